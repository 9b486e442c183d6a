// Microbenchmark: throughput of v_fma_f32 against v_pk_fma_f32 on independent register chains.
//   hipcc --offload-arch=gfx950 -O3 scripts/mb_pk_fma.hip -o build/mb_pk_fma && ./build/mb_pk_fma
// MI355X, round 1: 100.6 TFLOP/s scalar, 107.6 TFLOP/s packed - packed fp32 buys 7 %, not 2x, which is why the blend
// kernels are written with scalar FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k_scalar(float *out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_packed(float *out, int iters, float a, float b) {
  v2f x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  v2f av = {a, a}, bv = {b, b};
  for (int i = 0; i < iters; i++) {
    asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(av), "v"(bv));
  }
  v2f s = x0 + x1 + x2 + x3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main() {
  float *d; hipMalloc(&d, 1024 * 256 * 8 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * 8;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0); k_scalar<<<blocks, 256>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 8 * iters * (double)blocks * 256;
    printf("scalar fma: %.3f ms  %.1f TFLOP/s\n", ms, fl / ms * 1e-9);
    hipEventRecord(e0); k_packed<<<blocks, 256>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("pk fma    : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms * 1e-9);
  }
  return 0;
}
