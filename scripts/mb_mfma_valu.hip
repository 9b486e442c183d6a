// Microbenchmark: do fp32 MFMAs and plain VALU FMAs of DIFFERENT waves on one SIMD overlap?
//   hipcc --offload-arch=gfx950 -O3 scripts/mb_mfma_valu.hip -o build/mb_mfma_valu && ./build/mb_mfma_valu
// Three launches with the same grid (8 waves per SIMD): every wave runs FMAs; every wave runs
// v_mfma_f32_16x16x4_f32; even waves FMAs + odd waves MFMAs (half the work of each).  If the two pipes overlap the
// mixed launch takes about max(t_fma, t_mfma) / 2, if they share the issue slot about (t_fma + t_mfma) / 2.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: all FMA, 1: all MFMA, 2: mixed by wave parity
__global__ void k(float *out, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = MODE == 1 || (MODE == 2 && (wave & 1));
  float r = 0.0f;
  if (do_mfma) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int i = 0; i < iters; i++) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    r = a0.x + a1.y + a2.z + a3.w;
  } else {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float a = 1.0001f, b = 0.5f;
    for (int i = 0; i < 4 * iters; i++) {   // 32 FMAs per MFMA-loop iteration: comparable durations
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
float run(float *d, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<256 * 4, 512>>>(d, iters);   // 4 workgroups of 8 waves per CU: 8 waves per SIMD
  (void)hipEventRecord(e0);
  k<MODE><<<256 * 4, 512>>>(d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float *d; (void)hipMalloc(&d, 256 * 4 * 512 * 4);
  const int iters = 4000;
  const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters);
  printf("all FMA %.3f ms, all MFMA %.3f ms, mixed (half of each) %.3f ms; overlap would give %.3f, no overlap %.3f\n",
         t0, t1, t2, (t0 > t1 ? t0 : t1) / 2, (t0 + t1) / 2);
  return 0;
}
