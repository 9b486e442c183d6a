// Microbenchmark (round 5): packed fp32 arithmetic of gfx950 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 operations
// per lane per instruction on 64-bit register pairs) against the scalar forms, priced the way scripts/mb_issue_costs.hip
// prices everything else: cycles at 2.4 GHz of wall time per INSTRUCTION per SIMD with W waves sharing the SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/mb_pk_f32.hip -o build/mb/mb_pk_f32 && ./build/mb/mb_pk_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MIX>
__global__ void __launch_bounds__(512) k(float *out, int iters, float a, float b) {
  f32x2 x0 = {threadIdx.x * 1e-3f + 1.0f, 2.0f}, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f,
        x6 = x0 + 6.0f, x7 = x0 + 7.0f;
  const f32x2 aa = {a, a * 1.0001f}, bb = {b, b * 0.999f};
  for (int i = 0; i < iters; i++) {
    if constexpr (MIX == 0) {          // v_fma_f32 on the low halves (reference)
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0.x), "+v"(x1.x), "+v"(x2.x), "+v"(x3.x), "+v"(x4.x), "+v"(x5.x), "+v"(x6.x), "+v"(x7.x) : "v"(a), "v"(b));
    } else if constexpr (MIX == 1) {   // v_pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa), "v"(bb));
    } else if constexpr (MIX == 2) {   // v_pk_fma_f32, the multiplier's LOW half for both lanes of the pair (op_sel_hi: a broadcast without a move)
      asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel_hi:[1,0,1]\n"
                   "v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel_hi:[1,0,1]\n"
                   "v_pk_fma_f32 %4, %4, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel_hi:[1,0,1]\n"
                   "v_pk_fma_f32 %6, %6, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel_hi:[1,0,1]\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa), "v"(bb));
    } else if constexpr (MIX == 3) {   // v_pk_mul_f32
      asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa), "v"(bb));
    } else if constexpr (MIX == 4) {   // v_mul_f32 (VOP2 reference)
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(x0.x), "+v"(x1.x), "+v"(x2.x), "+v"(x3.x), "+v"(x4.x), "+v"(x5.x), "+v"(x6.x), "+v"(x7.x) : "v"(a), "v"(b));
    } else if constexpr (MIX == 5) {   // v_pk_add_f32
      asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                   "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa), "v"(bb));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0.x + x1.x + x2.x + x3.x + x4.x + x5.x + x6.x + x7.x + x0.y + x1.y + x2.y + x3.y + x4.y + x5.y + x6.y + x7.y;
}

template <int MIX>
static void run(const char *name, float *d_out, int cus) {
  const int iters = 4000;
  for (int W : {2, 4, 6, 8}) {
    const int threads = 64 * 4 * W > 512 ? 512 : 64 * 4 * W;
    const int wgs_per_cu = (64 * 4 * W) / threads;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MIX><<<cus * wgs_per_cu, threads>>>(d_out, 50, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MIX><<<cus * wgs_per_cu, threads>>>(d_out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s waves/SIMD %d: %7.3f ms  %6.2f cycles/instruction/SIMD @2.4GHz wall\n", name, W, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * W));
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float *d_out;
  hipMalloc(&d_out, (size_t)cus * 8 * 512 * 4);
  run<0>("v_fma_f32", d_out, cus);
  run<1>("v_pk_fma_f32 (two per lane)", d_out, cus);
  run<2>("v_pk_fma_f32, multiplier broadcast by op_sel", d_out, cus);
  run<4>("v_mul_f32", d_out, cus);
  run<3>("v_pk_mul_f32", d_out, cus);
  run<5>("v_pk_add_f32", d_out, cus);
  return 0;
}
