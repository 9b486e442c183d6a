// Microbenchmark: what one SIMD of gfx950 really sustains for the instruction mixes of the blend kernels.
//   hipcc --offload-arch=gfx950 -O3 scripts/mb_valu_peak.hip -o build/mb/mb_valu_peak && ./build/mb/mb_valu_peak
// Every kernel is launched with W waves per SIMD on every CU (W = 1, 2, 4, 8) and reports
//   cycles per wave-instruction per SIMD, from wall time at 2.4 GHz AND from s_memtime (the clock the chip held).
// Mixes: fma3 (3 distinct VGPR sources), fmac (VOP2, 2 sources + dst), mul2 (VOP2), add with an SGPR source,
// exp2 (transcendental), cndmask, and "blend": the forward's per-hit sequence on register operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
template <int MIX>
__global__ void __launch_bounds__(256) k_mix(float *out, unsigned long long *cyc, int iters, float a, float b, float sa) {
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if constexpr (MIX == 0) {        // v_fma_f32, 3 VGPR sources
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 1) { // v_fmac_f32 (VOP2): d += a * b
      asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                   "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 2) { // v_mul_f32 (VOP2), 2 VGPR sources
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 3) { // v_add_f32 with an SGPR source
      asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                   "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(sa), "v"(b));
    } else if constexpr (MIX == 4) { // v_exp_f32
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                   "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 5) { // v_fma_f32 with one SGPR source (VOP3, 2 VGPR + 1 SGPR)
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(sa), "v"(b));
    } else if constexpr (MIX == 6) { // 7 fma + 1 exp (the blend's transcendental density)
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %3, %3\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 7) { // v_cndmask_b32 (reads vcc) + v_cmp alternating
      asm volatile("v_cmp_le_f32 vcc, %8, %0\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_le_f32 vcc, %8, %2\n v_cndmask_b32 %3, %3, %9, vcc\n"
                   "v_cmp_le_f32 vcc, %8, %4\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_le_f32 vcc, %8, %6\n v_cndmask_b32 %7, %7, %9, vcc\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    } else if constexpr (MIX == 8) { // v_permlane32_swap (the 2DGS backward's first fold level)
      asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                   "v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 9) { // v_permlane16_swap
      asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                   "v_permlane16_swap_b32 %1, %2\n v_permlane16_swap_b32 %3, %4\n v_permlane16_swap_b32 %5, %6\n v_permlane16_swap_b32 %7, %0\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 10) { // v_add_f32 with a DPP source (row_ror:8), 8 independent chains
      asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 11) { // ds_bpermute_b32 (LDS crossbar) + the add that consumes it
      float t0_, t1_, t2_, t3_;
      const int addr = ((threadIdx.x ^ 32) & 63) << 2;
      asm volatile("ds_bpermute_b32 %8, %12, %0\n ds_bpermute_b32 %9, %12, %1\n ds_bpermute_b32 %10, %12, %2\n ds_bpermute_b32 %11, %12, %3\n"
                   "s_waitcnt lgkmcnt(0)\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %10\n v_add_f32 %7, %7, %11\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_)
                   : "v"(addr));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// The forward blend's per-hit sequence (gs3d_forward.hip blend_group), records broadcast from LDS exactly as there.
template <int GROUP>
__global__ void __launch_bounds__(64) k_blend(float *out, unsigned long long *cyc, int iters) {
  __shared__ float4 q_a[128], q_b[128];
  __shared__ float2 q_c[128];
  const int lane = threadIdx.x;
  for (int i = lane; i < 128; i += 64) {
    q_a[i] = make_float4(3.0f + (i & 7), 2.0f + (i >> 3 & 7), -0.05f - 0.001f * i, 0.01f);
    q_b[i] = make_float4(-0.04f, -1.0f - 0.01f * (i & 15), 0.5f, 0.25f);
    q_c[i] = make_float2(0.75f, 5.0f + i);
  }
  __syncthreads();
  const float pxf = (float)(lane & 7), pyf = (float)(lane >> 3);
  float T = 1.0f, C0 = 0, C1 = 0, C2 = 0, Dp = 0;
  uint32_t last = 0;
  int head = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    int hv = head;
    asm volatile("" : "+v"(hv));
    const float4 *ga = q_a + hv, *gb = q_b + hv;
    const float2 *gc = q_c + hv;
    float al[GROUP];
#pragma unroll
    for (int i = 0; i < GROUP; i++) {
      const float4 qa = ga[i];
      const float2 co = *reinterpret_cast<const float2 *>(&gb[i]);
      const float dx = qa.x - pxf, dy = qa.y - pyf;
      const float e = co.y + qa.z * dx * dx + co.x * dy * dy + qa.w * dx * dy;
      const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(e));
      al[i] = ((e <= co.y) & (alpha >= 1.0f / 255.0f)) ? alpha : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < GROUP; i++) {
      const float2 rg = *reinterpret_cast<const float2 *>(&gb[i].z);
      const float2 bz = gc[i];
      const float alpha = al[i];
      const float test_T = T * (1.0f - alpha);
      const bool ok = test_T >= 1e-4f;
      const float ae = ok ? alpha : 0.0f;
      const float w = ae * T;
      C0 += rg.x * w; C1 += rg.y * w; C2 += bz.x * w; Dp += bz.y * w;
      T = ok ? test_T : -fabsf(T);
      last = ae > 0.0f ? (uint32_t)(it * GROUP + i) : last;
    }
    head = (head + GROUP) & 127;
    if ((it & 15) == 15) T = 1.0f;   // keep pixels alive so the arithmetic stays representative
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = C0 + C1 + C2 + Dp + T + (float)last;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

static double median_cycles(unsigned long long *d_cyc, int n) {
  std::vector<unsigned long long> h(n);
  hipMemcpy(h.data(), d_cyc, n * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  return (double)h[n / 2];
}

template <int MIX>
void run_mix(const char *name, float *d, unsigned long long *c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;   // 256-thread blocks: 4 waves, one per SIMD
    k_mix<MIX><<<blocks, 256>>>(d, c, 100, 1.0001f, 0.5f, 0.25f);
    hipEventRecord(e0); k_mix<MIX><<<blocks, 256>>>(d, c, iters, 1.0001f, 0.5f, 0.25f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = 8.0 * iters * wps;
    const double cyc_wall = ms * 1e-3 * 2.4e9 / inst_per_simd;
    const double mc = median_cycles(c, blocks);   // s_memtime ticks at 100 MHz on gfx9? reported raw
    printf("%-10s waves/SIMD %d: %.3f ms  %.2f cyc/inst/SIMD @2.4GHz wall   memtime ticks/inst %.3f\n", name, wps, ms, cyc_wall, mc / inst_per_simd * wps);
  }
}

template <int GROUP>
void run_blend(float *d, unsigned long long *c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
    const int blocks = 1024 * wps;
    const int iters = 4096 / GROUP * 8;
    k_blend<GROUP><<<blocks, 64>>>(d, c, 16);
    hipEventRecord(e0); k_blend<GROUP><<<blocks, 64>>>(d, c, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double hits_per_simd = (double)iters * GROUP * wps;
    printf("blend G=%-2d waves/SIMD %d: %.3f ms  %.1f cyc/hit/SIMD @2.4GHz wall   memtime ticks/hit/wave %.2f\n", GROUP, wps, ms,
           ms * 1e-3 * 2.4e9 / hits_per_simd, median_cycles(c, blocks) / ((double)iters * GROUP));
  }
}

int main() {
  float *d; hipMalloc(&d, 1024 * 256 * 8 * 4);
  unsigned long long *c; hipMalloc(&c, 8192 * 8 * 2);
  run_mix<0>("fma3", d, c);
  run_mix<1>("fmac", d, c);
  run_mix<2>("mul2", d, c);
  run_mix<3>("add_sgpr", d, c);
  run_mix<5>("fma_sgpr", d, c);
  run_mix<4>("exp2", d, c);
  run_mix<6>("7fma+exp", d, c);
  run_mix<7>("cmp+cnd", d, c);
  run_mix<8>("perml32sw", d, c);
  run_mix<9>("perml16sw", d, c);
  run_mix<10>("add_dpp", d, c);
  run_mix<11>("bperm+add", d, c);   // 8 instructions per iteration: 4 ds_bpermute + 4 v_add
  run_blend<8>(d, c);
  run_blend<16>(d, c);
  return 0;
}
