// Microbenchmark (round 5): what ONE INSTRUCTION of each class the blend kernels use costs a SIMD of gfx950 when W waves
// share it - vector, scalar and LDS alike.  profiles/DESIGN_history_r01-r05.md section 8.0: the blend kernels are bound by instructions issued, so the
// price list that matters is per instruction CLASS, not per FLOP.
//   hipcc --offload-arch=gfx950 -O3 scripts/mb_issue_costs.hip -o build/mb/mb_issue_costs && ./build/mb/mb_issue_costs
// Every kernel runs W = 2, 4, 6, 8 waves per SIMD on every CU; eight independent register chains per wave; reported:
// cycles at 2.4 GHz of wall time per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MIX>
__global__ void __launch_bounds__(512) k(float *out, int iters, float a, float b, uint64_t sm) {
  __shared__ float4 lds[256];
  float x0 = threadIdx.x * 1e-3f + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  if (threadIdx.x < 256) lds[threadIdx.x] = make_float4(x0, x1, x2, x3);
  __syncthreads();
  uint32_t sacc = 0;
  f32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; i++) {
    if constexpr (MIX == 0) {          // v_fma_f32 (reference)
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 1) {   // v_cvt_pkrtz_f16_f32
      asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %8\n v_cvt_pkrtz_f16_f32 %1, %1, %8\n v_cvt_pkrtz_f16_f32 %2, %2, %8\n v_cvt_pkrtz_f16_f32 %3, %3, %8\n"
                   "v_cvt_pkrtz_f16_f32 %4, %4, %8\n v_cvt_pkrtz_f16_f32 %5, %5, %8\n v_cvt_pkrtz_f16_f32 %6, %6, %8\n v_cvt_pkrtz_f16_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 2) {   // v_fma_mix_f32 with an fp16 source (the split's remainder)
      asm volatile("v_fma_mix_f32 %0, %8, -1.0, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, -1.0, %1 op_sel_hi:[1,0,0]\n"
                   "v_fma_mix_f32 %2, %8, -1.0, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, -1.0, %3 op_sel_hi:[1,0,0]\n"
                   "v_fma_mix_f32 %4, %8, -1.0, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, -1.0, %5 op_sel_hi:[1,0,0]\n"
                   "v_fma_mix_f32 %6, %8, -1.0, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, -1.0, %7 op_sel_hi:[1,0,0]\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 3) {   // v_perm_b32
      asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                   "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 4) {   // v_and_b32 (VOP2 integer)
      asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                   "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 5) {   // v_rcp_f32
      asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if constexpr (MIX == 6) {   // v_cndmask_b32 with an SGPR-pair mask (no compare)
      asm volatile("v_cndmask_b32_e64 %0, %0, %8, %10\n v_cndmask_b32_e64 %1, %1, %8, %10\n v_cndmask_b32_e64 %2, %2, %8, %10\n v_cndmask_b32_e64 %3, %3, %8, %10\n"
                   "v_cndmask_b32_e64 %4, %4, %8, %10\n v_cndmask_b32_e64 %5, %5, %8, %10\n v_cndmask_b32_e64 %6, %6, %8, %10\n v_cndmask_b32_e64 %7, %7, %8, %10\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(sm));
    } else if constexpr (MIX == 7) {   // 8 x (v_fma_f32 + one scalar instruction): what a scalar instruction costs NEXT TO vector work
      asm volatile("v_fma_f32 %0, %0, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %2, %2, %9, %10\n s_add_u32 %8, %8, 1\n"
                   "v_fma_f32 %3, %3, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %4, %4, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %9, %10\n s_add_u32 %8, %8, 1\n"
                   "v_fma_f32 %6, %6, %9, %10\n s_add_u32 %8, %8, 1\n v_fma_f32 %7, %7, %9, %10\n s_add_u32 %8, %8, 1\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+s"(sacc) : "v"(a), "v"(b) : "scc");
    } else if constexpr (MIX == 8) {   // scalar instructions alone
      asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                   : "+s"(sacc) : : "scc");
    } else if constexpr (MIX == 9) {   // ds_read_b128, every lane the same address (the ring's broadcast read) + the wait
      float4 r0, r1, r2, r3;
      const uint32_t ad = (uint32_t)((i & 63) * 16);
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(ad) : "memory");
      x0 += r0.x; x1 += r1.y; x2 += r2.z; x3 += r3.w;   // (4 reads + 4 adds + 1 wait per iteration)
    } else if constexpr (MIX == 10) {  // ds_write_b32, lane-consecutive
      const uint32_t ad = (uint32_t)((threadIdx.x & 63) * 4);
      asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:256\n ds_write_b32 %0, %3 offset:512\n ds_write_b32 %0, %4 offset:768\n"
                   "ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %2 offset:1280\n ds_write_b32 %0, %3 offset:1536\n ds_write_b32 %0, %4 offset:1792\n"
                   : : "v"(ad), "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "memory");
    } else if constexpr (MIX == 11) {  // v_mfma_f32_16x16x4_f32 (four per iteration, one chain)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, x3, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x4, x5, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x6, x7, acc, 0, 0, 0);
    } else if constexpr (MIX == 12) {  // v_mfma_f32_16x16x32_f16 (four per iteration, one chain)
      f16x8 af, bf;
      for (int j = 0; j < 8; j++) { af[j] = (_Float16)x0; bf[j] = (_Float16)x1; }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc, 0, 0, 0);
    } else if constexpr (MIX == 13) {  // 4 x v_mfma_f32_16x16x4_f32 interleaved with 8 v_fma_f32: do they share the datapath?
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, acc, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n" : "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(a), "v"(b));
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, acc, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n" : "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, acc, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n" : "+v"(x6), "+v"(x7), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, acc, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n" : "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5) : "v"(a), "v"(b));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + acc[0] + acc[1] + acc[2] + acc[3] + (float)sacc;
}

template <int MIX>
static void run(const char *name, int per_iter, float *d_out, int cus) {
  const int iters = 4000;
  for (int W : {2, 4, 6, 8}) {
    const int threads = 64 * 4 * W > 512 ? 512 : 64 * 4 * W;          // waves of one workgroup spread over the CU's four SIMDs
    const int wgs_per_cu = (64 * 4 * W) / threads;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MIX><<<cus * wgs_per_cu, threads>>>(d_out, 50, 1.0001f, 0.5f, 0x5555555555555555ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MIX><<<cus * wgs_per_cu, threads>>>(d_out, iters, 1.0001f, 0.5f, 0x5555555555555555ull);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * per_iter * W;
    printf("%-34s waves/SIMD %d: %7.3f ms  %6.2f cycles/instruction/SIMD @2.4GHz wall\n", name, W, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float *d_out;
  hipMalloc(&d_out, (size_t)cus * 8 * 512 * 4);
  run<0>("v_fma_f32", 8, d_out, cus);
  run<1>("v_cvt_pkrtz_f16_f32", 8, d_out, cus);
  run<2>("v_fma_mix_f32 (f16 source)", 8, d_out, cus);
  run<3>("v_perm_b32", 8, d_out, cus);
  run<4>("v_and_b32", 8, d_out, cus);
  run<5>("v_rcp_f32", 8, d_out, cus);
  run<6>("v_cndmask_b32 (SGPR mask)", 8, d_out, cus);
  run<7>("v_fma_f32 + s_add_u32 (per pair)", 8, d_out, cus);
  run<8>("s_add_u32", 8, d_out, cus);
  run<9>("ds_read_b128 bcast x4+4 add+wait", 1, d_out, cus);
  run<10>("ds_write_b32", 8, d_out, cus);
  run<11>("v_mfma_f32_16x16x4_f32", 4, d_out, cus);
  run<12>("v_mfma_f32_16x16x32_f16", 4, d_out, cus);
  run<13>("4 x (mfma f32 + 2 v_fma_f32) (per group of 3)", 4, d_out, cus);
  return 0;
}
