// Microbenchmark: can the blend kernels take a hit's 48-byte record through the SCALAR cache (s_load into SGPRs, used as
// scalar operands) instead of broadcasting it from an LDS ring?  Same per-hit arithmetic in both kernels:
//   k_scalar: ids by s_load_dwordx4, records by s_load_dwordx8 + s_load_dwordx4, software-pipelined BATCH hits deep;
//   k_lds   : the forward's structure - 64 lanes gather 64 records, stage them in LDS, every hit is read back by all lanes.
// 30016 waves of one 8x8 pixel block each, 160 hits per wave, ids random with the locality of a tile (the four waves of a
// tile draw from the same 512 ids), records in a 1M x 48 B table.
//   hipcc --offload-arch=gfx950 -O3 scripts/mb_scalar_ring.hip -o build/mb/mb_scalar_ring && ./build/mb/mb_scalar_ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

typedef float f8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const f8 CF8;
typedef __attribute__((address_space(4))) const f4 CF4;
typedef __attribute__((address_space(4))) const u4 CU4;

struct Acc { float T, C0, C1, C2, Dp; uint32_t last; };

__device__ __forceinline__ void hit(Acc &s, float pxf, float pyf, float x, float y, float A, float B, float C, float L, float r,
                                    float g, float b, float z, uint32_t ord) {
  const float dx = x - pxf, dy = y - pyf;
  float e = A * (dx * dx);
  e = __builtin_fmaf(C, dy * dy, e);
  e = __builtin_fmaf(B, dx * dy, e);
  e = e + L;
  float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(e));
  alpha = ((e <= L) & (alpha >= 1.0f / 255.0f)) ? alpha : 0.0f;
  const float test_T = s.T * (1.0f - alpha);
  const bool ok = test_T >= 1e-4f;
  const float ae = ok ? alpha : 0.0f;
  const float w = ae * s.T;
  s.C0 = __builtin_fmaf(r, w, s.C0); s.C1 = __builtin_fmaf(g, w, s.C1); s.C2 = __builtin_fmaf(b, w, s.C2); s.Dp = __builtin_fmaf(z, w, s.Dp);
  s.T = ok ? test_T : -fabsf(s.T);
  s.last = ae > 0.0f ? ord : s.last;
}

template <int BATCH>
__global__ void __launch_bounds__(64) k_scalar(const float *__restrict__ rec, const uint32_t *__restrict__ ids, int nhits, float *out) {
  const int lane = threadIdx.x;
  const float pxf = (float)(lane & 7), pyf = (float)(lane >> 3);
  Acc s = {1.0f, 0, 0, 0, 0, 0u};
  const uint32_t *my = ids + (size_t)blockIdx.x * nhits;
  f8 ra[BATCH], na[BATCH];
  f4 rb[BATCH], nb[BATCH];
  auto load_batch = [&](int h0, f8 *a, f4 *b) {
#pragma unroll
    for (int j = 0; j < BATCH; j += 4) {
      const u4 id4 = *(CU4 *)(my + h0 + j);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const char *p = (const char *)rec + (size_t)id4[q] * 48;
        a[j + q] = *(CF8 *)p;
        b[j + q] = *(CF4 *)(p + 32);
      }
    }
  };
  load_batch(0, ra, rb);
  for (int h = 0; h < nhits; h += BATCH) {
    const int hn = h + BATCH < nhits ? h + BATCH : h;   // (re-reads the last batch at the end: harmless)
    load_batch(hn, na, nb);
#pragma unroll
    for (int j = 0; j < BATCH; j++)
      hit(s, pxf, pyf, ra[j][0], ra[j][1], ra[j][2], ra[j][3], ra[j][4], ra[j][5], ra[j][6], ra[j][7], rb[j][0], rb[j][1], h + j + 1);
#pragma unroll
    for (int j = 0; j < BATCH; j++) { ra[j] = na[j]; rb[j] = nb[j]; }
    if ((h & 63) == 0) s.T = 1.0f;
  }
  out[blockIdx.x * 64 + lane] = s.C0 + s.C1 + s.C2 + s.Dp + s.T + (float)s.last;
}

__global__ void __launch_bounds__(64) k_lds(const float *__restrict__ rec, const uint32_t *__restrict__ ids, int nhits, float *out) {
  __shared__ float4 q_a[128], q_b[128], q_c[128];
  const int lane = threadIdx.x;
  const float pxf = (float)(lane & 7), pyf = (float)(lane >> 3);
  Acc s = {1.0f, 0, 0, 0, 0, 0u};
  const uint32_t *my = ids + (size_t)blockIdx.x * nhits;
  auto fetch_id = [&](int bs) { return bs + lane < nhits ? my[bs + lane] : 0xFFFFFFFFu; };
  float4 a, b, c;
  uint32_t id0 = fetch_id(0);
  auto fetch_rec = [&](uint32_t id, float4 &a_, float4 &b_, float4 &c_) {
    if (id != 0xFFFFFFFFu) { const float4 *src = reinterpret_cast<const float4 *>(rec + (size_t)id * 12); a_ = src[0]; b_ = src[1]; c_ = src[2]; }
  };
  fetch_rec(id0, a, b, c);
  uint32_t id1 = fetch_id(64);
  int head = 0;
  for (int base = 0; base < nhits; base += 64) {
    float4 a1, b1, c1;
    fetch_rec(id1, a1, b1, c1);
    const uint32_t id2 = fetch_id(base + 128);
    const int n = nhits - base < 64 ? nhits - base : 64;
    if (id0 != 0xFFFFFFFFu) { q_a[(head + lane) & 127] = a; q_b[(head + lane) & 127] = b; q_c[(head + lane) & 127] = c; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int g = 0; g < n; g += 8) {
      int hv = (head + g) & 127;
      asm volatile("" : "+v"(hv));
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float4 qa = q_a[hv + i], qb = q_b[hv + i];
        const float2 qc = *reinterpret_cast<const float2 *>(&q_c[hv + i]);
        hit(s, pxf, pyf, qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w, qc.x, qc.y, base + g + i + 1);
      }
    }
    head = (head + 64) & 127;
    id0 = id1; a = a1; b = b1; c = c1; id1 = id2;
    s.T = 1.0f;
  }
  out[blockIdx.x * 64 + lane] = s.C0 + s.C1 + s.C2 + s.Dp + s.T + (float)s.last;
}

int main() {
  const int N = 1000000, waves = 30016, nhits = 160;
  std::vector<float> h_rec((size_t)N * 12);
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  for (int i = 0; i < N; i++) {
    float *r = &h_rec[(size_t)i * 12];
    r[0] = 8 * U(rng); r[1] = 8 * U(rng); r[2] = -0.05f - 0.1f * U(rng); r[3] = 0.02f * (U(rng) - 0.5f); r[4] = -0.05f - 0.1f * U(rng);
    r[5] = -2.0f * U(rng); r[6] = U(rng); r[7] = U(rng); r[8] = U(rng); r[9] = 3 + U(rng); r[10] = 0; r[11] = 0;
  }
  std::vector<uint32_t> h_ids((size_t)waves * nhits);
  for (int t = 0; t < waves / 4; t++) {           // a tile: 512 splats drawn at random from the table, shared by its four waves
    std::vector<uint32_t> pool(512);
    for (auto &p : pool) p = rng() % N;
    for (int q = 0; q < 4; q++)
      for (int k = 0; k < nhits; k++) h_ids[((size_t)t * 4 + q) * nhits + k] = pool[rng() % 512];
  }
  float *d_rec, *d_out; uint32_t *d_ids;
  hipMalloc(&d_rec, h_rec.size() * 4); hipMalloc(&d_ids, h_ids.size() * 4); hipMalloc(&d_out, (size_t)waves * 64 * 4);
  hipMemcpy(d_rec, h_rec.data(), h_rec.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_ids, h_ids.data(), h_ids.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time_it = [&](const char *name, auto launch) {
    launch();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    std::vector<float> h(64);
    hipMemcpy(h.data(), d_out, 256, hipMemcpyDeviceToHost);
    printf("%-14s %.1f us  (%.1f cyc/hit/SIMD @2.4GHz)  check %.5f\n", name, best * 1e3, best * 1e-3 * 2.4e9 / ((double)waves * nhits / 1024), h[5]);
  };
  time_it("lds ring", [&] { k_lds<<<waves, 64>>>(d_rec, d_ids, nhits, d_out); });
  time_it("scalar B=4", [&] { k_scalar<4><<<waves, 64>>>(d_rec, d_ids, nhits, d_out); });
  time_it("scalar B=8", [&] { k_scalar<8><<<waves, 64>>>(d_rec, d_ids, nhits, d_out); });
  return 0;
}
