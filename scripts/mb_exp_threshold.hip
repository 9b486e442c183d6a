// Probe: is  exp2(e) >= 1/255  (v_exp_f32, the blend kernels' `alpha >= 1/255` test)  the same predicate as  e >= E0  for ONE
// float E0?  If v_exp_f32 is non-decreasing across the threshold the two are the same bits for every input, and the blend
// kernels may decide "live" from the exponent itself, before (and without) the v_exp.
//   hipcc --offload-arch=gfx950 -O3 scripts/mb_exp_threshold.hip -o build/mb/mb_exp_threshold && ./build/mb/mb_exp_threshold
// Scans EVERY float in [-9, -7] (and, for the record, every float in [-126, 0]) and prints the step's position, or the
// first violation of monotonicity.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
__global__ void scan(uint32_t lo_bits, uint32_t n, float thr, unsigned long long *out) {
  // negative floats: larger bit pattern = smaller value.  value index k: bits = lo_bits - k, k = 0 .. n-1 ascends in value
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  const float e = __uint_as_float(lo_bits - (k < n ? k : n - 1));
  const bool live = __builtin_amdgcn_exp2f(e) >= thr;
  unsigned long long a = live ? (unsigned long long)k : ~0ull, b = live ? 0ull : (unsigned long long)k + 1;
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long a2 = __shfl_xor(a, off, 64), b2 = __shfl_xor(b, off, 64);
    a = a2 < a ? a2 : a; b = b2 > b ? b2 : b;
  }
  if ((threadIdx.x & 63) == 0) {
    if (a != ~0ull) atomicMin(&out[0], a);        // first live index
    if (b != 0ull) atomicMax(&out[1], b);         // 1 + last dead index
  }
}
int main() {
  const float thr = 1.0f / 255.0f;
  unsigned long long *d, h[2];
  hipMalloc(&d, 16);
  const float ranges[2][2] = {{-9.0f, -7.0f}, {-126.0f, -0.0f}};
  for (auto &r : ranges) {
    uint32_t lo, hi;
    memcpy(&lo, &r[0], 4); memcpy(&hi, &r[1], 4);
    const uint32_t n = lo - hi + 1;
    h[0] = ~0ull; h[1] = 0;
    hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    scan<<<(n + 255) / 256, 256>>>(lo, n, thr, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const uint32_t first_live = lo - (uint32_t)h[0];
    float e0; memcpy(&e0, &first_live, 4);
    printf("range [%g, %g]: %u floats; first live index %llu, 1 + last dead index %llu -> %s; E0 = %.9g (bits 0x%08x)\n", r[0], r[1], n,
           h[0], h[1], h[0] == h[1] ? "ONE STEP (monotone across the threshold)" : "NOT a step", e0, first_live);
  }
  return 0;
}
