// mb_wg_barrier.hip — what a tile-level decomposition of the blend kernels pays for keeping the four waves of a tile in
// ONE 256-thread workgroup: the barrier itself and, mostly, the waiting of three waves for the slowest one between two
// barriers.  Today every 8x8 block is its own one-wave workgroup and never waits.
//
// Work model: wave q of tile t runs hits[t * 4 + q] iterations of a loop body shaped like the blend forward's per-hit
// sequence (13 VALU incl. one v_exp, one LDS read).  The tile-level kernel cuts the tile's list into chunks of 256
// entries (n_chunks = ceil(list / 256), list = sum of the four blocks' hits / 1.85: the measured hits per list entry)
// and puts `barriers_per_chunk` __syncthreads() behind each chunk's share of every wave's iterations.
// hits[] = the block_hits of a real view if a file is given (scripts/dev/dump_block_hits.py writes it), else a synthetic
// draw with the same mean.   Occupancy is held at the forward's six waves per SIMD by the LDS footprint.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mb_wg_barrier.hip -o build/mb/mb_wg_barrier && ./build/mb/mb_wg_barrier [block_hits.bin]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kLdsPerWave = 6192 / 4;   // floats: the blend forward's LDS footprint per wave

__device__ __forceinline__ float body(float T, float e, float c, float &C0, float &C1, float &C2, float &D) {
  const float g = __builtin_amdgcn_exp2f(e);
  const bool live = g >= 0.0039f;
  const float al = live ? g : 0.0f;
  const float tt = __builtin_fmaf(-al, T, T);
  const bool ok = tt >= 1e-4f;
  const float ae = ok ? al : 0.0f;
  const float w = ae * T;
  C0 += c * w; C1 += (c + 1.0f) * w; C2 += (c - 1.0f) * w; D += e * w;
  return ok ? tt : -__builtin_fabsf(T);
}

template <int WAVES>   // waves per workgroup: 1 (today) or 4 (tile-level)
__global__ void __launch_bounds__(64 * WAVES, 2) k_blend(const int *__restrict__ hits, int tiles, int barriers_per_chunk, float *out) {
  __shared__ float lds[kLdsPerWave * WAVES];
  // (wave-uniform values are forced into SGPRs: left in VGPRs, the loop counter and the LDS addresses of the four-wave
  // form cost 1.6x the VALU instructions of the one-wave form - the first version of this file measured exactly that)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int tile, quad;
  if (WAVES == 4) { const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3; tile = kk * 8 + xcd; quad = wv; }
  else { const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3; tile = (kk >> 2) * 8 + xcd; quad = kk & 3; }
  if (tile >= tiles) return;
  float *my = lds + wv * kLdsPerWave;
  for (int i = lane; i < 96 * 4; i += 64) my[i] = (float)(i & 15) * -0.37f;
  const int h = __builtin_amdgcn_readfirstlane(hits[tile * 4 + quad]);
  int n_chunks = 1;
  if (WAVES == 4) {
    const int total = hits[tile * 4] + hits[tile * 4 + 1] + hits[tile * 4 + 2] + hits[tile * 4 + 3];
    n_chunks = __builtin_amdgcn_readfirstlane(max(1, ((int)(total / 1.85f) + 255) / 256));
    __syncthreads();
  }
  float T = 1.0f, C0 = 0, C1 = 0, C2 = 0, D = 0;
  int done = 0;
  for (int c = 0; c < n_chunks; c++) {
    const int upto = (int)((long long)h * (c + 1) / n_chunks);
    for (; done + 16 <= upto; done += 16) {
#pragma unroll
      for (int i = 0; i < 16; i++) T = body(T, my[((done + i) % 96) * 4 + 1] - lane * 0.01f, my[((done + i) % 96) * 4], C0, C1, C2, D);
    }
    for (; done < upto; done++) T = body(T, my[(done % 96) * 4 + 1] - lane * 0.01f, my[(done % 96) * 4], C0, C1, C2, D);
    if (WAVES == 4)
      for (int b = 0; b < barriers_per_chunk; b++) __syncthreads();
  }
  if (T + C0 + C1 + C2 + D == 12345.678f) out[0] = T;
}

template <typename F>
static float time_ms(F launch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  return best;
}

int main(int argc, char **argv) {
  const int tiles = 7500;
  std::vector<int> hits(tiles * 4);
  bool real = false;
  if (argc > 1) {
    FILE *f = fopen(argv[1], "rb");
    if (f) { real = fread(hits.data(), 4, hits.size(), f) == hits.size(); fclose(f); }
  }
  if (!real) {   // synthetic: mean 147 hits per block (4.42 M / 30 000), log-normal-ish spread, tile-correlated
    srand(7);
    for (int t = 0; t < tiles; t++) {
      const float tile_scale = 0.4f + 1.2f * (rand() / (float)RAND_MAX);
      for (int q = 0; q < 4; q++) hits[t * 4 + q] = (int)(147.0f * tile_scale * (0.5f + (rand() / (float)RAND_MAX)));
    }
  }
  long long total = 0, waited = 0;
  for (int t = 0; t < tiles; t++) {
    int mx = 0;
    for (int q = 0; q < 4; q++) { total += hits[t * 4 + q]; mx = hits[t * 4 + q] > mx ? hits[t * 4 + q] : mx; }
    waited += 4LL * mx;
  }
  printf("# block hits: %s, %d tiles, %.2f M (block, hit) iterations; 4 x max over a tile's blocks / sum = %.3f (the work a tile-level\n"
         "# workgroup holds wave slots for, relative to the work it does)\n", real ? argv[1] : "synthetic", tiles, total / 1e6, (double)waited / total);
  int *d_hits; float *d_out;
  hipMalloc(&d_hits, hits.size() * 4); hipMalloc(&d_out, 64);
  hipMemcpy(d_hits, hits.data(), hits.size() * 4, hipMemcpyHostToDevice);
  const int wgs1 = ((tiles + 7) / 8) * 8 * 4, wgs4 = ((tiles + 7) / 8) * 8;
  const float t1 = time_ms([&]() { k_blend<1><<<wgs1, 64>>>(d_hits, tiles, 0, d_out); }, 7);
  printf("%-52s %10.1f us\n", "one wave per 8x8 block (64-thread workgroups)", t1 * 1e3);
  for (int b = 0; b <= 3; b++) {
    const float t4 = time_ms([&]() { k_blend<4><<<wgs4, 256>>>(d_hits, tiles, b, d_out); }, 7);
    printf("one 256-thread workgroup per tile, %d barrier(s)/chunk %10.1f us   (x %.3f)\n", b, t4 * 1e3, t4 / t1);
  }
  return 0;
}
