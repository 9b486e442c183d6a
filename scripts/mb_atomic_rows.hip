// mb_atomic_rows.hip — what the memory side of gfx950 sustains for the blend backward's accumulator traffic: global float
// atomics shaped as WHOLE 64-byte rows, four rows per wave-instruction (lane = float of a row, 10 or 16 of 16 lanes
// active), at random rows of a 64 MB table (1 M Gaussians x 64 B).  MI355X_MICROARCH.md lists this shape as unmeasured;
// profiles/DESIGN_history_r01-r05.md section 8.1 extrapolated a floor of ~216 us for the 4.42 M (block, hit) rows of one S3 view from the 256-byte
// contiguous figure.  This measures it, together with what a tile-level redesign would change:
//   * rows per view 4.42 M (one per (8x8 block, hit)) against 2.43 M (one per (tile, splat));
//   * the four waves of a tile hitting the SAME rows at about the same time (today) against disjoint rows;
//   * plain 64-byte row stores and LDS float atomics (ds_add_f32 into a per-workgroup table) for comparison.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mb_atomic_rows.hip -o build/mb/mb_atomic_rows && ./build/mb/mb_atomic_rows
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

// MODE 0: float atomics, rows random per (wave, iteration, 16-lane group)
// MODE 1: the same rows for the four consecutive waves of a "tile" (wave >> 2 seeds the hash): contention as today
// MODE 2: plain stores of the rows
// MODE 3: float atomics, ONE row per instruction (16 lanes active, 48 idle)
// MODE 4: float atomics, rows drawn with the locality of a view: a tile's rows come from a window of 4096 rows that moves
//         with the tile (Gaussians of neighbouring tiles are not neighbours in memory, but one tile revisits its few hundred rows)
template <int MODE>
__global__ void __launch_bounds__(64) k_rows(float *acc, uint32_t n_rows, int iters, int active, int spacer) {
  const int lane = threadIdx.x;
  const uint32_t wave = blockIdx.x;
  const int grp = lane >> 4, col = lane & 15;
  float keep = 0.0f;
  for (int it = 0; it < iters; it++) {
    uint32_t row;
    if (MODE == 1) row = hash3(wave >> 2, it, grp) % n_rows;
    else if (MODE == 3) row = hash3(wave, it, 0) % n_rows;
    else if (MODE == 4) row = (hash3(wave >> 2, 0, 0) % (n_rows - 4096)) + hash3(wave >> 2, it, grp) % 4096u;
    else row = hash3(wave, it, grp) % n_rows;
    float *p = acc + (size_t)row * 16 + col;
    const bool on = MODE == 3 ? (grp == 0 && col < active) : col < active;
    if (on) {
      if (MODE == 2) __builtin_nontemporal_store(1.0f, p);
      else atomicAdd(p, 1.0f);
    }
    // `spacer` VALU instructions of dependent work between the atomics: the real kernel issues one group of four
    // atomic instructions per ~330 VALU instructions
    for (int s = 0; s < spacer; s++) keep = __builtin_fmaf(keep, 1.0001f, 0.5f);
  }
  if (keep == 123.456f) acc[0] = keep;
}

// LDS float atomics: a 256-thread workgroup adds rows into a table of `slots` rows of 16 floats in LDS (the tile-level
// table of the redesign), then flushes the table with ONE global atomic row per slot.
__global__ void __launch_bounds__(256) k_lds_table(float *acc, uint32_t n_rows, int iters, int slots) {
  extern __shared__ float tab[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, col = lane & 15;
  for (int i = threadIdx.x; i < slots * 16; i += 256) tab[i] = 0.0f;
  __syncthreads();
  for (int it = 0; it < iters; it++) {
    const uint32_t slot = hash3(blockIdx.x * 4 + wave, it, grp) % (uint32_t)slots;
    if (col < 10) atomicAdd(&tab[slot * 16 + col], 1.0f);   // ds_add_f32
  }
  __syncthreads();
  for (int s = threadIdx.x >> 4; s < slots; s += 16) {
    const uint32_t row = hash3(blockIdx.x, s, 77) % n_rows;
    if (col < 10) atomicAdd(acc + (size_t)row * 16 + col, tab[s * 16 + col]);
  }
}

template <typename F>
static float time_ms(F launch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  hipEventDestroy(a); hipEventDestroy(b);
  return best;
}

int main() {
  const uint32_t n_rows = 1000000;
  float *acc = nullptr;
  CHECK(hipMalloc(&acc, (size_t)n_rows * 64));
  CHECK(hipMemset(acc, 0, (size_t)n_rows * 64));
  const int waves = 30016;          // the blend backward's grid (one wave per 8x8 block of 1600x1200)
  printf("# 64-byte accumulator rows at random rows of a %u-row table (64 B each), %d one-wave workgroups\n", n_rows, waves);
  printf("# rows/instr = 4 unless noted; 'active' = lanes of 16 that carry a float; spacer = dependent VALU instructions between instructions\n");
  printf("%-44s %8s %10s %12s %14s %14s\n", "variant", "active", "spacer", "M rows", "us", "G rows/s");
  struct V { const char *name; int mode, active, spacer, iters; };
  const V vs[] = {
      {"atomic, random rows", 0, 10, 0, 40},
      {"atomic, random rows", 0, 16, 0, 40},
      {"atomic, random rows", 0, 10, 330, 40},
      {"atomic, 4 waves of a tile share rows", 1, 10, 0, 40},
      {"atomic, 4 waves of a tile share rows", 1, 10, 330, 40},
      {"atomic, tile-local window of 4096 rows", 4, 10, 0, 40},
      {"atomic, tile-local window of 4096 rows", 4, 10, 330, 40},
      {"plain nontemporal row stores", 2, 10, 0, 40},
      {"plain nontemporal row stores", 2, 16, 0, 40},
      {"atomic, ONE row per instruction", 3, 10, 0, 40},
  };
  for (const V &v : vs) {
    auto launch = [&]() {
      switch (v.mode) {
        case 0: k_rows<0><<<waves, 64>>>(acc, n_rows, v.iters, v.active, v.spacer); break;
        case 1: k_rows<1><<<waves, 64>>>(acc, n_rows, v.iters, v.active, v.spacer); break;
        case 2: k_rows<2><<<waves, 64>>>(acc, n_rows, v.iters, v.active, v.spacer); break;
        case 3: k_rows<3><<<waves, 64>>>(acc, n_rows, v.iters, v.active, v.spacer); break;
        case 4: k_rows<4><<<waves, 64>>>(acc, n_rows, v.iters, v.active, v.spacer); break;
      }
    };
    const float ms = time_ms(launch, 5);
    const double rows = (double)waves * v.iters * (v.mode == 3 ? 1 : 4);
    printf("%-44s %8d %10d %12.2f %14.1f %14.2f\n", v.name, v.active, v.spacer, rows / 1e6, ms * 1e3, rows / (ms * 1e-3) / 1e9);
  }
  // what one S3 view asks for: 4.42 M rows (one per (block, hit)) against 2.43 M (one per (tile, splat))
  for (int per_tile = 0; per_tile < 2; per_tile++) {
    const double want = per_tile ? 2.43e6 : 4.42e6;
    const int iters = (int)(want / 4 / waves + 0.5);
    auto launch = [&]() { if (per_tile) k_rows<0><<<waves, 64>>>(acc, n_rows, iters, 10, 0); else k_rows<1><<<waves, 64>>>(acc, n_rows, iters, 10, 0); };
    const float ms = time_ms(launch, 5);
    printf("%-44s %8d %10d %12.2f %14.1f %14.2f\n", per_tile ? "one view, row per (tile, splat), disjoint" : "one view, row per (block, hit), shared", 10, 0,
           (double)waves * iters * 4 / 1e6, ms * 1e3, (double)waves * iters * 4 / (ms * 1e-3) / 1e9);
  }
  // the tile-level table: 7500 workgroups of 256 threads, 592 (block, hit) rows each into a 324-slot LDS table, then 324 global rows
  {
    const int wgs = 7500, slots = 324, iters = 37;   // 4 waves x 37 iterations x 4 rows = 592 (block, hit) rows per tile (4.42 M / 7500)
    auto launch = [&]() { k_lds_table<<<wgs, 256, slots * 64>>>(acc, n_rows, iters, slots); };
    const float ms = time_ms(launch, 5);
    printf("%-44s %8d %10d %12.2f %14.1f %14s\n", "LDS table per tile + one global row per slot", 10, 0, (double)wgs * slots / 1e6, ms * 1e3, "-");
    printf("#   (%d workgroups x (%d LDS row-adds + %d global atomic rows))\n", wgs, 4 * iters * 4, slots);
  }
  hipFree(acc);
  return 0;
}
