// gs3d_forward.hip — 3DGS forward for gfx950: project -> per-tile count -> scan -> bucket -> per-tile LDS depth
// sort -> front-to-back blend.
//
// Replaces the forward half of `diff_gaussian_rasterization` as called at
// gs3dgs/gaussian_renderer/__init__.py:101-111.  The arithmetic follows oracle/gs3d_oracle.c (the CPU
// restatement of the published algorithm); the pipeline does not follow the CUDA original's
// duplicate-with-64-bit-keys + global radix sort: (tile,splat) pairs are counted per tile while projecting,
// bucketed by tile with one atomic per pair, and each tile's list (a few hundred entries) is depth-sorted in
// LDS by the workgroup that owns the tile.  Ties in depth are broken by splat index, which makes the order —
// and therefore the image — independent of atomic arrival order.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "exp_mfma.hpp"
#if defined(SCORP_FWD_MERGE) && SCORP_FWD_MERGE
#include "blend_group_asm.hpp"   // generated: python scripts/dev/gen_blend_group_asm.py (a parked experiment, off by default)
#endif

namespace scorp {
namespace {

// ---------------------------------------------------------------------------------------------------------
// K2: exclusive prefix sum of the per-tile counts (one workgroup; T is 7.5k at 1600x1200, <100k at 5400x4050).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
scan_tiles_kernel(const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_start, int tiles,
                  StateHeader *__restrict__ header) {
  __shared__ uint32_t s_wave[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (tiles + 1023) / 1024;
  const int lo = min(tiles, t * per), hi = min(tiles, lo + per);
  // the thread's counts stay in registers between the two passes when they fit (per <= 8: up to 8192 tiles)
  uint32_t cnt[8];
  uint32_t sum = 0;
  if (per <= 8) {
#pragma unroll
    for (int j = 0; j < 8; j++) cnt[j] = lo + j < hi ? tile_count[lo + j] : 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) sum += cnt[j];
  } else {
    for (int k = lo; k < hi; k++) sum += tile_count[k];
  }
  // inclusive scan over the 1024 partials: shuffles inside each wave, the 16 wave totals through LDS (two barriers
  // instead of the twenty of a Hillis-Steele over LDS)
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    uint32_t w = lane < 16 ? s_wave[lane] : 0u;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)w, off, 64);
      if (lane >= off) w += v;
    }
    if (lane < 16) s_wave[lane] = w;   // inclusive totals of waves 0..lane
  }
  __syncthreads();
  incl += wave > 0 ? s_wave[wave - 1] : 0u;
  const uint32_t total = s_wave[15];
  uint32_t run = incl - sum;
  uint2 *range = reinterpret_cast<uint2 *>(tile_start);   // (start, end) per tile
  if (per <= 8) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (lo + j < hi) range[lo + j] = make_uint2(run, run + cnt[j]);
      run += cnt[j];
    }
  } else {
    for (int k = lo; k < hi; k++) {
      const uint32_t c = tile_count[k];
      range[k] = make_uint2(run, run + c);
      run += c;
    }
  }
  if (t == 1023) {
    header->num_pairs = total;
    header->overflow = 0;
  }
}

// The same scan for up to kMaxLdsTiles tiles with the counts staged in LDS: coalesced, independent loads and stores (the
// kernel above walks `per` consecutive counts per thread with dependent, uncoalesced global loads - 56 us for the 37 500
// tiles of the align sweep's 15 stacked views, all of it latency).
__global__ void __launch_bounds__(1024)
scan_tiles_lds_kernel(const uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_start, int tiles,
                      StateHeader *__restrict__ header) {
  extern __shared__ uint32_t s_all[];
  __shared__ uint32_t s_wave[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int k = t; k < tiles; k += 1024) s_all[k] = tile_count[k];
  __syncthreads();
  const int per = (tiles + 1023) / 1024;
  const int lo = min(tiles, t * per), hi = min(tiles, lo + per);
  uint32_t sum = 0;
  for (int k = lo; k < hi; k++) sum += s_all[k];
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    uint32_t w = lane < 16 ? s_wave[lane] : 0u;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)w, off, 64);
      if (lane >= off) w += v;
    }
    if (lane < 16) s_wave[lane] = w;
  }
  __syncthreads();
  uint32_t run = incl - sum + (wave > 0 ? s_wave[wave - 1] : 0u);
  const uint32_t total = s_wave[15];
  uint2 *range = reinterpret_cast<uint2 *>(tile_start);   // (start, end) per tile
  for (int k = lo; k < hi; k++) {
    const uint32_t v = s_all[k];
    range[k] = make_uint2(run, run + v);
    run += v;
  }
  if (t == 0) {
    header->num_pairs = total;
    header->overflow = 0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// K3: bucket (tile,splat) pairs by tile. The per-tile counter doubles as the cursor (counted back down to zero,
// so it is clean for the next view). Slot order inside a tile is arbitrary; the sort below fixes it.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
scatter_pairs_kernel(int N, const BinRec *__restrict__ bin, const uint64_t *__restrict__ tile_mask,
                     uint32_t *__restrict__ tile_count,
                     const uint32_t *__restrict__ tile_start, int tiles_x, uint64_t *__restrict__ keys,
                     uint32_t capacity, StateHeader *__restrict__ header, uint32_t *__restrict__ header_copy) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    header->capacity = capacity;
    header->long_tiles = 0;   // (the sort's list of long tiles starts empty)
    const uint32_t np = header->num_pairs, ov = np > capacity ? 1u : header->overflow;
    if (np > capacity) header->overflow = 1;
    if (header_copy) { header_copy[0] = np; header_copy[1] = ov; header_copy[2] = capacity; header_copy[3] = 0u; }
  }
  if (i >= N) return;
  const uint4 raw = reinterpret_cast<const uint4 *>(bin)[i];
  const BinRec br = *reinterpret_cast<const BinRec *>(&raw);
  if ((br.radius & kRadiusMask) == 0) return;
  const uint64_t key = ((uint64_t)br.depth_bits << 32) | (uint32_t)i;
  for_each_tile(br.x0, br.y0, br.x1, br.y1, tile_mask[i], tiles_x, [&](int t) {
    const uint32_t slot = tile_start[2 * t] + atomicSub(&tile_count[t], 1u) - 1u;
    if (slot < capacity) keys[slot] = key;
  });
}

// ---------------------------------------------------------------------------------------------------------
// LDS-histogram binning (the default).  Device-scope atomics on random addresses execute at the memory side, one
// request per lane (~14 G/s measured here), so counting and bucketing 3.3 M pairs through global counters cost
// ~450 us.  Instead each block owns a contiguous range of Gaussians and a private per-tile histogram in LDS:
//   count : LDS atomics; the histogram is written out as one coalesced row  hist[block][tile];
//   scan  : per tile, an exclusive prefix over blocks (column of hist, coalesced across threads) + the tile total;
//           then the existing single-block scan of the totals gives tile_start;
//   scatter: the block reloads its row (+ tile_start) as LDS cursors and ranks its pairs with returning LDS atomics.
// Slot order inside a tile is arbitrary but deterministic; the per-tile depth sort fixes the final order.
// ---------------------------------------------------------------------------------------------------------
constexpr int kBinAhead = 4;         // Gaussians whose records one thread loads together
constexpr int kBinThreads = 1024;   // few Gaussians per thread: the count / scatter loops are latency chains (load -> LDS atomic -> store)

// Images with more tiles than one LDS histogram holds are binned in passes: workgroup (pass, block) owns the tile range
// [pass * tpp, (pass + 1) * tpp) of bin block `block` (blockIdx.x = pass * nb + block).
// kCells: the bins are cells of kCellTiles x kCellTiles tiles (`tiles` = number of cells, `tiles_x` = cells per row): the
// first level of the two-level binning
template <bool kCells>
__global__ void __launch_bounds__(kBinThreads)
count_tiles_lds_kernel(int N, int per_block, int nb, int tpp, int view_n, const BinRec *__restrict__ bin,
                       const uint64_t *__restrict__ tile_mask, int tiles, int tiles_x, uint32_t *__restrict__ block_hist,
                       StateHeader *__restrict__ header) {
  extern __shared__ uint32_t s_hist[];
  if (header && blockIdx.x == 0 && threadIdx.x == 0) { header->num_pairs = 0; header->overflow = 0; }   // scan_block_hist adds the totals up
  const int pass = blockIdx.x / nb, blk = blockIdx.x - pass * nb;
  const int t_lo = pass * tpp, nt = min(tiles - t_lo, tpp);
  for (int t = threadIdx.x; t < nt; t += kBinThreads) s_hist[t] = 0;
  __syncthreads();
  // (stacked views: pass v looks at view v's Gaussians only, [v * view_n, (v + 1) * view_n))
  const int g0 = view_n >= 0 ? pass * view_n : 0, g1 = view_n >= 0 ? g0 + view_n : N;
  const int lo = g0 + blk * per_block, hi = min(g1, lo + per_block);
  // (the loads of kBinAhead Gaussians are issued together: one thread walks 2 - 4 of them, and with a load per iteration that
  // was as many dependent round trips to memory)
  for (int i0 = lo + threadIdx.x; i0 < hi; i0 += kBinAhead * kBinThreads) {
    uint4 raws[kBinAhead];
    uint64_t masks[kBinAhead];
#pragma unroll
    for (int u = 0; u < kBinAhead; u++) {
      const int i = min(i0 + u * kBinThreads, hi - 1);
      raws[u] = reinterpret_cast<const uint4 *>(bin)[i];
      masks[u] = tile_mask[i];
    }
#pragma unroll
    for (int u = 0; u < kBinAhead; u++) {
      if (i0 + u * kBinThreads >= hi) break;
      const BinRec br = *reinterpret_cast<const BinRec *>(&raws[u]);
      const uint64_t mask = masks[u];
      if ((br.radius & kRadiusMask) == 0) continue;
      if constexpr (kCells) {
        for_each_tile_xy(br.x0, br.y0, br.x1, br.y1, mask, [&](int x, int y) {
          atomicAdd(&s_hist[(y / kCellTiles) * tiles_x + x / kCellTiles], 1u);
        });
      } else {
        for_each_tile(br.x0, br.y0, br.x1, br.y1, mask, tiles_x, [&](int t) {
          const uint32_t r = (uint32_t)(t - t_lo);
          if (r < (uint32_t)nt) atomicAdd(&s_hist[r], 1u);
        });
      }
    }
  }
  __syncthreads();
  uint32_t *row = block_hist + (size_t)blk * tiles + t_lo;
  for (int t = threadIdx.x; t < nt; t += kBinThreads) row[t] = s_hist[t];
}

// 32 tiles x 32 segments of the block range per workgroup (235 workgroups at 7500 tiles — enough to cover every CU;
// the first form, 64 x 16, left half the chip idle): each thread sums its blocks for one tile (coalesced 128-byte
// rows), the segment totals are exchanged through LDS, then the prefixes are written in place.
constexpr int kScanTiles = 32, kScanSegs = 32;
__global__ void __launch_bounds__(kScanTiles * kScanSegs)
scan_block_hist_kernel(int nb, int tiles, uint32_t *__restrict__ block_hist, uint32_t *__restrict__ tile_count,
                       StateHeader *__restrict__ header) {
  __shared__ uint32_t s_seg[kScanSegs][kScanTiles];
  const int tl = threadIdx.x % kScanTiles, seg = threadIdx.x / kScanTiles;
  const int t = blockIdx.x * kScanTiles + tl;
  const int per = (nb + kScanSegs - 1) / kScanSegs;
  const int b0 = min(nb, seg * per), b1 = min(nb, b0 + per);
  // the segment's counts stay in registers between the two passes (nb <= 512 -> at most 16 per thread), and all of
  // its loads are in flight together instead of one per loop iteration
  constexpr int kMaxPer = (kBinBlocksMax + kScanSegs - 1) / kScanSegs;
  uint32_t cnt[kMaxPer];
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < kMaxPer; j++) {
    const int b = b0 + j;
    cnt[j] = (t < tiles && b < b1) ? block_hist[(size_t)b * tiles + t] : 0u;
  }
#pragma unroll
  for (int j = 0; j < kMaxPer; j++) sum += cnt[j];
  s_seg[seg][tl] = sum;
  __syncthreads();
  uint32_t run = 0;
  for (int q = 0; q < seg; q++) run += s_seg[q][tl];
  if (t < tiles) {
#pragma unroll
    for (int j = 0; j < kMaxPer; j++) {
      const int b = b0 + j;
      if (b < b1) block_hist[(size_t)b * tiles + t] = run;
      run += cnt[j];
    }
    if (seg == kScanSegs - 1) tile_count[t] = run;
  }
  if (header && seg == kScanSegs - 1) {   // D = the sum of the tile totals: this workgroup's 32 (lanes 32..63 of its last wave)
    static_assert(kScanTiles == 32, "the last segment is the upper half of a wave");
    uint32_t tot = t < tiles ? run : 0u;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off, 64);
    if (tl == 0 && tot) atomicAdd(&header->num_pairs, tot);
  }
}

// kCells: bins are cells (see count_tiles_lds_kernel); `tile_start` is then the plain prefix cell_start[cells + 1] and a key
// carries its tile's index inside the cell in bits kCellShift.. of its low word (expand_cells_kernel strips it again)
template <bool kCells>
__global__ void __launch_bounds__(kBinThreads)
scatter_pairs_lds_kernel(int N, int per_block, int nb, int tpp, int view_n, const BinRec *__restrict__ bin,
                         const uint64_t *__restrict__ tile_mask, int tiles, int tiles_x,
                         const uint32_t *__restrict__ block_hist, uint32_t *__restrict__ tile_start,
                         const uint32_t *__restrict__ tile_count, uint64_t *__restrict__ keys, uint32_t capacity,
                         StateHeader *__restrict__ header, uint32_t *__restrict__ header_copy) {
  extern __shared__ uint32_t s_cur[];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    header->capacity = capacity;
    header->long_tiles = 0;   // (the sort's list of long tiles starts empty)
    const uint32_t np = header->num_pairs, ov = np > capacity ? 1u : header->overflow;
    if (np > capacity) header->overflow = 1;
    if (header_copy) { header_copy[0] = np; header_copy[1] = ov; header_copy[2] = capacity; header_copy[3] = 0u; }
  }
  const int pass = blockIdx.x / nb, blk = blockIdx.x - pass * nb;
  const int t_lo = pass * tpp, nt = min(tiles - t_lo, tpp);
  const uint32_t *row = block_hist + (size_t)blk * tiles + t_lo;
  if (tile_count) {
    // (one pass, tiles <= 8192) tile_start is not there yet: every workgroup takes the exclusive prefix of the tile
    // totals itself - 8 counts per thread, wave scans, the 16 wave totals through LDS - straight into its cursors;
    // workgroup 0 also writes it out for the sort and the blend kernels
    static_assert(kBinThreads == 1024, "8 counts per thread cover 8192 tiles");
    __shared__ uint32_t s_wave[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t cnt[8], rw[8], sum = 0;
    if ((tiles & 3) == 0 && 8 * t + 8 <= tiles) {   // rows of the histogram matrix are 16-byte aligned when tiles % 4 == 0
      const uint4 c0 = reinterpret_cast<const uint4 *>(tile_count)[2 * t], c1 = reinterpret_cast<const uint4 *>(tile_count)[2 * t + 1];
      const uint4 r0 = reinterpret_cast<const uint4 *>(row)[2 * t], r1 = reinterpret_cast<const uint4 *>(row)[2 * t + 1];
      cnt[0] = c0.x; cnt[1] = c0.y; cnt[2] = c0.z; cnt[3] = c0.w; cnt[4] = c1.x; cnt[5] = c1.y; cnt[6] = c1.z; cnt[7] = c1.w;
      rw[0] = r0.x; rw[1] = r0.y; rw[2] = r0.z; rw[3] = r0.w; rw[4] = r1.x; rw[5] = r1.y; rw[6] = r1.z; rw[7] = r1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const bool in = 8 * t + j < tiles;
        cnt[j] = in ? tile_count[8 * t + j] : 0u;
        rw[j] = in ? row[8 * t + j] : 0u;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) sum += cnt[j];
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wave; w++) run += s_wave[w];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (8 * t + j < tiles) {
        s_cur[8 * t + j] = run + rw[j];
        if (blockIdx.x == 0) {
          if constexpr (kCells) tile_start[8 * t + j] = run;
          else reinterpret_cast<uint2 *>(tile_start)[8 * t + j] = make_uint2(run, run + cnt[j]);
        }
      }
      run += cnt[j];
    }
    if (kCells && blockIdx.x == 0 && t == kBinThreads - 1) tile_start[tiles] = run;
  } else {
    for (int t = threadIdx.x; t < nt; t += kBinThreads) s_cur[t] = tile_start[2 * (t_lo + t)] + row[t];
  }
  __syncthreads();
  const int g0 = view_n >= 0 ? pass * view_n : 0, g1 = view_n >= 0 ? g0 + view_n : N;
  const int lo = g0 + blk * per_block, hi = min(g1, lo + per_block);
  for (int i0 = lo + threadIdx.x; i0 < hi; i0 += kBinAhead * kBinThreads) {
    uint4 raws[kBinAhead];
    uint64_t masks[kBinAhead];
#pragma unroll
    for (int u = 0; u < kBinAhead; u++) {
      const int i = min(i0 + u * kBinThreads, hi - 1);
      raws[u] = reinterpret_cast<const uint4 *>(bin)[i];
      masks[u] = tile_mask[i];
    }
#pragma unroll
    for (int u = 0; u < kBinAhead; u++) {
    const int i = i0 + u * kBinThreads;
    if (i >= hi) break;
    const BinRec br = *reinterpret_cast<const BinRec *>(&raws[u]);
    const uint64_t mask = masks[u];
    if ((br.radius & kRadiusMask) == 0) continue;
    const uint64_t key = ((uint64_t)br.depth_bits << 32) | (uint32_t)i;
    if constexpr (kCells) {
      for_each_tile_xy(br.x0, br.y0, br.x1, br.y1, mask, [&](int x, int y) {
        const uint32_t slot = atomicAdd(&s_cur[(y / kCellTiles) * tiles_x + x / kCellTiles], 1u);
        if (slot < capacity) keys[slot] = key | (uint64_t)((uint32_t)((y % kCellTiles) * kCellTiles + x % kCellTiles) << kCellShift);
      });
    } else {
      for_each_tile(br.x0, br.y0, br.x1, br.y1, mask, tiles_x, [&](int t) {
        const uint32_t r = (uint32_t)(t - t_lo);
        if (r < (uint32_t)nt) {
          const uint32_t slot = atomicAdd(&s_cur[r], 1u);
          if (slot < capacity) keys[slot] = key;
        }
      });
    }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Second level of the two-level binning: ONE WORKGROUP PER CELL.  The cell's keys (contiguous, in arbitrary order) are
// counted per tile (sixteen LDS counters), the sixteen tile buckets are laid out one after the other inside the cell's own
// range of the second key buffer, every tile's (start, end) is written, and the keys go to their buckets with the tile
// index stripped - so that what the per-tile sort reads is exactly what the one-level scatter produces.  Up to
// kExpandKeep keys per thread stay in registers between the two passes (a cell of S3 holds ~5 100 pairs); longer cells read
// the rest again.  A view that overflowed its reservation is clamped to `capacity` as everywhere else.
// (Staging the buckets in LDS and copying them out linearly - a wave's 64 stores consecutive instead of scattered over the
// cell's 40 KB - was built and is SLOWER, 19.1 against 16.2 us: with 64 KB of LDS two workgroups share a CU, and the kernel
// is a chain of round trips, not a stream of stores.)
// ---------------------------------------------------------------------------------------------------------
constexpr int kExpandThreads = 512, kExpandKeep = 12;
__global__ void __launch_bounds__(kExpandThreads)
expand_cells_kernel(const uint32_t *__restrict__ cell_start, const uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys_out,
                    uint32_t *__restrict__ tile_range, uint32_t capacity, int cells_x, int tiles_x, int tiles_y) {
  constexpr int kBins = kCellTiles * kCellTiles;
  __shared__ uint32_t s_cnt[kBins], s_cur[kBins];
  const int cell = blockIdx.x, tid = threadIdx.x;
  const uint32_t beg = min(cell_start[cell], capacity), end = min(cell_start[cell + 1], capacity);
  if (tid < kBins) s_cnt[tid] = 0;
  __syncthreads();
  uint64_t kk[kExpandKeep];
#pragma unroll
  for (int j = 0; j < kExpandKeep; j++) {
    const uint32_t i = beg + tid + j * kExpandThreads;
    kk[j] = i < end ? keys_in[i] : ~0ull;
  }
  auto bin_of = [](uint64_t k) { return (uint32_t)(k >> kCellShift) & (uint32_t)(kBins - 1); };
#pragma unroll
  for (int j = 0; j < kExpandKeep; j++)
    if (beg + tid + j * kExpandThreads < end) atomicAdd(&s_cnt[bin_of(kk[j])], 1u);
  for (uint32_t i = beg + tid + kExpandKeep * kExpandThreads; i < end; i += kExpandThreads) atomicAdd(&s_cnt[bin_of(keys_in[i])], 1u);
  __syncthreads();
  if (tid < kBins) {
    uint32_t run = beg;
    for (int b = 0; b < tid; b++) run += s_cnt[b];
    s_cur[tid] = run;
    const int tx = (cell % cells_x) * kCellTiles + tid % kCellTiles, ty = (cell / cells_x) * kCellTiles + tid / kCellTiles;
    if (tx < tiles_x && ty < tiles_y) reinterpret_cast<uint2 *>(tile_range)[ty * tiles_x + tx] = make_uint2(run, run + s_cnt[tid]);
  }
  __syncthreads();
  constexpr uint64_t kStrip = ~((uint64_t)(kBins - 1) << kCellShift);
#pragma unroll
  for (int j = 0; j < kExpandKeep; j++)
    if (beg + tid + j * kExpandThreads < end) keys_out[atomicAdd(&s_cur[bin_of(kk[j])], 1u)] = kk[j] & kStrip;
  for (uint32_t i = beg + tid + kExpandKeep * kExpandThreads; i < end; i += kExpandThreads) {
    const uint64_t k = keys_in[i];
    keys_out[atomicAdd(&s_cur[bin_of(k)], 1u)] = k & kStrip;
  }
}

// ---------------------------------------------------------------------------------------------------------
// K4: per-tile depth sort. One workgroup per tile; bitonic network (all-ascending "flip" form, so virtual +inf
// padding never moves) in LDS for lists up to kSortLds entries, in global memory (same network) beyond that.
// Output: point_list = splat indices front to back.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSortLds = 4096;  // 32 KiB of 64-bit keys

// Comparator i of a stage is always handled by thread i % 256, i.e. the 128-element block b = i / 64 belongs to wave
// b % 4 in every stage whose partners are < 128 apart.  Those stages (all of k <= 128 and every j <= 64) therefore
// need no workgroup barrier: a wave's LDS operations retire in order, a wave-level fence is enough.  Only the few
// stages that cross 128-element blocks synchronise the workgroup (3 of 45 for a 512-entry list).
__device__ __forceinline__ void ce(uint64_t &lo, uint64_t &hi) {   // compare-exchange, ascending
  const uint64_t u = lo, v = hi;
  const bool sw = u > v;
  lo = sw ? v : u; hi = sw ? u : v;
}

// kRegTail (LDS lists only): every thread also owns 4 consecutive keys; the stages whose partners are 1 or 2 apart —
// two per level, the ones with the worst LDS bank behaviour — and the whole of levels k = 2, 4 run on those four keys
// in registers, one 16-byte LDS round trip per level instead of one per stage (45 -> 28 LDS stages at 512 keys).
template <bool kWaveLocal, bool kRegTail, typename Ptr>
__device__ __forceinline__ void bitonic_sort(Ptr a, uint32_t n, uint32_t P) {
  const uint32_t tid = threadIdx.x;
  auto sync = [&](bool cross) {
    if (!kWaveLocal || cross) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  };
  constexpr uint64_t kInf = ~0ull;
  bool prev_cross = true;  // the loads that filled `a` came from all waves
  // Register phases keep the LDS stages' ownership (128-key block b belongs to wave b % 4), so they too need only a
  // wave-level fence: lanes 0-31 of wave w take block w (+8, +16, ...), lanes 32-63 block w + 4.
  auto reg_phase = [&](bool first) {
    sync(prev_cross);
    prev_cross = false;
    const uint32_t w = tid >> 6, l = tid & 63;
    for (uint32_t b = w + 4 * (l >> 5); 128 * b < P; b += 8) {
      const uint32_t e = 128 * b + 4 * (l & 31);
      if (e < n) {
        uint64_t v0 = a[e], v1 = e + 1 < n ? a[e + 1] : kInf, v2 = e + 2 < n ? a[e + 2] : kInf, v3 = e + 3 < n ? a[e + 3] : kInf;
        if (first) {
          ce(v0, v1); ce(v2, v3);   // k = 2
          ce(v0, v3); ce(v1, v2);   // k = 4: flip
          ce(v0, v1); ce(v2, v3);   //        j = 1
        } else {
          ce(v0, v2); ce(v1, v3);   // j = 2
          ce(v0, v1); ce(v2, v3);   // j = 1
        }
        a[e] = v0;
        if (e + 1 < n) a[e + 1] = v1;
        if (e + 2 < n) a[e + 2] = v2;
        if (e + 3 < n) a[e + 3] = v3;
      }
    }
  };
  uint32_t lk0 = 1;
  if (kRegTail && P >= 4) { reg_phase(true); lk0 = 3; }
  for (uint32_t lk = lk0; (1u << lk) <= P; lk++) {
    const uint32_t k = 1u << lk, half = k >> 1;
    {
      const bool cross = k > 128;
      sync(cross || prev_cross);
      prev_cross = cross;
      for (uint32_t i = tid; i < (P >> 1); i += 256) {  // flip step: lo <-> mirrored partner inside each k-block
        const uint32_t base = (i >> (lk - 1)) << lk, r = i & (half - 1);
        const uint32_t lo = base + r, hi = base + (k - 1 - r);
        if (hi < n) {
          const uint64_t u = a[lo], v = a[hi];
          if (u > v) { a[lo] = v; a[hi] = u; }
        }
      }
    }
    const int lj_min = kRegTail ? 2 : 0;
    for (int lj = (int)lk - 2; lj >= lj_min; lj--) {
      const uint32_t j = 1u << lj;
      const bool cross = j >= 128;
      sync(cross || prev_cross);
      prev_cross = cross;
      for (uint32_t i = tid; i < (P >> 1); i += 256) {
        const uint32_t lo = ((i >> lj) << (lj + 1)) + (i & (j - 1));
        const uint32_t hi = lo + j;
        if (hi < n) {
          const uint64_t u = a[lo], v = a[hi];
          if (u > v) { a[lo] = v; a[hi] = u; }
        }
      }
    }
    if (kRegTail) reg_phase(false);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// K4, lists of up to 1024 entries (nearly all of them): the same bitonic network with the keys IN REGISTERS.  Thread t
// owns the four consecutive keys 4t .. 4t+3, a wave 256 consecutive keys.  Every comparator of the network pairs
// element e with e ^ X (X = k - 1 for the flip step of level k, X = j for a half-cleaner), i.e. lane ^ (X / 4) with the
// registers in the same (cleaner) or reversed (flip) order, and the lower index keeps the minimum:
//   * X < 4                : inside a thread;
//   * lane masks 1,2,3,7,8,15: one DPP move per dword (quad_perm / row_half_mirror / row_ror:8 / row_mirror), 4 = 7 o 3;
//   * lane masks 16,31,32,63 : ds_bpermute (the LDS crossbar, no LDS memory);
//   * X >= 256             : between waves, through LDS (2 stages of 45 for 512 keys, 5 of 55 for 1024).
// No LDS round trip and no barrier for all the rest, which the LDS version paid per stage.  Waves whose keys are all
// padding leave at once (the network of size P never touches indices >= P).
// ---------------------------------------------------------------------------------------------------------
template <int M>
__device__ __forceinline__ uint32_t xor_lane32(uint32_t v, int bperm_addr) {
  if constexpr (M == 1) return __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  else if constexpr (M == 3) return __builtin_amdgcn_update_dpp(0u, v, 0x1B, 0xF, 0xF, false);   // quad_perm [3,2,1,0]
  else if constexpr (M == 7) return __builtin_amdgcn_update_dpp(0u, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  else if constexpr (M == 15) return __builtin_amdgcn_update_dpp(0u, v, 0x140, 0xF, 0xF, false); // row_mirror
  else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(0u, v, 0x128, 0xF, 0xF, false);  // row_ror:8
  else if constexpr (M == 4) return xor_lane32<7>(xor_lane32<3>(v, 0), 0);
  else return (uint32_t)__builtin_amdgcn_ds_bpermute(bperm_addr, (int)v);                          // 16, 31, 32, 63
}
template <int M>
__device__ __forceinline__ uint64_t xor_lane64(uint64_t v, int bperm_addr) {
  const uint32_t lo = xor_lane32<M>((uint32_t)v, bperm_addr), hi = xor_lane32<M>((uint32_t)(v >> 32), bperm_addr);
  return ((uint64_t)hi << 32) | lo;
}
// one network stage between lanes: element (lane, r) against (lane ^ M, FLIP ? 3 - r : r); keep_min per lane
template <int M, bool FLIP>
__device__ __forceinline__ void lane_stage(uint64_t (&k)[4], bool keep_min, int lane) {
  const int addr = ((lane ^ M) & 63) << 2;
  uint64_t p[4];
#pragma unroll
  for (int r = 0; r < 4; r++) p[r] = xor_lane64<M>(k[FLIP ? 3 - r : r], addr);
#pragma unroll
  for (int r = 0; r < 4; r++) k[r] = ((p[r] < k[r]) == keep_min) ? p[r] : k[r];
}
// one network stage between waves, through LDS: element e against e ^ X (X >= 256; FLIP: X = K - 1)
template <bool FLIP>
__device__ __forceinline__ void cross_stage(uint64_t (&k)[4], uint64_t *s_x, uint32_t base, uint32_t X, bool keep_min) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  u64x2 *mine = reinterpret_cast<u64x2 *>(s_x + base);
  u64x2 w0, w1;
  w0.x = k[0]; w0.y = k[1]; w1.x = k[2]; w1.y = k[3];
  mine[0] = w0; mine[1] = w1;
  __syncthreads();
  const u64x2 *theirs = reinterpret_cast<const u64x2 *>(s_x + (base ^ (X & ~3u)));
  const u64x2 t0 = theirs[0], t1 = theirs[1];
  const uint64_t q[4] = {t0.x, t0.y, t1.x, t1.y};
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint64_t p = q[FLIP ? 3 - r : r];
    k[r] = ((p < k[r]) == keep_min) ? p : k[r];
  }
  __syncthreads();   // everyone has read before the next cross stage overwrites
}
__device__ __forceinline__ void thread_tail(uint64_t (&k)[4]) {   // j = 2, j = 1 of any level
  ce(k[0], k[2]); ce(k[1], k[3]);
  ce(k[0], k[1]); ce(k[2], k[3]);
}
// level K of the network (flip, half-cleaners down to j = 4, then the in-thread tail), lane masks as template constants
template <int K>
__device__ __forceinline__ void sort_level(uint64_t (&k)[4], uint64_t *s_x, uint32_t base, int lane) {
  constexpr int FM = K / 4 - 1;   // lane mask of the flip
  if constexpr (FM < 64) lane_stage<FM, true>(k, (lane & (K / 8)) == 0, lane);
  else cross_stage<true>(k, s_x, base, (uint32_t)K - 1u, (base & (K / 2)) == 0);
#define SCORP_CLEAN(J)                                                                                   \
  if constexpr (K / 4 >= (J) && (J) >= 4) {                                                              \
    if constexpr ((J) / 4 < 64) lane_stage<((J) / 4 < 64 ? (J) / 4 : 1), false>(k, (lane & ((J) / 4)) == 0, lane); \
    else cross_stage<false>(k, s_x, base, (uint32_t)(J), (base & (J)) == 0);                             \
  }
  SCORP_CLEAN(256) SCORP_CLEAN(128) SCORP_CLEAN(64) SCORP_CLEAN(32) SCORP_CLEAN(16) SCORP_CLEAN(8) SCORP_CLEAN(4)
#undef SCORP_CLEAN
  thread_tail(k);
}

__global__ void __launch_bounds__(256)
sort_tiles_reg_kernel(const uint32_t *__restrict__ tile_start, const uint64_t *__restrict__ keys,
                      uint32_t *__restrict__ point_list, uint32_t capacity, uint32_t *__restrict__ long_list,
                      StateHeader *__restrict__ header) {
  __shared__ __attribute__((aligned(16))) uint64_t s_x[1024];
  const int tile = blockIdx.x;
  const uint32_t beg = min(tile_start[2 * tile], capacity), end = min(tile_start[2 * tile + 1], capacity);   // (start, end) per tile
  const uint32_t n = end - beg;
  if (n > 1024) {   // sort_tiles_long_kernel's: it walks the list of such tiles (usually empty) instead of every tile
    if (threadIdx.x == 0) long_list[atomicAdd(&header->long_tiles, 1u)] = (uint32_t)tile;
    return;
  }
  if (n == 0) return;
  const uint32_t base = 4 * threadIdx.x;
  uint32_t P = 4;
  while (P < n) P <<= 1;
  if (256u * (threadIdx.x >> 6) >= P) return;   // whole waves of padding leave (never single lanes: lanes exchange)
  const int lane = threadIdx.x & 63;
  constexpr uint64_t kInf = ~0ull;
  uint64_t k[4];
#pragma unroll
  for (int r = 0; r < 4; r++) k[r] = base + r < n ? keys[beg + base + r] : kInf;
  ce(k[0], k[1]); ce(k[2], k[3]);                          // k = 2
  ce(k[0], k[3]); ce(k[1], k[2]); ce(k[0], k[1]); ce(k[2], k[3]);   // k = 4: flip, j = 1
  if (P >= 8) sort_level<8>(k, s_x, base, lane);
  if (P >= 16) sort_level<16>(k, s_x, base, lane);
  if (P >= 32) sort_level<32>(k, s_x, base, lane);
  if (P >= 64) sort_level<64>(k, s_x, base, lane);
  if (P >= 128) sort_level<128>(k, s_x, base, lane);
  if (P >= 256) sort_level<256>(k, s_x, base, lane);
  if (P >= 512) sort_level<512>(k, s_x, base, lane);
  if (P >= 1024) sort_level<1024>(k, s_x, base, lane);
#pragma unroll
  for (int r = 0; r < 4; r++)
    if (base + r < n) point_list[beg + base + r] = (uint32_t)k[r];
}

// ---------------------------------------------------------------------------------------------------------
// K4, lists longer than 1024 entries - a second launch, so that the common case keeps its 8 KiB LDS footprint.
//   * up to kSortLds = 4096 entries: the list is cut into chunks of 1024 (256 threads x 4 keys).  Every chunk is sorted
//     by the register network above (levels 2 .. 1024) and parked in LDS; the remaining one or two levels of the
//     network (2048, 4096) run as their chunk-crossing stages on the LDS array (the flip, and for 4096 the half-cleaner
//     of distance 1024) followed, per chunk, by the half-cleaners 512 .. 1 in registers again.  A 4096-entry list costs
//     3 LDS stages + 4 + 8 register passes instead of the 78 LDS round trips of the plain LDS network (which this
//     replaced: 87 -> 57 us and less on the 4 x 100k-object scene of config #4, whose tiles hold 1-3 k splats);
//   * beyond: the plain network on global memory (one workgroup: its barriers order its own accesses).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void clean_chunk_1024(uint64_t (&k)[4], uint64_t *s_x, uint32_t base, int lane) {
  // the half-cleaners 512 .. 1 of a level above 1024, inside one 1024-key chunk
  cross_stage<false>(k, s_x, base, 512u, (base & 512u) == 0);
  cross_stage<false>(k, s_x, base, 256u, (base & 256u) == 0);
  lane_stage<32, false>(k, (lane & 32) == 0, lane);
  lane_stage<16, false>(k, (lane & 16) == 0, lane);
  lane_stage<8, false>(k, (lane & 8) == 0, lane);
  lane_stage<4, false>(k, (lane & 4) == 0, lane);
  lane_stage<2, false>(k, (lane & 2) == 0, lane);
  lane_stage<1, false>(k, (lane & 1) == 0, lane);
  thread_tail(k);
}

__device__ __forceinline__ void sort_long_tile(int tile, const uint32_t *__restrict__ tile_start, uint64_t *__restrict__ keys,
                                               uint32_t *__restrict__ point_list, uint32_t capacity, uint64_t *s_keys, uint64_t *s_x) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const uint32_t beg = min(tile_start[2 * tile], capacity), end = min(tile_start[2 * tile + 1], capacity);   // (start, end) per tile
  const uint32_t n = end - beg;
  if (n <= 1024) return;   // sort_tiles_reg_kernel's
  uint32_t P = 2048;
  while (P < n) P <<= 1;
  if (n > (uint32_t)kSortLds) {
    bitonic_sort<false, false>(keys + beg, n, P);
    for (uint32_t i = threadIdx.x; i < n; i += 256) point_list[beg + i] = (uint32_t)keys[beg + i];
    return;
  }
  const uint32_t base = 4 * threadIdx.x;
  const int lane = threadIdx.x & 63;
  constexpr uint64_t kInf = ~0ull;
  uint64_t k[4];
  auto park = [&](uint32_t c) {
    u64x2 w0, w1;
    w0.x = k[0]; w0.y = k[1]; w1.x = k[2]; w1.y = k[3];
    u64x2 *dst = reinterpret_cast<u64x2 *>(s_keys + c + base);
    dst[0] = w0; dst[1] = w1;
  };
  auto fetch = [&](uint32_t c) {
    const u64x2 *src = reinterpret_cast<const u64x2 *>(s_keys + c + base);
    const u64x2 w0 = src[0], w1 = src[1];
    k[0] = w0.x; k[1] = w0.y; k[2] = w1.x; k[3] = w1.y;
  };
  for (uint32_t c = 0; c < P; c += 1024) {   // levels 2 .. 1024, chunk by chunk, in registers
#pragma unroll
    for (int r = 0; r < 4; r++) k[r] = c + base + r < n ? keys[beg + c + base + r] : kInf;
    if (c < n) {   // (a chunk of nothing but padding is sorted as it is)
      uint32_t Pc = 4;   // the last chunk's network only as large as its real entries need: the padding never moves
      while (Pc < n - c && Pc < 1024) Pc <<= 1;
      ce(k[0], k[1]); ce(k[2], k[3]);
      ce(k[0], k[3]); ce(k[1], k[2]); ce(k[0], k[1]); ce(k[2], k[3]);
      if (Pc >= 8) sort_level<8>(k, s_x, base, lane);
      if (Pc >= 16) sort_level<16>(k, s_x, base, lane);
      if (Pc >= 32) sort_level<32>(k, s_x, base, lane);
      if (Pc >= 64) sort_level<64>(k, s_x, base, lane);
      if (Pc >= 128) sort_level<128>(k, s_x, base, lane);
      if (Pc >= 256) sort_level<256>(k, s_x, base, lane);
      if (Pc >= 512) sort_level<512>(k, s_x, base, lane);     // (workgroup-uniform: these two hold barriers)
      if (Pc >= 1024) sort_level<1024>(k, s_x, base, lane);
    }
    park(c);
  }
  for (uint32_t K = 2048; K <= P; K <<= 1) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (P >> 1); i += 256) {   // flip: lo <-> mirrored partner inside each K-block
      const uint32_t lo = (i / (K >> 1)) * K + (i & ((K >> 1) - 1)), hi = lo ^ (K - 1);
      const uint64_t u = s_keys[lo], v = s_keys[hi];
      if (u > v) { s_keys[lo] = v; s_keys[hi] = u; }
    }
    if (K == 4096) {
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < (P >> 1); i += 256) {   // half-cleaner of distance 1024
        const uint32_t lo = ((i >> 10) << 11) + (i & 1023u), hi = lo + 1024;
        const uint64_t u = s_keys[lo], v = s_keys[hi];
        if (u > v) { s_keys[lo] = v; s_keys[hi] = u; }
      }
    }
    __syncthreads();
    for (uint32_t c = 0; c < n; c += 1024) {   // (chunks at or above n hold padding only, before and after)
      fetch(c);
      clean_chunk_1024(k, s_x, base, lane);
      if (K == P) {   // last level: straight out
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (c + base + r < n) point_list[beg + c + base + r] = (uint32_t)k[r];
      } else {
        park(c);
      }
    }
  }
}

// A fixed, small grid walks the list of long tiles the register kernel left (header->long_tiles ids in long_list): with
// no long tile - the usual case - its workgroups read one word and leave, instead of one workgroup per tile doing so.
__global__ void __launch_bounds__(256)
sort_tiles_long_kernel(const uint32_t *__restrict__ tile_start, uint64_t *__restrict__ keys, uint32_t *__restrict__ point_list,
                       uint32_t capacity, const uint32_t *__restrict__ long_list, const StateHeader *__restrict__ header) {
  __shared__ __attribute__((aligned(16))) uint64_t s_keys[kSortLds];
  __shared__ __attribute__((aligned(16))) uint64_t s_x[1024];
  const uint32_t count = header->long_tiles;
  for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
    sort_long_tile((int)long_list[k], tile_start, keys, point_list, capacity, s_keys, s_x);
    __syncthreads();   // the next tile reuses the staging arrays
  }
}

// ---------------------------------------------------------------------------------------------------------
// K5: front-to-back blend with ONE WAVE PER 8x8 BLOCK as the unit (64-thread workgroups, no workgroup barriers), the
// forward twin of blend_backward_wave_kernel: the wave walks the tile's list front to back 64 entries at a time, each
// lane gathers one record and runs the exact conic-vs-block test, survivors are compacted into a per-wave LDS ring,
// and groups of 8 go through a straight-line alpha phase followed by the branch-free sequential blend (a splat that
// does not contribute to a pixel enters as alpha = 0, which changes nothing).  Stops as soon as all 64 pixels are
// saturated.  The four waves of a tile are numbered onto the same XCD.
// For the backward it leaves the block's HIT LIST (the splat ids that passed the test, in list order, compacted, in the
// dead key region of the pair buffer) and per pixel the final T and the 1-based position IN THAT LIST of the last splat
// that contributed: the backward replays the hit list and never looks at the tile's list again.
// ---------------------------------------------------------------------------------------------------------
// The exponent log2(opacity * G) of a group of 16 hits at the block's 64 pixels comes from the matrix cores (exp_mfma.hpp:
// three v_mfma_f32_32x32x16_bf16 against the lane's own monomials; register i of a lane = splat i at that lane's pixel):
// the lane that stages a hit turns its record into the six block-frame coefficients, cut into three bf16 terms each, and a
// ring slot holds those 48 bytes plus (r, g, b, depth).  The sequential blend then costs, per hit and pixel: v_exp, the
// two threshold selects, the T update and four accumulations - the seven VALU instructions of the Horner form (35 % of the
// old loop's issue time together with its LDS reads of the conic) are gone.  A hit's 1-based position in the block's hit
// list is arithmetic (hits are blended in ring order): no position array.
#ifndef SCORP_FWD_DB
#define SCORP_FWD_DB 0
#endif
#ifndef SCORP_FWD_FMA
#define SCORP_FWD_FMA 1
#endif
#ifndef SCORP_FWD_RING
#define SCORP_FWD_RING 80
#endif
#ifndef SCORP_FWD_MERGE
#define SCORP_FWD_MERGE 0     // 1: clamp-free full groups let adjacent hits with disjoint live pixels share an iteration (the parked
                              // experiment scripts/dev/gen_blend_group_asm.py: correct, fewer vector instructions, not faster)
#endif
constexpr int kFRing = SCORP_FWD_RING, kFChunk = 64, kFGroup = 16;   // ring: at most 15 left-over hits + 64 new ones

// __launch_bounds__(64, 6): six waves per SIMD = at most 80 registers.  The kernel needs 84 - 86 left to itself since it
// compacts the hit list (round 4); held to 80 it spills ONE register outside the per-hit loop and keeps its sixth wave
// (same box, rocprof: 157.4 us at five waves per SIMD, 153.4 at six; 148.7 before the compaction, which takes 22 us off
// the backward).  The MFMA results stay in VGPRs either way (no accumulator-register reads in front of the v_exp).
#ifndef SCORP_FWD_WAVES
#define SCORP_FWD_WAVES 6
#endif
#ifdef SCORP_FWD_TRACE
// diagnostic build only (scripts/dev/trace_forward.py): per wave (start, end) on the 100 MHz real-time counter and the
// hardware id words, to draw the occupancy timeline of one launch
__device__ unsigned long long g_fwd_trace[3 * 40000];
#endif
#ifdef SCORP_FWD_STATS
// diagnostic build only (scripts/dev/stats_forward.py): how full the 64 lanes are per blended hit, and how many
// iterations a wave would run if its hits were listed per 8x4 half / per 4x4 quadrant / per pixel instead of per block
__device__ unsigned long long g_fwd_stats[12];
#endif
// kScore (scorp_gs3d_render_score; never with kForBackward): the render-and-compare scoring of a pose hypothesis
// (align.StackedSweep) needs depth and alpha only and needs them only to be compared with a target: no colour is
// accumulated (three of the hit loop's thirteen vector instructions), no image is written, and the epilogue forms this
// pixel's |alpha - alpha*| + |nan_to_num(depth / alpha) - depth*| (scorp_gs3d_pose_score_accumulate's term), sums it over
// the block and leaves ONE float per block in the state's block_hits array - plain stores, added up in a fixed order by
// score_reduce_kernel (so the score, unlike the atomic form's, is the same bits from run to run).
template <bool kForBackward, bool kScore = false>
__global__ void __launch_bounds__(64, SCORP_FWD_WAVES)
blend_forward_wave_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                          const SplatRec *__restrict__ rec, uint32_t capacity, int W, int H, int tiles_x, int tiles,
                          const float *__restrict__ bg, float *__restrict__ out_color, float *__restrict__ out_depth,
                          float *__restrict__ out_alpha, float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                          uint32_t *__restrict__ hits, uint32_t *__restrict__ block_hits, float *__restrict__ out_depth_norm,
                          float4 *__restrict__ zero_buf, uint32_t zero_per_wave, uint32_t zero_total, int band_h,
                          const float *__restrict__ tgt_depth = nullptr, const float *__restrict__ tgt_alpha = nullptr,
                          int score_rows = 0) {
  static_assert(!(kForBackward && kScore), "the scoring form leaves nothing behind for a backward pass");
  __shared__ uint4 q_k[3][kFRing + 1];   // the three bf16 terms of a hit's six coefficients; slot kFRing stays zero
  __shared__ float4 q_col[kFRing];       // r, g, b, depth
  __shared__ uint32_t q_id[kFRing];      // the hit's splat: written to the block's hit list only if some pixel took it (blend_group)
  const int lane = threadIdx.x;
#ifdef SCORP_FWD_TRACE
  struct TraceEnd {
    unsigned long long t0; int lane; unsigned b;
    __device__ ~TraceEnd() {
      if (lane == 0 && b < 40000u) {
        g_fwd_trace[3 * b] = t0; g_fwd_trace[3 * b + 1] = __builtin_amdgcn_s_memrealtime();
        g_fwd_trace[3 * b + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32) |   // HW_REG_HW_ID
                                 (unsigned)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));                       // HW_REG_XCC_ID
      }
    }
  } trace_end{__builtin_amdgcn_s_memrealtime(), lane, blockIdx.x};
#endif
  if (zero_buf) {   // this wave's share of the buffer the launch was asked to clear (every workgroup of the grid takes part)
    const uint32_t z0 = blockIdx.x * zero_per_wave, z1 = min(z0 + zero_per_wave, zero_total);
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v zero = {0.0f, 0.0f, 0.0f, 0.0f};   // streaming stores: the rows are not read before the backward, keep them out of L2
    for (uint32_t z = z0 + lane; z < z1; z += 64) __builtin_nontemporal_store(zero, reinterpret_cast<f4v *>(zero_buf) + z);
  }
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int tile = (kk >> 2) * 8 + xcd, quad = kk & 3;
  if (tile >= tiles) return;
  const int bx = (tile % tiles_x) * kTile + (quad & 1) * 8, by = (tile / tiles_x) * kTile + (quad >> 1) * 8;
  const int px = bx + (lane & 7), py = by + (lane >> 3);
  const bool inside = px < W && py < H;
  // stacked views (ScorpGs3dInputs.num_views): the records hold each view's OWN pixel coordinates, so the block's frame
  // is taken relative to the first row of its band (band_h = rows per view, 0 = one view)
  const int byl = band_h > 0 ? by % band_h : by;
  const float bx0 = (float)bx, bx1 = (float)(bx + 7), by0 = (float)byl, by1 = (float)(byl + 7);
  const float cx = (float)bx + 3.5f, cy = (float)byl + 3.5f;
  const uint32_t beg = min(tile_start[2 * tile], capacity), end = min(tile_start[2 * tile + 1], capacity);   // (start, end) per tile
  const uint32_t n = end - beg;
  uint32_t *my_hits = hits + (size_t)quad * capacity + beg;
  if (lane < 3) q_k[lane][kFRing] = make_uint4(0u, 0u, 0u, 0u);
  const uint4 basis = pixel_basis_frag(lane);
  const bool a_on = a_operand_active(lane);
  const int a_slot = a_operand_slot(lane);
  float T = inside ? 1.0f : -1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f, Dp = 0.0f;   // T < 0: pixel finished (see below)
  uint32_t last = 0;
  int head = 0, count = 0;   // head stays a multiple of kFGroup (only a wave's final group is partial), so the
                             // slots of a group are head + i without wrap-around: one LDS base, immediate offsets
  uint32_t nh = 0;           // hits found so far (wave-uniform)
  uint32_t consumed = 0;     // hits blended so far ((block, splat) iterations: the P statistic)
  uint32_t kept_n = 0;       // ... of which some pixel took: the length of the hit list left for the backward (wave-uniform)
  uint32_t hot_end = 0;      // hits [0, hot_end) may hold a splat with opacity > 0.99 (wave-uniform; see blend_group)
#ifdef SCORP_FWD_STATS
  uint32_t st_hits = 0, st_live = 0, st_any = 0, st_q[4] = {0, 0, 0, 0}, st_h[2] = {0, 0}, st_lane = 0, st_pairs = 0;
  uint32_t st_g16 = 0, st_g64 = 0, st_g8 = 0, st_gq[4] = {0, 0, 0, 0}, st_cq[4] = {0, 0, 0, 0}, st_hq[4] = {0, 0, 0, 0}, st_cn = 0;
  uint64_t st_prev = 0;
  bool st_have = false;
#endif
  // The chunk's gathers (list entry -> record) are dependent loads of ~1 us each; they are software-pipelined: while
  // chunk c is blended the records of chunk c+1 and the list entries of chunk c+2 are already in flight.
  auto fetch_id = [&](uint32_t bs) { return (bs + lane < n) ? point_list[beg + bs + lane] : 0xFFFFFFFFu; };
  auto fetch_rec = [&](uint32_t id_, float4 &a_, float4 &b_, float4 &c_) {
    if (id_ != 0xFFFFFFFFu) {
      const float4 *src = reinterpret_cast<const float4 *>(rec + id_);
      a_ = src[0]; b_ = src[1]; c_ = src[2];
    }
  };
  float4 a, b, c;
  uint32_t id0 = fetch_id(0);
  fetch_rec(id0, a, b, c);
  uint32_t id1 = fetch_id(kFChunk);
  for (uint32_t base = 0; base < n; base += kFChunk) {
    if (__ballot(T > 0.0f) == 0) break;
    float4 a1, b1, c1;
    fetch_rec(id1, a1, b1, c1);
    const uint32_t id2 = fetch_id(base + 2 * kFChunk);
    bool hit = false;
    if (id0 != 0xFFFFFFFFu)   // the record holds -k * conic and k * cutoff (k > 0): the test is scale-invariant
      hit = rec_is_indefinite(c.z) || conic_min_over_box(a.x, a.y, -a.z, -0.5f * a.w, -b.x, bx0, bx1, by0, by1) <= c.z;
    const uint64_t m = __ballot(hit);
    if (hit) {
      const uint32_t rank = nh + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      int qi = head + count + (int)(rank - nh);
      qi = qi >= kFRing ? qi - kFRing : qi;
      uint4 k0, k1, k2;
      splat_block_coefs(a.x, a.y, a.z, a.w, b.x, b.y, cx, cy, k0, k1, k2);
      k0.w = guard_limit_pack(c.w);   // (spare K slots: exp_mfma.hpp)
      q_k[0][qi] = k0; q_k[1][qi] = k1; q_k[2][qi] = k2;
      q_col[qi] = make_float4(b.z, b.w, c.x, c.y);
      if constexpr (kForBackward) q_id[qi] = id0;
    }
    count += __builtin_popcountll(m);
    nh += (uint32_t)__builtin_popcountll(m);
    // (b.y = log2(opacity); the margin keeps exp2(e) <= 0.99 for every other hit whatever the rounding of e)
    // ... and an indefinite conic needs the `power > 0` guard, which the same instantiation carries
    if (__ballot(hit && (b.y > kLog2AlphaMax - 1e-4f || rec_is_indefinite(c.z))) != 0) hot_end = nh;   // groups up to this chunk's last hit keep the clamp
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool last_chunk = base + kFChunk >= n;
    // full groups run straight-line (nslots is the compile-time kFGroup); only a wave's final group is partial
    // log2(opacity * G) of the 16 hits in ring slots h .. h + 15 at this lane's pixel (three MFMAs)
    auto exponents = [&](int h) -> f32x16 {
      int hv = h;
      asm volatile("" : "+v"(hv));
      const int sl = a_on ? hv + a_slot : kFRing;      // the other half of the lanes feeds zeros (exp_mfma.hpp)
      return block_exponents(q_k[0][sl], q_k[1][sl], q_k[2][sl], basis);
    };
    auto blend_group = [&](auto full, auto clampy, int nslots, const f32x16 &e) -> bool {
      constexpr bool kFull = decltype(full)::value;
      // alpha = min(0.99, opacity * G) and G <= 1: the clamp can only bind for a splat whose opacity itself exceeds 0.99.
      // Groups that hold no such hit (all but a few per cent on any scene: opacity > 0.99 is sigmoid(x), x > 4.6) run
      // without the v_min - one VALU instruction of the fourteen per hit.
      constexpr bool kClamp = decltype(clampy)::value;
      int hv = head;
      asm volatile("" : "+v"(hv));   // keep the group's LDS bases in VGPRs (else every ds_read re-moves an SGPR base)
      const float4 *gc = q_col + hv;
      const uint4 *gk = q_k[0] + hv;
      bool all_done = false;
      uint32_t lastg = 0;   // 1-based slot of the group's last contributor to this pixel (inline constants, no SGPR moves)
      uint32_t kept16 = 0;  // wave-uniform: bit i = slot i was taken by at least one pixel of the block
#pragma unroll
      for (int i = 0; i < kFGroup; i++) {
        if (kFull || i < nslots) {  // wave-uniform
          if (kFull && i == kFGroup / 2) {   // every pixel saturated already: the second half is not needed
            if (__ballot(T > 0.0f) == 0) { all_done = true; break; }
          }
          const float4 col = gc[i];
          const float g_o = __builtin_amdgcn_exp2f(e[i]);
          const float alpha = kClamp ? fminf(kAlphaMax, g_o) : g_o;
          bool live = alpha >= kAlphaMin;
          if constexpr (kClamp) live = live & (g_o <= guard_limit_unpack(gk[i].w));   // the reference's `power > 0` skip (exp_mfma.hpp)
          const float al = live ? alpha : 0.0f;
          // A saturated pixel is latched by the SIGN of T: the splat that would take T below 1e-4 is not blended and
          // flips T negative, after which every test_T is negative too (a live pixel always has T >= 1e-4, and
          // alpha = 0 leaves test_T = T exactly).
#if SCORP_FWD_FMA
          const float test_T = __builtin_fmaf(-al, T, T);
#else
          const float test_T = T * (1.0f - al);
#endif
          const bool ok = test_T >= kTMin;
          const float ae = ok ? al : 0.0f;
          const float w = ae * T;
          if constexpr (!kScore) { C0 += col.x * w; C1 += col.y * w; C2 += col.z * w; }
          Dp += col.w * w;
          T = ok ? test_T : -fabsf(T);
          if constexpr (kForBackward) {
            const bool took = ok & live;
            lastg = took ? (uint32_t)(i + 1) : lastg;
            // (each ballot of a plain compare IS that compare's lane mask, and the AND of two 64-bit scalars is scalar work; the
            // ballot of `took` itself - an AND of two conditions - goes through a VGPR 0 / 1 and a second v_cmp: +2 VALU per hit)
            kept16 |= ((__builtin_amdgcn_ballot_w64(ok) & __builtin_amdgcn_ballot_w64(live)) != 0ull ? 1u : 0u) << i;
          }
#ifdef SCORP_FWD_STATS
          {
            const uint64_t m = __ballot(ok & live);
            st_hits += 1; st_live += (uint32_t)__builtin_popcountll(m);
            st_any += m != 0;
            st_q[0] += (m & 0x000000000F0F0F0Full) != 0; st_q[1] += (m & 0x00000000F0F0F0F0ull) != 0;
            st_q[2] += (m & 0x0F0F0F0F00000000ull) != 0; st_q[3] += (m & 0xF0F0F0F000000000ull) != 0;
            st_h[0] += (m & 0x00000000FFFFFFFFull) != 0; st_h[1] += (m & 0xFFFFFFFF00000000ull) != 0;
            st_lane += (ok & live) ? 1u : 0u;
            // lock-step cost of per-quadrant lists when the four lists are synchronised every 8 / 16 / 64 block hits
            // (the backward's matrix pass holds a fixed number of slots): sum over such windows of the largest quadrant count
            if (m != 0) {
              const uint32_t qb[4] = {(m & 0x000000000F0F0F0Full) != 0, (m & 0x00000000F0F0F0F0ull) != 0,
                                      (m & 0x0F0F0F0F00000000ull) != 0, (m & 0xF0F0F0F000000000ull) != 0};
              for (int q_ = 0; q_ < 4; q_++) { st_gq[q_] += qb[q_]; st_cq[q_] += qb[q_]; st_hq[q_] += qb[q_]; }
              st_cn += 1;
              if ((st_cn & 7) == 0) { st_g8 += max(max(st_hq[0], st_hq[1]), max(st_hq[2], st_hq[3])); st_hq[0] = st_hq[1] = st_hq[2] = st_hq[3] = 0; }
              if ((st_cn & 15) == 0) { st_g16 += max(max(st_gq[0], st_gq[1]), max(st_gq[2], st_gq[3])); st_gq[0] = st_gq[1] = st_gq[2] = st_gq[3] = 0; }
              if ((st_cn & 63) == 0) { st_g64 += max(max(st_cq[0], st_cq[1]), max(st_cq[2], st_cq[3])); st_cq[0] = st_cq[1] = st_cq[2] = st_cq[3] = 0; }
            }
            // greedy pairing of ADJACENT hits whose live lanes are disjoint (no pixel sees both: they could share an iteration)
            if (st_have && (st_prev & m) == 0) { st_pairs += 1; st_have = false; }
            else { st_prev = m; st_have = true; }
          }
#endif
        }
      }
      if constexpr (kForBackward) {
        // The block's HIT LIST keeps only the hits some pixel took.  A hit passes the exact block test when the conic's
        // minimum over the 8x8 box is below the cutoff - a minimum that may lie between pixel centres, or behind pixels that
        // are already saturated: 6.6 % of the hits of an S3 view have no taker (scripts/dev/stats_forward.py), and the
        // backward, which replays the list, would run its whole per-hit pipeline (and an atomic row of zeros' worth of
        // bookkeeping) for each.  Positions are those of the compacted list: kept hits before this group + rank in it.
        if (lastg) last = kept_n + (uint32_t)__builtin_popcount(kept16 & ((1u << lastg) - 1u));
        if (lane < nslots && ((kept16 >> lane) & 1u))   // (v_mbcnt: the kept slots below this lane's, no lane mask held in a register)
          my_hits[kept_n + __builtin_amdgcn_mbcnt_lo(kept16, 0u)] = q_id[hv + lane];
        kept_n += (uint32_t)__builtin_popcount(kept16);
        __builtin_amdgcn_sched_barrier(0);   // (keeps this tail out of the next group's exponent MFMAs: their sixteen results and
                                             // the tail's temporaries together cost the sixth wave per SIMD)
      }
      head = head + kFGroup == kFRing ? 0 : head + kFGroup;
      count -= nslots;
      consumed += (uint32_t)nslots;
      return all_done;
    };
#if SCORP_FWD_MERGE
    // full clamp-free groups: adjacent hits whose live pixels are disjoint share an iteration (blend_group_asm.hpp: the
    // sixteen slots as one hand-allocated assembly block; profiles/DESIGN_history_r01-r05.md section 8.0 (2))
    auto blend_group_merged = [&](const f32x16 &e) -> bool {
      const uint32_t gcb = (uint32_t)(size_t)(const void *)(q_col + head);   // LDS byte address of the group's first slot
      bool all_done;
      if constexpr (kForBackward) {
        uint32_t lastg = 0, kept16 = 0;
        all_done = blend_group_merged_asm_fb(e, gcb, T, C0, C1, C2, Dp, lastg, kept16);
        int hv = head;
        asm volatile("" : "+v"(hv));
        if (lastg) last = kept_n + (uint32_t)__builtin_popcount(kept16 & ((1u << lastg) - 1u));
        if (lane < kFGroup && ((kept16 >> lane) & 1u)) my_hits[kept_n + __builtin_amdgcn_mbcnt_lo(kept16, 0u)] = q_id[hv + lane];
        kept_n += (uint32_t)__builtin_popcount(kept16);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        all_done = blend_group_merged_asm_img(e, gcb, T, C0, C1, C2, Dp);
      }
      head = head + kFGroup == kFRing ? 0 : head + kFGroup;
      count -= kFGroup;
      consumed += (uint32_t)kFGroup;
      return all_done;
    };
#endif
    bool all_done = false;   // every pixel saturated: checked twice per group, not only once per 64 list entries
#if SCORP_FWD_DB
    // The exponents of the NEXT group are issued to the matrix cores before the current group is blended: the three
    // dependent MFMAs (~100 cycles before the first result can be read) then run under the blend instead of in front of it.
    if (count >= kFGroup) {
      f32x16 e = exponents(head);
      for (;;) {
        const bool more = count >= 2 * kFGroup;
        f32x16 e2 = e;
        if (more) e2 = exponents(head + kFGroup == kFRing ? 0 : head + kFGroup);
        const bool gd = consumed < hot_end ? blend_group(std::true_type{}, std::true_type{}, kFGroup, e)
                                           : blend_group(std::true_type{}, std::false_type{}, kFGroup, e);
        if (gd || __ballot(T > 0.0f) == 0) { all_done = true; break; }
        if (!more) break;
        e = e2;
      }
    }
#else
    while (count >= kFGroup) {
      const f32x16 e = exponents(head);
#if SCORP_FWD_MERGE
      const bool gd = consumed < hot_end ? blend_group(std::true_type{}, std::true_type{}, kFGroup, e) : blend_group_merged(e);
#else
      const bool gd = consumed < hot_end ? blend_group(std::true_type{}, std::true_type{}, kFGroup, e)
                                         : blend_group(std::true_type{}, std::false_type{}, kFGroup, e);
#endif
      if (gd || __ballot(T > 0.0f) == 0) { all_done = true; break; }
    }
#endif
    if (all_done) break;
    if (last_chunk && count > 0) blend_group(std::false_type{}, std::true_type{}, count, exponents(head));   // (one group per wave: not worth a variant)
    id0 = id1; a = a1; b = b1; c = c1; id1 = id2;
  }
  if constexpr (kForBackward) {
    if (lane == 0) block_hits[tile * 4 + quad] = consumed;   // (block, splat) iterations this wave ran: the P statistic
  }
#ifdef SCORP_FWD_STATS
  {
    uint32_t lm = st_lane;
    for (int off = 32; off >= 1; off >>= 1) lm = max(lm, (uint32_t)__shfl_xor((int)lm, off, 64));
    if (lane == 0) {
      atomicAdd(&g_fwd_stats[0], (unsigned long long)st_hits); atomicAdd(&g_fwd_stats[1], (unsigned long long)st_live);
      atomicAdd(&g_fwd_stats[2], (unsigned long long)st_any);
      atomicAdd(&g_fwd_stats[3], (unsigned long long)(st_q[0] + st_q[1] + st_q[2] + st_q[3]));
      atomicAdd(&g_fwd_stats[4], (unsigned long long)max(max(st_q[0], st_q[1]), max(st_q[2], st_q[3])));
      atomicAdd(&g_fwd_stats[5], (unsigned long long)(st_h[0] + st_h[1]));
      atomicAdd(&g_fwd_stats[6], (unsigned long long)max(st_h[0], st_h[1]));
      atomicAdd(&g_fwd_stats[7], (unsigned long long)lm);
      atomicAdd(&g_fwd_stats[8], (unsigned long long)st_pairs);
      atomicAdd(&g_fwd_stats[9], (unsigned long long)(st_g8 + max(max(st_hq[0], st_hq[1]), max(st_hq[2], st_hq[3]))));
      atomicAdd(&g_fwd_stats[10], (unsigned long long)(st_g16 + max(max(st_gq[0], st_gq[1]), max(st_gq[2], st_gq[3]))));
      atomicAdd(&g_fwd_stats[11], (unsigned long long)(st_g64 + max(max(st_cq[0], st_cq[1]), max(st_cq[2], st_cq[3]))));
    }
  }
#endif
  if constexpr (kScore) {
    float term = 0.0f;
    if (inside) {
      const float a_ = 1.0f - fabsf(T);
      const size_t tp = (size_t)(py % score_rows) * W + px;     // (the target is ONE hypothesis' stack of views)
      term = fabsf(a_ - tgt_alpha[tp]) + fabsf(nan_to_num00(Dp / a_) - tgt_depth[tp]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) term += __shfl_xor(term, off, 64);
    if (lane == 0) reinterpret_cast<float *>(block_hits)[tile * 4 + quad] = term;
    return;
  }
  if (inside) {
    const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
    T = fabsf(T);
    if constexpr (kForBackward) {
      final_T[pix] = T;
      n_contrib[pix] = last;
    }
    out_color[pix] = C0 + T * bg[0];
    out_color[HW + pix] = C1 + T * bg[1];
    out_color[2 * HW + pix] = C2 + T * bg[2];
    out_depth[pix] = Dp;
    out_alpha[pix] = 1.0f - T;   // = sum of the blend weights (sum_i alpha_i T_i telescopes to 1 - T)
    if (out_depth_norm) out_depth_norm[pix] = nan_to_num00(Dp / (1.0f - T));   // render()'s depth, as render_tail_kernel forms it
  }
}

// acc[j] += scale * (the per-block terms of band j), added in a FIXED order and without atomics, in two small launches
// (a band of the sweep holds 150 000 terms: one workgroup walking them is a chain of 600 dependent round trips, ~100 us):
// kScoreSlices workgroups per band each sum a contiguous slice into partial[band][slice], then one workgroup per band adds
// its kScoreSlices partial sums.
constexpr int kScoreSlices = 128;
__device__ __forceinline__ float block_sum_256(float sum, float *s_part) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
  __syncthreads();
  return (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}
__global__ void __launch_bounds__(256)
score_slices_kernel(const float *__restrict__ block_terms, int blocks_per_score, float *__restrict__ partial) {
  __shared__ float s_part[4];
  const int band = blockIdx.x / kScoreSlices, slice = blockIdx.x % kScoreSlices;
  const int per = (blocks_per_score + kScoreSlices - 1) / kScoreSlices;
  const int i0 = slice * per, i1 = min(i0 + per, blocks_per_score);
  const float *src = block_terms + (size_t)band * blocks_per_score;
  float sum = 0.0f;
  for (int i = i0 + (int)threadIdx.x; i < i1; i += 256) sum += src[i];
  const float tot = block_sum_256(sum, s_part);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256)
score_reduce_kernel(const float *__restrict__ partial, float scale, float *__restrict__ acc) {
  __shared__ float s_part[4];
  const float tot = block_sum_256(threadIdx.x < kScoreSlices ? partial[blockIdx.x * kScoreSlices + threadIdx.x] : 0.0f, s_part);
  if (threadIdx.x == 0) acc[blockIdx.x] += scale * tot;
}

int validate(const ScorpGs3dInputs *in) {
  if (!in) { set_error("inputs is NULL"); return SCORP_ERR_INVALID; }
  if (in->num_gaussians < 0 || in->image_width <= 0 || in->image_height <= 0) {
    set_error("bad sizes: N=%d W=%d H=%d", in->num_gaussians, in->image_width, in->image_height);
    return SCORP_ERR_INVALID;
  }
  if (in->image_width > 16 * 65535 || in->image_height > 16 * 65535) {
    set_error("image larger than 65535 tiles per axis"); return SCORP_ERR_INVALID;
  }
  if (in->num_gaussians > 0) {
    if (!in->means3D || !in->opacities) { set_error("means3D / opacities is NULL"); return SCORP_ERR_INVALID; }
    if (in->shs_rest && !in->shs) { set_error("shs_rest given without shs (the degree-0 block)"); return SCORP_ERR_INVALID; }
    if ((in->shs == nullptr) == (in->colors_precomp == nullptr)) {
      set_error("provide exactly one of shs / colors_precomp"); return SCORP_ERR_INVALID;
    }
    const bool sr = in->scales != nullptr && in->rotations != nullptr;
    if (sr == (in->cov3D_precomp != nullptr) || ((in->scales != nullptr) != (in->rotations != nullptr))) {
      set_error("provide exactly one of scales+rotations / cov3D_precomp"); return SCORP_ERR_INVALID;
    }
    if (in->shs) {
      if (in->sh_degree < 0 || in->sh_degree > 3) { set_error("sh_degree %d not in 0..3", in->sh_degree); return SCORP_ERR_INVALID; }
      if (in->sh_coeffs < (in->sh_degree + 1) * (in->sh_degree + 1)) {
        set_error("sh_coeffs %d < (sh_degree+1)^2", in->sh_coeffs); return SCORP_ERR_INVALID;
      }
    }
  }
  if (!in->bg || !in->viewmatrix || !in->projmatrix || !in->campos) {
    set_error("bg / viewmatrix / projmatrix / campos is NULL"); return SCORP_ERR_INVALID;
  }
  if (in->num_views < 0) { set_error("num_views %d is negative", in->num_views); return SCORP_ERR_INVALID; }
  if (in->num_views > 1) {
    const long long nt = (long long)in->num_views * in->num_gaussians, ht = (long long)in->num_views * in->image_height;
    if (in->image_height % kTile != 0) { set_error("num_views > 1 needs image_height to be a multiple of %d", kTile); return SCORP_ERR_INVALID; }
    if (nt > 0x7FFFFFFFll || ht > 16 * 65535ll) { set_error("num_views x N or num_views x H too large"); return SCORP_ERR_INVALID; }
  }
  // SH rows and quaternions are fetched as 16-byte words (and SH rows by direct global -> LDS loads)
  if ((((uintptr_t)in->shs | (uintptr_t)in->shs_rest | (uintptr_t)in->rotations) & 15) != 0) {
    set_error("shs / shs_rest / rotations must be 16-byte aligned"); return SCORP_ERR_INVALID;
  }
  return SCORP_OK;
}

}  // namespace
}  // namespace scorp

namespace scorp {
int bin_count_and_scan(const StateLayout &L, char *base, int N, int debug, hipStream_t stream) {
  uint32_t *tile_count = (uint32_t *)(base + L.tile_count);
  if (L.two_level) {
    // first level: the bins are cells (one pass, the histogram matrix is nb x cells); the totals land in tile_count[0 .. cells)
    const int per_block = (max(N, 1) + L.nb - 1) / L.nb;
    uint32_t *block_hist = (uint32_t *)(base + L.block_hist);
    {
      ProfScope prof(kKCountTiles, stream);
      count_tiles_lds_kernel<true><<<L.nb, kBinThreads, (size_t)L.cells * 4, stream>>>(
          N, per_block, L.nb, L.cells, -1, (const BinRec *)(base + L.bin), (const uint64_t *)(base + L.tile_mask), L.cells, L.cells_x,
          block_hist, (StateHeader *)(base + L.header));
      scan_block_hist_kernel<<<(L.cells + kScanTiles - 1) / kScanTiles, kScanTiles * kScanSegs, 0, stream>>>(
          L.nb, L.cells, block_hist, tile_count, (StateHeader *)(base + L.header));
    }
    SCORP_KERNEL_CHECK("count_cells", debug, stream);
    return SCORP_OK;
  }
  if (L.lds_binning) {
    const int per_block = (max(L.bin_n() >= 0 ? L.bin_n() : N, 1) + L.nb - 1) / L.nb;
    uint32_t *block_hist = (uint32_t *)(base + L.block_hist);
    {
      ProfScope prof(kKCountTiles, stream);
      const int tpp = L.tiles_per_pass();
      // histograms above 64 KiB need the kernels' dynamic-LDS limit raised.  The attribute is PER DEVICE (a process that
      // drives a second GPU must set it there too), so it is set whenever such a launch is about to happen - two cheap
      // host calls, no process-global flag, no data race between threads - and its result is checked.
      if ((size_t)tpp * 4 > 64 * 1024) {
        SCORP_HIP_CHECK(hipFuncSetAttribute((const void *)count_tiles_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsTiles * 4));
        SCORP_HIP_CHECK(hipFuncSetAttribute((const void *)scatter_pairs_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsTiles * 4));
      }
      count_tiles_lds_kernel<false><<<L.nb * L.bin_passes(), kBinThreads, (size_t)tpp * 4, stream>>>(
          N, per_block, L.nb, tpp, L.bin_n(), (const BinRec *)(base + L.bin), (const uint64_t *)(base + L.tile_mask), L.tiles, L.tiles_x,
          block_hist, L.scan_in_scatter() ? (StateHeader *)(base + L.header) : nullptr);
      scan_block_hist_kernel<<<(L.tiles + kScanTiles - 1) / kScanTiles, kScanTiles * kScanSegs, 0, stream>>>(
          L.nb, L.tiles, block_hist, tile_count, L.scan_in_scatter() ? (StateHeader *)(base + L.header) : nullptr);
    }
    SCORP_KERNEL_CHECK("count_tiles", debug, stream);
  }
  if (!L.scan_in_scatter()) {
    ProfScope prof(kKScanTiles, stream);
    if (L.tiles > 8192 && L.tiles <= kMaxLdsTiles) {
      if ((size_t)L.tiles * 4 > 64 * 1024)
        SCORP_HIP_CHECK(hipFuncSetAttribute((const void *)scan_tiles_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsTiles * 4));
      scan_tiles_lds_kernel<<<1, 1024, (size_t)L.tiles * 4, stream>>>(tile_count, (uint32_t *)(base + L.tile_start), L.tiles,
                                                                      (StateHeader *)(base + L.header));
    } else {
      scan_tiles_kernel<<<1, 1024, 0, stream>>>(tile_count, (uint32_t *)(base + L.tile_start), L.tiles,
                                                (StateHeader *)(base + L.header));
    }
  }
  SCORP_KERNEL_CHECK("scan_tiles", debug, stream);
  return SCORP_OK;
}

int bin_scatter_and_sort(const StateLayout &L, const PairLayout &P, char *base, char *pb, int N, uint32_t capacity,
                         int debug, hipStream_t stream, uint32_t *header_copy) {
  uint32_t *tile_count = (uint32_t *)(base + L.tile_count), *tile_start = (uint32_t *)(base + L.tile_start);
  uint64_t *keys = (uint64_t *)(pb + P.keys);
  uint32_t *point_list = (uint32_t *)(pb + P.list);
  StateHeader *header = (StateHeader *)(base + L.header);
  if (L.two_level) {
    // pairs -> cell buckets (first key buffer) -> tile buckets (second key buffer), then the per-tile sort reads the second
    uint32_t *cell_start = (uint32_t *)(base + L.cell_start);
    uint64_t *keys2 = (uint64_t *)(pb + P.keys2);
    {
      ProfScope prof(kKScatterPairs, stream);
      const int per_block = (max(N, 1) + L.nb - 1) / L.nb;
      scatter_pairs_lds_kernel<true><<<L.nb, kBinThreads, (size_t)L.cells * 4, stream>>>(
          N, per_block, L.nb, L.cells, -1, (const BinRec *)(base + L.bin), (const uint64_t *)(base + L.tile_mask), L.cells, L.cells_x,
          (const uint32_t *)(base + L.block_hist), cell_start, tile_count, keys, capacity, header, header_copy);
      expand_cells_kernel<<<L.cells, kExpandThreads, 0, stream>>>(cell_start, keys, keys2, tile_start, capacity, L.cells_x, L.tiles_x, L.tiles_y);
    }
    SCORP_KERNEL_CHECK("scatter_cells", debug, stream);
    {
      ProfScope prof(kKSortTiles, stream);
      sort_tiles_reg_kernel<<<L.tiles, 256, 0, stream>>>(tile_start, keys2, point_list, capacity, tile_count, header);
      sort_tiles_long_kernel<<<L.tiles < 512 ? L.tiles : 512, 256, 0, stream>>>(tile_start, keys2, point_list, capacity, tile_count, header);
    }
    SCORP_KERNEL_CHECK("sort_tiles", debug, stream);
    return SCORP_OK;
  }
  {
    ProfScope prof(kKScatterPairs, stream);
    if (L.lds_binning) {
      const int per_block = (max(L.bin_n() >= 0 ? L.bin_n() : N, 1) + L.nb - 1) / L.nb;
      const int tpp = L.tiles_per_pass();
      scatter_pairs_lds_kernel<false><<<L.nb * L.bin_passes(), kBinThreads, (size_t)tpp * 4, stream>>>(
          N, per_block, L.nb, tpp, L.bin_n(), (const BinRec *)(base + L.bin), (const uint64_t *)(base + L.tile_mask), L.tiles, L.tiles_x,
          (const uint32_t *)(base + L.block_hist), tile_start, L.scan_in_scatter() ? tile_count : nullptr, keys, capacity,
          header, header_copy);
    } else {
      scatter_pairs_kernel<<<(max(N, 1) + 255) / 256, 256, 0, stream>>>(
          N, (const BinRec *)(base + L.bin), (const uint64_t *)(base + L.tile_mask), tile_count, tile_start, L.tiles_x,
          keys, capacity, header, header_copy);
    }
  }
  SCORP_KERNEL_CHECK("scatter_pairs", debug, stream);
  {
    ProfScope prof(kKSortTiles, stream);
    // (tile_count is dead once the pairs are scattered - the next preprocess rewrites it - and holds the long tiles' ids)
    sort_tiles_reg_kernel<<<L.tiles, 256, 0, stream>>>(tile_start, keys, point_list, capacity, tile_count, header);
    sort_tiles_long_kernel<<<L.tiles < 512 ? L.tiles : 512, 256, 0, stream>>>(tile_start, keys, point_list, capacity, tile_count, header);
  }
  SCORP_KERNEL_CHECK("sort_tiles", debug, stream);
  return SCORP_OK;
}
}  // namespace scorp

// The debug entry points' view of the tile lists: tile_start[tiles + 1] in RASTER tile order with the lists concatenated in
// that order - whatever order they have in the pair buffer (cell-major under the two-level binning).
int scorp::copy_tile_lists_raster(const StateLayout &L, const PairLayout &P, const void *state, const void *pairs, uint64_t capacity,
                                  uint32_t num_pairs, uint32_t *tile_start, uint32_t *point_list, hipStream_t stream) {
  const size_t n = num_pairs < capacity ? num_pairs : (size_t)capacity;
  uint32_t *range = (uint32_t *)malloc(((size_t)L.tiles + 1) * 8), *list = (uint32_t *)malloc((n ? n : 1) * 4);
  if (!range || !list) { free(range); free(list); set_error("out of host memory"); return SCORP_ERR_INVALID; }
  hipError_t e = hipMemcpyAsync(range, (const char *)state + L.tile_start, (size_t)L.tiles * 8, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(list, (const char *)pairs + P.list, n * 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) { free(range); free(list); set_error("copy of the tile lists failed: %s", hipGetErrorString(e)); return SCORP_ERR_HIP; }
  uint32_t run = 0;
  for (int t = 0; t < L.tiles; t++) {
    const uint32_t b = range[2 * t] < n ? range[2 * t] : (uint32_t)n, en = range[2 * t + 1] < n ? range[2 * t + 1] : (uint32_t)n;
    if (tile_start) tile_start[t] = run;
    if (point_list && en > b) memcpy(point_list + run, list + b, (size_t)(en - b) * 4);
    run += en > b ? en - b : 0;
  }
  if (tile_start) tile_start[L.tiles] = run;
  free(range); free(list);
  return SCORP_OK;
}

using namespace scorp;

extern "C" size_t scorp_gs3d_state_bytes(int32_t N, int32_t W, int32_t H) { return StateLayout(N, W, H).total; }
extern "C" size_t scorp_gs3d_pairs_bytes(uint64_t capacity) { return PairLayout(capacity).total; }

extern "C" int scorp_gs3d_preprocess(const ScorpGs3dInputs *in, int32_t *out_radii, void *state, size_t state_bytes,
                                     scorp_stream_t stream) {
  return preprocess3d_impl(in, out_radii, nullptr, state, state_bytes, stream);
}

int scorp::preprocess3d_impl(const ScorpGs3dInputs *in, int32_t *out_radii, uint8_t *out_visible, void *state,
                             size_t state_bytes, scorp_stream_t stream_) {
  if (int e = validate(in)) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const int V = in->num_views > 1 ? in->num_views : 1;          // V views stacked vertically: V * N virtual Gaussians
  const int N = V * in->num_gaussians, W = in->image_width, H = V * in->image_height;
  const StateLayout L(N, W, H, false, V);
  if (V > 1 && !L.lds_binning) { set_error("num_views > 1: the stacked image has too many tiles"); return SCORP_ERR_INVALID; }
  if (!state || state_bytes < L.total || ((uintptr_t)state & 255)) {
    set_error("state buffer NULL, misaligned or too small (%zu < %zu)", state_bytes, L.total);
    return SCORP_ERR_INVALID;
  }
  if (N > 0 && !out_radii) { set_error("out_radii is NULL"); return SCORP_ERR_INVALID; }
  char *base = (char *)state;
  uint32_t *tile_count = (uint32_t *)(base + L.tile_count);
  if (!L.lds_binning) SCORP_HIP_CHECK(hipMemsetAsync(tile_count, 0, ((size_t)L.tiles + 1) * 4, stream));
  if (N > 0) {
    ProfScope prof(kKPreprocess, stream);
    launch_preprocess(in, L, (SplatRec *)(base + L.rec), (BinRec *)(base + L.bin), (uint64_t *)(base + L.tile_mask), out_radii,
                      tile_count, out_visible, stream);
    SCORP_KERNEL_CHECK("preprocess", in->debug, stream);
  }
  if (int e = bin_count_and_scan(L, base, N, in->debug, stream)) return e;
  return SCORP_OK;
}

extern "C" int scorp_gs3d_num_pairs(const void *state, scorp_stream_t stream_, uint64_t *num_pairs) {
  if (!state || !num_pairs) { set_error("state / num_pairs is NULL"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  StateHeader h;
  SCORP_HIP_CHECK(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, stream));
  SCORP_HIP_CHECK(hipStreamSynchronize(stream));
  *num_pairs = h.num_pairs;
  return SCORP_OK;
}

extern "C" int scorp_gs3d_check_overflow(const void *state, scorp_stream_t stream_, uint64_t *num_pairs) {
  if (!state) { set_error("state is NULL"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  StateHeader h;
  SCORP_HIP_CHECK(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, stream));
  SCORP_HIP_CHECK(hipStreamSynchronize(stream));
  if (num_pairs) *num_pairs = h.num_pairs;
  if (h.overflow) {
    set_error("pair buffer overflow: %u pairs needed, capacity %u", h.num_pairs, h.capacity);
    return SCORP_ERR_OVERFLOW;
  }
  return SCORP_OK;
}

int scorp::render3d_impl(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity, float *out_color,
                         float *out_depth, float *out_alpha, float *out_depth_norm, void *zero_buf, size_t zero_bytes,
                         scorp_stream_t stream_, bool for_backward, uint32_t *header_copy) {
  if (int e = validate(in)) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const int V = in->num_views > 1 ? in->num_views : 1;
  const int N = V * in->num_gaussians, W = in->image_width, H = V * in->image_height;
  const StateLayout L(N, W, H, false, V);
  const PairLayout P(capacity);
  if (V > 1 && for_backward) { set_error("num_views > 1 is forward only: use scorp_gs3d_render_image"); return SCORP_ERR_INVALID; }
  if (!state || ((uintptr_t)state & 255) || !pairs || ((uintptr_t)pairs & 255)) {
    set_error("state / pairs buffer NULL or not 256-byte aligned"); return SCORP_ERR_INVALID;
  }
  if (capacity > 0xFFFFFFFFull) { set_error("capacity above 2^32-1 pairs"); return SCORP_ERR_INVALID; }
  if (!out_color || !out_depth || !out_alpha) { set_error("output image pointer is NULL"); return SCORP_ERR_INVALID; }
  char *base = (char *)state, *pb = (char *)pairs;
  uint32_t *tile_start = (uint32_t *)(base + L.tile_start);
  uint32_t *point_list = (uint32_t *)(pb + P.list);
  if (int e = bin_scatter_and_sort(L, P, base, pb, N, (uint32_t)capacity, in->debug, stream, header_copy)) return e;
  {
    ProfScope prof(kKBlendForward, stream);
    const int blocks = ((L.tiles + 7) / 8) * 8 * 4;
    const size_t zero_total = zero_buf ? zero_bytes / 16 : 0, zero_per_wave = (zero_total + blocks - 1) / blocks;
    if (zero_buf && (((uintptr_t)zero_buf & 15) || (zero_bytes & 15) || zero_total > 0xFFFFFFFFull)) {
      set_error("render3d_impl: zero_buf must be 16-byte aligned, a multiple of 16 bytes, below 64 GiB"); return SCORP_ERR_INVALID;
    }
    auto bk = for_backward ? blend_forward_wave_kernel<true> : blend_forward_wave_kernel<false>;
    bk<<<blocks, 64, 0, stream>>>(
        tile_start, point_list, (const SplatRec *)(base + L.rec), (uint32_t)capacity, W, H, L.tiles_x, L.tiles, in->bg,
        out_color, out_depth, out_alpha, (float *)(base + L.final_T), (uint32_t *)(base + L.n_contrib),
        (uint32_t *)(pb + P.hits), (uint32_t *)(base + L.block_hits), out_depth_norm, (float4 *)zero_buf,
        (uint32_t)zero_per_wave, (uint32_t)zero_total, V > 1 ? in->image_height : 0, nullptr, nullptr, 0);
  }
  SCORP_KERNEL_CHECK("blend_forward", in->debug, stream);
  return SCORP_OK;
}

extern "C" int scorp_gs3d_render(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity,
                                 float *out_color, float *out_depth, float *out_alpha, scorp_stream_t stream) {
  return render3d_impl(in, state, pairs, capacity, out_color, out_depth, out_alpha, nullptr, nullptr, 0, stream, true);
}

extern "C" int scorp_gs3d_render_image(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity,
                                       float *out_color, float *out_depth, float *out_alpha, scorp_stream_t stream) {
  return render3d_impl(in, state, pairs, capacity, out_color, out_depth, out_alpha, nullptr, nullptr, 0, stream, false);
}

extern "C" int scorp_gs3d_render_score(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity,
                                       const float *tgt_depth, const float *tgt_alpha, int32_t rows_per_score, float scale,
                                       float *acc, scorp_stream_t stream_) {
  if (int e = validate(in)) return e;
  hipStream_t stream = (hipStream_t)stream_;
  const int V = in->num_views > 1 ? in->num_views : 1;
  const int N = V * in->num_gaussians, W = in->image_width, H = V * in->image_height;
  if (!tgt_depth || !tgt_alpha || !acc || rows_per_score <= 0 || rows_per_score % kTile != 0 || H % rows_per_score != 0) {
    set_error("scorp_gs3d_render_score: targets / acc NULL, or rows_per_score (%d) not a multiple of 16 dividing the %d rows", rows_per_score, H);
    return SCORP_ERR_INVALID;
  }
  const StateLayout L(N, W, H, false, V);
  const PairLayout P(capacity);
  if (!state || ((uintptr_t)state & 255) || !pairs || ((uintptr_t)pairs & 255)) { set_error("state / pairs NULL or misaligned"); return SCORP_ERR_INVALID; }
  if (capacity > 0xFFFFFFFFull) { set_error("capacity above 2^32-1 pairs"); return SCORP_ERR_INVALID; }
  char *base = (char *)state, *pb = (char *)pairs;
  if (int e = bin_scatter_and_sort(L, P, base, pb, N, (uint32_t)capacity, in->debug, stream, nullptr)) return e;
  const uint32_t *tile_start = (const uint32_t *)(base + L.tile_start);
  const uint32_t *point_list = (const uint32_t *)(pb + P.list);
  {
    ProfScope prof(kKBlendForward, stream);
    const int blocks = ((L.tiles + 7) / 8) * 8 * 4;
    blend_forward_wave_kernel<false, true><<<blocks, 64, 0, stream>>>(
        tile_start, point_list, (const SplatRec *)(base + L.rec), (uint32_t)capacity, W, H, L.tiles_x, L.tiles, in->bg,
        nullptr, nullptr, nullptr, nullptr, nullptr, (uint32_t *)(pb + P.hits), (uint32_t *)(base + L.block_hits), nullptr, nullptr,
        0u, 0u, V > 1 ? in->image_height : 0, tgt_depth, tgt_alpha, rows_per_score);
    // the blocks of score j are the tiles of rows [j, j + 1) * rows_per_score: contiguous tile ids, four blocks each
    const int scores = H / rows_per_score, blocks_per_score = (rows_per_score / kTile) * L.tiles_x * 4;
    float *partial = (float *)(base + L.final_T);   // (the per-pixel state of a backward pass: unused by this form)
    if ((size_t)scores * kScoreSlices > (size_t)W * H) { set_error("scorp_gs3d_render_score: too many bands for the image"); return SCORP_ERR_INVALID; }
    score_slices_kernel<<<scores * kScoreSlices, 256, 0, stream>>>((const float *)(base + L.block_hits), blocks_per_score, partial);
    score_reduce_kernel<<<scores, 256, 0, stream>>>(partial, scale, acc);
  }
  SCORP_KERNEL_CHECK("blend_forward_score", in->debug, stream);
  return SCORP_OK;
}

extern "C" int scorp_gs3d_debug_geom(const void *state, int32_t N, int32_t W, int32_t H, float *xy, float *depth,
                                     float *conic_opacity, float *rgb, int32_t *rect, scorp_stream_t stream_) {
  if (!state) { set_error("state is NULL"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  const StateLayout L(N, W, H);
  if (N <= 0) return SCORP_OK;
  SplatRec *hrec = (SplatRec *)malloc((size_t)N * sizeof(SplatRec));
  BinRec *hbin = (BinRec *)malloc((size_t)N * sizeof(BinRec));
  if (!hrec || !hbin) { free(hrec); free(hbin); set_error("host allocation failed"); return SCORP_ERR_INVALID; }
  hipError_t e = hipMemcpyAsync(hrec, (const char *)state + L.rec, (size_t)N * sizeof(SplatRec), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(hbin, (const char *)state + L.bin, (size_t)N * sizeof(BinRec), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) { free(hrec); free(hbin); set_error("debug_geom copy failed: %s", hipGetErrorString(e)); return SCORP_ERR_HIP; }
  for (int i = 0; i < N; i++) {
    const bool vis = (hbin[i].radius & kRadiusMask) != 0;
    const SplatRec z = {};
    const SplatRec &s = vis ? hrec[i] : z;
    if (xy) { xy[2 * i] = s.x; xy[2 * i + 1] = s.y; }
    if (depth) depth[i] = s.depth;
    if (conic_opacity) {   // back from the exponent form
      conic_opacity[4 * i] = s.A * (-1.0f / kConicScale); conic_opacity[4 * i + 1] = s.B * (-0.5f / kConicScale);
      conic_opacity[4 * i + 2] = s.C * (-1.0f / kConicScale); conic_opacity[4 * i + 3] = s.o;
    }
    if (rgb) { rgb[3 * i] = s.r; rgb[3 * i + 1] = s.g; rgb[3 * i + 2] = s.b; }
    if (rect) { rect[4 * i] = vis ? hbin[i].x0 : 0; rect[4 * i + 1] = vis ? hbin[i].y0 : 0; rect[4 * i + 2] = vis ? hbin[i].x1 : 0; rect[4 * i + 3] = vis ? hbin[i].y1 : 0; }
  }
  free(hrec); free(hbin);
  return SCORP_OK;
}

extern "C" int scorp_gs3d_debug_work(const void *state, int32_t N, int32_t W, int32_t H, uint64_t *out3, scorp_stream_t stream_) {
  if (!state || !out3) { set_error("NULL argument to scorp_gs3d_debug_work"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  const StateLayout L(N, W, H);
  const size_t nb = (size_t)L.tiles * 4, hw = (size_t)W * H;
  uint32_t *hb = (uint32_t *)malloc(nb * 4), *hc = (uint32_t *)malloc(hw * 4);
  if (!hb || !hc) { free(hb); free(hc); set_error("host allocation failed"); return SCORP_ERR_INVALID; }
  hipError_t e = hipMemcpyAsync(hb, (const char *)state + L.block_hits, nb * 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(hc, (const char *)state + L.n_contrib, hw * 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) { free(hb); free(hc); set_error("debug_work copy failed: %s", hipGetErrorString(e)); return SCORP_ERR_HIP; }
  uint64_t fwd = 0, bwd = 0;
  for (size_t b = 0; b < nb; b++) fwd += hb[b];
  const int tiles_x = L.tiles_x;
  for (int t = 0; t < L.tiles; t++)
    for (int q = 0; q < 4; q++) {   // the backward replays, per block, up to the deepest last contributor of its pixels
      const int bx = (t % tiles_x) * kTile + (q & 1) * 8, by = (t / tiles_x) * kTile + (q >> 1) * 8;
      uint32_t m = 0;
      for (int y = by; y < by + 8 && y < H; y++)
        for (int x = bx; x < bx + 8 && x < W; x++) m = hc[(size_t)y * W + x] > m ? hc[(size_t)y * W + x] : m;
      bwd += m;
    }
  out3[0] = fwd; out3[1] = bwd; out3[2] = (uint64_t)nb;
  free(hb); free(hc);
  return SCORP_OK;
}

extern "C" int scorp_gs3d_debug_tiles(const void *state, const void *pairs, uint64_t capacity, int32_t N, int32_t W,
                                      int32_t H, uint32_t *tile_start, uint32_t *point_list, scorp_stream_t stream_) {
  if (!state || !pairs) { set_error("state / pairs is NULL"); return SCORP_ERR_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  const StateLayout L(N, W, H);
  const PairLayout P(capacity);
  StateHeader h;
  SCORP_HIP_CHECK(hipMemcpyAsync(&h, state, sizeof(h), hipMemcpyDeviceToHost, stream));
  SCORP_HIP_CHECK(hipStreamSynchronize(stream));
  return copy_tile_lists_raster(L, P, state, pairs, capacity, h.num_pairs, tile_start, point_list, stream);
}

#ifdef SCORP_FWD_STATS
extern "C" int scorp_debug_fwd_stats(unsigned long long *out, int reset) {
  static const unsigned long long zero[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(scorp::g_fwd_stats), 96) != hipSuccess) return -2;
  return reset && hipMemcpyToSymbol(HIP_SYMBOL(scorp::g_fwd_stats), zero, 96) != hipSuccess ? -2 : 0;
}
#endif
#ifdef SCORP_FWD_TRACE
extern "C" int scorp_debug_fwd_trace(unsigned long long *out, int n_waves) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(scorp::g_fwd_trace), (size_t)n_waves * 24) == hipSuccess ? 0 : -2;
}
#endif
