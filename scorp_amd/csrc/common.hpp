// common.hpp — shared device records, workspace layout and host-side error plumbing of libscorp_gs.
// gfx950 only (wave64, 160 KiB LDS); no dual-backend paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "scorp_gs.h"

namespace scorp {

// ---- constants of the published 3DGS algorithm (named as in oracle/gs3d_oracle.c) ----
constexpr int kTile = 16;
constexpr float kNearZ = 0.2f;
constexpr float kDilation = 0.3f;
constexpr float kFovGuard = 1.3f;
constexpr float kRadiusSigma = 3.0f;
constexpr float kLambdaFloor = 0.1f;
constexpr float kAlphaMax = 0.99f;
constexpr float kLog2AlphaMax = -0.014499569695115089f;   // log2(kAlphaMax): only a splat with log2(opacity) above it can reach the clamp
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kTMin = 0.0001f;
// `alpha >= 1/255` decided from the exponent: v_exp_f32 is non-decreasing across the threshold - scripts/mb_exp_threshold.hip
// evaluates EVERY float in [-126, 0] and finds ONE step, at this value (profiles/r05_mb_exp_threshold.txt) - so
//     __builtin_amdgcn_exp2f(e) >= kAlphaMin   <=>   e >= kExp2AlphaMin        for every float e (NaN: false on both sides),
// the same bits, and the blend kernels know which pixels a hit is live on before (and without) taking the v_exp.
constexpr uint32_t kExp2AlphaMinBits = 0xC0FFD1BEu;   // -7.99435329
constexpr float kWEps = 0.0000001f;
constexpr float kDet2Eps = 0.0000001f;

// ---- device records ----
// One projected splat as the blend kernels gather it: three 16-byte loads from one 48-byte record.
// The conic is stored the way the blend exponent uses it (splat_exponent below): with k = log2(e)/2,
//   A = -k * conic_xx, B = -2k * conic_xy, C = -k * conic_yy, L = log2(opacity),  log2(opacity * G) = L + A dx^2 + B dx dy + C dy^2,
// so a hit travels from the record to a wave's LDS without per-chunk arithmetic (and could travel by LDS-DMA).
struct alignas(16) SplatRec {
  float x, y, A, B;         // pixel-space centre, scaled conic A, B
  float C, L, r, g;         // scaled conic C, log2(opacity), colour r, g
  float b, depth, kcut, o;  // colour b, view-space z, cutoff: alpha >= 1/255 only where k * q <= kcut
                            // (q = conic quadratic form, kcut = k * (2 ln(255 o) plus slack)), opacity
};
constexpr float kConicScale = 0.72134752044448170f;   // k = log2(e) / 2
static_assert(sizeof(SplatRec) == 48, "SplatRec must be 48 bytes");

// What the binning kernels stream per Gaussian (16 bytes, coalesced).
struct alignas(16) BinRec {
  uint16_t x0, y0, x1, y1;  // tile rectangle, max exclusive
  uint32_t depth_bits;      // fp32 bits of view-space z (> 0, so unsigned order == float order)
  int32_t radius;           // bits 0..26: pixel radius (0 = culled); bit 27: surfel faces away (2DGS normal flipped);
                            // bits 28..30: SH colour clamp mask (r,g,b)
};
static_assert(sizeof(BinRec) == 16, "BinRec must be 16 bytes");
constexpr int kClampShift = 28;
constexpr int32_t kRadiusMask = 0x07FFFFFF;
constexpr int kFlipBit = 27;

struct StateHeader {
  uint32_t num_pairs;  // D of the last preprocess
  uint32_t overflow;   // 1 if the last render needed more than `capacity`
  uint32_t capacity;   // capacity the last render ran with
  uint32_t long_tiles; // tiles whose list has more than 1024 entries (their ids: the dead tile_count array), see sort_tiles_*
  uint32_t _pad[12];
};
static_assert(sizeof(StateHeader) == 64, "StateHeader must be 64 bytes");

// Per-Gaussian screen-space gradient accumulator filled by the blend backward (float atomics).  Rows are 64 bytes, 40
// used: float atomics execute at the memory side in 64-byte requests (MI355X_MICROARCH.md, "Global float atomics") and
// the blend backward is bound by their rate, so a splat's ten sums must never straddle two requests (with 48-byte rows
// every second row did: 1.5 requests per (block, splat) instead of 1).
constexpr int kAccStride = 16;  // dx, dy, dA, dB, dC, dopacity, dr, dg, db, ddepth, 6 x pad

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#ifdef __HIPCC__
// Exact culling.  A splat contributes to a pixel only if alpha = o * exp(-q/2) >= 1/255, q = A dx^2 + 2B dx dy + C dy^2,
// i.e. q <= 2 ln(255 o).  This returns min q over the pixel box [bx0,bx1] x [by0,by1]: 0 if the centre is inside,
// otherwise the smallest of the four edge minima (q is convex, so the box minimum lies on an edge).  A box whose
// minimum exceeds the splat's `kcut` (2 ln(255 o) plus 1% + 0.02 of slack) cannot receive a single contributing
// pixel, so dropping the (box, splat) pair changes no output bit.  Lane-parallel: one lane tests one splat.
__device__ __forceinline__ float conic_min_over_box(float cx, float cy, float A, float B, float C, float bx0, float bx1,
                                                    float by0, float by1) {
  const float x0 = bx0 - cx, x1 = bx1 - cx, y0 = by0 - cy, y1 = by1 - cy;
  if (x0 <= 0.0f && x1 >= 0.0f && y0 <= 0.0f && y1 >= 0.0f) return 0.0f;
  const float iC = __builtin_amdgcn_rcpf(C), iA = __builtin_amdgcn_rcpf(A);   // 1 ulp: far inside the cull slack
  float best = 3.4e38f;
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const float xe = e ? x1 : x0;
    const float dy = fminf(fmaxf(-B * xe * iC, y0), y1);
    best = fminf(best, A * xe * xe + 2.0f * B * xe * dy + C * dy * dy);
    const float ye = e ? y1 : y0;
    const float dx = fminf(fmaxf(-B * ye * iA, x0), x1);
    best = fminf(best, A * dx * dx + 2.0f * B * dx * ye + C * ye * ye);
  }
  return best;
}

// log2(opacity * G) of a splat at a pixel: L + A' dx^2 + C' dy^2 + B' dx dy with (dx, dy) = splat centre - pixel, the conic
// pre-scaled by -log2(e)/2 (B' by -log2(e)) and L = log2(opacity).  Horner form, 5 VALU after dx, dy; the forward and the
// backward blend share it so that both see the same alpha bit for bit.
__device__ __forceinline__ float splat_exponent(float dx, float dy, float A, float B, float C, float L) {
  return __builtin_fmaf(__builtin_fmaf(A, dx, B * dy), dx, __builtin_fmaf(C * dy, dy, L));
}

// A float in the constant address space: uniform loads through such a pointer go to the scalar cache (s_load) instead of
// occupying VMEM slots.  Only for data no kernel of the same launch writes (camera matrices).
typedef __attribute__((address_space(4))) float CFloat;

// v + (v moved across lanes by the DPP control CTRL): the building block of the in-row reductions (gs2d.hip).
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), CTRL, 0xF, 0xF, false));
}

// Wave-wide maximum of a u32 (also of non-negative floats, through their bits), uniform result: four DPP steps inside
// each row of 16 lanes (xor 1, xor 2, half-mirror, mirror), two row broadcasts, one v_readlane.  No LDS round trips
// (the __shfl_xor form is six dependent ds_bpermute).
// CALL IT WITH ALL 64 LANES ACTIVE (wave-uniform control flow only): the result is read from lane 63, and the row
// broadcasts (row_bcast15 / row_bcast31) take their source from lanes that must have executed the steps before.  Those two
// DPP controls exist on GFX9-family targets only - this library is built for gfx950 and nothing else (build.py: ARCH).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libscorp_gs is written for gfx950 (wave64, row_bcast DPP, 160 KiB LDS): no other target is supported"
#endif
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  // old = 0 (the identity of an unsigned maximum) lets the compiler fold each move into v_max_u32_dpp; the rows are joined
  // by the two row broadcasts and one v_readlane (it was four v_readlane and three s_max behind unfolded v_mov_dpp pairs)
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror: every lane holds its row's
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast15 into rows 1, 3
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast31 into rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Tile mask convention: bit (ty - y0) * 8 + (tx - x0) for rectangles of at most 8 x 8 tiles; ~0 = whole rectangle.
constexpr uint64_t kMaskAll = ~0ull;
template <typename F>
__device__ __forceinline__ void for_each_tile_xy(int x0, int y0, int x1, int y1, uint64_t mask, F f) {   // f(tile x, tile y)
  if (mask == kMaskAll) {
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) f(x, y);
  } else {
    while (mask) {
      const int b = __builtin_ctzll(mask);
      mask &= mask - 1;
      f(x0 + (b & 7), y0 + (b >> 3));
    }
  }
}
template <typename F>
__device__ __forceinline__ void for_each_tile(int x0, int y0, int x1, int y1, uint64_t mask, int tiles_x, F f) {
  if (mask == kMaskAll) {
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) f(y * tiles_x + x);
  } else {
    while (mask) {
      const int b = __builtin_ctzll(mask);
      mask &= mask - 1;
      f((y0 + (b >> 3)) * tiles_x + x0 + (b & 7));
    }
  }
}
#endif

// Forward state layout (one caller-owned blob, 256-byte aligned sub-buffers).
constexpr int kMaxLdsTiles = 40000;   // tiles per pass of the LDS-histogram binning (156.25 KiB of the CU's 160 KiB): up to
                                      // 3200 x 3200 pixels - or the align sweep's 15 stacked 800x800 views, 37 500
                                      // tiles - in ONE pass; larger images (the align loop renders at up to 1.5^3 x
                                      // the base resolution, cameras.py:139-148) take ceil(tiles / this) passes over
                                      // the Gaussians, each pass owning a contiguous range of tiles
constexpr int kBinBlocksMax = 256;    // blocks of the LDS-histogram binning (each owns a contiguous Gaussian range)

// As many blocks as the cap allows, down to 256 Gaussians each, while the histogram matrix hist[block][tile] stays
// within 2^21 counters (8 MB written once and read twice); never fewer than one block per 4096 Gaussians.  A
// 100k-Gaussian object binned by 25 workgroups of 4096 left nine tenths of the chip idle for its count and scatter
// passes (15 + 25 us of a 180 us render at 800x800).
inline int bin_blocks(int N, int tiles) {
  const int coarse = (N + 4095) / 4096, fine = (N + 255) / 256, fit = (1 << 21) / (tiles > 0 ? tiles : 1);
  int b = fine < fit ? fine : fit;
  if (b < coarse) b = coarse;
  return b < 1 ? 1 : (b > kBinBlocksMax ? kBinBlocksMax : b);
}

// Two-level binning (round 5).  The (tile, splat) pairs of a view are bucketed FIRST by cell - kCellTiles x kCellTiles tiles,
// 64 x 64 pixels - and only then, one workgroup per cell, by tile: a bin block's pairs for one cell are a run of ~20
// consecutive 8-byte keys (they were 1.4 scattered keys per (block, tile): the scatter wrote 110 MB for 35 MB, history
// section 4.10), the block x bin histogram matrix shrinks sixteenfold, and the second level's stores stay inside one cell's
// ~40 KB.  A level-one key carries the tile's index inside its cell in bits 28..31 of its low word, which is why the path
// takes fewer than 2^28 (virtual) Gaussians; the tile lists then lie in memory in CELL-major order, so every tile has its
// own (start, end) pair: tile_start[2 t], tile_start[2 t + 1] (both paths write that form).
constexpr int kCellTiles = 4;
constexpr int kCellShift = 28;          // level-one keys: tile-in-cell index in bits 28..31 of the low word
constexpr int kMaxCells = 8192;         // the scatter's prologue scans this many bin totals (8 per thread)
#ifndef SCORP_TWO_LEVEL_MIN_N
#define SCORP_TWO_LEVEL_MIN_N 0
#endif
// Fewer cells than this: the one-level kernels.  The second level runs ONE workgroup per cell (expand_cells_kernel) and the
// first level's LDS atomics land on `cells` counters: a 256 x 256 image has 16 cells - sixteen workgroups would move every
// pair of a 100 k-Gaussian object while 240 CUs idle (the advisor's round-5 finding; profiles/r06_small_image_binning.txt
// has the measurement behind the number).
// Measured (scripts/dev/time_small_image.py, 100 k Gaussians, count + scatter [+ expand] us, two-level | one-level): 256x256
// (16 cells) 56 | 20, 384x384 (36) 46 | 19, 512x512 (64) 39 | 20, 800x800 (169) 45 | 35; S3 (1 M Gaussians, 475 cells) 48 | 52.
#ifndef SCORP_TWO_LEVEL_MIN_CELLS
#define SCORP_TWO_LEVEL_MIN_CELLS 256
#endif

// SCORP_ONE_LEVEL_BINNING=1 in the environment (read once per process): every view takes the one-level binning - what images
// beyond kMaxCells cells or 2^28 (virtual) Gaussians take anyway - so that the tests can hold that path against the oracle as well.
inline bool one_level_binning_forced() {
  static const bool forced = [] { const char *e = getenv("SCORP_ONE_LEVEL_BINNING"); return e && e[0] == '1'; }();
  return forced;
}

// (SCORP_TWO_LEVEL_MIN_CELLS in the environment overrides the compiled-in threshold: A/B runs)
inline int two_level_min_cells() {
  static const int v = [] { const char *e = getenv("SCORP_TWO_LEVEL_MIN_CELLS"); return e && e[0] ? atoi(e) : SCORP_TWO_LEVEL_MIN_CELLS; }();
  return v;
}

struct StateLayout {
  size_t header, rec, bin, tile_mask, tile_count, tile_start, cell_start, final_T, n_contrib, block_hits, block_hist, total;
  int tiles_x, tiles_y, tiles, nb;
  int cells_x, cells_y, cells;
  bool two_level;      // bins of the count / scan / scatter kernels are cells; expand_cells_kernel makes the tile buckets
  int views, view_n;   // stacked views (ScorpGs3dInputs.num_views): binning pass v owns view v's Gaussians AND tiles
  bool lds_binning;
  // (the three below describe the ONE-level binning; the two-level one is a single pass over everything)
  int bin_passes() const { return views > 1 ? views : (tiles + kMaxLdsTiles - 1) / kMaxLdsTiles; }
  int tiles_per_pass() const { return views > 1 ? tiles / views : (tiles < kMaxLdsTiles ? tiles : kMaxLdsTiles); }
  int bin_n() const { return views > 1 ? view_n : -1; }   // Gaussians a pass looks at (-1: all of them)
  // up to 8192 tiles (one pass, 8 counts per thread) every scatter workgroup scans the tile totals itself while it sets
  // up its LDS cursors, and the one-workgroup scan launch between the count and the scatter is dropped
  bool scan_in_scatter() const { return lds_binning && (two_level || (tiles <= 8192 && views <= 1)); }
  // views_ > 1: N and H are the TOTALS of views_ stacked views (N / views_ Gaussians, H / views_ rows each).  View v's
  // Gaussians only reach the tiles of band v, so the binning runs as views_ passes, pass v over view v's Gaussians with a
  // histogram of view v's tiles only: nb blocks PER VIEW and a block-histogram matrix of nb x tiles counters instead of
  // 256 x tiles (38 MB at the align sweep's 15 x 800x800).  Only nb - and with it the size of the last region - depends
  // on views_, so a state buffer sized by the one-view formula on the totals (scorp_gs3d_state_bytes) always suffices.
  StateLayout(int N, int W, int H, bool mode2d = false, int views_ = 1) {
    views = views_ > 1 ? views_ : 1;
    view_n = N / views;
    tiles_x = (W + kTile - 1) / kTile;
    tiles_y = (H + kTile - 1) / kTile;
    tiles = tiles_x * tiles_y;
    size_t n = N > 0 ? (size_t)N : 1, hw = (size_t)(W > 0 ? W : 1) * (size_t)(H > 0 ? H : 1);
    size_t off = 0;
    header = off; off = align_up(off + sizeof(StateHeader), 256);
    rec = off; off = align_up(off + n * (mode2d ? (size_t)96 : sizeof(SplatRec)), 256);
    bin = off; off = align_up(off + n * sizeof(BinRec), 256);
    tile_mask = off; off = align_up(off + n * 8, 256);   // per splat: which tiles of its (<= 8x8) rectangle it can reach
    cells_x = (tiles_x + kCellTiles - 1) / kCellTiles;
    cells_y = (tiles_y + kCellTiles - 1) / kCellTiles;
    cells = cells_x * cells_y;
    tile_count = off; off = align_up(off + ((size_t)tiles + 1) * 4, 256);
    tile_start = off; off = align_up(off + ((size_t)tiles + 1) * 8, 256);   // (start, end) per tile
    cell_start = off; off = align_up(off + ((size_t)cells + 1) * 4, 256);
    final_T = off; off = align_up(off + hw * 4 * (mode2d ? 3 : 1), 256);     // 2DGS also keeps M1, M2 per pixel
    n_contrib = off; off = align_up(off + hw * 4 * (mode2d ? 2 : 1), 256);   // ... and the median contributor
    block_hits = off; off = align_up(off + (size_t)tiles * 4 * 4, 256);      // per 8x8 block: hits the blend forward replayed
    nb = bin_blocks(N, tiles);
    lds_binning = tiles <= 64 * kMaxLdsTiles;   // (beyond: the global-atomic fallback; 150 M pixels)
    // (stacked views too: the V x N virtual Gaussians of the stacked image are binned in ONE pass - a cell that straddles two
    // views' bands is only a bucket - instead of V passes with a tile histogram each and a one-workgroup scan of all tiles)
    two_level = lds_binning && N < (1 << kCellShift) && N >= SCORP_TWO_LEVEL_MIN_N && cells <= kMaxCells &&
                cells >= two_level_min_cells() && !one_level_binning_forced();
    if (views > 1 && !two_level) {
      // (a view's blocks write only the view's segment of their histogram row, so nb is the number of blocks PER VIEW; more
      // than ~16 bought nothing at 15 views: the scatter is bound by its 8-byte pair stores, 64 blocks measured 69 vs 68 us)
      const int per_view = bin_blocks(view_n, tiles / views), cap = kBinBlocksMax / views > 16 ? kBinBlocksMax / views : 16;
      nb = per_view < cap ? per_view : cap;
    }
    block_hist = off; off = align_up(off + (lds_binning ? (size_t)nb * tiles * 4 : 0), 256);
    total = off;
  }
};

// Pair buffer: keys[capacity] u64 (depth_bits<<32 | splat) bucketed by tile, then the depth-sorted point_list[capacity]
// u32.  The keys are dead once the lists are sorted; the blend forward then leaves there, per 8x8 block, what the
// backward replays: 3DGS the block's HIT LIST hits[quad][capacity] u32 (the splats that passed the exact
// ellipse-vs-block test, in list order, compacted; 16 bytes per pair, so the region is twice the keys), 2DGS one
// verdict byte per (quad, entry).
struct PairLayout {
  size_t keys, keys2, hits, list, total;   // keys2: the tile buckets of the two-level binning (keys: its cell buckets)
  explicit PairLayout(uint64_t capacity) {
    size_t c = capacity > 0 ? (size_t)capacity : 1;
    keys = 0;
    keys2 = c * 8;
    hits = 0;
    list = align_up(c * 16, 256);
    total = align_up(list + c * 4, 256);
  }
};

// Scratch of the deterministic backward (SCORP_BACKWARD_DETERMINISTIC), 3-D and 2-D:
// [accumulator rows][pair_base N + 1][block sums of the scan][one flag byte per row][4 x capacity rows of row_floats floats]
constexpr int kPairScanBlock = 1024;   // Gaussians per workgroup of the pair-count scan (gs3d_backward.hip)
struct DetLayout {
  size_t acc, pair_base, block_sums, flags, partial, total;
  int scan_blocks;
  DetLayout(int N, uint64_t capacity, int row_floats) {
    const size_t n = N > 0 ? (size_t)N : 1, c = capacity > 0 ? (size_t)capacity : 1;
    scan_blocks = (int)((n + kPairScanBlock - 1) / kPairScanBlock);
    size_t off = 0;
    acc = off; off = align_up(off + n * row_floats * sizeof(float), 256);
    pair_base = off; off = align_up(off + (n + 1) * sizeof(uint32_t), 256);
    block_sums = off; off = align_up(off + (size_t)scan_blocks * sizeof(uint32_t), 256);
    flags = off; off = align_up(off + c * 4, 256);
    partial = off; off = align_up(off + c * 4 * row_floats * sizeof(float), 256);
    total = off;
  }
};
int launch_pair_base(int N, const BinRec *bin, const uint64_t *tile_mask, uint32_t *block_sums, uint32_t *pair_base,
                     hipStream_t stream);

// ---- in-library kernel timing (api.hip) ----
enum KernelId {
  kKPreprocess = 0, kKCountTiles, kKScanTiles, kKScatterPairs, kKSortTiles, kKBlendForward, kKBlendBackward, kKPreprocessBackward,
  kKLossForward, kKLossBackward, kKKnn, kKAdam, kKPreprocess2d, kKBlendForward2d, kKBlendBackward2d,
  kKPreprocessBackward2d, kKMapsForward2d, kKMapsBackward2d, kKNumKernels
};
extern bool g_prof_on;
extern uint64_t g_prof_mask;
void prof_begin(int kernel_id, hipStream_t stream);
void prof_end(int kernel_id, hipStream_t stream);
struct ProfScope {  // brackets one kernel launch with an event pair when profiling is enabled
  int id; hipStream_t s;
  ProfScope(int id_, hipStream_t s_) : id(id_), s(s_) { if (g_prof_on) prof_begin(id, s); }
  ~ProfScope() { if (g_prof_on) prof_end(id, s); }
};

// torch.nan_to_num(x, 0, 0): what render() applies to depth / alpha (gaussian_renderer/__init__.py:113-120)
__device__ __forceinline__ float nan_to_num00(float x) {
  if (x != x) return 0.0f;
  if (x == __builtin_inff()) return 0.0f;
  if (x == -__builtin_inff()) return -3.402823466e+38f;
  return x;
}

// ---- per-Gaussian kernels (gs3d_pergaussian.hip) ----
// out_visible (optional): radii > 0 as bytes, what scorp_gs3d_render_tail would write
void launch_preprocess(const ScorpGs3dInputs *in, const StateLayout &L, SplatRec *rec, BinRec *bin, uint64_t *tile_mask,
                       int32_t *radii, uint32_t *tile_count, uint8_t *out_visible, hipStream_t stream);
// ---- the public preprocess / render with the outputs of scorp_gs3d_render_tail produced on the way (gs3d_forward.hip):
// out_visible by the per-Gaussian kernel, out_depth_norm = nan_to_num(depth / alpha) by the blend forward's epilogue;
// either may be NULL.  scorp_gs3d_train_view uses them instead of a tail launch.
int preprocess3d_impl(const ScorpGs3dInputs *in, int32_t *out_radii, uint8_t *out_visible, void *state, size_t state_bytes,
                      scorp_stream_t stream);
// zero_buf / zero_bytes (optional, a multiple of 16 bytes): memory the blend forward's waves clear on the way (the
// backward's accumulator rows: the forward is VALU-bound and its stores are free, a separate fill is 9 us per view).
// header_copy (optional, 16 bytes): {num_pairs, overflow, capacity, long_tiles} of this render, written by the scatter
// kernel next to the state header itself - a caller that wants the overflow word on the device after the state blob is
// gone (the one-call views) gets it without a copy launch.
int render3d_impl(const ScorpGs3dInputs *in, void *state, void *pairs, uint64_t capacity, float *out_color, float *out_depth,
                  float *out_alpha, float *out_depth_norm, void *zero_buf, size_t zero_bytes, scorp_stream_t stream,
                  bool for_backward, uint32_t *header_copy = nullptr);
// Fused Adam + densification statistics in the per-Gaussian backward (scorp_gs3d_train_view with `adam`): what the kernel
// needs, already reduced to floats by the launcher.  Leaves in the order xyz, features_dc, features_rest, opacity, scaling,
// rotation; m[k] == NULL: the leaf is frozen (no update).
struct AdamEpi {
  int on;
  float *m[6], *v[6];
  float step_size[6];                     // lr / bias_correction1
  float omb1, beta2, omb2, eps, inv_sqrt_bc2;
  const uint32_t *skip;                   // nothing is updated if *skip != 0 when the kernel runs
  uint32_t *skipped_counter;              // ... and this word is incremented once instead (NULL: not counted)
  float *max_radii2D, *accum, *denom;     // per-view densification statistics (all three or none)
};
void launch_preprocess_backward(const ScorpGs3dInputs *in, const StateLayout &L, const BinRec *bin, const float *acc,
                                const ScorpGs3dGrads *grads, hipStream_t stream, const AdamEpi *adam = nullptr);
int backward3d_impl(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity, const float *dL_dcolor,
                    const float *dL_ddepth, const float *dL_dalpha, const ScorpGs3dGrads *grads, void *scratch,
                    size_t scratch_bytes, uint32_t flags, scorp_stream_t stream, const AdamEpi *adam);
int backward2d_impl(const ScorpGs3dInputs *in, const void *state, const void *pairs, uint64_t capacity, const float *dL_dcolor,
                    const float *dL_dallmap, const ScorpGs3dGrads *grads, void *scratch, size_t scratch_bytes, uint32_t flags,
                    scorp_stream_t stream, const AdamEpi *adam);
// one Adam update, the operation order of torch.optim.Adam (single-tensor, non-capturable): exp_avg.lerp_, addcmul_,
// sqrt / div / add, addcdiv_; (1 - beta) is formed in double on the host, as torch does.  Shared by adam_kernel
// (aux_kernels.hip) and the fused epilogue of preprocess_backward_kernel so that the two give the same bits.
#ifdef __HIPCC__
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float omb1, float b2, float omb2,
                                         float step_size, float inv_sqrt_bc2, float eps) {
#pragma clang fp contract(off)   // every product and sum rounded on its own, wherever the function is inlined: the two call sites
                                 // must not differ by which pairs the compiler happens to fuse into an fma
  m = m + (g - m) * omb1;
  v = v * b2 + omb2 * g * g;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
}
// The optimizer's streams are NONTEMPORAL: the two moments are touched once per iteration and by nobody else, the parameters'
// next reader (the next view's preprocess) comes ~1 ms and > 2 GB of traffic later.  Same box, S3 training iteration:
// 985 - 988 it/s with plain loads and stores, 1 070 with nontemporal ones on the SH streams alone (profiles/r06_fused_adam_step.txt).
#ifndef SCORP_ADAM_NT
#define SCORP_ADAM_NT 1   // nontemporal loads / stores on the optimizer's streams (0: plain, for A/B builds)
#endif
__device__ __forceinline__ float4 adam_ld4(const float4 *p) {
#if SCORP_ADAM_NT
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
  return make_float4(t.x, t.y, t.z, t.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ void adam_st4(float4 *p, const float4 x) {
#if SCORP_ADAM_NT
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 t; t.x = x.x; t.y = x.y; t.z = x.z; t.w = x.w;
  __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(p));
#else
  *p = x;
#endif
}
__device__ __forceinline__ float adam_ld1(const float *p) {
#if SCORP_ADAM_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void adam_st1(float *p, float x) {
#if SCORP_ADAM_NT
  __builtin_nontemporal_store(x, p);
#else
  *p = x;
#endif
}
#endif

// ---- binning shared by the 3DGS and 2DGS paths (gs3d_forward.hip) ----
int copy_tile_lists_raster(const StateLayout &L, const PairLayout &P, const void *state, const void *pairs, uint64_t capacity,
                           uint32_t num_pairs, uint32_t *tile_start, uint32_t *point_list, hipStream_t stream);
int bin_count_and_scan(const StateLayout &L, char *state_base, int N, int debug, hipStream_t stream);
int bin_scatter_and_sort(const StateLayout &L, const PairLayout &P, char *state_base, char *pairs_base, int N,
                         uint32_t capacity, int debug, hipStream_t stream, uint32_t *header_copy = nullptr);

// ---- loss (loss.hip): the exported pair, with the option of leaving the three loss values to the backward launch ----
int loss_forward_impl(const float *img, const float *gt, const float *mask, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                      float *out_loss3, void *workspace, size_t workspace_bytes, int32_t need_backward, bool finalize,
                      hipStream_t stream);
int loss_backward_impl(const float *img, const float *gt, const float *mask, int32_t C, int32_t H, int32_t W, float lambda_dssim,
                       const void *workspace, const float *grad_out, float *grad_img, float *out_loss3, hipStream_t stream);

// ---- host error plumbing ----
void set_error(const char *fmt, ...);

#define SCORP_HIP_CHECK(expr)                                                                    \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      scorp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return SCORP_ERR_HIP;                                                                      \
    }                                                                                            \
  } while (0)

// After a kernel launch: always catch launch errors; in debug mode also sync and catch execution errors.
#define SCORP_KERNEL_CHECK(name, debug, stream)                                                   \
  do {                                                                                            \
    hipError_t _e = hipGetLastError();                                                            \
    if (_e == hipSuccess && (debug)) _e = hipStreamSynchronize(stream);                           \
    if (_e != hipSuccess) {                                                                       \
      scorp::set_error("kernel %s failed: %s", name, hipGetErrorString(_e));                      \
      return SCORP_ERR_HIP;                                                                       \
    }                                                                                             \
  } while (0)

}  // namespace scorp
