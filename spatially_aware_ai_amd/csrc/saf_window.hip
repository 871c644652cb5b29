// saf_window.hip -- the windowed, voxel-major path of saf_fuse_frames: per window of 128 frames four
// classification launches (32 frames each: projection, depth test, TSDF in registers, one frame-mask plane; any grid; a frame
// and occlusion cull per brick) and one row kernel that reads and writes every touched feature row once -- in its order-free
// form (a row's samples summed in registers, one blend per row: section 4.6c; the default) or its frame-ordered form
// (SAF_WIN_FORM=rows: bit-identical to the per-frame pipeline of saf_fuse.hip).  Selected by saf_fuse_frames for calls of 16
// or more frames of one shape.  The host side schedules a call as a list of (sub-volume, window) units.
#include <chrono>
#include <vector>

#include "saf_window_dev.h"

namespace saf {
namespace {


// Pixel-major image for the windowed path, whose taps are read from global memory (L2): row p holds
// the D channels of map position p contiguously (a wave's tap load is one contiguous D*4 bytes), row P
// is the zero row of the taps outside the map.

// ------------------------------------------------------------------------------------------
// fuse, voxel-major over a WINDOW of up to 128 frames (saf_fuse_frames with many frames).
//
// A running mean is applied hit by hit, but nothing forces a row to travel to HBM between two hits.
// Per window, on the caller's stream:
//   classify_bricks_kernel  (one launch per 32 frames) every voxel against 32 frames (the full-grid sweep of
//                           clipfusion.py:647-695): TSDF running mean kept in registers across the frames
//                           and written once, one 32-bit frame mask per voxel into that launch's mask plane;
//   fuse_window_kernel      every touched voxel's D-row is read ONCE, the voxel's hits are applied -- frame-ordered form: in
//                           frame order, the same s*a + old*b with a = 1/(w+1), bit-identical to fusing the frames one
//                           after the other; order-free form: summed, then one blend -- and written ONCE.  Row bytes fall by
//                           the window's hits-per-voxel ratio (2.8 for incoherent depth, 14 for a coherent scene).
//
// fuse_window_kernel: waves work independently (no workgroup barrier after the prologue).  A wave takes
// pieces of 256 consecutive voxels, compacts the touched ones, and per chunk of <= 64 touched voxels
// (<= kHitCap hits):
//   lane-parallel over HITS  : projection, a, b, the map cell of the hit (staged in LDS);
//   lane-parallel over VOXELS: rgb / weight / label side, hit by hit;
//   rows, in sub-chunks of SR rows (<= 64 hits): the rows are brought into LDS by LDS-DMA, the hits are
//   regrouped frame-major by (frame, map cell) -- every hit of a group blends the same four map rows,
//   which are loaded from the window's map images (L2) ONCE per group, P groups in flight -- each hit
//   updates its row in LDS, and the rows are streamed back.
// ------------------------------------------------------------------------------------------
constexpr int kWinMinFrames = 16;  // shorter calls run the per-frame pipeline
#ifndef SAF_WIN_HITCAP
#define SAF_WIN_HITCAP 512  // hits of a chunk (the staging area in LDS: 6 words per hit and wave -- 48 KB per workgroup).  128 until round 5: a coherent
#endif                      // scene's chunk was then 9 rows (14 hits each) -- 1.5 sub-chunks, the lane-parallel phases a quarter full; 512: 48.3 -> 46.4 ms
                            // on scene B, 76.1 -> 75.0 on depth A (profiles/r05/row_knobs.txt)
constexpr int kHitCap = SAF_WIN_HITCAP;
static_assert(kHitCap >= kWin, "a voxel's hits of one window must fit a chunk");
constexpr int kWinThreads = 256;
constexpr int kWinWaves = kWinThreads / 64;
constexpr int kPiece = 256;

// The frames of ONE classification launch travel by value (a window's 128 x 6 pointers would not fit the 4 KB
// kernel-argument segment); the launch also files them in the window's frame table in device memory (workspace header),
// which prep_rows_kernel and the row kernel read.
struct ClsArgs {
  int n, H, W;  // frames of this launch (<= kClsFrames), image size
  int slot;     // index of the launch's first frame within the window (0, 32, 64, 96)
  int count;    // 1: this launch counts its frames in stats[2] (a window classified slab by slab counts them once)
  float guard_x, guard_y;  // SAF_CLS_GUARD: W * 2^-20, H * 2^-20 (2: always the reference's chain -- images wider than 8192)
  float mid_x, mid_y;      // (W - 1) / 2, (H - 1) / 2
  unsigned long long* verify;  // SAF_CLS_GUARD = 2 (development): disagreements of the two paths are counted here
  int tiles_x8;     // TILED instantiations: tiles per image row of the frames' tiled depth copies
  int depth_bytes;  // bytes of one depth image as the launch reads it (the padded tiled copy, or H * W * 4)
  const float* depth[kClsFrames];
  const float* rgb[kClsFrames];
  const float* pose[kClsFrames];
  const float* K[kClsFrames];
  const float* label_map[kClsFrames];
  const float* feat_map[kClsFrames];
};

__device__ __forceinline__ void file_frames(const ClsArgs& ca, WinTable* __restrict__ tab, int tid) {
  if (tab && blockIdx.x == 0 && tid < ca.n) {
    const int k = ca.slot + tid;
    tab->rgb[k] = ca.rgb[tid];
    tab->pose[k] = ca.pose[tid];
    tab->K[k] = ca.K[tid];
    tab->label_map[k] = ca.label_map[tid];
    tab->feat_map[k] = ca.feat_map[tid];
  }
}

// img_elems: elements (f32, or bf16 when `as_bf16`) from one frame's image to the next.  bf16 images (bf16 volumes in the
// order-free form): a tap is 16 bytes per 8 channels instead of 32 -- half the bytes through the L1 path that bounds the row
// kernel, and the window's table (128 frames x 36 cells x D) at 4.7 MB instead of 9.4 sits in an XCD's L2.  The features are
// rounded ONCE to the volume's precision (round to nearest even); exact when the backbone emitted bf16 (BASELINE config 3).
__global__ __launch_bounds__(256) void prep_rows_kernel(const WinTable* __restrict__ tab, void* __restrict__ imgs,
                                                        int img_elems, int D, int P, int as_bf16) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= (P + 1) * D) return;
  const float* __restrict__ feat_map = tab->feat_map[blockIdx.y];
  const int c = o % D, p = o / D;
  const float x = p < P ? feat_map[(size_t)c * P + p] : 0.0f;
  if (as_bf16)
    static_cast<uint16_t*>(imgs)[(size_t)blockIdx.y * img_elems + o] = (uint16_t)f32_to_bf16_bits(x);
  else
    static_cast<float*>(imgs)[(size_t)blockIdx.y * img_elems + o] = x;
}


// ClipSeemFusion's image side (round 6): a hit's bilinear rgb sample is four 12-byte taps in two image rows
// and its class one 4-byte tap of a third image -- 13 gather instructions per 64 hits and 3.3 distinct lines per hit from images
// no cache holds (128 x 4.9 MB per window): +50 M line fetches per window.  One packed image per frame of {r, g, b, label} pixels
// in 4 x 2-pixel tiles of a line: four 16-byte gathers and 1.9 lines.  A thread per pixel; the ragged last tile column / row is padding.
__global__ __launch_bounds__(256) void prep_rgbl_kernel(const WinTable* __restrict__ tab, float4* __restrict__ out, int H, int W,
                                                        int tiles_x, int px_pad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int f = blockIdx.y, y = i / W, x = i - y * W;
  const float* __restrict__ rgb = tab->rgb[f] + (size_t)i * 3;
  const float* __restrict__ lab = tab->label_map[f];
  out[(size_t)f * px_pad + rgbl_offset(x, y, tiles_x)] = make_float4(rgb[0], rgb[1], rgb[2], lab ? lab[i] : 0.0f);
}

#ifdef SAF_WIN_TIMING  // development aid: per-phase wave cycles of the window kernel, printed by the host
__device__ unsigned long long g_win_t[16];
#define WT_DECL unsigned long long wt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt_last_ = __builtin_readcyclecounter(), wt_c0_ = wt_last_, wt_r0_ = __builtin_amdgcn_s_memrealtime()
#define WT(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); wt_[k] += n_ - wt_last_; wt_last_ = n_; } while (0)
#define WT_FLUSH do { if (lane == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_win_t[k_], wt_[k_]); \
    if (lane == 0 && threadIdx.x < 64 && blockIdx.x == 0) { /* the shader clock over this workgroup's life: s_memtime ticks per 100 MHz tick */ \
      atomicAdd(&g_win_t[8], __builtin_readcyclecounter() - wt_c0_); atomicAdd(&g_win_t[9], __builtin_amdgcn_s_memrealtime() - wt_r0_); } } while (0)
#else
#define WT_DECL
#define WT(k)
#define WT_FLUSH
#endif

#ifndef SAF_WIN_MAPS16_BUILD
#define SAF_WIN_MAPS16_BUILD 1  // bf16 volumes, order-free form: bf16 map images (0: f32 images, the A/B)
#endif
#ifndef SAF_WIN_P2
#define SAF_WIN_P2 2  // tap groups in flight: 2 -> 133 VGPRs, so that two row-kernel waves leave room for classification waves on a SIMD (alone, 2 / 3 / 4 / 6 time the same)
#endif
#ifndef SAF_CLS_WPE
#define SAF_CLS_WPE 5  // classify_bricks_kernel: waves per SIMD the register budget is set for (78 VGPRs, no spills)
#endif
#ifndef SAF_CLS_OCCL
#define SAF_CLS_OCCL 1  // the occlusion cull of the classification (depth tile maxima); 0: the frame-wide largest depth only
#endif
#ifndef SAF_CLS_BUFLD
#define SAF_CLS_BUFLD 1  // depth gathers through a buffer descriptor of the frame's image (0: 64-bit addresses)
#endif
#ifndef SAF_CLS_GUARD
#define SAF_CLS_GUARD 1  // the voxel's pixel from the quotients themselves wherever no lane is near a rounding boundary (0: always the reference's chain; 2: both, disagreements counted)
#endif
#ifndef SAF_CLS_GUARD_EPS
#define SAF_CLS_GUARD_EPS 0x1p-20f  // the guard band per pixel of image width / height (the chain's error bound is 11.1 * 2^-24 = 0.69 of it)
#endif
#ifndef SAF_CLS_BOX
#define SAF_CLS_BOX 1  // the brick's frame cull tests the box's extents (0: its bounding sphere, rounds 2-3)
#endif
#ifndef SAF_CLS_SKIPDEAD
#define SAF_CLS_SKIPDEAD 0  // 1: depth gathers of voxel slots without a single pixel in the wave are not issued.  Measured SLOWER (round 6, same box,
                            // profiles/r06/classification_skipdead_ab.txt: job 75.26 -> 75.7 ms, config 3 69.5 -> 70.0): the ballot and branch per slot cost
                            // the vector-bound kernel more than the skipped requests save the texture-address path
#endif
#ifndef SAF_CLS_FU
#define SAF_CLS_FU 1   // frames classified together: with the frame cull, occupancy hides the depth gathers better than batching does (1: 1.13 ms, 2: 1.17, 4: 1.29, 8: 2.08 per launch)
#endif
#ifndef SAF_CLS_CUBE
#define SAF_CLS_CUBE 0  // 1: voxel j of a lane lies in the brick's j-th 4 x 4 x 4 cube (z = 4 j + lane / 16), so that the 64 depth gathers of one
                        // instruction land in a cube's pixel footprint instead of the whole 16-voxel column's.  Measured (round 5,
                        // profiles/r05/classification_variants.txt): no difference, 76.4 vs 76.6 ms per job -- a lane keeps 4 consecutive z
#endif
#ifndef SAF_CLS_ABL
#define SAF_CLS_ABL 0  // development (same results): 1 = every depth gather issued twice (the second one's pixel mirrored in its row):
#endif                 // doubles the classification's load on the texture-address unit and nothing else; 2 = the projection's vector
                       // arithmetic twice, no memory request (profiles/r05/classification_variants.txt)
#ifndef SAF_WIN_SPLIT_LOG2
#define SAF_WIN_SPLIT_LOG2 2  // a piece is handed out in 2^k parts (quarters measured best: halves 6967, quarters 7165, eighths 6049 frames/s on the coherent scene)
#endif
#ifndef SAF_WIN_SR2
#define SAF_WIN_SR2 5  // 5-row sub-chunks: 71 KB of LDS per workgroup, two of them leave room for a classification workgroup
#endif
#ifndef SAF_WIN_WPE
#define SAF_WIN_WPE 2
#endif
#ifndef SAF_WIN_OF_SR2
#define SAF_WIN_OF_SR2 6  // order-free form, D = 512: rows of a sub-chunk = accumulator sets in registers (8 VGPRs each).  6: 167 VGPRs (f32) / 176 (bf16):
                          // two row waves still leave a SIMD room for two classification waves (2 x 176 + 2 x 80 = 512); job 76.1 / 75.4 / 75.1 ms for 5 / 6 / 7
                          // (7: 176 / 189 VGPRs -- the bf16 kernel would push the classification down to one wave)
#endif
#ifndef SAF_WIN_OF_P2
#define SAF_WIN_OF_P2 2
#endif
#ifndef SAF_WIN_LABEL_RUNS
#define SAF_WIN_LABEL_RUNS 1  // label histogram: one add per run of equal classes of a voxel's hits (0: one atomic per hit, rounds 1-4)
#endif
#ifndef SAF_WIN_ABL
#define SAF_WIN_ABL 0  // development: ablations of the order-free kernel (WRONG results): 1 taps "outside" (instructions issue, no
#endif                 // request), 2 no tap instructions, 4 no row loads, 8 no row stores, 16 no multiply-adds
#ifndef SAF_WIN_OF_LDPOL
#define SAF_WIN_OF_LDPOL 2  // cache policy of the order-free kernel's row loads / stores (aux bits of the buffer instructions: 2 = nt)
#endif
#ifndef SAF_WIN_OF_STPOL
#define SAF_WIN_OF_STPOL 2
#endif
#ifndef SAF_WIN_OF_WPE
#define SAF_WIN_OF_WPE 2  // waves per SIMD the order-free kernel's register budget is set for
#endif
constexpr int kUnitVox = kPiece >> SAF_WIN_SPLIT_LOG2;  // voxels of a unit of work (a quarter piece)
// OF = the order-free form of the row kernel: a row's samples of the window are summed in
// REGISTERS and the row is blended once, (w0 old + sum) / (w0 + k); no row lives in LDS.
#ifndef SAF_WIN_OF_SR2_BF16
#define SAF_WIN_OF_SR2_BF16 5  // ... of a bf16 volume: 5 rows -- 168 registers instead of 178: two row waves (registers come in blocks of eight) then leave a
                               // SIMD room for TWO classification waves of 80, with six (184 each) for one; config 3's job 66.0 -> 65.2 ms, the f32 kernel
                               // (168 registers with six rows) loses with five: 75.1 -> 75.8 (profiles/r06/config3_rgbl_ab.txt)
#endif
template <int CPL, bool OF = false, bool BF16 = false>
struct WinCfg {
  static constexpr int SR = OF ? (CPL == 1 ? 8 : (CPL == 2 ? (BF16 ? SAF_WIN_OF_SR2_BF16 : SAF_WIN_OF_SR2) : (CPL == 3 ? 3 : 2)))
                               : (CPL == 1 ? 8 : (CPL == 2 ? SAF_WIN_SR2 : 3));  // rows of a sub-chunk (LDS resident / register sums)
  static constexpr int P = OF ? (CPL == 1 ? 4 : (CPL == 2 ? SAF_WIN_OF_P2 : 1))
                              : (CPL == 1 ? 6 : (CPL == 2 ? SAF_WIN_P2 : 2));  // tap groups in flight
  // dynamic LDS layout (bytes)
  static constexpr size_t rows_off = 0;
  static constexpr size_t rows_bytes = OF ? 0 : (size_t)kWinWaves * SR * CPL * 64 * sizeof(float4);
  static constexpr size_t stage_off = rows_off + rows_bytes;  // 6 arrays of kHitCap words per wave
  static constexpr size_t stage_bytes = (size_t)kWinWaves * 6 * kHitCap * 4;
  static constexpr size_t tm_off = stage_off + stage_bytes;
  static constexpr size_t tm_bytes = (size_t)kWinWaves * kUnitVox * 4 * kMaskWords;  // a unit's touched voxels: masks
  static constexpr size_t tv_off = tm_off + tm_bytes;
  static constexpr size_t tv_bytes = (size_t)kWinWaves * kUnitVox * 2;  // ... and local ids
  static constexpr size_t ptr_off = tv_off + tv_bytes;
  static constexpr size_t ptr_bytes = (size_t)2 * kWin * sizeof(const float*);
  static constexpr size_t cam_off = ptr_off + ptr_bytes;
  static constexpr size_t total = cam_off + (size_t)kWin * sizeof(Cam);
};

// Classification of one piece (256 consecutive voxels, a lane owns 4 of them) against every frame of a
// window (clipfusion.py:647-679): the voxels' TSDF running mean is kept in registers across the frames
// (clipfusion.py:681-695 with B = 1, frame after frame -- order dependent) and written once; mk4[j] collects
// the frame bitmask of voxel j.  KFU frames at a time: their depth gathers are in flight together.
// The classification of a lane's 4 consecutive voxels (flat indices nb .. nb+3, world coordinates xw/yw/zw,
// inb = inside the grid) against the frames of `live` (bit k = frame k of the launch), ascending.
// Where pixel (x, y) of a frame's depth image lies (in floats).  TILED: in the window's re-laid-out copy, 4 x 8-pixel tiles of
// 128 bytes -- a brick's footprint in a frame is a blob of a few dozen pixels across SEVERAL image rows, and row-major every one
// of those rows is a cache line of its own: 37 lines per (brick, frame) against 17 tiled (a simulation of the benchmark's
// geometry), i.e. half the classification's requests on the L2 -> L1 path it shares with the row kernel.
#ifndef SAF_CLS_TILE_WL2
#define SAF_CLS_TILE_WL2 2  // log2 of the tile's width in pixels; its height is 32 / width (a tile = one 128-byte line).  4 x 8: a brick is 16
                            // voxels tall and scans are gravity-aligned, so its footprint is taller than wide -- 17 lines per (brick, frame)
                            // against 19 for 8 x 4, 25 for 16 x 2, 37 row-major; job 73.5 / 74.0 / 74.2 / 75.8 ms (profiles/r05/depth_tiles.txt)
#endif
constexpr int kTileWL2 = SAF_CLS_TILE_WL2, kTileHL2 = 5 - SAF_CLS_TILE_WL2;
template <bool TILED>
__device__ __forceinline__ int depth_offset(int x, int y, int width, int tiles_x8) {
  return TILED ? (((y >> kTileHL2) * tiles_x8 + (x >> kTileWL2)) << 5) + ((y & ((1 << kTileHL2) - 1)) << kTileWL2) + (x & ((1 << kTileWL2) - 1))
               : y * width + x;
}
// nearest_index (saf_common.h) with the pixel kept as (x, y): offset in the launch's depth layout, or -1
template <bool TILED>
__device__ __forceinline__ int nearest_offset(float gx, float gy, const Cam& c, int width, int tiles_x8) {
  const float xn = __builtin_rintf(unnormalize(gx, c.sfx)), yn = __builtin_rintf(unnormalize(gy, c.sfy));
  const bool inb = (xn > -1.0f) && (xn < c.fw) && (yn > -1.0f) && (yn < c.fh);
  return inb ? depth_offset<TILED>((int)xn, (int)yn, width, tiles_x8) : -1;
}
template <int KFU, bool SUM, bool VERIFY = false, bool TILED = false>
__device__ __forceinline__ void classify_voxels(const KVol& v, const ClsArgs& wa, const Cam* __restrict__ s_cam, uint32_t nb,
                                                const float (&xw)[4], const float (&yw)[4], const float (&zw)[4],
                                                const bool (&inb)[4], float rtrunc, bool tsdf_aligned,
                                                uint32_t live, uint32_t (&mk4)[4], unsigned long long& nt_done,
                                                unsigned long long& tsdf_rows_done) {
  float told[4];
  int tw[4];
  constexpr uint32_t kZS = SAF_CLS_CUBE ? 4u : 1u;  // flat-index distance of a lane's voxels j and j + 1
  const bool vec = kZS == 1u && tsdf_aligned && inb[3] && (nb & 3u) == 0u;
  if (vec) {
    const float4 t4 = *reinterpret_cast<const float4*>(v.tsdf + nb);
    const int4 w4 = *reinterpret_cast<const int4*>(v.tsdf_w + nb);
    told[0] = t4.x; told[1] = t4.y; told[2] = t4.z; told[3] = t4.w;
    tw[0] = w4.x; tw[1] = w4.y; tw[2] = w4.z; tw[3] = w4.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      told[j] = inb[j] ? v.tsdf[nb + kZS * j] : 0.0f;
      tw[j] = inb[j] ? v.tsdf_w[nb + kZS * j] : 0;
    }
  }
  uint32_t touched = 0;  // bit j: voxel j's TSDF changed
  // kFU frames at a time: all their depth gathers are in flight together, then the frames are
  // applied one after the other (the TSDF running mean is order dependent)
  constexpr int kFU = KFU;
  while (live) {
    int fr[kFU];  // local frame index (bit of `live`), -1 past the end
#pragma unroll
    for (int u = 0; u < kFU; ++u) {
      fr[u] = live ? __ffs((int)live) - 1 : -1;
      live &= live - (live ? 1u : 0u);
    }
    int pix[kFU][4];  // >= 0: pixel; -1: in view, no pixel (zeros padding); -2: not in view
    float pz[kFU][4];
#pragma unroll
    for (int u = 0; u < kFU; ++u) {
      const bool on = fr[u] >= 0;
      const Cam cam = s_cam[on ? fr[u] : 0];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#if SAF_CLS_GUARD
        // The voxel's pixel without the normalise / un-normalise round trip (clipfusion.py:654-661 and grid_sample's own
        // arithmetic: +0.5, /W, *2, -1, +1, *W/2, -0.5, round): that chain returns the quotient qu = u / z to within
        // W * 11.1 * 2^-24 (every rounding of it at its largest, |qu| <= 2 W; beyond that the voxel is out of view by a margin no
        // rounding reaches), so wherever qu and qv are farther than W * 2^-20 / H * 2^-20 from every half-integer, the pixel is
        // (rint(qu), rint(qv)), "in view" is -0.5 < qu < W - 0.5 (both ends are half-integers) and an in-view pixel is inside
        // the image.  If ANY lane of the wave is closer than that (14 % of the voxel slots at 640 x 480) the whole wave takes
        // the reference's chain for this voxel ...
        const Uvz hq = project_uvz(cam, xw[j], yw[j], zw[j]);
#if SAF_CLS_ABL & 2  // the guarded path's arithmetic a second time on a shifted point, result kept alive: twice the vector work, no memory request
        {
          const Uvz h2 = project_uvz(cam, xw[j] + 0.25f, yw[j] - 0.25f, zw[j] + 0.125f);
          const float r2 = __builtin_amdgcn_rcpf(h2.z);
          const float a2 = h2.u * r2, b2 = h2.v * r2;
          const float c2 = __builtin_rintf(a2), d2 = __builtin_rintf(b2);
          const bool n2 = fabsf(a2 - c2) > 0.5f - __builtin_fmaf(fabsf(a2), 0x1p-21f, wa.guard_x) ||
                          fabsf(b2 - d2) > 0.5f - __builtin_fmaf(fabsf(b2), 0x1p-21f, wa.guard_y);
          const bool v2 = fabsf(a2 - wa.mid_x) < cam.sfx && fabsf(b2 - wa.mid_y) < cam.sfy && (h2.z > 0.0f);
          const int p2 = (v2 && !n2) ? (int)d2 * wa.W + (int)c2 : -2;
          asm volatile("" ::"v"(p2));
        }
#endif
        // ... and, on the guarded path, without the two IEEE divisions either: au = u * rcp(z) is within |qu| * 0.75 * 2^-22 of
        // the quotient (v_rcp_f32: 1 ulp; one rounding of the product), so the band is widened by |au| * 2^-21.  (z below
        // 2^-100 -- the camera inside the voxel -- takes the reference's path: rcp overflows there.)
        const float rz = __builtin_amdgcn_rcpf(hq.z);
        const float au = hq.u * rz, av = hq.v * rz;
        const float ru = __builtin_rintf(au), rv = __builtin_rintf(av);
        const bool near = fabsf(au - ru) > 0.5f - __builtin_fmaf(fabsf(au), 0x1p-21f, wa.guard_x) ||
                          fabsf(av - rv) > 0.5f - __builtin_fmaf(fabsf(av), 0x1p-21f, wa.guard_y) ||
                          fabsf(hq.z) < 0x1p-100f;  // (NaN: not near, and not in view below)
        int pixel;
        if (__builtin_amdgcn_ballot_w64(near) != 0ull) {
          const Proj p = finish_from_uv(cam, hq.u / hq.z, hq.v / hq.z, hq.z);
          const bool in_view = on && inb[j] && (fabsf(p.gx) <= 1.0f) && (fabsf(p.gy) <= 1.0f) && (p.z > 0.0f);
          const int px = nearest_offset<TILED>(p.gx, p.gy, cam, wa.W, wa.tiles_x8);
          pixel = in_view ? (px >= 0 ? px : -1) : -2;
        } else {
          const bool in_view = on && inb[j] && fabsf(au - wa.mid_x) < cam.sfx && fabsf(av - wa.mid_y) < cam.sfy && (hq.z > 0.0f);
          pixel = in_view ? depth_offset<TILED>((int)ru, (int)rv, wa.W, wa.tiles_x8) : -2;
        }
        if constexpr (VERIFY || SAF_CLS_GUARD > 1) {  // the self-check (SAF_CLS_VERIFY=1 at run time): both paths, disagreements counted in stats[7]
          const Proj p = finish_from_uv(cam, hq.u / hq.z, hq.v / hq.z, hq.z);
          const bool in_view = on && inb[j] && (fabsf(p.gx) <= 1.0f) && (fabsf(p.gy) <= 1.0f) && (p.z > 0.0f);
          const int px = nearest_offset<TILED>(p.gx, p.gy, cam, wa.W, wa.tiles_x8);
          const int want = in_view ? (px >= 0 ? px : -1) : -2;
          if (want != pixel && wa.verify) atomicAdd(wa.verify, 1ull);
        }
        pix[u][j] = pixel;
        pz[u][j] = hq.z;
        continue;
#endif
        const Proj p = project(cam, xw[j], yw[j], zw[j]);
        const bool in_view = on && inb[j] && (fabsf(p.gx) <= 1.0f) && (fabsf(p.gy) <= 1.0f) && (p.z > 0.0f);
        const int px = nearest_offset<TILED>(p.gx, p.gy, cam, wa.W, wa.tiles_x8);
        pix[u][j] = in_view ? (px >= 0 ? px : -1) : -2;
        pz[u][j] = p.z;
      }
    }
    float depth[kFU][4];
#pragma unroll
    for (int u = 0; u < kFU; ++u) {
      // through a buffer descriptor of the frame's depth image (the frame is wave-uniform): a 32-bit offset per lane instead of
      // a 64-bit address, and "no pixel" (pix < 0: an offset beyond the image) reads 0 by the range check -- zeros padding
#if SAF_CLS_BUFLD
      const __amdgpu_buffer_rsrc_t dimg = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(wa.depth[fr[u] >= 0 ? fr[u] : 0]), 0, wa.depth_bytes, 0x00020000);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#if SAF_CLS_SKIPDEAD
        // a voxel slot no lane of the wave has a pixel for (a brick at the edge of the view, z <= 0) issues no gather: an
        // instruction whose 64 offsets all lie beyond the image makes no memory request but still takes its turn in the
        // texture-address unit, the path the classification shares with the row kernel beside it
        depth[u][j] = 0.0f;
        if (__builtin_amdgcn_ballot_w64(pix[u][j] >= 0) != 0ull)
#endif
        depth[u][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dimg, pix[u][j] * 4, 0, 0));
      }
#if SAF_CLS_ABL & 1
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = pix[u][j] / wa.W, col = pix[u][j] - row * wa.W;
        const int twin = pix[u][j] >= 0 && !TILED ? row * wa.W + (wa.W - 1 - col) : pix[u][j];
        const float d2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dimg, twin * 4, 0, 0));
        asm volatile("" ::"v"(d2));
      }
#endif
#else
      const float* __restrict__ dimg = wa.depth[fr[u] >= 0 ? fr[u] : 0];
#pragma unroll
      for (int j = 0; j < 4; ++j) depth[u][j] = pix[u][j] >= 0 ? dimg[pix[u][j]] : 0.0f;
#endif
    }
#pragma unroll
    for (int u = 0; u < kFU; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in_view = pix[u][j] != -2;
        const float num = depth[u][j] - pz[u][j];
        const float sdf = num == INFINITY ? INFINITY : div_by_uniform(num, v.trunc, rtrunc);
        if (in_view && fabsf(sdf) <= 1.0f) mk4[j] |= 1u << fr[u];
        if (in_view && sdf > -1.0f) {
          const float t = sdf > 1.0f ? 1.0f : sdf;
          const int w1 = tw[j] + 1;
          if (SUM) {
            told[j] = told[j] + t;
          } else {
            const float rw = __builtin_amdgcn_rcpf((float)w1);
            told[j] = t * rw + told[j] * ((float)tw[j] * rw);
          }
          tw[j] = w1;
          touched |= 1u << j;
          ++nt_done;
        }
      }
    }
  }
  if (touched) {
    tsdf_rows_done += (unsigned long long)__popc(touched);
    if (vec) {
      *reinterpret_cast<float4*>(v.tsdf + nb) = make_float4(told[0], told[1], told[2], told[3]);
      *reinterpret_cast<int4*>(v.tsdf_w + nb) = make_int4(tw[0], tw[1], tw[2], tw[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (touched & (1u << j)) {
          v.tsdf[nb + kZS * j] = told[j];
          v.tsdf_w[nb + kZS * j] = tw[j];
        }
      }
    }
  }
}

// Counters of a classification launch.  One atomic per WAVE on stats[1] / stats[6] would be 131 k atomics on
// two addresses per launch -- they serialise in L2 and took 0.7 ms of a 1.8 ms kernel.  The four waves of a
// workgroup are summed in LDS and added to one of 64 shards in the workspace header; the window's row kernel
// folds the shards into stats[].
// `cull`: the wave's (brick, frame) pairs {tested, behind, far, outside the frustum, occluded} (wave-uniform; round 6): the same
// route, 16 shards of five words behind the frame table -- thread k + 1 of the workgroup adds word k.
__device__ __forceinline__ void cls_accumulate(unsigned long long nt_done, unsigned long long tsdf_rows_done, const uint32_t (&cull)[kCullWords],
                                               int lane, int wave, unsigned long long (&s_acc)[4][2], uint32_t (&s_cull)[4][kCullWords],
                                               unsigned long long* __restrict__ cls_acc) {
  for (int o = 32; o > 0; o >>= 1) {
    nt_done += __shfl_xor(nt_done, o);
    tsdf_rows_done += __shfl_xor(tsdf_rows_done, o);
  }
  if (lane == 0) {
    s_acc[wave][0] = nt_done;
    s_acc[wave][1] = tsdf_rows_done;
#pragma unroll
    for (int k = 0; k < kCullWords; ++k) s_cull[wave][k] = cull[k];
  }
  __syncthreads();
  if (threadIdx.x == 0 && cls_acc) {
    const unsigned long long a = s_acc[0][0] + s_acc[1][0] + s_acc[2][0] + s_acc[3][0];
    const unsigned long long b = s_acc[0][1] + s_acc[1][1] + s_acc[2][1] + s_acc[3][1];
    unsigned long long* shard = cls_acc + 2 * (blockIdx.x % kClsShards);
    if (a) atomicAdd(&shard[0], a);
    if (b) atomicAdd(&shard[1], b);
  }
  if (threadIdx.x >= 1 && threadIdx.x <= kCullWords && cls_acc) {
    const int k = threadIdx.x - 1;
    const unsigned long long c = (unsigned long long)s_cull[0][k] + s_cull[1][k] + s_cull[2][k] + s_cull[3][k];
    unsigned long long* cs = cls_acc + (kCullAccOff - kClsAccOff) / sizeof(unsigned long long);
    if (c) atomicAdd(&cs[k * kCullShards + (blockIdx.x % kCullShards)], c);
  }
}

// ---- the classification: one launch per 32 frames of a window (one mask plane), a wave per 4 x 4 x 16 brick ----
// ANY grid: bricks at the upper faces are partial (their lanes outside the grid are masked), the brick columns are walked
// in 8 x 8 tiles over the brick grid padded to whole tiles (a wave whose brick lies in the padding has nothing to do).
// (Rounds 1-3 had a second kernel over linear 256-voxel pieces for grids that do not tile into bricks -- the reference's own
// 127 x 104 x 116, 61 x 60 x 59, ... --: no frame cull, 21 spilled registers.)
// (One launch over all 64 frames of a window would keep the TSDF in registers twice as long but put 64 depth-image
// footprints in L2 at once: 4.2 ms against 2 x 1.9 ms.)
// A z-column piece is almost never outside a frame's view as a whole; a compact brick often is -- outside
// the frustum, or farther than anything the frame has seen (z beyond the image's largest depth + trunc:
// neither valid nor tsdf_valid).  The wave tests its brick against the launch's 32 frames lane-parallel
// (lane k tests frame k: bounding sphere against the frustum planes and the depth bound, generous margins)
// and classifies only the frames that survive; the arithmetic of the surviving frames is unchanged.
constexpr int kBrickX = 4, kBrickY = 4, kBrickZ = 16;

constexpr int kMaxDepthTiles = 4096;  // tiles per frame of the depth pyramid's one level (tile side 16 px, doubled until they fit)
// workspace: the tile maxima of up to kTileWindows windows of a call (a window's slabs, and the same window of later slabs of a
// slab-by-slab call, reuse them: 128 depth launches of a 512-frame job in eight slabs were 4.7 % of it), each window: 4 KB for
// the frames' largest / smallest tile maximum, then kWin x kMaxDepthTiles floats
constexpr int kTileWindows = 4;  // (8 until round 5: with the tiled depth copies a slot is 160 MB at 640 x 480; a 512-frame call has four windows)
// ... and, when the workspace was sized for the frames' image size (saf_fuse_workspace_bytes_for_frames), the window's depth
// images re-laid-out in tiles of one cache line (depth_px_pad floats per frame; 0: the classification reads the frames' own images)
constexpr size_t tile_win_bytes(size_t depth_px_pad) { return 4096 + (size_t)kWin * kMaxDepthTiles * sizeof(float) + (size_t)kWin * depth_px_pad * sizeof(float); }
inline size_t depth_px_padded(int H, int W) {
  return (size_t)((H + (1 << (5 - SAF_CLS_TILE_WL2)) - 1) >> (5 - SAF_CLS_TILE_WL2)) * (size_t)((W + (1 << SAF_CLS_TILE_WL2) - 1) >> SAF_CLS_TILE_WL2) * 32;
}

// Largest depth of every frame of a launch, and of every tile of 2^ts_log2 x 2^ts_log2 pixels of it: max(depth, 0), NaN
// ignored, +inf kept.  dmax_bits[k] (non-negative floats order like their bit patterns: an integer atomicMax) feeds the
// coarse cull of a brick against a frame, tmax[k][tile] the fine one: a brick whose nearest voxel centre lies farther than
// trunc behind the largest depth of the pixels it projects onto is OCCLUDED in that frame -- sdf < -1 for every voxel,
// neither valid nor tsdf_valid (clipfusion.py:669-679) -- and is not classified at all.  One wave per tile.
// (dmax[k] / dmax[kClsFrames + k], the largest / the SMALLEST of frame k's tile maxima, come from depth_reduce_kernel: a
// brick nearer than the smallest tile maximum + trunc cannot be occluded anywhere in the frame and skips the tile lookups.
// A first form added every tile's maximum to the two words of its frame with atomics: 2400 atomics per address and frame
// serialise in L2 -- 0.3 ms per launch, more than the whole classification of a 128^3 grid.)
// `tiled` (optional): the frames' depth images re-laid-out in tiles (depth_offset<true>), `img_pad` floats per frame --
// this kernel reads every pixel of the window's depth images once anyway.
__global__ __launch_bounds__(256) void depth_max_kernel(ClsArgs wa, int ts_log2, int tiles_x, int n_tiles, float* __restrict__ tmax,
                                                        float* __restrict__ tiled, int tiles_x8, int img_pad) {
  const int lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  const float* __restrict__ d = wa.depth[blockIdx.y];
  float* __restrict__ td = tiled ? tiled + (size_t)blockIdx.y * img_pad : nullptr;
  const int ts = 1 << ts_log2, tx = tile % tiles_x, ty = tile / tiles_x;
  float m = 0.0f;
  // four loads in flight per lane (a 16 x 16 tile is exactly one round)
  for (int i0 = lane; i0 < ts * ts; i0 += 256) {
    float x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + 64 * k;
      const int px = (tx << ts_log2) + (i & (ts - 1)), py = (ty << ts_log2) + (i >> ts_log2);
      const bool ok = i < ts * ts && px < wa.W && py < wa.H;
      x[k] = ok ? d[(size_t)py * wa.W + px] : 0.0f;
      if (ok && td) td[depth_offset<true>(px, py, wa.W, tiles_x8)] = x[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) m = x[k] > m ? x[k] : m;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_xor(m, o);
    m = t > m ? t : m;
  }
  if (lane == 0) tmax[(size_t)blockIdx.y * kMaxDepthTiles + tile] = m;
}
// one workgroup per frame: the largest and the smallest of its tile maxima
__global__ __launch_bounds__(256) void depth_reduce_kernel(const float* __restrict__ tmax, int n_tiles, float* __restrict__ dmax) {
  __shared__ float s_hi[4], s_lo[4];
  const float* __restrict__ t = tmax + (size_t)blockIdx.x * kMaxDepthTiles;
  float hi = 0.0f, lo = INFINITY;
  for (int i = threadIdx.x; i < n_tiles; i += 256) {
    const float x = t[i];
    hi = x > hi ? x : hi;
    lo = x < lo ? x : lo;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float a = __shfl_xor(hi, o), b = __shfl_xor(lo, o);
    hi = a > hi ? a : hi;
    lo = b < lo ? b : lo;
  }
  if ((threadIdx.x & 63) == 0) {
    s_hi[threadIdx.x >> 6] = hi;
    s_lo[threadIdx.x >> 6] = lo;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    dmax[blockIdx.x] = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    dmax[kWin + blockIdx.x] = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
  }
}

template <bool SUM, bool VERIFY = false, bool TILED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SAF_CLS_WPE))) void classify_bricks_kernel(
    KVol v, ClsArgs wa, const float* __restrict__ dmax, const float* __restrict__ tmax, int ts_log2, int tiles_x,
    uint32_t* __restrict__ hitmask,
    unsigned long long* __restrict__ stats, unsigned long long* __restrict__ cls_acc, WinTable* __restrict__ tab) {
  __shared__ Cam s_cam[kClsFrames];
  __shared__ unsigned long long s_acc[4][2];
  __shared__ uint32_t s_cull[4][kCullWords];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_f = wa.n;
  if (tid < n_f) s_cam[tid] = load_cam(wa.pose[tid], wa.K[tid], wa.W, wa.H);
  file_frames(wa, tab, tid);
  __syncthreads();
  if (stats && wa.count && tid == 0 && blockIdx.x == 0) atomicAdd(&stats[2], (unsigned long long)n_f);
  const uint32_t nbx = ((uint32_t)v.nx + kBrickX - 1) / kBrickX, nby = ((uint32_t)v.ny + kBrickY - 1) / kBrickY;
  const uint32_t nbz = ((uint32_t)v.nz + kBrickZ - 1) / kBrickZ;
  // bricks in flight form a compact box: z fastest, then 8 x 8 tiles of brick columns (32 x 32 voxels) over the padded brick grid
  const uint32_t tiles_y = (nby + 7u) / 8u;
  const uint32_t q = blockIdx.x * 4u + (uint32_t)wave;
  const uint32_t bz = q % nbz, t = q / nbz, tt = t / 64u, r = t % 64u;
  const uint32_t bx = (tt / tiles_y) * 8u + r / 8u, by = (tt % tiles_y) * 8u + r % 8u;
  const bool on = bx < nbx && by < nby;  // (no early return: the workgroup meets again in cls_accumulate)
  const int ix = (int)bx * kBrickX + (lane & 3), iy = (int)by * kBrickY + ((lane >> 2) & 3);
  // a lane's four voxels: z = iz0 + kZS j (SAF_CLS_CUBE: one per 4 x 4 x 4 cube of the brick; else four consecutive ones)
  constexpr int kZS = SAF_CLS_CUBE ? 4 : 1;
  const int iz0 = (int)bz * kBrickZ + (SAF_CLS_CUBE ? (lane >> 4) : (lane >> 4) * 4);
  const bool col_in = on && ix < v.nx && iy < v.ny;
  const int ixc = min(ix, v.nx - 1), iyc = min(iy, v.ny - 1);
  const uint32_t nb = ((uint32_t)ixc * (uint32_t)v.ny + (uint32_t)iyc) * (uint32_t)v.nz + (uint32_t)min(iz0, v.nz - 1);
  float xw[4], yw[4], zw[4];
  bool inb[4];
  const float x_l = v.ax[ixc], y_l = v.ay[iyc];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    xw[j] = x_l;
    yw[j] = y_l;
    zw[j] = v.az[min(iz0 + kZS * j, v.nz - 1)];
    inb[j] = col_in && iz0 + kZS * j < v.nz;
  }
  // ---- which frames can touch this brick at all?  lane k tests frame k
  uint32_t live;
  uint32_t cull[kCullWords];  // (brick, frame) pairs of this wave: tested, and dropped by reason (stats[8..12])
  {
    // (the brick's voxel centres that lie inside the grid; a brick in the padding is clamped onto the grid: it is not classified)
    const int bx0 = min((int)bx * kBrickX, v.nx - 1), by0 = min((int)by * kBrickY, v.ny - 1), bz0 = min((int)bz * kBrickZ, v.nz - 1);
    const float x0 = v.ax[bx0], x1 = v.ax[min(bx0 + kBrickX - 1, v.nx - 1)];
    const float y0 = v.ay[by0], y1 = v.ay[min(by0 + kBrickY - 1, v.ny - 1)];
    const float z0 = v.az[bz0], z1 = v.az[min(bz0 + kBrickZ - 1, v.nz - 1)];
    const float cxw = 0.5f * (x0 + x1), cyw = 0.5f * (y0 + y1), czw = 0.5f * (z0 + z1);
    const float hx = 0.5f * (x1 - x0), hy = 0.5f * (y1 - y0), hz = 0.5f * (z1 - z0);
    const float rho = sqrtf(hx * hx + hy * hy + hz * hz) * 1.02f + 1e-4f;  // voxel CENTRES are what is tested
    bool dead = false;
    int why = 0;  // what dropped the frame: 1 behind the camera, 2 beyond the frame's largest depth + trunc, 3 outside the frustum, 4 occluded
    if (lane < n_f) {
      const Cam c = s_cam[lane];
      const bool pinhole = c.k01 == 0.0f && c.k10 == 0.0f && c.k20 == 0.0f && c.k21 == 0.0f && c.k22 == 1.0f;
      const float dx = cxw - c.tx, dy = cyw - c.ty, dz = czw - c.tz;
      const float xc = c.r00 * dx + c.r10 * dy + c.r20 * dz;  // R^T (X - t)
      const float yc = c.r01 * dx + c.r11 * dy + c.r21 * dz;
      const float zc = c.r02 * dx + c.r12 * dy + c.r22 * dz;
      const float far_z = dmax[lane] + v.trunc * 1.01f + 1e-4f;  // beyond it: sdf < -1 for every pixel (inf: never)
      // the sphere argument needs a rigid pose (R orthonormal: distances survive R^T) and z = the camera-space
      // z (third row of K = 0 0 1); anything else is classified without a cull
      const float n0 = c.r00 * c.r00 + c.r10 * c.r10 + c.r20 * c.r20, n1 = c.r01 * c.r01 + c.r11 * c.r11 + c.r21 * c.r21;
      const float n2 = c.r02 * c.r02 + c.r12 * c.r12 + c.r22 * c.r22;
      const float d01 = c.r00 * c.r01 + c.r10 * c.r11 + c.r20 * c.r21, d02 = c.r00 * c.r02 + c.r10 * c.r12 + c.r20 * c.r22;
      const float d12 = c.r01 * c.r02 + c.r11 * c.r12 + c.r21 * c.r22;
      const bool rigid = fabsf(n0 - 1.0f) < 1e-3f && fabsf(n1 - 1.0f) < 1e-3f && fabsf(n2 - 1.0f) < 1e-3f &&
                         fabsf(d01) < 1e-3f && fabsf(d02) < 1e-3f && fabsf(d12) < 1e-3f;
      const bool z_is_depth = c.k20 == 0.0f && c.k21 == 0.0f && c.k22 == 1.0f;
      // (SAF_CLS_BOX, round 4: the brick is 4 x 4 x 16 voxels -- its bounding sphere is four times as wide as the brick is
      //  across -- so the tests use the BOX's extent along each direction instead: for a direction n in camera space the
      //  box reaches |n . ex| hx + |n . ey| hy + |n . ez| hz from its centre, ex / ey / ez = the world axes in camera space)
      const float mg = 1.02f, me = 1e-4f;
      const float ext_x = SAF_CLS_BOX ? (fabsf(c.r00) * hx + fabsf(c.r10) * hy + fabsf(c.r20) * hz) * mg + me : rho;
      const float ext_y = SAF_CLS_BOX ? (fabsf(c.r01) * hx + fabsf(c.r11) * hy + fabsf(c.r21) * hz) * mg + me : rho;
      const float ext_z = SAF_CLS_BOX ? (fabsf(c.r02) * hx + fabsf(c.r12) * hy + fabsf(c.r22) * hz) * mg + me : rho;
      dead = rigid && z_is_depth && (zc + ext_z <= 0.0f || zc - ext_z > far_z);
      why = dead ? (zc + ext_z <= 0.0f ? 1 : 2) : 0;
      if (pinhole && rigid) {
        // in view <=> -0.5 <= u/z <= W - 0.5 and -0.5 <= v/z <= H - 0.5 with u = k00 x + k02 z, v = k11 y + k12 z:
        // four planes through the camera centre; the brick is outside if its centre is farther than its reach behind one
        const float a1 = c.k02 + 0.5f, a2 = c.k02 - c.fw + 0.5f, b1 = c.k12 + 0.5f, b2 = c.k12 - c.fh + 0.5f;
        auto reach = [&](float nx, float ny, float nz) {  // of the box along (nx, ny, nz) (camera space, not normalised)
          return (fabsf(nx * c.r00 + ny * c.r01 + nz * c.r02) * hx + fabsf(nx * c.r10 + ny * c.r11 + nz * c.r12) * hy +
                  fabsf(nx * c.r20 + ny * c.r21 + nz * c.r22) * hz) * mg;
        };
        const float n1 = sqrtf(c.k00 * c.k00 + a1 * a1), n2 = sqrtf(c.k00 * c.k00 + a2 * a2);
        const float m1 = sqrtf(c.k11 * c.k11 + b1 * b1), m2 = sqrtf(c.k11 * c.k11 + b2 * b2);
        const float d1 = (c.k00 * xc + a1 * zc) / n1, d2 = (c.k00 * xc + a2 * zc) / n2;
        const float e1 = (c.k11 * yc + b1 * zc) / m1, e2 = (c.k11 * yc + b2 * zc) / m2;
        const float rd1 = SAF_CLS_BOX ? reach(c.k00, 0.0f, a1) / n1 + me : rho, rd2 = SAF_CLS_BOX ? reach(c.k00, 0.0f, a2) / n2 + me : rho;
        const float re1 = SAF_CLS_BOX ? reach(0.0f, c.k11, b1) / m1 + me : rho, re2 = SAF_CLS_BOX ? reach(0.0f, c.k11, b2) / m2 + me : rho;
        const bool fx_pos = c.k00 > 0.0f, fy_pos = c.k11 > 0.0f;  // the usual orientation; otherwise no frustum cull
        dead = dead || (fx_pos && (d1 < -rd1 || d2 > rd2)) || (fy_pos && (e1 < -re1 || e2 > re2));
        why = dead && !why ? 3 : why;
        // ---- occlusion: the largest depth over the pixels the brick can project onto (SAF_CLS_OCCL=0 at build time: off).
        // A voxel's pixel is round(u), u = k00 x / z + k02 (clipfusion.py:651-661 undone: grid_sample's un-normalisation gives
        // back the pixel coordinate); over the sphere's bounding box x in [xc - rho, xc + rho], z in [zc - rho, zc + rho] (z > 0)
        // u is monotone in x and in 1 / z: a conservative pixel rectangle, widened by a pixel, at most 16 tiles of it.
        const float zn = zc - ext_z, zf = zc + ext_z;
        if (SAF_CLS_OCCL && !dead && z_is_depth && fx_pos && fy_pos && zn > 1e-3f && zn > dmax[kWin + lane] + v.trunc * 1.01f + 1e-4f) {
          const float rn = 1.0f / zn, rf = 1.0f / zf;
          const float xl = xc - ext_x, xh = xc + ext_x, yl = yc - ext_y, yh = yc + ext_y;
          const float ul = c.k00 * (xl >= 0.0f ? xl * rf : xl * rn) + c.k02, uh = c.k00 * (xh >= 0.0f ? xh * rn : xh * rf) + c.k02;
          const float vl = c.k11 * (yl >= 0.0f ? yl * rf : yl * rn) + c.k12, vh = c.k11 * (yh >= 0.0f ? yh * rn : yh * rf) + c.k12;
          // (1e-3 relative for the reciprocals' rounding, a pixel for round-half-even and the reference's own arithmetic)
          const float su = 1e-3f * (fabsf(ul) + fabsf(uh)) + 1.0f, sv = 1e-3f * (fabsf(vl) + fabsf(vh)) + 1.0f;
          const int px0 = max((int)floorf(fmaxf(ul - su, -1.0f)), 0), px1 = min((int)ceilf(fminf(uh + su, c.fw)), wa.W - 1);
          const int py0 = max((int)floorf(fmaxf(vl - sv, -1.0f)), 0), py1 = min((int)ceilf(fminf(vh + sv, c.fh)), wa.H - 1);
          if (px0 <= px1 && py0 <= py1) {
            const int tx0 = px0 >> ts_log2, tx1 = px1 >> ts_log2, ty0 = py0 >> ts_log2, ty1 = py1 >> ts_log2;
            if (tx1 - tx0 < 4 && ty1 - ty0 < 4) {
              // the 4 x 4 tiles from (tx0, ty0), clamped onto the rectangle (a tile may be read twice): 16 loads in flight
              const float* __restrict__ tm = tmax + (size_t)lane * kMaxDepthTiles;
              float t[16];
#pragma unroll
              for (int k = 0; k < 16; ++k) t[k] = tm[min(ty0 + (k >> 2), ty1) * tiles_x + min(tx0 + (k & 3), tx1)];
              float m = 0.0f;
#pragma unroll
              for (int k = 0; k < 16; ++k) m = t[k] > m ? t[k] : m;
              dead = zn > m + v.trunc * 1.01f + 1e-4f;  // (inf: never)
              why = dead ? 4 : 0;
            }
          }
        }
      }
    }
    live = on ? (uint32_t)__ballot(lane < n_f && !dead) : 0u;
    cull[0] = on ? (uint32_t)n_f : 0u;
#pragma unroll
    for (int k = 1; k < kCullWords; ++k) cull[k] = on ? (uint32_t)__popcll(__ballot(why == k)) : 0u;
  }
  const float rtrunc = 1.0f / v.trunc;
  const bool tsdf_aligned = (((uintptr_t)v.tsdf | (uintptr_t)v.tsdf_w) & 15) == 0;
  unsigned long long nt_done = 0, tsdf_rows_done = 0;
  uint32_t mk4[4] = {0u, 0u, 0u, 0u};
  if (live)
    classify_voxels<SAF_CLS_FU, SUM, VERIFY, TILED>(v, wa, s_cam, nb, xw, yw, zw, inb, rtrunc, tsdf_aligned, live, mk4, nt_done, tsdf_rows_done);
  // every voxel of the grid gets its mask word (the row kernel reads them all); 16 bytes at once where the four lie in the grid
  // and the run is aligned (always, when nz is a multiple of 4)
  if (kZS == 1 && inb[3] && (nb & 3u) == 0u) {
    *reinterpret_cast<uint4*>(hitmask + nb) = make_uint4(mk4[0], mk4[1], mk4[2], mk4[3]);
  } else {  // (cube order: the four lanes of a column write 16 consecutive bytes per instruction)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (inb[j]) hitmask[nb + kZS * j] = mk4[j];
  }
  cls_accumulate(nt_done, tsdf_rows_done, cull, lane, wave, s_acc, s_cull, stats ? cls_acc : nullptr);
}

// One hit of a sub-chunk, held by the lane with the hit's index.
struct WinHit {
  int row;  // row slot in the LDS buffer
  float a, b, nw, ne, sw, se;
};
template <int CPL>
struct WinCtx {
  __amdgpu_buffer_rsrc_t maps;  // the window's map images as a buffer: a load beyond its end returns zeros and moves nothing
  int img_vecs, DV, npx, npy, zero_row, lane;
  float4* rows;
};
typedef unsigned int win_v4u __attribute__((vector_size(16)));
__device__ __forceinline__ float4 win_tap_load(__amdgpu_buffer_rsrc_t maps, uint32_t byte_off) {
  const win_v4u v = __builtin_amdgcn_raw_buffer_load_b128(maps, (int)byte_off, 0, 0);
  return __builtin_bit_cast(float4, v);
}
// Channel chunk c of a lane (a float4 of the D-channel map row).  f32 volume: lane + 64 c.  bf16 volume:
// a lane's 16-byte row unit holds 8 channels = chunks 2 (lane + 64 (c / 2)) + c % 2.
template <bool BF16>
__device__ __forceinline__ constexpr int win_chunk_off(int c) {
  return BF16 ? (c >> 1) * 128 + (c & 1) : c * 64;
}
// bf16 volume: the rows of a sub-chunk travel through registers (16 bytes = 8 channels per lane and unit)
// and are widened into the f32 LDS rows once they have landed.
template <int SR, int UPL>
struct WinRaw {
  uint4 u[SR * UPL];
  int nrows;
};

// Per lane: the byte offsets of the four map rows of the lane's hit (frame image + tap position, or "outside");
// read at a group's first lane.
struct WinGroupOffs {
  int nw, ne, sw, se;
};
template <int CPL>
__device__ __forceinline__ WinGroupOffs win_group_offsets(const WinCtx<CPL>& cx, uint32_t k) {
  const int fb = (int)(k >> 16), x0 = (int)(k & 255u) - 2, y0 = (int)((k >> 8) & 255u) - 2;
  const bool x0ok = x0 >= 0 && x0 < cx.npx, x1ok = x0 + 1 >= 0 && x0 + 1 < cx.npx;
  const bool y0ok = y0 >= 0 && y0 < cx.npy, y1ok = y0 + 1 >= 0 && y0 + 1 < cx.npy;
  const int base = fb * cx.img_vecs;  // < 2^27: kWin images of (P + 1) * D / 4 float4 each
  // BYTE offsets into the buffer of map images.  A tap outside the map (zeros padding; 17 % of the taps of a 5 x 7 map)
  // gets an offset beyond the buffer: the hardware's range check returns zeros without a request to L1 / L2 -- and the
  // L1's miss queue is what bounds this kernel.
  WinGroupOffs o;
  o.nw = (x0ok && y0ok) ? (base + (y0 * cx.npx + x0) * cx.DV) * 16 : (int)kTapOutside;
  o.ne = (x1ok && y0ok) ? (base + (y0 * cx.npx + x0 + 1) * cx.DV) * 16 : (int)kTapOutside;
  o.sw = (x0ok && y1ok) ? (base + ((y0 + 1) * cx.npx + x0) * cx.DV) * 16 : (int)kTapOutside;
  o.se = (x1ok && y1ok) ? (base + ((y0 + 1) * cx.npx + x0 + 1) * cx.DV) * 16 : (int)kTapOutside;
  return o;
}

// NB groups (runs of lanes hl[u] .. hl[u + 1] - 1 of hits in group order): request the four map rows of every group, then
// blend group after group into the LDS rows (the waits are counted: group u is processed while the rows of groups
// u+1.. are still in flight).
template <int NB, int CPL, bool SUM, bool BF16, int SR>
__device__ __forceinline__ void win_batch(const WinCtx<CPL>& cx, const int (&hl)[NB + 1], bool first, const WinGroupOffs& go,
                                          const WinHit& rec, const WinRaw<SR, BF16 ? CPL / 2 : 1>& raw) {
  float4 tp[NB][4][CPL];
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    // the group's four map rows as float4 offsets into the window's images (computed lane-parallel by
    // win_group_offsets: the row kernel issues as many scalar as vector instructions, this keeps the
    // per-group address arithmetic off the scalar unit)
    const int o_nw = __builtin_amdgcn_readlane(go.nw, hl[u]), o_ne = __builtin_amdgcn_readlane(go.ne, hl[u]);
    const int o_sw = __builtin_amdgcn_readlane(go.sw, hl[u]), o_se = __builtin_amdgcn_readlane(go.se, hl[u]);
    const uint32_t lane_off = (uint32_t)(BF16 ? 2 * cx.lane : cx.lane) * 16u;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t off = lane_off + (uint32_t)win_chunk_off<BF16>(c) * 16u;
      tp[u][0][c] = win_tap_load(cx.maps, (uint32_t)o_nw + off);
      tp[u][1][c] = win_tap_load(cx.maps, (uint32_t)o_ne + off);
      tp[u][2][c] = win_tap_load(cx.maps, (uint32_t)o_sw + off);
      tp[u][3][c] = win_tap_load(cx.maps, (uint32_t)o_se + off);
    }
  }
  if (first) {  // the sub-chunk's first batch: its rows (issued before these loads) have landed after this
    // the BUILTIN, not inline asm: the compiler's wait-count pass must see that the LDS-DMA has been
    // waited for, or it drains vmcnt before every later LDS read (each row's store waited for the last)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt and lgkmcnt untouched
    if (BF16) {
      constexpr int UPL = CPL / 2;
#pragma unroll
      for (int r = 0; r < SR; ++r) {
        if (r < raw.nrows) {
#pragma unroll
          for (int k = 0; k < UPL; ++k) {
            const uint4 w = raw.u[r * UPL + k];
            cx.rows[(r * CPL + 2 * k) * 64 + cx.lane] = make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
            cx.rows[(r * CPL + 2 * k + 1) * 64 + cx.lane] = make_float4(bf16_lo(w.z), bf16_hi(w.z), bf16_lo(w.w), bf16_hi(w.w));
          }
        }
      }
    }
    wave_lds_sync();
  }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    for (int l = hl[u]; l < hl[u + 1]; ++l) {  // the group's hits: lanes hl[u] .. hl[u + 1] - 1
      const int r = __builtin_amdgcn_readlane(rec.row, l);
      const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.a), l));
      const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.b), l));
      Bilin w;
      w.x0 = 0; w.y0 = 0;
      w.nw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.nw), l));
      w.ne = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.ne), l));
      w.sw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.sw), l));
      w.se = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.se), l));
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        float4* rp = cx.rows + (r * CPL + c) * 64 + cx.lane;
        const float4 sv = lerp_taps(tp[u][0][c], tp[u][1][c], tp[u][2][c], tp[u][3][c], w);
        float4 nv = blend(sv, *rp, a, b, SUM);
        if (BF16) {  // the per-frame path stores bf16 after every hit: round to nearest even, keep as f32
          const uint32_t p01 = pack_bf16(nv.x, nv.y), p23 = pack_bf16(nv.z, nv.w);  // two hardware conversions
          nv.x = bf16_lo(p01); nv.y = bf16_hi(p01); nv.z = bf16_lo(p23); nv.w = bf16_hi(p23);
        }
        *rp = nv;
      }
    }
  }
}

// ---- order-free form (OF): a row's samples of the window are summed in registers ----
// A row hit k times in a window goes from `old` (weight w0) to (w0 old + the sum of its k samples) / (w0 + k): the running
// mean of clipfusion.py:715-721 with the window's k updates folded into one (order-free up to fp32 rounding, SURVEY
// section 7).  The accumulator of row r of the sub-chunk is acc[r]: loaded with the old row (global -> registers, no LDS),
// scaled by w0 when it has landed, every hit adds nw t0 + ne t1 + sw t2 + se t3 with packed FMAs whose weights come from
// SGPRs (v_readlane), one multiply by 1 / (w0 + k) on the way out.  Per hit: no LDS access, no blend.
typedef float win_v2f __attribute__((ext_vector_type(2)));
// (the accumulators are PAIRS of channels from end to end -- what v_pk_fma_f32 reads and writes: carried as float4 through
//  the loops below, every loop entry and exit repacked them with v_mov copies into a second register set)
template <int CPL>
__device__ __forceinline__ void of_add_row(win_v2f (&a)[2 * CPL], const float4 (&tp)[4][CPL], float nw, float ne, float sw,
                                           float se) {
  const win_v2f w[4] = {{nw, nw}, {ne, ne}, {sw, sw}, {se, se}};
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a[2 * c] = __builtin_elementwise_fma((win_v2f){tp[t][c].x, tp[t][c].y}, w[t], a[2 * c]);
      a[2 * c + 1] = __builtin_elementwise_fma((win_v2f){tp[t][c].z, tp[t][c].w}, w[t], a[2 * c + 1]);
    }
  }
}
// One batch of NB groups.  The hit's row must be a compile-time register set: a chain of compares on the (wave-uniform) row
// number merges the sets behind phi copies (or, sunk into one body, behind a dynamically indexed array in scratch memory).
// So the loops are turned inside out: for every group u and every row R of the sub-chunk (both unrolled), the lanes
// rm[R] & [hl[u], hl[u + 1]) are that row's hits of that group -- usually none or one -- and each body adds into acc[R]
// from tp[u] in place.  Within a row the hits stay in (frame, lane) order: the sum is reproducible.
template <int NB, int CPL, bool SUM, bool BF16, int SR>
__device__ __forceinline__ void win_batch_of(const WinCtx<CPL>& cx, const int (&hl)[NB + 1], int nb, const WinGroupOffs& go,
                                             const WinHit& rec, const unsigned long long (&rm)[SR],
                                             win_v2f (&acc)[SR][2 * CPL]) {
  if constexpr (BF16 && SAF_WIN_MAPS16_BUILD) {  // bf16 map images: one 16-byte load per tap and unit of 8 channels, widened in the FMA operands
    constexpr int UPL = CPL / 2;
    uint4 tq[NB][4][UPL];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const bool real = u < nb;
      const int hu = real ? hl[u] : 0;
      const int o4[4] = {real ? __builtin_amdgcn_readlane(go.nw, hu) : (int)(kTapOutside + 0x10000u * (4 * u)),
                         real ? __builtin_amdgcn_readlane(go.ne, hu) : (int)(kTapOutside + 0x10000u * (4 * u + 1)),
                         real ? __builtin_amdgcn_readlane(go.sw, hu) : (int)(kTapOutside + 0x10000u * (4 * u + 2)),
                         real ? __builtin_amdgcn_readlane(go.se, hu) : (int)(kTapOutside + 0x10000u * (4 * u + 3))};
#pragma unroll
      for (int k = 0; k < UPL; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          tq[u][t][k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(cx.maps, (int)((uint32_t)o4[t] + (uint32_t)cx.lane * 16u + (uint32_t)k * 1024u), 0, 0));
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const unsigned long long below_hi = hl[u + 1] >= 64 ? ~0ull : ((1ull << hl[u + 1]) - 1ull);
      const unsigned long long range = below_hi & ~((1ull << hl[u]) - 1ull);
#pragma unroll
      for (int R = 0; R < SR; ++R) {
        unsigned long long m = rm[R] & range;
        while (m) {
          const int l = __ffsll((long long)m) - 1;
          m &= m - 1ull;
          const float wt[4] = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.nw), l)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.ne), l)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.sw), l)),
                               __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.se), l))};
#pragma unroll
          for (int k = 0; k < UPL; ++k) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const uint4 w = tq[u][t][k];
              const win_v2f ww = {wt[t], wt[t]};
              acc[R][4 * k + 0] = __builtin_elementwise_fma((win_v2f){bf16_lo(w.x), bf16_hi(w.x)}, ww, acc[R][4 * k + 0]);
              acc[R][4 * k + 1] = __builtin_elementwise_fma((win_v2f){bf16_lo(w.y), bf16_hi(w.y)}, ww, acc[R][4 * k + 1]);
              acc[R][4 * k + 2] = __builtin_elementwise_fma((win_v2f){bf16_lo(w.z), bf16_hi(w.z)}, ww, acc[R][4 * k + 2]);
              acc[R][4 * k + 3] = __builtin_elementwise_fma((win_v2f){bf16_lo(w.w), bf16_hi(w.w)}, ww, acc[R][4 * k + 3]);
            }
          }
        }
      }
    }
    return;
  }
  float4 tp[NB][4][CPL];
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const bool real = u < nb && !(SAF_WIN_ABL & 1);  // (hl[u] = 64 for an empty group when the pass is full: no such lane to read)
    const int hu = real ? hl[u] : 0;
    // (distinct offsets for an empty group's taps: identical loads would be merged into one)
    const int o_nw = real ? __builtin_amdgcn_readlane(go.nw, hu) : (int)(kTapOutside + 0x10000u * (4 * u));
    const int o_ne = real ? __builtin_amdgcn_readlane(go.ne, hu) : (int)(kTapOutside + 0x10000u * (4 * u + 1));
    const int o_sw = real ? __builtin_amdgcn_readlane(go.sw, hu) : (int)(kTapOutside + 0x10000u * (4 * u + 2));
    const int o_se = real ? __builtin_amdgcn_readlane(go.se, hu) : (int)(kTapOutside + 0x10000u * (4 * u + 3));
    const uint32_t lane_off = (uint32_t)(BF16 ? 2 * cx.lane : cx.lane) * 16u;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const uint32_t off = lane_off + (uint32_t)win_chunk_off<BF16>(c) * 16u;
      if (SAF_WIN_ABL & 2) {
        const float f = __builtin_bit_cast(float, (uint32_t)o_nw + off);
        tp[u][0][c] = tp[u][1][c] = tp[u][2][c] = tp[u][3][c] = make_float4(f, f, f, f);
        continue;
      }
      tp[u][0][c] = win_tap_load(cx.maps, (uint32_t)o_nw + off);
      tp[u][1][c] = win_tap_load(cx.maps, (uint32_t)o_ne + off);
      tp[u][2][c] = win_tap_load(cx.maps, (uint32_t)o_sw + off);
      tp[u][3][c] = win_tap_load(cx.maps, (uint32_t)o_se + off);
    }
  }
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    // lanes hl[u] .. hl[u + 1] - 1 (hl[u + 1] <= 64)
    const unsigned long long below_hi = hl[u + 1] >= 64 ? ~0ull : ((1ull << hl[u + 1]) - 1ull);
    const unsigned long long range = below_hi & ~((1ull << hl[u]) - 1ull);
#pragma unroll
    for (int R = 0; R < SR; ++R) {
      unsigned long long m = rm[R] & range;
      while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        const float nw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.nw), l));
        const float ne = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.ne), l));
        const float sw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.sw), l));
        const float se = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rec.se), l));
        if (SAF_WIN_ABL & 16) {
#pragma unroll
          for (int c = 0; c < CPL; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(tp[u][t][c].x), "v"(tp[u][t][c].y), "v"(tp[u][t][c].z), "v"(tp[u][t][c].w), "s"(nw), "s"(ne), "s"(sw), "s"(se));
        } else {
          of_add_row<CPL>(acc[R], tp[u], nw, ne, sw, se);
        }
      }
    }
  }
}



template <int CPL, bool SUM, bool BF16, bool OF>
__device__ __forceinline__ void
fuse_window_body(const KVol& v, const WinArgs& wa, const WinTable* __restrict__ tab, const float* __restrict__ map_imgs, int img_vecs,
                 unsigned long long* __restrict__ stats, unsigned int* __restrict__ piece_ctr,
                 const uint32_t* __restrict__ hitmask, uint32_t mask_plane, const unsigned long long* __restrict__ cls_acc,
                 int xcd_order) {
  using Cfg = WinCfg<CPL, OF, BF16>;
  constexpr int SR = Cfg::SR;
  // (s_setprio 1 / 3 here, ahead of the classification waves that share the SIMDs, changes nothing: 105.4 / 105.7 / 105.8 ms)
  // a bf16 sub-chunk also holds its raw rows in registers until they are widened: one tap group fewer in flight
#ifndef SAF_WIN_OF_P16
#define SAF_WIN_OF_P16 0  // tap groups in flight with bf16 map images at D = 512 (0: as for f32 images; their registers are half as many)
#endif
  constexpr int P = OF ? ((BF16 && SAF_WIN_MAPS16_BUILD && CPL == 2 && SAF_WIN_OF_P16 > 0) ? SAF_WIN_OF_P16 : Cfg::P)
                       : (BF16 && Cfg::P > 2 ? Cfg::P - 1 : (BF16 && CPL == 4 ? 1 : Cfg::P));  // (bf16, D = 1024: two groups of 64 tap registers in flight spilled)
  extern __shared__ __align__(16) unsigned char s_dyn[];
#ifdef SAF_WIN_PRIO
  __builtin_amdgcn_s_setprio(SAF_WIN_PRIO);
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4* rows = reinterpret_cast<float4*>(s_dyn + Cfg::rows_off) + (size_t)wave * SR * CPL * 64;
  uint32_t* stage = reinterpret_cast<uint32_t*>(s_dyn + Cfg::stage_off) + (size_t)wave * 6 * kHitCap;
  uint32_t* s_hf = stage;  // voxel lane (6 bits) | cell << 6 (16 bits) | frame of the window << 22 (7 bits)
  int* s_hw = reinterpret_cast<int*>(stage + kHitCap);
  float* s_ha = reinterpret_cast<float*>(stage + 2 * kHitCap);
  float* s_hb = reinterpret_cast<float*>(stage + 3 * kHitCap);
  float* s_hgx = reinterpret_cast<float*>(stage + 4 * kHitCap);
  float* s_hgy = reinterpret_cast<float*>(stage + 5 * kHitCap);
  uint32_t* s_tm = reinterpret_cast<uint32_t*>(s_dyn + Cfg::tm_off) + wave * kUnitVox * kMaskWords;
  uint16_t* s_tv = reinterpret_cast<uint16_t*>(s_dyn + Cfg::tv_off) + wave * kUnitVox;
  const float** s_rgb = reinterpret_cast<const float**>(s_dyn + Cfg::ptr_off);
  const float** s_lab = s_rgb + kWin;
  Cam* s_cam = reinterpret_cast<Cam*>(s_dyn + Cfg::cam_off);

  if (stats && blockIdx.x == 0 && tid < kClsShards) {  // the classification launches' sharded counters (see cls_accumulate)
    unsigned long long a = cls_acc[2 * tid], b = cls_acc[2 * tid + 1];
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
    }
    if (tid == 0) {
      if (a) atomicAdd(&stats[1], a);
      if (b) atomicAdd(&stats[6], b);
    }
    fold_cull_shards(cls_acc, stats, tid);
  }
  static_assert(kWinThreads >= kWin, "one thread per frame loads the window's cameras");
  if (tid < wa.F) {
    s_cam[tid] = load_cam(tab->pose[tid], tab->K[tid], wa.W, wa.H);
    s_rgb[tid] = tab->rgb[tid];
    s_lab[tid] = tab->label_map[tid];
  }
  __syncthreads();
  const int DV = v.D >> 2;
  const int DVM = (BF16 && OF && SAF_WIN_MAPS16_BUILD) ? v.D >> 3 : v.D >> 2;  // 16-byte vectors of a MAP row (bf16 map images in the order-free form of a bf16 volume)
  const float half_px = (float)wa.npx / 2.0f, half_py = (float)wa.npy / 2.0f;
  const int zero_row = wa.npx * wa.npy;
  int chs[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) chs[c] = lane + c * 64;
  constexpr int UPL = BF16 ? CPL / 2 : 1;  // 16-byte units of a bf16 row per lane
  float4* feat = reinterpret_cast<float4*>(v.feat);
  float4* featb = reinterpret_cast<float4*>(v.feat);  // bf16 volume: D / 8 units of 16 bytes per row
  const __amdgpu_buffer_rsrc_t maps_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(map_imgs), 0, (int)((size_t)wa.F * img_vecs * sizeof(float4)), 0x00020000);
  KFrame kf;  // per-hit view of a frame for the scalar side
  kf.H = wa.H; kf.W = wa.W; kf.npy = wa.npy; kf.npx = wa.npx; kf.rgb_bilinear = wa.rgb_bilinear;
  kf.depth = nullptr; kf.pose = nullptr; kf.K = nullptr;
  unsigned long long hits_done = 0, rows_done = 0;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  WT_DECL;
  // persistent grid (2 workgroups of 4 waves per CU, 71 KB of LDS each).  Pieces are handed out by an
  // atomic counter: a coherent scene concentrates its hits in few pieces (a wall = whole columns).
  const uint32_t n_pieces = (v.N + kPiece - 1) / kPiece;
  // The unit of work is a QUARTER of a piece (the voxels of 16 lanes): a column inside a wall carries
  // thousands of hits, and whoever draws it last decides when the kernel ends.
  // Order of the units.  Linear (xcd_order == 0): one counter, units in index order -- the units in flight are a few whole
  // x-planes.  XCD-compact (grids of 16 x 16-column tiles with nz a multiple of 64): workgroup i runs on XCD i % 8 (round-
  // robin dispatch), every XCD draws from its own counter and walks its own tiles (tile t belongs to XCD t % 8) z-section by
  // z-section, so the 256 units an XCD has in flight are one 16 x 16 x 64-voxel box: the map taps its L2 must hold are the
  // few map positions that box sees in each frame.  An XCD that runs out of units helps the next one.
  constexpr uint32_t kSplit = 1u << SAF_WIN_SPLIT_LOG2;
  const uint32_t tiles_y = (uint32_t)v.ny / 16u, n_tiles = ((uint32_t)v.nx / 16u) * tiles_y;
  const uint32_t zsecs = (uint32_t)v.nz / (kPiece / kSplit);  // units of a column
  const uint32_t upt = 256u * zsecs;                            // units of a tile
  uint32_t xcd = blockIdx.x & 7u, xcd_tries = 0;
  for (;;) {
    uint32_t unit = 0;
    if (xcd_order) {
      const uint32_t my_tiles = (n_tiles + 7u - xcd) / 8u;
      uint32_t j = 0;
      if (lane == 0) j = atomicAdd(piece_ctr + xcd, 1u);
      j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
      if (j >= my_tiles * upt) {
        if (++xcd_tries == 8u) break;
        xcd = (xcd + 1u) & 7u;
        continue;
      }
      const uint32_t tl = j / upt, within = j - tl * upt, zs = within >> 8, col = within & 255u;
      const uint32_t t = tl * 8u + xcd, tx = t / tiles_y, ty = t - tx * tiles_y;
      const uint32_t X = tx * 16u + (col >> 4), Y = ty * 16u + (col & 15u);
      unit = (X * (uint32_t)v.ny + Y) * zsecs + zs;  // nz % 64 == 0: a column is zsecs whole units
    } else {
      if (lane == 0) unit = atomicAdd(piece_ctr, 1u);
      unit = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit);
    }
    const uint32_t piece = unit >> SAF_WIN_SPLIT_LOG2;
    if (piece >= n_pieces) break;
    const bool mine = (uint32_t)(lane >> (6 - SAF_WIN_SPLIT_LOG2)) == (unit & ((1u << SAF_WIN_SPLIT_LOG2) - 1u));
    const uint32_t piece_base = piece * (uint32_t)kPiece;
    // ---- the piece's touched voxels: (local id, frame mask) left by classify_bricks_kernel, compacted into LDS
    uint32_t mk4[4][kMaskWords];
    {
      const uint32_t nb = piece_base + (uint32_t)lane * 4u;
      // mask word w of every voxel lives in plane w (written by the classification launch of frames 32 w ..)
#pragma unroll
      for (int w = 0; w < kMaskWords; ++w) {
        const uint32_t* mrow = hitmask + (size_t)w * mask_plane + nb;
        if (w * 32 >= wa.F || !mine) {
          mk4[0][w] = mk4[1][w] = mk4[2][w] = mk4[3][w] = 0u;  // a short window has no such plane; other units' voxels
        } else if (nb + 3u < v.N) {
          const uint4 t = *reinterpret_cast<const uint4*>(mrow);
          mk4[0][w] = t.x; mk4[1][w] = t.y; mk4[2][w] = t.z; mk4[3][w] = t.w;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) mk4[k][w] = nb + k < v.N ? mrow[k] : 0u;
        }
      }
    }
    int T = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t any = 0u;
#pragma unroll
      for (int w = 0; w < kMaskWords; ++w) any |= mk4[k][w];
      const bool touched = mine && any != 0u;
      const unsigned long long bal = __ballot(touched);
      if (touched) {
        const int slot = T + __popcll(bal & lt_mask);
        s_tv[slot] = (uint16_t)(lane * 4 + k);
#pragma unroll
        for (int w = 0; w < kMaskWords; ++w) s_tm[slot * kMaskWords + w] = mk4[k][w];
      }
      T += __popcll(bal);
    }
    wave_lds_sync();
    WT(0);
    rows_done += (unsigned long long)T;
    int pos = 0;
    while (pos < T) {
      const int cnt = min(64, T - pos);
      const uint32_t vl = lane < cnt ? s_tv[pos + lane] : 0u;
      uint32_t mk[kMaskWords];
      int h = 0;
#pragma unroll
      for (int w = 0; w < kMaskWords; ++w) {
        mk[w] = lane < cnt ? s_tm[(pos + lane) * kMaskWords + w] : 0u;
        h += __popc(mk[w]);
      }
      // inclusive prefix sum of h over the wave
      int incl = h;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const int prefix = incl - h;
      const unsigned long long fits = __ballot(lane < cnt && incl <= kHitCap);
      const int m = fits == ~0ull ? 64 : (__ffsll((long long)~fits) - 1);  // leading voxels whose hits fit (>= 1)
      const bool active = lane < m;
      const uint32_t n_l = piece_base + vl;
      const int w0 = active ? v.weight[n_l] : 0;
      const int htot = __builtin_amdgcn_readlane(incl, m - 1);
      // ---- expand the masks into the hit list (frame order within a voxel)
      if (active) {
        int r = 0;
#pragma unroll
        for (int w = 0; w < kMaskWords; ++w) {
          uint32_t mm = mk[w];
          while (mm) {
            const int fbit = __ffs((int)mm) - 1 + 32 * w;
            mm &= mm - 1u;
            s_hf[prefix + r] = (uint32_t)lane | ((uint32_t)fbit << 22);
            ++r;
          }
        }
      }
      wave_lds_sync();
      WT(1);
      // ---- lane-parallel over hits: projection, the hit's map cell, the frame's rgb sample and the label
      //      count (clipfusion.py:647-659, :701-706; clip_seem_fusion.py:786-822).  The samples come from up
      //      to 64 different images: fetched here, one round of 64 hits at a time, they cost two exposed
      //      latencies per chunk instead of one per hit of the chunk's busiest voxel.  They are parked in
      //      the a / b / w slots of the staging area until the blend below replaces them.
      float* s_hs2 = reinterpret_cast<float*>(s_hw);
      for (int j0 = 0; j0 < htot; j0 += 64) {
        const int j = j0 + lane;
        const uint32_t hf = j < htot ? s_hf[j] : 0u;
        const uint32_t n = (uint32_t)__shfl((int)n_l, (int)(hf & 63u));
        int lbl = -1;  // the hit's panoptic class, -1: none
        if (j < htot) {
          const int fb = (int)(hf >> 22);
          int ix, iy, iz;
          voxel_coords(v, n, ix, iy, iz);
          const Cam cam = s_cam[fb];
          const Proj p = project(cam, v.ax[ix], v.ay[iy], v.az[iz]);
          s_hgx[j] = p.gx;
          s_hgy[j] = p.gy;
          // the hit's map cell: (y0, x0) of its four taps; every cell wholly outside the map is one cell
          const Bilin bw = bilinear_setup(p.gx, p.gy, half_px, half_py);
          const int cx = min(max(bw.x0, -2), wa.npx) + 2, cy = min(max(bw.y0, -2), wa.npy) + 2;
          s_hf[j] = hf | ((uint32_t)((cy << 8) | cx) << 6);
          kf.rgb = s_rgb[fb];
          kf.label_map = s_lab[fb];
          float s0, s1, s2, lpk = 0.0f;
          // (wa.rgbl: ClipSeemFusion's bilinear rgb and the pixel's class from the window's packed images -- four 16-byte gathers)
          const int pix = wa.rgbl ? sample_rgbl_lane(wa.rgbl + (size_t)fb * wa.rgbl_px, wa.rgbl_tiles_x, kf, cam, p.gx, p.gy, s0, s1, s2, lpk)
                                  : sample_rgb_lane(kf, cam, p.gx, p.gy, s0, s1, s2);
          s_ha[j] = s0;
          s_hb[j] = s1;
          s_hs2[j] = s2;
#if !SAF_WIN_LABEL_RUNS
          count_label_lane<true>(v, kf, n, pix, stats);
#endif
          // the hit's class (clip_seem_fusion.py:786-791: nearest sample of the panoptic map, .long())
          if (SAF_WIN_LABEL_RUNS && v.labels && kf.label_map) {
            const float lraw = wa.rgbl ? lpk : kf.label_map[pix >= 0 ? pix : 0];
            const long long l = (long long)(pix >= 0 ? lraw : 0.f);
            const bool ok = l >= 0 && l < v.n_classes;
            lbl = ok ? (int)l : -1;
            if (!ok && stats) atomicAdd(&stats[3], 1ull);
          }
        }
        if (SAF_WIN_LABEL_RUNS && v.labels && s_lab[0]) {
          // Label histogram (clip_seem_fusion.py:820-822), one add per RUN: a voxel's hits lie in consecutive lanes in frame
          // order, and what a panoptic model says about a static scene is the same frame after frame -- the first lane of a
          // run of equal classes of one voxel adds the run's length (integer adds: bit for bit the per-hit histogram; a run
          // that straddles two rounds of 64 hits is two adds).  Maps that are iid per pixel have runs of one: as before.
          const int prev_l = __shfl_up(lbl, 1);
          const uint32_t prev_v = (uint32_t)__shfl_up((int)(hf & 63u), 1);
          const bool head = lbl >= 0 && (lane == 0 || prev_l != lbl || prev_v != (hf & 63u));
          const unsigned long long stops = __ballot(head || lbl < 0);  // where a run cannot continue
          const unsigned long long above = lane == 63 ? 0ull : (stops >> (lane + 1));
          const int len = above ? __ffsll((long long)above) : 64 - lane;
          if (head) atomicAdd(v.labels + (int64_t)n * v.n_classes + lbl, len);
        }
      }
      wave_lds_sync();
      WT(2);
      // ---- lane-parallel over voxels: the rgb running mean over the voxel's hits in frame order, in
      //      registers (clipfusion.py:715-721); a = 1/(w+1), b = w*a of every hit take the samples' slots
      if (active) {
        float* dst = v.rgb + (int64_t)n_l * 3;
        float o0 = dst[0], o1 = dst[1], o2 = dst[2];
        for (int r = 0; r < h; ++r) {
          const int j = prefix + r;
          const int wi = w0 + r;
          const float a = 1.0f / (float)(wi + 1), b = (float)wi * a;
          o0 = blend(s_ha[j], o0, a, b, SUM);
          o1 = blend(s_hb[j], o1, a, b, SUM);
          o2 = blend(s_hs2[j], o2, a, b, SUM);
          s_ha[j] = a;
          s_hb[j] = b;
        }
        dst[0] = o0; dst[1] = o1; dst[2] = o2;
        v.weight[n_l] = w0 + h;
      }
      wave_lds_sync();
      WT(3);
      // ---- rows: sub-chunks of <= SR rows and <= 64 hits; a single row with more than 64 hits (windows longer than 64
      //      frames) is a sub-chunk of its own whose hits are applied in passes of 64 over the LDS-resident row
      int i0 = 0;
      while (i0 < m) {
        const int pbase = __builtin_amdgcn_readlane(prefix, i0);
        // up to SR rows and kSubHits hits (a voxel has at most kWin <= kSubHits): more than 64 hits are applied in passes
        constexpr int kSubHits = 128;
        static_assert(kSubHits >= kWin && kSubHits <= kHitCap && kSubHits <= 128, "two blocks of 64 hits at most");
        // (round 5, profiles/r05/row_knobs.txt: sub-chunks cut at 64 hits -- one pass, never a presort -- are SLOWER on the coherent
        //  scene, 7.9 -> 9.0 ms per window: what a sub-chunk costs is mostly per sub-chunk, not per pass)
        const unsigned long long okm = __ballot(lane >= i0 && lane < m && lane < i0 + SR && (incl - pbase) <= kSubHits);
        const int nrows = __popcll(okm);  // >= 1
        const int nh = __builtin_amdgcn_readlane(incl, i0 + nrows - 1) - pbase;  // 1..kSubHits
        WinRaw<SR, UPL> raw;
        raw.nrows = nrows;
        const WinCtx<CPL> cx{maps_rsrc, img_vecs, DVM, wa.npx, wa.npy, zero_row, lane, rows};
        // order-free form: acc[r] = the old row (never scaled: the hits' weights are, by 1 / w0), then the window's samples
        win_v2f acc[OF ? SR : 1][2 * CPL];
        // Several rows with more than 64 hits between them (coherent scenes: ~15 hits per row): the staging entries of the
        // sub-chunk are brought into (frame, row) order first -- ranks from the rows' frame masks as below, for both blocks
        // of 64 hits, every staging field permuted through the free s_hw array -- and the passes then find them sorted.
        // (round 5, profiles/r05/row_knobs.txt: a FRAME-MAJOR walk instead -- the frames taken in ascending order off the union of
        //  the rows' masks, per frame the rows and their record lanes found from the masks lane-parallel, nothing sorted: 27 instead
        //  of 54 vector instructions per hit, bit-identical sums, and 43 % SLOWER on the coherent scene, 9 % on random depth: every
        //  group costs a serial mask -> ballot -> scalar -> readlane -> load round trip, which the sort pays once per pass)
        const bool presorted = nrows > 1 && nh > 64;
        if (presorted) {
          int rk[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int j = pbase + 64 * q + lane;
            const bool hq = 64 * q + lane < nh;
            const uint32_t hf = hq ? s_hf[j] : 0u;
            const int my_row = (int)(hf & 63u) - i0;
            const uint32_t f = hf >> 22, fwd = f >> 5, fbit = f & 31u;
            int r_ = 0;
#pragma unroll
            for (int r = 0; r < SR; ++r) {
              if (r < nrows) {
                uint32_t at_f = 0u;
#pragma unroll
                for (int w = 0; w < kMaskWords; ++w) {
                  const uint32_t mw = (uint32_t)__builtin_amdgcn_readlane((int)mk[w], i0 + r);
                  const uint32_t low = (uint32_t)w < fwd ? 0xffffffffu : ((uint32_t)w == fwd ? (1u << fbit) - 1u : 0u);
                  r_ += __popc(mw & low);
                  at_f = fwd == (uint32_t)w ? mw : at_f;
                }
                r_ += (r < my_row && ((at_f >> fbit) & 1u)) ? 1 : 0;
              }
            }
            rk[q] = hq ? r_ : 64 * q + lane;  // entries past the end keep their place
          }
          uint32_t* fields[5] = {s_hf, reinterpret_cast<uint32_t*>(s_ha), reinterpret_cast<uint32_t*>(s_hb),
                                 reinterpret_cast<uint32_t*>(s_hgx), reinterpret_cast<uint32_t*>(s_hgy)};
          uint32_t* tmp = reinterpret_cast<uint32_t*>(s_hw);
#pragma unroll
          for (int fi = 0; fi < 5; ++fi) {
            const uint32_t v0 = fields[fi][pbase + lane];
            const uint32_t v1 = 64 + lane < nh ? fields[fi][pbase + 64 + lane] : 0u;
            wave_lds_sync();
            tmp[rk[0]] = v0;
            if (64 + lane < nh) tmp[rk[1]] = v1;
            wave_lds_sync();
            fields[fi][pbase + lane] = tmp[lane];
            if (64 + lane < nh) fields[fi][pbase + 64 + lane] = tmp[64 + lane];
            wave_lds_sync();
          }
        }
        for (int h0 = 0; h0 < nh; h0 += 64) {
          // hit h0 + l of the sub-chunk (staging entry pbase + h0 + l) lives in lane l: its row, a, b and tap weights
          const bool hit = h0 + lane < nh;
          const int sj = pbase + h0 + lane;
          const uint32_t hfl = hit ? s_hf[sj] : 0xffffffffu;
          const uint32_t key = hfl >> 6;  // frame << 16 | cell
          WinHit rec;
          rec.row = (int)(hfl & 63u) - i0;
          rec.a = hit ? s_ha[sj] : 0.0f;
          rec.b = hit ? s_hb[sj] : 0.0f;
          {
            const Bilin w = bilinear_setup(hit ? s_hgx[sj] : 0.0f, hit ? s_hgy[sj] : 0.0f, half_px, half_py);
            rec.nw = w.nw; rec.ne = w.ne; rec.sw = w.sw; rec.se = w.se;
          }
          if constexpr (OF && !SUM && !BF16) {
            // f32 running mean: the accumulator starts as the OLD row, unscaled (it is loaded straight into the registers and
            // first touched by whichever hit comes first), so every sample of a row is weighted by 1 / w0 instead -- lane-
            // parallel, on the hit's four tap weights -- and the row leaves as acc x w0 / (w0 + k).  A fresh row (w0 = 0)
            // starts at zero: weight 1, leaves as acc / k.
            const int w0h = __shfl(w0, (int)(hfl & 63u));
            const float rs = w0h == 0 ? 1.0f : 1.0f / (float)w0h;
            rec.nw *= rs; rec.ne *= rs; rec.sw *= rs; rec.se *= rs;
          }
          int rank = 0;
          if (h0 == 0) {
            // (the staging reads above come BEFORE the LDS-DMA below: the compiler drains vmcnt ahead of any LDS
            //  read that follows an LDS-DMA, which would expose the rows' whole latency right here)
            // the rows, global -> LDS (one LDS-DMA moves a wave's 64 x 16 B = one 1 KiB piece of a row)
            // A voxel of weight 0 has never been written: its row is all zeros by construction (rows are only
            // written together with a weight increment, clipfusion.py:715-721), so it is not read -- in a fresh
            // volume that is every row's first window.  (For the running mean the old row would be multiplied by
            // b = 0 anyway.)  Zero rows are written to LDS first: an LDS store after an LDS-DMA makes the compiler
            // drain vmcnt.
            uint32_t fresh = 0;  // bit r: row r of the sub-chunk is untouched so far
            const uint32_t fwd = (key >> 16) >> 5, fbit = (key >> 16) & 31u;  // the hit's frame: mask word and bit
            uint32_t low[kMaskWords];                                        // the frames before it
#pragma unroll
            for (int w = 0; w < kMaskWords; ++w)
              low[w] = (uint32_t)w < fwd ? 0xffffffffu : ((uint32_t)w == fwd ? (1u << fbit) - 1u : 0u);
#pragma unroll
            for (int r = 0; r < SR; ++r) {
              if (r < nrows && (__builtin_amdgcn_readlane(w0, i0 + r) == 0 || (OF && (SAF_WIN_ABL & 4)))) {
                fresh |= 1u << r;
                if (!BF16 && !OF) {
#pragma unroll
                  for (int c = 0; c < CPL; ++c) rows[(r * CPL + c) * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
              }
            }
            // (the ranks are only used to sort a sub-chunk of several rows that was not presorted: on a coherent scene -- 85 hits per
            //  sub-chunk, nearly always presorted -- this loop was ~200 wasted vector instructions per sub-chunk)
            if (nrows > 1 && !presorted)
#pragma unroll
            for (int r = 0; r < SR; ++r) {
              if (r < nrows) {
                // the hit's place in (frame, row) order, from the rows' frame masks: hits of earlier frames in this row,
                // plus this frame's hit of it if the row comes first
                uint32_t m[kMaskWords], at_f = 0u;
#pragma unroll
                for (int w = 0; w < kMaskWords; ++w) {
                  m[w] = (uint32_t)__builtin_amdgcn_readlane((int)mk[w], i0 + r);
                  rank += __popc(m[w] & low[w]);
                  at_f = fwd == (uint32_t)w ? m[w] : at_f;
                }
                rank += (r < rec.row && ((at_f >> fbit) & 1u)) ? 1 : 0;
              }
            }
            if constexpr (OF) {
              // The registers the old rows are loaded into are the ones the previous sub-chunk STORED from, and a store's data
              // registers may not be rewritten before the store has left: the compiler guards every such write with a
              // vmcnt wait -- placed between the rows' loads it serialises them (load, wait for it AND the stores, load, ...:
              // five memory latencies per sub-chunk, 15 % of the kernel).  ONE wait here, after the rank arithmetic above has
              // given the stores time to leave, and the loads below are issued back to back.
              __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
              // (zeroes first, loads after: a register write behind a load in flight waits for the load)
#pragma unroll
              for (int r = 0; r < SR; ++r) {
                // bf16: the accumulator holds the samples only, the old row waits in `raw`; f32: a fresh or absent row starts at 0
                if (BF16 || r >= nrows || (fresh & (1u << r))) {
#pragma unroll
                  for (int c = 0; c < 2 * CPL; ++c) acc[r][c] = (win_v2f){0.f, 0.f};
                }
                if (BF16 && (r >= nrows || (fresh & (1u << r)))) {
#pragma unroll
                  for (int k = 0; k < UPL; ++k) raw.u[r * UPL + k] = make_uint4(0u, 0u, 0u, 0u);
                }
              }
            }
#pragma unroll
            for (int r = 0; r < SR; ++r) {
              if (r < nrows) {
                const int64_t row = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)n_l, i0 + r) * DV;
                if (fresh & (1u << r)) {
                  if (BF16 && !OF) {
#pragma unroll
                    for (int k = 0; k < UPL; ++k) raw.u[r * UPL + k] = make_uint4(0u, 0u, 0u, 0u);
                  }
                } else if (BF16) {
#pragma unroll
                  for (int k = 0; k < UPL; ++k) {
                    const float4 t = ld_stream(featb + (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)n_l, i0 + r) * (DV / 2) +
                                               lane + k * 64);
                    raw.u[r * UPL + k] = make_uint4(__builtin_bit_cast(uint32_t, t.x), __builtin_bit_cast(uint32_t, t.y),
                                                    __builtin_bit_cast(uint32_t, t.z), __builtin_bit_cast(uint32_t, t.w));
                  }
                } else if constexpr (OF) {  // the old row, global -> the accumulator registers (never through LDS)
                  // through a buffer descriptor of the ROW (four SGPRs): the per-lane offset is one register, the same for every
                  // row and never rewritten -- with 64-bit addresses in VGPRs every row's loads and stores would wait for the
                  // previous row's to leave before their address registers may be recomputed
                  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(feat + row, 0, v.D * 4, 0x00020000);
#pragma unroll
                  for (int c = 0; c < CPL; ++c) {
                    const float4 t = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, lane * 16 + c * 1024, 0, SAF_WIN_OF_LDPOL));
                    acc[r][2 * c] = (win_v2f){t.x, t.y};
                    acc[r][2 * c + 1] = (win_v2f){t.z, t.w};
                  }
                } else {
#pragma unroll
                  for (int c = 0; c < CPL; ++c)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(feat + row + chs[c]),
                                                     (__attribute__((address_space(3))) void*)(rows + (r * CPL + c) * 64),
                                                     16, 0, 2);
                }
              }
            }
          }
          WT(4);
          // Groups = hits of one frame in one map cell (they blend the same four map rows, loaded once per group), frames
          // ascending, so that a row's hits stay in frame order.  The hits are brought into (frame, row) order by a
          // permutation computed lane-parallel from the rows' frame masks (`rank`; a single row's hits are in frame order
          // already) -- a group is then a run of lanes with one key, found with one neighbour compare and a ballot.  (The
          // earlier form walked the frames with ballots, one serial scalar round trip per group: 11 % of the kernel.)
          WinHit srt = rec;
          uint32_t skey = key;
          if (nrows > 1 && !presorted) {
            const int to = (hit ? rank : lane) * 4;  // lanes without a hit keep their place
            srt.row = __builtin_amdgcn_ds_permute(to, rec.row);
            srt.a = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, rec.a)));
            srt.b = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, rec.b)));
            srt.nw = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, rec.nw)));
            srt.ne = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, rec.ne)));
            srt.sw = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, rec.sw)));
            srt.se = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(to, __builtin_bit_cast(int, rec.se)));
            skey = (uint32_t)__builtin_amdgcn_ds_permute(to, (int)key);
          }
          const uint32_t prev_key = (uint32_t)__shfl_up((int)skey, 1);
          unsigned long long heads = __ballot(hit && (lane == 0 || skey != prev_key));  // `hit` lanes are lanes 0 .. n - 1
          const int n_pass = nh - h0 < 64 ? nh - h0 : 64;
          WT(5);
          const WinGroupOffs go = win_group_offsets(cx, skey);
          unsigned long long rm[OF ? SR : 1];  // order-free form: the lanes (hits in group order) of every row of the sub-chunk
          if constexpr (OF) {
#pragma unroll
            for (int r = 0; r < SR; ++r) rm[r] = __ballot(lane < n_pass && srt.row == r);
          }
          bool first = h0 == 0;
          while (heads) {
            int hl[P + 1];
            int nb = 0;
#pragma unroll
            for (int u = 0; u < P; ++u) {
              if (heads) {
                hl[u] = __ffsll((long long)heads) - 1;
                heads &= heads - 1ull;
                nb = u + 1;
              } else {
                hl[u] = n_pass;
              }
            }
            hl[P] = heads ? __ffsll((long long)heads) - 1 : n_pass;
            if constexpr (OF) {
              // ONE instantiation: a batch with fewer than P groups has empty groups at its end (hl[u] = hl[u + 1] = n_pass), whose
              // taps are "outside the map" -- no memory request -- and whose hit loops find no lane
              win_batch_of<P, CPL, SUM, BF16, SR>(cx, hl, nb, go, srt, rm, acc);
            } else {
            switch (nb) {
                case 1: { const int h1[2] = {hl[0], hl[1]}; win_batch<1, CPL, SUM, BF16, SR>(cx, h1, first, go, srt, raw); break; }
                case 2: if constexpr (P >= 2) { const int h2[3] = {hl[0], hl[1], hl[2]}; win_batch<2, CPL, SUM, BF16, SR>(cx, h2, first, go, srt, raw); } break;
                case 3: if constexpr (P >= 3) { const int h3[4] = {hl[0], hl[1], hl[2], hl[3]}; win_batch<3, CPL, SUM, BF16, SR>(cx, h3, first, go, srt, raw); } break;
                case 4: if constexpr (P >= 4) { const int h4[5] = {hl[0], hl[1], hl[2], hl[3], hl[4]}; win_batch<4, CPL, SUM, BF16, SR>(cx, h4, first, go, srt, raw); } break;
                case 5: if constexpr (P >= 5) { const int h5[6] = {hl[0], hl[1], hl[2], hl[3], hl[4], hl[5]}; win_batch<5, CPL, SUM, BF16, SR>(cx, h5, first, go, srt, raw); } break;
                default: if constexpr (P >= 6) { const int h6[7] = {hl[0], hl[1], hl[2], hl[3], hl[4], hl[5], hl[6]}; win_batch<6, CPL, SUM, BF16, SR>(cx, h6, first, go, srt, raw); } break;
              }
            }
            first = false;
          }
          WT(6);  // (inside the pass loop: a second pass's records must not be charged with the first pass's tap batches)
        }
        // nothing is outstanding here (every tap load has been consumed); the explicit wait only tells the
        // compiler's wait-count pass so, or it would drain vmcnt -- i.e. the previous row's store -- before
        // each row's LDS read below
        __builtin_amdgcn_s_waitcnt(0x0F70);  // (order-free form: see below -- nothing is outstanding either)
        if constexpr (!OF) wave_lds_sync();
        WT(6);
#pragma unroll
        for (int r = 0; r < SR; ++r) {
          if (r < nrows) {
            const int64_t row = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)n_l, i0 + r) * DV;
            if constexpr (OF) {  // (w0 old + the window's samples) / (w0 + k), one rounding to bf16 per window
              // IN PLACE, and stored from the row's own registers through the row's buffer descriptor: through shared data or
              // address temporaries every row's stores would wait for the previous row's to leave (a store's registers may not
              // be rewritten before that): five store latencies per sub-chunk, 17 % of the kernel.  The explicit vmcnt(0)
              // above (free: every tap load has been consumed) tells the compiler's wait-count pass that no older store -- the
              // chunk's rgb / weight stores -- still reads the scratch registers of the arithmetic below.
              const int k_r = __builtin_amdgcn_readlane(h, i0 + r), w0_r = __builtin_amdgcn_readlane(w0, i0 + r);
              const float inv = 1.0f / (float)(w0_r + k_r);
              // f32: acc = old + (sum of samples) / w0 (or the plain sum for a fresh row); bf16: acc = the sum of samples
              const float fac = SUM ? 1.0f : ((BF16 || w0_r == 0) ? inv : (float)w0_r * inv);
              if (BF16) {
                const float f = SUM ? 1.0f : (float)w0_r;
#pragma unroll
                for (int k = 0; k < UPL; ++k) {
                  const uint4 w = raw.u[r * UPL + k];  // + w0 x old (SUM: + old), then the mean, packed into the same registers
                  const win_v2f a0 = acc[r][4 * k], a1 = acc[r][4 * k + 1], a2 = acc[r][4 * k + 2], a3 = acc[r][4 * k + 3];
                  uint4 q;
                  q.x = pack_bf16(__builtin_fmaf(bf16_lo(w.x), f, a0.x) * fac, __builtin_fmaf(bf16_hi(w.x), f, a0.y) * fac);
                  q.y = pack_bf16(__builtin_fmaf(bf16_lo(w.y), f, a1.x) * fac, __builtin_fmaf(bf16_hi(w.y), f, a1.y) * fac);
                  q.z = pack_bf16(__builtin_fmaf(bf16_lo(w.z), f, a2.x) * fac, __builtin_fmaf(bf16_hi(w.z), f, a2.y) * fac);
                  q.w = pack_bf16(__builtin_fmaf(bf16_lo(w.w), f, a3.x) * fac, __builtin_fmaf(bf16_hi(w.w), f, a3.y) * fac);
                  raw.u[r * UPL + k] = q;
                  if (!(SAF_WIN_ABL & 8))
                    st_stream(featb + (row / 2) + lane + k * 64, make_float4(__builtin_bit_cast(float, q.x), __builtin_bit_cast(float, q.y),
                                                                             __builtin_bit_cast(float, q.z), __builtin_bit_cast(float, q.w)));
                }
              } else {
                const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(feat + row, 0, v.D * 4, 0x00020000);
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                  if (!SUM) {
                    acc[r][2 * c] = acc[r][2 * c] * (win_v2f){fac, fac};
                    acc[r][2 * c + 1] = acc[r][2 * c + 1] * (win_v2f){fac, fac};
                  }
                  const float4 o = make_float4(acc[r][2 * c].x, acc[r][2 * c].y, acc[r][2 * c + 1].x, acc[r][2 * c + 1].y);
                  if (SAF_WIN_ABL & 8)
                    asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
                  else
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(win_v4u, o), rr, lane * 16 + c * 1024, 0, SAF_WIN_OF_STPOL);
                }
              }
            } else if (BF16) {
#pragma unroll
              for (int k = 0; k < UPL; ++k) {
                const float4 lo = rows[(r * CPL + 2 * k) * 64 + lane], hi = rows[(r * CPL + 2 * k + 1) * 64 + lane];
                float4 o;  // the LDS values are bf16-exact already: the packing is lossless
                o.x = __builtin_bit_cast(float, pack_bf16(lo.x, lo.y));
                o.y = __builtin_bit_cast(float, pack_bf16(lo.z, lo.w));
                o.z = __builtin_bit_cast(float, pack_bf16(hi.x, hi.y));
                o.w = __builtin_bit_cast(float, pack_bf16(hi.z, hi.w));
                st_stream(featb + (row / 2) + lane + k * 64, o);
              }
            } else {
#pragma unroll
              for (int c = 0; c < CPL; ++c) st_stream(&feat[row + chs[c]], rows[(r * CPL + c) * 64 + lane]);
            }
          }
        }
        if constexpr (!OF) wave_lds_sync();  // the row buffer is rewritten by the next sub-chunk
        WT(7);
        i0 += nrows;
      }
      hits_done += (unsigned long long)htot;
      pos += m;
      wave_lds_sync();  // the staging area is rewritten by the next chunk
    }
  }
  WT_FLUSH;
  if (stats && lane == 0) {
    if (hits_done) atomicAdd(&stats[0], hits_done);
    if (rows_done) atomicAdd(&stats[5], rows_done);  // rows read-modify-written by this window
  }
}

// The kernels proper.  Registers are handed out in blocks of eight and a SIMD has 512: two row waves of up to 176 leave room for TWO
// classification waves of 80 beside them, two of 177 (= 184) for one.  The order-free D = 512 kernels -- the benchmark's -- are
// therefore capped at 176 (amdgpu_num_vgpr takes no template-dependent value: hence two wrappers around one body); the bf16 one
// had crept to 177 in round 5, and config 3's classification beside it lost its second wave per SIMD.
#ifndef SAF_WIN_OF_VGPRS
#define SAF_WIN_OF_VGPRS 176
#endif
template <int CPL, bool SUM, bool BF16, bool OF>
__global__ __launch_bounds__(kWinThreads)
__attribute__((amdgpu_waves_per_eu(OF ? SAF_WIN_OF_WPE : SAF_WIN_WPE, OF ? SAF_WIN_OF_WPE : SAF_WIN_WPE))) void
fuse_window_kernel(KVol v, WinArgs wa, const WinTable* __restrict__ tab, const float* __restrict__ map_imgs, int img_vecs,
                   unsigned long long* __restrict__ stats, unsigned int* __restrict__ piece_ctr,
                   const uint32_t* __restrict__ hitmask, uint32_t mask_plane, const unsigned long long* __restrict__ cls_acc,
                   int xcd_order) {
  fuse_window_body<CPL, SUM, BF16, OF>(v, wa, tab, map_imgs, img_vecs, stats, piece_ctr, hitmask, mask_plane, cls_acc, xcd_order);
}
template <int CPL, bool SUM, bool BF16>
__global__ __launch_bounds__(kWinThreads) __attribute__((amdgpu_num_vgpr(SAF_WIN_OF_VGPRS))) void
fuse_window_kernel_of176(KVol v, WinArgs wa, const WinTable* __restrict__ tab, const float* __restrict__ map_imgs, int img_vecs,
                         unsigned long long* __restrict__ stats, unsigned int* __restrict__ piece_ctr,
                         const uint32_t* __restrict__ hitmask, uint32_t mask_plane, const unsigned long long* __restrict__ cls_acc,
                         int xcd_order) {
  fuse_window_body<CPL, SUM, BF16, true>(v, wa, tab, map_imgs, img_vecs, stats, piece_ctr, hitmask, mask_plane, cls_acc, xcd_order);
}

// ---------------------------------------------------------------------------------------------
// Windowed (voxel-major) path of saf_fuse_frames: see fuse_window_kernel.
// Workspace: the common header (piece counter), the kWin pixel-major map images of one window, and the
// window's frame bitmasks (kMaskWords words per voxel).
// ---------------------------------------------------------------------------------------------
struct WinLayout {
  size_t img_bytes, maps_bytes, mask_bytes, tile_off, tile_win, rgbl_off, cmax_off, total;
  uint32_t mask_plane;
};
// rgbl_px_pad > 0 (with depth_px_pad: the layout of saf_fuse_workspace_bytes_for_frames for a volume that counts labels): room for
// ONE window's packed {r, g, b, label} images behind the tile region
WinLayout win_layout(int64_t n_vox, int D, int P, bool bricks = false, size_t depth_px_pad = 0, size_t rgbl_px_pad = 0) {
  WinLayout w;
  w.img_bytes = ((size_t)D * (P + 1) * sizeof(float) + 255) & ~(size_t)255;
  w.maps_bytes = (size_t)kWin * w.img_bytes;
  w.mask_plane = (uint32_t)((n_vox + 63) & ~(int64_t)63);  // words per mask plane (16-byte aligned planes)
  w.mask_bytes = ((size_t)w.mask_plane * sizeof(uint32_t) * kMaskWords + 255) & ~(size_t)255;
  w.tile_off = kHdrTotal + w.maps_bytes + 2 * w.mask_bytes;  // the classification's depth tile maxima (one launch's)
  w.tile_win = (tile_win_bytes(depth_px_pad) + 255) & ~(size_t)255;
  w.rgbl_off = w.tile_off + kTileWindows * w.tile_win;
  w.cmax_off = w.rgbl_off + (((size_t)kWin * rgbl_px_pad * sizeof(float4) + 255) & ~(size_t)255);  // the brick form's channel maxima and camera table
  // the brick form's segment pools (6.5 GB at 256^3) only where that form can run: the row forms end at cmax_off
  w.total = w.cmax_off + (bricks ? brick_aux_bytes_est(n_vox, D) : 0);
  return w;
}

using WinFn = void (*)(KVol, WinArgs, const WinTable*, const float*, int, unsigned long long*, unsigned int*, const uint32_t*,
                       uint32_t, const unsigned long long*, int);
template <int CPL, bool OF>
WinFn pick_win(bool sum, bool bf16) {
  if (OF && CPL == 2 && SAF_WIN_OF_VGPRS > 0) {  // the benchmark's kernels: within 176 registers (see fuse_window_kernel_of176)
    if (bf16) return sum ? fuse_window_kernel_of176<2, true, true> : fuse_window_kernel_of176<2, false, true>;
    return sum ? fuse_window_kernel_of176<2, true, false> : fuse_window_kernel_of176<2, false, false>;
  }
  if (bf16) {
    if (CPL % 2 != 0) return nullptr;
    constexpr int C2 = CPL % 2 == 0 ? CPL : 2;
    return sum ? fuse_window_kernel<C2, true, true, OF> : fuse_window_kernel<C2, false, true, OF>;
  }
  return sum ? fuse_window_kernel<CPL, true, false, OF> : fuse_window_kernel<CPL, false, false, OF>;
}
template <int CPL>
WinFn pick_win(bool sum, bool bf16, bool of, size_t* lds) {
  *lds = of ? WinCfg<CPL, true>::total : WinCfg<CPL, false>::total;
  return of ? pick_win<CPL, true>(sum, bf16) : pick_win<CPL, false>(sum, bf16);
}

// Shapes the windowed path takes; everything else runs the per-frame pipeline.
}  // namespace

size_t window_workspace_bytes(int64_t n_vox, int D, int P, bool bricks, int H, int W, bool labels) {
  const bool fr = H > 0 && W > 0;
  return win_layout(n_vox, D, P, bricks, fr ? depth_px_padded(H, W) : 0, fr && labels ? rgbl_px_padded(H, W) : 0).total;
}

// SAF_WIN_FORM (read per call): "rows" = the frame-ordered row kernel (bit-identical to fusing frame after frame), "sums" =
// its order-free form (a row's samples of the window summed in registers, one blend per row: feature values within fp32
// rounding of the sequential path, everything else exact), "bricks" = saf_brick.hip.  Default: sums.
bool window_form_sums() {
  const char* e = getenv("SAF_WIN_FORM");
  return !e || e[0] == 's';
}

// Frames per window: 128 (SAF_WINDOW_FRAMES) unless SAF_WIN_FRAMES=64 asks for the shorter form (read per call).
int window_frames() {
  const char* e = getenv("SAF_WIN_FRAMES");
  return e && atoi(e) == 64 ? 64 : kWin;
}

bool window_ok(const KVol& kv, const saf_frame* frames, int32_t n_frames, size_t workspace_bytes) {
  const bool enabled = !(getenv("SAF_WINDOW") && getenv("SAF_WINDOW")[0] == '0');
  if (!enabled || n_frames < kWinMinFrames) return false;
  // SAF_WINDOW_BF16=0 keeps bf16 volumes on the per-frame pipeline
  const bool bf16_on = !(getenv("SAF_WINDOW_BF16") && getenv("SAF_WINDOW_BF16")[0] == '0');
  if (kv.bf16 && !bf16_on) return false;
  const saf_frame& fr0 = frames[0];
  const WinLayout wl0 = win_layout(kv.N, kv.D, fr0.npy * fr0.npx);
  const bool bricks = brick_form_ok(kv) && workspace_bytes > wl0.cmax_off && brick_aux_fits(kv, workspace_bytes - wl0.cmax_off);
  if (!bricks) {  // the frame-ordered row kernel: whole 1 KiB pieces of a row per wave instruction
    if (kv.D % 256 != 0 || kv.D > 1024) return false;
    if (kv.bf16 && kv.D % 512 != 0) return false;  // a lane moves 8 bf16 channels: 512 per wave
  }
  const saf_frame& f0 = frames[0];
  for (int32_t i = 0; i < n_frames; ++i) {
    const saf_frame& f = frames[i];
    if (f.height != f0.height || f.width != f0.width || f.npy != f0.npy || f.npx != f0.npx ||
        f.rgb_bilinear != f0.rgb_bilinear || (f.label_map == nullptr) != (f0.label_map == nullptr))
      return false;
  }
  if (f0.npx + 3 > 255 || f0.npy + 3 > 255) return false;  // a hit's map cell travels as two bytes
  const WinLayout wl = win_layout(kv.N, kv.D, f0.npy * f0.npx);
  if (wl.maps_bytes >= (size_t)kTapOutside) return false;  // the taps are buffer loads with 31-bit byte offsets
  return workspace_bytes >= wl.cmax_off;  // (the brick form's own region was checked above: brick_aux_fits)
}

// x-planes [x0, x0 + nx) of a volume as a volume of their own: the same buffers, offset (a slab of x-planes is a contiguous
// range of the flat voxel index; the axis table starts at x0, so every decision is that of the full volume's voxels).
KVol slab_kvol(const KVol& kv, int x0, int nx) {
  KVol o = kv;
  const int64_t rows = (int64_t)x0 * kv.ny * kv.nz;
  o.nx = nx;
  o.N = (uint32_t)((int64_t)nx * kv.ny * kv.nz);
  o.ax = kv.ax + x0;
  o.tsdf = kv.tsdf + rows;
  o.tsdf_w = kv.tsdf_w + rows;
  o.weight = kv.weight + rows;
  o.rgb = kv.rgb + 3 * rows;
  o.feat = kv.feat + rows * (kv.bf16 ? kv.D / 2 : kv.D);
  if (kv.labels) o.labels = kv.labels + rows * kv.n_classes;
  return o;
}

// The schedule of a windowed call is a list of UNITS, each a (sub-volume, window of frames) pair with its own
// classification launches and its own row kernel; unit u + 1 is classified (auxiliary stream) beside unit u's row
// kernel.  Units of one call:
//   * default: one unit per window over the whole volume (SAF_WIN_SLABS / SAF_WIN_W0_SLABS: windows cut into slabs of
//     x-planes -- nothing hides a unit's classification except the row kernel of the unit before it);
//   * `slabs` (saf_fuse_frames_slabs): every frame into slab 0, then every frame into slab 1, ...: a finished slab is
//     never touched again (its event is recorded behind its last row kernel: the merge of the slab-pipelined
//     multi-GPU job starts there), and the first classification of slab s + 1 runs beside the last row kernel of slab s.
struct WinUnit {
  KVol kv;
  int f0, F;     // frames [f0, f0 + F) of the call
  int count;     // counts its frames in stats[2]
  int done;      // index of the slab whose event is recorded behind this unit's row kernel, or -1
  int window;    // index of the window (units of one window in a row share its map images)
};

int fuse_many_windowed(const KVol& kv, const saf_frame* frames, int32_t n_frames, void* workspace, size_t workspace_bytes,
                       uint64_t* stats, saf_profiler* prof, hipStream_t s, const WinOverlap* ov, const WinSlabs* slabs, bool recycled) {
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  int rc = SAF_OK;
  KFrame kf0;
  if ((rc = make_kframe(&frames[0], &kf0))) return rc;
  for (int32_t i = 1; i < n_frames; ++i) {
    KFrame t;
    if ((rc = make_kframe(&frames[i], &t))) return rc;
  }
  const int P = kf0.npy * kf0.npx;
  // Two layouts of the workspace: with room for the window's depth images re-laid-out in tiles (a workspace sized by
  // saf_fuse_workspace_bytes_for_frames) or without (the classification then reads the frames' own row-major images).
  // SAF_CLS_TILED=0 (read per call): never tiled; 2: the first unit of a call reads the tiled copies too (tests: a single-window call).
  const size_t dpx = depth_px_padded(kf0.H, kf0.W);
  // (a volume that counts labels: the frames-sized layout also holds one window of packed {r, g, b, label} images)
  const size_t rpx = kv.labels ? rgbl_px_padded(kf0.H, kf0.W) : 0;
  const WinLayout wl_lin = win_layout(kv.N, kv.D, P), wl_til = win_layout(kv.N, kv.D, P, false, dpx, rpx);
  const char* til_env = getenv("SAF_CLS_TILED");
  bool tiled = !(til_env && til_env[0] == '0') && dpx * sizeof(float) < (size_t)1 << 31 && workspace_bytes >= wl_til.cmax_off;
  if (tiled && brick_form_ok(kv)) {  // the brick form's pools follow the tile region: both must fit, or neither moves
    const size_t a_lin = workspace_bytes > wl_lin.cmax_off ? workspace_bytes - wl_lin.cmax_off : 0;
    const size_t a_til = workspace_bytes - wl_til.cmax_off;
    if (a_lin > 0 && brick_aux_fits(kv, a_lin) && !(a_til > 0 && brick_aux_fits(kv, a_til))) tiled = false;
  }
  const WinLayout wl = tiled ? wl_til : wl_lin;
  // ClipSeemFusion's image side from packed images (SAF_WIN_RGBL=0, read per call: from the frames' own images -- the A/B)
  const bool rgbl_on = wl.cmax_off > wl.rgbl_off && kf0.rgb_bilinear && kf0.label_map && !(getenv("SAF_WIN_RGBL") && getenv("SAF_WIN_RGBL")[0] == '0');
  const bool sum = kv.accum == SAF_SUM;
  int img_vecs = (int)(wl.img_bytes / sizeof(float4));
  const int prep_blocks = (kv.D * (P + 1) + 255) / 256;
  const size_t aux_bytes = workspace_bytes > wl.cmax_off ? workspace_bytes - wl.cmax_off : 0;
  const bool brick_form = brick_form_ok(kv) && aux_bytes > 0 && brick_aux_fits(kv, aux_bytes);
  const int split = brick_form && brick_split() ? 1 : 0;
  WinFn fn = nullptr;
  size_t win_lds = 0;
  // SAF_WIN_MAPS16=0 (read per call): a bf16 volume keeps fp32 map images -- it takes the frame-ordered kernel, bit-identical
  // to the per-frame bf16 pipeline -- instead of the order-free form's bf16 images (one rounding of every tap to the volume's
  // precision, exact when the backbone emitted bf16: BASELINE config 3).  fp32 volumes are not affected.
  const char* m16 = getenv("SAF_WIN_MAPS16");
  const bool of = window_form_sums() && !(kv.bf16 != 0 && m16 && m16[0] == '0');
  if (!brick_form) switch (kv.D / 256) {
    case 1: fn = pick_win<1>(sum, kv.bf16 != 0, of, &win_lds); break;
    case 2: fn = pick_win<2>(sum, kv.bf16 != 0, of, &win_lds); break;
    case 3: fn = pick_win<3>(sum, kv.bf16 != 0, of, &win_lds); break;
    default: fn = pick_win<4>(sum, kv.bf16 != 0, of, &win_lds); break;
  }
  // bf16 volume in the order-free form: the window's map images are kept in bf16 (what the kernel's BF16 && OF instantiations read)
  const bool maps16 = !brick_form && of && kv.bf16 != 0 && SAF_WIN_MAPS16_BUILD;
  const size_t img_bytes16 = ((size_t)kv.D * (P + 1) * 2 + 255) & ~(size_t)255;
  if (maps16) img_vecs = (int)(img_bytes16 / sizeof(float4));
  if (!brick_form) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)win_lds);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute(LDS=%zu): %s", win_lds, hipGetErrorString(e));
  }
  float* maps = reinterpret_cast<float*>(ws + kHdrTotal);
  const int wgs_env = getenv("SAF_WIN_WGS") ? atoi(getenv("SAF_WIN_WGS")) : 0;
  const char* xcd_env = getenv("SAF_WIN_XCD");
  static_assert(kClsAccOff + kClsShards * 2 * sizeof(unsigned long long) <= kTableOff && kClsAccOff >= 256,
                "workspace header layout");
  // the depth tiles of the classification's occlusion cull: 16 x 16 pixels, doubled until a frame has at most kMaxDepthTiles
  int ts_log2 = 4;
  while (((kf0.W + (1 << ts_log2) - 1) >> ts_log2) * ((kf0.H + (1 << ts_log2) - 1) >> ts_log2) > kMaxDepthTiles) ++ts_log2;
  const int tiles_x = (kf0.W + (1 << ts_log2) - 1) >> ts_log2, n_tiles = tiles_x * ((kf0.H + (1 << ts_log2) - 1) >> ts_log2);
  int tile_window[kTileWindows];  // which window's tile maxima a slot of the tile region holds (-1: none)
  for (int k = 0; k < kTileWindows; ++k) tile_window[k] = -1;
  const int wlen = window_frames();
  const int n_win = (n_frames + wlen - 1) / wlen;
  auto win_frames = [&](int w) { return n_frames - w * wlen < wlen ? n_frames - w * wlen : wlen; };

  // ---- the units of this call
  std::vector<WinUnit> units;
  if (slabs && slabs->n > 0) {
    for (int k = 0; k < slabs->n; ++k) {
      if (slabs->x0[k] < 0 || slabs->nx[k] <= 0 || slabs->x0[k] + slabs->nx[k] > kv.nx) return fail(SAF_E_INVALID, "slab %d outside the volume", k);
      for (int w = 0; w < n_win; ++w)
        units.push_back(WinUnit{slab_kvol(kv, slabs->x0[k], slabs->nx[k]), w * wlen, win_frames(w), k == 0, w + 1 == n_win ? k : -1, k * n_win + w});
    }
  } else {
    // SAF_WIN_SLABS (read per call; default 1 = off): EVERY window slab by slab, window after window -- units of one size, so
    // that unit u + 1's classification is as long as unit u's row kernel is; SAF_WIN_W0_SLABS: only the first window (measured:
    // no gain -- the second window's whole classification then has only the last slab's row kernel to hide behind).  Slabs are
    // whole multiples of 16 x-planes (the row kernel's XCD-compact unit order, the classification's bricks).
    const char* e0 = getenv("SAF_WIN_W0_SLABS");
    const char* e1 = getenv("SAF_WIN_SLABS");
    auto fit = [&](int n) {
      if (!ov || brick_form || n < 2 || kv.nx % 16 != 0) return 1;
      while (n > 1 && (kv.nx / 16) % n != 0) --n;
      return n;
    };
    const int ns = fit(e1 ? atoi(e1) : 1), n0 = ns > 1 ? ns : fit(e0 ? atoi(e0) : 1);
    for (int w = 0; w < n_win; ++w) {
      const int n = w == 0 ? n0 : ns;
      for (int k = 0; k < n; ++k)
        units.push_back(WinUnit{n == 1 ? kv : slab_kvol(kv, k * (kv.nx / n), kv.nx / n), w * wlen, win_frames(w), k == 0, -1, w});
    }
  }
  const int n_units = (int)units.size();

  struct Geom {
    uint32_t cls_wgs, grid;  // workgroups of a classification launch (4 bricks each), of the row kernel
    int xcd_order;
  };
  auto geom = [&](const KVol& u) {
    Geom g;
    const uint32_t n_pieces = (uint32_t)(((int64_t)u.N + kPiece - 1) / kPiece);
    const uint32_t row_wgs = (n_pieces + kWinWaves - 1) / kWinWaves;
    // (the frame-ordered form keeps its rows in LDS: with 512-hit chunks 115 KB per workgroup at D = 512 -- ONE fits a CU, and a grid
    //  of two per CU would leave half of the persistent workgroups waiting for the others to finish)
    const int fit = win_lds > 0 ? (int)((160 * 1024) / win_lds) : 2;
    g.grid = (uint32_t)device_cus() * (wgs_env > 0 ? wgs_env : (of ? SAF_WIN_OF_WPE : (fit < 1 ? 1 : (fit > 2 ? 2 : fit))));
    if (g.grid > row_wgs) g.grid = row_wgs;
    // the classification's bricks: the brick grid padded to whole 8 x 8 tiles of brick columns
    const uint32_t tx = ((uint32_t)u.nx + 8 * kBrickX - 1) / (8 * kBrickX), ty = ((uint32_t)u.ny + 8 * kBrickY - 1) / (8 * kBrickY);
    const uint32_t nbz = ((uint32_t)u.nz + kBrickZ - 1) / kBrickZ;
    g.cls_wgs = (tx * ty * 64u * nbz + 3u) / 4u;
    // units of the row kernel in XCD-compact order (see the kernel); SAF_WIN_XCD=0: linear order
    g.xcd_order = !(xcd_env && xcd_env[0] == '0') && u.nx % 16 == 0 && u.ny % 16 == 0 && u.nz % kUnitVox == 0 && u.N % kPiece == 0 ? 1 : 0;
    return g;
  };

  // Two streams.  The classification (VALU-bound; TSDF, depth images, one mask plane per 32 frames) of unit u + 1 runs
  // on `cs` while the row kernel (memory-bound) of unit u runs on the caller's stream: the row kernel leaves LDS and
  // registers free for classification workgroups beside it.  Masks and header (counters, frame table) are double-
  // buffered by unit parity:
  //   classify(u) -> fuse(u)       event cls_done[u & 1]
  //   fuse(u) -> classify(u + 2)   event fuse_done[u & 1]  (same mask buffer and header)
  // Without `ov` everything is queued on the caller's stream in order.
  hipStream_t cs = ov ? ov->aux : s;
  if (ov) {
    if (hipEventRecord(ov->fork, s) != hipSuccess || hipStreamWaitEvent(cs, ov->fork, 0) != hipSuccess)
      return fail(SAF_E_HIP, "windowed path: could not fork the classification stream");
  }
  const bool trace = getenv("SAF_WIN_TRACE") != nullptr;  // development: host-side timeline of this call on stderr
  const auto t_call = std::chrono::steady_clock::now();
  auto mark = [&](const char* what, int w) {
    if (trace)
      fprintf(stderr, "[win trace] %8.3f ms  %s %d\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(), what, w);
  };
  // The depth tile maxima (and tiled copies) of one window: depth_max_kernel + depth_reduce_kernel per 32 frames, on stream `st`.
  auto depth_tiles = [&](int widx, int f0, int F, hipStream_t st) {
    float* dmax_w = reinterpret_cast<float*>(ws + wl.tile_off + (size_t)(widx % kTileWindows) * wl.tile_win);
    float* tmax_w = dmax_w + 1024;
    float* tdepth_w = tmax_w + (size_t)kWin * kMaxDepthTiles;
    for (int fb = 0; fb < F; fb += kClsFrames) {
      ClsArgs ca;
      ca.n = fb + kClsFrames < F ? kClsFrames : F - fb;
      ca.H = kf0.H; ca.W = kf0.W;
      for (int k = 0; k < kClsFrames; ++k) ca.depth[k] = frames[f0 + fb + (k < ca.n ? k : 0)].depth;
      float* tmax = tmax_w + (size_t)fb * kMaxDepthTiles;
      hipLaunchKernelGGL(depth_max_kernel, dim3((n_tiles + 3) / 4, ca.n), dim3(256), 0, st, ca, ts_log2, tiles_x, n_tiles, tmax,
                         tiled ? tdepth_w + (size_t)fb * dpx : nullptr, (kf0.W + (1 << SAF_CLS_TILE_WL2) - 1) >> SAF_CLS_TILE_WL2, (int)dpx);
      hipLaunchKernelGGL(depth_reduce_kernel, dim3(ca.n), dim3(256), 0, st, tmax, n_tiles, dmax_w + fb);
    }
  };
  // The later windows' tiles AHEAD of time, on the caller's stream: it is idle until the first window has been classified, and
  // every such pair of small launches inside the classification chain (16 per 512-frame job, ~70 us each beside a row kernel)
  // lengthens the chain that a job's time follows.  Window 0's stay in front of its classification.
  const bool pre_tiles = ov && ov->tiles && !(slabs && slabs->n > 0) && n_units == n_win && n_win >= 2 &&
                         !(getenv("SAF_WIN_PRETILES") && getenv("SAF_WIN_PRETILES")[0] == '0');
  if (pre_tiles) {
    for (int w = 1; w < n_win && w < kTileWindows; ++w) {
      depth_tiles(w, w * wlen, win_frames(w), s);
      tile_window[w % kTileWindows] = w;
    }
    if (hipEventRecord(ov->tiles, s) != hipSuccess) return fail(SAF_E_HIP, "hipEventRecord(tiles)");
  }
  auto classify = [&](int ui) -> int {
    const WinUnit& u = units[ui];
    const Geom g = geom(u.kv);
    const int F = u.F, f0 = u.f0, par = ui & 1;
    unsigned char* hdr = ws + (size_t)par * kHdrBytes;
    uint32_t* masks = reinterpret_cast<uint32_t*>(ws + kHdrTotal + wl.maps_bytes + (size_t)par * wl.mask_bytes);
    // the window's depth tile maxima: computed when a unit of the window first needs them
    const int widx = f0 / wlen, tslot = widx % kTileWindows;
    float* dmax_w = reinterpret_cast<float*>(ws + wl.tile_off + (size_t)tslot * wl.tile_win);  // [kWin] largest, [kWin] smallest
    float* tmax_w = dmax_w + 1024;
    float* tdepth_w = tmax_w + (size_t)kWin * kMaxDepthTiles;  // (tiled layout only) the window's depth images in tiles
    const bool tiles_cached = tile_window[tslot] == widx;
    tile_window[tslot] = widx;
    unsigned long long* cls_acc = reinterpret_cast<unsigned long long*>(hdr + kClsAccOff);
    WinTable* tab = reinterpret_cast<WinTable*>(hdr + kTableOff);
    mark("classify: begin", ui);
    if (ov && ui >= 2 && hipStreamWaitEvent(cs, ov->fuse_done[par], 0) != hipSuccess) return fail(SAF_E_HIP, "hipStreamWaitEvent");
    if (pre_tiles && ui == 1 && hipStreamWaitEvent(cs, ov->tiles, 0) != hipSuccess) return fail(SAF_E_HIP, "hipStreamWaitEvent(tiles)");
    // header: unit counters, dmax, the classification launches' counter shards, the frame table
    if (hipMemsetAsync(hdr, 0, kHdrBytes, cs) != hipSuccess) return fail(SAF_E_HIP, "hipMemsetAsync(workspace header)");
    mark("classify: header memset queued", ui);
    for (int fb = 0; fb < F; fb += kClsFrames) {
      ClsArgs ca;
      ca.n = fb + kClsFrames < F ? kClsFrames : F - fb;
      ca.H = kf0.H; ca.W = kf0.W; ca.slot = fb; ca.count = u.count;
      ca.guard_x = kf0.W <= 8192 ? (float)kf0.W * SAF_CLS_GUARD_EPS : 2.0f;
      ca.guard_y = kf0.H <= 8192 ? (float)kf0.H * SAF_CLS_GUARD_EPS : 2.0f;
      ca.mid_x = (float)(kf0.W - 1) * 0.5f; ca.mid_y = (float)(kf0.H - 1) * 0.5f;
      ca.verify = stats ? reinterpret_cast<unsigned long long*>(stats) + 7 : nullptr;
      ca.tiles_x8 = (kf0.W + (1 << SAF_CLS_TILE_WL2) - 1) >> SAF_CLS_TILE_WL2;  // tiles per image row
      // (unit 0 has the chip to itself -- everything the caller queued before is done, nothing of this call runs yet --, where the
      //  classification is bound by its vector instructions and the tile offset costs 5 % (0.92 vs 0.97 ms per launch): it reads the
      //  frames' own images; the tiled copies pay where the address path is shared, i.e. for every later unit)
      const bool use_tiled = tiled && (ui > 0 || (til_env && til_env[0] == '2'));
      ca.depth_bytes = use_tiled ? (int)(dpx * sizeof(float)) : kf0.H * kf0.W * 4;
      for (int k = 0; k < kClsFrames; ++k) {
        const saf_frame& fr = frames[f0 + fb + (k < ca.n ? k : 0)];
        ca.depth[k] = fr.depth; ca.rgb[k] = fr.rgb; ca.pose[k] = fr.pose; ca.K[k] = fr.K; ca.label_map[k] = fr.label_map;
        ca.feat_map[k] = fr.feat_map;
      }
      uint32_t* plane = masks + (size_t)(fb / kClsFrames) * wl.mask_plane;
      // the frames' largest depths feed the bricks' frame cull
      float* dmax = dmax_w + fb;
      float* tmax = tmax_w + (size_t)fb * kMaxDepthTiles;
      float* tdepth = tdepth_w + (size_t)fb * dpx;
      if (!tiles_cached && fb == 0) depth_tiles(widx, f0, F, cs);  // (reads the frames' own images; writes the tile maxima and, in the tiled layout, the copies)
      if (use_tiled)
        for (int k = 0; k < kClsFrames; ++k) ca.depth[k] = tdepth + (size_t)(k < ca.n ? k : 0) * dpx;
      ScopedPair t(prof, 1, f0 + fb, cs);
      // SAF_CLS_VERIFY=1 (read per call): the self-checking classification -- every voxel slot computes the reference's pixel chain
      // as well and counts disagreements with the guarded path in stats[7] (tests; tools/cls_guard_verify.py)
      const bool verify = getenv("SAF_CLS_VERIFY") && getenv("SAF_CLS_VERIFY")[0] == '1' && stats;
      auto kfn = use_tiled ? (verify ? (sum ? classify_bricks_kernel<true, true, true> : classify_bricks_kernel<false, true, true>)
                                 : (sum ? classify_bricks_kernel<true, false, true> : classify_bricks_kernel<false, false, true>))
                       : (verify ? (sum ? classify_bricks_kernel<true, true, false> : classify_bricks_kernel<false, true, false>)
                                 : (sum ? classify_bricks_kernel<true, false, false> : classify_bricks_kernel<false, false, false>));
      hipLaunchKernelGGL(kfn, dim3(g.cls_wgs), dim3(256), 0, cs, u.kv, ca, dmax, tmax, ts_log2, tiles_x, plane,
                         reinterpret_cast<unsigned long long*>(stats), cls_acc, tab);
    }
    int r = check_launch("classify_bricks_kernel");
    if (r) return r;
    if (split) {  // the brick form's build kernel: the window's hit records, sorted groups and scalar side, into the segment pool
      WinArgs wa;
      wa.F = F; wa.H = kf0.H; wa.W = kf0.W; wa.npy = kf0.npy; wa.npx = kf0.npx; wa.rgb_bilinear = kf0.rgb_bilinear;
      wa.rgbl = nullptr; wa.rgbl_px = 0; wa.rgbl_tiles_x = 0;
      ScopedPair t(prof, 3, f0, cs);
      if ((r = launch_brick_build(u.kv, wa, tab, wl.img_bytes, reinterpret_cast<unsigned long long*>(stats), masks, wl.mask_plane,
                                  ws + wl.cmax_off, aux_bytes, par, cs)))
        return r;
    }
    mark("classify: launches queued", ui);
    if (ov && hipEventRecord(ov->cls_done[par], cs) != hipSuccess) return fail(SAF_E_HIP, "hipEventRecord");
    return SAF_OK;
  };
  if ((rc = classify(0))) return rc;
  // A recycled volume (saf_fuse_frames_recycled): the rows of the voxels that are still unwritten when the call is over have to be
  // zeroed -- on a coherent scene five sixths of the volume, 28 GB of stores at 256^3 x 512 and a tenth of the job when they follow
  // it.  When every unit covers the whole volume they are written on the classification stream BESIDE the last unit's row kernel
  // instead (that stream has nothing left to do): by then every earlier unit has updated `weight`, and the last unit's hit masks
  // say which rows its row kernel writes -- the two kernels' rows are disjoint.  What it buys is small (the row kernel slows down
  // by nearly what the clear takes: saf_misc.hip, clear_rows).  SAF_WIN_CLEAR_BESIDE=0 (read per call): behind the last row
  // kernel, on the caller's stream.
  const bool clear_beside = recycled && ov && !brick_form && !(slabs && slabs->n > 0) && n_units == n_win &&
                            !(getenv("SAF_WIN_CLEAR_BESIDE") && getenv("SAF_WIN_CLEAR_BESIDE")[0] == '0');
  int maps_of = -1;  // the window whose map images the workspace holds
  for (int ui = 0; ui < n_units && rc == SAF_OK; ++ui) {
    const WinUnit& u = units[ui];
    const Geom g = geom(u.kv);
    const int F = u.F, f0 = u.f0, par = ui & 1;
    WinArgs wa;
    wa.F = F; wa.H = kf0.H; wa.W = kf0.W; wa.npy = kf0.npy; wa.npx = kf0.npx; wa.rgb_bilinear = kf0.rgb_bilinear;
    wa.rgbl = rgbl_on && !brick_form ? reinterpret_cast<const float4*>(ws + wl.rgbl_off) : nullptr;
    wa.rgbl_px = (int)rgbl_px_padded(kf0.H, kf0.W); wa.rgbl_tiles_x = (kf0.W + 3) >> 2;
    unsigned char* hdr = ws + (size_t)par * kHdrBytes;
    const WinTable* tab = reinterpret_cast<const WinTable*>(hdr + kTableOff);
    uint32_t* masks = reinterpret_cast<uint32_t*>(ws + kHdrTotal + wl.maps_bytes + (size_t)par * wl.mask_bytes);
    if (ov && ui + 1 < n_units && (rc = classify(ui + 1))) break;  // queued now: it runs beside this unit's row kernel
    if (ov && hipStreamWaitEvent(s, ov->cls_done[par], 0) != hipSuccess) { rc = fail(SAF_E_HIP, "hipStreamWaitEvent"); break; }
    if (clear_beside && ui + 1 == n_units) {
      // (queued behind this unit's classification; the row kernel of the unit before must have stored its weights)
      if (ui >= 1 && hipStreamWaitEvent(cs, ov->fuse_done[par ^ 1], 0) != hipSuccess) { rc = fail(SAF_E_HIP, "hipStreamWaitEvent"); break; }
      if ((rc = launch_clear_unwritten(u.kv, masks, wl.mask_plane, (F + kClsFrames - 1) / kClsFrames, cs))) break;
    }
    if (maps_of != u.window) {  // (the slabs of one window share its map images)
      ScopedPair t(prof, 0, f0, s);
      hipLaunchKernelGGL(prep_rows_kernel, dim3(prep_blocks, F), dim3(256), 0, s, tab, static_cast<void*>(maps),
                         maps16 ? (int)(img_bytes16 / 2) : (int)(wl.img_bytes / sizeof(float)), kv.D, P, maps16 ? 1 : 0);
      if ((rc = check_launch("prep_rows_kernel"))) break;
      if (wa.rgbl) {
        hipLaunchKernelGGL(prep_rgbl_kernel, dim3((kf0.H * kf0.W + 255) / 256, F), dim3(256), 0, s, tab, const_cast<float4*>(wa.rgbl), kf0.H, kf0.W,
                           wa.rgbl_tiles_x, wa.rgbl_px);
        if ((rc = check_launch("prep_rgbl_kernel"))) break;
      }
      maps_of = u.window;
    }
    if (brick_form) {
      ScopedPair t(prof, 2, f0, s);
      rc = launch_fuse_bricks(u.kv, wa, tab, maps, wl.img_bytes, reinterpret_cast<unsigned long long*>(stats),
                              reinterpret_cast<unsigned int*>(hdr), masks, wl.mask_plane,
                              reinterpret_cast<const unsigned long long*>(hdr + kClsAccOff),
                              ws + wl.cmax_off, aux_bytes, par, split, s);
    } else {
      ScopedPair t(prof, 2, f0, s);
      hipLaunchKernelGGL(fn, dim3(g.grid), dim3(kWinThreads), win_lds, s, u.kv, wa, tab, maps, img_vecs,
                         reinterpret_cast<unsigned long long*>(stats), reinterpret_cast<unsigned int*>(hdr), masks, wl.mask_plane,
                         reinterpret_cast<const unsigned long long*>(hdr + kClsAccOff), g.xcd_order);
      rc = check_launch("fuse_window_kernel");
    }
    if (rc) break;
    mark("rows: queued", ui);
    // a finished slab of a recycled volume: its rows that are still unwritten are zeroed before anyone is told it is finished
    if (u.done >= 0 && recycled && (rc = launch_clear_unwritten(u.kv, nullptr, 0, 0, s))) break;
    if (u.done >= 0 && slabs && slabs->done && slabs->done[u.done] &&
        hipEventRecord(static_cast<hipEvent_t>(slabs->done[u.done]), s) != hipSuccess) { rc = fail(SAF_E_HIP, "hipEventRecord(slab done)"); break; }
    if (ov && hipEventRecord(ov->fuse_done[par], s) != hipSuccess) { rc = fail(SAF_E_HIP, "hipEventRecord"); break; }
    if (!ov && ui + 1 < n_units) rc = classify(ui + 1);
  }
  if (ov) {  // whatever was queued on the classification stream is ordered before later work of the caller (error paths too)
    if (hipEventRecord(ov->join, cs) == hipSuccess) (void)hipStreamWaitEvent(s, ov->join, 0);
  }
  if (recycled && !clear_beside && !(slabs && slabs->n > 0) && rc == SAF_OK) rc = launch_clear_unwritten(kv, nullptr, 0, 0, s);
#ifdef SAF_WIN_TIMING
  {
    (void)hipStreamSynchronize(s);
    unsigned long long t[16];
    if (hipMemcpyFromSymbol(t, HIP_SYMBOL(g_win_t), sizeof(t)) == hipSuccess) {
      unsigned long long tot = 0;
      for (int k = 0; k < 8; ++k) tot += t[k];
      fprintf(stderr, "[win timing] masks %.1f%% expand %.1f%% project %.1f%% scalars %.1f%% records + row issue %.1f%% groups %.1f%% tap batches %.1f%% row store %.1f%% (total %.3g wave-cycles)\n",
              100.0 * t[0] / tot, 100.0 * t[1] / tot, 100.0 * t[2] / tot, 100.0 * t[3] / tot, 100.0 * t[4] / tot,
              100.0 * t[5] / tot, 100.0 * t[6] / tot, 100.0 * t[7] / tot, (double)tot);
      if (t[9]) fprintf(stderr, "[win timing] shader clock during the row kernels of this call: %.3f GHz (s_memtime / s_memrealtime x 100 MHz)\n", 0.1 * (double)t[8] / (double)t[9]);
      memset(t, 0, sizeof(t));
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_win_t), t, sizeof(t));
    }
  }
#endif
  return rc;
}


// ---------------------------------------------------------------------------------------------
// Streaming sessions (saf_fuse_session_*, round 6): the one-frame-per-integrate() queue hands its frames over 32 at a time.
//
// A separate saf_fuse_frames call per flushed window exposes, every time, the window's whole classification (3.7 ms at 256^3:
// nothing of the call runs beside it) -- and the queue cannot flush before the window's 128th frame has been staged (3 ms).  A
// session keeps ONE pipeline over its calls and classifies as the frames arrive: every push of 32 frames is one classification
// launch (one mask plane) on the classification stream; when a window's last plane is queued its row kernel follows on the
// caller's stream, and the next window's launches -- pushed while it runs -- run beside it, as units u and u + 1 of one
// fuse_many_windowed call do.  Same kernels, same arguments per launch, same window cuts: results are bit for bit those of one
// saf_fuse_frames call over the same frames.  (The row forms only: the brick form builds its segments per window.)
// ---------------------------------------------------------------------------------------------
namespace {
struct StreamPlan {  // what fuse_many_windowed derives at its top, for one (volume, frame shape, workspace)
  WinLayout wl;
  bool tiled, sum, maps16;
  size_t dpx, img_bytes16, win_lds;
  int P, img_vecs, prep_blocks, ts_log2, tiles_x, n_tiles, wlen;
  WinFn fn;
};
int stream_plan(const KVol& kv, const KFrame& kf0, size_t workspace_bytes, StreamPlan* pl) {
  pl->P = kf0.npy * kf0.npx;
  pl->dpx = depth_px_padded(kf0.H, kf0.W);
  const WinLayout wl_lin = win_layout(kv.N, kv.D, pl->P),
                  wl_til = win_layout(kv.N, kv.D, pl->P, false, pl->dpx, kv.labels ? rgbl_px_padded(kf0.H, kf0.W) : 0);
  const char* til_env = getenv("SAF_CLS_TILED");
  pl->tiled = !(til_env && til_env[0] == '0') && pl->dpx * sizeof(float) < (size_t)1 << 31 && workspace_bytes >= wl_til.cmax_off;
  pl->wl = pl->tiled ? wl_til : wl_lin;
  if (workspace_bytes < pl->wl.cmax_off) return fail(SAF_E_WORKSPACE, "session: workspace too small");
  pl->sum = kv.accum == SAF_SUM;
  pl->img_vecs = (int)(pl->wl.img_bytes / sizeof(float4));
  pl->prep_blocks = (kv.D * (pl->P + 1) + 255) / 256;
  const char* m16 = getenv("SAF_WIN_MAPS16");
  const bool of = window_form_sums() && !(kv.bf16 != 0 && m16 && m16[0] == '0');
  switch (kv.D / 256) {
    case 1: pl->fn = pick_win<1>(pl->sum, kv.bf16 != 0, of, &pl->win_lds); break;
    case 2: pl->fn = pick_win<2>(pl->sum, kv.bf16 != 0, of, &pl->win_lds); break;
    case 3: pl->fn = pick_win<3>(pl->sum, kv.bf16 != 0, of, &pl->win_lds); break;
    default: pl->fn = pick_win<4>(pl->sum, kv.bf16 != 0, of, &pl->win_lds); break;
  }
  if (!pl->fn) return fail(SAF_E_UNSUPPORTED, "session: no row kernel for this width");
  pl->maps16 = of && kv.bf16 != 0 && SAF_WIN_MAPS16_BUILD;
  pl->img_bytes16 = ((size_t)kv.D * (pl->P + 1) * 2 + 255) & ~(size_t)255;
  if (pl->maps16) pl->img_vecs = (int)(pl->img_bytes16 / sizeof(float4));
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pl->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->win_lds);
  if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute(LDS=%zu): %s", pl->win_lds, hipGetErrorString(e));
  pl->ts_log2 = 4;
  while (((kf0.W + (1 << pl->ts_log2) - 1) >> pl->ts_log2) * ((kf0.H + (1 << pl->ts_log2) - 1) >> pl->ts_log2) > kMaxDepthTiles) ++pl->ts_log2;
  pl->tiles_x = (kf0.W + (1 << pl->ts_log2) - 1) >> pl->ts_log2;
  pl->n_tiles = pl->tiles_x * ((kf0.H + (1 << pl->ts_log2) - 1) >> pl->ts_log2);
  pl->wlen = window_frames();
  return SAF_OK;
}
}  // namespace

bool stream_ok(const KVol& kv, const saf_frame* frames, int32_t n_frames, size_t workspace_bytes) {
  // what the windowed ROW forms take (window_ok's shape rules without its minimum of 16 frames per call: a push may be short)
  if (brick_form_ok(kv)) return false;
  if (getenv("SAF_WINDOW") && getenv("SAF_WINDOW")[0] == '0') return false;
  if (kv.bf16 && getenv("SAF_WINDOW_BF16") && getenv("SAF_WINDOW_BF16")[0] == '0') return false;
  if (kv.D % 256 != 0 || kv.D > 1024 || (kv.bf16 && kv.D % 512 != 0) || n_frames < 1) return false;
  const saf_frame& f0 = frames[0];
  for (int32_t i = 1; i < n_frames; ++i) {
    const saf_frame& f = frames[i];
    if (f.height != f0.height || f.width != f0.width || f.npy != f0.npy || f.npx != f0.npx || f.rgb_bilinear != f0.rgb_bilinear ||
        (f.label_map == nullptr) != (f0.label_map == nullptr))
      return false;
  }
  if (f0.npx + 3 > 255 || f0.npy + 3 > 255) return false;
  const WinLayout wl = win_layout(kv.N, kv.D, f0.npy * f0.npx);
  return wl.maps_bytes < (size_t)kTapOutside && workspace_bytes >= wl.cmax_off;
}

// the open window's row kernel: behind its last classification launch
int stream_close(void* workspace, size_t workspace_bytes, uint64_t* stats, hipStream_t s, const WinOverlap* ov, WinStream* st, bool preopen) {
  if (!st->open || st->filled == 0) return SAF_OK;
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  StreamPlan pl;
  int rc = stream_plan(st->kv, st->kf0, workspace_bytes, &pl);
  if (rc) return rc;
  const KVol& kv = st->kv;
  const int par = st->n_windows & 1, F = st->filled;
  if (hipEventRecord(ov->cls_done[par], ov->aux) != hipSuccess || hipStreamWaitEvent(s, ov->cls_done[par], 0) != hipSuccess)
    return fail(SAF_E_HIP, "session: could not order the row kernel behind its classification");
  unsigned char* hdr = ws + (size_t)par * kHdrBytes;
  const WinTable* tab = reinterpret_cast<const WinTable*>(hdr + kTableOff);
  uint32_t* masks = reinterpret_cast<uint32_t*>(ws + kHdrTotal + pl.wl.maps_bytes + (size_t)par * pl.wl.mask_bytes);
  float* maps = reinterpret_cast<float*>(ws + kHdrTotal);
  WinArgs wa;
  wa.F = F; wa.H = st->kf0.H; wa.W = st->kf0.W; wa.npy = st->kf0.npy; wa.npx = st->kf0.npx; wa.rgb_bilinear = st->kf0.rgb_bilinear;
  const bool rgbl_on = pl.wl.cmax_off > pl.wl.rgbl_off && st->kf0.rgb_bilinear && st->kf0.label_map &&
                       !(getenv("SAF_WIN_RGBL") && getenv("SAF_WIN_RGBL")[0] == '0');
  wa.rgbl = rgbl_on ? reinterpret_cast<const float4*>(ws + pl.wl.rgbl_off) : nullptr;
  wa.rgbl_px = (int)rgbl_px_padded(st->kf0.H, st->kf0.W); wa.rgbl_tiles_x = (st->kf0.W + 3) >> 2;
  hipLaunchKernelGGL(prep_rows_kernel, dim3(pl.prep_blocks, F), dim3(256), 0, s, tab, static_cast<void*>(maps),
                     pl.maps16 ? (int)(pl.img_bytes16 / 2) : (int)(pl.wl.img_bytes / sizeof(float)), kv.D, pl.P, pl.maps16 ? 1 : 0);
  if ((rc = check_launch("prep_rows_kernel"))) return rc;
  if (wa.rgbl) {
    hipLaunchKernelGGL(prep_rgbl_kernel, dim3((wa.H * wa.W + 255) / 256, F), dim3(256), 0, s, tab, const_cast<float4*>(wa.rgbl), wa.H, wa.W,
                       wa.rgbl_tiles_x, wa.rgbl_px);
    if ((rc = check_launch("prep_rgbl_kernel"))) return rc;
  }
  // (the row kernel's grid and unit order: as fuse_many_windowed's geom())
  const bool of = window_form_sums() && !(kv.bf16 != 0 && getenv("SAF_WIN_MAPS16") && getenv("SAF_WIN_MAPS16")[0] == '0');
  const int wgs_env = getenv("SAF_WIN_WGS") ? atoi(getenv("SAF_WIN_WGS")) : 0;
  const char* xcd_env = getenv("SAF_WIN_XCD");
  const uint32_t n_pieces = (uint32_t)(((int64_t)kv.N + kPiece - 1) / kPiece), row_wgs = (n_pieces + kWinWaves - 1) / kWinWaves;
  uint32_t grid = (uint32_t)device_cus() * (wgs_env > 0 ? wgs_env : (of ? SAF_WIN_OF_WPE : 2));
  if (grid > row_wgs) grid = row_wgs;
  const int xcd_order = !(xcd_env && xcd_env[0] == '0') && kv.nx % 16 == 0 && kv.ny % 16 == 0 && kv.nz % kUnitVox == 0 && kv.N % kPiece == 0 ? 1 : 0;
  hipLaunchKernelGGL(pl.fn, dim3(grid), dim3(kWinThreads), pl.win_lds, s, kv, wa, tab, maps, pl.img_vecs,
                     reinterpret_cast<unsigned long long*>(stats), reinterpret_cast<unsigned int*>(hdr), masks, pl.wl.mask_plane,
                     reinterpret_cast<const unsigned long long*>(hdr + kClsAccOff), xcd_order);
  if ((rc = check_launch("fuse_window_kernel"))) return rc;
  if (hipEventRecord(ov->fuse_done[par], s) != hipSuccess) return fail(SAF_E_HIP, "hipEventRecord");
  st->n_windows += 1;
  st->filled = 0;
  st->open = false;
  if (preopen) {
    // The NEXT window is opened here, on the classification stream, right behind this window's last launch: its header's wait
    // (the row kernel of two windows ago) and its memset are then out of the way when the next frames arrive, and the first
    // classification launch of window w + 1 reaches the chip a few microseconds BEFORE the row kernel of window w (which still has a
    // cross-stream wait and prep_rows_kernel in front of it) -- as in one saf_fuse_frames call, where that launch runs at its
    // alone speed while the row kernel's workgroups find their places (profiles/r06/api_b1_timeline.txt: 1.05 ms against 8.4).
    const int parn = st->n_windows & 1;
    unsigned char* hdrn = ws + (size_t)parn * kHdrBytes;
    if (st->n_windows >= 2 && hipStreamWaitEvent(ov->aux, ov->fuse_done[parn], 0) != hipSuccess) return fail(SAF_E_HIP, "hipStreamWaitEvent");
    if (hipMemsetAsync(hdrn, 0, kHdrBytes, ov->aux) != hipSuccess) return fail(SAF_E_HIP, "hipMemsetAsync(workspace header)");
    st->open = true;
  }
  return SAF_OK;
}

// The depth tiles (maxima, tiled copies) of frames that WILL be pushed next, in order, computed on the stream they were staged on --
// a call ahead of their push: by the time the host pushes them (after staging the next 32), an event recorded behind this call has
// long completed, the push finds that out with hipEventQuery and queues its classification launch with NO cross-stream wait in front
// (a barrier packet between two launches of the classification chain is 0.12 ms of idle chain: profiles/r06/api_b1_timeline.txt).
int stream_prepare(const KVol& kv, const saf_frame* frames, int32_t n_frames, void* workspace, size_t workspace_bytes, hipStream_t ts,
                   WinStream* st) {
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  int rc = SAF_OK;
  KFrame kf0;
  if ((rc = make_kframe(&frames[0], &kf0))) return rc;
  StreamPlan pl;
  if ((rc = stream_plan(kv, kf0, workspace_bytes, &pl))) return rc;
  if (st->prepared < st->pushed) st->prepared = st->pushed;
  int done = 0;
  while (done < n_frames) {
    // where frame number `prepared` of the session will lie: window = closed windows + what is still to be pushed ahead of it
    const long long ahead = st->prepared - st->pushed + (st->open ? st->filled : 0);
    const int widx = st->n_windows + (int)(ahead / pl.wlen), fb = (int)(ahead % pl.wlen), tslot = widx % kTileWindows;
    float* dmax_w = reinterpret_cast<float*>(ws + pl.wl.tile_off + (size_t)tslot * pl.wl.tile_win);
    float* tmax_w = dmax_w + 1024;
    float* tdepth_w = tmax_w + (size_t)kWin * kMaxDepthTiles;
    ClsArgs ca;
    ca.n = n_frames - done < kClsFrames ? n_frames - done : kClsFrames;
    if (fb + ca.n > pl.wlen) ca.n = pl.wlen - fb;
    ca.H = kf0.H; ca.W = kf0.W;
    for (int k = 0; k < kClsFrames; ++k) ca.depth[k] = frames[done + (k < ca.n ? k : 0)].depth;
    float* tmax = tmax_w + (size_t)fb * kMaxDepthTiles;
    hipLaunchKernelGGL(depth_max_kernel, dim3((pl.n_tiles + 3) / 4, ca.n), dim3(256), 0, ts, ca, pl.ts_log2, pl.tiles_x, pl.n_tiles, tmax,
                       pl.tiled ? tdepth_w + (size_t)fb * pl.dpx : nullptr, (kf0.W + (1 << SAF_CLS_TILE_WL2) - 1) >> SAF_CLS_TILE_WL2, (int)pl.dpx);
    hipLaunchKernelGGL(depth_reduce_kernel, dim3(ca.n), dim3(256), 0, ts, tmax, pl.n_tiles, dmax_w + fb);
    if ((rc = check_launch("depth tiles"))) return rc;
    st->prepared += ca.n;
    done += ca.n;
  }
  return SAF_OK;
}

int stream_push(const KVol& kv, const saf_frame* frames, int32_t n_frames, void* workspace, size_t workspace_bytes, uint64_t* stats,
                hipStream_t s, hipEvent_t ready, hipStream_t tile_stream, const WinOverlap* ov, WinStream* st) {
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  int rc = SAF_OK;
  KFrame kf0;
  if ((rc = make_kframe(&frames[0], &kf0))) return rc;
  for (int32_t i = 1; i < n_frames; ++i) {
    KFrame t;
    if ((rc = make_kframe(&frames[i], &t))) return rc;
  }
  if (st->have_shape && (kf0.H != st->kf0.H || kf0.W != st->kf0.W || kf0.npy != st->kf0.npy || kf0.npx != st->kf0.npx ||
                         kf0.rgb_bilinear != st->kf0.rgb_bilinear || (kf0.label_map == nullptr) != (st->kf0.label_map == nullptr)))
    return fail(SAF_E_INVALID, "the frames of a streaming session share their shapes: finish the session first");
  if (st->open && st->filled % kClsFrames != 0)
    return fail(SAF_E_INVALID, "the open window holds %d frames: only a window's LAST push may be short of a multiple of %d (finish the session)", st->filled, kClsFrames);
  StreamPlan pl;
  if ((rc = stream_plan(kv, kf0, workspace_bytes, &pl))) return rc;
  st->kv = kv; st->kf0 = kf0; st->have_shape = true;
  hipStream_t cs = ov->aux;
  // What the classification of these frames waits for: their staging.  With `ready` (an event the caller recorded behind it) ONLY
  // that -- forking from `s` would also wait for the row kernel queued there, the very kernel these launches are to run beside
  // (the first version did: every window's classification started when the previous row kernel ended; api_b1 0.895 of bulk).
  // A session's first push forks from `s` as well: whatever the caller queued there before the session comes first.
  // (frames whose depth tiles stream_prepare has computed: if `ready` -- recorded behind that -- has COMPLETED, nothing is waited for)
  const bool prepared = st->prepared >= st->pushed + n_frames;
  const bool ready_done = prepared && ready && hipEventQuery(ready) == hipSuccess;
  if (prepared) tile_stream = nullptr;
  if ((!ready && !tile_stream) || (st->n_windows == 0 && !st->open && st->pushed == 0)) {
    if (hipEventRecord(ov->fork, s) != hipSuccess || hipStreamWaitEvent(cs, ov->fork, 0) != hipSuccess)
      return fail(SAF_E_HIP, "session: could not fork the classification stream");
  }
  // (with a tile stream the classification waits for the tiles' event, recorded THERE behind the frames' staging: one barrier per
  //  launch instead of two -- every cross-stream wait is ~0.05 ms of gap in the classification chain)
  if (ready && !tile_stream && !ready_done && hipStreamWaitEvent(cs, ready, 0) != hipSuccess) return fail(SAF_E_HIP, "session: hipStreamWaitEvent(ready)");
  // `tile_stream` (the stream the frames were staged on, idle otherwise): the launches' depth tile maxima and tiled copies are
  // computed THERE, behind the staging, and the classification waits for them -- two small launches per 32 frames (0.1 ms beside a
  // row kernel) that would otherwise sit in the classification chain, which a window's time follows
  hipStream_t ts = tile_stream ? tile_stream : cs;
  const char* til_env = getenv("SAF_CLS_TILED");
  const bool verify = getenv("SAF_CLS_VERIFY") && getenv("SAF_CLS_VERIFY")[0] == '1' && stats;
  int done = 0;
  while (done < n_frames) {
    if (st->open && st->filled >= pl.wlen && (rc = stream_close(workspace, workspace_bytes, stats, s, ov, st, true))) return rc;
    const int par = st->n_windows & 1, widx = st->n_windows, tslot = widx % kTileWindows;
    unsigned char* hdr = ws + (size_t)par * kHdrBytes;
    uint32_t* masks = reinterpret_cast<uint32_t*>(ws + kHdrTotal + pl.wl.maps_bytes + (size_t)par * pl.wl.mask_bytes);
    float* dmax_w = reinterpret_cast<float*>(ws + pl.wl.tile_off + (size_t)tslot * pl.wl.tile_win);
    float* tmax_w = dmax_w + 1024;
    float* tdepth_w = tmax_w + (size_t)kWin * kMaxDepthTiles;
    unsigned long long* cls_acc = reinterpret_cast<unsigned long long*>(hdr + kClsAccOff);
    WinTable* tab = reinterpret_cast<WinTable*>(hdr + kTableOff);
    if (!st->open) {  // a window starts: its header (counters, frame table) and mask planes were the window's two before
      if (st->n_windows >= 2 && hipStreamWaitEvent(cs, ov->fuse_done[par], 0) != hipSuccess) return fail(SAF_E_HIP, "hipStreamWaitEvent");
      if (hipMemsetAsync(hdr, 0, kHdrBytes, cs) != hipSuccess) return fail(SAF_E_HIP, "hipMemsetAsync(workspace header)");
      st->open = true;
      st->filled = 0;
    }
    const int fb = st->filled;  // a multiple of 32 (checked above / by the loop)
    ClsArgs ca;
    ca.n = n_frames - done < kClsFrames ? n_frames - done : kClsFrames;
    if (fb + ca.n > pl.wlen) ca.n = pl.wlen - fb;
    ca.H = kf0.H; ca.W = kf0.W; ca.slot = fb; ca.count = 1;
    ca.guard_x = kf0.W <= 8192 ? (float)kf0.W * SAF_CLS_GUARD_EPS : 2.0f;
    ca.guard_y = kf0.H <= 8192 ? (float)kf0.H * SAF_CLS_GUARD_EPS : 2.0f;
    ca.mid_x = (float)(kf0.W - 1) * 0.5f; ca.mid_y = (float)(kf0.H - 1) * 0.5f;
    ca.verify = stats ? reinterpret_cast<unsigned long long*>(stats) + 7 : nullptr;
    ca.tiles_x8 = (kf0.W + (1 << SAF_CLS_TILE_WL2) - 1) >> SAF_CLS_TILE_WL2;
    // (the session's first window has the chip to itself: it reads the frames' own images, as a call's first unit does)
    const bool use_tiled = pl.tiled && (widx > 0 || (til_env && til_env[0] == '2'));
    ca.depth_bytes = use_tiled ? (int)(pl.dpx * sizeof(float)) : kf0.H * kf0.W * 4;
    for (int k = 0; k < kClsFrames; ++k) {
      const saf_frame& fr = frames[done + (k < ca.n ? k : 0)];
      ca.depth[k] = fr.depth; ca.rgb[k] = fr.rgb; ca.pose[k] = fr.pose; ca.K[k] = fr.K; ca.label_map[k] = fr.label_map;
      ca.feat_map[k] = fr.feat_map;
    }
    float* dmax = dmax_w + fb;
    float* tmax = tmax_w + (size_t)fb * kMaxDepthTiles;
    float* tdepth = tdepth_w + (size_t)fb * pl.dpx;
    // this launch's depth tile maxima (and tiled copies), from the frames' own images
    if (!prepared) {
      hipLaunchKernelGGL(depth_max_kernel, dim3((pl.n_tiles + 3) / 4, ca.n), dim3(256), 0, ts, ca, pl.ts_log2, pl.tiles_x, pl.n_tiles, tmax,
                         pl.tiled ? tdepth : nullptr, ca.tiles_x8, (int)pl.dpx);
      hipLaunchKernelGGL(depth_reduce_kernel, dim3(ca.n), dim3(256), 0, ts, tmax, pl.n_tiles, dmax);
      if (tile_stream && (hipEventRecord(ov->tiles, ts) != hipSuccess || hipStreamWaitEvent(cs, ov->tiles, 0) != hipSuccess))
        return fail(SAF_E_HIP, "session: could not order the classification behind its depth tiles");
    }
    if (use_tiled)
      for (int k = 0; k < kClsFrames; ++k) ca.depth[k] = tdepth + (size_t)(k < ca.n ? k : 0) * pl.dpx;
    uint32_t* plane = masks + (size_t)(fb / kClsFrames) * pl.wl.mask_plane;
    const uint32_t tx = ((uint32_t)kv.nx + 8 * kBrickX - 1) / (8 * kBrickX), ty = ((uint32_t)kv.ny + 8 * kBrickY - 1) / (8 * kBrickY);
    const uint32_t nbz = ((uint32_t)kv.nz + kBrickZ - 1) / kBrickZ, cls_wgs = (tx * ty * 64u * nbz + 3u) / 4u;
    auto kfn = use_tiled ? (verify ? (pl.sum ? classify_bricks_kernel<true, true, true> : classify_bricks_kernel<false, true, true>)
                                   : (pl.sum ? classify_bricks_kernel<true, false, true> : classify_bricks_kernel<false, false, true>))
                         : (verify ? (pl.sum ? classify_bricks_kernel<true, true, false> : classify_bricks_kernel<false, true, false>)
                                   : (pl.sum ? classify_bricks_kernel<true, false, false> : classify_bricks_kernel<false, false, false>));
    hipLaunchKernelGGL(kfn, dim3(cls_wgs), dim3(256), 0, cs, kv, ca, dmax, tmax, pl.ts_log2, pl.tiles_x, plane,
                       reinterpret_cast<unsigned long long*>(stats), cls_acc, tab);
    if ((rc = check_launch("classify_bricks_kernel"))) return rc;
    st->filled += ca.n;
    st->pushed += ca.n;
    done += ca.n;
  }
  // a full window's row kernel follows at once: the launches of the next pushes run beside it
  if (st->open && st->filled >= pl.wlen) rc = stream_close(workspace, workspace_bytes, stats, s, ov, st, true);
  return rc;
}

}  // namespace saf
