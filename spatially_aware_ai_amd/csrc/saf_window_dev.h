// saf_window_dev.h -- what the two row kernels of the windowed path (saf_window.hip: frame-ordered rows; saf_brick.hip:
// brick-resident rows with shared map taps) and the host side of saf_window.hip share: the workspace header layout,
// the window's frame table, and the launcher of the brick form.
#pragma once
#include "saf_fuse_dev.h"

namespace saf {

constexpr size_t kHdrBytes = 8192;   // one workspace header (unit counters, dmax, counter shards, the window's frame table)
constexpr size_t kHdrTotal = 2 * kHdrBytes;  // two of them, and two mask buffers: window w + 1 is classified while window w's rows are fused

constexpr int kWin = SAF_WINDOW_FRAMES;  // frames of the longest window: four 32-bit mask words per voxel
static_assert(kWin == 128, "the mask layout and the 7-bit frame field assume windows of up to 128 frames");
constexpr int kMaskWords = kWin / 32;
constexpr int kClsFrames = 32;  // frames of one classification launch = one mask plane

struct WinTable {
  const float* rgb[kWin];
  const float* pose[kWin];
  const float* K[kWin];
  const float* label_map[kWin];
  const float* feat_map[kWin];
};
struct WinArgs {  // what is common to a window's frames
  int F, H, W, npy, npx, rgb_bilinear;
  // (round 6) ClipSeemFusion's image side: the window's rgb and label images re-laid-out as ONE image per frame of 16-byte
  // {r, g, b, label} pixels in 4 x 2-pixel tiles of one cache line (prep_rgbl_kernel); NULL: the frames' own images
  const float4* rgbl;
  int rgbl_px, rgbl_tiles_x;  // padded pixels per frame, tiles per image row
};
// where pixel (x, y) lies in a frame's packed image
__device__ __forceinline__ int rgbl_offset(int x, int y, int tiles_x) { return (((y >> 1) * tiles_x + (x >> 2)) << 3) + ((y & 1) << 2) + (x & 3); }
inline size_t rgbl_px_padded(int H, int W) { return (size_t)((H + 1) >> 1) * (size_t)((W + 3) >> 2) * 8; }
// The same sample from a frame's PACKED image ({r, g, b, label} pixels, tiled: saf_window_dev.h rgbl_offset) -- ClipSeemFusion's
// bilinear rgb (clip_seem_fusion.py:793-798) as four 16-byte gathers instead of twelve 4-byte ones, and the panoptic class of the
// nearest pixel (:786-791) from the `.w` of the tap it coincides with (rint(x) is floor(x) or floor(x) + 1) instead of a
// thirteenth gather from a third image.  Same taps, same weights, same arithmetic: bit for bit sample_rgb_lane + the label lookup.
__device__ __forceinline__ int sample_rgbl_lane(const float4* __restrict__ img, int tiles_x, const KFrame& f, const Cam& cam, float gx,
                                                float gy, float& s0, float& s1, float& s2, float& lraw) {
  const float xn = __builtin_rintf(unnormalize(gx, cam.sfx)), yn = __builtin_rintf(unnormalize(gy, cam.sfy));
  const bool inb = (xn > -1.0f) && (xn < cam.fw) && (yn > -1.0f) && (yn < cam.fh);
  const int pix = inb ? (int)yn * f.W + (int)xn : -1;
  const Bilin bi = bilinear_setup(gx, gy, cam.sfx, cam.sfy);
  const bool x0ok = bi.x0 >= 0 && bi.x0 < f.W, x1ok = bi.x0 + 1 >= 0 && bi.x0 + 1 < f.W;
  const bool y0ok = bi.y0 >= 0 && bi.y0 < f.H, y1ok = bi.y0 + 1 >= 0 && bi.y0 + 1 < f.H;
  const bool knw = x0ok && y0ok, kne = x1ok && y0ok, ksw = x0ok && y1ok, kse = x1ok && y1ok;
  float4 nw = img[knw ? rgbl_offset(bi.x0, bi.y0, tiles_x) : 0], ne = img[kne ? rgbl_offset(bi.x0 + 1, bi.y0, tiles_x) : 0];
  float4 sw = img[ksw ? rgbl_offset(bi.x0, bi.y0 + 1, tiles_x) : 0], se = img[kse ? rgbl_offset(bi.x0 + 1, bi.y0 + 1, tiles_x) : 0];
  // the nearest pixel's class: the tap at (xn - x0, yn - y0)
  const bool east = inb && (int)xn != bi.x0, south = inb && (int)yn != bi.y0;
  lraw = south ? (east ? se.w : sw.w) : (east ? ne.w : nw.w);
  // out-of-image taps: value forced to +0 (x * 0 would keep NaN/inf of pixel 0 alive)
  if (!knw) nw = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!kne) ne = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!ksw) sw = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!kse) se = make_float4(0.f, 0.f, 0.f, 0.f);
  s0 = lerp_taps(nw.x, ne.x, sw.x, se.x, bi);
  s1 = lerp_taps(nw.y, ne.y, sw.y, se.y, bi);
  s2 = lerp_taps(nw.z, ne.z, sw.z, se.z, bi);
  return pix;
}

constexpr size_t kTableOff = 2048;  // WinTable in the workspace header
static_assert(kTableOff + sizeof(WinTable) <= kHdrBytes, "workspace header layout");

// Counters of a classification launch, sharded in the workspace header (see cls_accumulate in saf_window.hip); the
// window's row kernel folds the shards into stats[].
constexpr int kClsShards = 64;
constexpr size_t kClsAccOff = 1024;  // 64 x {tsdf updates, tsdf voxels} u64 in the workspace header
// ... and the bricks' frame cull (round 6): (brick, frame) pairs tested and dropped, by reason -- stats[8..12].  16 shards of
// {tested, behind the camera, beyond the frame's largest depth, outside the frustum, occluded} behind the frame table.
constexpr int kCullShards = 16, kCullWords = 5;
constexpr size_t kCullAccOff = kTableOff + sizeof(WinTable);
static_assert(kCullAccOff % 8 == 0 && kCullAccOff + kCullShards * kCullWords * sizeof(unsigned long long) <= kHdrBytes, "workspace header layout");
constexpr int kStatCull = 8;  // first of the five words in stats[]
// Layout: word k of shard j at k * kCullShards + j.  Folded by the first wave of a window's row kernel (`cls_acc` = header +
// kClsAccOff as the row kernels receive it; all 64 lanes of the wave call).
__device__ __forceinline__ void fold_cull_shards(const unsigned long long* __restrict__ cls_acc, unsigned long long* __restrict__ stats, int lane) {
  const unsigned long long* cull = cls_acc + (kCullAccOff - kClsAccOff) / sizeof(unsigned long long);
#pragma unroll
  for (int k = 0; k < kCullWords; ++k) {
    unsigned long long x = lane < kCullShards ? cull[k * kCullShards + lane] : 0ull;
    for (int o = kCullShards / 2; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if (lane == 0 && x) atomicAdd(&stats[kStatCull + k], x);
  }
}

constexpr uint32_t kTapOutside = 0x80000000u;  // byte offset of a tap outside the map: beyond any buffer

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The brick form of the window's row kernel (saf_brick.hip).  `ctr`: the header's unit counters (8 words, zeroed),
// `map_imgs`: the window's pixel-major map images (img_bytes each), `hitmask`: the window's mask planes.
// `aux`: the rest of the workspace (saf_fuse_workspace_bytes reserves brick_aux_bytes_est): the channels' largest magnitudes and, per window parity, the camera table and the
// pool of segments the build kernel (after the window's classification, on its stream) leaves for the walk kernel.
bool brick_form_ok(const KVol& kv);
bool brick_split();
size_t brick_aux_bytes_est(int64_t n_vox, int D);
bool brick_aux_fits(const KVol& kv, size_t avail);
int launch_brick_build(const KVol& kv, const WinArgs& wa, const WinTable* tab, size_t img_bytes, unsigned long long* stats,
                       const uint32_t* hitmask, uint32_t mask_plane, void* aux, size_t aux_bytes, int parity, hipStream_t s);
int launch_fuse_bricks(const KVol& kv, const WinArgs& wa, const WinTable* tab, const float* map_imgs, size_t img_bytes,
                       unsigned long long* stats, unsigned int* ctr, const uint32_t* hitmask, uint32_t mask_plane,
                       const unsigned long long* cls_acc, void* aux, size_t aux_bytes, int parity, int split, hipStream_t s);

}  // namespace saf
