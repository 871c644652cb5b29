// saf_fuse_dev.h -- device-side building blocks shared by the per-frame pipeline (saf_fuse.hip) and the
// windowed path (saf_window.hip): kernel descriptors, the bilinear blend arithmetic, the rgb / weight / label
// side of one valid voxel, and the host-side profiler pair.  Everything lives in an anonymous namespace: each
// translation unit gets its own copy.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "saf_common.h"
#include "saf_host.h"

#pragma clang fp contract(off)

// Pool of event pairs; opaque to callers (include/saf.h).
struct saf_profiler {
  struct Pair {
    hipEvent_t a, b;
    int cls;
  };
  Pair* pairs;
  int capacity, used;
  int stride;  // record only frames whose index within the call is a multiple of this
};

namespace saf {

// kernel-side descriptors (POD, passed by value; shared by both translation units)
struct KVol {
  int nx, ny, nz, D, n_classes, accum, bf16;
  uint32_t N;
  float trunc;
  const float *ax, *ay, *az;
  float* tsdf;
  int* tsdf_w;
  int* weight;
  float* rgb;
  float* feat;
  int* labels;
  FastDiv div_nz, div_ny;
};

struct KFrame {
  int H, W, npy, npx, rgb_bilinear;
  const float *depth, *rgb, *pose, *K, *label_map;
};

namespace {

FastDiv make_fastdiv(uint32_t d) {
  // q = (n * mul) >> shift is exact for every n < 2^31: with L = ceil(log2 d), S = 31 + L and
  // mul = ceil(2^S / d), the error term e = mul*d - 2^S is < d <= 2^L, so n*e < 2^S.
  uint32_t L = 0;
  while ((1ull << L) < d) ++L;
  FastDiv f;
  f.shift = 31 + L;
  f.mul = (uint32_t)(((1ull << f.shift) + d - 1) / d);
  f.d = d;
  f.pad = 0;
  return f;
}

__device__ __forceinline__ void voxel_coords(const KVol& v, uint32_t n, int& ix, int& iy, int& iz) {
  uint32_t t = fdiv(n, v.div_nz);
  iz = (int)(n - t * (uint32_t)v.nz);
  uint32_t x = fdiv(t, v.div_ny);
  iy = (int)(t - x * (uint32_t)v.ny);
  ix = (int)x;
}

// ------------------------------------------------------------------------------------------
// fuse: gather + running-mean RMW of the valid voxel rows
// ------------------------------------------------------------------------------------------
template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
};
template <>
struct VecT<1> {
  using type = float;
};

__device__ __forceinline__ float4 lerp_taps(float4 a, float4 b, float4 c, float4 d, const Bilin& w) {
  // (nw_val*nw + ne_val*ne) + sw_val*sw + se_val*se, left to right (GridSamplerKernel.cpp)
  float4 r;
  r.x = ((a.x * w.nw + b.x * w.ne) + c.x * w.sw) + d.x * w.se;
  r.y = ((a.y * w.nw + b.y * w.ne) + c.y * w.sw) + d.y * w.se;
  r.z = ((a.z * w.nw + b.z * w.ne) + c.z * w.sw) + d.z * w.se;
  r.w = ((a.w * w.nw + b.w * w.ne) + c.w * w.sw) + d.w * w.se;
  return r;
}
__device__ __forceinline__ float lerp_taps(float a, float b, float c, float d, const Bilin& w) {
  return ((a * w.nw + b * w.ne) + c * w.sw) + d * w.se;
}
__device__ __forceinline__ float4 blend(float4 s, float4 old, float a, float b, bool sum) {
  float4 r;
  if (sum) {
    r.x = old.x + s.x; r.y = old.y + s.y; r.z = old.z + s.z; r.w = old.w + s.w;
  } else {
    // clip_feat.T * a + self.clip_feat[valid] * b          clipfusion.py:720
    r.x = s.x * a + old.x * b; r.y = s.y * a + old.y * b;
    r.z = s.z * a + old.z * b; r.w = s.w * a + old.w * b;
  }
  return r;
}
__device__ __forceinline__ float blend(float s, float old, float a, float b, bool sum) {
  return sum ? old + s : s * a + old * b;
}

struct Taps {
  int o_nw, o_ne, o_sw, o_se;  // tap positions in [0, P]; P = the zero column (outside the map)
};
__device__ __forceinline__ Taps tap_offsets(const Bilin& b, int npx, int npy) {
  const bool x0 = b.x0 >= 0 && b.x0 < npx, x1 = b.x0 + 1 >= 0 && b.x0 + 1 < npx;
  const bool y0 = b.y0 >= 0 && b.y0 < npy, y1 = b.y0 + 1 >= 0 && b.y0 + 1 < npy;
  const int zero = npx * npy;
  Taps t;
  t.o_nw = (x0 && y0) ? b.y0 * npx + b.x0 : zero;
  t.o_ne = (x1 && y0) ? b.y0 * npx + b.x0 + 1 : zero;
  t.o_sw = (x0 && y1) ? (b.y0 + 1) * npx + b.x0 : zero;
  t.o_se = (x1 && y1) ? (b.y0 + 1) * npx + b.x0 + 1 : zero;
  return t;
}

// rgb / weight / label side of one valid voxel, done by lane `gl` of the group of `G` lanes.
__device__ __forceinline__ void fuse_scalars(const KVol& v, const KFrame& f, const Cam& cam, uint32_t n, float gx,
                                             float gy, int w0, float a, float b, int gl, int G,
                                             unsigned long long* stats) {
  const bool sum = v.accum == SAF_SUM;
  if (gl < 3) {
    const int pix = nearest_index(gx, gy, cam, f.W);
    Bilin bi;
    int x0ok = 0, x1ok = 0, y0ok = 0, y1ok = 0;
    if (f.rgb_bilinear) {
      bi = bilinear_setup(gx, gy, cam.sfx, cam.sfy);
      x0ok = bi.x0 >= 0 && bi.x0 < f.W;
      x1ok = bi.x0 + 1 >= 0 && bi.x0 + 1 < f.W;
      y0ok = bi.y0 >= 0 && bi.y0 < f.H;
      y1ok = bi.y0 + 1 >= 0 && bi.y0 + 1 < f.H;
    }
    for (int ch = gl; ch < 3; ch += G) {
      float s;
      if (f.rgb_bilinear) {  // clip_seem_fusion.py:793-798
        const float* img = f.rgb + ch;
        const int64_t r0 = (int64_t)bi.y0 * f.W, r1 = r0 + f.W;
        const float nw = (x0ok && y0ok) ? img[(r0 + bi.x0) * 3] : 0.f;
        const float ne = (x1ok && y0ok) ? img[(r0 + bi.x0 + 1) * 3] : 0.f;
        const float sw = (x0ok && y1ok) ? img[(r1 + bi.x0) * 3] : 0.f;
        const float se = (x1ok && y1ok) ? img[(r1 + bi.x0 + 1) * 3] : 0.f;
        s = lerp_taps(nw, ne, sw, se, bi);
      } else {  // clipfusion.py:701-706
        s = pix >= 0 ? f.rgb[(int64_t)pix * 3 + ch] : 0.f;
      }
      float* dst = v.rgb + (int64_t)n * 3 + ch;
      *dst = blend(s, *dst, a, b, sum);
    }
    if (gl == 0) {
      v.weight[n] = w0 + 1;  // clipfusion.py:715, :721
      if (v.labels && f.label_map) {
        // labels = grid_sample(pano_seg.float(), nearest); one_hot(labels.long())  clip_seem_fusion.py:786-822
        const float lf = pix >= 0 ? f.label_map[pix] : 0.f;
        const long long l = (long long)lf;
        if (l >= 0 && l < v.n_classes) {
          int* c = v.labels + (int64_t)n * v.n_classes + l;
          *c = *c + 1;
        } else if (stats) {
          atomicAdd(&stats[3], 1ull);
        }
      }
    }
  }
}

// Block until the sweep of this frame has published all its blocks (device-side dependency: no
// event packet sits between consecutive fuse kernels on the caller's stream).  The sweep never
// waits on anything and always fits beside a fuse workgroup, so this cannot deadlock; the spin is
// bounded anyway (~2 s of the 100 MHz wall clock) and reports through stats[4].
// rgb / weight / label side of one valid voxel handled entirely by ONE lane (the lane-parallel
// part of fuse_rows_kernel): the three channel loads are issued together, then blended and stored.
// The frame's rgb sample for one valid voxel (nearest in ClipFusion, bilinear in ClipSeemFusion); returns the
// nearest pixel index (-1 outside), which the label lookup shares.
__device__ __forceinline__ int sample_rgb_lane(const KFrame& f, const Cam& cam, float gx, float gy, float& s0, float& s1,
                                               float& s2) {
  const int pix = nearest_index(gx, gy, cam, f.W);
  if (f.rgb_bilinear) {  // clip_seem_fusion.py:793-798
    const Bilin bi = bilinear_setup(gx, gy, cam.sfx, cam.sfy);
    const bool x0ok = bi.x0 >= 0 && bi.x0 < f.W, x1ok = bi.x0 + 1 >= 0 && bi.x0 + 1 < f.W;
    const bool y0ok = bi.y0 >= 0 && bi.y0 < f.H, y1ok = bi.y0 + 1 >= 0 && bi.y0 + 1 < f.H;
    const int64_t r0 = (int64_t)bi.y0 * f.W, r1 = r0 + f.W;
    const float* pnw = f.rgb + ((x0ok && y0ok) ? (r0 + bi.x0) * 3 : 0);
    const float* pne = f.rgb + ((x1ok && y0ok) ? (r0 + bi.x0 + 1) * 3 : 0);
    const float* psw = f.rgb + ((x0ok && y1ok) ? (r1 + bi.x0) * 3 : 0);
    const float* pse = f.rgb + ((x1ok && y1ok) ? (r1 + bi.x0 + 1) * 3 : 0);
    const float mnw = (x0ok && y0ok) ? 1.f : 0.f, mne = (x1ok && y0ok) ? 1.f : 0.f;
    const float msw = (x0ok && y1ok) ? 1.f : 0.f, mse = (x1ok && y1ok) ? 1.f : 0.f;
    // out-of-image taps: value forced to +0 (x * 0 would keep NaN/inf of pixel 0 alive)
    float nw[3], ne[3], sw[3], se[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      nw[ch] = pnw[ch]; ne[ch] = pne[ch]; sw[ch] = psw[ch]; se[ch] = pse[ch];
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      nw[ch] = mnw != 0.f ? nw[ch] : 0.f; ne[ch] = mne != 0.f ? ne[ch] : 0.f;
      sw[ch] = msw != 0.f ? sw[ch] : 0.f; se[ch] = mse != 0.f ? se[ch] : 0.f;
    }
    s0 = lerp_taps(nw[0], ne[0], sw[0], se[0], bi);
    s1 = lerp_taps(nw[1], ne[1], sw[1], se[1], bi);
    s2 = lerp_taps(nw[2], ne[2], sw[2], se[2], bi);
  } else {  // clipfusion.py:701-706
    const float* px = f.rgb + (int64_t)(pix >= 0 ? pix : 0) * 3;
    const float t0 = px[0], t1 = px[1], t2 = px[2];
    s0 = pix >= 0 ? t0 : 0.f;
    s1 = pix >= 0 ? t1 : 0.f;
    s2 = pix >= 0 ? t2 : 0.f;
  }
  return pix;
}
// labels = grid_sample(pano_seg.float(), nearest); one_hot(labels.long())          clip_seem_fusion.py:786-822
// ATOMIC: several hits of one voxel may be counted by different lanes at the same time (integer adds commute).
template <bool ATOMIC>
__device__ __forceinline__ void count_label_lane(const KVol& v, const KFrame& f, uint32_t n, int pix,
                                                 unsigned long long* stats) {
  if (v.labels && f.label_map) {
    const float lraw = f.label_map[pix >= 0 ? pix : 0];
    const float lf = pix >= 0 ? lraw : 0.f;
    const long long l = (long long)lf;
    if (l >= 0 && l < v.n_classes) {
      int* c = v.labels + (int64_t)n * v.n_classes + l;
      if (ATOMIC)
        atomicAdd(c, 1);
      else
        *c = *c + 1;
    } else if (stats) {
      atomicAdd(&stats[3], 1ull);
    }
  }
}
__device__ __forceinline__ void fuse_scalars_lane(const KVol& v, const KFrame& f, const Cam& cam, uint32_t n,
                                                  float gx, float gy, int w0, float a, float b,
                                                  unsigned long long* stats) {
  const bool sum = v.accum == SAF_SUM;
  float s0, s1, s2;
  const int pix = sample_rgb_lane(f, cam, gx, gy, s0, s1, s2);
  float* dst = v.rgb + (int64_t)n * 3;
  const float o0 = dst[0], o1 = dst[1], o2 = dst[2];
  dst[0] = blend(s0, o0, a, b, sum);
  dst[1] = blend(s1, o1, a, b, sum);
  dst[2] = blend(s2, o2, a, b, sum);
  v.weight[n] = w0 + 1;  // clipfusion.py:715, :721
  count_label_lane<false>(v, f, n, pix, stats);
}

int make_kframe(const saf_frame* fr, KFrame* kf) {
  if (!fr) return fail(SAF_E_INVALID, "frame is NULL");
  if (fr->height <= 0 || fr->width <= 0 || fr->npy <= 0 || fr->npx <= 0)
    return fail(SAF_E_INVALID, "bad frame shape %dx%d map %dx%d", fr->height, fr->width, fr->npy, fr->npx);
  if ((int64_t)fr->height * fr->width >= (1ll << 30)) return fail(SAF_E_UNSUPPORTED, "image too large");
  if (!fr->depth || !fr->rgb || !fr->pose || !fr->K || !fr->feat_map) return fail(SAF_E_INVALID, "frame has a NULL buffer");
  kf->H = fr->height; kf->W = fr->width; kf->npy = fr->npy; kf->npx = fr->npx;
  kf->rgb_bilinear = fr->rgb_bilinear;
  kf->depth = fr->depth; kf->rgb = fr->rgb; kf->pose = fr->pose; kf->K = fr->K;
  kf->label_map = fr->label_map;
  return SAF_OK;
}


struct ScopedPair {
  saf_profiler* p;
  hipStream_t s;
  int idx;
  ScopedPair(saf_profiler* prof, int cls, int64_t frame_no, hipStream_t stream) : p(prof), s(stream), idx(-1) {
    if (p && p->used < p->capacity && frame_no % p->stride == 0) {
      idx = p->used++;
      p->pairs[idx].cls = cls;
      (void)hipEventRecord(p->pairs[idx].a, s);
    }
  }
  ~ScopedPair() {
    if (idx >= 0) (void)hipEventRecord(p->pairs[idx].b, s);
  }
};

}  // namespace

// the windowed path (saf_window.hip), called by saf_fuse_frames
size_t window_workspace_bytes(int64_t n_vox, int D, int P, bool bricks, int H = 0, int W = 0, bool labels = false);
int window_frames();  // frames per window of this call (SAF_WINDOW_FRAMES, or 64 with SAF_WIN_FRAMES=64)
bool window_ok(const KVol& kv, const saf_frame* frames, int32_t n_frames, size_t workspace_bytes);
// Streams / events for running the classification of window w + 1 beside the row kernel of window w (may be NULL:
// everything on the caller's stream).
struct WinOverlap {
  hipStream_t aux;
  hipEvent_t fork, join, cls_done[2], fuse_done[2];
  hipEvent_t tiles;  // (may be NULL) the later windows' depth tiles, computed on the caller's stream ahead of their classification
};
// saf_fuse_frames_slabs: every frame into x-planes [x0[k], x0[k] + nx[k]) for k = 0 .. n - 1 in turn; done[k] (may be NULL):
// a hipEvent_t recorded on the caller's stream behind slab k's last row kernel.
struct WinSlabs {
  int n;
  const int32_t *x0, *nx;
  void* const* done;
};
// A streaming session (saf_fuse_session_*; saf_window.hip): the open window's state between two pushes.
struct WinStream {
  bool open = false;        // a window is open: `filled` of its frames are classified, its row kernel is not launched
  int filled = 0;
  int n_windows = 0;        // windows closed so far (parity of headers / mask planes, tile slots, "first window")
  bool have_shape = false;  // kv / kf0 are those of the session's frames
  KVol kv;
  KFrame kf0;
  long long pushed = 0;     // frames classified so far in this session
  long long prepared = 0;   // frames whose depth tiles are computed (stream_prepare runs ahead of stream_push: >= pushed, or behind it when unused)
};
int stream_prepare(const KVol& kv, const saf_frame* frames, int32_t n_frames, void* workspace, size_t workspace_bytes, hipStream_t ts,
                   WinStream* st);
bool stream_ok(const KVol& kv, const saf_frame* frames, int32_t n_frames, size_t workspace_bytes);
int stream_push(const KVol& kv, const saf_frame* frames, int32_t n_frames, void* workspace, size_t workspace_bytes, uint64_t* stats,
                hipStream_t s, hipEvent_t ready, hipStream_t tile_stream, const WinOverlap* ov, WinStream* st);
int stream_close(void* workspace, size_t workspace_bytes, uint64_t* stats, hipStream_t s, const WinOverlap* ov, WinStream* st, bool preopen);
// recycled: the volume's feature rows were not cleared when its scalars were (saf_fuse_frames_recycled) -- the rows of voxels
// whose weight is still 0 when the call is over are zeroed by it, beside the last window's row kernel where the schedule allows.
int fuse_many_windowed(const KVol& kv, const saf_frame* frames, int32_t n_frames, void* workspace, size_t workspace_bytes,
                       uint64_t* stats, saf_profiler* prof, hipStream_t s, const WinOverlap* ov, const WinSlabs* slabs = nullptr,
                       bool recycled = false);
// saf_misc.hip (clear_rows): zero the feature rows of the voxels n in [0, kv.N) with weight[n] == 0 and -- masks may be NULL -- no
// bit set in masks[p * mask_plane + n] for p < n_planes (the hit masks of a window whose row kernel may be running: it writes exactly
// the rows with a bit set, this kernel only the others).
int clear_rows(void* feat, const int* weight, int64_t first, int64_t n_rows, int esz, int row_bytes, const uint32_t* masks,
               size_t mask_plane, int n_planes, hipStream_t s);
inline int launch_clear_unwritten(const KVol& kv, const uint32_t* masks, size_t mask_plane, int n_planes, hipStream_t s) {
  const int esz = kv.bf16 ? 2 : 4;
  return clear_rows(kv.feat, kv.weight, 0, (int64_t)kv.N, esz, kv.D * esz, masks, mask_plane, n_planes, s);
}
KVol slab_kvol(const KVol& kv, int x0, int nx);

}  // namespace saf
