// saf_fuse.hip -- projective voxel fusion of one RGB-D frame on gfx950 (MI355X).
//
// Replaces ClipFusion.integrate / ClipSeemFusion.integrate after the backbone calls
// (reference clipfusion.py:647-721, clip_seem_fusion.py:697-822).  Three launches per frame:
//
//   prep   : re-lays the frame's feature map [D,npy,npx] -> [npy*npx][D] (so a voxel's D-vector
//            of one tap is contiguous) and zeroes the compact-list counters.
//   sweep  : one pass over ALL voxels (a2-a4): project the voxel centre, nearest-pixel depth
//            test, TSDF running mean for tsdf_valid voxels, and wave-ballot + LDS compaction of
//            the `valid` voxels of a 4096-voxel chunk into one of 16 compact lists (one global
//            atomic per block).  Compute-bound; no volume row is touched here.
//   fuse   : (a5-a7) workgroups stage the re-laid feature map in LDS (71,680 B for 512x5x7) and
//            walk the compact lists; a group of G lanes owns one voxel row: 4-tap bilinear
//            sample from LDS, running-mean read-modify-write of the D-row with 16-byte accesses
//            per lane, several rows in flight per wave; rgb / weight / label counter ride along.
//            HBM-bound: algorithmic bytes = Nv * (2*D*4 + 32 [+8]) per frame.
//
// All arithmetic that selects voxels is shared with the oracle's restatement via saf_common.h.
#include <math.h>
#include <string.h>

#include "saf_common.h"
#include "saf_host.h"

#pragma clang fp contract(off)

namespace saf {

char* err_buf() {
  static thread_local char buf[kErrLen] = "";
  return buf;
}

namespace {

// ------------------------------------------------------------------------------------------
// kernel-side descriptors (POD, passed by value)
// ------------------------------------------------------------------------------------------
struct KVol {
  int nx, ny, nz, D, n_classes, accum;
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

struct WsLayout {
  size_t counts_off, map_off, lists_off, total;
  uint32_t n_blocks, list_cap;
};

WsLayout ws_layout(int64_t n_vox, int D, int P) {
  WsLayout w;
  w.n_blocks = (uint32_t)((n_vox + kSweepChunk - 1) / kSweepChunk);
  uint32_t per_list = (w.n_blocks + kNumLists - 1) / kNumLists;
  w.list_cap = per_list * kSweepChunk;
  w.counts_off = 0;
  w.map_off = 256;
  size_t map_bytes = ((size_t)D * (P + 1) * sizeof(float) + 255) & ~(size_t)255;
  w.lists_off = w.map_off + map_bytes;
  w.total = w.lists_off + (size_t)kNumLists * w.list_cap * sizeof(uint32_t);
  return w;
}

// ------------------------------------------------------------------------------------------
// prep: feature map [Dm>=D][P] -> [P][D]; zero list counters
// ------------------------------------------------------------------------------------------
__global__ void prep_kernel(const float* __restrict__ feat_map, float* __restrict__ map_t, int D, int P,
                            unsigned long long* __restrict__ counts) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kNumLists) counts[i] = 0ull;
  if (i < D * P) {
    int p = i / D, c = i - p * D;
    map_t[i] = feat_map[(size_t)c * P + p];
  } else if (i < D * (P + 1)) {
    map_t[i] = 0.0f;  // row P: the zero padding every out-of-map tap reads
  }
}

// ------------------------------------------------------------------------------------------
// sweep: classify every voxel, TSDF running mean, compact the valid ones
// ------------------------------------------------------------------------------------------
constexpr int kSweepIlp = 4;  // voxels in flight per thread: hides table / depth / TSDF latency

__global__ __launch_bounds__(kSweepThreads) void sweep_kernel(KVol v, KFrame f,
                                                               unsigned long long* __restrict__ counts,
                                                               uint32_t* __restrict__ lists, uint32_t list_cap) {
  __shared__ uint32_t s_buf[kSweepChunk];
  __shared__ uint32_t s_count, s_base, s_nt;
  const int tid = threadIdx.x, lane = tid & 63;
  const Cam cam = load_cam(f.pose, f.K, f.W, f.H);
  if (tid == 0) {
    s_count = 0;
    s_nt = 0;
  }
  __syncthreads();
  const uint32_t chunk_base = blockIdx.x * (uint32_t)kSweepChunk;
  uint32_t nt_local = 0;
  for (int k0 = 0; k0 < kSweepPerThread; k0 += kSweepIlp) {
    uint32_t n[kSweepIlp];
    Proj p[kSweepIlp];
    int pix[kSweepIlp];
    bool in_view[kSweepIlp], valid[kSweepIlp], tv[kSweepIlp];
    float depth[kSweepIlp], sdf[kSweepIlp], told[kSweepIlp];
    int w0[kSweepIlp];
    // phase 1: voxel centre -> image (clipfusion.py:647-659); all table loads issued together
    float xw[kSweepIlp], yw[kSweepIlp], zw[kSweepIlp];
#pragma unroll
    for (int j = 0; j < kSweepIlp; ++j) {
      n[j] = chunk_base + (uint32_t)(k0 + j) * kSweepThreads + tid;
      const uint32_t nc = n[j] < v.N ? n[j] : v.N - 1;
      int ix, iy, iz;
      voxel_coords(v, nc, ix, iy, iz);
      xw[j] = v.ax[ix];
      yw[j] = v.ay[iy];
      zw[j] = v.az[iz];
    }
#pragma unroll
    for (int j = 0; j < kSweepIlp; ++j) {
      p[j] = project(cam, xw[j], yw[j], zw[j]);
      // _valid = (grid.abs() <= 1).all(dim=1) & (z > 0)            clipfusion.py:673
      in_view[j] = (n[j] < v.N) && (fabsf(p[j].gx) <= 1.0f) && (fabsf(p[j].gy) <= 1.0f) && (p[j].z > 0.0f);
      pix[j] = in_view[j] ? nearest_index(p[j].gx, p[j].gy, cam, f.W) : -1;
    }
    // phase 2: nearest-pixel depth (zeros padding), all gathers in flight together
#pragma unroll
    for (int j = 0; j < kSweepIlp; ++j) depth[j] = pix[j] >= 0 ? f.depth[pix[j]] : 0.0f;
    // phase 3: classify, issue the TSDF loads
#pragma unroll
    for (int j = 0; j < kSweepIlp; ++j) {
      sdf[j] = (depth[j] - p[j].z) / v.trunc;        // clipfusion.py:669
      valid[j] = in_view[j] && fabsf(sdf[j]) <= 1.0f;  // :678
      tv[j] = in_view[j] && sdf[j] > -1.0f;           // tsdf_valid, :679
      if (tv[j]) {
        w0[j] = v.tsdf_w[n[j]];
        told[j] = v.tsdf[n[j]];
      }
    }
    // phase 4: running mean of the clamped sdf, clipfusion.py:681-695 with B = 1
#pragma unroll
    for (int j = 0; j < kSweepIlp; ++j) {
      if (tv[j]) {
        const float t = sdf[j] > 1.0f ? 1.0f : sdf[j];
        const int w1 = w0[j] + 1;
        float nt;
        if (v.accum == SAF_SUM) {
          nt = told[j] + t;
        } else {
          const float a = (float)w1;
          const float b = (float)w0[j] / (float)w1;
          nt = t / a + told[j] * b;
        }
        v.tsdf[n[j]] = nt;
        v.tsdf_w[n[j]] = w1;
        ++nt_local;
      }
    }
    // phase 5: wave64 ballot + prefix popcount -> slots in the block's LDS buffer
#pragma unroll
    for (int j = 0; j < kSweepIlp; ++j) {
      const unsigned long long m = __ballot(valid[j]);
      if (m) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_count, (uint32_t)__popcll(m));
        base = __shfl(base, 0);
        if (valid[j]) s_buf[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = n[j];
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nt_local += __shfl_down(nt_local, o);
  if (lane == 0 && nt_local) atomicAdd(&s_nt, nt_local);
  __syncthreads();
  const uint32_t total = s_count;
  // Lists are chosen by a hash of the block index: a plain modulo would alias with the grid's
  // y-bands (ny*nz/4096 blocks per x-slab) and concentrate the shell in a few lists.
  const uint32_t list = (blockIdx.x ^ (blockIdx.x >> 4) ^ (blockIdx.x >> 9)) % kNumLists;
  // ONE global atomic per block: low word = valid entries appended to this list (returns the
  // block's base slot), high word = tsdf-valid voxels (statistics, summed by the fuse kernel).
  if (tid == 0 && (total | s_nt))
    s_base = (uint32_t)atomicAdd(&counts[list], ((unsigned long long)s_nt << 32) | total);
  if (total) {
    __syncthreads();
    uint32_t* dst = lists + (size_t)list * list_cap + s_base;
    for (uint32_t i = tid; i < total; i += kSweepThreads) dst[i] = s_buf[i];
  }
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
__device__ __forceinline__ float4 vzero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float vzero(float) { return 0.f; }

struct Taps {
  int o_nw, o_ne, o_sw, o_se;  // offsets (in vector units) of the four taps in the [P+1][DV] map
};
// Taps outside the map read row P, which holds zeros: grid_sample's padding_mode="zeros" with
// unconditional loads (no branches in the row loop).
__device__ __forceinline__ Taps tap_offsets(const Bilin& b, int npx, int npy, int DV) {
  const bool x0 = b.x0 >= 0 && b.x0 < npx, x1 = b.x0 + 1 >= 0 && b.x0 + 1 < npx;
  const bool y0 = b.y0 >= 0 && b.y0 < npy, y1 = b.y0 + 1 >= 0 && b.y0 + 1 < npy;
  const int zero_row = npx * npy;
  Taps t;
  t.o_nw = ((x0 && y0) ? b.y0 * npx + b.x0 : zero_row) * DV;
  t.o_ne = ((x1 && y0) ? b.y0 * npx + b.x0 + 1 : zero_row) * DV;
  t.o_sw = ((x0 && y1) ? (b.y0 + 1) * npx + b.x0 : zero_row) * DV;
  t.o_se = ((x1 && y1) ? (b.y0 + 1) * npx + b.x0 + 1 : zero_row) * DV;
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

// Once per frame (block 0, thread 0): fold the per-list counters into the caller's statistics.
__device__ __forceinline__ void add_frame_stats(const unsigned long long* __restrict__ counts,
                                                unsigned long long* __restrict__ stats) {
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long nv = 0, nt = 0;
    for (int l = 0; l < kNumLists; ++l) {
      const unsigned long long c = counts[l];
      nv += c & 0xffffffffull;
      nt += c >> 32;
    }
    atomicAdd(&stats[0], nv);
    atomicAdd(&stats[1], nt);
    atomicAdd(&stats[2], 1ull);
  }
}

// rgb / weight / label side of one valid voxel handled entirely by ONE lane (the lane-parallel
// part of fuse_rows_kernel): the three channel loads are issued together, then blended and stored.
__device__ __forceinline__ void fuse_scalars_lane(const KVol& v, const KFrame& f, const Cam& cam, uint32_t n,
                                                  float gx, float gy, int w0, float a, float b,
                                                  unsigned long long* stats) {
  const bool sum = v.accum == SAF_SUM;
  const int pix = nearest_index(gx, gy, cam, f.W);
  float s0, s1, s2;
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
  float* dst = v.rgb + (int64_t)n * 3;
  const float o0 = dst[0], o1 = dst[1], o2 = dst[2];
  dst[0] = blend(s0, o0, a, b, sum);
  dst[1] = blend(s1, o1, a, b, sum);
  dst[2] = blend(s2, o2, a, b, sum);
  v.weight[n] = w0 + 1;  // clipfusion.py:715, :721
  if (v.labels && f.label_map) {
    // labels = grid_sample(pano_seg.float(), nearest); one_hot(labels.long())  clip_seem_fusion.py:786-822
    const float lraw = f.label_map[pix >= 0 ? pix : 0];
    const float lf = pix >= 0 ? lraw : 0.f;
    const long long l = (long long)lf;
    if (l >= 0 && l < v.n_classes) {
      int* c = v.labels + (int64_t)n * v.n_classes + l;
      *c = *c + 1;
    } else if (stats) {
      atomicAdd(&stats[3], 1ull);
    }
  }
}

// VEC: floats per lane access (4 when D % 4 == 0).  CPL: vector chunks per lane (compile-time,
// 0 = runtime loop).  U: voxel rows in flight per lane group.  LDS_MAP: feature map staged in LDS.
template <int VEC, int CPL, int U, bool LDS_MAP>
__global__ __launch_bounds__(kFuseThreads) void fuse_kernel(KVol v, KFrame f,
                                                             const unsigned long long* __restrict__ counts,
                                                             const uint32_t* __restrict__ lists, uint32_t list_cap,
                                                             const float* __restrict__ map_t, int g_log2,
                                                             unsigned long long* __restrict__ stats) {
  using V = typename VecT<VEC>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int tid = threadIdx.x;
  const int DV = v.D / VEC;  // vector chunks per row
  const int P = f.npy * f.npx;
  const V* map;
  if (LDS_MAP) {
    V* s_map = reinterpret_cast<V*>(s_raw);
    const V* src = reinterpret_cast<const V*>(map_t);
    for (int i = tid; i < (P + 1) * DV; i += kFuseThreads) s_map[i] = src[i];
    __syncthreads();
    map = s_map;
  } else {
    map = reinterpret_cast<const V*>(map_t);
  }
  const Cam cam = load_cam(f.pose, f.K, f.W, f.H);
  const float half_px = (float)f.npx / 2.0f, half_py = (float)f.npy / 2.0f;
  const bool sum = v.accum == SAF_SUM;

  const int G = 1 << g_log2;
  const int lane = tid & 63, wave = tid >> 6;
  const int slot = lane >> g_log2, gl = lane & (G - 1);
  const int epw = 64 >> g_log2;  // entries per wave per step
  const uint32_t list = blockIdx.x % kNumLists;
  const uint32_t wg_in_list = blockIdx.x / kNumLists, wgs_per_list = gridDim.x / kNumLists;
  const uint32_t count = (uint32_t)counts[list];
  const uint32_t* lst = lists + (size_t)list * list_cap;
  add_frame_stats(counts, stats);
  const uint32_t sid = (wg_in_list * (kFuseThreads / 64) + wave) * epw + slot;
  const uint32_t stride = wgs_per_list * (kFuseThreads / 64) * epw;
  V* feat = reinterpret_cast<V*>(v.feat);
  constexpr int C = CPL > 0 ? CPL : 1;

  for (uint32_t e0 = sid; e0 < count; e0 += stride * U) {
    uint32_t n[U];
    bool act[U];
    float gx[U], gy[U], a[U], b[U];
    int w0[U];
    Bilin bf[U];
    Taps tp[U];
    V old[U][C];
    // phase 1: entries
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const uint32_t e = e0 + (uint32_t)j * stride;
      act[j] = e < count;
      n[j] = act[j] ? lst[e] : 0u;
    }
    // phase 2: issue the row loads of all U rows
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (act[j]) {
        w0[j] = v.weight[n[j]];
        if (CPL > 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const int ch = gl + c * G;
            if (ch < DV) old[j][c] = feat[(int64_t)n[j] * DV + ch];
          }
        }
      }
    }
    // phase 3: projection + taps (overlaps the loads)
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (act[j]) {
        int ix, iy, iz;
        voxel_coords(v, n[j], ix, iy, iz);
        const Proj p = project(cam, v.ax[ix], v.ay[iy], v.az[iz]);
        gx[j] = p.gx;
        gy[j] = p.gy;
        bf[j] = bilinear_setup(p.gx, p.gy, half_px, half_py);
        tp[j] = tap_offsets(bf[j], f.npx, f.npy, DV);
      }
    }
    // phase 4: blend + store
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (act[j]) {
        // a = 1 / new_weight ; b = weight * a                    clipfusion.py:716-717
        a[j] = 1.0f / (float)(w0[j] + 1);
        b[j] = (float)w0[j] * a[j];
        const Taps t = tp[j];
        if (CPL > 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const int ch = gl + c * G;
            if (ch < DV) {
              const V s = lerp_taps(map[t.o_nw + ch], map[t.o_ne + ch], map[t.o_sw + ch], map[t.o_se + ch], bf[j]);
              feat[(int64_t)n[j] * DV + ch] = blend(s, old[j][c], a[j], b[j], sum);
            }
          }
        } else {
          for (int ch = gl; ch < DV; ch += G) {
            const V s = lerp_taps(map[t.o_nw + ch], map[t.o_ne + ch], map[t.o_sw + ch], map[t.o_se + ch], bf[j]);
            V* dst = feat + (int64_t)n[j] * DV + ch;
            *dst = blend(s, *dst, a[j], b[j], sum);
          }
        }
        fuse_scalars(v, f, cam, n[j], gx[j], gy[j], w0[j], a[j], b[j], gl, G, stats);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// fuse, main path (D % 4 == 0, <= 4 vector chunks per lane): each wave owns a contiguous slice
// of one compact list and walks it in batches of 64 entries.
//   lane-parallel part : lane l <-> entry l of the batch: projection, bilinear setup, 1/w
//                        weights, and the whole scalar side (rgb / weight / label counter) as
//                        64-wide gathers -- done once per voxel instead of once per lane.
//   row loop           : groups of G lanes RMW one D-row each; the owning lane's (n, a, b, tap
//                        cell, tap fractions) are broadcast with readlane (G = 64: scalar
//                        registers, scalar row base) or ds_bpermute; R rows are kept in flight
//                        per group by a rotating register pipeline (load row i+R right after
//                        storing row i), so HBM latency is covered by work, not by occupancy.
// ------------------------------------------------------------------------------------------
template <bool G64>
__device__ __forceinline__ int bcast_i(int x, int src) {
  return G64 ? __builtin_amdgcn_readlane(x, src) : __shfl(x, src);
}
template <bool G64>
__device__ __forceinline__ float bcast_f(float x, int src) {
  return __builtin_bit_cast(float, bcast_i<G64>(__builtin_bit_cast(int, x), src));
}

struct RowCtx {
  uint32_t n;
  float a, b;
  Bilin w;
  Taps t;
};

template <bool G64>
__device__ __forceinline__ RowCtx fetch_row(int src, uint32_t n_l, float a_l, float b_l, int xy_l, float wx_l,
                                            float wy_l, int npx, int npy, int DV) {
  RowCtx c;
  c.n = (uint32_t)bcast_i<G64>((int)n_l, src);
  c.a = bcast_f<G64>(a_l, src);
  c.b = bcast_f<G64>(b_l, src);
  const int xy = bcast_i<G64>(xy_l, src);
  const float wx = bcast_f<G64>(wx_l, src), wy = bcast_f<G64>(wy_l, src);
  // identical operations to bilinear_setup(): the fractions travel, the products are redone
  const float ex = 1.0f - wx, sy = 1.0f - wy;
  c.w.x0 = (xy & 0xffff) - 1;
  c.w.y0 = (xy >> 16) - 1;
  c.w.nw = sy * ex;
  c.w.ne = sy * wx;
  c.w.sw = wy * ex;
  c.w.se = wy * wx;
  c.t = tap_offsets(c.w, npx, npy, DV);
  return c;
}

// The row loop is written without per-lane guards so that it compiles to straight-line code with
// counted waits: lanes whose chunk index would fall past the row re-do the last chunk (identical
// value to the same address), and lane groups past the end of a batch re-do the batch's last
// row inside the SAME wave instruction as its owner (same loads, same stores).
template <int CPL, int R, bool G64, bool LDS_MAP>
__global__ __launch_bounds__(kFuseThreads) void fuse_rows_kernel(KVol v, KFrame f,
                                                                  const unsigned long long* __restrict__ counts,
                                                                  const uint32_t* __restrict__ lists,
                                                                  uint32_t list_cap, const float* __restrict__ map_t,
                                                                  int g_log2, unsigned long long* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int tid = threadIdx.x;
  const int DV = v.D >> 2;
  const int P = f.npy * f.npx;
  const float4* map;
  if (LDS_MAP) {
    float4* s_map = reinterpret_cast<float4*>(s_raw);
    const float4* src = reinterpret_cast<const float4*>(map_t);
    for (int i = tid; i < (P + 1) * DV; i += kFuseThreads) s_map[i] = src[i];
    __syncthreads();
    map = s_map;
  } else {
    map = reinterpret_cast<const float4*>(map_t);
  }
  const Cam cam = load_cam(f.pose, f.K, f.W, f.H);
  const float half_px = (float)f.npx / 2.0f, half_py = (float)f.npy / 2.0f;
  const bool sum = v.accum == SAF_SUM;
  add_frame_stats(counts, stats);

  const int G = G64 ? 64 : (1 << g_log2);
  const int lane = tid & 63, wave = tid >> 6;
  const int slot = G64 ? 0 : (lane >> g_log2), gl = lane & (G - 1);
  const int epw = G64 ? 1 : (64 >> g_log2);  // rows per wave step
  int chs[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) chs[c] = min(gl + c * G, DV - 1);
  const uint32_t list = blockIdx.x % kNumLists;
  const uint32_t count = (uint32_t)counts[list];
  const uint32_t* lst = lists + (size_t)list * list_cap;
  // static, balanced partition of the list over the waves that serve it
  const uint32_t waves_per_list = (gridDim.x / kNumLists) * (kFuseThreads / 64);
  const uint32_t wave_in_list = (blockIdx.x / kNumLists) * (kFuseThreads / 64) + wave;
  const uint32_t q = (count + waves_per_list - 1) / waves_per_list;
  const uint32_t begin = wave_in_list * q;
  const uint32_t end = begin + q < count ? begin + q : count;
  float4* feat = reinterpret_cast<float4*>(v.feat);

  for (uint32_t b0 = begin; b0 < end; b0 += 64) {
    const int nb = (int)(end - b0 < 64u ? end - b0 : 64u);  // wave-uniform, >= 1
    // ---------------- lane-parallel part: lane l <-> entry b0 + l ----------------
    uint32_t n_l = 0;
    float a_l = 0.f, b_l = 0.f, wx_l = 0.f, wy_l = 0.f;
    int xy_l = 0;
    if (lane < nb) {
      n_l = lst[b0 + lane];
      int ix, iy, iz;
      voxel_coords(v, n_l, ix, iy, iz);
      const Proj p = project(cam, v.ax[ix], v.ay[iy], v.az[iz]);
      const int w0 = v.weight[n_l];
      // a = 1 / new_weight ; b = weight * a                        clipfusion.py:716-717
      a_l = 1.0f / (float)(w0 + 1);
      b_l = (float)w0 * a_l;
      const float x = unnormalize(p.gx, half_px), y = unnormalize(p.gy, half_py);
      const float xw = __builtin_floorf(x), yn = __builtin_floorf(y);
      wx_l = x - xw;
      wy_l = y - yn;
      // tap cell, clamped to [-1, size] (everything outside is zero padding anyway)
      const int x0 = (int)fminf(fmaxf(xw, -1.0f), (float)f.npx), y0 = (int)fminf(fmaxf(yn, -1.0f), (float)f.npy);
      xy_l = (x0 + 1) | ((y0 + 1) << 16);
      fuse_scalars_lane(v, f, cam, n_l, p.gx, p.gy, w0, a_l, b_l, stats);
    }
    // ---------------- row loop: groups of G lanes, R rows in flight per group ----------------
    const int steps = (nb + epw - 1) / epw;
    const int last = nb - 1;
#define SAF_FETCH(i) fetch_row<G64>(min((i) * epw + slot, last), n_l, a_l, b_l, xy_l, wx_l, wy_l, f.npx, f.npy, DV)
#define SAF_LOAD(ctx_, old_)                                                           \
  _Pragma("unroll") for (int c = 0; c < CPL; ++c) old_[c] = ld_stream(&feat[(int64_t)(ctx_).n * DV + chs[c]]);
#define SAF_STORE(ctx_, old_)                                                                                      \
  _Pragma("unroll") for (int c = 0; c < CPL; ++c) {                                                                \
    const float4 sv = lerp_taps(map[(ctx_).t.o_nw + chs[c]], map[(ctx_).t.o_ne + chs[c]],                          \
                                map[(ctx_).t.o_sw + chs[c]], map[(ctx_).t.o_se + chs[c]], (ctx_).w);               \
    st_stream(&feat[(int64_t)(ctx_).n * DV + chs[c]], blend(sv, old_[c], (ctx_).a, (ctx_).b, sum));               \
  }
    int i0 = 0;
    if (steps >= R) {
      RowCtx ctx[R];
      float4 old[R][CPL];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        ctx[r] = SAF_FETCH(r);
        SAF_LOAD(ctx[r], old[r]);
      }
      // steady state: store row i, immediately refill its slot with row i + R
      for (; i0 + 2 * R <= steps; i0 += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          SAF_STORE(ctx[r], old[r]);
          ctx[r] = SAF_FETCH(i0 + R + r);
          SAF_LOAD(ctx[r], old[r]);
        }
      }
      // drain
#pragma unroll
      for (int r = 0; r < R; ++r) {
        SAF_STORE(ctx[r], old[r]);
      }
      i0 += R;
    }
    for (; i0 < steps; ++i0) {  // fewer than R rows left
      const RowCtx c1 = SAF_FETCH(i0);
      float4 o1[CPL];
      SAF_LOAD(c1, o1);
      SAF_STORE(c1, o1);
    }
#undef SAF_FETCH
#undef SAF_LOAD
#undef SAF_STORE
  }
}

using FuseFn = void (*)(KVol, KFrame, const unsigned long long*, const uint32_t*, uint32_t, const float*, int,
                        unsigned long long*);

template <int VEC, int CPL, int U>
FuseFn pick_lds(bool lds) {
  return lds ? fuse_kernel<VEC, CPL, U, true> : fuse_kernel<VEC, CPL, U, false>;
}

template <int CPL, int R>
FuseFn pick_rows(bool g64, bool lds) {
  if (g64) return lds ? fuse_rows_kernel<CPL, R, true, true> : fuse_rows_kernel<CPL, R, true, false>;
  return lds ? fuse_rows_kernel<CPL, R, false, true> : fuse_rows_kernel<CPL, R, false, false>;
}

int launch_fuse(const KVol& kv, const KFrame& kf, const WsLayout& w, unsigned char* ws, unsigned long long* stats,
                hipStream_t s) {
  const int D = kv.D, P = kf.npy * kf.npx;
  const int VEC = (D % 4 == 0) ? 4 : 1;
  const int DV = D / VEC;
  int g_log2 = 0;
  while ((1 << g_log2) < DV && g_log2 < 6) ++g_log2;
  const int G = 1 << g_log2;
  const int cpl = (DV + G - 1) / G;
  const size_t map_bytes = (size_t)D * (P + 1) * sizeof(float);
  const bool lds = map_bytes <= 144 * 1024;
  FuseFn fn;
  if (VEC == 4 && cpl >= 1 && cpl <= 4) {
    const bool g64 = G == 64;
    switch (cpl) {
      case 1: fn = pick_rows<1, 4>(g64, lds); break;
      case 2: fn = pick_rows<2, 4>(g64, lds); break;
      case 3: fn = pick_rows<3, 2>(g64, lds); break;
      default: fn = pick_rows<4, 2>(g64, lds); break;
    }
  } else if (VEC == 4) {
    fn = pick_lds<4, 0, 1>(lds);
  } else {
    fn = (cpl == 1) ? pick_lds<1, 1, 4>(lds) : pick_lds<1, 0, 1>(lds);
  }
  const size_t shmem = lds ? map_bytes : 0;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute(LDS=%zu): %s", shmem, hipGetErrorString(e));
  }
  // workgroups resident per CU: limited by the LDS image (160 KiB / CU) and 2048 threads / CU
  int per_cu = shmem ? (int)((160 * 1024) / (shmem + 64)) : 4;
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  int grid = device_cus() * per_cu;
  grid = ((grid + kNumLists - 1) / kNumLists) * kNumLists;
  const unsigned long long* counts = reinterpret_cast<const unsigned long long*>(ws + w.counts_off);
  const uint32_t* lists = reinterpret_cast<const uint32_t*>(ws + w.lists_off);
  const float* map_t = reinterpret_cast<const float*>(ws + w.map_off);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kFuseThreads), shmem, s, kv, kf, counts, lists, w.list_cap, map_t, g_log2,
                     stats);
  return check_launch("fuse_kernel");
}

int make_kvol(const saf_volume* vol, KVol* kv) {
  if (!vol) return fail(SAF_E_INVALID, "volume is NULL");
  if (vol->nx <= 0 || vol->ny <= 0 || vol->nz <= 0 || vol->feat_dim <= 0)
    return fail(SAF_E_INVALID, "bad volume shape %dx%dx%d D=%d", vol->nx, vol->ny, vol->nz, vol->feat_dim);
  const int64_t N = n_voxels(vol);
  if (N >= (1ll << 31)) return fail(SAF_E_UNSUPPORTED, "volumes of 2^31 voxels or more are not supported");
  if (vol->feat_dtype != SAF_F32) return fail(SAF_E_UNSUPPORTED, "feat_dtype %d: only SAF_F32 so far", vol->feat_dtype);
  if (vol->accum_mode != SAF_RUNNING_MEAN && vol->accum_mode != SAF_SUM)
    return fail(SAF_E_INVALID, "bad accum_mode %d", vol->accum_mode);
  if (!vol->axis_x || !vol->axis_y || !vol->axis_z || !vol->tsdf || !vol->tsdf_weight || !vol->weight || !vol->rgb ||
      !vol->clip_feat)
    return fail(SAF_E_INVALID, "volume has a NULL buffer");
  if (!(vol->trunc > 0.0f)) return fail(SAF_E_INVALID, "trunc must be positive");
  if (vol->feat_dim % 4 == 0 && ((uintptr_t)vol->clip_feat & 15)) return fail(SAF_E_INVALID, "clip_feat must be 16-byte aligned");
  if (vol->n_classes < 0 || (vol->n_classes > 0 && !vol->labels_one_hot && false))
    return fail(SAF_E_INVALID, "bad n_classes");
  kv->nx = vol->nx; kv->ny = vol->ny; kv->nz = vol->nz;
  kv->D = vol->feat_dim;
  kv->n_classes = vol->labels_one_hot ? vol->n_classes : 0;
  kv->accum = vol->accum_mode;
  kv->N = (uint32_t)N;
  kv->trunc = vol->trunc;
  kv->ax = vol->axis_x; kv->ay = vol->axis_y; kv->az = vol->axis_z;
  kv->tsdf = vol->tsdf; kv->tsdf_w = vol->tsdf_weight; kv->weight = vol->weight;
  kv->rgb = vol->rgb;
  kv->feat = static_cast<float*>(vol->clip_feat);
  kv->labels = kv->n_classes ? vol->labels_one_hot : nullptr;
  kv->div_nz = make_fastdiv((uint32_t)vol->nz);
  kv->div_ny = make_fastdiv((uint32_t)vol->ny);
  return SAF_OK;
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

}  // namespace
}  // namespace saf

// Pool of event pairs; opaque to callers (include/saf.h).
struct saf_profiler {
  struct Pair {
    hipEvent_t a, b;
    int cls;
  };
  Pair* pairs;
  int capacity, used;
};

namespace saf {
namespace {

struct ScopedPair {
  saf_profiler* p;
  hipStream_t s;
  int idx;
  ScopedPair(saf_profiler* prof, int cls, hipStream_t stream) : p(prof), s(stream), idx(-1) {
    if (p && p->used < p->capacity) {
      idx = p->used++;
      p->pairs[idx].cls = cls;
      (void)hipEventRecord(p->pairs[idx].a, s);
    }
  }
  ~ScopedPair() {
    if (idx >= 0) (void)hipEventRecord(p->pairs[idx].b, s);
  }
};

int fuse_one(const KVol& kv, const saf_frame* frame, void* workspace, size_t workspace_bytes, uint64_t* stats,
             saf_profiler* prof, hipStream_t s) {
  KFrame kf;
  int rc = make_kframe(frame, &kf);
  if (rc) return rc;
  const int P = kf.npy * kf.npx;
  const WsLayout w = ws_layout(kv.N, kv.D, P);
  if (!workspace || ((uintptr_t)workspace & 255)) return fail(SAF_E_INVALID, "workspace must be 256-byte aligned");
  if (workspace_bytes < w.total)
    return fail(SAF_E_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, w.total);
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  unsigned long long* counts = reinterpret_cast<unsigned long long*>(ws + w.counts_off);
  uint32_t* lists = reinterpret_cast<uint32_t*>(ws + w.lists_off);
  float* map_t = reinterpret_cast<float*>(ws + w.map_off);
  unsigned long long* st = reinterpret_cast<unsigned long long*>(stats);

  const int prep_items = kv.D * (P + 1) > kNumLists ? kv.D * (P + 1) : kNumLists;
  {
    ScopedPair t(prof, 0, s);
    hipLaunchKernelGGL(prep_kernel, dim3((prep_items + 255) / 256), dim3(256), 0, s, frame->feat_map, map_t, kv.D, P,
                       counts);
  }
  if ((rc = check_launch("prep_kernel"))) return rc;
  {
    ScopedPair t(prof, 1, s);
    hipLaunchKernelGGL(sweep_kernel, dim3(w.n_blocks), dim3(kSweepThreads), 0, s, kv, kf, counts, lists, w.list_cap);
  }
  if ((rc = check_launch("sweep_kernel"))) return rc;
  ScopedPair t(prof, 2, s);
  return launch_fuse(kv, kf, w, ws, st, s);
}

}  // namespace
}  // namespace saf

using namespace saf;

extern "C" {

const char* saf_last_error(void) { return err_buf(); }
int saf_abi_version(void) { return SAF_ABI_VERSION; }

size_t saf_fuse_workspace_bytes(int64_t n_vox, int32_t feat_dim, int32_t npy, int32_t npx) {
  if (n_vox <= 0 || feat_dim <= 0 || npy <= 0 || npx <= 0) return 0;
  return ws_layout(n_vox, feat_dim, npy * npx).total;
}

int saf_fuse_frame(const saf_volume* vol, const saf_frame* frame, void* workspace, size_t workspace_bytes,
                   uint64_t* stats, void* stream) {
  KVol kv;
  int rc = make_kvol(vol, &kv);
  if (rc) return rc;
  return fuse_one(kv, frame, workspace, workspace_bytes, stats, nullptr, static_cast<hipStream_t>(stream));
}

int saf_fuse_frames_profiled(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, void* workspace,
                             size_t workspace_bytes, uint64_t* stats, saf_profiler* profiler, void* stream) {
  KVol kv;
  int rc = make_kvol(vol, &kv);
  if (rc) return rc;
  if (n_frames < 0 || (n_frames > 0 && !frames)) return fail(SAF_E_INVALID, "bad frame array");
  for (int32_t i = 0; i < n_frames; ++i) {
    rc = fuse_one(kv, &frames[i], workspace, workspace_bytes, stats, profiler, static_cast<hipStream_t>(stream));
    if (rc) return rc;
  }
  return SAF_OK;
}

int saf_fuse_frames(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, void* workspace,
                    size_t workspace_bytes, uint64_t* stats, void* stream) {
  return saf_fuse_frames_profiled(vol, frames, n_frames, workspace, workspace_bytes, stats, nullptr, stream);
}

saf_profiler* saf_profiler_create(int32_t capacity_pairs) {
  if (capacity_pairs <= 0) return nullptr;
  saf_profiler* p = new saf_profiler;
  p->pairs = new saf_profiler::Pair[capacity_pairs];
  p->capacity = 0;
  p->used = 0;
  for (int i = 0; i < capacity_pairs; ++i) {
    if (hipEventCreate(&p->pairs[i].a) != hipSuccess || hipEventCreate(&p->pairs[i].b) != hipSuccess) break;
    p->capacity = i + 1;
  }
  return p;
}

void saf_profiler_destroy(saf_profiler* p) {
  if (!p) return;
  for (int i = 0; i < p->capacity; ++i) {
    (void)hipEventDestroy(p->pairs[i].a);
    (void)hipEventDestroy(p->pairs[i].b);
  }
  delete[] p->pairs;
  delete p;
}

void saf_profiler_reset(saf_profiler* p) {
  if (p) p->used = 0;
}

int saf_profiler_read(saf_profiler* p, int32_t kernel_class, double* total_ms, int64_t* launches) {
  if (!p || !total_ms || !launches) return fail(SAF_E_INVALID, "profiler_read: bad arguments");
  double tot = 0;
  int64_t n = 0;
  for (int i = 0; i < p->used; ++i) {
    if (p->pairs[i].cls != kernel_class) continue;
    float ms = 0.f;
    hipError_t e = hipEventElapsedTime(&ms, p->pairs[i].a, p->pairs[i].b);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipEventElapsedTime: %s", hipGetErrorString(e));
    tot += ms;
    ++n;
  }
  *total_ms = tot;
  *launches = n;
  return SAF_OK;
}

}  // extern "C"
