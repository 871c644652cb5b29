// saf_fuse.hip -- projective voxel fusion of RGB-D frames on gfx950 (MI355X).
//
// Replaces ClipFusion.integrate / ClipSeemFusion.integrate after the backbone calls
// (reference clipfusion.py:647-721, clip_seem_fusion.py:697-822).  Two launches per frame:
//
//   sweep  : one pass over ALL voxels (rows a2-a4 of SURVEY.md §8): project the voxel centre,
//            nearest-pixel depth test, TSDF running mean for tsdf_valid voxels (16-byte runs of 4
//            consecutive voxels), wave-ballot + LDS compaction of the `valid` voxels of a 4096-voxel
//            chunk into one of 16 compact lists (one global atomic per block), write-through
//            publication of the lists + a completion counter.  VALU-bound; no feature row is touched.
//   fuse   : (a5-a7) workgroups stage the frame's feature map into LDS as a conflict-free image and
//            walk the compact lists; a group of G lanes owns one voxel row: 4-tap bilinear sample
//            from LDS, running-mean read-modify-write of the D-row with 16-byte non-temporal
//            accesses, R rows in flight per group; rgb / weight / label counter are done
//            lane-parallel, once per voxel.  HBM-bound: Nv * (2*D*s + 36 [+8]) bytes per frame.
//
// saf_fuse_frames pipelines the two over frames: sweeps run ahead on an auxiliary stream, fuse
// kernels back to back on the caller's stream, sweep(i) -> fuse(i) through a device-side counter
// (DESIGN.md §4).  All arithmetic that selects voxels is shared with the oracle's restatement in
// spirit and checked bit-for-bit against it (saf_common.h).
#include <math.h>
#include <stdlib.h>
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

// Workspace: a header with eight rotating sets of list counters (frame i uses set i & 7 and zeroes
// set (i+1) & 7 for its successor) and the sweep-completion counter, then kListBuffers buffers
// (the sweep/fuse pipeline of saf_fuse_frames lets the sweep run ahead), each holding the compact
// lists of one frame and -- only for feature maps too large for LDS -- the map image.
constexpr int kListBuffers = 4;      // the sweep may run up to 4 frames ahead of the fuse
// Counter set i & 7 is zeroed by sweep(i - 1) and read by fuse(i).  sweep(j) may start once
// fuse(j - kListBuffers) is done, so the set of fuse(i) can be re-zeroed (by sweep(i + 7)) only
// after fuse(i + 3) -- hence after fuse(i) -- has finished.  (With 4 sets, sweep(i + 3) could zero
// the set fuse(i) is about to read.)
constexpr int kCounterSets = 8;
static_assert(kCounterSets > kListBuffers + 1, "a counter set must outlive the fuse kernel that reads it");
constexpr size_t kSweepDoneOff = (size_t)kCounterSets * kNumLists * sizeof(unsigned long long);  // 1024
constexpr int kDoneShards = 8;       // kDoneShards x u64: sweep blocks finished since the call started
constexpr size_t kHdrBytes = 2048;
struct WsLayout {
  size_t map_off, lists_off, half, total;
  uint32_t n_blocks, list_cap;
  bool lds_map;
};

// Feature-map image used by the fuse kernels: [D/VEC][Ppad][VEC] floats, i.e. channel-major like
// the backbone's [D][P] output but with channels grouped per lane access (VEC = 4 when D % 4 == 0)
// and each group padded to an ODD number of tap positions Ppad >= P + 1.  Position P (and the
// padding) holds zeros = grid_sample's zero padding.  Lane l reads 16 bytes at
// ((ch_l * Ppad + tap) * 16): consecutive lanes are 4*Ppad words apart, and 4*odd is
// conflict-free over the 64 LDS banks for every 16-lane group of a ds_read_b128.
__host__ __device__ inline int map_ppad(int P) { return (P + 1) | 1; }
size_t lds_map_bytes(int D, int P) { return (size_t)D * map_ppad(P) * sizeof(float); }

WsLayout ws_layout(int64_t n_vox, int D, int P) {
  WsLayout w;
  w.n_blocks = (uint32_t)((n_vox + kSweepChunk - 1) / kSweepChunk);
  uint32_t per_list = (w.n_blocks + kNumLists - 1) / kNumLists;
  w.list_cap = per_list * kSweepChunk;
  w.lds_map = lds_map_bytes(D, P) <= 144 * 1024;
  w.map_off = 0;  // the map image, used when !lds_map
  size_t map_bytes = w.lds_map ? 0 : ((lds_map_bytes(D, P) + 255) & ~(size_t)255);
  w.lists_off = w.map_off + map_bytes;
  w.half = w.lists_off + (size_t)kNumLists * w.list_cap * sizeof(uint32_t);
  w.half = (w.half + 255) & ~(size_t)255;
  w.total = kHdrBytes + kListBuffers * w.half;
  return w;
}

// ------------------------------------------------------------------------------------------
// Feature-map image: element (c, p) of the backbone's [D][P] map goes to float index
// ((c / VEC) * Ppad + p) * VEC + c % VEC; everything else is zero.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void fill_map_image(float* __restrict__ dst, const float* __restrict__ feat_map, int D,
                                               int P, int vec, int tid, int nthreads) {
  const int ppad = map_ppad(P);
  const int total = D * ppad;
  const int vshift = vec == 4 ? 2 : 0;
  const float rp = 1.0f / (float)ppad;
  for (int o = tid; o < total; o += nthreads) {
    const int k = o & (vec - 1), t = o >> vshift;  // t = cv * ppad + p
    // t / ppad via floor((t + 0.5) * fl(1/ppad)): exact for t < 2^22 (see sdiv)
    const int cv = (int)(((float)t + 0.5f) * rp), p = t - cv * ppad;
    dst[o] = p < P ? feat_map[(size_t)((cv << vshift) + k) * P + p] : 0.0f;
  }
}

// Pixel-major image for the windowed path, whose taps are read from global memory (L2): row p holds
// the D channels of map position p contiguously (a wave's tap load is one contiguous D*4 bytes), row P
// is the zero row of the taps outside the map.
struct PrepArgs {
  const float* feat_map[64];  // >= kWin
};
__global__ __launch_bounds__(256) void prep_rows_kernel(PrepArgs pa, float* __restrict__ imgs, int img_floats, int D,
                                                        int P) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= (P + 1) * D) return;
  const float* __restrict__ feat_map = pa.feat_map[blockIdx.y];
  const int c = o % D, p = o / D;
  imgs[(size_t)blockIdx.y * img_floats + o] = p < P ? feat_map[(size_t)c * P + p] : 0.0f;
}

// Only when the image does not fit LDS: build it once per frame in the workspace.
__global__ __launch_bounds__(256) void prep_kernel(const float* __restrict__ feat_map, float* __restrict__ map_img,
                                                   int D, int P, int vec) {
  fill_map_image(map_img, feat_map, D, P, vec, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// ------------------------------------------------------------------------------------------
// sweep: classify every voxel, TSDF running mean, compact the valid ones.
//
// A block owns kSweepChunk = 4096 consecutive voxels; a thread owns 4 runs of 4 CONSECUTIVE
// voxels (flat index n..n+3, i.e. along z), so the TSDF value / weight of a run is one 16-byte
// access and a wave covers 256 consecutive voxels.  The four voxels of a run are processed in
// lock step (ILP 4): all depth gathers, then all TSDF loads, are in flight together.
// ------------------------------------------------------------------------------------------
constexpr int kRun = 4;
constexpr int kRunsPerThread = kSweepPerThread / kRun;

// t / d for 0 <= t < 2^13, 1 <= d: floor((t + 0.5) * fl(1/d)) -- the product's error (< 2^-10/d) is
// far below the 0.5/d distance of (t + 0.5)/d from any integer.  Divisors above 4096 take compares
// (t < 4096 + d <= 2d there).
struct SmallDiv {
  float r;
  uint32_t d;
};
__device__ __forceinline__ uint32_t sdiv(uint32_t t, const SmallDiv& f) {
  if (f.d > 4096u) return t >= f.d ? 1u : 0u;
  return (uint32_t)(((float)t + 0.5f) * f.r);
}

__global__ __launch_bounds__(kSweepThreads) void sweep_kernel(KVol v, KFrame f,
                                                               unsigned long long* __restrict__ counts,
                                                               unsigned long long* __restrict__ next_counts,
                                                               uint32_t* __restrict__ lists, uint32_t list_cap,
                                                               unsigned long long* __restrict__ sweep_done) {
  __shared__ uint32_t s_buf[kSweepChunk];
  __shared__ float s_axes[kAxisLds];
  __shared__ uint32_t s_count, s_base, s_nt;
  const int tid = threadIdx.x, lane = tid & 63;
  const Cam cam = load_cam(f.pose, f.K, f.W, f.H);
  const bool lds_axes = v.nx + v.ny + v.nz <= kAxisLds;
  if (lds_axes) {
    for (int i = tid; i < v.nx + v.ny + v.nz; i += kSweepThreads)
      s_axes[i] = i < v.nx ? v.ax[i] : (i < v.nx + v.ny ? v.ay[i - v.nx] : v.az[i - v.nx - v.ny]);
  }
  if (tid == 0) {
    s_count = 0;
    s_nt = 0;
  }
  if (blockIdx.x == 0 && tid < kNumLists) next_counts[tid] = 0ull;  // the successor frame's counters
  __syncthreads();
  const float* ax = lds_axes ? s_axes : v.ax;
  const float* ay = lds_axes ? s_axes + v.nx : v.ay;
  const float* az = lds_axes ? s_axes + v.nx + v.ny : v.az;
  // coordinates of the block's first voxel (uniform), then small per-thread offsets
  const uint32_t chunk_base = blockIdx.x * (uint32_t)kSweepChunk;
  int bx, by, bz;
  voxel_coords(v, chunk_base < v.N ? chunk_base : 0u, bx, by, bz);
  const SmallDiv dz{1.0f / (float)v.nz, (uint32_t)v.nz}, dy{1.0f / (float)v.ny, (uint32_t)v.ny};
  const float rtrunc = 1.0f / v.trunc;
  const bool aligned = (((uintptr_t)v.tsdf | (uintptr_t)v.tsdf_w) & 15) == 0;
  uint32_t nt_local = 0;
  for (int g = 0; g < kRunsPerThread; ++g) {
    const uint32_t off = (uint32_t)g * (kSweepThreads * kRun) + (uint32_t)tid * kRun;  // < 4096
    const uint32_t n0 = chunk_base + off;
    uint32_t n[kRun];
    Proj p[kRun];
    int pix[kRun];
    bool in_view[kRun], valid[kRun], tv[kRun];
    float depth[kRun], sdf[kRun];
    float xw[kRun], yw[kRun], zw[kRun];
    // phase A: voxel centre -> image (clipfusion.py:647-659)
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
      n[j] = n0 + j;
      const uint32_t tz = (uint32_t)bz + off + j;  // < 4096 + nz
      const uint32_t qz = sdiv(tz, dz);
      const uint32_t ty = (uint32_t)by + qz;
      const uint32_t qy = sdiv(ty, dy);
      const int iz = (int)(tz - qz * (uint32_t)v.nz), iy = (int)(ty - qy * (uint32_t)v.ny);
      const int ix = min(bx + (int)qy, v.nx - 1);  // clamped only for the (unused) lanes past N
      xw[j] = ax[ix];
      yw[j] = ay[iy];
      zw[j] = az[iz];
    }
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
      p[j] = project(cam, xw[j], yw[j], zw[j]);
      // _valid = (grid.abs() <= 1).all(dim=1) & (z > 0)            clipfusion.py:673
      in_view[j] = (n[j] < v.N) && (fabsf(p[j].gx) <= 1.0f) && (fabsf(p[j].gy) <= 1.0f) && (p[j].z > 0.0f);
      pix[j] = in_view[j] ? nearest_index(p[j].gx, p[j].gy, cam, f.W) : -1;
    }
    // phase B: nearest-pixel depth (zeros padding), all gathers in flight together
#pragma unroll
    for (int j = 0; j < kRun; ++j) depth[j] = pix[j] >= 0 ? f.depth[pix[j]] : 0.0f;
    // phase C: classify (clipfusion.py:669-679)
    bool any_tv = false;
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
      // (depth - z) / trunc: exact fast division; a = +inf (infinite depth reading) must stay +inf so
      // that the voxel counts as in front of the surface, NaN / -inf fail both tests either way
      const float num = depth[j] - p[j].z;
      sdf[j] = num == INFINITY ? INFINITY : div_by_uniform(num, v.trunc, rtrunc);
      valid[j] = in_view[j] && fabsf(sdf[j]) <= 1.0f;
      tv[j] = in_view[j] && sdf[j] > -1.0f;
      any_tv |= tv[j];
    }
    // phase D: TSDF running mean of the clamped sdf, clipfusion.py:681-695 with B = 1.  The run's
    // four values travel as one 16-byte vector; untouched elements are written back unchanged
    // (this thread is the only writer of its run).
    if (any_tv) {
      float told[kRun];
      int w0[kRun];
      const bool vec = aligned && (n0 + kRun <= v.N);
      if (vec) {
        const float4 t4 = *reinterpret_cast<const float4*>(v.tsdf + n0);
        const int4 w4 = *reinterpret_cast<const int4*>(v.tsdf_w + n0);
        told[0] = t4.x; told[1] = t4.y; told[2] = t4.z; told[3] = t4.w;
        w0[0] = w4.x; w0[1] = w4.y; w0[2] = w4.z; w0[3] = w4.w;
      } else {
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
          told[j] = tv[j] ? v.tsdf[n[j]] : 0.0f;
          w0[j] = tv[j] ? v.tsdf_w[n[j]] : 0;
        }
      }
#pragma unroll
      for (int j = 0; j < kRun; ++j) {
        if (tv[j]) {
          const float t = sdf[j] > 1.0f ? 1.0f : sdf[j];
          const int w1 = w0[j] + 1;
          if (v.accum == SAF_SUM) {
            told[j] = told[j] + t;
          } else {
            // batch_tsdf / new_weight + tsdf * (tsdf_weight / new_weight).  The TSDF is a VALUE
            // (compared at 1e-4), not an index: both quotients use the hardware reciprocal (1 ulp).
            const float rw = __builtin_amdgcn_rcpf((float)w1);
            told[j] = t * rw + told[j] * ((float)w0[j] * rw);
          }
          w0[j] = w1;
          ++nt_local;
        }
      }
      if (vec) {
        *reinterpret_cast<float4*>(v.tsdf + n0) = make_float4(told[0], told[1], told[2], told[3]);
        *reinterpret_cast<int4*>(v.tsdf_w + n0) = make_int4(w0[0], w0[1], w0[2], w0[3]);
      } else {
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
          if (tv[j]) {
            v.tsdf[n[j]] = told[j];
            v.tsdf_w[n[j]] = w0[j];
          }
        }
      }
    }
    {
      // phase E: compaction -- wave64 ballots + prefix popcounts, ONE LDS atomic per wave and run
      unsigned long long m[kRun];
      uint32_t cnt = 0;
#pragma unroll
      for (int j = 0; j < kRun; ++j) {
        m[j] = __ballot(valid[j]);
        cnt += (uint32_t)__popcll(m[j]);
      }
      if (cnt) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_count, cnt);
        base = __shfl(base, 0);
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int j = 0; j < kRun; ++j) {
          if (valid[j]) s_buf[base + (uint32_t)__popcll(m[j] & lt)] = n[j];
          base += (uint32_t)__popcll(m[j]);
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nt_local += __shfl_down(nt_local, o);
  if (lane == 0 && nt_local) atomicAdd(&s_nt, nt_local);
  __syncthreads();
  {
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
      // write-through (sc1) stores: the entries are the only thing the concurrently running fuse
      // kernel reads from this kernel, and it reads them with sc1 loads
      for (uint32_t i = tid; i < total; i += kSweepThreads)
        __hip_atomic_store(&dst[i], s_buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // Publish (CDNA guide, inter-workgroup hand-off without a release fence): every storing wave
  // drains its write-through stores, the workgroup joins, then ONE lane bumps a shard of the
  // completion counter that the fuse kernel polls.  No L2 write-back is forced.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0)
    (void)__hip_atomic_fetch_add(&sweep_done[blockIdx.x % kDoneShards], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
__device__ __forceinline__ unsigned long long sweep_blocks_done(const unsigned long long* __restrict__ sweep_done) {
  unsigned long long n = 0;
#pragma unroll
  for (int k = 0; k < kDoneShards; ++k) n += __hip_atomic_load(&sweep_done[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return n;
}
__device__ __forceinline__ bool wait_for_sweep(const unsigned long long* __restrict__ sweep_done,
                                               unsigned long long target, unsigned long long* __restrict__ stats) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    int ok = 1;
    if (sweep_blocks_done(sweep_done) < target) {
      const unsigned long long t0 = wall_clock64();
      while (sweep_blocks_done(sweep_done) < target) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 200000000ull) {
          ok = 0;
          if (stats) atomicAdd(&stats[4], 1ull);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

// Once per frame (block 0, thread 0): fold the per-list counters into the caller's statistics.
__device__ __forceinline__ void add_frame_stats(const unsigned long long* __restrict__ counts,
                                                unsigned long long* __restrict__ stats) {
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long nv = 0, nt = 0;
    for (int l = 0; l < kNumLists; ++l) {
      const unsigned long long c = __hip_atomic_load(&counts[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                                                             const float* __restrict__ feat_map, const float* __restrict__ map_t, int g_log2,
                                                             unsigned long long* __restrict__ stats,
                                                             const unsigned long long* __restrict__ sweep_done,
                                                             unsigned long long sweep_target) {
  using V = typename VecT<VEC>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int tid = threadIdx.x;
  const int DV = v.D / VEC;  // vector chunks per row
  const int P = f.npy * f.npx;
  const int ppad = map_ppad(P);
  const V* map;
  if (LDS_MAP) {
    fill_map_image(reinterpret_cast<float*>(s_raw), feat_map, v.D, P, VEC, tid, kFuseThreads);
    __syncthreads();
    map = reinterpret_cast<const V*>(s_raw);
  } else {
    map = reinterpret_cast<const V*>(map_t);
  }
  if (!wait_for_sweep(sweep_done, sweep_target, stats)) return;  // lists + counters are published
  const Cam cam = load_cam(f.pose, f.K, f.W, f.H);
  const float half_px = (float)f.npx / 2.0f, half_py = (float)f.npy / 2.0f;
  const bool sum = v.accum == SAF_SUM;

  const int G = 1 << g_log2;
  const int lane = tid & 63, wave = tid >> 6;
  const int slot = lane >> g_log2, gl = lane & (G - 1);
  const int epw = 64 >> g_log2;  // entries per wave per step
  const uint32_t list = blockIdx.x % kNumLists;
  const uint32_t wg_in_list = blockIdx.x / kNumLists, wgs_per_list = gridDim.x / kNumLists;
  const uint32_t count = (uint32_t)__hip_atomic_load(&counts[list], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
      n[j] = act[j] ? __hip_atomic_load(&lst[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
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
        tp[j] = tap_offsets(bf[j], f.npx, f.npy);
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
              const int mb = ch * ppad;
              const V s = lerp_taps(map[mb + t.o_nw], map[mb + t.o_ne], map[mb + t.o_sw], map[mb + t.o_se], bf[j]);
              feat[(int64_t)n[j] * DV + ch] = blend(s, old[j][c], a[j], b[j], sum);
            }
          }
        } else {
          for (int ch = gl; ch < DV; ch += G) {
            const int mb = ch * ppad;
            const V s = lerp_taps(map[mb + t.o_nw], map[mb + t.o_ne], map[mb + t.o_sw], map[mb + t.o_se], bf[j]);
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
                                            float wy_l, int npx, int npy) {
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
  c.t = tap_offsets(c.w, npx, npy);
  return c;
}

// The row loop is written without per-lane guards so that it compiles to straight-line code with
// counted waits: lanes whose chunk index would fall past the row re-do the last chunk (identical
// value to the same address), and lane groups past the end of a batch re-do the batch's last
// row inside the SAME wave instruction as its owner (same loads, same stores).
// BF16: the feature volume holds bfloat16 (a 16-byte lane access = 8 channels = two map chunks);
// samples are blended in fp32 exactly as for the fp32 volume and rounded to nearest-even once per
// update.
template <int CPL, int R, bool G64, bool LDS_MAP, bool SUM, bool BF16>
__global__ __launch_bounds__(kFuseThreads) void fuse_rows_kernel(KVol v, KFrame f,
                                                                  const unsigned long long* __restrict__ counts,
                                                                  const uint32_t* __restrict__ lists,
                                                                  uint32_t list_cap, const float* __restrict__ feat_map,
                                                                  const float* __restrict__ map_t, int g_log2,
                                                                  unsigned long long* __restrict__ stats,
                                                                  const unsigned long long* __restrict__ sweep_done,
                                                                  unsigned long long sweep_target) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int tid = threadIdx.x;
  const int DV = v.D >> 2;
  const int P = f.npy * f.npx;
  const int ppad = map_ppad(P);
  const float4* map;
  if (LDS_MAP) {
    fill_map_image(reinterpret_cast<float*>(s_raw), feat_map, v.D, P, 4, tid, kFuseThreads);
    __syncthreads();
    map = reinterpret_cast<const float4*>(s_raw);
  } else {
    map = reinterpret_cast<const float4*>(map_t);
  }
  if (!wait_for_sweep(sweep_done, sweep_target, stats)) return;  // lists + counters are published
  const Cam cam = load_cam(f.pose, f.K, f.W, f.H);
  const float half_px = (float)f.npx / 2.0f, half_py = (float)f.npy / 2.0f;
  constexpr bool sum = SUM;
  add_frame_stats(counts, stats);

  const int G = G64 ? 64 : (1 << g_log2);
  const int lane = tid & 63, wave = tid >> 6;
  const int slot = G64 ? 0 : (lane >> g_log2), gl = lane & (G - 1);
  const int epw = G64 ? 1 : (64 >> g_log2);  // rows per wave step
  constexpr int KV = BF16 ? 2 : 1;  // map chunks (4 channels each) per 16-byte row unit
  const int DU = DV / KV;           // 16-byte units per row
  int chs[CPL], mbs[CPL][KV];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    chs[c] = min(gl + c * G, DU - 1);
#pragma unroll
    for (int k = 0; k < KV; ++k) mbs[c][k] = (chs[c] * KV + k) * ppad;
  }
  const uint32_t list = blockIdx.x % kNumLists;
  const uint32_t count = (uint32_t)__hip_atomic_load(&counts[list], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
      n_l = __hip_atomic_load(&lst[b0 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
#define SAF_FETCH(i) fetch_row<G64>(min((i) * epw + slot, last), n_l, a_l, b_l, xy_l, wx_l, wy_l, f.npx, f.npy)
#define SAF_LOAD(ctx_, old_)                                                           \
  _Pragma("unroll") for (int c = 0; c < CPL; ++c) old_[c] = ld_stream(&feat[(int64_t)(ctx_).n * DU + chs[c]]);
#define SAF_TAPS(c_, k_, ctx_)                                                                         \
  lerp_taps(map[mbs[c_][k_] + (ctx_).t.o_nw], map[mbs[c_][k_] + (ctx_).t.o_ne], map[mbs[c_][k_] + (ctx_).t.o_sw], \
            map[mbs[c_][k_] + (ctx_).t.o_se], (ctx_).w)
#define SAF_STORE(ctx_, old_)                                                                          \
  _Pragma("unroll") for (int c = 0; c < CPL; ++c) {                                                    \
    float4 out_;                                                                                       \
    if (BF16) {                                                                                        \
      const uint32_t w0_ = __builtin_bit_cast(uint32_t, old_[c].x), w1_ = __builtin_bit_cast(uint32_t, old_[c].y); \
      const uint32_t w2_ = __builtin_bit_cast(uint32_t, old_[c].z), w3_ = __builtin_bit_cast(uint32_t, old_[c].w); \
      const float4 n0_ = blend(SAF_TAPS(c, 0, ctx_), make_float4(bf16_lo(w0_), bf16_hi(w0_), bf16_lo(w1_), bf16_hi(w1_)), \
                               (ctx_).a, (ctx_).b, sum);                                               \
      const float4 n1_ = blend(SAF_TAPS(c, KV - 1, ctx_), make_float4(bf16_lo(w2_), bf16_hi(w2_), bf16_lo(w3_), bf16_hi(w3_)), \
                               (ctx_).a, (ctx_).b, sum);                                               \
      out_.x = __builtin_bit_cast(float, pack_bf16(n0_.x, n0_.y));                                     \
      out_.y = __builtin_bit_cast(float, pack_bf16(n0_.z, n0_.w));                                     \
      out_.z = __builtin_bit_cast(float, pack_bf16(n1_.x, n1_.y));                                     \
      out_.w = __builtin_bit_cast(float, pack_bf16(n1_.z, n1_.w));                                     \
    } else {                                                                                           \
      out_ = blend(SAF_TAPS(c, 0, ctx_), old_[c], (ctx_).a, (ctx_).b, sum);                            \
    }                                                                                                  \
    st_stream(&feat[(int64_t)(ctx_).n * DU + chs[c]], out_);                                           \
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
#undef SAF_TAPS
#undef SAF_STORE
  }
}

// ------------------------------------------------------------------------------------------
// fuse, voxel-major over a WINDOW of up to 64 frames (saf_fuse_frames with many frames).
//
// A running mean is applied hit by hit, but nothing forces a row to travel to HBM between two hits.
// Per window, on the caller's stream:
//   classify_window_kernel  (one launch per 32 frames) every voxel against 32 frames (the full-grid sweep of
//                           clipfusion.py:647-695): TSDF running mean kept in registers across the frames
//                           and written once, one 32-bit frame mask per voxel into that launch's mask plane;
//   fuse_window_kernel      every touched voxel's D-row is read ONCE, the voxel's hits are applied in frame
//                           order -- the same s*a + old*b with a = 1/(w+1), so the result is bit-identical to
//                           fusing the frames one after the other -- and written ONCE.  Row bytes fall by the
//                           window's hits-per-voxel ratio (1.8 for incoherent depth, 7 for a coherent scene).
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
constexpr int kWin = SAF_WINDOW_FRAMES;  // frames per window: two 32-bit mask words per voxel
static_assert(kWin == 64, "the mask layout and the 6-bit frame field assume 64-frame windows");
constexpr int kMaskWords = kWin / 32;
constexpr int kWinMinFrames = 16;  // shorter calls run the per-frame pipeline
#ifndef SAF_WIN_HITCAP
#define SAF_WIN_HITCAP 128
#endif
constexpr int kHitCap = SAF_WIN_HITCAP;
constexpr int kWinThreads = 256;
constexpr int kWinWaves = kWinThreads / 64;
constexpr int kPiece = 256;

struct WinArgs {
  int F, H, W, npy, npx, rgb_bilinear;
  const float* depth[kWin];
  const float* rgb[kWin];
  const float* pose[kWin];
  const float* K[kWin];
  const float* label_map[kWin];
};

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave execute in order; this only stops the compiler from moving them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef SAF_WIN_TIMING  // development aid: per-phase wave cycles of the window kernel, printed by the host
__device__ unsigned long long g_win_t[16];
#define WT_DECL unsigned long long wt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wt_last_ = __builtin_readcyclecounter()
#define WT(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); wt_[k] += n_ - wt_last_; wt_last_ = n_; } while (0)
#define WT_FLUSH do { if (lane == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_win_t[k_], wt_[k_]); } while (0)
#else
#define WT_DECL
#define WT(k)
#define WT_FLUSH
#endif

#ifndef SAF_WIN_P2
#define SAF_WIN_P2 4
#endif
#ifndef SAF_WIN_SR2
#define SAF_WIN_SR2 6
#endif
#ifndef SAF_WIN_WPE
#define SAF_WIN_WPE 2
#endif
template <int CPL>
struct WinCfg {
  static constexpr int SR = CPL == 1 ? 8 : (CPL == 2 ? SAF_WIN_SR2 : 3);  // rows of a sub-chunk (LDS resident)
  static constexpr int P = CPL == 1 ? 6 : (CPL == 2 ? SAF_WIN_P2 : 2);  // tap groups in flight
  // dynamic LDS layout (bytes)
  static constexpr size_t rows_off = 0;
  static constexpr size_t rows_bytes = (size_t)kWinWaves * SR * CPL * 64 * sizeof(float4);
  static constexpr size_t stage_off = rows_off + rows_bytes;  // 6 arrays of kHitCap words per wave
  static constexpr size_t stage_bytes = (size_t)kWinWaves * 6 * kHitCap * 4;
  static constexpr size_t tm_off = stage_off + stage_bytes;
  static constexpr size_t tm_bytes = (size_t)kWinWaves * kPiece * 4 * kMaskWords;
  static constexpr size_t tv_off = tm_off + tm_bytes;
  static constexpr size_t tv_bytes = (size_t)kWinWaves * kPiece * 2;
  static constexpr size_t ptr_off = tv_off + tv_bytes;
  static constexpr size_t ptr_bytes = (size_t)2 * kWin * sizeof(const float*);
  static constexpr size_t cam_off = ptr_off + ptr_bytes;
  static constexpr size_t total = cam_off + (size_t)kWin * sizeof(Cam);
};

// Classification of one piece (256 consecutive voxels, a lane owns 4 of them) against every frame of a
// window (clipfusion.py:647-679): the voxels' TSDF running mean is kept in registers across the frames
// (clipfusion.py:681-695 with B = 1, frame after frame -- order dependent) and written once; mk4[j] collects
// the frame bitmask of voxel j.  KFU frames at a time: their depth gathers are in flight together.
template <int KFU, bool SUM>
__device__ __forceinline__ void classify_piece(const KVol& v, const WinArgs& wa, const Cam* __restrict__ s_cam,
                                               uint32_t piece_base, int lane, float rtrunc, bool tsdf_aligned,
                                               int f_begin, int f_end, uint32_t (&mk4)[4], unsigned long long& nt_done,
                                               unsigned long long& tsdf_rows_done) {
  const uint32_t nb = piece_base + (uint32_t)lane * 4u;
  float xw[4], yw[4], zw[4], told[4];
  int tw[4];
  bool inb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    inb[j] = nb + j < v.N;
    int ix, iy, iz;
    voxel_coords(v, inb[j] ? nb + j : 0u, ix, iy, iz);
    xw[j] = v.ax[ix];
    yw[j] = v.ay[iy];
    zw[j] = v.az[iz];
  }
  const bool vec = tsdf_aligned && nb + 4u <= v.N;
  if (vec) {
    const float4 t4 = *reinterpret_cast<const float4*>(v.tsdf + nb);
    const int4 w4 = *reinterpret_cast<const int4*>(v.tsdf_w + nb);
    told[0] = t4.x; told[1] = t4.y; told[2] = t4.z; told[3] = t4.w;
    tw[0] = w4.x; tw[1] = w4.y; tw[2] = w4.z; tw[3] = w4.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      told[j] = inb[j] ? v.tsdf[nb + j] : 0.0f;
      tw[j] = inb[j] ? v.tsdf_w[nb + j] : 0;
    }
  }
  uint32_t touched = 0;  // bit j: voxel j's TSDF changed
  // kFU frames at a time: all their depth gathers are in flight together, then the frames are
  // applied one after the other (the TSDF running mean is order dependent)
  constexpr int kFU = KFU;
  for (int f0 = f_begin; f0 < f_end; f0 += kFU) {
    int pix[kFU][4];  // >= 0: pixel; -1: in view, no pixel (zeros padding); -2: not in view
    float pz[kFU][4];
#pragma unroll
    for (int u = 0; u < kFU; ++u) {
      const bool live = f0 + u < f_end;
      const Cam cam = s_cam[live ? f0 + u : 0];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const Proj p = project(cam, xw[j], yw[j], zw[j]);
        const bool in_view = live && inb[j] && (fabsf(p.gx) <= 1.0f) && (fabsf(p.gy) <= 1.0f) && (p.z > 0.0f);
        const int px = nearest_index(p.gx, p.gy, cam, wa.W);
        pix[u][j] = in_view ? (px >= 0 ? px : -1) : -2;
        pz[u][j] = p.z;
      }
    }
    float depth[kFU][4];
#pragma unroll
    for (int u = 0; u < kFU; ++u) {
      const float* __restrict__ dimg = wa.depth[f0 + u < f_end ? f0 + u : 0];
#pragma unroll
      for (int j = 0; j < 4; ++j) depth[u][j] = pix[u][j] >= 0 ? dimg[pix[u][j]] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kFU; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in_view = pix[u][j] != -2;
        const float num = depth[u][j] - pz[u][j];
        const float sdf = num == INFINITY ? INFINITY : div_by_uniform(num, v.trunc, rtrunc);
        if (in_view && fabsf(sdf) <= 1.0f) mk4[j] |= 1u << (f0 + u - f_begin);
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
          v.tsdf[nb + j] = told[j];
          v.tsdf_w[nb + j] = tw[j];
        }
      }
    }
  }
}

// classify_window_kernel: one launch per 32 frames of a window [f_begin, f_end); leaves that mask word of
// every voxel in its plane of `hitmask` and the updated TSDF.  (One launch over all 64 frames keeps the TSDF
// in registers twice as long but puts 64 depth-image footprints in L2 at once: 4.2 ms against 2 x 1.9 ms.)
template <bool SUM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void classify_window_kernel(
    KVol v, WinArgs wa, int f_begin, int f_end, int tile, uint32_t* __restrict__ hitmask,
    unsigned long long* __restrict__ stats) {
  __shared__ Cam s_cam[kWin];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid >= f_begin && tid < f_end) s_cam[tid] = load_cam(wa.pose[tid], wa.K[tid], wa.W, wa.H);
  __syncthreads();
  if (stats && tid == 0 && blockIdx.x == 0) atomicAdd(&stats[2], (unsigned long long)(f_end - f_begin));
  const uint32_t n_pieces = (v.N + kPiece - 1) / kPiece;
  uint32_t piece = blockIdx.x * 4u + (uint32_t)wave;
  if (piece >= n_pieces) return;
  if (tile > 0) {
    // Workgroups are dispatched in index order, so the few thousand pieces in flight at any time are
    // consecutive indices.  In linear order those are a few whole x-planes of the grid -- seen face-on
    // they cover the whole depth image of a frame.  Walking the (x-plane, piece-in-plane) rectangle in
    // tiles of tile x tile keeps the pieces in flight inside a compact box, whose footprint in every
    // frame's depth image is small enough for the frames of this launch to stay in L2 together.
    const uint32_t ppx = (uint32_t)(((int64_t)v.ny * v.nz) / kPiece);  // pieces per x-plane (exact, checked by the host)
    const uint32_t T = (uint32_t)tile, tj = ppx / T, per_tile = T * T;
    const uint32_t t = piece / per_tile, r = piece - t * per_tile;
    const uint32_t tx = t / tj, ty = t - tx * tj;
    piece = (tx * T + r / T) * ppx + ty * T + (r - (r / T) * T);
  }
  const float rtrunc = 1.0f / v.trunc;
  const bool tsdf_aligned = (((uintptr_t)v.tsdf | (uintptr_t)v.tsdf_w) & 15) == 0;
  unsigned long long nt_done = 0, tsdf_rows_done = 0;
  const uint32_t nb = piece * (uint32_t)kPiece + (uint32_t)lane * 4u;
  uint32_t mk4[4] = {0u, 0u, 0u, 0u};
  classify_piece<4, SUM>(v, wa, s_cam, piece * (uint32_t)kPiece, lane, rtrunc, tsdf_aligned, f_begin, f_end, mk4, nt_done,
                         tsdf_rows_done);
  if (nb + 3u < v.N) {
    *reinterpret_cast<uint4*>(hitmask + nb) = make_uint4(mk4[0], mk4[1], mk4[2], mk4[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (nb + k < v.N) hitmask[nb + k] = mk4[k];
  }
  for (int o = 32; o > 0; o >>= 1) {
    nt_done += __shfl_xor(nt_done, o);
    tsdf_rows_done += __shfl_xor(tsdf_rows_done, o);
  }
  if (stats && lane == 0) {
    if (nt_done) atomicAdd(&stats[1], nt_done);
    if (tsdf_rows_done) atomicAdd(&stats[6], tsdf_rows_done);
  }
}

// One hit of a sub-chunk, held by the lane with the hit's index.
struct WinHit {
  int row;  // row slot in the LDS buffer
  float a, b, nw, ne, sw, se;
};
template <int CPL>
struct WinCtx {
  const float4* imgs;
  int img_vecs, DV, npx, npy, zero_row, lane;
  float4* rows;
};
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
__device__ __forceinline__ float bf16_round(float x) { return bf16_lo(f32_to_bf16_bits(x)); }

// NB groups: request the four map rows of every group, then blend group after group into the LDS rows
// (the waits are counted: group u is processed while the rows of groups u+1.. are still in flight).
template <int NB, int CPL, bool SUM, bool BF16, int SR>
__device__ __forceinline__ void win_batch(const WinCtx<CPL>& cx, int g0, uint32_t gk, uint32_t gm_lo, uint32_t gm_hi,
                                          const WinHit& rec, const WinRaw<SR, BF16 ? CPL / 2 : 1>& raw) {
  float4 tp[NB][4][CPL];
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)gk, g0 + u);
    const int fb = (int)(k >> 16), x0 = (int)(k & 255u) - 2, y0 = (int)((k >> 8) & 255u) - 2;
    const bool x0ok = x0 >= 0 && x0 < cx.npx, x1ok = x0 + 1 >= 0 && x0 + 1 < cx.npx;
    const bool y0ok = y0 >= 0 && y0 < cx.npy, y1ok = y0 + 1 >= 0 && y0 + 1 < cx.npy;
    const int o_nw = (x0ok && y0ok) ? y0 * cx.npx + x0 : cx.zero_row;
    const int o_ne = (x1ok && y0ok) ? y0 * cx.npx + x0 + 1 : cx.zero_row;
    const int o_sw = (x0ok && y1ok) ? (y0 + 1) * cx.npx + x0 : cx.zero_row;
    const int o_se = (x1ok && y1ok) ? (y0 + 1) * cx.npx + x0 + 1 : cx.zero_row;
    const float4* img = cx.imgs + (int64_t)fb * cx.img_vecs + (BF16 ? 2 * cx.lane : cx.lane);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      tp[u][0][c] = img[o_nw * cx.DV + win_chunk_off<BF16>(c)];
      tp[u][1][c] = img[o_ne * cx.DV + win_chunk_off<BF16>(c)];
      tp[u][2][c] = img[o_sw * cx.DV + win_chunk_off<BF16>(c)];
      tp[u][3][c] = img[o_se * cx.DV + win_chunk_off<BF16>(c)];
    }
  }
  if (g0 == 0) {  // the sub-chunk's rows (issued before these loads) have landed after this
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
    unsigned long long mm = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)gm_lo, g0 + u) |
                            ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)gm_hi, g0 + u) << 32);
    while (mm) {
      const int l = __ffsll((long long)mm) - 1;
      mm &= mm - 1ull;
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
          nv.x = bf16_round(nv.x); nv.y = bf16_round(nv.y); nv.z = bf16_round(nv.z); nv.w = bf16_round(nv.w);
        }
        *rp = nv;
      }
    }
  }
}

template <int CPL, bool SUM, bool BF16>
__global__ __launch_bounds__(kWinThreads) __attribute__((amdgpu_waves_per_eu(SAF_WIN_WPE, SAF_WIN_WPE))) void
fuse_window_kernel(KVol v, WinArgs wa, const float* __restrict__ map_imgs, int img_vecs,
                   unsigned long long* __restrict__ stats, unsigned int* __restrict__ piece_ctr,
                   const uint32_t* __restrict__ hitmask, uint32_t mask_plane) {
  using Cfg = WinCfg<CPL>;
  constexpr int SR = Cfg::SR;
  // a bf16 sub-chunk also holds its raw rows in registers until they are widened: one tap group fewer in flight
  constexpr int P = BF16 && Cfg::P > 2 ? Cfg::P - 1 : Cfg::P;
  extern __shared__ __align__(16) unsigned char s_dyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4* rows = reinterpret_cast<float4*>(s_dyn + Cfg::rows_off) + (size_t)wave * SR * CPL * 64;
  uint32_t* stage = reinterpret_cast<uint32_t*>(s_dyn + Cfg::stage_off) + (size_t)wave * 6 * kHitCap;
  uint32_t* s_hf = stage;  // voxel lane (6 bits) | cell << 6 (16 bits) | frame of the window << 22 (6 bits)
  int* s_hw = reinterpret_cast<int*>(stage + kHitCap);
  float* s_ha = reinterpret_cast<float*>(stage + 2 * kHitCap);
  float* s_hb = reinterpret_cast<float*>(stage + 3 * kHitCap);
  float* s_hgx = reinterpret_cast<float*>(stage + 4 * kHitCap);
  float* s_hgy = reinterpret_cast<float*>(stage + 5 * kHitCap);
  uint32_t* s_tm = reinterpret_cast<uint32_t*>(s_dyn + Cfg::tm_off) + wave * kPiece * kMaskWords;
  uint16_t* s_tv = reinterpret_cast<uint16_t*>(s_dyn + Cfg::tv_off) + wave * kPiece;
  const float** s_rgb = reinterpret_cast<const float**>(s_dyn + Cfg::ptr_off);
  const float** s_lab = s_rgb + kWin;
  Cam* s_cam = reinterpret_cast<Cam*>(s_dyn + Cfg::cam_off);

  if (tid < wa.F) {
    s_cam[tid] = load_cam(wa.pose[tid], wa.K[tid], wa.W, wa.H);
    s_rgb[tid] = wa.rgb[tid];
    s_lab[tid] = wa.label_map[tid];
  }
  __syncthreads();
  const int DV = v.D >> 2;
  const float half_px = (float)wa.npx / 2.0f, half_py = (float)wa.npy / 2.0f;
  const int zero_row = wa.npx * wa.npy;
  int chs[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) chs[c] = lane + c * 64;
  constexpr int UPL = BF16 ? CPL / 2 : 1;  // 16-byte units of a bf16 row per lane
  float4* feat = reinterpret_cast<float4*>(v.feat);
  float4* featb = reinterpret_cast<float4*>(v.feat);  // bf16 volume: D / 8 units of 16 bytes per row
  const float4* imgs = reinterpret_cast<const float4*>(map_imgs);
  KFrame kf;  // per-hit view of a frame for the scalar side
  kf.H = wa.H; kf.W = wa.W; kf.npy = wa.npy; kf.npx = wa.npx; kf.rgb_bilinear = wa.rgb_bilinear;
  kf.depth = nullptr; kf.pose = nullptr; kf.K = nullptr;
  unsigned long long hits_done = 0, rows_done = 0;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  WT_DECL;
  // persistent grid (2 workgroups of 4 waves per CU, 71 KB of LDS each).  Pieces are handed out by an
  // atomic counter: a coherent scene concentrates its hits in few pieces (a wall = whole columns).
  const uint32_t n_pieces = (v.N + kPiece - 1) / kPiece;
  for (;;) {
    uint32_t piece = 0;
    if (lane == 0) piece = atomicAdd(piece_ctr, 1u);
    piece = (uint32_t)__builtin_amdgcn_readfirstlane((int)piece);
    if (piece >= n_pieces) break;
    const uint32_t piece_base = piece * (uint32_t)kPiece;
    // ---- the piece's touched voxels: (local id, frame mask) left by classify_window_kernel, compacted into LDS
    uint32_t mk4[4][kMaskWords];
    {
      const uint32_t nb = piece_base + (uint32_t)lane * 4u;
      // mask word w of every voxel lives in plane w (written by the classification launch of frames 32 w ..)
#pragma unroll
      for (int w = 0; w < kMaskWords; ++w) {
        const uint32_t* mrow = hitmask + (size_t)w * mask_plane + nb;
        if (w * 32 >= wa.F) {
          mk4[0][w] = mk4[1][w] = mk4[2][w] = mk4[3][w] = 0u;  // a short window has no second plane
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
      const bool touched = (mk4[k][0] | mk4[k][1]) != 0u;
      const unsigned long long bal = __ballot(touched);
      if (touched) {
        const int slot = T + __popcll(bal & lt_mask);
        s_tv[slot] = (uint16_t)(lane * 4 + k);
        s_tm[slot * kMaskWords] = mk4[k][0];
        s_tm[slot * kMaskWords + 1] = mk4[k][1];
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
      const uint32_t mk0 = lane < cnt ? s_tm[(pos + lane) * kMaskWords] : 0u;
      const uint32_t mk1 = lane < cnt ? s_tm[(pos + lane) * kMaskWords + 1] : 0u;
      const int h = __popc(mk0) + __popc(mk1);
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
        unsigned long long mm = (unsigned long long)mk0 | ((unsigned long long)mk1 << 32);
        int r = 0;
        while (mm) {
          const int fbit = __ffsll((long long)mm) - 1;
          mm &= mm - 1ull;
          s_hf[prefix + r] = (uint32_t)lane | ((uint32_t)fbit << 22);
          s_hw[prefix + r] = w0 + r;
          ++r;
        }
      }
      wave_lds_sync();
      WT(1);
      // ---- lane-parallel over hits: projection, a = 1/(w+1), b = w*a          clipfusion.py:647-659, :716-717
      for (int j0 = 0; j0 < htot; j0 += 64) {
        const int j = j0 + lane;
        const uint32_t hf = j < htot ? s_hf[j] : 0u;
        const uint32_t n = (uint32_t)__shfl((int)n_l, (int)(hf & 63u));
        if (j < htot) {
          const int fb = (int)(hf >> 22);
          int ix, iy, iz;
          voxel_coords(v, n, ix, iy, iz);
          const Cam cam = s_cam[fb];
          const Proj p = project(cam, v.ax[ix], v.ay[iy], v.az[iz]);
          const int wi = s_hw[j];
          const float a = 1.0f / (float)(wi + 1);
          s_ha[j] = a;
          s_hb[j] = (float)wi * a;
          s_hgx[j] = p.gx;
          s_hgy[j] = p.gy;
          // the hit's map cell: (y0, x0) of its four taps; every cell wholly outside the map is one cell
          const Bilin bw = bilinear_setup(p.gx, p.gy, half_px, half_py);
          const int cx = min(max(bw.x0, -2), wa.npx) + 2, cy = min(max(bw.y0, -2), wa.npy) + 2;
          s_hf[j] = hf | ((uint32_t)((cy << 8) | cx) << 6);
        }
      }
      wave_lds_sync();
      WT(2);
      // ---- lane-parallel over voxels: rgb / weight / label side, hit by hit in frame order
      if (active) {
        for (int r = 0; r < h; ++r) {
          const int j = prefix + r;
          const int fb = (int)(s_hf[j] >> 22);
          kf.rgb = s_rgb[fb];
          kf.label_map = s_lab[fb];
          const Cam cam = s_cam[fb];
          fuse_scalars_lane(v, kf, cam, n_l, s_hgx[j], s_hgy[j], w0 + r, s_ha[j], s_hb[j], stats);
        }
      }
      WT(3);
      // ---- rows: sub-chunks of <= SR rows and <= 64 hits
      int i0 = 0;
      while (i0 < m) {
        const int pbase = __builtin_amdgcn_readlane(prefix, i0);
        const unsigned long long okm = __ballot(lane >= i0 && lane < m && lane < i0 + SR && (incl - pbase) <= 64);
        const int nrows = __popcll(okm);                                         // >= 1 (a voxel has <= 64 hits)
        const int nh = __builtin_amdgcn_readlane(incl, i0 + nrows - 1) - pbase;  // 1..64
        // hit l of the sub-chunk (staging entry pbase + l) lives in lane l: its row, a, b and tap weights
        const bool hit = lane < nh;
        const uint32_t hfl = hit ? s_hf[pbase + lane] : 0xffffffffu;
        const uint32_t key = hfl >> 6;  // frame << 16 | cell
        WinHit rec;
        rec.row = (int)(hfl & 63u) - i0;
        rec.a = hit ? s_ha[pbase + lane] : 0.0f;
        rec.b = hit ? s_hb[pbase + lane] : 0.0f;
        {
          const Bilin w = bilinear_setup(hit ? s_hgx[pbase + lane] : 0.0f, hit ? s_hgy[pbase + lane] : 0.0f, half_px, half_py);
          rec.nw = w.nw; rec.ne = w.ne; rec.sw = w.sw; rec.se = w.se;
        }
        // (the staging reads above come BEFORE the LDS-DMA below: the compiler drains vmcnt ahead of any LDS
        //  read that follows an LDS-DMA, which would expose the rows' whole latency right here)
        // the rows, global -> LDS (one LDS-DMA moves a wave's 64 x 16 B = one 1 KiB piece of a row)
        unsigned long long fmask = 0;  // frames with a hit in this sub-chunk
        WinRaw<SR, UPL> raw;
#pragma unroll
        for (int r = 0; r < SR; ++r) {
          if (r < nrows) {
            const int64_t row = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)n_l, i0 + r) * DV;
            fmask |= (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mk0, i0 + r) |
                     ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mk1, i0 + r) << 32);
            if (BF16) {
#pragma unroll
              for (int k = 0; k < UPL; ++k) {
                const float4 t = ld_stream(featb + (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)n_l, i0 + r) * (DV / 2) +
                                           lane + k * 64);
                raw.u[r * UPL + k] = make_uint4(__builtin_bit_cast(uint32_t, t.x), __builtin_bit_cast(uint32_t, t.y),
                                                __builtin_bit_cast(uint32_t, t.z), __builtin_bit_cast(uint32_t, t.w));
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
        raw.nrows = nrows;
        // groups = hits of one frame in one map cell, frames ascending (a row's hits stay in frame order);
        // group g is kept in lane g: its key and the lane mask of its members
        uint32_t gk = 0, gm_lo = 0, gm_hi = 0;
        int G = 0;
        while (fmask) {
          const uint32_t f = (uint32_t)__ffsll((long long)fmask) - 1u;
          fmask &= fmask - 1ull;
          unsigned long long rem = __ballot(hit && (key >> 16) == f);
          while (rem) {
            const int l0 = __ffsll((long long)rem) - 1;
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, l0);
            const unsigned long long mm = __ballot(hit && key == k0);
            rem &= ~mm;
            if (lane == G) {
              gk = k0;
              gm_lo = (uint32_t)mm;
              gm_hi = (uint32_t)(mm >> 32);
            }
            ++G;
          }
        }
        WT(4);
        const WinCtx<CPL> cx{imgs, img_vecs, DV, wa.npx, wa.npy, zero_row, lane, rows};
        for (int g0 = 0; g0 < G; g0 += P) {
          const int nb = min(P, G - g0);
          switch (nb) {
            case 1: win_batch<1, CPL, SUM, BF16, SR>(cx, g0, gk, gm_lo, gm_hi, rec, raw); break;
            case 2: win_batch<(P >= 2 ? 2 : 1), CPL, SUM, BF16, SR>(cx, g0, gk, gm_lo, gm_hi, rec, raw); break;
            case 3: win_batch<(P >= 3 ? 3 : 1), CPL, SUM, BF16, SR>(cx, g0, gk, gm_lo, gm_hi, rec, raw); break;
            case 4: win_batch<(P >= 4 ? 4 : 1), CPL, SUM, BF16, SR>(cx, g0, gk, gm_lo, gm_hi, rec, raw); break;
            case 5: win_batch<(P >= 5 ? 5 : 1), CPL, SUM, BF16, SR>(cx, g0, gk, gm_lo, gm_hi, rec, raw); break;
            default: win_batch<(P >= 6 ? 6 : 1), CPL, SUM, BF16, SR>(cx, g0, gk, gm_lo, gm_hi, rec, raw); break;
          }
        }
        // nothing is outstanding here (every tap load has been consumed); the explicit wait only tells the
        // compiler's wait-count pass so, or it would drain vmcnt -- i.e. the previous row's store -- before
        // each row's LDS read below
        __builtin_amdgcn_s_waitcnt(0x0F70);
        wave_lds_sync();
        WT(6);
#pragma unroll
        for (int r = 0; r < SR; ++r) {
          if (r < nrows) {
            const int64_t row = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)n_l, i0 + r) * DV;
            if (BF16) {
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
        wave_lds_sync();  // the row buffer is rewritten by the next sub-chunk
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

using FuseFn = void (*)(KVol, KFrame, const unsigned long long*, const uint32_t*, uint32_t, const float*,
                        const float*, int, unsigned long long*, const unsigned long long*, unsigned long long);

template <int VEC, int CPL, int U>
FuseFn pick_lds(bool lds) {
  return lds ? fuse_kernel<VEC, CPL, U, true> : fuse_kernel<VEC, CPL, U, false>;
}

template <int CPL, int R, bool SUM, bool BF16>
FuseFn pick_rows3(bool g64, bool lds) {
  if (g64)
    return lds ? fuse_rows_kernel<CPL, R, true, true, SUM, BF16> : fuse_rows_kernel<CPL, R, true, false, SUM, BF16>;
  return lds ? fuse_rows_kernel<CPL, R, false, true, SUM, BF16> : fuse_rows_kernel<CPL, R, false, false, SUM, BF16>;
}
template <int CPL, int R>
FuseFn pick_rows(bool g64, bool lds, bool sum, bool bf16) {
  if (bf16) return sum ? pick_rows3<CPL, R, true, true>(g64, lds) : pick_rows3<CPL, R, false, true>(g64, lds);
  return sum ? pick_rows3<CPL, R, true, false>(g64, lds) : pick_rows3<CPL, R, false, false>(g64, lds);
}

int launch_fuse(const KVol& kv, const KFrame& kf, const WsLayout& w, const float* feat_map,
                const unsigned long long* counts, const unsigned char* half, unsigned long long* stats,
                const unsigned long long* sweep_done, unsigned long long sweep_target, bool shared_cus,
                hipStream_t s) {
  const int D = kv.D, P = kf.npy * kf.npx;
  const bool bf16 = kv.bf16 != 0;
  const int VEC = (D % 4 == 0) ? 4 : 1;
  const int DV = D / VEC;
  const int units = bf16 ? D / 8 : DV;  // 16-byte row units (vector path)
  int g_log2 = 0;
  while ((1 << g_log2) < units && g_log2 < 6) ++g_log2;
  const int G = 1 << g_log2;
  const int cpl = (units + G - 1) / G;
  const bool lds = w.lds_map;
  const bool sum = kv.accum == SAF_SUM;
  FuseFn fn;
  if (bf16) {
    if (D % 8 != 0 || cpl > 4) return fail(SAF_E_UNSUPPORTED, "bf16 volume needs feat_dim %% 8 == 0 and <= 2048");
    const bool g64 = G == 64;
    switch (cpl) {
      case 1: fn = pick_rows<1, 8>(g64, lds, sum, true); break;
      case 2: fn = pick_rows<2, 4>(g64, lds, sum, true); break;
      case 3: fn = pick_rows<3, 2>(g64, lds, sum, true); break;
      default: fn = pick_rows<4, 2>(g64, lds, sum, true); break;
    }
  } else if (VEC == 4 && cpl >= 1 && cpl <= 4) {
    const bool g64 = G == 64;
    switch (cpl) {
      case 1: fn = pick_rows<1, 4>(g64, lds, sum, false); break;
      case 2: fn = pick_rows<2, 4>(g64, lds, sum, false); break;
      case 3: fn = pick_rows<3, 2>(g64, lds, sum, false); break;
      default: fn = pick_rows<4, 2>(g64, lds, sum, false); break;
    }
  } else if (VEC == 4) {
    fn = pick_lds<4, 0, 1>(lds);
  } else {
    fn = (cpl == 1) ? pick_lds<1, 1, 4>(lds) : pick_lds<1, 0, 1>(lds);
  }
  const size_t shmem = lds ? lds_map_bytes(D, P) : 0;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute(LDS=%zu): %s", shmem, hipGetErrorString(e));
  }
  // 512-thread workgroups (8 waves at <= 128 VGPRs = half of each SIMD's register file, one LDS
  // image of the map).  Alone on the chip two of them share a CU; in the saf_fuse_frames pipeline
  // ONE per CU is launched so that the other half of the registers, ~80 KB of LDS and 16 wave slots
  // stay free for four sweep blocks of the next frame: both kernels are then resident on every CU
  // and neither can starve the other at dispatch.
  static const int grid_env = getenv("SAF_FUSE_GRID") ? atoi(getenv("SAF_FUSE_GRID")) : 0;
  // one workgroup per CU is enough in-flight rows to saturate HBM (alone: 276 us with 208..256
  // workgroups, 286 us with 512); in the pipeline 13/16 of the CUs (208 on MI355X) measured best
  int grid = grid_env > 0 ? grid_env : (shared_cus ? (device_cus() * 13) / 16 : device_cus());
  grid = ((grid + kNumLists - 1) / kNumLists) * kNumLists;
  const uint32_t* lists = reinterpret_cast<const uint32_t*>(half + w.lists_off);
  const float* map_t = reinterpret_cast<const float*>(half + w.map_off);
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kFuseThreads), shmem, s, kv, kf, counts, lists, w.list_cap, feat_map, map_t,
                     g_log2, stats, sweep_done, sweep_target);
  return check_launch("fuse_kernel");
}

int make_kvol(const saf_volume* vol, KVol* kv) {
  if (!vol) return fail(SAF_E_INVALID, "volume is NULL");
  if (vol->nx <= 0 || vol->ny <= 0 || vol->nz <= 0 || vol->feat_dim <= 0)
    return fail(SAF_E_INVALID, "bad volume shape %dx%dx%d D=%d", vol->nx, vol->ny, vol->nz, vol->feat_dim);
  const int64_t N = n_voxels(vol);
  if (N >= (1ll << 31)) return fail(SAF_E_UNSUPPORTED, "volumes of 2^31 voxels or more are not supported");
  if (vol->feat_dtype != SAF_F32 && vol->feat_dtype != SAF_BF16)
    return fail(SAF_E_UNSUPPORTED, "feat_dtype %d: SAF_F32 and SAF_BF16 are implemented", vol->feat_dtype);
  if (vol->feat_dtype == SAF_BF16 && (vol->feat_dim % 8 != 0 || ((uintptr_t)vol->clip_feat & 15)))
    return fail(SAF_E_UNSUPPORTED, "a bf16 volume needs feat_dim %% 8 == 0 and a 16-byte aligned buffer");
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
  kv->bf16 = vol->feat_dtype == SAF_BF16;
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
  int stride;  // record only frames whose index within the call is a multiple of this
};

namespace saf {
namespace {

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

struct FrameJob {
  KFrame kf;
  WsLayout w;
  const float* feat_map;
};

int make_job(const KVol& kv, const saf_frame* frame, void* workspace, size_t workspace_bytes, FrameJob* job) {
  int rc = make_kframe(frame, &job->kf);
  if (rc) return rc;
  job->feat_map = frame->feat_map;
  job->w = ws_layout(kv.N, kv.D, job->kf.npy * job->kf.npx);
  if (!workspace || ((uintptr_t)workspace & 255)) return fail(SAF_E_INVALID, "workspace must be 256-byte aligned");
  if (workspace_bytes < job->w.total)
    return fail(SAF_E_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, job->w.total);
  return SAF_OK;
}

inline unsigned long long* counter_set(unsigned char* ws, int64_t frame_no) {
  return reinterpret_cast<unsigned long long*>(ws) + (size_t)(frame_no & (kCounterSets - 1)) * kNumLists;
}
inline unsigned long long* sweep_done_ptr(unsigned char* ws) {
  return reinterpret_cast<unsigned long long*>(ws + kSweepDoneOff);
}

// sweep (and, for maps too large for LDS, the map image) of frame `frame_no` into list buffer `buf`
int launch_classify(const KVol& kv, const FrameJob& job, unsigned char* ws, unsigned char* buf, int64_t frame_no,
                    saf_profiler* prof, hipStream_t s) {
  const WsLayout& w = job.w;
  uint32_t* lists = reinterpret_cast<uint32_t*>(buf + w.lists_off);
  int rc;
  if (!w.lds_map) {
    const int P = job.kf.npy * job.kf.npx;
    const int items = kv.D * map_ppad(P);
    ScopedPair t(prof, 0, frame_no, s);
    hipLaunchKernelGGL(prep_kernel, dim3((items + 255) / 256), dim3(256), 0, s, job.feat_map,
                       reinterpret_cast<float*>(buf + w.map_off), kv.D, P, (kv.D % 4 == 0) ? 4 : 1);
    if ((rc = check_launch("prep_kernel"))) return rc;
  }
  ScopedPair t(prof, 1, frame_no, s);
  hipLaunchKernelGGL(sweep_kernel, dim3(w.n_blocks), dim3(kSweepThreads), 0, s, kv, job.kf,
                     counter_set(ws, frame_no), counter_set(ws, frame_no + 1), lists, w.list_cap, sweep_done_ptr(ws));
  return check_launch("sweep_kernel");
}

int launch_rows(const KVol& kv, const FrameJob& job, unsigned char* ws, unsigned char* buf, int64_t frame_no,
                uint64_t* stats, bool shared_cus, saf_profiler* prof, hipStream_t s) {
  ScopedPair t(prof, 2, frame_no, s);
  // the fuse kernel starts once the sweeps of frames 0..frame_no have published all their blocks
  const unsigned long long target = (unsigned long long)(frame_no + 1) * job.w.n_blocks;
  return launch_fuse(kv, job.kf, job.w, job.feat_map, counter_set(ws, frame_no), buf,
                     reinterpret_cast<unsigned long long*>(stats), sweep_done_ptr(ws), target, shared_cus, s);
}

#define SAF_HIP_TRY(call)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) { rc = fail(SAF_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); goto done; } \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Windowed (voxel-major) path of saf_fuse_frames: see fuse_window_kernel.
// Workspace: the common header (piece counter), the kWin pixel-major map images of one window, and the
// window's frame bitmasks (kMaskWords words per voxel).
// ---------------------------------------------------------------------------------------------
struct WinLayout {
  size_t img_bytes, maps_bytes, mask_bytes, total;
  uint32_t mask_plane;
};
WinLayout win_layout(int64_t n_vox, int D, int P) {
  WinLayout w;
  w.img_bytes = ((size_t)D * (P + 1) * sizeof(float) + 255) & ~(size_t)255;
  w.maps_bytes = (size_t)kWin * w.img_bytes;
  w.mask_plane = (uint32_t)((n_vox + 63) & ~(int64_t)63);  // words per mask plane (16-byte aligned planes)
  w.mask_bytes = ((size_t)w.mask_plane * sizeof(uint32_t) * kMaskWords + 255) & ~(size_t)255;
  w.total = kHdrBytes + w.maps_bytes + w.mask_bytes;
  return w;
}

using WinFn = void (*)(KVol, WinArgs, const float*, int, unsigned long long*, unsigned int*, const uint32_t*, uint32_t);
template <int CPL>
WinFn pick_win(bool sum, bool bf16) {
  if (bf16) {
    if (CPL % 2 != 0) return nullptr;
    constexpr int C2 = CPL % 2 == 0 ? CPL : 2;
    return sum ? fuse_window_kernel<C2, true, true> : fuse_window_kernel<C2, false, true>;
  }
  return sum ? fuse_window_kernel<CPL, true, false> : fuse_window_kernel<CPL, false, false>;
}

// Shapes the windowed path takes; everything else runs the per-frame pipeline.
bool window_ok(const KVol& kv, const saf_frame* frames, int32_t n_frames, size_t workspace_bytes) {
  static const bool enabled = !(getenv("SAF_WINDOW") && getenv("SAF_WINDOW")[0] == '0');
  if (!enabled || n_frames < kWinMinFrames || kv.D % 256 != 0 || kv.D > 1024) return false;
  // bf16 volumes: implemented and bit-identical, but no faster than the per-frame pipeline (rows are half
  // the bytes, the map taps are not: 3527 vs 3685 frames/s with labels, 4310 vs 4055 on the coherent
  // scene) -- opt-in with SAF_WINDOW_BF16=1
  static const bool bf16_on = getenv("SAF_WINDOW_BF16") && getenv("SAF_WINDOW_BF16")[0] == '1';
  if (kv.bf16 && (!bf16_on || kv.D % 512 != 0)) return false;  // a lane moves 8 bf16 channels: 512 per wave
  const saf_frame& f0 = frames[0];
  for (int32_t i = 0; i < n_frames; ++i) {
    const saf_frame& f = frames[i];
    if (f.height != f0.height || f.width != f0.width || f.npy != f0.npy || f.npx != f0.npx ||
        f.rgb_bilinear != f0.rgb_bilinear || (f.label_map == nullptr) != (f0.label_map == nullptr))
      return false;
  }
  if (f0.npx + 3 > 255 || f0.npy + 3 > 255) return false;  // a hit's map cell travels as two bytes
  return workspace_bytes >= win_layout(kv.N, kv.D, f0.npy * f0.npx).total;
}

int fuse_many_windowed(const KVol& kv, const saf_frame* frames, int32_t n_frames, void* workspace, uint64_t* stats,
                       saf_profiler* prof, hipStream_t s) {
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  int rc = SAF_OK;
  KFrame kf0;
  if ((rc = make_kframe(&frames[0], &kf0))) return rc;
  for (int32_t i = 1; i < n_frames; ++i) {
    KFrame t;
    if ((rc = make_kframe(&frames[i], &t))) return rc;
  }
  const int P = kf0.npy * kf0.npx;
  const WinLayout wl = win_layout(kv.N, kv.D, P);
  const bool sum = kv.accum == SAF_SUM;
  const int img_vecs = (int)(wl.img_bytes / sizeof(float4));
  const int prep_blocks = (kv.D * (P + 1) + 255) / 256;
  WinFn fn;
  size_t win_lds;
  switch (kv.D / 256) {
    case 1: fn = pick_win<1>(sum, kv.bf16 != 0); win_lds = WinCfg<1>::total; break;
    case 2: fn = pick_win<2>(sum, kv.bf16 != 0); win_lds = WinCfg<2>::total; break;
    case 3: fn = pick_win<3>(sum, kv.bf16 != 0); win_lds = WinCfg<3>::total; break;
    default: fn = pick_win<4>(sum, kv.bf16 != 0); win_lds = WinCfg<4>::total; break;
  }
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)win_lds);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute(LDS=%zu): %s", win_lds, hipGetErrorString(e));
  }
  float* maps = reinterpret_cast<float*>(ws + kHdrBytes);
  uint32_t* masks = reinterpret_cast<uint32_t*>(ws + kHdrBytes + wl.maps_bytes);
  unsigned int* piece_ctr = reinterpret_cast<unsigned int*>(ws);
  static const int wgs_env = getenv("SAF_WIN_WGS") ? atoi(getenv("SAF_WIN_WGS")) : 0;
  uint32_t grid = (uint32_t)device_cus() * (wgs_env > 0 ? wgs_env : 2);
  const uint32_t n_pieces = (uint32_t)(((int64_t)kv.N + kPiece - 1) / kPiece);
  const uint32_t n_wgs = (n_pieces + kWinWaves - 1) / kWinWaves;
  if (grid > n_wgs) grid = n_wgs;
  // Everything is ordered on the caller's stream: classification, map images, row kernel, window after
  // window.  Running the classification of window w+1 beside the row kernel of window w (second stream)
  // was measured: the pair costs the sum of the two either way (both are limited by the memory system),
  // and with 64-frame windows the row kernel's 156 KB of LDS per CU leave no room for it.
  static const int tile_env = getenv("SAF_WIN_TILE") ? atoi(getenv("SAF_WIN_TILE")) : -1;
  int tile = tile_env >= 0 ? tile_env : 32;
  {
    const int64_t plane = (int64_t)kv.ny * kv.nz;
    const int64_t ppx = plane / kPiece;
    if (plane % kPiece != 0) tile = 0;
    while (tile >= 8 && (ppx % tile != 0 || kv.nx % tile != 0)) tile >>= 1;
    if (tile < 8) tile = 0;  // linear order
  }
  const int n_win = (n_frames + kWin - 1) / kWin;
  for (int w = 0; w < n_win; ++w) {
    const int f0 = w * kWin;
    const int F = n_frames - f0 < kWin ? n_frames - f0 : kWin;
    WinArgs wa;
    wa.F = F; wa.H = kf0.H; wa.W = kf0.W; wa.npy = kf0.npy; wa.npx = kf0.npx; wa.rgb_bilinear = kf0.rgb_bilinear;
    PrepArgs pa;
    for (int k = 0; k < kWin; ++k) {
      const saf_frame& fr = frames[f0 + (k < F ? k : 0)];
      wa.depth[k] = fr.depth; wa.rgb[k] = fr.rgb; wa.pose[k] = fr.pose; wa.K[k] = fr.K; wa.label_map[k] = fr.label_map;
      pa.feat_map[k] = fr.feat_map;
    }
    for (int fb = 0; fb < F; fb += 32) {
      const int fe = fb + 32 < F ? fb + 32 : F;
      uint32_t* plane = masks + (size_t)(fb / 32) * wl.mask_plane;
      ScopedPair t(prof, 1, f0 + fb, s);
      if (sum)
        hipLaunchKernelGGL(classify_window_kernel<true>, dim3(n_wgs), dim3(256), 0, s, kv, wa, fb, fe, tile, plane,
                           reinterpret_cast<unsigned long long*>(stats));
      else
        hipLaunchKernelGGL(classify_window_kernel<false>, dim3(n_wgs), dim3(256), 0, s, kv, wa, fb, fe, tile, plane,
                           reinterpret_cast<unsigned long long*>(stats));
    }
    if ((rc = check_launch("classify_window_kernel"))) return rc;
    if (hipMemsetAsync(piece_ctr, 0, sizeof(unsigned int), s) != hipSuccess)
      return fail(SAF_E_HIP, "hipMemsetAsync(piece counter)");
    {
      ScopedPair t(prof, 0, f0, s);
      hipLaunchKernelGGL(prep_rows_kernel, dim3(prep_blocks, F), dim3(256), 0, s, pa, maps,
                         (int)(wl.img_bytes / sizeof(float)), kv.D, P);
    }
    if ((rc = check_launch("prep_rows_kernel"))) return rc;
    {
      ScopedPair t(prof, 2, f0, s);
      hipLaunchKernelGGL(fn, dim3(grid), dim3(kWinThreads), win_lds, s, kv, wa, maps, img_vecs,
                         reinterpret_cast<unsigned long long*>(stats), piece_ctr, masks, wl.mask_plane);
    }
    if ((rc = check_launch("fuse_window_kernel"))) return rc;
  }
#ifdef SAF_WIN_TIMING
  {
    (void)hipStreamSynchronize(s);
    unsigned long long t[16];
    if (hipMemcpyFromSymbol(t, HIP_SYMBOL(g_win_t), sizeof(t)) == hipSuccess) {
      unsigned long long tot = 0;
      for (int k = 0; k < 8; ++k) tot += t[k];
      fprintf(stderr, "[win timing] masks %.1f%% expand %.1f%% project %.1f%% scalars %.1f%% records+groups+row issue %.1f%% - %.1f%% tap batches %.1f%% row store %.1f%% (total %.3g wave-cycles)\n",
              100.0 * t[0] / tot, 100.0 * t[1] / tot, 100.0 * t[2] / tot, 100.0 * t[3] / tot, 100.0 * t[4] / tot,
              100.0 * t[5] / tot, 100.0 * t[6] / tot, 100.0 * t[7] / tot, (double)tot);
      memset(t, 0, sizeof(t));
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_win_t), t, sizeof(t));
    }
  }
#endif
  return rc;
}

// Frames in order.  A single frame runs sweep -> fuse on the caller's stream.  For two or more
// frames the sweeps (VALU-bound; they touch only the TSDF buffers and their own list buffer) are
// queued back to back on an auxiliary stream and may run up to kListBuffers frames ahead, while the
// caller's stream carries the HBM-bound fuse kernels back to back:
//   sweep(i) -> fuse(i)        device-side: fuse workgroups poll the sweep-completion counter
//   fuse(i)  -> sweep(i + 4)   reuse of list buffer i & 3: an event recorded on the caller's
//                              stream after every second fuse kernel (few packets between them)
// Everything is ordered after what the caller already queued on `s` (fork event), and complete,
// as far as `s` is concerned, when the last fuse kernel is (it has waited for every sweep).
int fuse_many(const KVol& kv, const saf_frame* frames, int32_t n_frames, void* workspace, size_t workspace_bytes,
              uint64_t* stats, saf_profiler* prof, hipStream_t s) {
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  int rc = SAF_OK;
  if (!workspace || ((uintptr_t)workspace & 255)) return fail(SAF_E_INVALID, "workspace must be 256-byte aligned");
  if (window_ok(kv, frames, n_frames, workspace_bytes))
    return fuse_many_windowed(kv, frames, n_frames, workspace, stats, prof, s);
  // counters and the completion counter start at zero; afterwards every sweep zeroes its successor's set
  if (hipMemsetAsync(ws, 0, kHdrBytes, s) != hipSuccess) return fail(SAF_E_HIP, "hipMemsetAsync(workspace header)");
  // SAF_PIPELINE=0 keeps everything on the caller's stream (debugging / per-kernel timing)
  static const bool pipeline = !(getenv("SAF_PIPELINE") && getenv("SAF_PIPELINE")[0] == '0');
  if (n_frames == 1 || !pipeline) {
    for (int32_t i = 0; i < n_frames; ++i) {
      FrameJob job;
      if ((rc = make_job(kv, &frames[i], workspace, workspace_bytes, &job))) return rc;
      unsigned char* buf = ws + kHdrBytes;
      if ((rc = launch_classify(kv, job, ws, buf, i, prof, s))) return rc;
      if ((rc = launch_rows(kv, job, ws, buf, i, stats, false, prof, s))) return rc;
    }
    return SAF_OK;
  }
  constexpr int kEvRing = 4;
  hipStream_t aux = nullptr;
  hipEvent_t fork = nullptr, fused[kEvRing] = {nullptr, nullptr, nullptr, nullptr};
  SAF_HIP_TRY(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
  SAF_HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  for (int b = 0; b < kEvRing; ++b) SAF_HIP_TRY(hipEventCreateWithFlags(&fused[b], hipEventDisableTiming));
  SAF_HIP_TRY(hipEventRecord(fork, s));
  SAF_HIP_TRY(hipStreamWaitEvent(aux, fork, 0));
  for (int32_t i = 0; i < n_frames; ++i) {
    FrameJob job;
    if ((rc = make_job(kv, &frames[i], workspace, workspace_bytes, &job))) goto done;
    unsigned char* buf = ws + kHdrBytes + (size_t)(i % kListBuffers) * job.w.half;
    if (i >= kListBuffers) {
      // buffer i & 3 was last read by fuse(i - 4); events exist after the odd-numbered fuse kernels:
      // the first one at or after i - 4 is j = (i - 4) | 1  (<= i - 3, already recorded)
      const int32_t j = (i - kListBuffers) | 1;
      SAF_HIP_TRY(hipStreamWaitEvent(aux, fused[(j >> 1) % kEvRing], 0));
    }
    if ((rc = launch_classify(kv, job, ws, buf, i, prof, aux))) goto done;
    if ((rc = launch_rows(kv, job, ws, buf, i, stats, true, prof, s))) goto done;
    if (i & 1) SAF_HIP_TRY(hipEventRecord(fused[(i >> 1) % kEvRing], s));
  }
done:
  // destroying a stream / event with work in flight is deferred by the runtime until it drains
  if (fork) (void)hipEventDestroy(fork);
  for (int b = 0; b < kEvRing; ++b)
    if (fused[b]) (void)hipEventDestroy(fused[b]);
  if (aux) (void)hipStreamDestroy(aux);
  return rc;
}

}  // namespace
}  // namespace saf

using namespace saf;

extern "C" {

const char* saf_last_error(void) { return err_buf(); }
int saf_abi_version(void) { return SAF_ABI_VERSION; }

size_t saf_fuse_workspace_bytes(int64_t n_vox, int32_t feat_dim, int32_t npy, int32_t npx) {
  if (n_vox <= 0 || feat_dim <= 0 || npy <= 0 || npx <= 0) return 0;
  const size_t a = ws_layout(n_vox, feat_dim, npy * npx).total, b = win_layout(n_vox, feat_dim, npy * npx).total;
  return a > b ? a : b;
}

int saf_fuse_frames_profiled(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, void* workspace,
                             size_t workspace_bytes, uint64_t* stats, saf_profiler* profiler, void* stream) {
  KVol kv;
  int rc = make_kvol(vol, &kv);
  if (rc) return rc;
  if (n_frames < 0 || (n_frames > 0 && !frames)) return fail(SAF_E_INVALID, "bad frame array");
  if (n_frames == 0) return SAF_OK;
  return fuse_many(kv, frames, n_frames, workspace, workspace_bytes, stats, profiler, static_cast<hipStream_t>(stream));
}

int saf_fuse_frames(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, void* workspace,
                    size_t workspace_bytes, uint64_t* stats, void* stream) {
  return saf_fuse_frames_profiled(vol, frames, n_frames, workspace, workspace_bytes, stats, nullptr, stream);
}

int saf_fuse_frame(const saf_volume* vol, const saf_frame* frame, void* workspace, size_t workspace_bytes,
                   uint64_t* stats, void* stream) {
  if (!frame) return fail(SAF_E_INVALID, "frame is NULL");
  return saf_fuse_frames_profiled(vol, frame, 1, workspace, workspace_bytes, stats, nullptr, stream);
}

saf_profiler* saf_profiler_create(int32_t capacity_pairs) {
  if (capacity_pairs <= 0) return nullptr;
  saf_profiler* p = new saf_profiler;
  p->pairs = new saf_profiler::Pair[capacity_pairs];
  p->capacity = 0;
  p->used = 0;
  p->stride = 1;
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

void saf_profiler_set_stride(saf_profiler* p, int32_t stride) {
  if (p) p->stride = stride > 0 ? stride : 1;
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
