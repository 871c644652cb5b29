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
// (DESIGN.md §4).  Calls of 16 or more frames of one shape take the windowed path of saf_window.hip instead
// (bit-identical).  All arithmetic that selects voxels is shared with the oracle's restatement in
// spirit and checked bit-for-bit against it (saf_common.h); the device building blocks shared with the
// windowed path live in saf_fuse_dev.h.
#include <mutex>

#include "saf_window_dev.h"

namespace saf {

char* err_buf() {
  static thread_local char buf[kErrLen] = "";
  return buf;
}

namespace {

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

__device__ __forceinline__ unsigned long long sweep_blocks_done(const unsigned long long* __restrict__ sweep_done) {
  unsigned long long n = 0;
#pragma unroll
  for (int k = 0; k < kDoneShards; ++k) n += __hip_atomic_load(&sweep_done[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return n;
}
// Asynchronous error latch (host-mapped pinned word, one per device): a fuse workgroup that gives up waiting
// for its frame's sweep sets it; every later saf_fuse_* / saf_poll_async_error call on the host then fails with
// SAF_E_HIP instead of leaving a silently incomplete volume behind (stats[4] counts the workgroups as before).
__device__ int* g_async_latch = nullptr;

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
          if (g_async_latch) __hip_atomic_store(g_async_latch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
  const int grid_env = getenv("SAF_FUSE_GRID") ? atoi(getenv("SAF_FUSE_GRID")) : 0;
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

// The auxiliary stream and events of the two-stream pipeline are pooled per device (creating and destroying
// a stream and six events per call cost tens of microseconds on every short integrate()).  A call takes a set
// from the pool and returns it when it has queued its work: the next user queues behind it on the same stream,
// and an event re-recorded later does not disturb a wait that was queued earlier.
constexpr int kEvRing = 4;
struct PipeRes {
  int device;
  hipStream_t aux;
  hipEvent_t fork, join, fused[kEvRing], tiles;
  PipeRes* next;
};
std::mutex g_pipe_mu;
PipeRes* g_pipe_free = nullptr;

PipeRes* pipe_acquire() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pipe_mu);
    for (PipeRes** pp = &g_pipe_free; *pp; pp = &(*pp)->next) {
      if ((*pp)->device == dev) {
        PipeRes* r = *pp;
        *pp = r->next;
        return r;
      }
    }
  }
  PipeRes* r = new PipeRes();
  r->device = dev;
  r->next = nullptr;
  // SAF_CLS_PRIORITY=1 (read when a device's first pipeline is made; development): the classification stream at the device's
  // highest priority -- its launches run beside the row kernel AND, behind integrate(), beside the staging of later frames
  int lo_pri = 0, hi_pri = 0;
  const char* pe = getenv("SAF_CLS_PRIORITY");
  const bool pri = pe && atoi(pe) != 0 && hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri) == hipSuccess;
  bool ok = (pri ? hipStreamCreateWithPriority(&r->aux, hipStreamNonBlocking, hi_pri) : hipStreamCreateWithFlags(&r->aux, hipStreamNonBlocking)) == hipSuccess &&
            hipEventCreateWithFlags(&r->fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&r->join, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&r->tiles, hipEventDisableTiming) == hipSuccess;
  for (int b = 0; ok && b < kEvRing; ++b) ok = hipEventCreateWithFlags(&r->fused[b], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    delete r;  // (handles created so far are leaked: this only happens when the runtime is out of resources)
    return nullptr;
  }
  return r;
}
void pipe_release(PipeRes* r) {
  std::lock_guard<std::mutex> lk(g_pipe_mu);
  r->next = g_pipe_free;
  g_pipe_free = r;
}

// Host side of the asynchronous error latch (see g_async_latch): one mapped pinned word per device.
constexpr int kMaxDevices = 64;
int* g_latch_host[kMaxDevices] = {};
std::mutex g_latch_mu;

int ensure_latch() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return SAF_OK;  // no latch: stats[4] still counts
  std::lock_guard<std::mutex> lk(g_latch_mu);
  if (g_latch_host[dev]) return SAF_OK;
  int* h = nullptr;
  int* d = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&h), sizeof(int), hipHostMallocMapped) != hipSuccess) return SAF_OK;
  *h = 0;
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(g_async_latch), &d, sizeof(d)) != hipSuccess) {
    (void)hipHostFree(h);
    return SAF_OK;
  }
  g_latch_host[dev] = h;
  return SAF_OK;
}
int poll_latch() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return SAF_OK;
  int* h = g_latch_host[dev];
  if (h && *reinterpret_cast<volatile int*>(h) != 0) {
    // reported ONCE, then cleared: the call that dropped frames is what is wrong, not the device -- a stall that has passed
    // (a debugger stop, another tenant) must not disable fusion for the life of the process
    *reinterpret_cast<volatile int*>(h) = 0;
    return fail(SAF_E_HIP, "an earlier saf_fuse_frames call on this device dropped frames: fuse workgroups timed out "
                           "waiting for their frame's sweep (stats[4] of that volume); that volume is incomplete");
  }
  return SAF_OK;
}

#define SAF_HIP_TRY(call)                                                                  \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) { rc = fail(SAF_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); goto done; } \
  } while (0)

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
              uint64_t* stats, saf_profiler* prof, hipStream_t s, const WinSlabs* slabs = nullptr, bool recycled = false) {
  unsigned char* ws = static_cast<unsigned char*>(workspace);
  int rc = SAF_OK;
  if (!workspace || ((uintptr_t)workspace & 255)) return fail(SAF_E_INVALID, "workspace must be 256-byte aligned");
  if (recycled && !window_ok(kv, frames, n_frames, workspace_bytes)) {
    // the per-frame pipeline reads every row it updates: the rows of weight-0 voxels are zeroed first (of the slabs, if the
    // call names slabs: voxels outside them are not this call's)
    if (slabs && slabs->n > 0) {
      for (int k = 0; k < slabs->n; ++k) {
        if (slabs->x0[k] < 0 || slabs->nx[k] <= 0 || slabs->x0[k] + slabs->nx[k] > kv.nx) return fail(SAF_E_INVALID, "slab %d outside the volume", k);
        if ((rc = launch_clear_unwritten(slab_kvol(kv, slabs->x0[k], slabs->nx[k]), nullptr, 0, 0, s))) return rc;
      }
    } else if ((rc = launch_clear_unwritten(kv, nullptr, 0, 0, s))) {
      return rc;
    }
    recycled = false;
  }
  if (window_ok(kv, frames, n_frames, workspace_bytes)) {
    // SAF_WIN_OVERLAP=0: every kernel of the windowed path on the caller's stream (read per call: same-process A/Bs)
    // (a call of one window overlaps too: its first window is classified slab by slab beside its own row kernels)
    const char* ov_env = getenv("SAF_WIN_OVERLAP");
    PipeRes* pr = (ov_env && ov_env[0] == '0') ? nullptr : pipe_acquire();
    if (!pr) return fuse_many_windowed(kv, frames, n_frames, workspace, workspace_bytes, stats, prof, s, nullptr, slabs, recycled);
    WinOverlap ov;
    ov.aux = pr->aux; ov.fork = pr->fork; ov.join = pr->join;
    ov.cls_done[0] = pr->fused[0]; ov.cls_done[1] = pr->fused[1]; ov.fuse_done[0] = pr->fused[2]; ov.fuse_done[1] = pr->fused[3];
    ov.tiles = pr->tiles;
    rc = fuse_many_windowed(kv, frames, n_frames, workspace, workspace_bytes, stats, prof, s, &ov, slabs, recycled);
    pipe_release(pr);
    return rc;
  }
  if (slabs && slabs->n > 0) {  // the per-frame pipeline, slab after slab (every (slab, frame) pair counts as a frame in stats[2])
    for (int k = 0; k < slabs->n; ++k) {
      if (slabs->x0[k] < 0 || slabs->nx[k] <= 0 || slabs->x0[k] + slabs->nx[k] > kv.nx) return fail(SAF_E_INVALID, "slab %d outside the volume", k);
      if ((rc = fuse_many(slab_kvol(kv, slabs->x0[k], slabs->nx[k]), frames, n_frames, workspace, workspace_bytes, stats, prof, s))) return rc;
      if (slabs->done && slabs->done[k] && hipEventRecord(static_cast<hipEvent_t>(slabs->done[k]), s) != hipSuccess)
        return fail(SAF_E_HIP, "hipEventRecord(slab done)");
    }
    return SAF_OK;
  }
  // counters and the completion counter start at zero; afterwards every sweep zeroes its successor's set
  if (hipMemsetAsync(ws, 0, kHdrBytes, s) != hipSuccess) return fail(SAF_E_HIP, "hipMemsetAsync(workspace header)");
  // SAF_PIPELINE=0 keeps everything on the caller's stream (debugging / per-kernel timing)
  const bool pipeline = !(getenv("SAF_PIPELINE") && getenv("SAF_PIPELINE")[0] == '0');
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
  PipeRes* pr = pipe_acquire();
  if (!pr) return fail(SAF_E_HIP, "could not create the sweep stream / events of the per-frame pipeline");
  hipStream_t aux = pr->aux;
  SAF_HIP_TRY(hipEventRecord(pr->fork, s));
  SAF_HIP_TRY(hipStreamWaitEvent(aux, pr->fork, 0));
  for (int32_t i = 0; i < n_frames; ++i) {
    FrameJob job;
    if ((rc = make_job(kv, &frames[i], workspace, workspace_bytes, &job))) goto done;
    unsigned char* buf = ws + kHdrBytes + (size_t)(i % kListBuffers) * job.w.half;
    if (i >= kListBuffers) {
      // buffer i & 3 was last read by fuse(i - 4); events exist after the odd-numbered fuse kernels:
      // the first one at or after i - 4 is j = (i - 4) | 1  (<= i - 3, already recorded)
      const int32_t j = (i - kListBuffers) | 1;
      SAF_HIP_TRY(hipStreamWaitEvent(aux, pr->fused[(j >> 1) % kEvRing], 0));
    }
    if ((rc = launch_classify(kv, job, ws, buf, i, prof, aux))) goto done;
    if ((rc = launch_rows(kv, job, ws, buf, i, stats, true, prof, s))) goto done;
    if (i & 1) SAF_HIP_TRY(hipEventRecord(pr->fused[(i >> 1) % kEvRing], s));
  }
done:
  // join: the sweeps' TSDF stores become visible to later work on `s` only at the end of the sweep
  // kernels, which the last fuse kernel does not order -- the caller's stream waits for the aux stream
  // (also on the error paths: whatever was queued on aux must not outlive the call unordered)
  if (hipEventRecord(pr->join, aux) == hipSuccess) (void)hipStreamWaitEvent(s, pr->join, 0);
  pipe_release(pr);
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
  // without the volume's dtype: room for the brick form wherever SOME dtype of this width would take it
  const char* e = getenv("SAF_WIN_FORM");
  const bool rows_always = feat_dim % 512 == 0 && feat_dim <= 1024;  // f32 and bf16 volumes both take the row kernel
  const bool bricks = feat_dim % 64 == 0 && feat_dim <= 8192 && !(e && (e[0] == 'r' || e[0] == 's')) && (!rows_always || (e && e[0] == 'b'));
  const size_t a = ws_layout(n_vox, feat_dim, npy * npx).total, b = window_workspace_bytes(n_vox, feat_dim, npy * npx, bricks);
  return a > b ? a : b;
}

size_t saf_fuse_workspace_bytes_for(const saf_volume* vol, int32_t npy, int32_t npx) {
  KVol kv;
  if (!vol || npy <= 0 || npx <= 0 || make_kvol(vol, &kv)) return 0;
  const size_t a = ws_layout(kv.N, kv.D, npy * npx).total, b = window_workspace_bytes(kv.N, kv.D, npy * npx, brick_form_ok(kv));
  return a > b ? a : b;
}

size_t saf_fuse_workspace_bytes_for_frames(const saf_volume* vol, int32_t npy, int32_t npx, int32_t height, int32_t width) {
  KVol kv;
  if (!vol || npy <= 0 || npx <= 0 || height <= 0 || width <= 0 || make_kvol(vol, &kv)) return 0;
  const size_t a = ws_layout(kv.N, kv.D, npy * npx).total;
  const size_t b = window_workspace_bytes(kv.N, kv.D, npy * npx, brick_form_ok(kv), height, width, kv.labels != nullptr);
  return a > b ? a : b;
}

int saf_fuse_frames_profiled(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, void* workspace,
                             size_t workspace_bytes, uint64_t* stats, saf_profiler* profiler, void* stream) {
  KVol kv;
  int rc = make_kvol(vol, &kv);
  if (rc) return rc;
  if (n_frames < 0 || (n_frames > 0 && !frames)) return fail(SAF_E_INVALID, "bad frame array");
  if (n_frames == 0) return SAF_OK;
  if ((rc = poll_latch())) return rc;
  ensure_latch();
  return fuse_many(kv, frames, n_frames, workspace, workspace_bytes, stats, profiler, static_cast<hipStream_t>(stream));
}

int saf_fuse_frames_recycled(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, void* workspace,
                             size_t workspace_bytes, uint64_t* stats, saf_profiler* profiler, void* stream) {
  KVol kv;
  int rc = make_kvol(vol, &kv);
  if (rc) return rc;
  if (n_frames < 0 || (n_frames > 0 && !frames)) return fail(SAF_E_INVALID, "bad frame array");
  if ((rc = poll_latch())) return rc;
  ensure_latch();
  if (n_frames == 0) return launch_clear_unwritten(kv, nullptr, 0, 0, static_cast<hipStream_t>(stream));
  return fuse_many(kv, frames, n_frames, workspace, workspace_bytes, stats, profiler, static_cast<hipStream_t>(stream), nullptr, true);
}

int saf_fuse_frames_slabs(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, const int32_t* slab_x0,
                          const int32_t* slab_nx, int32_t n_slabs, void* const* slab_done_events, int32_t recycled,
                          void* workspace, size_t workspace_bytes, uint64_t* stats, saf_profiler* profiler, void* stream) {
  KVol kv;
  int rc = make_kvol(vol, &kv);
  if (rc) return rc;
  if (n_frames < 0 || (n_frames > 0 && !frames)) return fail(SAF_E_INVALID, "bad frame array");
  if (n_slabs <= 0 || !slab_x0 || !slab_nx) return fail(SAF_E_INVALID, "bad slab list");
  if ((rc = poll_latch())) return rc;
  ensure_latch();
  const WinSlabs sl{n_slabs, slab_x0, slab_nx, slab_done_events};
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_frames == 0) {  // nothing to fuse: a recycled volume's slabs are cleared, every event is recorded
    for (int k = 0; k < n_slabs; ++k) {
      if (slab_x0[k] < 0 || slab_nx[k] <= 0 || slab_x0[k] + slab_nx[k] > kv.nx) return fail(SAF_E_INVALID, "slab %d outside the volume", k);
      if (recycled && (rc = launch_clear_unwritten(slab_kvol(kv, slab_x0[k], slab_nx[k]), nullptr, 0, 0, s))) return rc;
      if (slab_done_events && slab_done_events[k] && hipEventRecord(static_cast<hipEvent_t>(slab_done_events[k]), s) != hipSuccess)
        return fail(SAF_E_HIP, "hipEventRecord(slab done)");
    }
    return SAF_OK;
  }
  return fuse_many(kv, frames, n_frames, workspace, workspace_bytes, stats, profiler, s, &sl, recycled != 0);
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

int saf_poll_async_error(void) { return poll_latch(); }

// ---- streaming sessions (round 6): consecutive windowed calls as ONE unit pipeline ----
struct saf_fuse_session {
  PipeRes* pr = nullptr;
  WinStream st;
  const void* feat = nullptr;  // identity of the volume and workspace the open window belongs to
  void* workspace = nullptr;
  size_t workspace_bytes = 0;
  uint64_t* stats = nullptr;
};

saf_fuse_session* saf_fuse_session_create(void) { return new saf_fuse_session; }

static bool session_overlap(saf_fuse_session* ss, WinOverlap* ov) {
  if (!ss->pr && !(ss->pr = pipe_acquire())) return false;
  PipeRes* pr = ss->pr;
  ov->aux = pr->aux; ov->fork = pr->fork; ov->join = pr->join;
  ov->cls_done[0] = pr->fused[0]; ov->cls_done[1] = pr->fused[1]; ov->fuse_done[0] = pr->fused[2]; ov->fuse_done[1] = pr->fused[3];
  ov->tiles = pr->tiles;
  return true;
}

int saf_fuse_session_ok(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, size_t workspace_bytes) {
  KVol kv;
  if (make_kvol(vol, &kv) || n_frames <= 0 || !frames) return -1;
  if (getenv("SAF_WIN_OVERLAP") && getenv("SAF_WIN_OVERLAP")[0] == '0') return 0;
  return stream_ok(kv, frames, n_frames, workspace_bytes) ? 1 : 0;
}

int saf_fuse_session_push(saf_fuse_session* ss, const saf_volume* vol, const saf_frame* frames, int32_t n_frames, void* workspace,
                          size_t workspace_bytes, uint64_t* stats, void* stream, void* ready_event, void* tile_stream) {
  if (!ss) return fail(SAF_E_INVALID, "session is NULL");
  KVol kv;
  int rc = make_kvol(vol, &kv);
  if (rc) return rc;
  if (n_frames <= 0 || !frames) return fail(SAF_E_INVALID, "bad frame array");
  if (!workspace || ((uintptr_t)workspace & 255)) return fail(SAF_E_INVALID, "workspace must be 256-byte aligned");
  if (!stream_ok(kv, frames, n_frames, workspace_bytes) || (getenv("SAF_WIN_OVERLAP") && getenv("SAF_WIN_OVERLAP")[0] == '0'))
    return fail(SAF_E_UNSUPPORTED, "a streaming session takes what the windowed row forms take on two streams (saf_fuse_session_ok)");
  if (ss->st.have_shape && (ss->feat != kv.feat || ss->workspace != workspace || ss->workspace_bytes != workspace_bytes || ss->stats != stats))
    return fail(SAF_E_INVALID, "a session continues on the same volume, workspace and counters: finish it first");
  if ((rc = poll_latch())) return rc;
  ensure_latch();
  WinOverlap ov;
  if (!session_overlap(ss, &ov)) return fail(SAF_E_HIP, "could not create the classification stream / events of a session");
  ss->feat = kv.feat; ss->workspace = workspace; ss->workspace_bytes = workspace_bytes; ss->stats = stats;
  return stream_push(kv, frames, n_frames, workspace, workspace_bytes, stats, static_cast<hipStream_t>(stream),
                     static_cast<hipEvent_t>(ready_event), static_cast<hipStream_t>(tile_stream), &ov, &ss->st);
}

int saf_fuse_session_prepare(saf_fuse_session* ss, const saf_volume* vol, const saf_frame* frames, int32_t n_frames, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (!ss) return fail(SAF_E_INVALID, "session is NULL");
  KVol kv;
  int rc = make_kvol(vol, &kv);
  if (rc) return rc;
  if (n_frames <= 0 || !frames) return fail(SAF_E_INVALID, "bad frame array");
  if (!workspace || ((uintptr_t)workspace & 255)) return fail(SAF_E_INVALID, "workspace must be 256-byte aligned");
  if (!stream_ok(kv, frames, n_frames, workspace_bytes)) return fail(SAF_E_UNSUPPORTED, "saf_fuse_session_prepare: not a shape a session takes");
  return stream_prepare(kv, frames, n_frames, workspace, workspace_bytes, static_cast<hipStream_t>(stream), &ss->st);
}

int saf_fuse_session_finish(saf_fuse_session* ss, void* stream) {
  if (!ss) return fail(SAF_E_INVALID, "session is NULL");
  int rc = SAF_OK;
  WinOverlap ov;
  if (ss->st.open && ss->st.filled > 0) {
    if (!session_overlap(ss, &ov)) return fail(SAF_E_HIP, "could not create the classification stream / events of a session");
    rc = stream_close(ss->workspace, ss->workspace_bytes, ss->stats, static_cast<hipStream_t>(stream), &ov, &ss->st, false);
  }
  ss->st = WinStream();  // the next push starts a new pipeline (its first window's classification alone on the chip)
  return rc;
}

int saf_fuse_session_abandon(saf_fuse_session* ss) {
  if (!ss) return fail(SAF_E_INVALID, "session is NULL");
  ss->st = WinStream();  // (the open window's classification ran -- TSDF and masks of a volume that is being discarded; no row kernel follows)
  return SAF_OK;
}

int saf_fuse_session_pending(const saf_fuse_session* ss) { return ss && ss->st.open ? ss->st.filled : 0; }

void saf_fuse_session_destroy(saf_fuse_session* ss) {
  if (!ss) return;
  if (ss->pr) pipe_release(ss->pr);  // (what is queued on its stream stays ordered: the next user queues behind it)
  delete ss;
}

int saf_fuse_path(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, size_t workspace_bytes) {
  KVol kv;
  if (make_kvol(vol, &kv) || n_frames <= 0 || !frames) return -1;
  return window_ok(kv, frames, n_frames, workspace_bytes) ? 1 : 0;
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
