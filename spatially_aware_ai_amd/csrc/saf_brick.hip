// saf_brick.hip -- the brick form of the windowed path's row kernel (clipfusion.py:699-721,
// clip_seem_fusion.py:752-822 for a window of up to 128 frames at once).
//
// The frame-ordered row kernel (saf_window.hip) pulls the four map rows of EVERY hit from L2 through the CU's L1:
// 8 KB per hit, 4.6 x the kernel's HBM bytes, and it runs at the L2's gather rate.  The taps of a hit depend only on
// (frame, map cell), and a compact brick of voxels sees one or two cells of a frame: here a workgroup owns a brick of
// 4 x 4 x 4 voxels, keeps an accumulator for ALL its touched rows in LDS -- a slab of up to 256 channels at a time -- and
// walks the brick's hits grouped by (frame, map cell): a group's four map rows are loaded ONCE per slab and blended
// into every member row.
//
//   per brick     the voxels' frame masks (left by the classification) -> hits in frame-major order (LDS bit matrix
//                 frame x voxel built with LDS atomics, prefix sums over frames); per hit: projection, map cell,
//                 bilinear weights, rgb sample, label count; per voxel, in frame order: the rgb running mean and the
//                 weight (exactly the per-frame arithmetic); hit records sorted by (frame, cell) in windows of 64 and
//                 a table of the groups (tap offsets, first hit) built once;
//   per slab      four WALKER waves take batches of groups round-robin: a lane owns CPL consecutive channels, the tap rows
//                 of a batch are requested one batch ahead (two register buffers), every hit's sample sum(w_t tap_t)
//                 goes into the LDS accumulator; four MOVER waves hold the rows' old pieces in registers (requested one
//                 slab ahead) and after the walk write old * b + acc * a and clear the accumulator.  Only the movers have
//                 row traffic in flight: a wave's vector-memory operations retire in order, so a walker's tap loads never
//                 queue behind row loads and stores.
//
// Arithmetic: a row's new value is (w0 old + sum of its samples) / (w0 + k) -- the running mean of clipfusion.py:715-721
// with the k updates of the window folded into one (order-free within fp32 rounding); weights, rgb,
// labels and which voxels are touched are exactly those of the frame-ordered kernels.
//
// The sum of a row's samples is kept in 32-bit FIXED POINT.  gfx950's LDS adds floats atomically at about three lanes
// per clock (ds_add_f32: 193 cycles per wave instruction, tools/lds_atomic_bench.hip -- the first form of this kernel
// spent 70 % of its time there) and integers at the LDS's write rate (ds_add_u32: one wave instruction per 4 cycles and CU,
// tools/valu_probe.hip).  A sample is a convex combination of
// map values, so with M = the largest magnitude of the slab's channels in the window's maps (chan_max_kernel) and k_max =
// the most hits any row of the brick takes in this round, |sum| <= k_max M: every sample is scaled by the power of two
// 2^(30 - ceil(log2 k_max) - ceil(log2 M)) (exact; folded into the bilinear weights), rounded to the nearest integer (the
// only error: half a unit of M 2^-28 at k_max = 4, of M 2^-23 at 128 -- no more than the rounding of an fp32 running mean
// of values near M) and added with ds_add_u32.  Integer adds commute: the result does not depend on the order in which
// the walkers' adds arrive and is reproducible bit for bit.  A window whose maps hold a non-finite value (no number to
// scale by) takes float atomics instead: slow, NaN / inf propagate as in the reference.  SAF_WIN_FORM=rows selects the
// frame-ordered kernel (bit-identical to fusing frame after frame).
#include <cstddef>
#include <type_traits>

#include "saf_window_dev.h"

namespace saf {
namespace {

constexpr int kBX = 4, kBY = 4, kBZ = 4;   // brick shape (voxels); a thread of wave 0 owns one voxel
constexpr int kBV = kBX * kBY * kBZ;       // 64
static_assert(kBV == 64, "one wave of voxel threads, one 64-bit word of the bit matrix per frame");
constexpr int kWalkers = 4, kMovers = 4;   // waves 0 .. 3 walk, waves 4 .. 7 move rows (a quarter of the rows each: their
                                           // old pieces must fit a wave's registers at 4 waves per SIMD: two workgroups of 8
                                           // waves per CU, two waves of each on every SIMD; workgroups of 6 waves at 3 per
                                           // SIMD never ran two to a CU)
constexpr int kBWaves = kWalkers + kMovers;
constexpr int kBThreads = kBWaves * 64;
constexpr int kHitThreads = kWalkers * 64;
constexpr int kBuildThreads = 128;        // the build kernel: two waves per brick, eight bricks per CU
constexpr int kWindows = 4;                // 64-hit windows of a round, sorted one by one
#ifndef SAF_BRICK_GROUP_CAP
#define SAF_BRICK_GROUP_CAP 16
#endif
constexpr int kGroupCap = SAF_BRICK_GROUP_CAP;  // most hits of one group (a power of two, <= 64)
constexpr int kHC = kWindows * 64;         // hit records of one round (a brick with more hits takes its frames in rounds)
static_assert(kHC >= kBV, "a frame's hits fit a round");
#ifndef SAF_BRICK_P
#define SAF_BRICK_P 2
#endif
constexpr int kP = SAF_BRICK_P;            // groups of one batch: 4 tap loads each, two batches in flight

#ifdef SAF_BRICK_TIMING  // development aid: per-phase wave cycles, by role, printed by the host after every launch
__device__ unsigned long long g_brick_t[32];
__device__ unsigned long long g_walk_t[8];
#define BT_DECL unsigned long long bt_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, bt_last_ = __builtin_readcyclecounter(), wt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define BT(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); bt_[k] += n_ - bt_last_; bt_last_ = n_; } while (0)
#define BT_FLUSH do { if (lane == 0) { for (int k_ = 0; k_ < 12; ++k_) atomicAdd(&g_brick_t[(wave >= kWalkers ? 16 : 0) + k_], bt_[k_]); for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&g_walk_t[k_], wt_[k_]); } } while (0)
// ... and the walkers' time inside a slab's walk: [0] issue, [1] records + weights, [2] waiting for the taps, [3] the hits
#define WT_PARAM , unsigned long long* wt_
#define WT_ARG , wt_
#define WT(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); wt_[k] += n_ - wt_[7]; wt_[7] = n_; } while (0)
#else
#define BT_DECL
#define BT(k)
#define BT_FLUSH
#define WT_PARAM
#define WT_ARG
#define WT(k)
#endif

constexpr int kCamFloats = 21;  // the per-frame part of Cam: pose[:3,:4], K (the image-size terms are launch-uniform)

// What the walk needs of one round of one brick -- the IMAGE the build kernel leaves in the pool (a SEGMENT) and the walk
// kernel copies into LDS as it stands.  The group table comes last: a segment is read up to its last used group.
// The movers' part first (they still read it while the walkers, done with a round, already replace theirs) ...
struct alignas(16) SegRows {
  uint32_t hdr[16];              // R, nh, G, kexp, G of the brick's next round
  uint32_t rown[kBV];            // flat voxel index of a row
  float rowA[kBV], rowB[kBV];    // new = old * b + acc * a
  uint8_t rowfresh[kBV];         // row was never written (weight 0): nothing to read
};
// ... then the walkers'
struct alignas(16) SegWalk {
  uint32_t rec_k[kHC];           // row (7) | cell x (8) << 7 | cell y (8) << 15 | frame (7) << 23
  float2 rec_g[kHC];             // a hit's bilinear fractions (wx, wy)
  uint16_t grp_start[kHC + 8];   // first hit of a group; [G] = number of hits
  uint4 grp_off[kHC + 1];        // a group's four map rows: byte offsets into the window's images (slab 0, lane 0)
};
struct alignas(16) SegImage {
  SegRows rows;
  SegWalk walk;
};
constexpr uint32_t kSegBytes = (sizeof(SegImage) + 255u) & ~255u;
constexpr uint32_t kSegFixed = sizeof(SegRows) + offsetof(SegWalk, grp_off);  // a segment's used bytes: kSegFixed + (G + 1) * 16
constexpr int kRows16 = (int)(sizeof(SegRows) / 16);
static_assert(kSegFixed % 16 == 0 && sizeof(SegImage) % 16 == 0 && sizeof(SegRows) % 16 == 0 &&
                  sizeof(SegImage) == sizeof(SegRows) + sizeof(SegWalk), "segment layout");

template <int CPL>
struct alignas(16) BrickLds {
  int acc[kBV * 64 * CPL];       // [row][channel of the lane][lane]; during the build it stages the hits' rgb samples
  SegRows rows[2];               // the round at work and the one after it
  SegWalk walk;
  uint32_t mf[kWin * 2];         // bit matrix: frame x voxel
  uint16_t off[kWin + 8];        // exclusive prefix of the frames' hit counts
  uint8_t vrow[kBV];             // voxel -> row of this round
  uint32_t slabm[128];           // the slabs' largest map magnitudes (bits), read once per launch
  uint32_t misc[32];
};
// The build kernel's LDS: everything but the accumulator (the rgb staging gets an array of its own).
struct alignas(16) BuildLds {
  float stage[kHC * 3];
  SegRows rows[1];               // (rows[0] and walk: one SegImage, written to the pool as it stands)
  SegWalk walk;
  uint32_t mf[kWin * 2];
  uint16_t off[kWin + 8];
  uint8_t vrow[kBV];
  uint32_t misc[32];
};

// Control words of a window's pool (zeroed by the host before the build kernel)
struct BrickCtl {
  uint32_t seg_next;    // segments handed out
  uint32_t over_n;      // bricks that found the pool exhausted: the walk kernel builds them itself
  uint32_t over_next;   // overflow bricks taken
  uint32_t pad;
  uint32_t list_n[8];   // listed bricks per XCD (each entry: first segment, rounds, groups of the first round)
  uint32_t walk_next[8];  // ... and how many of them the walk kernel has taken
  unsigned long long acc[64][2];  // the build workgroups' counters (hits, rows), sharded
};

static_assert(sizeof(BrickLds<4>) <= 78 * 1024, "two workgroups per CU and room for the classification beside them");
static_assert(kHC * 3 * sizeof(float) <= sizeof(int) * kBV * 64, "the rgb staging lives in the accumulator");
static_assert(offsetof(BuildLds, walk) == offsetof(BuildLds, rows) + sizeof(SegRows), "rows[0] and walk form one SegImage");

// LDS-only barrier: the walkers' tap loads and the movers' row traffic stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ int rfl(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float rl_f(float x, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
// floor(x + 0.5) as an integer (one instruction; v_rndne + v_cvt would be two)
__device__ __forceinline__ int cvt_rpi(float x) {
  int q;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(x));
  return q;
}

__device__ __forceinline__ Cam cam_from(const float* __restrict__ c, const Cam& u) {
  Cam r = u;
  r.r00 = c[0]; r.r01 = c[1]; r.r02 = c[2]; r.tx = c[3];
  r.r10 = c[4]; r.r11 = c[5]; r.r12 = c[6]; r.ty = c[7];
  r.r20 = c[8]; r.r21 = c[9]; r.r22 = c[10]; r.tz = c[11];
  r.k00 = c[12]; r.k01 = c[13]; r.k02 = c[14]; r.k10 = c[15]; r.k11 = c[16]; r.k12 = c[17];
  r.k20 = c[18]; r.k21 = c[19]; r.k22 = c[20];
  return r;
}

// position of voxel t's hit among the hits of a frame (voxel order): the set bits of the frame's row below bit t
__device__ __forceinline__ int rank_in_frame(const uint32_t* __restrict__ mf_row, int t) {
  int r = __popc(mf_row[t >> 5] & ((1u << (t & 31)) - 1u));
  if (t >= 32) r += __popc(mf_row[0]);
  return r;
}

// CPL consecutive channels of a map row
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <int CPL>
struct TapVec {
  float v[CPL];
};
template <int CPL>
__device__ __forceinline__ TapVec<CPL> tap_load(__amdgpu_buffer_rsrc_t maps, uint32_t byte_off) {
  TapVec<CPL> t;
  if constexpr (CPL == 1) {
    t.v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(maps, (int)byte_off, 0, 0));
  } else if constexpr (CPL == 2) {
    // (the whole vector is bit-cast: element access r[i] on the builtin's result makes hipcc 7.2 load ONE dword and use it for
    //  every element)
    const float2 f = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(maps, (int)byte_off, 0, 0));
    t.v[0] = f.x; t.v[1] = f.y;
  } else {
    const float4 f = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(maps, (int)byte_off, 0, 0));
    t.v[0] = f.x; t.v[1] = f.y; t.v[2] = f.z; t.v[3] = f.w;
  }
  return t;
}

// ---- the walk of one slab: batches of kP consecutive groups, handed out round-robin to the walkers
struct WalkBatch {
  int hl[kP + 1];  // first hit of each group of the batch; hl[kP] = the batch's end
};
struct WalkCtx {
  __amdgpu_buffer_rsrc_t maps;
  int G, lane;
  uint32_t lane_off;  // slab * (256 CPL) + lane * (4 CPL) bytes
};

template <int CPL>
__device__ __forceinline__ void walk_issue(const BrickLds<CPL>& L, const WalkCtx& cx, int batch, WalkBatch& b,
                                           TapVec<CPL> (&tp)[kP][4] WT_PARAM) {
  const int g0 = batch * kP;
  const int hs = (int)L.walk.grp_start[min(g0 + min(cx.lane, kP), cx.G)];  // lanes 0 .. kP: the groups' first hits ([G] = the end)
#pragma unroll
  for (int u = 0; u <= kP; ++u) b.hl[u] = __builtin_amdgcn_readlane(hs, u);
#pragma unroll
  for (int u = 0; u < kP; ++u) {
#ifdef SAF_BRICK_NOTAPS  // ablation: every tap outside the buffer (no L2 -> L1 traffic, same instruction stream)
    const uint4 o = L.walk.grp_off[cx.G];
#else
    const uint4 o = L.walk.grp_off[min(g0 + u, cx.G)];  // entry G: four offsets beyond the buffer (no such group: nothing moves)
#endif
    tp[u][0] = tap_load<CPL>(cx.maps, o.x + cx.lane_off);
    tp[u][1] = tap_load<CPL>(cx.maps, o.y + cx.lane_off);
    tp[u][2] = tap_load<CPL>(cx.maps, o.z + cx.lane_off);
    tp[u][3] = tap_load<CPL>(cx.maps, o.w + cx.lane_off);
  }
  WT(0);
}

// FX: fixed-point accumulation (`scale` is folded into the weights); otherwise float atomics
template <int CPL, bool FX>
__device__ __forceinline__ void walk_process(BrickLds<CPL>& L, const WalkCtx& cx, const WalkBatch& b,
                                             const TapVec<CPL> (&tp)[kP][4], float scale, bool pending WT_PARAM) {
  for (int c0 = b.hl[0]; c0 < b.hl[kP]; c0 += 64) {  // (one round trip unless the batch has more than 64 hits)
    const int slot = min(c0 + cx.lane, b.hl[kP] - 1);
    const uint32_t key = L.walk.rec_k[slot];
    const float2 g = L.walk.rec_g[slot];
    // the bilinear weights (bilinear_setup's products), times the slab's fixed-point scale (a power of two: exact)
    const float ex = 1.0f - g.x, sy = 1.0f - g.y;
    const float w_nw = (sy * ex) * scale, w_ne = (sy * g.x) * scale, w_sw = (g.y * ex) * scale, w_se = (g.y * g.x) * scale;
    const int rowbase = (int)(key & 127u) * (64 * CPL) + cx.lane;
    // the records are here before the loops start: without this (the BUILTIN: the wait-count pass must see it) the pass puts
    // an lgkmcnt(0) at the head of the per-hit loop, where it waits for the previous hit's ds_add every time round
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); vmcnt, expcnt untouched
    const int c1 = c0 + 64;
#ifdef SAF_BRICK_TIMING  // (the taps of this batch; the next batch's 4 kP loads may stay in flight)
    WT(1);
    static_assert(kP == 2, "the wait below counts the next batch's loads");
    if (pending) __builtin_amdgcn_s_waitcnt(0x0F78); else __builtin_amdgcn_s_waitcnt(0x0F70);
    WT(2);
#endif
#pragma unroll
    for (int u = 0; u < kP; ++u) {
      const int l0 = max(b.hl[u], c0), l1 = min(b.hl[u + 1], c1);
#ifdef SAF_BRICK_NOHITS  // ablation: no per-hit work at all
      for (int l = l0; l < l0; ++l) {
#else
      for (int l = l0; l < l1; ++l) {
#endif
        const int li = l - c0;
        const float wnw = rl_f(w_nw, li), wne = rl_f(w_ne, li), wsw = rl_f(w_sw, li), wse = rl_f(w_se, li);
        const int idx = __builtin_amdgcn_readlane(rowbase, li) - li + cx.lane;  // the hit's row, this lane
        // (two channels per instruction: v_pk_mul_f32 / v_pk_fma_f32 with the hit's weight broadcast from an SGPR; the same
        //  products and FMAs, channel by channel, as the scalar form)
        float sv[CPL];
        if constexpr (CPL % 2 == 0) {
#pragma unroll
          for (int c = 0; c < CPL; c += 2) {
            const v2f_t t0 = {tp[u][0].v[c], tp[u][0].v[c + 1]}, t1 = {tp[u][1].v[c], tp[u][1].v[c + 1]};
            const v2f_t t2 = {tp[u][2].v[c], tp[u][2].v[c + 1]}, t3 = {tp[u][3].v[c], tp[u][3].v[c + 1]};
            v2f_t s2 = t0 * (v2f_t){wnw, wnw};
            s2 = __builtin_elementwise_fma(t1, (v2f_t){wne, wne}, s2);
            s2 = __builtin_elementwise_fma(t2, (v2f_t){wsw, wsw}, s2);
            s2 = __builtin_elementwise_fma(t3, (v2f_t){wse, wse}, s2);
            sv[c] = s2.x; sv[c + 1] = s2.y;
          }
        } else {
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            float s1 = tp[u][0].v[c] * wnw;
            s1 = __builtin_fmaf(tp[u][1].v[c], wne, s1);
            s1 = __builtin_fmaf(tp[u][2].v[c], wsw, s1);
            sv[c] = __builtin_fmaf(tp[u][3].v[c], wse, s1);
          }
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const float s = sv[c];
#ifdef SAF_BRICK_NOATOM  // ablation: a plain LDS store instead of the atomic add (wrong sums, same instruction count)
          if (true) {
            L.acc[idx + c * 64] = cvt_rpi(s);
          } else if (FX) {
#else
          if (FX) {
#endif
            __hip_atomic_fetch_add(&L.acc[idx + c * 64], cvt_rpi(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            __hip_atomic_fetch_add(reinterpret_cast<float*>(&L.acc[idx + c * 64]), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    WT(3);
  }
}

template <int CPL, bool FX>
__device__ __forceinline__ void walk_slab(BrickLds<CPL>& L, const WalkCtx& wc, int wave, float scale WT_PARAM) {
  const int n_batch = (wc.G + kP - 1) / kP;
  WalkBatch bA, bB;
  TapVec<CPL> tpA[kP][4], tpB[kP][4];
  int batch = wave;
  if (batch >= n_batch) return;
#ifdef SAF_BRICK_TIMING
  wt_[7] = __builtin_readcyclecounter();
  wt_[4] += (unsigned long long)n_batch;
  wt_[5] += (unsigned long long)L.walk.grp_start[wc.G];
#endif
  walk_issue<CPL>(L, wc, batch, bA, tpA WT_ARG);
  for (;;) {
    const bool hasB = batch + kWalkers < n_batch;
    if (hasB) walk_issue<CPL>(L, wc, batch + kWalkers, bB, tpB WT_ARG);
    walk_process<CPL, FX>(L, wc, bA, tpA, scale, hasB WT_ARG);
    if (!hasB) break;
    batch += 2 * kWalkers;
    const bool hasA = batch < n_batch;
    if (hasA) walk_issue<CPL>(L, wc, batch, bA, tpA WT_ARG);
    walk_process<CPL, FX>(L, wc, bB, tpB, scale, hasA WT_ARG);
    if (!hasA) break;
  }
}

// The fixed-point scale of a slab: 2^e with e = 30 - kexp - ceil(log2 M), M = the largest magnitude of its channels in the
// window's maps (bits of |x|), so that k_max samples of magnitude <= M sum to less than 2^31.  Returned as the exponent.
__device__ __forceinline__ int scale_exp(uint32_t mbits, int kexp) {
  const int ex = (int)(mbits >> 23) - 127 + ((mbits & 0x7fffffu) ? 1 : 0);  // M <= 2^ex (denormals: ex = -127 or -126)
  const int e = 30 - kexp - ex;
  return min(max(e, -96), 96);  // (a slab of values below 2^-66 is resolved to 2^-96; an all-zero slab to anything)
}
__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (uint32_t)(e + 127) << 23); }
template <int CPL>
__device__ __forceinline__ uint32_t slab_max_bits(const uint32_t* __restrict__ cmax, int D, int p) {
  uint32_t m = 0u;
#pragma unroll
  for (int j = 0; j < CPL; ++j) m = max(m, cmax[D + p * CPL + j]);
  return m;
}

// What the build and the walk kernel share of the pool
constexpr uint32_t kOverWords = 1 + kBV + 3;  // (68: entries stay 16-byte aligned)
struct BrickPool {
  unsigned char* segs;   // cap segments of kSegBytes
  uint4* list;           // 8 lists (one per XCD) of list_stride entries; per listed brick: first segment, rounds, groups of round 0
  uint32_t list_stride;
  uint32_t* over;        // per overflow brick kOverWords words: the brick's code, then its voxels' weights before this window
  BrickCtl* ctl;
  uint32_t cap;
  int split;             // 0: no build kernel ran -- the walk kernel takes every brick from the XCD counters and builds it itself
};

// One source for both kernels.  BUILD: one workgroup of 4 waves per brick, small LDS, many per CU: the per-brick build
// (hit records, scalar side, sorted groups), written to the pool as segments.  !BUILD: the persistent walk kernel (4 walker +
// 4 mover waves): segments from the pool, then the bricks of the overflow list (or, without a build kernel, every brick),
// which it builds itself.
// REBUILD (walk kernel only): this instantiation carries the build code too -- it builds every brick itself (no build kernel:
// pool.split == 0) or the bricks of the overflow list (pool.split == 3: a small second launch behind the walk of the pool's
// segments).  The walk of the pool's segments (pool.split == 2) is compiled WITHOUT it: with the build code inlined the walk
// kernel spilled 23-95 registers at its cap of 128.
template <int CPL, bool SUM, bool BF16, bool BUILD, bool REBUILD = true>
// (round 5: the walk instantiations that carry the build code -- the small launch over the overflow list, and SAF_BRICK_SPLIT=0 --
//  may use the registers of two waves per SIMD instead of four: they spilled 6-96 VGPRs at the cap of 128.  Their workgroups are
//  independent -- work is handed out by counters -- so fewer of them resident at once only lengthens those rare launches.)
__global__ __launch_bounds__(BUILD ? kBuildThreads : kBThreads) __attribute__((amdgpu_waves_per_eu((!BUILD && REBUILD) ? 2 : 4, 4))) void fuse_brick_kernel(
    KVol v, WinArgs wa, const WinTable* __restrict__ tab, const float* __restrict__ map_imgs, uint32_t img_bytes,
    unsigned long long* __restrict__ stats, unsigned int* __restrict__ ctr, const uint32_t* __restrict__ hitmask,
    uint32_t mask_plane, const unsigned long long* __restrict__ cls_acc, const uint32_t* __restrict__ cmax,
    const float* __restrict__ cams, BrickPool pool) {
  using Lds = typename std::conditional<BUILD, BuildLds, BrickLds<CPL>>::type;
  constexpr int kNT = BUILD ? kBuildThreads : kBThreads;
  constexpr int kNHit = BUILD ? kBuildThreads : kHitThreads;  // threads that build hit records
  constexpr int kNW = kNHit / 64;                             // ... and waves that sort windows
  extern __shared__ __align__(16) unsigned char s_dyn[];
  Lds& L = *reinterpret_cast<Lds*>(s_dyn);
  // (values read from LDS or derived from the thread index are divergent to the compiler: what is wave-uniform is said
  //  so with readfirstlane -- scalar branches, loop counters in SGPRs)
  const int tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
  const int F = wa.F;
  constexpr int kSlabCh = 64 * CPL;

  if (!BUILD && stats && blockIdx.x == 0 && tid < kClsShards && pool.split != 3) {  // the classification launches' sharded counters (cls_accumulate)
    unsigned long long a = cls_acc[2 * tid], b = cls_acc[2 * tid + 1];
    unsigned long long c = 0, d = 0;  // ... and the build kernel's (hits, rows)
    if (pool.split) { c = pool.ctl->acc[tid][0]; d = pool.ctl->acc[tid][1]; }
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o);
      b += __shfl_xor(b, o);
      c += __shfl_xor(c, o);
      d += __shfl_xor(d, o);
    }
    if (tid == 0) {
      if (a) atomicAdd(&stats[1], a);
      if (b) atomicAdd(&stats[6], b);
      if (c) atomicAdd(&stats[0], c);
      if (d) atomicAdd(&stats[5], d);
    }
    fold_cull_shards(cls_acc, stats, tid);
  }
  const Cam ucam = load_cam(tab->pose[0], tab->K[0], wa.W, wa.H);  // its image-size terms are the same for every frame
  const int n_pass = v.D / kSlabCh;
  const float half_px = (float)wa.npx / 2.0f, half_py = (float)wa.npy / 2.0f;
  float4* feat = reinterpret_cast<float4*>(v.feat);
  const __amdgpu_buffer_rsrc_t maps_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(map_imgs), 0, (int)((size_t)F * img_bytes), 0x00020000);
  const uint32_t row_bytes = (uint32_t)v.D * 4u;
  KFrame kf;
  kf.H = wa.H; kf.W = wa.W; kf.npy = wa.npy; kf.npx = wa.npx; kf.rgb_bilinear = wa.rgb_bilinear;
  kf.depth = nullptr; kf.pose = nullptr; kf.K = nullptr;
  unsigned long long hits_done = 0, rows_done = 0;
  // fixed-point accumulation unless a map value of this window is not finite (chan_max_kernel leaves the largest |x| of all
  // channels behind the per-channel and per-64-channel maxima)
  const bool fx = BUILD ? true : rfl((int)(cmax[v.D + v.D / 64] < 0x7f800000u)) != 0;
  BT_DECL;

  // Bricks are handed out XCD by XCD: workgroup i runs on XCD i % 8 (round-robin dispatch), every XCD draws from its own
  // counter and walks its own tiles of 4 x 4 brick columns (tile t belongs to XCD t % 8) in sections of 4 bricks along z,
  // so the 64 bricks an XCD has in flight are one 16 x 16 x 16-voxel box whose map taps stay in its L2.  An XCD that
  // runs out helps the next one.  Bricks beyond a ragged grid's edge are skipped, voxels beyond it masked.
  const uint32_t nbx = ((uint32_t)v.nx + kBX - 1) / kBX, nby = ((uint32_t)v.ny + kBY - 1) / kBY, nbz = ((uint32_t)v.nz + kBZ - 1) / kBZ;
  const uint32_t tiles_y = (nby + 3u) / 4u, n_tiles = ((nbx + 3u) / 4u) * tiles_y;
  const uint32_t zsecs = (nbz + 3u) / 4u, upt = zsecs * 64u;
  uint32_t xcd = blockIdx.x & 7u, xcd_tries = 0;

  // The next segment's image is fetched by the walkers when they have walked a round's last slab, while the movers write
  // that slab out (3 x 16 bytes per walker thread): the walkers' part of the image replaces the current one, which only
  // they read; the movers' part goes to the other of two buffers.
  constexpr int kImgRegs = (int)((sizeof(SegImage) / 16 + kHitThreads - 1) / kHitThreads);
  [[maybe_unused]] bool ahead = false;       // (the segment loop is running)
  [[maybe_unused]] bool ahead_same = false;  // the next segment is this brick's next round ...
  [[maybe_unused]] uint32_t ahead_seg = 0;   // ... this one; otherwise the next brick's first (L.misc[24..])
  [[maybe_unused]] bool take_entry = false;  // a brick's first round: one mover thread draws the workgroup's next brick
  int cur = 0;  // the rows buffer of the round at work
  // The workgroup's next listed brick: from its XCD's list, then (that one exhausted) the next XCD's ...  One thread; the
  // entry (first segment, rounds, groups of round 0, found) goes to L.misc[20 + 4 slot ..]: two slots, the brick at work and
  // the one after it; the list position is kept in L.misc[28..29].
  int eb = 0;  // the slot of the brick at work
  [[maybe_unused]] auto draw_entry = [&](const int slot) {
    if constexpr (!BUILD) {
      uint32_t lxx = L.misc[28], tries = L.misc[29];
      uint4 en = make_uint4(0u, 0u, 0u, 0u);
      while (tries < 8u) {
        const uint32_t n = atomicAdd(&pool.ctl->walk_next[lxx], 1u);
        if (n < pool.ctl->list_n[lxx]) {
          en = pool.list[(size_t)lxx * pool.list_stride + n];
          en.w = 1u;
          break;
        }
        lxx = (lxx + 1u) & 7u;
        ++tries;
      }
      L.misc[20 + 4 * slot] = en.x; L.misc[21 + 4 * slot] = en.y; L.misc[22 + 4 * slot] = en.z; L.misc[23 + 4 * slot] = en.w;
      L.misc[28] = lxx; L.misc[29] = tries;
    }
  };
  // ---- the slabs of one round: R rows, G groups, kexp = ceil(log2 of the most hits a row takes); the records, the group
  //      table and the rows' coefficients are in LDS
  [[maybe_unused]] auto run_slabs = [&](const int R, const int G, const int kexp) {
    if constexpr (!BUILD) {
      // ---- slabs of 64 CPL channels.  The two roles run different code between the same barriers (their registers --
      //      the movers' row pieces, the walkers' two tap batches -- are never live together).
      if (wave >= kWalkers) {
        // A mover's 16-byte piece: PPR pieces per row slab, RPI rows per iteration, the movers take rows alternately.
        constexpr int kEB = BF16 ? 2 : 4;                 // bytes per stored channel
        constexpr int kPPR = kSlabCh * kEB / 16;          // pieces per row slab (4 .. 64)
        constexpr int kRPI = 64 / kPPR;                   // rows per iteration and mover (1 .. 16)
        constexpr int kMI = ((kBV + kMovers - 1) / kMovers + kRPI - 1) / kRPI;  // iterations
        constexpr int kCPP = 16 / kEB;                    // channels per piece
        const int row_vecs = v.D * kEB / 16;              // 16-byte units of a row
        float4 old[kMI];
        // (opaque to the optimiser: the iterations' row numbers, LDS offsets and row pointers would otherwise be computed once
        //  at the top of the kernel and kept -- a hundred spilled registers)
        int m_sub = lane / kPPR, m_pc = lane % kPPR;
        asm volatile("" : "+v"(m_sub), "+v"(m_pc));
        // the first slab's old pieces: they arrive during the first walk
#pragma unroll
        for (int i = 0; i < kMI; ++i) {
          const int mr = (i * kRPI + m_sub) * kMovers + (wave - kWalkers);
          old[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (i * kRPI * kMovers < R && mr < R && !L.rows[cur].rowfresh[mr]) old[i] = ld_stream(feat + (int64_t)L.rows[cur].rown[mr] * row_vecs + m_pc);
        }
        if (take_entry && tid == kHitThreads) draw_entry(eb ^ 1);  // (published by the slab's barriers)
        int* acc = L.acc;
        for (int p = 0; p < n_pass; ++p) {
          lds_barrier();  // the walkers have added this slab's samples
          BT(6);
          asm volatile("" : "+v"(m_sub), "+v"(m_pc));
          const float inv = fx ? pow2f(-scale_exp((uint32_t)rfl((int)L.slabm[p & 127]), kexp)) : 1.0f;
          // new = old * b + acc * a.  The old pieces were requested a whole walk ago: one explicit wait (the BUILTIN, which
          // the wait-count pass sees) -- left to itself the pass puts a vmcnt(0) into every conditional iteration, where it
          // waits for the store of the iteration before: a memory round trip per iteration
          __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
          for (int i = 0; i < kMI; ++i) {
            const int mr = (i * kRPI + m_sub) * kMovers + (wave - kWalkers);
            if (i * kRPI * kMovers < R && mr < R) {
              const float a = L.rows[cur].rowA[mr], b = L.rows[cur].rowB[mr];
              // channel ch of the slab lives in lane ch / CPL, slot ch % CPL of the walkers' layout
              float sm[kCPP];
#pragma unroll
              for (int j = 0; j < kCPP; ++j) {
                const int ch = m_pc * kCPP + j;
                int* ap = &acc[mr * kSlabCh + (ch % CPL) * 64 + ch / CPL];
                const int raw = *ap;
                *ap = 0;
                sm[j] = (fx ? (float)raw * inv : __builtin_bit_cast(float, raw)) * a;
              }
              float4* gp = feat + (int64_t)L.rows[cur].rown[mr] * row_vecs + (int64_t)p * kPPR + m_pc;
              float4 o;
              if constexpr (BF16) {
                const uint32_t ox = __builtin_bit_cast(uint32_t, old[i].x), oy = __builtin_bit_cast(uint32_t, old[i].y);
                const uint32_t oz = __builtin_bit_cast(uint32_t, old[i].z), ow = __builtin_bit_cast(uint32_t, old[i].w);
                o.x = __builtin_bit_cast(float, pack_bf16(bf16_lo(ox) * b + sm[0], bf16_hi(ox) * b + sm[1]));
                o.y = __builtin_bit_cast(float, pack_bf16(bf16_lo(oy) * b + sm[2], bf16_hi(oy) * b + sm[3]));
                o.z = __builtin_bit_cast(float, pack_bf16(bf16_lo(oz) * b + sm[4], bf16_hi(oz) * b + sm[5]));
                o.w = __builtin_bit_cast(float, pack_bf16(bf16_lo(ow) * b + sm[6], bf16_hi(ow) * b + sm[7]));
              } else {
                o.x = old[i].x * b + sm[0]; o.y = old[i].y * b + sm[1];
                o.z = old[i].z * b + sm[2]; o.w = old[i].w * b + sm[3];
              }
#ifndef SAF_BRICK_NOROWS  // ablation: no row stores (and no row loads below)
              st_stream(gp, o);
#else
              asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
#endif
            }
          }
          // (a loop of its own: behind each store, a load would make the wait-count pass drain the queue in every
          //  iteration; here the first use of `old` in the next slab finds them all arrived)
          if (p + 1 < n_pass) {
#pragma unroll
            for (int i = 0; i < kMI; ++i) {
              const int mr = (i * kRPI + m_sub) * kMovers + (wave - kWalkers);
#ifndef SAF_BRICK_NOROWS
              if (i * kRPI * kMovers < R && mr < R && !L.rows[cur].rowfresh[mr])
                old[i] = ld_stream(feat + (int64_t)L.rows[cur].rown[mr] * row_vecs + (int64_t)(p + 1) * kPPR + m_pc);
#endif
            }
          }
          BT(7);
          lds_barrier();  // the accumulator is clear again
          BT(8);
        }
      } else {
        WalkCtx wc;
        wc.maps = maps_rsrc; wc.G = G; wc.lane = lane;
        for (int p = 0; p < n_pass; ++p) {
          wc.lane_off = (uint32_t)p * (kSlabCh * 4u) + (uint32_t)lane * (4u * CPL);
          if (fx)
            walk_slab<CPL, true>(L, wc, wave, pow2f(scale_exp((uint32_t)rfl((int)L.slabm[p & 127]), kexp)) WT_ARG);
          else
            walk_slab<CPL, false>(L, wc, wave, 1.0f WT_ARG);
          BT(6);
          lds_barrier();  // this slab's samples are in the accumulator
          // the last slab is walked: the next segment's image
          const uint32_t* nx_ent = &L.misc[20 + 4 * (eb ^ 1)];
          const uint32_t nx_seg = ahead_same ? ahead_seg : (uint32_t)rfl((int)nx_ent[0]);
          const int next_n16 = (int)(kSegFixed / 16) + 1 + min(rfl((int)(ahead_same ? L.rows[cur].hdr[4] : nx_ent[2])), kHC);
          if (p + 1 == n_pass && ahead && (ahead_same || rfl((int)nx_ent[3]) != 0)) {
            // (a native vector type: as HIP's uint4 -- a struct around a union -- the conditionally loaded image lived in scratch)
            typedef unsigned int img_u4 __attribute__((ext_vector_type(4)));
            const img_u4* next_sg = reinterpret_cast<const img_u4*>(pool.segs + (size_t)nx_seg * kSegBytes);
            img_u4 img[kImgRegs];
#pragma unroll
            for (int k = 0; k < kImgRegs; ++k) {
              const int i16 = tid + k * kHitThreads;
              if (i16 < next_n16) img[k] = next_sg[i16];
            }
            img_u4* d_rows = reinterpret_cast<img_u4*>(&L.rows[cur ^ 1]);
            img_u4* d_walk = reinterpret_cast<img_u4*>(&L.walk) - kRows16;
#pragma unroll
            for (int k = 0; k < kImgRegs; ++k) {
              const int i16 = tid + k * kHitThreads;
              if (i16 < next_n16) (i16 < kRows16 ? d_rows : d_walk)[i16] = img[k];
            }
          }
          BT(7);
          lds_barrier();  // the movers have written the slab and cleared the accumulator
          BT(8);
        }
      }
    }
  };

  // ---- BUILD: one round's records, group table and row coefficients go to the pool
  uint32_t seg_base = 0;  // first segment of this brick
  int seg_round = 0;
  uint32_t list_slot = 0;  // this brick's entry in its XCD's list
  [[maybe_unused]] auto write_segment = [&](const int R, const int nh, const int G, const int kexp) {
    if constexpr (BUILD) {
      if (seg_base == 0xffffffffu) {
        lds_barrier();
        return;
      }
      if (tid == 0) {
        L.rows[cur].hdr[0] = (uint32_t)R; L.rows[cur].hdr[1] = (uint32_t)nh; L.rows[cur].hdr[2] = (uint32_t)G; L.rows[cur].hdr[3] = (uint32_t)kexp;
        L.rows[cur].hdr[4] = (uint32_t)kHC;
        // the walk kernel reads a segment up to its last group: a first round's count is in the list, a later one's in
        // the header of the round before
        if (seg_round == 0)
          pool.list[list_slot].z = (uint32_t)G;
        else
          reinterpret_cast<uint32_t*>(pool.segs + (size_t)(seg_base + (uint32_t)seg_round - 1u) * kSegBytes)[4] = (uint32_t)G;
      }
      lds_barrier();
      uint4* dst = reinterpret_cast<uint4*>(pool.segs + (size_t)(seg_base + (uint32_t)seg_round) * kSegBytes);
      const uint4* src = reinterpret_cast<const uint4*>(&L.rows[0]);  // (rows[0] and walk: one image)
      const int n16 = (int)(kSegFixed / 16) + G + 1;
      for (int i = tid; i < n16; i += kNT) dst[i] = src[i];
      ++seg_round;
      lds_barrier();  // the arrays are rewritten by the next round
    }
  };

  // voxel of this thread within a brick (wave 0)
  const int lx = (tid >> 4) & 3, ly = (tid >> 2) & 3, lz = tid & 3;
  const bool vth = tid < kBV;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;

  if constexpr (!BUILD) {
    // the accumulator starts clear and every slab leaves it clear
    {
      int4* acc4 = reinterpret_cast<int4*>(L.acc);
      for (int i = tid; i < kBV * kSlabCh / 4; i += kNT) acc4[i] = make_int4(0, 0, 0, 0);
    }
    for (int p = tid; p < n_pass && p < 128; p += kNT) L.slabm[p] = slab_max_bits<CPL>(cmax, v.D, p);
    lds_barrier();
    // ---- the bricks the build kernel has prepared.  Every XCD has its own list (the build workgroups of its tiles filled
    //      it) and its workgroups draw from it -- the map taps and rows of neighbouring bricks stay in one L2.  A brick is
    //      drawn while the one before it is walked, and the walkers fetch a segment's image as soon as they have walked
    //      the last slab of the segment before: it lands while the movers write that slab out.
    if (pool.split == 1 || pool.split == 2) {
      if (tid == kHitThreads) {
        L.misc[28] = blockIdx.x & 7u;
        L.misc[29] = 0u;
        draw_entry(0);
      }
      lds_barrier();
      ahead = true;
      bool have_image = false;  // the segment's image is already in LDS
      while (rfl((int)L.misc[23 + 4 * eb]) != 0) {
        const uint32_t s0 = (uint32_t)rfl((int)L.misc[20 + 4 * eb]), n_r = (uint32_t)rfl((int)L.misc[21 + 4 * eb]);
        const uint32_t g0 = (uint32_t)rfl((int)L.misc[22 + 4 * eb]);
        for (uint32_t r = 0; r < n_r; ++r) {
          if (have_image) {
            cur ^= 1;
          } else {  // (the workgroup's first segment)
            const uint4* sg = reinterpret_cast<const uint4*>(pool.segs + (size_t)(s0 + r) * kSegBytes);
            uint4* d_rows = reinterpret_cast<uint4*>(&L.rows[cur]);
            uint4* d_walk = reinterpret_cast<uint4*>(&L.walk) - kRows16;
            const int n16 = (int)(kSegFixed / 16) + (int)(r == 0 ? g0 : (uint32_t)kHC) + 1;
            for (int i16 = tid; i16 < n16; i16 += kNT) (i16 < kRows16 ? d_rows : d_walk)[i16] = sg[i16];
            lds_barrier();
          }
          const int R = rfl((int)L.rows[cur].hdr[0]), G = rfl((int)L.rows[cur].hdr[2]), kexp = rfl((int)L.rows[cur].hdr[3]);
          ahead_same = r + 1 < n_r;
          ahead_seg = s0 + r + 1;
          take_entry = r == 0;
          BT(5);
          run_slabs(R, G, kexp);
          have_image = true;
          // (a brick's next round reads the rows this one has written: see the end of a round below.  The round's last
          //  barrier has also published the next image.)
          if (ahead_same) {
            if (wave >= kWalkers) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            lds_barrier();
          }
        }
        eb ^= 1;
      }
      ahead = false;  // (the bricks built here have nothing to fetch ahead)
      take_entry = false;
      cur = 0;
    }
  }

  // ---- bricks built here: BUILD: the workgroup's own brick; the walk kernel: the bricks of the overflow list, or (no build
  //      kernel) every brick, drawn XCD by XCD
  if constexpr (BUILD || REBUILD) {
  bool build_done = false;
  // The walk kernel behind a build kernel only rebuilds what the pool had no room for: the build kernel has already done
  // those bricks' scalar side (rgb, weights, labels, counters) -- the next window's build kernel may be running beside this
  // kernel and reads them -- so it is not done again, and the weight before this window is the stored one minus the hits.
  const bool redo = !BUILD && pool.split != 0;
  for (;;) {
    if (tid == 0) {
      uint32_t code = 0xffffffffu;
      if constexpr (BUILD) {
        if (!build_done) {
          const uint32_t bxcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
          const uint32_t my_tiles = (n_tiles + 7u - bxcd) / 8u;
          if (j < my_tiles * upt) {
            const uint32_t tl = j / upt, within = j - tl * upt, zs = within >> 6, k = within & 63u;
            const uint32_t t = tl * 8u + bxcd, tx = t / tiles_y, ty = t - tx * tiles_y;
            const uint32_t bx = tx * 4u + ((k >> 2) & 3u), by = ty * 4u + (k & 3u), bz = zs * 4u + (k >> 4);
            if (bx < nbx && by < nby && bz < nbz) code = bx | (by << 10) | (bz << 20);
          }
        }
      } else if (pool.split) {
        const uint32_t o = atomicAdd(&pool.ctl->over_next, 1u);
        if (o < pool.ctl->over_n) {
          code = pool.over[(size_t)o * kOverWords];
          L.misc[6] = o;
        }
      } else
      while (xcd_tries < 8u) {
        const uint32_t my_tiles = (n_tiles + 7u - xcd) / 8u;
        const uint32_t j = atomicAdd(ctr + xcd, 1u);
        if (j >= my_tiles * upt) {
          ++xcd_tries;
          xcd = (xcd + 1u) & 7u;
          continue;
        }
        const uint32_t tl = j / upt, within = j - tl * upt, zs = within >> 6, k = within & 63u;
        const uint32_t t = tl * 8u + xcd, tx = t / tiles_y, ty = t - tx * tiles_y;
        const uint32_t bx = tx * 4u + ((k >> 2) & 3u), by = ty * 4u + (k & 3u), bz = zs * 4u + (k >> 4);
        if (bx >= nbx || by >= nby || bz >= nbz) continue;
        code = bx | (by << 10) | (bz << 20);
        break;
      }
      L.misc[0] = code;
    }
    lds_barrier();
    const uint32_t code = (uint32_t)rfl((int)L.misc[0]);
    if (code == 0xffffffffu) break;
    build_done = true;
    const int bx = (int)(code & 1023u), by = (int)((code >> 10) & 1023u), bz = (int)(code >> 20);
    const int ix = bx * kBX + lx, iy = by * kBY + ly, iz = bz * kBZ + lz;
    const bool inb = vth && ix < v.nx && iy < v.ny && iz < v.nz;
    const uint32_t n = ((uint32_t)ix * (uint32_t)v.ny + (uint32_t)iy) * (uint32_t)v.nz + (uint32_t)iz;
    uint32_t mk[kMaskWords];
#pragma unroll
    for (int w = 0; w < kMaskWords; ++w) mk[w] = (inb && w * 32 < F) ? hitmask[(size_t)w * mask_plane + n] : 0u;
    uint32_t any = 0u;
#pragma unroll
    for (int w = 0; w < kMaskWords; ++w) any |= mk[w];
    // (the zeroing of the bit matrix rides on this barrier)
    for (int i = tid; i < kWin * 2; i += kNT) L.mf[i] = 0u;
    if (wave == 0) {
      const unsigned long long wany = __ballot(any != 0u);
      if (lane == 0) L.misc[4] = wany != 0ull ? 1u : 0u;
    }
    lds_barrier();
    BT(0);
    if (rfl((int)L.misc[4]) == 0) continue;
    int w_cur = 0;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (any) {
      if (!redo) w_cur = v.weight[n];
      if (redo) {
        // (NOT the stored weight minus this window's hits: the next window's build kernel may already have added its own)
        w_cur = (int)pool.over[(size_t)L.misc[6] * kOverWords + 1 + tid];
      } else {
        const float* src = v.rgb + (int64_t)n * 3;
        o0 = src[0]; o1 = src[1]; o2 = src[2];
        rows_done += 1ull;
      }
      // ---- bit matrix frame x voxel
#pragma unroll
      for (int w = 0; w < kMaskWords; ++w) {
        uint32_t mm = mk[w];
        while (mm) {
          const int f = __ffs((int)mm) - 1 + 32 * w;
          mm &= mm - 1u;
          atomicOr(&L.mf[f * 2 + (tid >> 5)], 1u << (tid & 31));
        }
      }
    }
    lds_barrier();
    // ---- hits per frame and their exclusive prefix (frames beyond F have none)
    {
      int cf = 0;
      if (tid < kWin) cf = __popc(L.mf[tid * 2]) + __popc(L.mf[tid * 2 + 1]);
      int incl = cf;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      if (tid == 63) L.misc[1] = (uint32_t)incl;
      lds_barrier();
      if (tid < kWin) {
        const int add = wave == 1 ? rfl((int)L.misc[1]) : 0;
        L.off[tid] = (uint16_t)(incl - cf + add);
        if (tid == kWin - 1) L.off[kWin] = (uint16_t)(incl + add);
      }
      lds_barrier();
    }
    if (tid == 0 && !redo) hits_done += (unsigned long long)L.off[kWin];
    BT(1);

    // ---- rounds: consecutive frames whose hits fit the record arrays
    auto round_end = [&](const int f0) {
      const int base = rfl((int)L.off[f0]);
      int f1 = F;
      if (rfl((int)L.off[F]) - base > kHC) {
        int lo = f0 + 1, hi = F;  // a single frame has at most kBV <= kHC hits
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (rfl((int)L.off[mid]) - base <= kHC) lo = mid; else hi = mid - 1;
        }
        f1 = lo;
      }
      return f1;
    };
    if constexpr (BUILD) {
      // the brick's rounds get consecutive segments of the pool; a brick that finds the pool exhausted goes to the overflow
      // list untouched (the walk kernel builds it itself)
      int n_r = 0;
      for (int f = 0; f < F;) {
        const int f1 = round_end(f);
        if (rfl((int)L.off[f1]) - rfl((int)L.off[f]) == 0) break;
        ++n_r;
        f = f1;
      }
      if (tid == 0) {
        const uint32_t b0 = atomicAdd(&pool.ctl->seg_next, (uint32_t)n_r);
        if (b0 + (uint32_t)n_r <= pool.cap) {
          const uint32_t lx = blockIdx.x & 7u;  // the XCD this brick's tile belongs to
          const uint32_t slot = lx * pool.list_stride + atomicAdd(&pool.ctl->list_n[lx], 1u);
          pool.list[slot] = make_uint4(b0, (uint32_t)n_r, 0u, 0u);
          L.misc[5] = b0;
          L.misc[7] = slot;
        } else {
          const uint32_t o = atomicAdd(&pool.ctl->over_n, 1u);
          pool.over[(size_t)o * kOverWords] = code;
          L.misc[5] = 0xffffffffu;
          L.misc[6] = o;
        }
      }
      lds_barrier();
      seg_base = (uint32_t)rfl((int)L.misc[5]);
      list_slot = (uint32_t)rfl((int)L.misc[7]);
      seg_round = 0;
      if (seg_base == 0xffffffffu && vth)  // the walk kernel rebuilds this brick: it needs the weights as they are NOW
        pool.over[(size_t)L.misc[6] * kOverWords + 1 + tid] = (uint32_t)w_cur;
      // (a brick without segments still gets its scalar side here; the walk kernel rebuilds its records)
    }
    int f0 = 0;
    while (f0 < F) {
      const int base = rfl((int)L.off[f0]);
      const int f1 = round_end(f0);
      const int nh = rfl((int)L.off[f1]) - base;
      if (nh == 0) break;  // only when nothing is left
      // ---- the voxels' hits of this round: rows, record slots
      uint32_t rm[kMaskWords];
      int k_v = 0;
#pragma unroll
      for (int w = 0; w < kMaskWords; ++w) {
        const int lo_b = f0 - 32 * w, hi_b = f1 - 32 * w;  // bits [lo_b, hi_b) of word w
        const uint32_t m_lo = lo_b <= 0 ? 0xffffffffu : (lo_b >= 32 ? 0u : ~((1u << lo_b) - 1u));
        const uint32_t m_hi = hi_b <= 0 ? 0u : (hi_b >= 32 ? 0xffffffffu : ((1u << hi_b) - 1u));
        rm[w] = mk[w] & m_lo & m_hi;
        k_v += __popc(rm[w]);
      }
      const bool touched = k_v > 0;
      const unsigned long long bal = __ballot(touched);
      if (wave == 0) {
        int km = k_v;  // the most hits any row takes in this round (bounds the fixed-point sums)
        for (int o = 32; o > 0; o >>= 1) km = max(km, __shfl_xor(km, o));
        if (lane == 0) { L.misc[2] = (uint32_t)__popcll(bal); L.misc[8] = (uint32_t)km; }
        const int row = __popcll(bal & lt_mask);
        if (touched) {
          L.vrow[tid] = (uint8_t)row;
          L.rows[cur].rown[row] = n;
          L.rows[cur].rowfresh[row] = w_cur == 0 ? 1 : 0;  // never written: all zeros by construction, not read
#pragma unroll
          for (int w = 0; w < kMaskWords; ++w) {
            uint32_t mm = rm[w];
            while (mm) {
              const int f = __ffs((int)mm) - 1 + 32 * w;
              mm &= mm - 1u;
              const int slot = (int)L.off[f] - base + rank_in_frame(&L.mf[f * 2], tid);
              L.walk.rec_k[slot] = (uint32_t)tid | ((uint32_t)f << 8);
            }
          }
        }
      }
      lds_barrier();
      const int R = rfl((int)L.misc[2]);
      const int k_max = rfl((int)L.misc[8]);
      const int kexp = k_max <= 1 ? 0 : 32 - __builtin_clz((unsigned)(k_max - 1));  // ceil(log2 k_max)
      BT(2);

      if (wave < kWalkers) {
        // ---- walkers: per hit: projection, map cell, bilinear fractions, the frame's rgb sample and label count
        //      (clipfusion.py:647-659, :701-713; clip_seem_fusion.py:786-822)
        float* stage;
        if constexpr (BUILD) stage = L.stage; else stage = reinterpret_cast<float*>(L.acc);
        for (int j = tid; j < nh; j += kNHit) {
          const uint32_t vf = L.walk.rec_k[j];
          const int tv = (int)(vf & 127u), f = (int)(vf >> 8);
          const int hx = bx * kBX + ((tv >> 4) & 3), hy = by * kBY + ((tv >> 2) & 3), hz = bz * kBZ + (tv & 3);
          const uint32_t hn = ((uint32_t)hx * (uint32_t)v.ny + (uint32_t)hy) * (uint32_t)v.nz + (uint32_t)hz;
          const Cam cam = cam_from(cams + f * kCamFloats, ucam);
          const Proj p = project(cam, v.ax[hx], v.ay[hy], v.az[hz]);
          const Bilin bw = bilinear_setup(p.gx, p.gy, half_px, half_py);
          // the fractions bilinear_setup built its weights from (nw = (1 - wy)(1 - wx), ...), recomputed the same way
          const float ux = unnormalize(p.gx, half_px), uy = unnormalize(p.gy, half_py);
          const float wx = ux - __builtin_floorf(ux), wy = uy - __builtin_floorf(uy);
          const int cx = min(max(bw.x0, -2), wa.npx) + 2, cy = min(max(bw.y0, -2), wa.npy) + 2;
          L.walk.rec_g[j] = make_float2(wx, wy);
          L.walk.rec_k[j] = (uint32_t)L.vrow[tv] | ((uint32_t)cx << 7) | ((uint32_t)cy << 15) | ((uint32_t)f << 23);
          if (!redo) {
            kf.rgb = tab->rgb[f];
            kf.label_map = tab->label_map[f];
            float s0, s1, s2;
            const int pix = sample_rgb_lane(kf, cam, p.gx, p.gy, s0, s1, s2);
            stage[j * 3] = s0;
            stage[j * 3 + 1] = s1;
            stage[j * 3 + 2] = s2;
            count_label_lane<true>(v, kf, hn, pix, stats);
          }
        }
      }
      lds_barrier();
      BT(3);
      // ---- per voxel, in frame order: the rgb running mean and the weight, exactly as frame after frame
      //      (clipfusion.py:715-721); the row's coefficients for the folded update of the feature row
      if (wave == 0 && touched) {
        const float* stage;
        if constexpr (BUILD) stage = L.stage; else stage = reinterpret_cast<const float*>(L.acc);
        const int row = (int)L.vrow[tid];
        int r = 0;
#pragma unroll
        for (int w = 0; w < kMaskWords; ++w) {
          uint32_t mm = redo ? 0u : rm[w];
          while (mm) {
            const int f = __ffs((int)mm) - 1 + 32 * w;
            mm &= mm - 1u;
            const int slot = (int)L.off[f] - base + rank_in_frame(&L.mf[f * 2], tid);
            const int wi = w_cur + r;
            const float a = 1.0f / (float)(wi + 1), b = (float)wi * a;
            o0 = blend(stage[slot * 3], o0, a, b, SUM);
            o1 = blend(stage[slot * 3 + 1], o1, a, b, SUM);
            o2 = blend(stage[slot * 3 + 2], o2, a, b, SUM);
            ++r;
          }
        }
        const float a = SUM ? 1.0f : 1.0f / (float)(w_cur + k_v);
        L.rows[cur].rowA[row] = a;
        L.rows[cur].rowB[row] = SUM ? 1.0f : (float)w_cur * a;
        w_cur += k_v;
      }
      lds_barrier();
      BT(4);
      // ---- everybody clears the accumulator (the staging is done with); the walkers sort their window of 64 hits by
      //      (frame, cell): a group = a run of hits that blend the same four map rows
      if constexpr (!BUILD) {
        int4* acc4 = reinterpret_cast<int4*>(L.acc);
        for (int i = tid; i < R * (kSlabCh / 4); i += kNT) acc4[i] = make_int4(0, 0, 0, 0);
      }
      if (wave < kWalkers) {
        for (int win = wave; win < kWindows; win += kNW) {
          const int wb = win * 64, cnt = min(64, nh - wb);  // (cnt <= 0: no such window)
          const bool hv = lane < cnt;
          const uint32_t key = hv ? L.walk.rec_k[wb + lane] : 0xffffffffu;
          const float2 g = L.walk.rec_g[wb + lane];
          const uint32_t sk = key >> 7;
          int rank = 0;
          if (cnt > 0) {
            for (int j = 0; j < 64; ++j) {
              const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)sk, j);
              rank += (kj < sk || (kj == sk && j < lane)) ? 1 : 0;
            }
          }
          wave_lds_sync();
          if (hv) {
            L.walk.rec_k[wb + rank] = key;
            L.walk.rec_g[wb + rank] = g;
          }
          wave_lds_sync();
          const uint32_t skey = hv ? L.walk.rec_k[wb + lane] : 0xffffffffu;
          const uint32_t gk = skey >> 7, pg = (uint32_t)__shfl_up((int)gk, 1);
          // (a group of more than kGroupCap hits is cut into pieces with the same four map rows: one wave walks a batch's
          //  hits one after the other -- in a coherent scene a brick's 64 voxels fall into one cell of a frame, and the
          //  other walkers would watch one of them walk them all)
          const uint32_t pgc = (uint32_t)__shfl_up((int)gk, kGroupCap);
          const unsigned long long heads =
              __ballot(hv && (lane == 0 || gk != pg || (lane >= kGroupCap && (lane & (kGroupCap - 1)) == 0 && gk == pgc)));
          if (lane == 0) L.misc[16 + win] = (uint32_t)__popcll(heads);
        }
      }
      lds_barrier();
      // ---- the table of groups: first hit, and the four map rows as byte offsets into the window's images; a tap outside
      //      the map (zeros padding) gets an offset beyond the buffer: the range check returns zero and moves nothing
      const int g_w0 = rfl((int)L.misc[16]), g_w1 = rfl((int)L.misc[17]), g_w2 = rfl((int)L.misc[18]), g_w3 = rfl((int)L.misc[19]);
      const int G = g_w0 + g_w1 + g_w2 + g_w3;
      if (wave < kWalkers) {
        for (int win = wave; win < kWindows; win += kNW) {
          const int wb = win * 64, cnt = min(64, nh - wb);
          const bool hv = lane < cnt;
          const uint32_t skey = hv ? L.walk.rec_k[wb + lane] : 0xffffffffu;
          const uint32_t gk = skey >> 7, pg = (uint32_t)__shfl_up((int)gk, 1);
          // (a group of more than kGroupCap hits is cut into pieces with the same four map rows: one wave walks a batch's
          //  hits one after the other -- in a coherent scene a brick's 64 voxels fall into one cell of a frame, and the
          //  other walkers would watch one of them walk them all)
          const uint32_t pgc = (uint32_t)__shfl_up((int)gk, kGroupCap);
          const unsigned long long heads =
              __ballot(hv && (lane == 0 || gk != pg || (lane >= kGroupCap && (lane & (kGroupCap - 1)) == 0 && gk == pgc)));
          const int gbase = win == 0 ? 0 : (win == 1 ? g_w0 : (win == 2 ? g_w0 + g_w1 : g_w0 + g_w1 + g_w2));
          if ((heads >> lane) & 1ull) {
            const int gi = gbase + __popcll(heads & lt_mask);
            const int fb = (int)(skey >> 23), x0 = (int)((skey >> 7) & 255u) - 2, y0 = (int)((skey >> 15) & 255u) - 2;
            const bool x0ok = x0 >= 0 && x0 < wa.npx, x1ok = x0 + 1 >= 0 && x0 + 1 < wa.npx;
            const bool y0ok = y0 >= 0 && y0 < wa.npy, y1ok = y0 + 1 >= 0 && y0 + 1 < wa.npy;
            const uint32_t ib = (uint32_t)fb * img_bytes;
            uint4 o;
            o.x = (x0ok && y0ok) ? ib + (uint32_t)(y0 * wa.npx + x0) * row_bytes : kTapOutside;
            o.y = (x1ok && y0ok) ? ib + (uint32_t)(y0 * wa.npx + x0 + 1) * row_bytes : kTapOutside;
            o.z = (x0ok && y1ok) ? ib + (uint32_t)((y0 + 1) * wa.npx + x0) * row_bytes : kTapOutside;
            o.w = (x1ok && y1ok) ? ib + (uint32_t)((y0 + 1) * wa.npx + x0 + 1) * row_bytes : kTapOutside;
            L.walk.grp_off[gi] = o;
            L.walk.grp_start[gi] = (uint16_t)(wb + lane);
          }
        }
        if (tid == 0) {
          L.walk.grp_off[G] = make_uint4(kTapOutside, kTapOutside, kTapOutside, kTapOutside);
          L.walk.grp_start[G] = (uint16_t)nh;
        }
      }
      lds_barrier();
      BT(5);

      if constexpr (BUILD) {
        write_segment(R, nh, G, kexp);
      } else {
        run_slabs(R, G, kexp);
      }
      if (!BUILD && f1 < F) {
        // Another round follows: its rows may be read by a different mover than the one that has just written them, and
        // nothing orders two waves' accesses to one address -- every mover's stores are in L2 before anybody goes on
        // (the loads are nontemporal: served by L2, never by the CU's L1).
        if (wave >= kWalkers) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        lds_barrier();
      }
      f0 = f1;
    }
    if (any && !redo) {
      float* dst = v.rgb + (int64_t)n * 3;
      dst[0] = o0; dst[1] = o1; dst[2] = o2;
      v.weight[n] = w_cur;
    }
    BT(9);
  }
  }  // BUILD || REBUILD
  BT(10);
  BT_FLUSH;
  if (stats) {
    for (int o = 32; o > 0; o >>= 1) rows_done += __shfl_xor(rows_done, o);
    if constexpr (BUILD) {  // one workgroup per brick: sharded, folded into stats[] by the walk kernel
      unsigned long long* sh = pool.ctl->acc[blockIdx.x & 63u];
      if (tid == 0 && (hits_done | rows_done)) {
        atomicAdd(&sh[0], hits_done);
        atomicAdd(&sh[1], rows_done);
      }
    } else {
      if (lane == 0 && rows_done) atomicAdd(&stats[5], rows_done);  // rows read-modify-written by this window
      if (tid == 0 && hits_done) atomicAdd(&stats[0], hits_done);
    }
  }
}

using BrickFn = void (*)(KVol, WinArgs, const WinTable*, const float*, uint32_t, unsigned long long*, unsigned int*,
                         const uint32_t*, uint32_t, const unsigned long long*, const uint32_t*, const float*, BrickPool);

// cmax[c] = bits of the largest |x| of channel c over the window's pixel-major map images, cmax[D + c / 64] = of every
// group of 64 channels, cmax[D + D / 64] = of all channels (non-negative floats order like their bit patterns; NaN > inf >
// every finite value)
__global__ __launch_bounds__(256) void chan_max_kernel(const float* __restrict__ imgs, int img_floats, int D, int P, int F,
                                                       uint32_t* __restrict__ cmax) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  uint32_t m = 0u;
  if (c < D) {
    const int f1 = min(F, (int)(blockIdx.y + 1) * 8);
    for (int f = blockIdx.y * 8; f < f1; ++f)
      for (int p = 0; p < P; ++p) m = max(m, __builtin_bit_cast(uint32_t, imgs[(size_t)f * img_floats + (size_t)p * D + c]) & 0x7fffffffu);
    if (m) atomicMax(&cmax[c], m);
  }
  uint32_t w = m;
  for (int o = 32; o > 0; o >>= 1) w = max(w, (uint32_t)__shfl_xor((int)w, o));
  if ((threadIdx.x & 63) == 0 && w && c < D) {  // (D is a multiple of 64: a wave is one group)
    atomicMax(&cmax[D + c / 64], w);
    atomicMax(&cmax[D + D / 64], w);
  }
}

// the per-frame part of the window's cameras: pose[:3, :4] and K, 21 floats per frame
__global__ __launch_bounds__(128) void cam_table_kernel(const WinTable* __restrict__ tab, int F, float* __restrict__ cams) {
  const int f = threadIdx.x;
  if (f >= F) return;
  const float* __restrict__ pose = tab->pose[f];
  const float* __restrict__ K = tab->K[f];
  float* o = cams + f * kCamFloats;
  for (int i = 0; i < 12; ++i) o[i] = pose[i];
  for (int i = 0; i < 9; ++i) o[12 + i] = K[i];
}

size_t cmax_words(int D) { return (size_t)D + (size_t)D / 64 + 1; }
size_t cmax_bytes(int D) { return (cmax_words(D) * sizeof(uint32_t) + 255) & ~(size_t)255; }

template <int CPL, bool REBUILD>
BrickFn pick_brick(bool sum, bool bf16) {
  if (bf16) return sum ? fuse_brick_kernel<CPL, true, true, false, REBUILD> : fuse_brick_kernel<CPL, false, true, false, REBUILD>;
  return sum ? fuse_brick_kernel<CPL, true, false, false, REBUILD> : fuse_brick_kernel<CPL, false, false, false, REBUILD>;
}

uint32_t brick_count(const KVol& kv) {
  return (((uint32_t)kv.nx + kBX - 1) / kBX) * (((uint32_t)kv.ny + kBY - 1) / kBY) * (((uint32_t)kv.nz + kBZ - 1) / kBZ);
}
size_t pad256(size_t x) { return (x + 255) & ~(size_t)255; }
// The aux region of the workspace: the channel maxima, then per window parity the camera table and the segment pool.  The
// lists are sized by the grid's bricks; the segments get what is left of `avail` (at most a round and a quarter per brick:
// a brick that finds the pool exhausted is built by the walk kernel, slower, never wrong).
struct AuxLayout {
  size_t cams, ctl, list, over, segs, parity_bytes;
  uint32_t cap, list_stride;
  bool fits;
};
// workgroups of the build kernel (one per brick position of the XCD-compact order, ragged edges included) per XCD = the
// length of an XCD's list
uint32_t build_wgs_per_xcd(const KVol& kv) {
  const uint32_t nbx = ((uint32_t)kv.nx + kBX - 1) / kBX, nby = ((uint32_t)kv.ny + kBY - 1) / kBY, nbz = ((uint32_t)kv.nz + kBZ - 1) / kBZ;
  const uint32_t n_tiles = ((nbx + 3u) / 4u) * ((nby + 3u) / 4u), upt = ((nbz + 3u) / 4u) * 64u;
  return ((n_tiles + 7u) / 8u) * upt;
}
size_t aux_fixed(uint32_t nb, uint32_t list_entries) {
  return pad256((size_t)kWin * kCamFloats * sizeof(float)) + pad256(sizeof(BrickCtl)) + pad256((size_t)list_entries * sizeof(uint4)) +
         pad256((size_t)nb * kOverWords * sizeof(uint32_t));
}
AuxLayout aux_layout(const KVol& kv, size_t avail) {
  AuxLayout a;
  const uint32_t nb = brick_count(kv);
  a.cams = 0;
  a.ctl = a.cams + pad256((size_t)kWin * kCamFloats * sizeof(float));
  a.list = a.ctl + pad256(sizeof(BrickCtl));
  a.list_stride = build_wgs_per_xcd(kv);
  a.over = a.list + pad256((size_t)8 * a.list_stride * sizeof(uint4));
  a.segs = a.over + pad256((size_t)nb * kOverWords * sizeof(uint32_t));
  a.fits = avail >= cmax_bytes(kv.D) + 2 * a.segs;
  const size_t per_parity = a.fits ? (avail - cmax_bytes(kv.D)) / 2 : a.segs;
  a.parity_bytes = per_parity & ~(size_t)255;
  const size_t want = (size_t)nb + nb / 4 + 64, room = (a.parity_bytes - a.segs) / kSegBytes;
  a.cap = (uint32_t)(room < want ? room : want);
  if (const char* e = getenv("SAF_BRICK_POOL_CAP")) {  // development / tests: a small pool drives bricks into the overflow list
    const long c = atol(e);
    if (c >= 0 && (uint32_t)c < a.cap) a.cap = (uint32_t)c;
  }
  return a;
}
BrickPool make_pool(const KVol& kv, void* aux, size_t aux_bytes, int parity, int split, const float** cams) {
  const AuxLayout a = aux_layout(kv, aux_bytes);
  unsigned char* p = static_cast<unsigned char*>(aux) + cmax_bytes(kv.D) + (size_t)parity * a.parity_bytes;
  BrickPool pool;
  pool.segs = p + a.segs;
  pool.list = reinterpret_cast<uint4*>(p + a.list);
  pool.list_stride = a.list_stride;
  pool.over = reinterpret_cast<uint32_t*>(p + a.over);
  pool.ctl = reinterpret_cast<BrickCtl*>(p + a.ctl);
  pool.cap = a.cap;
  pool.split = split;
  *cams = reinterpret_cast<const float*>(p + a.cams);
  return pool;
}

}  // namespace

// The brick form takes every grid shape (partial bricks are masked) and every feat_dim that is a multiple of 64.  It is
// the default for the feature widths the frame-ordered row kernel does not take (it wants whole 1 KiB pieces of a row:
// feat_dim a multiple of 256 up to 1024, of 512 for bf16) -- those used to fall back to the per-frame pipeline -- and what
// SAF_WIN_FORM=bricks (read per call) asks for; SAF_WIN_FORM=rows never uses it.
bool brick_form_ok(const KVol& kv) {
  const char* e = getenv("SAF_WIN_FORM");
  if (e && e[0] == 'r') return false;
  if (kv.D % 64 != 0 || kv.D > 8192) return false;
  const uint32_t nbx = ((uint32_t)kv.nx + kBX - 1) / kBX, nby = ((uint32_t)kv.ny + kBY - 1) / kBY, nbz = ((uint32_t)kv.nz + kBZ - 1) / kBZ;
  if (nbx >= 1024u || nby >= 1024u || nbz >= 1024u) return false;  // the brick code of a workgroup is three 10-bit fields
  if (e && e[0] == 'b') return true;
  const bool rows_take_it = kv.D % 256 == 0 && kv.D <= 1024 && (!kv.bf16 || kv.D % 512 == 0);
  return !rows_take_it;
}
// SAF_BRICK_SPLIT=0 (read per call): no build kernel, the walk kernel builds every brick itself
bool brick_split() {
  const char* e = getenv("SAF_BRICK_SPLIT");
  return !(e && e[0] == '0');
}

// What saf_fuse_workspace_bytes reserves (it knows the number of voxels, not the grid's shape): room for the lists and a full
// pool of a grid with a tenth more bricks than n_vox / 64 (ragged edges); brick_aux_fits() says whether a given grid fits.
size_t brick_aux_bytes_est(int64_t n_vox, int D) {
  const uint32_t nb = (uint32_t)(n_vox / 64 + n_vox / 640 + 4096);
  return cmax_bytes(D) + 2 * (aux_fixed(nb, 2 * nb) + ((size_t)nb + nb / 4 + 64) * kSegBytes);
}
bool brick_aux_fits(const KVol& kv, size_t avail) { return aux_layout(kv, avail).fits; }

// The build kernel of a window (after its classification, on the classification's stream): camera table, pool control
// words, one workgroup per brick.
int launch_brick_build(const KVol& kv, const WinArgs& wa, const WinTable* tab, size_t img_bytes, unsigned long long* stats,
                       const uint32_t* hitmask, uint32_t mask_plane, void* aux, size_t aux_bytes, int parity, hipStream_t s) {
  const float* cams;
  const BrickPool pool = make_pool(kv, aux, aux_bytes, parity, 1, &cams);
  if (hipMemsetAsync(pool.ctl, 0, sizeof(BrickCtl), s) != hipSuccess) return fail(SAF_E_HIP, "hipMemsetAsync(brick pool control)");
  hipLaunchKernelGGL(cam_table_kernel, dim3(1), dim3(128), 0, s, tab, wa.F, const_cast<float*>(cams));
  const uint32_t grid = 8u * build_wgs_per_xcd(kv);
  const bool sum = kv.accum == SAF_SUM;
  // (the build does not depend on the feature dtype or width: one instantiation per accumulation mode)
  BrickFn fn = sum ? fuse_brick_kernel<1, true, false, true> : fuse_brick_kernel<1, false, false, true>;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(kBuildThreads), sizeof(BuildLds), s, kv, wa, tab, nullptr, (uint32_t)img_bytes, stats, nullptr,
                     hitmask, mask_plane, nullptr, nullptr, cams, pool);
  return check_launch("fuse_brick_kernel (build)");
}

int launch_fuse_bricks(const KVol& kv, const WinArgs& wa, const WinTable* tab, const float* map_imgs, size_t img_bytes,
                       unsigned long long* stats, unsigned int* ctr, const uint32_t* hitmask, uint32_t mask_plane,
                       const unsigned long long* cls_acc, void* aux, size_t aux_bytes, int parity, int split, hipStream_t s) {
  uint32_t* cmax = static_cast<uint32_t*>(aux);
  const float* cams;
  BrickPool pool = make_pool(kv, aux, aux_bytes, parity, split, &cams);
  // the channels' largest magnitudes over this window's maps (the scales of the fixed-point sums)
  if (hipMemsetAsync(cmax, 0, cmax_bytes(kv.D), s) != hipSuccess) return fail(SAF_E_HIP, "hipMemsetAsync(channel maxima)");
  hipLaunchKernelGGL(chan_max_kernel, dim3((kv.D + 255) / 256, (wa.F + 7) / 8), dim3(256), 0, s, map_imgs,
                     (int)(img_bytes / sizeof(float)), kv.D, wa.npy * wa.npx, wa.F, cmax);
  if (!split) hipLaunchKernelGGL(cam_table_kernel, dim3(1), dim3(128), 0, s, tab, wa.F, const_cast<float*>(cams));
  const bool sum = kv.accum == SAF_SUM, bf16 = kv.bf16 != 0;
  // behind a build kernel: the walk of the pool's segments (no build code in it), then the overflow list, if any, by a small
  // launch of the instantiation that can build; without a build kernel that instantiation does everything
  BrickFn fn, fn_re;
  size_t lds;
  if (kv.D % 256 == 0) {
    fn = pick_brick<4, false>(sum, bf16); fn_re = pick_brick<4, true>(sum, bf16); lds = sizeof(BrickLds<4>);
  } else if (kv.D % 128 == 0) {
    fn = pick_brick<2, false>(sum, bf16); fn_re = pick_brick<2, true>(sum, bf16); lds = sizeof(BrickLds<2>);
  } else {
    fn = pick_brick<1, false>(sum, bf16); fn_re = pick_brick<1, true>(sum, bf16); lds = sizeof(BrickLds<1>);
  }
  for (BrickFn f : {fn, fn_re}) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e));
  }
  const int wgs_env = getenv("SAF_BRICK_WGS") ? atoi(getenv("SAF_BRICK_WGS")) : 0;
  const uint32_t grid = (uint32_t)device_cus() * (uint32_t)(wgs_env > 0 ? wgs_env : 2);
  if (split) {
    pool.split = 2;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBThreads), lds, s, kv, wa, tab, map_imgs, (uint32_t)img_bytes, stats, ctr, hitmask,
                       mask_plane, cls_acc, cmax, cams, pool);
    pool.split = 3;  // (a workgroup that finds the overflow list empty leaves at once)
    hipLaunchKernelGGL(fn_re, dim3((uint32_t)device_cus()), dim3(kBThreads), lds, s, kv, wa, tab, map_imgs, (uint32_t)img_bytes, stats, ctr,
                       hitmask, mask_plane, cls_acc, cmax, cams, pool);
  } else {
    fn = fn_re;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBThreads), lds, s, kv, wa, tab, map_imgs, (uint32_t)img_bytes, stats, ctr, hitmask,
                       mask_plane, cls_acc, cmax, cams, pool);
  }
#ifdef SAF_BRICK_TIMING
  {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(fn), kBThreads, lds);
    fprintf(stderr, "[brick timing] occupancy: %d workgroups of %d threads per CU (LDS %zu bytes each)\n", nb, kBThreads, lds);
    (void)hipStreamSynchronize(s);
    unsigned long long t[32];
    if (hipMemcpyFromSymbol(t, HIP_SYMBOL(g_brick_t), sizeof(t)) == hipSuccess) {
      const char* names[11] = {"fetch+masks", "bit matrix+prefix", "rows+scatter", "hit records|prefetch", "scalar side", "sort+table",
                               "walk|wait walk", "wait|combine", "wait combine|wait", "tail", "end"};
      for (int role = 0; role < 2; ++role) {
        unsigned long long tot = 0;
        for (int k = 0; k < 11; ++k) tot += t[role * 16 + k];
        fprintf(stderr, "[brick timing] %s:", role ? "movers " : "walkers");
        for (int k = 0; k < 11; ++k) fprintf(stderr, " %s %.1f%%", names[k], 100.0 * t[role * 16 + k] / (double)(tot ? tot : 1));
        fprintf(stderr, " (total %.3g wave-cycles)\n", (double)tot);
      }
      memset(t, 0, sizeof(t));
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_brick_t), t, sizeof(t));
      unsigned long long w[8];
      if (hipMemcpyFromSymbol(w, HIP_SYMBOL(g_walk_t), sizeof(w)) == hipSuccess) {
        const double tot = (double)(w[0] + w[1] + w[2] + w[3]);
        fprintf(stderr, "[brick timing] walk: issue %.1f%% records %.1f%% tap wait %.1f%% hits %.1f%% (%.3g wave-cycles; %llu batch-waves, %llu hit-waves)\n",
                100.0 * w[0] / tot, 100.0 * w[1] / tot, 100.0 * w[2] / tot, 100.0 * w[3] / tot, tot, w[4], w[5]);
        memset(w, 0, sizeof(w));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_walk_t), w, sizeof(w));
      }
    }
  }
#endif
  return check_launch("fuse_brick_kernel");
}

}  // namespace saf
