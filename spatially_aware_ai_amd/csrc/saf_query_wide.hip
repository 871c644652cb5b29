// saf_query_wide.hip -- many-query cosine scan on the 16-bit matrix cores (BASELINE config 5:
// a 1000-query CLIP-text scan over a 256^3 x 512 fp16 volume, the query_mesh.py path at scale).
//
// This is a dense contraction (2*N*D*Q flop, 17 TFLOP at config 5), MFMA-bound, so the layout is a
// GEMM's -- but one operand is the whole HBM-resident volume, read exactly once:
//   * a wave owns 32 feature rows and keeps them in registers for ALL queries: lane (r = lane & 31,
//     h = lane >> 5) holds A[r][16 s + 8 h + j] (j = 0..7) for every K step s -- the operand layout of
//     v_mfma_f32_32x32x{16}_{f16,bf16}; each 16-byte fragment is loaded straight from the lane's own
//     row (D = 512: 128 VGPRs per lane);
//   * the text embeddings (rounded once to the volume's dtype by a small pre-kernel) stream through a
//     double-buffered LDS tile of 32 queries; rows padded by 16 bytes so that the 16 lanes of a
//     ds_read_b128 group hit distinct bank quads; the 8 waves of a workgroup (256 feature rows) share it;
//   * per tile 32 MFMAs per wave accumulate a 32x32 fp32 block, scaled by the row's 1/norm (computed from
//     the same registers) and written out as scores in the requested dtype.
#include <type_traits>

#include "saf_common.h"
#include "saf_host.h"

namespace saf {
namespace {

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kWThreads = 512;
constexpr int kWWaves = kWThreads / 64;
constexpr int kWTile = 32;  // queries per LDS tile

// fp32 text [Q][tstride] -> 16-bit [Qpad][D] (rows >= Q zero)
template <int FT>
__global__ void text_to_16_kernel(const float* __restrict__ text, int Q, int64_t tstride, int D, int Qpad,
                                  uint16_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Qpad * D) return;
  const int q = i / D, k = i - q * D;
  const float v = q < Q ? text[(int64_t)q * tstride + k] : 0.0f;
  if (FT == SAF_BF16) {
    out[i] = (uint16_t)f32_to_bf16_bits(v);
  } else {
    const _Float16 hv = (_Float16)v;
    out[i] = __builtin_bit_cast(uint16_t, hv);
  }
}

template <int FT>
__device__ __forceinline__ float elem16_to_f32(uint16_t b) {
  if (FT == SAF_BF16) return __builtin_bit_cast(float, (uint32_t)b << 16);
  return (float)__builtin_bit_cast(_Float16, b);
}

// acc + lo^2 + hi^2 of a pair of 16-bit elements
template <int FT>
__device__ __forceinline__ float dot2_self(uint32_t w, float acc) {
  if (FT == SAF_BF16) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    const bf2_t v = __builtin_bit_cast(bf2_t, w);
    return __builtin_amdgcn_fdot2_f32_bf16(v, v, acc, false);
  }
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const h2_t v = __builtin_bit_cast(h2_t, w);
  return __builtin_amdgcn_fdot2(v, v, acc, false);
}

template <int OT>
__device__ __forceinline__ void store_score(void* __restrict__ out, int64_t idx, float v) {
  if (OT == SAF_F32) {
    static_cast<float*>(out)[idx] = v;
  } else if (OT == SAF_BF16) {
    static_cast<uint16_t*>(out)[idx] = (uint16_t)f32_to_bf16_bits(v);
  } else {
    const _Float16 hv = (_Float16)v;
    static_cast<uint16_t*>(out)[idx] = __builtin_bit_cast(uint16_t, hv);
  }
}

template <int FT>
__device__ __forceinline__ f32x16_t mfma16(const uint4& a, const uint4& b, const f32x16_t& c) {
  if (FT == SAF_BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
}

// Four consecutive scores of one feature row -> one 8-byte (16-bit out) or 16-byte (fp32 out) store.
template <int OT>
__device__ __forceinline__ void store_scores4(void* __restrict__ out, int64_t idx, float v0, float v1, float v2,
                                              float v3) {
  if (OT == SAF_F32) {
    *reinterpret_cast<float4*>(static_cast<float*>(out) + idx) = make_float4(v0, v1, v2, v3);
  } else if (OT == SAF_BF16) {
    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(out) + idx) = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
  } else {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 a, b;
    a.x = (_Float16)v0; a.y = (_Float16)v1; b.x = (_Float16)v2; b.y = (_Float16)v3;
    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(out) + idx) =
        make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
  }
}

// KS = D / 16 K steps (register-resident feature fragments: 4 VGPRs each).
//
// Operand roles: the TEXT tile is the MFMA's A operand and the FEATURE rows its B operand (both
// fragments have the same lane layout: row/col = lane & 31, k = 16 s + 8 h + j), so the accumulator
// holds C[query][feature row]: lane (r, h) owns feature row r and, in register group g = i >> 2, the four
// CONSECUTIVE queries 8 g + 4 h + (i & 3).  The epilogue is then per lane: its own row's 1/norm (no
// shuffles) and one vector store of 4 scores per group.
template <int FT, int OT, int KS>
__global__ __launch_bounds__(kWThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void query_wide_kernel(const uint16_t* __restrict__ feats, int64_t n_rows,
                                                                int64_t fstride, const uint16_t* __restrict__ text16,
                                                                int Q, int Qpad, float scale, int normalize,
                                                                void* __restrict__ out, int64_t ostride) {
  constexpr int D = KS * 16;
  constexpr int ROWB = D * 2 + 16;  // padded LDS row in bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char s_tiles[];  // 2 x [32][ROWB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_qt = Qpad / kWTile;
  const int64_t n_blocks = (n_rows + 32 * kWWaves - 1) / (32 * kWWaves);
  // cooperative tile copy: 32 rows x D*2 bytes = 32 * (D/8) 16-byte pieces over 512 threads
  constexpr int PIECES = kWTile * (D / 8);
  constexpr int PPT = (PIECES + kWThreads - 1) / kWThreads;  // pieces per thread (4 at D = 512)
  const bool vec_ok = (OT == SAF_F32 ? (ostride % 4 == 0) : (ostride % 4 == 0)) &&
                      (((uintptr_t)out & 15) == 0);
  for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int64_t row0 = (blk * kWWaves + wave) * 32;
    const int64_t row_true = row0 + r;
    const int64_t row = row_true < n_rows ? row_true : n_rows - 1;  // padded lanes recompute the last row
    const uint16_t* arow = feats + row * fstride + 8 * h;
    // the wave's 32 rows, register resident for every query tile
    uint4 a[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) a[s] = *reinterpret_cast<const uint4*>(arow + 16 * s);
    float inv = scale;
    if (normalize) {
      float ss = 0.0f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const uint32_t w[4] = {a[s].x, a[s].y, a[s].z, a[s].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = elem16_to_f32<FT>((uint16_t)(w[j] & 0xffffu)), hi = elem16_to_f32<FT>((uint16_t)(w[j] >> 16));
          ss = __builtin_fmaf(lo, lo, ss);
          ss = __builtin_fmaf(hi, hi, ss);
        }
      }
      ss += __shfl_xor(ss, 32);                              // both halves of row r
      // SAF_NORM_L2: nan_to_num, an all-zero row scores 0; SAF_NORM_L2_CLAMP: norm.clamp_min(0.1)
      inv = normalize == SAF_NORM_L2_CLAMP ? scale / fmaxf(sqrtf(ss), 0.1f) : (ss > 0.0f ? scale / sqrtf(ss) : 0.0f);
    }

    __syncthreads();  // the previous row block is done with both LDS buffers
    for (int p = tid; p < PIECES; p += kWThreads) {
      const int q = p / (D / 8), c = p - q * (D / 8);
      *reinterpret_cast<uint4*>(s_tiles + q * ROWB + c * 16) =
          *reinterpret_cast<const uint4*>(text16 + (int64_t)q * D + c * 8);
    }
    __syncthreads();
    for (int qt = 0; qt < n_qt; ++qt) {
      const unsigned char* cur = s_tiles + (size_t)(qt & 1) * kWTile * ROWB;
      unsigned char* nxt = s_tiles + (size_t)((qt + 1) & 1) * kWTile * ROWB;
      // Next tile: at D = 512 a text row is exactly one 1 KiB LDS-DMA piece (global_load_lds_dwordx4:
      // 64 lanes x 16 B, written linearly from a wave-uniform LDS base), so each wave copies 4 rows
      // straight into the other buffer with no registers; it lands while this tile's MFMAs run and
      // is covered by the barrier's vmcnt(0).  Narrower rows are staged through registers.
      constexpr bool kDma = (D == 512);
      typedef unsigned int w1_u4 __attribute__((ext_vector_type(4)));  // (native: as HIP's uint4 the conditionally loaded pieces lived in scratch)
      w1_u4 stage[kDma ? 1 : PPT];
      const bool more = qt + 1 < n_qt;
      if (more) {
        if (kDma) {
#pragma unroll
          for (int k = 0; k < kWTile / kWWaves; ++k) {
            const int q = wave * (kWTile / kWWaves) + k;
            const uint16_t* src = text16 + (int64_t)((qt + 1) * kWTile + q) * D + lane * 8;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)src,
                (__attribute__((address_space(3))) void*)(nxt + q * ROWB), 16, 0, 0);
          }
        } else {
#pragma unroll
          for (int k = 0; k < PPT; ++k) {
            const int p = tid + k * kWThreads;
            if (p < PIECES) {
              const int q = p / (D / 8), c = p - q * (D / 8);
              stage[k] = *reinterpret_cast<const w1_u4*>(text16 + (int64_t)((qt + 1) * kWTile + q) * D + c * 8);
            }
          }
        }
      }
      f32x16_t acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
      const unsigned char* trow = cur + r * ROWB + 16 * h;  // text row (query qt*32 + r), k half h
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const uint4 t = *reinterpret_cast<const uint4*>(trow + 32 * s);
        acc = mfma16<FT>(t, a[s], acc);  // C[query][feature row]
      }
      if (more && !kDma) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          const int p = tid + k * kWThreads;
          if (p < PIECES) {
            const int q = p / (D / 8), c = p - q * (D / 8);
            *reinterpret_cast<w1_u4*>(nxt + q * ROWB + c * 16) = stage[k];
          }
        }
      }
      // The barrier comes BEFORE this tile's output stores: what it waits for (vmcnt counts loads and
      // stores in issue order) is then the LDS-DMA of the next tile plus the PREVIOUS tile's stores, which
      // have had a whole MFMA phase to drain; this tile's stores overlap the next tile's MFMAs.
      if (kDma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // tile qt consumed by every wave; tile qt+1 fully written
      if (row_true < n_rows) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int q0 = qt * kWTile + 8 * g + 4 * h;
          const int64_t idx = row_true * ostride + q0;
          const float v0 = acc[4 * g] * inv, v1 = acc[4 * g + 1] * inv, v2 = acc[4 * g + 2] * inv,
                      v3 = acc[4 * g + 3] * inv;
          if (vec_ok && q0 + 3 < Q) {
            store_scores4<OT>(out, idx, v0, v1, v2, v3);
          } else {
            if (q0 < Q) store_score<OT>(out, idx, v0);
            if (q0 + 1 < Q) store_score<OT>(out, idx + 1, v1);
            if (q0 + 2 < Q) store_score<OT>(out, idx + 2, v2);
            if (q0 + 3 < Q) store_score<OT>(out, idx + 3, v3);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// v2: 64 feature rows per wave at ONE wave per SIMD (the whole 512-register file), fused epilogues.
//
// v1 above spends one ds_read_b128 of text per MFMA and two waves per SIMD share the matrix pipe, meeting at a
// barrier every 32 MFMAs.  Here a wave keeps TWO 32-row fragments of the volume in registers (256 VGPRs at
// D = 512), so every text fragment read from LDS feeds two MFMAs and a tile of 32 queries is 64 back-to-back
// MFMAs per wave between barriers; the query tiles run as one continuous double-buffered sequence across row
// blocks (the next block's first tile is already in flight during the last tile of this one).
//
// Epilogues (saf_wide_epilogue): SCORES writes the N x Q scores (16-bit outputs as 16-byte stores after a
// half-wave exchange); VS_BACKGROUND turns every target column into softmax([backgrounds..., target])[-1]
// (query_mesh.py:36-39, hypersim_eval.py:76-81) from a per-row log-sum-exp of the background scores; ROW_ARGMAX
// keeps only the best query and its score per row (eval_scannet_segmentation.py:553-560, first label of the
// argsort); QUERY_MAX keeps only the best row and its score per query.  The last two write no N x Q output.
// ------------------------------------------------------------------------------------------------------------

typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream_u4(const uint16_t* p) {  // the volume is read once: keep it out of the caches' way
  const v4u_t v = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// One 1 KiB LDS-DMA piece (64 lanes x 16 bytes, written linearly from the wave-uniform LDS byte address `lds_addr`), issued
// from inline assembly so that the COMPILER DOES NOT KNOW an LDS write is pending: for the builtin it orders every later
// ds_read behind the transfer (no alias information: `s_waitcnt vmcnt(0)` right after the issue -- rounds 2-3 paid the
// whole L2 latency of the next tile at the start of every step: the "3.3 ms of LDS-DMA" of DESIGN 4.4).  The ordering is
// this file's own: a counted `s_waitcnt vmcnt` ahead of the barrier behind which the tile is read (w2_wait_vm).  An
// operation the compiler's counter does not know about only makes its own waits longer, never shorter.
#ifndef SAF_W2_ASMDMA
#define SAF_W2_ASMDMA 1
#endif
// K consecutive text rows (1 KiB each: D = 512) from sbase (wave-uniform: SGPRs) + the lane's 16 bytes into K consecutive padded
// LDS rows (ROWB bytes apart) from the wave-uniform byte address lds_addr.  One statement: M0 and the lane offset step from
// piece to piece (a VALU instruction sits between every write of M0 and the transfer that reads it).
#define SAF_W2_DMA_NEXT "s_add_u32 m0, m0, %5\n\tv_add_u32 %1, 0x400, %1\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
template <int K, int ROWB>
__device__ __forceinline__ void w2_dma_rows(const uint16_t* sbase, uint32_t lane16, uint32_t lds_addr) {
  static_assert(K == 4 || K == 8, "pieces per wave and tile");
#if SAF_W2_ASMDMA
  uint32_t m0_saved, voff;  // (M0 is the compiler's: handed back as found)
  if (K == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\tv_mov_b32 %1, %2\n\tglobal_load_lds_dwordx4 %1, %3\n\t" SAF_W2_DMA_NEXT SAF_W2_DMA_NEXT
                 SAF_W2_DMA_NEXT "s_nop 0\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved), "=&v"(voff) : "v"(lane16), "s"(sbase), "s"(lds_addr), "n"(ROWB) : "memory", "scc");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\tv_mov_b32 %1, %2\n\tglobal_load_lds_dwordx4 %1, %3\n\t" SAF_W2_DMA_NEXT SAF_W2_DMA_NEXT
                 SAF_W2_DMA_NEXT SAF_W2_DMA_NEXT SAF_W2_DMA_NEXT SAF_W2_DMA_NEXT SAF_W2_DMA_NEXT "s_nop 0\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved), "=&v"(voff) : "v"(lane16), "s"(sbase), "s"(lds_addr), "n"(ROWB) : "memory", "scc");
#else
#pragma unroll
  for (int k = 0; k < K; ++k)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(sbase) + lane16 + k * 1024),
                                     (__attribute__((address_space(3))) void*)(uintptr_t)(lds_addr + k * ROWB), 16, 0, 0);
#endif
}
// One piece with a per-lane byte offset from a wave-uniform base (the next row block's rows: see query_wide3_kernel).
__device__ __forceinline__ void w3_dma_piece(const uint16_t* sbase, uint32_t lane_off, uint32_t lds_addr) {
  uint32_t m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_nop 0\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved) : "v"(lane_off), "s"(sbase), "s"(lds_addr) : "memory");
}
// `s_waitcnt vmcnt(n)` for a wave-uniform n known only at run time (the instruction takes an immediate): the largest listed
// count <= n -- waiting for more than asked is always correct.
__device__ __forceinline__ void w2_wait_vm(int n) {
  if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n >= 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// float -> u32 that orders like the float (for integer atomic max)
__device__ __forceinline__ uint32_t ordered_bits(float f) {
  const uint32_t b = __builtin_bit_cast(uint32_t, f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t o) {
  return __builtin_bit_cast(float, (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// 8 consecutive 16-bit scores per lane from the two half-waves' groups g (even) and g + 1: after the exchange
// lanes 0-31 hold queries 8 g .. 8 g + 7 and lanes 32-63 queries 8 g + 8 .. 8 g + 15 of their row (T21).
template <int OT>
__device__ __forceinline__ uint2 pack4_16(float v0, float v1, float v2, float v3) {
  if (OT == SAF_BF16) return make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 a, b;
  a.x = (_Float16)v0; a.y = (_Float16)v1; b.x = (_Float16)v2; b.y = (_Float16)v3;
  return make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
}

// Development (-DSAF_W2_STAMP): s_memtime stamps around the segments of a step, summed per wave of workgroup 0 and read
// back by saf_debug_w2_stamps (tools/w2_stamps.py).  Costs about a tenth of the kernel's time.
#ifdef SAF_W2_STAMP
__device__ unsigned long long g_w2_stamp[8 * 8];
#define W2_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[k] += t_ - t_last; t_last = t_; } while (0)
#else
#define W2_STAMP(k) do { } while (0)
#endif

struct Wide2Args {
  const uint16_t* feats;
  int64_t n_rows, fstride;
  const uint16_t* text16;  // [Qpad][D]; VS_BACKGROUND: tile 0 = backgrounds (zero padded), targets from tile 1
  int Q, Qpad;             // Q = columns that exist (VS_BACKGROUND: 32 + number of targets)
  float scale;
  int normalize, n_bg, flags;
  void* out;               // SCORES / VS_BACKGROUND
  int64_t ostride;
  int32_t* out_index;      // ROW_ARGMAX [n_rows]
  float* out_value;        // ROW_ARGMAX [n_rows]
  unsigned long long* qkeys;  // QUERY_MAX [Qpad]: ordered(score) << 32 | low 32 bits of ~row (smaller row wins ties)
  int64_t row_offset;
  int safe_wait;  // SAF_W2_SAFE_WAIT=1 (read per call): vmcnt(0) in front of every tile's barrier instead of the counted wait
};

// What the epilogue of a tile needs to know about the tile (it runs one step later, beside the next tile's MFMAs,
// possibly after the workgroup has moved on to its next row block).  NF = 32-row fragments per wave.
template <int NF>
struct W2Tile {
  float inv[NF];     // scale / row norm of the lane's feature rows
  int64_t row[NF];   // their true indices (may be >= n_rows on the last block)
  int qt;            // query tile
};
// per-row-block epilogue state (ROW_ARGMAX: running best; VS_BACKGROUND: log-sum-exp of the backgrounds)
template <int NF>
struct W2State {
  float best_v[NF], lse[NF];
  int best_q[NF];
  const float* inv_lds;  // QUERY_MAX: the wave's 32 NF (scale / norm) values in LDS, by row of the wave (read four at a time: the
                         // rows 8 g + 4 h + 0..3 behind accumulator registers 4 g .. 4 g + 3 of this lane)
};

template <int OT, int EPI, int NF>
__device__ __forceinline__ void w2_epilogue(const Wide2Args& wa, const f32x16_t (&c)[NF], const W2Tile<NF>& t, W2State<NF>& st,
                                            int r, int h, int n_qt, bool vec_ok) {
  // lane (r, h): feature row t.row[f] in fragment f; register 4 g + i holds query qt*32 + 8 g + 4 h + i
  const int qbase = t.qt * kWTile + 4 * h;
  if (EPI == SAF_QW_SCORES || EPI == SAF_QW_VS_BACKGROUND) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = c[f][i] * t.inv[f];
      int col0 = qbase;  // output column of register 0
      bool write = true;
      if (EPI == SAF_QW_VS_BACKGROUND) {
        if (t.qt == 0) {  // tile 0 holds the backgrounds: per-row log-sum-exp of their scaled scores, no output
          float m = -INFINITY;
#pragma unroll
          for (int i = 0; i < 16; ++i) m = (8 * (i >> 2) + 4 * h + (i & 3) < wa.n_bg) ? fmaxf(m, v[i]) : m;
          m = fmaxf(m, __shfl_xor(m, 32));
          float e = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) e += (8 * (i >> 2) + 4 * h + (i & 3) < wa.n_bg) ? __expf(v[i] - m) : 0.f;
          e += __shfl_xor(e, 32);
          st.lse[f] = m + __logf(e);
          write = false;
        } else {
          // softmax([bg..., target])[-1] = 1 / (1 + exp(lse_bg - z_target))
          const bool rescale = (wa.flags & 1) != 0;  // query_mesh.py:39: ((r - 0.5) * 2).clamp(0, 1)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            // (the same arithmetic as w2_fast_piece: interior and edge tiles of one output agree bit for bit)
            const float p = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(c[f][i], -t.inv[f] * 1.4426950408889634f,
                                                                                                  st.lse[f] * 1.4426950408889634f)));
            v[i] = __builtin_amdgcn_fmed3f(__builtin_fmaf(p, rescale ? 2.0f : 1.0f, rescale ? -1.0f : 0.0f), 0.0f, 1.0f);
          }
          col0 = qbase - kWTile;
        }
      }
      if (write) {
        const int ncols = EPI == SAF_QW_VS_BACKGROUND ? wa.Q - kWTile : wa.Q;
        const int64_t row = t.row[f];
        if (OT == SAF_F32) {
          if (row < wa.n_rows) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int q0 = col0 + 8 * g;
              float* o = static_cast<float*>(wa.out) + row * wa.ostride + q0;
              if (vec_ok && q0 + 3 < ncols) {
                *reinterpret_cast<float4*>(o) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
              } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  if (q0 + i < ncols) o[i] = v[4 * g + i];
              }
            }
          }
        } else {
          // pairs of groups (g, g + 1): after the half-wave exchange lanes 0-31 hold columns 16 gp .. + 7 and
          // lanes 32-63 columns 16 gp + 8 .. + 15 of their row: one 16-byte store each
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            uint2 lo = pack4_16<OT>(v[8 * gp], v[8 * gp + 1], v[8 * gp + 2], v[8 * gp + 3]);       // group 2 gp
            uint2 hi = pack4_16<OT>(v[8 * gp + 4], v[8 * gp + 5], v[8 * gp + 6], v[8 * gp + 7]);   // group 2 gp + 1
            auto rx = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
            auto ry = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
            // lanes 0-31: [own group 2gp | upper half's group 2gp]; lanes 32-63: [lower half's group 2gp+1 | own]
            const uint4 w = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            const int qv = (col0 - 4 * h) + 16 * gp + 8 * h;  // first of this lane's 8 columns
            if (row < wa.n_rows) {
              uint16_t* o = static_cast<uint16_t*>(wa.out) + row * wa.ostride + qv;
              if (vec_ok && qv + 7 < ncols) {
                *reinterpret_cast<uint4*>(o) = w;
              } else {
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int i = 0; i < 8; ++i)
                  if (qv + i < ncols) o[i] = (uint16_t)(ww[i >> 1] >> (16 * (i & 1)));
              }
            }
          }
        }
      }
    }
  } else if (EPI == SAF_QW_ROW_ARGMAX) {
    const bool last = t.qt == n_qt - 1;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      float bv = t.qt == 0 ? -INFINITY : st.best_v[f];
      int bq = t.qt == 0 ? 0 : st.best_q[f];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int q = qbase + 8 * (i >> 2) + (i & 3);
        // the RAW dot products are compared: a row's scale / norm is one non-negative factor for all of its queries (a negative
        // `scale` arrives as negated text, see the host side), multiplied in once, into the row's best
        float x = c[f][i];
        if (last && q >= wa.Q) x = -INFINITY;  // zero-padded text rows are no candidates
        if (x > bv) { bv = x; bq = q; }        // queries ascend: the first maximum stays
      }
      if (last) {
        // both halves of a row: the larger score, the smaller query on ties
        const float ov = __shfl_xor(bv, 32);
        const int oq = __shfl_xor(bq, 32);
        if (ov > bv || (ov == bv && oq < bq)) { bv = ov; bq = oq; }
        if (h == 0 && t.row[f] < wa.n_rows) { wa.out_index[t.row[f]] = bq; wa.out_value[t.row[f]] = bv * t.inv[f]; }
      }
      st.best_v[f] = bv;
      st.best_q[f] = bq;
    }
  } else {  // SAF_QW_QUERY_MAX.  The operands are SWAPPED for this epilogue (C[feature row][query]): lane (n, h) owns query
    // qt*32 + n and, in register i of fragment f, feature row 32 f + 8 (i >> 2) + 4 h + (i & 3) of the wave -- the maximum over
    // rows is a per-lane chain over the registers, the two halves meet once, and one atomic instruction per tile and wave
    // carries 32 queries (st.invc: the rows' 1/norm in this layout).
    const int q = t.qt * kWTile + r;
    const int64_t base = t.row[0] - r;  // the wave's first row
    float bv = -INFINITY;
    int bm = 0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = 32 * f + 8 * (i >> 2) + 4 * h + (i & 3);
        const float x = base + m < wa.n_rows ? c[f][i] * st.inv_lds[m] : -INFINITY;
        if (x > bv) { bv = x; bm = m; }  // rows ascend: the first maximum stays
      }
    }
    unsigned long long key = bv > -INFINITY ? ((unsigned long long)ordered_bits(bv) << 32) | (uint32_t) ~(uint32_t)(base + bm + wa.row_offset) : 0ull;
    const unsigned long long other = __shfl_xor(key, 32);
    key = other > key ? other : key;
    if (h == 0 && q < wa.Q && key) atomicMax(&wa.qkeys[q], key);
  }
}

// The same epilogue cut into pieces that sit BETWEEN the MFMAs of the next tile.  A wave issues in order: a block of
// epilogue instructions in front of the MFMA block costs its full issue time on top of the matrix pipe's, while six
// vector instructions fit in the shadow of every v_mfma_f32_32x32x16 (8 of its 32 cycles hold the issue port).  So for
// interior tiles -- every row of the wave valid, aligned output, all 32 columns real: a wave-uniform condition -- the
// previous tile's accumulator register i is finished right after MFMA step i * KS / 16 of the current tile, branch-free.
// `w2_fast_piece(i)` does register i of every fragment; stores go out after registers 7 and 15 (8 scores per lane each).
template <int NF>
struct W2Fast {
  float v[NF][16];                 // scaled scores / probabilities of the previous tile (SCORES, VS_BACKGROUND)
  float iv[NF][4];                // QUERY_MAX: 1/norm of the rows behind the current group of four registers
  float ka[NF], kb[NF];           // VS_BACKGROUND: the row's -inv log2(e) and lse log2(e)
  float bv; int bm;               // QUERY_MAX: the lane's best score of the tile so far and its row (without the 4 h of the half)
};

template <int OT, int EPI, int NF>
__device__ __forceinline__ void w2_fast_piece(int i, const Wide2Args& wa, const f32x16_t (&c)[NF], const W2Tile<NF>& t,
                                              W2State<NF>& st, W2Fast<NF>& fs, int r, int h, int n_qt) {
  const int qbase = t.qt * kWTile + 4 * h;
  if (EPI == SAF_QW_SCORES || EPI == SAF_QW_VS_BACKGROUND) {
    const bool rescale = (wa.flags & 1) != 0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      float x;
      if (EPI == SAF_QW_VS_BACKGROUND) {
        // 1 / (1 + exp(lse - z)), z = c * inv, as 1 / (1 + 2^(c ka + kb)) with the row's ka = -inv log2(e), kb = lse log2(e);
        // query_mesh.py:39's ((p - 0.5) * 2).clamp(0, 1) is clamp(2 p - 1): one FMA whose constants the flag picks (p itself
        // lies in (0, 1]: the clamp is harmless without the rescale).  Six vector instructions per score instead of eleven.
        if (i == 0) {
          fs.ka[f] = -t.inv[f] * 1.4426950408889634f;
          fs.kb[f] = st.lse[f] * 1.4426950408889634f;
        }
        const float p = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(c[f][i], fs.ka[f], fs.kb[f])));
        x = __builtin_amdgcn_fmed3f(__builtin_fmaf(p, rescale ? 2.0f : 1.0f, rescale ? -1.0f : 0.0f), 0.0f, 1.0f);
      } else {
        x = c[f][i] * t.inv[f];
      }
      fs.v[f][i] = x;
      if ((i & 7) == 7) {  // registers 8 gp .. 8 gp + 7 are complete: this lane's 8 columns of the pair of groups gp
        const int gp = i >> 3;
        const int col0 = EPI == SAF_QW_VS_BACKGROUND ? qbase - kWTile : qbase;
        const float* v = fs.v[f];
        if (OT == SAF_F32) {
          float* o = static_cast<float*>(wa.out) + t.row[f] * wa.ostride + col0 + 16 * gp;
          *reinterpret_cast<float4*>(o) = make_float4(v[8 * gp], v[8 * gp + 1], v[8 * gp + 2], v[8 * gp + 3]);
          *reinterpret_cast<float4*>(o + 8) = make_float4(v[8 * gp + 4], v[8 * gp + 5], v[8 * gp + 6], v[8 * gp + 7]);
        } else {
          uint2 lo = pack4_16<OT>(v[8 * gp], v[8 * gp + 1], v[8 * gp + 2], v[8 * gp + 3]);
          uint2 hi = pack4_16<OT>(v[8 * gp + 4], v[8 * gp + 5], v[8 * gp + 6], v[8 * gp + 7]);
          auto rx = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
          auto ry = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
          // (tried: the tile transposed through a per-wave LDS image so that four lanes write the 64 contiguous bytes a
          //  row has in this tile -- two fully coalesced stores per tile instead of these: heat maps 30.3 vs 27.4 ms)
          uint16_t* o = static_cast<uint16_t*>(wa.out) + t.row[f] * wa.ostride + (col0 - 4 * h) + 16 * gp + 8 * h;
          *reinterpret_cast<uint4*>(o) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        }
      }
    }
  } else if (EPI == SAF_QW_ROW_ARGMAX) {
    const int q = qbase + 8 * (i >> 2) + (i & 3);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const float x = c[f][i];               // raw dot products: see w2_epilogue
      const bool better = x > st.best_v[f];  // queries ascend: the first maximum stays
      st.best_v[f] = better ? x : st.best_v[f];
      st.best_q[f] = better ? q : st.best_q[f];
    }
  } else {  // SAF_QW_QUERY_MAX (swapped operands: see w2_epilogue): the lane's chain over its rows
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int m = 32 * f + 8 * (i >> 2) + (i & 3);
      if ((i & 3) == 0) {
        const float4 v4 = *reinterpret_cast<const float4*>(st.inv_lds + m + 4 * h);
        fs.iv[f][0] = v4.x; fs.iv[f][1] = v4.y; fs.iv[f][2] = v4.z; fs.iv[f][3] = v4.w;
      }
      const float x = c[f][i] * fs.iv[f][i & 3];
      if (f == 0 && i == 0) {
        fs.bv = x; fs.bm = m;
      } else {
        const bool better = x > fs.bv;  // rows ascend: the first maximum stays
        fs.bv = better ? x : fs.bv;
        fs.bm = better ? m : fs.bm;
      }
    }
  }
}

// NF = 2: 64 rows per wave, 4 waves per workgroup, one wave per SIMD (the whole 512-register file);
// NF = 1: 32 rows per wave, 8 waves per workgroup, two waves per SIMD (the two interleave on the matrix pipe by themselves).
// TH = threads per workgroup: 512 (one workgroup per CU) or, for NF = 1, 256 (two independent workgroups per CU: while
// one waits at its barrier or loads its next rows, the other one has the matrix pipes).
template <int FT, int OT, int KS, int EPI, int NF, int TH>
__global__ __launch_bounds__(TH) __attribute__((amdgpu_waves_per_eu(NF == 2 ? 1 : 2, NF == 2 ? 1 : 2))) void
query_wide2_kernel(Wide2Args wa) {
  constexpr int kThreads = TH, kWaves = kThreads / 64, kRows = 32 * NF;
  constexpr int D = KS * 16;
  constexpr int ROWB = D * 2 + 16;  // padded LDS row in bytes: the 16 lanes of a ds_read_b128 group hit distinct bank quads
  constexpr bool kDma = (D == 512);  // a text row is exactly one 1 KiB LDS-DMA piece
  extern __shared__ __attribute__((aligned(16))) unsigned char s_tiles[];  // 2 x [32][ROWB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_qt = wa.Qpad / kWTile;
  const int64_t rows_per_wg = (int64_t)kWaves * kRows;
  const int64_t n_blocks = (wa.n_rows + rows_per_wg - 1) / rows_per_wg;
  const int64_t my_blocks = blockIdx.x < n_blocks ? (n_blocks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t n_steps = my_blocks * n_qt;  // (row block, query tile) pairs of this workgroup, in order
  if (n_steps == 0) return;
  constexpr int PIECES = kWTile * (D / 8);
  constexpr int PPT = (PIECES + kThreads - 1) / kThreads;
  const bool vec_ok = (wa.ostride % 8 == 0) && (((uintptr_t)wa.out & 15) == 0);

  for (int p = tid; p < PIECES; p += kThreads) {  // tile 0 of the first block
    const int q = p / (D / 8), c = p - q * (D / 8);
    *reinterpret_cast<uint4*>(s_tiles + q * ROWB + c * 16) = *reinterpret_cast<const uint4*>(wa.text16 + (int64_t)q * D + c * 8);
  }

  uint4 a[NF][KS];
  W2Tile<NF> cur, prev;
  W2State<NF> st;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    st.best_v[f] = -INFINITY; st.best_q[f] = 0; st.lse[f] = 0.f;
    cur.inv[f] = 0.f; cur.row[f] = 0;
  }
  float* s_inv = reinterpret_cast<float*>(s_tiles + 2 * kWTile * ROWB) + wave * kRows;  // QUERY_MAX only (W2State::inv_lds)
  st.inv_lds = s_inv;
  cur.qt = 0;
  prev = cur;
  f32x16_t acc[2][NF];  // [step parity][fragment]: the tile being accumulated and the previous one awaiting its epilogue
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)s_tiles;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int tail_ops = 0;  // vector-memory operations the last step issued behind its LDS-DMA (a lower bound)

  // One step = one (row block, query tile) pair.  In program order: barrier (tile in LDS) -> transfer of the next tile
  // issued -> the first text fragments requested -> the PREVIOUS tile's epilogue (vector work and stores that do not
  // depend on what follows) -> the MFMAs, each group consuming one fragment and requesting the one AHEAD steps ahead.
  int qt_run = 0;
  int64_t blk_run = blockIdx.x;
#ifdef SAF_W2_STAMP
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
  auto step_body = [&](int64_t step, f32x16_t (&c)[NF], const f32x16_t (&pc)[NF]) {
    const int qt = qt_run;  // step % n_qt, kept incrementally (the 64-bit division cost ~250 cycles of every step: tools/w2_stamps.py)
    qt_run = qt + 1 < n_qt ? qt + 1 : 0;
    W2_STAMP(0);  // the tail of the step before: last epilogue pieces, loop control
    if (qt == 0) {
      // A new row block: its rows per wave, register resident for every query tile.  A 32-row fragment per wave-load
      // touches 32 rows x 32 bytes (each lane holds 16 bytes of its OWN row: the MFMA operand layout), and nothing runs
      // beside this phase: 3.7 of 19.4 ms at config 5.  Tried and measured slower: requesting the next block's rows
      // behind the last tile's MFMAs (a load into a register an MFMA is still reading stalls the in-order issue: 20.8 ms),
      // staggered workgroup phases, two independent workgroups per CU.
      const int64_t blk = blk_run;
      blk_run += gridDim.x;
      const int64_t row0 = (blk * kWaves + wave) * kRows;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        cur.row[f] = row0 + 32 * f + r;
        const int64_t rr = cur.row[f] < wa.n_rows ? cur.row[f] : wa.n_rows - 1;  // padded lanes recompute the last row
        const uint16_t* pa = wa.feats + rr * wa.fstride + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[f][s] = ld_stream_u4(pa + 16 * s);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        cur.inv[f] = wa.scale;
        if (wa.normalize) {
          // the row's squared norm: one v_dot2_f32_{f16,bf16} per pair of elements (exact products, fp32 sums) -- a quarter of the
          // instructions of widening and two FMAs per pair; two chains, added at the end
          float ss = 0.f, ss2 = 0.f;
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const uint32_t w[4] = {a[f][s].x, a[f][s].y, a[f][s].z, a[f][s].w};
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
              ss = dot2_self<FT>(w[j], ss);
              ss2 = dot2_self<FT>(w[j + 1], ss2);
            }
          }
          ss += ss2;
          ss += __shfl_xor(ss, 32);
          // SAF_NORM_L2_CLAMP: norm.clamp_min(0.1); SAF_NORM_L2 with nan_to_num: an all-zero row scores 0
          cur.inv[f] = wa.normalize == SAF_NORM_L2_CLAMP ? wa.scale / fmaxf(sqrtf(ss), 0.1f)
                                                         : (ss > 0.0f ? wa.scale / sqrtf(ss) : 0.0f);
        }
      }
    }
    cur.qt = qt;
    if (qt == 0) W2_STAMP(1);  // a new block's rows: issue, norms (the waits for the data are in the first tile's MFMAs)
    const unsigned char* curb = s_tiles + (size_t)(step & 1) * kWTile * ROWB;
    unsigned char* nxt = s_tiles + (size_t)((step + 1) & 1) * kWTile * ROWB;
    // This tile's text is in LDS (its transfer was issued a step ago) and every wave is done reading the other buffer.
    // The wait is COUNTED: the wave's `tail_ops` youngest vector-memory operations are the previous tile's last stores (or
    // its atomic), issued behind the transfer -- they need not have landed (vmcnt retires in issue order).
#if SAF_W2_ASMDMA
    // (the count is this file's claim about what the compiler emits behind the transfer: SAF_W2_SAFE_WAIT=1 drains instead,
    //  and tests/test_gpu_parity.py::test_wide_scan_counted_wait_equals_the_draining_wait compares the two bit for bit)
    w2_wait_vm(kDma && !wa.safe_wait ? tail_ops : 0);
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    W2_STAMP(7);  // the wait for this wave's pieces of the tile
    __syncthreads();
    W2_STAMP(2);  // the barrier  // the wait for the tile and the barrier
    const bool more = step + 1 < n_steps;
    const int qt_next = qt + 1 < n_qt ? qt + 1 : 0;
    typedef unsigned int w2_u4 __attribute__((ext_vector_type(4)));  // (a native vector: HIP's uint4 -- a struct around a union -- stayed in scratch memory here)
    w2_u4 stage[kDma ? 1 : PPT];
    // the next tile by LDS-DMA, 32 / kWaves pieces per wave, ahead of the tile's first MFMA.  (Tried: of the two waves that
    // share a SIMD one issuing here and the other one behind the first half of its MFMAs -- the second site splits the MFMA
    // loop: heat maps 24.8 vs 23.3 ms --; every wave issuing behind its 8th / 16th MFMA: 17.4 / 17.8 vs 17.7 ms for the row
    // argmax, 23.2 / 24.3 vs 23.1 for the heat maps.)
    auto dma_next = [&]() {
      constexpr int K = kWTile / kWaves;
      w2_dma_rows<K, ROWB>(wa.text16 + (int64_t)(qt_next * kWTile + wave_u * K) * D, (uint32_t)lane * 16u,
                           lds_base + (uint32_t)((step + 1) & 1) * kWTile * ROWB + wave_u * K * ROWB);
    };
    if (more) {
      if (kDma) {
        dma_next();
      } else {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          const int p = tid + k * kThreads;
          if (p < PIECES) {
            const int q = p / (D / 8), cc = p - q * (D / 8);
            stage[k] = *reinterpret_cast<const w2_u4*>(wa.text16 + (int64_t)(qt_next * kWTile + q) * D + cc * 8);
          }
        }
      }
    }
    W2_STAMP(3);  // the issue of the next tile's transfer
#ifndef SAF_W2_AHEAD
#define SAF_W2_AHEAD 6  // text fragments requested ahead of the MFMA that consumes them.  8 spilled 5-18 registers in the NF = 1, D = 512 instantiations (scratch traffic inside the tile loop: VERDICT round 3); 6: none
#endif
    constexpr int kAhead = EPI == SAF_QW_QUERY_MAX && NF == 1 ? SAF_W2_AHEAD - 2 : SAF_W2_AHEAD;  // (the per-query maximum's chain state; -1: 3 spilled registers)
    constexpr int AHEAD = KS < kAhead ? KS : kAhead;
    uint4 t[KS];
    // (tried: fragments straight from a fragment-ordered copy of the text in L2 / L1, no LDS, no barrier: 24.0 vs 19.3 ms)
    const unsigned char* trow = curb + r * ROWB + 16 * h;  // text row (query qt*32 + r), k half h
#define SAF_W2_TOFF(s) (32 * (s))
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) t[s] = *reinterpret_cast<const uint4*>(trow + SAF_W2_TOFF(s));
    // Is the previous tile an interior one (wave-uniform)?  Then its epilogue rides between this tile's MFMAs.
    bool fast = step > 0;
#pragma unroll
    for (int f = 0; f < NF; ++f) fast = fast && __all(prev.row[f] < wa.n_rows);
    if (EPI == SAF_QW_SCORES) fast = fast && vec_ok && (prev.qt + 1) * kWTile <= wa.Q;
    if (EPI == SAF_QW_VS_BACKGROUND) fast = fast && vec_ok && prev.qt > 0 && (prev.qt + 1) * kWTile <= wa.Q;
    if (EPI == SAF_QW_ROW_ARGMAX) fast = fast && prev.qt > 0 && prev.qt < n_qt - 1;  // first / last tile: reset / write-out
    if (EPI == SAF_QW_QUERY_MAX) fast = fast && (prev.qt + 1) * kWTile <= wa.Q;
#ifdef SAF_W2_NO_FAST
    fast = false;
#endif
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr bool kSwap = EPI == SAF_QW_QUERY_MAX;
    if (!fast) {
      if (step > 0) w2_epilogue<OT, EPI, NF>(wa, pc, prev, st, r, h, n_qt, vec_ok);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int f = 0; f < NF; ++f) c[f] = kSwap ? mfma16<FT>(a[f][s], t[s], s == 0 ? zero16 : c[f]) : mfma16<FT>(t[s], a[f][s], s == 0 ? zero16 : c[f]);  // C[query][feature row]; kSwap: C[feature row][query]
        if (s + AHEAD < KS) t[s + AHEAD] = *reinterpret_cast<const uint4*>(trow + SAF_W2_TOFF(s + AHEAD));
      }
      tail_ops = 0;
    } else {
      W2Fast<NF> fs;
      // ROW_ARGMAX: the tile's best per lane is a chain over the 16 accumulator registers of the previous tile -- compare
      // (v_cmp) behind one MFMA, the two selects behind the NEXT one: the MFMA and the LDS read between them are the wait
      // states a v_cndmask needs behind the v_cmp that wrote its mask (the compiler filled them with s_nop), and the winner's
      // query is selected from constants (its index inside the tile), the tile's base added once per tile.  3 instructions per
      // register instead of 5 in a kernel the power limit binds.
      float ra_v[NF], ra_x[NF];
      int ra_i[NF];
      bool ra_b[NF];
      int ra_pending = -1;
      auto ra_idc = [](int i) { return 8 * (i >> 2) + (i & 3); };
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int f = 0; f < NF; ++f) c[f] = kSwap ? mfma16<FT>(a[f][s], t[s], s == 0 ? zero16 : c[f]) : mfma16<FT>(t[s], a[f][s], s == 0 ? zero16 : c[f]);
        if (s + AHEAD < KS) t[s + AHEAD] = *reinterpret_cast<const uint4*>(trow + SAF_W2_TOFF(s + AHEAD));
        if (EPI == SAF_QW_ROW_ARGMAX || EPI == SAF_QW_QUERY_MAX) {
          if (ra_pending >= 0) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              ra_v[f] = ra_b[f] ? ra_x[f] : ra_v[f];
              ra_i[f] = ra_b[f] ? ra_idc(ra_pending) : ra_i[f];
            }
            ra_pending = -1;
          }
#pragma unroll
          for (int i = (s * 16) / KS; i < ((s + 1) * 16) / KS; ++i) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              if (EPI == SAF_QW_QUERY_MAX) {  // (swapped operands: the chain runs over the lane's ROWS, each with its own 1 / norm)
                if ((i & 3) == 0) {
                  const float4 v4 = *reinterpret_cast<const float4*>(st.inv_lds + 32 * f + 8 * (i >> 2) + 4 * h);
                  fs.iv[f][0] = v4.x; fs.iv[f][1] = v4.y; fs.iv[f][2] = v4.z; fs.iv[f][3] = v4.w;
                }
                ra_x[f] = pc[f][i] * fs.iv[f][i & 3];
              } else {
                ra_x[f] = pc[f][i];  // raw dot products: see w2_epilogue
              }
              if (i == 0) {
                ra_v[f] = ra_x[f];
                ra_i[f] = 0;
              } else {
                ra_b[f] = ra_x[f] > ra_v[f];  // queries ascend: the first maximum stays
              }
            }
            if (i > 0) ra_pending = i;
          }
        } else {
#pragma unroll
          for (int i = (s * 16) / KS; i < ((s + 1) * 16) / KS; ++i) w2_fast_piece<OT, EPI, NF>(i, wa, pc, prev, st, fs, r, h, n_qt);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the pieces where they are: between the MFMAs
      }
      if (EPI == SAF_QW_ROW_ARGMAX || EPI == SAF_QW_QUERY_MAX) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          if (ra_pending >= 0) {
            ra_v[f] = ra_b[f] ? ra_x[f] : ra_v[f];
            ra_i[f] = ra_b[f] ? ra_idc(ra_pending) : ra_i[f];
          }
          if (EPI == SAF_QW_ROW_ARGMAX) {
            const bool better = ra_v[f] > st.best_v[f];  // tiles ascend: the first maximum stays
            st.best_v[f] = better ? ra_v[f] : st.best_v[f];
            st.best_q[f] = better ? prev.qt * kWTile + 4 * h + ra_i[f] : st.best_q[f];
          } else if (f == 0) {
            fs.bv = ra_v[0]; fs.bm = ra_i[0];
          } else {
            const bool better = ra_v[f] > fs.bv;  // rows ascend with the fragment: the first maximum stays
            fs.bv = better ? ra_v[f] : fs.bv;
            fs.bm = better ? 32 * f + ra_i[f] : fs.bm;
          }
        }
      }
      // what this step issued behind its transfer, at the very least: the stores of registers 8-15 (one per fragment for
      // 16-bit scores, two for fp32) or the atomic below
      tail_ops = EPI == SAF_QW_ROW_ARGMAX ? 0 : EPI == SAF_QW_QUERY_MAX ? 1 : NF * (OT == SAF_F32 ? 2 : 1);
      if (EPI == SAF_QW_QUERY_MAX) {  // the two halves of a query's rows, then one atomic instruction: 32 queries
        const uint32_t row = (uint32_t)(prev.row[0] - r + wa.row_offset) + (uint32_t)fs.bm + 4u * (uint32_t)h;
        const uint32_t khi = ordered_bits(fs.bv), klo = ~row;
        const unsigned long long key = ((unsigned long long)khi << 32) | klo;
        const unsigned long long other = __shfl_xor(key, 32);
        const unsigned long long best = other > key ? other : key;
        if (h == 0) atomicMax(&wa.qkeys[prev.qt * kWTile + r], best);
      }
    }
    if (EPI == SAF_QW_QUERY_MAX && qt == 0) {
      // The new block's 1/norms, by row -- only now: this step's epilogue pieces were the LAST tile of the block before.
#pragma unroll
      for (int f = 0; f < NF; ++f)
        if (h == 0) s_inv[32 * f + r] = cur.inv[f];  // (read back by this wave only: the LDS keeps a wave's accesses in order)
    }
    if (qt == 0) W2_STAMP(4); else W2_STAMP(5);  // the tile: LDS reads, MFMAs, the previous tile's epilogue (4: first tile of a block, waits for its rows)
    if (more && !kDma) {
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int p = tid + k * kThreads;
        if (p < PIECES) {
          const int q = p / (D / 8), cc = p - q * (D / 8);
          *reinterpret_cast<w2_u4*>(nxt + q * ROWB + cc * 16) = stage[k];
        }
      }
    }
    prev = cur;
  };

  int64_t step = 0;
  for (; step + 1 < n_steps; step += 2) {
    step_body(step, acc[0], acc[1]);
    step_body(step + 1, acc[1], acc[0]);
  }
  if (step < n_steps) {
    step_body(step, acc[0], acc[1]);
    w2_epilogue<OT, EPI, NF>(wa, acc[0], prev, st, r, h, n_qt, vec_ok);
  } else {
    w2_epilogue<OT, EPI, NF>(wa, acc[1], prev, st, r, h, n_qt, vec_ok);
  }
#ifdef SAF_W2_STAMP
  if (blockIdx.x == 0 && lane == 0 && wave < 8) {
    seg[6] = (unsigned long long)n_steps;
    for (int k = 0; k < 8; ++k) g_w2_stamp[wave * 8 + k] = seg[k];
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------
// v3: the v2 scan on v_mfma_f32_16x16x32 instead of v_mfma_f32_32x32x16 (round 5).
//
// Both shapes cost the same cycles per flop, and v2 is bound by the clock the chip holds under the matrix load,
// not by idle cycles -- and that clock depends on the MFMA shape: MI355X_MICROARCH.md, "DVFS give-back" item 7, measures
// 1.12-1.14 x the FLOP/s for 16x16x32 loops whose operands are re-read from LDS (cdna_hip_programming.md rule 28: build both at
// the same output tile per wave, keep the faster by wall, on random data).  Same tile per wave as v2's default geometry (32 rows
// x 32 queries, 8 waves, two per SIMD), same LDS tiles, transfers, barrier and waits; what changes is the operand layout:
//   lane (c = lane & 15, g = lane >> 4) holds, of the wave's 32 rows, rows c and 16 + c (row blocks rb = 0, 1) -- 16 bytes of
//   every 64-byte k-step (k = 32 s + 8 g + j): a[rb][s], KS registers-of-four in all, as before;
//   a text fragment m = 2 s + qb is query block qb (16 queries) x k-step s: 1 KiB, read once from LDS, feeds the two row blocks'
//   MFMAs (2 x 16 cycles: the LDS bytes per matrix cycle of v2);
//   acc[rb][qb] (4 registers): query 16 qb + 4 g + i of the tile x the lane's row of block rb (QUERY_MAX, operands swapped:
//   row 16 rb + 4 g + i of the wave x query 16 qb + c).
// A row is shared by four lanes (v2: two): the reductions over a row's queries meet over xor 16 and xor 32; 16-bit scores leave as
// 16-byte stores after v_permlane16_swap (the odd 16-lane rows of one operand against the even rows of the other).
// ------------------------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#ifndef SAF_W3_PAIR
#define SAF_W3_PAIR 0  // 1: heat maps (VS_BACKGROUND, 16-bit out): a row's two 64-byte halves of one output line -- two consecutive tiles -- meet in the
                       // LDS and leave as ONE full-line store (0: a 64-byte store per row and tile: 46.5 GB written for 33.6 GB of scores).  Built and
                       // measured in round 6, results identical, SLOWER: 21.2 / 21.5 -> 22.7 / 22.95 ms on one box (profiles/r06/
                       // wide_scan_paired_stores_ab.txt) -- the epilogue sits at the issue port's limit: a ds_write, two ds_read
                       // and their wait per row block and pair of tiles cost more than the 13 GB of half-line write traffic they remove
                       // 2 (round 6, later): the low half held in four registers for one tile, the two halves stored back to back (heat maps and raw
                       // scores; no LDS, no extra instruction).  The counters say pairing removes the amplification (34.2 GB written); the clock
                       // says nobody was waiting for it: 21.0-21.2 -> 21.6 ms, raw scores 19.3 -> 19.7-19.9 (same file).  Not kept either.
#endif
#ifndef SAF_W3_PREFETCH
#define SAF_W3_PREFETCH 1  // the next row block's first pieces through the LDS (query_wide3_kernel, kPref); 0: every row from HBM at the block change
#endif

template <int FT>
__device__ __forceinline__ f32x4_t mfma32(const uint4& a, const uint4& b, const f32x4_t& c) {
  if (FT == SAF_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
}

struct W3Tile {
  float inv[2];     // scale / row norm of the lane's row of block rb
  int64_t row[2];   // its true index (may be >= n_rows on the last block)
  int qt;
};
struct W3State {
  float best_v[2];
  float lse0, lse1;  // VS_BACKGROUND: log-sum-exp of the backgrounds of the lane's rows (two scalars: as an array indexed by the row
                     // block the fp32-output instantiations kept the whole struct in scratch memory)
  int best_q[2];
  const float* inv_lds;  // QUERY_MAX: the wave's 32 (scale / norm) values in LDS, by row of the wave
  unsigned char* pair_lds;  // SAF_W3_PAIR=1: the wave's 4 KiB of output lines in LDS ([row block][row][128 bytes]), or nullptr
  int pair;                 // this step's epilogue tile: 0 = stores its half lines itself, 1 = low half, staged only, 2 = high half: stage, then full lines
  uint32_t hold[2][4];      // SAF_W3_PAIR=2: the low half's 16 bytes per row block, in registers until the high half is ready
};

template <int OT, int EPI>
__device__ __forceinline__ void w3_epilogue(const Wide2Args& wa, const f32x4_t (&acc)[2][2], const W3Tile& t, W3State& st, int c,
                                            int g, int n_qt, bool vec_ok) {
  const int qlane = t.qt * kWTile + 4 * g;  // the lane's first query of block qb = 0 (plain layout)
  if (EPI == SAF_QW_SCORES || EPI == SAF_QW_VS_BACKGROUND) {
    // (the two row blocks as two instances of one body with a constant index: left as a loop, the fp32-output instantiations kept
    //  it rolled -- `st` indexed at run time, i.e. in scratch memory)
    auto block = [&](auto rb_c) {
      constexpr int rb = decltype(rb_c)::value;
      float v[8];  // v[4 qb + i]: query 16 qb + 4 g + i of the tile
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = acc[rb][j >> 2][j & 3] * t.inv[rb];
      int colt = t.qt * kWTile;  // output column of the tile's query 0
      bool write = true;
      // (the new log-sum-exp leaves its branch as a VALUE and is stored unconditionally below: a float stored to `st` on one path
      //  and floats stored to the output on the other were merged into one store through a selected FLAT pointer -- `st` in
      //  scratch memory in every fp32-output instantiation)
      float lse_new = rb == 0 ? st.lse0 : st.lse1;
      if (EPI == SAF_QW_VS_BACKGROUND) {
        if (t.qt == 0) {  // tile 0 holds the backgrounds: per-row log-sum-exp of their scaled scores, no output
          float m = -INFINITY;
#pragma unroll
          for (int j = 0; j < 8; ++j) m = (16 * (j >> 2) + 4 * g + (j & 3) < wa.n_bg) ? fmaxf(m, v[j]) : m;
          m = fmaxf(m, __shfl_xor(m, 16));
          m = fmaxf(m, __shfl_xor(m, 32));
          float e = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) e += (16 * (j >> 2) + 4 * g + (j & 3) < wa.n_bg) ? __expf(v[j] - m) : 0.f;
          e += __shfl_xor(e, 16);
          e += __shfl_xor(e, 32);
          lse_new = m + __logf(e);
          write = false;
        } else {
          const bool rescale = (wa.flags & 1) != 0;  // query_mesh.py:39: ((r - 0.5) * 2).clamp(0, 1)
          const float ka = -t.inv[rb] * 1.4426950408889634f, kb = (rb == 0 ? st.lse0 : st.lse1) * 1.4426950408889634f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float p = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc[rb][j >> 2][j & 3], ka, kb)));
            v[j] = __builtin_amdgcn_fmed3f(__builtin_fmaf(p, rescale ? 2.0f : 1.0f, rescale ? -1.0f : 0.0f), 0.0f, 1.0f);
          }
          colt -= kWTile;
        }
      }
      if (rb == 0) st.lse0 = lse_new; else st.lse1 = lse_new;
      if (write) {
        const int ncols = EPI == SAF_QW_VS_BACKGROUND ? wa.Q - kWTile : wa.Q;
        const int64_t row = t.row[rb];
        if (OT == SAF_F32) {
          if (row < wa.n_rows) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
              const int q0 = colt + 16 * qb + 4 * g;
              float* o = static_cast<float*>(wa.out) + row * wa.ostride + q0;
              if (vec_ok && q0 + 3 < ncols) {
                *reinterpret_cast<float4*>(o) = make_float4(v[4 * qb], v[4 * qb + 1], v[4 * qb + 2], v[4 * qb + 3]);
              } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  if (q0 + i < ncols) o[i] = v[4 * qb + i];
              }
            }
          }
        } else {
          // the lane's two groups of four (8 bytes each) against the neighbouring 16-lane row's: even rows end with queries
          // 4 g .. 4 g + 7, odd rows with 16 + 4 (g - 1) .. + 7 of their feature row -- one 16-byte store each
          const uint2 lo = pack4_16<OT>(v[0], v[1], v[2], v[3]);  // block 0: queries 4 g ..
          const uint2 hi = pack4_16<OT>(v[4], v[5], v[6], v[7]);  // block 1: queries 16 + 4 g ..
          auto rx = __builtin_amdgcn_permlane16_swap(lo.x, hi.x, false, false);
          auto ry = __builtin_amdgcn_permlane16_swap(lo.y, hi.y, false, false);
          const uint32_t ww[4] = {rx[0], ry[0], rx[1], ry[1]};
          const int qv = colt + ((g & 1) ? 16 + 4 * (g - 1) : 4 * g);  // first of this lane's 8 columns
          if (row < wa.n_rows) {
            uint16_t* o = static_cast<uint16_t*>(wa.out) + row * wa.ostride + qv;
            if (vec_ok && qv + 7 < ncols) {
              typedef unsigned int w3_u4 __attribute__((ext_vector_type(4)));
              const w3_u4 w = {ww[0], ww[1], ww[2], ww[3]};
              *reinterpret_cast<w3_u4*>(o) = w;
            } else {
#pragma unroll
              for (int i = 0; i < 8; ++i)
                if (qv + i < ncols) o[i] = (uint16_t)(ww[i >> 1] >> (16 * (i & 1)));
            }
          }
        }
      }
    };
    block(std::integral_constant<int, 0>{});
    block(std::integral_constant<int, 1>{});
  } else if (EPI == SAF_QW_ROW_ARGMAX) {
    const bool last = t.qt == n_qt - 1;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float bv = t.qt == 0 ? -INFINITY : st.best_v[rb];
      int bq = t.qt == 0 ? 0 : st.best_q[rb];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = qlane + 16 * (j >> 2) + (j & 3);
        float x = acc[rb][j >> 2][j & 3];  // RAW dot products (see w2_epilogue)
        if (last && q >= wa.Q) x = -INFINITY;
        if (x > bv) { bv = x; bq = q; }  // the lane's queries ascend: the first maximum stays
      }
      if (last) {  // the four lanes of a row: the larger score, the smaller query on ties
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
          const float ov = __shfl_xor(bv, off);
          const int oq = __shfl_xor(bq, off);
          if (ov > bv || (ov == bv && oq < bq)) { bv = ov; bq = oq; }
        }
        if (g == 0 && t.row[rb] < wa.n_rows) { wa.out_index[t.row[rb]] = bq; wa.out_value[t.row[rb]] = bv * t.inv[rb]; }
      }
      st.best_v[rb] = bv;
      st.best_q[rb] = bq;
    }
  } else {  // SAF_QW_QUERY_MAX, operands swapped: acc[rb][qb][i] = row 16 rb + 4 g + i of the wave x query 16 qb + c
    const int64_t base = t.row[0] - c;  // the wave's first row
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int q = t.qt * kWTile + 16 * qb + c;
      float bv = -INFINITY;
      int bm = 0;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const float4 iv = *reinterpret_cast<const float4*>(st.inv_lds + 16 * rb + 4 * g);
        const float ivs[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = 16 * rb + 4 * g + i;
          const float x = base + m < wa.n_rows ? acc[rb][qb][i] * ivs[i] : -INFINITY;
          if (x > bv) { bv = x; bm = m; }  // the lane's rows ascend: the first maximum stays
        }
      }
      unsigned long long key = bv > -INFINITY ? ((unsigned long long)ordered_bits(bv) << 32) | (uint32_t) ~(uint32_t)(base + bm + wa.row_offset) : 0ull;
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const unsigned long long other = __shfl_xor(key, off);
        key = other > key ? other : key;
      }
      if (g == 0 && q < wa.Q && key) atomicMax(&wa.qkeys[q], key);
    }
  }
}

// The same epilogue in sixteen pieces that sit BETWEEN the MFMAs of the next tile (see w2_fast_piece: a wave issues in order, and
// a few vector instructions fit in the shadow of every MFMA), for interior tiles: every row of the wave valid, aligned output, all
// 32 columns real -- wave-uniform, branch-free.  Piece k finishes one accumulator register per row block (SCORES / VS_BACKGROUND /
// ROW_ARGMAX: k = 8 rb + j, j = 4 qb + i) or per query block (QUERY_MAX: k = 8 qb + 4 rb + i); what leaves the wave (a store per
// row block, an atomic per query block) follows pieces 7 and 15.
struct W3Fast {
  float v[2][8];     // SCORES / VS_BACKGROUND: the row block's finished values
  float ka[2], kb[2];
  float iv[4];       // QUERY_MAX: 1/norm of the four rows behind the current accumulator
  float cv; int ci;  // the running best of the current chain
  uint32_t k0hi, k0lo;  // QUERY_MAX: query block 0's key, until block 1's is complete
};
template <int OT, int EPI>
__device__ __forceinline__ void w3_piece(int k, const Wide2Args& wa, const f32x4_t (&pacc)[2][2], const W3Tile& t, W3State& st,
                                         W3Fast& fs, int c, int g) {
  if (EPI == SAF_QW_SCORES || EPI == SAF_QW_VS_BACKGROUND) {
    const int rb = k >> 3, j = k & 7;
    const float raw = pacc[rb][j >> 2][j & 3];
    float x;
    if (EPI == SAF_QW_VS_BACKGROUND) {
      const bool rescale = (wa.flags & 1) != 0;
      if (j == 0) {
        fs.ka[rb] = -t.inv[rb] * 1.4426950408889634f;
        fs.kb[rb] = (rb == 0 ? st.lse0 : st.lse1) * 1.4426950408889634f;
      }
      const float p = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(raw, fs.ka[rb], fs.kb[rb])));
      x = __builtin_amdgcn_fmed3f(__builtin_fmaf(p, rescale ? 2.0f : 1.0f, rescale ? -1.0f : 0.0f), 0.0f, 1.0f);
    } else {
      x = raw * t.inv[rb];
    }
    fs.v[rb][j] = x;
    if (j == 7) {
      const int colt = (EPI == SAF_QW_VS_BACKGROUND ? t.qt - 1 : t.qt) * kWTile;
      const float* v = fs.v[rb];
      if (OT == SAF_F32) {
        float* o = static_cast<float*>(wa.out) + t.row[rb] * wa.ostride + colt + 4 * g;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(o + 16) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        const uint2 lo = pack4_16<OT>(v[0], v[1], v[2], v[3]);
        const uint2 hi = pack4_16<OT>(v[4], v[5], v[6], v[7]);
        auto rx = __builtin_amdgcn_permlane16_swap(lo.x, hi.x, false, false);
        auto ry = __builtin_amdgcn_permlane16_swap(lo.y, hi.y, false, false);
        typedef unsigned int w3_u4 __attribute__((ext_vector_type(4)));
        const w3_u4 w = {rx[0], ry[0], rx[1], ry[1]};
        const int col0 = (g & 1) ? 16 + 4 * (g - 1) : 4 * g;  // the lane's 8 consecutive columns of the tile
        if (SAF_W3_PAIR == 2 && (EPI == SAF_QW_VS_BACKGROUND || EPI == SAF_QW_SCORES) && st.pair != 0) {
          // the line's two halves leave back to back: the low half waited one tile in four registers
          if (st.pair == 1) {
            st.hold[rb][0] = w[0]; st.hold[rb][1] = w[1]; st.hold[rb][2] = w[2]; st.hold[rb][3] = w[3];
          } else {
            uint16_t* o = static_cast<uint16_t*>(wa.out) + t.row[rb] * wa.ostride + colt + col0;
            const w3_u4 lo = {st.hold[rb][0], st.hold[rb][1], st.hold[rb][2], st.hold[rb][3]};
            *reinterpret_cast<w3_u4*>(o - kWTile) = lo;
            *reinterpret_cast<w3_u4*>(o) = w;
          }
        } else if (SAF_W3_PAIR == 1 && EPI == SAF_QW_VS_BACKGROUND && st.pair != 0) {
          // the row's line in LDS: this tile's 64 bytes in its half; the high half then reads whole lines back -- lane l takes
          // 16 bytes (l & 7) of row (l >> 3) + 8 h -- and a wave's store instruction writes eight full 128-byte lines
          unsigned char* line = st.pair_lds + (rb * 16 + c) * 128 + (st.pair == 2 ? 64 : 0) + col0 * 2;
          *reinterpret_cast<w3_u4*>(line) = w;
          if (st.pair == 2) {
            const int lane = 16 * g + c;
            const int64_t row0 = t.row[rb] - c;  // the row block's first row
            uint16_t* ob = static_cast<uint16_t*>(wa.out) + (colt - kWTile) + 8 * (lane & 7);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int r = (lane >> 3) + 8 * h;
              const w3_u4 full = *reinterpret_cast<const w3_u4*>(st.pair_lds + (rb * 16 + r) * 128 + (lane & 7) * 16);
              *reinterpret_cast<w3_u4*>(ob + (row0 + r) * wa.ostride) = full;
            }
          }
        } else {
          uint16_t* o = static_cast<uint16_t*>(wa.out) + t.row[rb] * wa.ostride + colt + col0;
          *reinterpret_cast<w3_u4*>(o) = w;
        }
      }
    }
  } else if (EPI == SAF_QW_ROW_ARGMAX) {
    const int rb = k >> 3, j = k & 7;
    const float x = pacc[rb][j >> 2][j & 3];  // raw dot products: see w2_epilogue
    if (j == 0) {
      fs.cv = x; fs.ci = 0;
    } else {
      const bool better = x > fs.cv;  // the lane's queries ascend: the first maximum stays
      fs.cv = better ? x : fs.cv;
      fs.ci = better ? 16 * (j >> 2) + (j & 3) : fs.ci;
    }
    if (j == 7) {
      const bool better = fs.cv > st.best_v[rb];  // tiles ascend: the first maximum stays
      st.best_v[rb] = better ? fs.cv : st.best_v[rb];
      st.best_q[rb] = better ? t.qt * kWTile + 4 * g + fs.ci : st.best_q[rb];
    }
  } else {  // SAF_QW_QUERY_MAX (operands swapped): the chain over the lane's eight rows of query 16 qb + c
    const int qb = k >> 3, rb = (k >> 2) & 1, i = k & 3;
    if (i == 0) {
      const float4 v4 = *reinterpret_cast<const float4*>(st.inv_lds + 16 * rb + 4 * g);
      fs.iv[0] = v4.x; fs.iv[1] = v4.y; fs.iv[2] = v4.z; fs.iv[3] = v4.w;
    }
    const float x = pacc[rb][qb][i] * fs.iv[i];
    const int m = 16 * rb + 4 * g + i;
    if ((k & 7) == 0) {
      fs.cv = x; fs.ci = m;
    } else {
      const bool better = x > fs.cv;  // the lane's rows ascend: the first maximum stays
      fs.cv = better ? x : fs.cv;
      fs.ci = better ? m : fs.ci;
    }
    if ((k & 7) == 7) {
      const uint32_t row = (uint32_t)(t.row[0] - c + wa.row_offset) + (uint32_t)fs.ci;
      const uint32_t khi = ordered_bits(fs.cv), klo = ~row;
      if (qb == 0) {
        fs.k0hi = khi; fs.k0lo = klo;
      } else {
        // The four lanes of a query meet without the LDS: v_permlane16_swap trades block 0's key of the odd 16-lane rows for
        // block 1's key of the even rows -- even rows then hold both halves of a pair of rows for block 0, odd rows for block 1 --,
        // v_permlane32_swap brings the other pair of rows; rows 0 and 1 issue ONE atomic instruction for the tile's 32 queries.
        auto sh = __builtin_amdgcn_permlane16_swap(fs.k0hi, khi, false, false);
        auto sl = __builtin_amdgcn_permlane16_swap(fs.k0lo, klo, false, false);
        const unsigned long long ka = ((unsigned long long)sh[0] << 32) | sl[0], kb = ((unsigned long long)sh[1] << 32) | sl[1];
        unsigned long long key = ka > kb ? ka : kb;
        auto th = __builtin_amdgcn_permlane32_swap((uint32_t)(key >> 32), (uint32_t)(key >> 32), false, false);
        auto tl = __builtin_amdgcn_permlane32_swap((uint32_t)key, (uint32_t)key, false, false);
        const unsigned long long other = g >= 2 ? ((unsigned long long)th[0] << 32) | tl[0] : ((unsigned long long)th[1] << 32) | tl[1];
        key = other > key ? other : key;
        if (g < 2) atomicMax(&wa.qkeys[t.qt * kWTile + 16 * g + c], key);
      }
    }
  }
}

template <int FT, int OT, int KS, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void query_wide3_kernel(Wide2Args wa) {
  constexpr int kThreads = 512, kWaves = 8, kRows = 32;
  constexpr int D = KS * 16, S = KS / 2;  // S k-steps of 32
  constexpr int ROWB = D * 2 + 16;        // padded LDS row: the 16 lanes of a fragment's row group hit distinct bank quads
  constexpr bool kDma = (D == 512);
  constexpr bool kSwap = EPI == SAF_QW_QUERY_MAX;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_tiles[];  // 2 x [32][ROWB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n_qt = wa.Qpad / kWTile;
  const int64_t rows_per_wg = (int64_t)kWaves * kRows;
  const int64_t n_blocks = (wa.n_rows + rows_per_wg - 1) / rows_per_wg;
  const int64_t my_blocks = blockIdx.x < n_blocks ? (n_blocks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t n_steps = my_blocks * n_qt;
  if (n_steps == 0) return;
  constexpr int PIECES = kWTile * (D / 8);
  constexpr int PPT = (PIECES + kThreads - 1) / kThreads;
  const bool vec_ok = (wa.ostride % 8 == 0) && (((uintptr_t)wa.out & 15) == 0);

  for (int p = tid; p < PIECES; p += kThreads) {  // tile 0 of the first block
    const int q = p / (D / 8), cc = p - q * (D / 8);
    *reinterpret_cast<uint4*>(s_tiles + q * ROWB + cc * 16) = *reinterpret_cast<const uint4*>(wa.text16 + (int64_t)q * D + cc * 8);
  }

  uint4 a[2][S];
  W3Tile cur, prev;
  W3State st;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    st.best_v[rb] = -INFINITY; st.best_q[rb] = 0;
    cur.inv[rb] = 0.f; cur.row[rb] = 0;
  }
  st.lse0 = st.lse1 = 0.f;
  float* s_inv = reinterpret_cast<float*>(s_tiles + 2 * kWTile * ROWB) + wave * kRows;  // QUERY_MAX only
  st.inv_lds = s_inv;
  // SAF_W3_PAIR: 4 KiB of output lines per wave behind the tiles (the heat maps use no row prefetch region); lines must be lines:
  // output rows 128-byte aligned
  constexpr bool kPair = ((SAF_W3_PAIR == 1 && EPI == SAF_QW_VS_BACKGROUND) ||
                          (SAF_W3_PAIR == 2 && (EPI == SAF_QW_VS_BACKGROUND || EPI == SAF_QW_SCORES))) && OT != SAF_F32;
  const bool pair_ok = kPair && vec_ok && (((uintptr_t)wa.out & 127) == 0) && ((wa.ostride * 2) % 128 == 0);
  st.pair_lds = kPair && SAF_W3_PAIR == 1 ? s_tiles + 2 * kWTile * ROWB + 1024 + wave * 4096 : nullptr;
  st.pair = 0;
  bool pair_staged = false;  // the previous tile's halves wait in the LDS (wave-uniform)
  cur.qt = 0;
  prev = cur;
  f32x4_t acc[2][2][2];  // [step parity][row block][query block]
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)s_tiles;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int tail_ops = 0;
  int qt_run = 0;
  int64_t blk_run = blockIdx.x;
  // The block change is the scan's largest idle stretch: every wave replaces its 32 KiB of rows at once, a CU pulls HBM at 20-25 GB/s
  // (profiles/r05/gather_probe_movers.log), nothing computes meanwhile -- 2.9 of 15.4 ms (SAF_W3_NO_RELOAD).  The registers have no
  // room for a second set, but the LDS has 93 KiB beside the two text tiles: kPref of a wave's 32 one-KiB pieces (piece p = row block
  // p / S, k-step p % S: exactly register a[p / S][p % S] of every lane, lane-linear) of the NEXT block travel there by LDS-DMA while
  // this block computes -- one piece per step, issued last in the step so that it is the step's youngest vector-memory operation and
  // the counted wait in front of the next barrier lets it fly -- and the block change reads them back with ds_read_b128.
  // Measured, one box, ms with / without (profiles/r05/wide_scan_prefetch_ab.txt): best voxel per query 16.9 / 17.45, raw scores
  // 20.05 / 20.65 -- a third of the pieces, a sixth of what SAF_W3_NO_RELOAD suggests the block change costs (that ablation also feeds
  // the matrix pipes constant operands, which raises the clock) --; row argmax 15.8 / 15.85 (its block change already overlaps the
  // last tile's write-out), heat maps 21.8 / 21.7: those two keep loading every row at the block change.
  constexpr int kPref = (kDma && SAF_W3_PREFETCH && (EPI == SAF_QW_SCORES || EPI == SAF_QW_QUERY_MAX)) ? 11 : 0;
  constexpr uint32_t kPrefOff = 2 * kWTile * ROWB + 1024;  // behind the tiles and QUERY_MAX's 1 KiB of 1/norms
  int pref_cnt = 0;       // pieces of the next block issued so far (wave-uniform)
  bool pref_ok = false;   // the next block exists and all 32 rows of this wave's part of it do
  const uint16_t* pref_base = wa.feats;
  const bool pref_stride_ok = wa.fstride < ((int64_t)1 << 24);  // a lane's byte offset from its wave's first row fits 32 bits
  const uint32_t pref_lds = lds_base + kPrefOff + (uint32_t)wave_u * (uint32_t)(kPref * 1024);

  auto step_body = [&](int64_t step, f32x4_t (&cacc)[2][2], const f32x4_t (&pacc)[2][2]) __attribute__((always_inline)) {
    const int qt = qt_run;
    qt_run = qt + 1 < n_qt ? qt + 1 : 0;
#ifdef SAF_W3_NO_RELOAD  // (development: only the first block's rows are loaded -- wrong results, the scan without its block-change bubble)
    if (qt == 0 && step > 0) {
      const int64_t blk = blk_run;
      blk_run += gridDim.x;
      const int64_t row0 = (blk * kWaves + wave) * kRows;
      cur.row[0] = row0 + c; cur.row[1] = row0 + 16 + c;
    }
    if (qt == 0 && step == 0) {
#else
    if (qt == 0) {  // a new row block: its rows per wave, register resident for every query tile
#endif
      const int64_t blk = blk_run;
      blk_run += gridDim.x;
      const int64_t row0 = (blk * kWaves + wave) * kRows;
      // pieces of this block that the block before sent to the LDS (none for the first block).  They are old -- the last one was
      // issued kPref steps into that block and every later step waited for operations younger than it -- unless the block is short
      const int ready = pref_ok ? pref_cnt : 0;
      if (kPref && ready > 0 && n_qt < kPref + 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        cur.row[rb] = row0 + 16 * rb + c;
        const int64_t rr = cur.row[rb] < wa.n_rows ? cur.row[rb] : wa.n_rows - 1;  // padded lanes recompute the last row
        const uint16_t* pa = wa.feats + rr * wa.fstride + 8 * g;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const int p = rb * S + s;
          if (p < kPref && p < ready)
            a[rb][s] = *reinterpret_cast<const uint4*>(s_tiles + kPrefOff + (size_t)(wave * kPref + p) * 1024 + lane * 16);
          else
            a[rb][s] = ld_stream_u4(pa + 32 * s);
        }
      }
      if (kPref) {  // the block after this one
        const int64_t row0n = (blk_run * kWaves + wave_u) * kRows;  // (wave-uniform by construction: the transfer's base lives in SGPRs)
        pref_ok = pref_stride_ok && blk_run < n_blocks && row0n + kRows <= wa.n_rows;
        pref_base = wa.feats + (pref_ok ? row0n : 0) * wa.fstride;
        pref_cnt = 0;
      }
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        cur.inv[rb] = wa.scale;
        if (wa.normalize) {
          float ss = 0.f, ss2 = 0.f;
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const uint32_t w[4] = {a[rb][s].x, a[rb][s].y, a[rb][s].z, a[rb][s].w};
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
              ss = dot2_self<FT>(w[j], ss);
              ss2 = dot2_self<FT>(w[j + 1], ss2);
            }
          }
          ss += ss2;
          ss += __shfl_xor(ss, 16);
          ss += __shfl_xor(ss, 32);
          cur.inv[rb] = wa.normalize == SAF_NORM_L2_CLAMP ? wa.scale / fmaxf(sqrtf(ss), 0.1f)
                                                          : (ss > 0.0f ? wa.scale / sqrtf(ss) : 0.0f);
        }
      }
    }
    cur.qt = qt;
    const unsigned char* curb = s_tiles + (size_t)(step & 1) * kWTile * ROWB;
    unsigned char* nxt = s_tiles + (size_t)((step + 1) & 1) * kWTile * ROWB;
#if SAF_W2_ASMDMA
    w2_wait_vm(kDma && !wa.safe_wait ? tail_ops : 0);
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();  // this tile's text is in LDS, every wave is done reading the other buffer
    const bool more = step + 1 < n_steps;
    const int qt_next = qt + 1 < n_qt ? qt + 1 : 0;
    typedef unsigned int w3s_u4 __attribute__((ext_vector_type(4)));
    w3s_u4 stage[kDma ? 1 : PPT];
    if (more) {
      if (kDma) {
        constexpr int K = kWTile / kWaves;
        w2_dma_rows<K, ROWB>(wa.text16 + (int64_t)(qt_next * kWTile + wave_u * K) * D, (uint32_t)lane * 16u,
                             lds_base + (uint32_t)((step + 1) & 1) * kWTile * ROWB + wave_u * K * ROWB);
      } else {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
          const int p = tid + k * kThreads;
          if (p < PIECES) {
            const int q = p / (D / 8), cc = p - q * (D / 8);
            stage[k] = *reinterpret_cast<const w3s_u4*>(wa.text16 + (int64_t)(qt_next * kWTile + q) * D + cc * 8);
          }
        }
      }
    }
#ifndef SAF_W3_AHEAD
#define SAF_W3_AHEAD 4  // text fragments requested ahead of their MFMAs.  One box, ms (heat maps / row argmax / query max / scores): 2: 21.4 / 15.9 / 17.6 / 20.3,
#endif                  // 3: 21.5 / 15.85 / 17.45 / 20.1, 4: 21.5 / 15.85 / 17.35 / 20.2, 6 (v2's): 21.6 / 16.3 / 17.8 / 20.3; 8 spills (25.3 / 15.7 / 18.8 / 20.5)
    constexpr int kAhead = EPI == SAF_QW_QUERY_MAX ? SAF_W3_AHEAD - 1 : SAF_W3_AHEAD;  // (the per-query maximum's chain state)
    constexpr int AHEAD = KS < kAhead ? KS : kAhead;
    uint4 t[KS];  // fragment m = 2 s + qb
    const unsigned char* trow = curb + c * ROWB + 16 * g;
#define SAF_W3_TOFF(m) ((((m) & 1) * 16) * ROWB + 64 * ((m) >> 1))
#pragma unroll
    for (int m = 0; m < AHEAD; ++m) t[m] = *reinterpret_cast<const uint4*>(trow + SAF_W3_TOFF(m));
    // Is the previous tile an interior one (wave-uniform)?  Then its epilogue rides between this tile's MFMAs.
    bool fast = step > 0 && __all(prev.row[0] < wa.n_rows && prev.row[1] < wa.n_rows);
    if (EPI == SAF_QW_SCORES) fast = fast && vec_ok && (prev.qt + 1) * kWTile <= wa.Q;
    if (EPI == SAF_QW_VS_BACKGROUND) fast = fast && vec_ok && prev.qt > 0 && (prev.qt + 1) * kWTile <= wa.Q;
    if (EPI == SAF_QW_ROW_ARGMAX) fast = fast && prev.qt > 0 && prev.qt < n_qt - 1;  // first / last tile: reset / write-out
    if (EPI == SAF_QW_QUERY_MAX) fast = fast && (prev.qt + 1) * kWTile <= wa.Q;
#ifdef SAF_W3_NO_FAST
    fast = false;
#endif
    if (kPair) {
      // low half (an even output tile): staged only -- if the NEXT tile's epilogue takes this path too (same block, a whole tile,
      // and not the workgroup's very last one, whose epilogue runs behind the loop); high half: completes what was staged
      st.pair = 0;
      if (fast && pair_ok) {
        const int ot = EPI == SAF_QW_VS_BACKGROUND ? prev.qt - 1 : prev.qt;  // the output tile: two of them make a 128-byte line
        if ((ot & 1) == 0) st.pair = (prev.qt + 1 < n_qt && (prev.qt + 2) * kWTile <= wa.Q && step + 1 < n_steps) ? 1 : 0;
        else st.pair = pair_staged ? 2 : 0;
      }
      pair_staged = st.pair == 1;
    }
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    if (!fast) {
#ifndef SAF_W3_NO_EPI  // (development: the scan without its epilogues -- wrong results, the matrix core's time)
      if (step > 0) w3_epilogue<OT, EPI>(wa, pacc, prev, st, c, g, n_qt, vec_ok);
#endif
#pragma unroll
      for (int m = 0; m < KS; ++m) {
        const int s = m >> 1, qb = m & 1;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          cacc[rb][qb] = kSwap ? mfma32<FT>(a[rb][s], t[m], s == 0 ? zero4 : cacc[rb][qb]) : mfma32<FT>(t[m], a[rb][s], s == 0 ? zero4 : cacc[rb][qb]);
        if (m + AHEAD < KS) t[m + AHEAD] = *reinterpret_cast<const uint4*>(trow + SAF_W3_TOFF(m + AHEAD));
      }
      tail_ops = 0;
    } else {
      W3Fast fs;
#pragma unroll
      for (int m = 0; m < KS; ++m) {
        const int s = m >> 1, qb = m & 1;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          cacc[rb][qb] = kSwap ? mfma32<FT>(a[rb][s], t[m], s == 0 ? zero4 : cacc[rb][qb]) : mfma32<FT>(t[m], a[rb][s], s == 0 ? zero4 : cacc[rb][qb]);
        if (m + AHEAD < KS) t[m + AHEAD] = *reinterpret_cast<const uint4*>(trow + SAF_W3_TOFF(m + AHEAD));
#ifndef SAF_W3_NO_EPI
#pragma unroll
        for (int k = (m * 16) / KS; k < ((m + 1) * 16) / KS; ++k) w3_piece<OT, EPI>(k, wa, pacc, prev, st, fs, c, g);
#endif
        __builtin_amdgcn_sched_barrier(0);  // keep the pieces where they are: between the MFMAs
      }
      // what this step issued behind its transfer, at the very least: a store per row block (two for fp32) or the tile's atomic
      // -- every one of them by all lanes of a group that always exists
#ifndef SAF_W3_NO_EPI
      tail_ops = EPI == SAF_QW_ROW_ARGMAX ? 0 : EPI == SAF_QW_QUERY_MAX ? 1 : (OT == SAF_F32 ? 4 : 2);
      if (kPair && st.pair == 1) tail_ops = 0;  // (staged: no store this step; a completed pair: four)
      if (kPair && st.pair == 2) tail_ops = 4;
#else
      tail_ops = 0;
#endif
    }
    if (kPref && pref_ok && qt >= 1 && pref_cnt < kPref) {
      // one piece of the next block's rows, the step's youngest operation (see kPref)
      const int rbp = pref_cnt >= S ? 1 : 0, sp = pref_cnt - rbp * S;
      const uint32_t lane_off = (uint32_t)((16 * rbp + c) * (int)wa.fstride + 32 * sp + 8 * g) * 2u;
      w3_dma_piece(pref_base, lane_off, pref_lds + (uint32_t)pref_cnt * 1024u);
      ++pref_cnt;
      tail_ops += 1;
    }
    if (EPI == SAF_QW_QUERY_MAX && qt == 0) {
      // the new block's 1/norms, by row of the wave -- only now: this step's epilogue was the LAST tile of the block before
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        if (g == 0) s_inv[16 * rb + c] = cur.inv[rb];
    }
    if (more && !kDma) {
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const int p = tid + k * kThreads;
        if (p < PIECES) {
          const int q = p / (D / 8), cc = p - q * (D / 8);
          *reinterpret_cast<w3s_u4*>(nxt + q * ROWB + cc * 16) = stage[k];
        }
      }
    }
    prev = cur;
  };

  int64_t step = 0;
  for (; step + 1 < n_steps; step += 2) {
    step_body(step, acc[0], acc[1]);
    step_body(step + 1, acc[1], acc[0]);
  }
  if (step < n_steps) {
    step_body(step, acc[0], acc[1]);
    w3_epilogue<OT, EPI>(wa, acc[0], prev, st, c, g, n_qt, vec_ok);
  } else {
    w3_epilogue<OT, EPI>(wa, acc[1], prev, st, c, g, n_qt, vec_ok);
  }
}

__global__ void qkeys_decode_kernel(const unsigned long long* __restrict__ keys, int Q, float* __restrict__ out_value,
                                    int64_t* __restrict__ out_row, int64_t row_hi) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q) return;
  const unsigned long long k = keys[q];
  out_value[q] = k ? from_ordered_bits((uint32_t)(k >> 32)) : -INFINITY;
  // the low word is ~(row + offset) mod 2^32; row_hi restores the bits above 32 (volumes here have < 2^32 rows)
  out_row[q] = k ? (int64_t)(uint32_t)~(uint32_t)k + row_hi : -1;
}

template <int FT, int OT, int KS>
int launch_wide(const uint16_t* feats, int64_t n_rows, int64_t fstride, const uint16_t* text16, int Q, int Qpad,
                float scale, int normalize, void* out, int64_t ostride, hipStream_t s) {
  constexpr size_t shmem = 2 * (size_t)kWTile * (KS * 32 + 16);
  auto fn = query_wide_kernel<FT, OT, KS>;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)shmem);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  int64_t blocks = (n_rows + 32 * kWWaves - 1) / (32 * kWWaves);
  const int64_t cap = device_cus();
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kWThreads), shmem, s, feats, n_rows, fstride, text16, Q, Qpad,
                     scale, normalize, out, ostride);
  return check_launch("query_wide_kernel");
}

template <int FT, int OT>
int launch_wide_ks(int D, const uint16_t* feats, int64_t n_rows, int64_t fstride, const uint16_t* text16, int Q,
                   int Qpad, float scale, int normalize, void* out, int64_t ostride, hipStream_t s) {
  switch (D) {
    case 128: return launch_wide<FT, OT, 8>(feats, n_rows, fstride, text16, Q, Qpad, scale, normalize, out, ostride, s);
    case 256: return launch_wide<FT, OT, 16>(feats, n_rows, fstride, text16, Q, Qpad, scale, normalize, out, ostride, s);
    case 512: return launch_wide<FT, OT, 32>(feats, n_rows, fstride, text16, Q, Qpad, scale, normalize, out, ostride, s);
    default: return fail(SAF_E_UNSUPPORTED, "wide scan: feat_dim must be 128, 256 or 512 (got %d)", D);
  }
}

template <int FT>
int launch_wide_ot(int ot, int D, const uint16_t* feats, int64_t n_rows, int64_t fstride, const uint16_t* text16, int Q,
                   int Qpad, float scale, int normalize, void* out, int64_t ostride, hipStream_t s) {
  switch (ot) {
    case SAF_F32: return launch_wide_ks<FT, SAF_F32>(D, feats, n_rows, fstride, text16, Q, Qpad, scale, normalize, out, ostride, s);
    case SAF_F16: return launch_wide_ks<FT, SAF_F16>(D, feats, n_rows, fstride, text16, Q, Qpad, scale, normalize, out, ostride, s);
    case SAF_BF16: return launch_wide_ks<FT, SAF_BF16>(D, feats, n_rows, fstride, text16, Q, Qpad, scale, normalize, out, ostride, s);
    default: return fail(SAF_E_INVALID, "wide scan: bad out_dtype %d", ot);
  }
}

template <int FT, int OT, int KS, int EPI, int NF, int TH>
int launch_wide2_nf(const Wide2Args& wa, hipStream_t s) {
  constexpr size_t shmem = 2 * (size_t)kWTile * (KS * 32 + 16) + (EPI == SAF_QW_QUERY_MAX ? (TH / 64) * 32 * NF * sizeof(float) : 0);
  auto fn = query_wide2_kernel<FT, OT, KS, EPI, NF, TH>;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)shmem);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int64_t rows_per_wg = (TH / 64) * 32 * NF;
  int64_t blocks = (wa.n_rows + rows_per_wg - 1) / rows_per_wg;
  const int64_t cap = (int64_t)device_cus() * (NF == 1 && TH == 256 ? 2 : 1);  // persistent: one or two workgroups per CU
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(TH), shmem, s, wa);
  return check_launch("query_wide2_kernel");
}

template <int FT, int OT, int KS, int EPI>
int launch_wide3(const Wide2Args& wa, hipStream_t s) {
  // two text tiles, 1 KiB of 1/norms (QUERY_MAX), and at D = 512 eleven 1 KiB pieces per wave of the next block's rows (kPref)
  constexpr size_t shmem = 2 * (size_t)kWTile * (KS * 32 + 16) + 1024 +
                           ((KS == 32 && SAF_W3_PREFETCH && (EPI == SAF_QW_SCORES || EPI == SAF_QW_QUERY_MAX)) ? 8 * 11 * 1024 : 0) +
                           ((SAF_W3_PAIR == 1 && EPI == SAF_QW_VS_BACKGROUND && OT != SAF_F32) ? 8 * 4096 : 0);
  auto fn = query_wide3_kernel<FT, OT, KS, EPI>;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int64_t rows_per_wg = 8 * 32;
  int64_t blocks = (wa.n_rows + rows_per_wg - 1) / rows_per_wg;
  const int64_t cap = (int64_t)device_cus();  // persistent: one workgroup per CU
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(512), shmem, s, wa);
  return check_launch("query_wide3_kernel");
}

// The shipped geometry is 32 rows per wave, 8 waves per workgroup.  Development builds: -DSAF_W2_NF2 adds the 64-row instantiations
// (SAF_WIDE_ROWS=64 in the environment picks them; the fp32-out heat maps among them spill ten registers), -DSAF_W2_TWO_WGS two
// workgroups of four waves per CU (SAF_WIDE_ROWS=33).
template <int FT, int OT, int KS, int EPI>
int launch_wide2(const Wide2Args& wa, hipStream_t s) {
  const int rows_env = getenv("SAF_WIDE_ROWS") ? atoi(getenv("SAF_WIDE_ROWS")) : 0;
  (void)rows_env;
  // The scan runs on v_mfma_f32_16x16x32 (query_wide3_kernel: 4-9 % faster than the 32x32x16 form at the same tile per wave, the
  // chip holds a higher clock on it); SAF_WIDE_MFMA=32 (read per call) selects query_wide2_kernel.
  const char* shape = getenv("SAF_WIDE_MFMA");
  if (!(shape && atoi(shape) == 32)) return launch_wide3<FT, OT, KS, EPI>(wa, s);
#ifdef SAF_W2_NF2
  if (rows_env == 64) return launch_wide2_nf<FT, OT, KS, EPI, 2, 256>(wa, s);
#endif
#ifdef SAF_W2_TWO_WGS
  if (rows_env == 33) return launch_wide2_nf<FT, OT, KS, EPI, 1, 256>(wa, s);  // two workgroups of 4 waves per CU
#endif
  // measured at config 5 (ms, 32 vs 64 rows per wave): heat maps 27.5 / 31.4, best query per voxel 19.3 / 20.4, raw scores
  // 25.7 / 27.8 (rounds 2-3: the best voxel per query ran at 64 -- its reduction over the lanes of a wave was per tile; with the
  // operands swapped there is none)
  return launch_wide2_nf<FT, OT, KS, EPI, 1, 512>(wa, s);
}

template <int FT, int KS>
int launch_wide2_epi(int epi, int ot, const Wide2Args& wa, hipStream_t s) {
  switch (epi) {
    case SAF_QW_SCORES:
      switch (ot) {
        case SAF_F32: return launch_wide2<FT, SAF_F32, KS, SAF_QW_SCORES>(wa, s);
        case SAF_F16: return launch_wide2<FT, SAF_F16, KS, SAF_QW_SCORES>(wa, s);
        case SAF_BF16: return launch_wide2<FT, SAF_BF16, KS, SAF_QW_SCORES>(wa, s);
      }
      break;
    case SAF_QW_VS_BACKGROUND:
      switch (ot) {
        case SAF_F32: return launch_wide2<FT, SAF_F32, KS, SAF_QW_VS_BACKGROUND>(wa, s);
        case SAF_F16: return launch_wide2<FT, SAF_F16, KS, SAF_QW_VS_BACKGROUND>(wa, s);
        case SAF_BF16: return launch_wide2<FT, SAF_BF16, KS, SAF_QW_VS_BACKGROUND>(wa, s);
      }
      break;
    case SAF_QW_ROW_ARGMAX: return launch_wide2<FT, SAF_F32, KS, SAF_QW_ROW_ARGMAX>(wa, s);
    case SAF_QW_QUERY_MAX: return launch_wide2<FT, SAF_F32, KS, SAF_QW_QUERY_MAX>(wa, s);
  }
  return fail(SAF_E_INVALID, "wide scan: bad epilogue %d / out_dtype %d", epi, ot);
}

// fp32 text -> 16-bit tiles.  SCORES / ROW_ARGMAX / QUERY_MAX: row q of the text is row q.  VS_BACKGROUND: the
// first n_bg rows (the shared background prompts) fill tile 0, zero padded to 32 rows; target t is row 32 + t.
template <int FT>
__global__ void text_tiles_kernel(const float* __restrict__ text, int Q, int64_t tstride, int D, int Qpad, int n_bg,
                                  uint16_t* __restrict__ out, unsigned long long* __restrict__ qkeys, int negate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (qkeys && i < Qpad) qkeys[i] = 0ull;
  if (i >= Qpad * D) return;
  const int q = i / D, k = i - q * D;
  int src = q;
  if (n_bg > 0) src = q < kWTile ? (q < n_bg ? q : -1) : q - kWTile + n_bg;
  float v = (src >= 0 && src < Q) ? text[(int64_t)src * tstride + k] : 0.0f;
  if (negate) v = -v;  // ROW_ARGMAX with a negative scale: -text and |scale| give the same scores, and the rows' factor stays >= 0
  const int o = i;
  if (FT == SAF_BF16) {
    out[o] = (uint16_t)f32_to_bf16_bits(v);
  } else {
    const _Float16 hv = (_Float16)v;
    out[o] = __builtin_bit_cast(uint16_t, hv);
  }
}

// columns the kernel sees (Q of Wide2Args) and their padding to whole tiles
inline int wide2_cols(int n_text, int epi, int n_bg) { return epi == SAF_QW_VS_BACKGROUND ? kWTile + (n_text - n_bg) : n_text; }
inline size_t wide2_ws_bytes(int n_text, int D, int epi, int n_bg) {
  const size_t qpad = ((size_t)wide2_cols(n_text, epi, n_bg) + kWTile - 1) / kWTile * kWTile;
  return ((qpad * D * 2 + 255) & ~(size_t)255) + qpad * sizeof(unsigned long long);
}

}  // namespace
}  // namespace saf

using namespace saf;

extern "C" {

size_t saf_query_wide_workspace_bytes(int32_t n_text, int32_t feat_dim) {
  if (n_text <= 0 || feat_dim <= 0) return 0;
  const size_t qpad = ((size_t)n_text + kWTile - 1) / kWTile * kWTile;
  return (qpad * feat_dim * 2 + 255) & ~(size_t)255;
}

int saf_query_scan_wide(const void* feats, int32_t feat_dtype, int64_t n_rows, int64_t feat_stride, int32_t feat_dim,
                        const float* text, int32_t n_text, int64_t text_stride, float scale, int32_t normalize,
                        void* out, int32_t out_dtype, int64_t out_stride, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (feat_dtype != SAF_F16 && feat_dtype != SAF_BF16)
    return fail(SAF_E_UNSUPPORTED, "wide scan: features must be SAF_F16 or SAF_BF16");
  if (!feats || !text || !out || n_rows < 0 || n_text <= 0 || feat_stride < feat_dim || text_stride < feat_dim ||
      out_stride < n_text)
    return fail(SAF_E_INVALID, "wide scan: bad arguments");
  if (((uintptr_t)feats & 15) || (feat_stride % 8) != 0) return fail(SAF_E_INVALID, "wide scan: feature rows must be 16-byte aligned");
  if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < saf_query_wide_workspace_bytes(n_text, feat_dim))
    return fail(SAF_E_WORKSPACE, "wide scan: workspace needs %zu bytes", saf_query_wide_workspace_bytes(n_text, feat_dim));
  if (n_rows == 0) return SAF_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int Qpad = (n_text + kWTile - 1) / kWTile * kWTile;
  uint16_t* text16 = static_cast<uint16_t*>(workspace);
  const int items = Qpad * feat_dim;
  if (feat_dtype == SAF_BF16)
    hipLaunchKernelGGL(text_to_16_kernel<SAF_BF16>, dim3((items + 255) / 256), dim3(256), 0, s, text, n_text, text_stride,
                       feat_dim, Qpad, text16);
  else
    hipLaunchKernelGGL(text_to_16_kernel<SAF_F16>, dim3((items + 255) / 256), dim3(256), 0, s, text, n_text, text_stride,
                       feat_dim, Qpad, text16);
  int rc = check_launch("text_to_16_kernel");
  if (rc) return rc;
  const uint16_t* f = static_cast<const uint16_t*>(feats);
  return feat_dtype == SAF_BF16
             ? launch_wide_ot<SAF_BF16>(out_dtype, feat_dim, f, n_rows, feat_stride, text16, n_text, Qpad, scale, normalize,
                                        out, out_stride, s)
             : launch_wide_ot<SAF_F16>(out_dtype, feat_dim, f, n_rows, feat_stride, text16, n_text, Qpad, scale, normalize,
                                       out, out_stride, s);
}

size_t saf_query_wide_ex_workspace_bytes(int32_t n_text, int32_t feat_dim, int32_t epilogue, int32_t n_background) {
  if (n_text <= 0 || feat_dim <= 0 || n_background < 0 || n_background > n_text) return 0;
  return wide2_ws_bytes(n_text, feat_dim, epilogue, n_background);
}

int saf_query_scan_wide_ex(const void* feats, int32_t feat_dtype, int64_t n_rows, int64_t feat_stride, int32_t feat_dim,
                           const float* text, int32_t n_text, int64_t text_stride, float scale, int32_t normalize,
                           int32_t epilogue, int32_t n_background, int32_t flags, void* out, int32_t out_dtype,
                           int64_t out_stride, int32_t* out_index, float* out_value, int64_t* out_row, int64_t row_offset,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (feat_dtype != SAF_F16 && feat_dtype != SAF_BF16)
    return fail(SAF_E_UNSUPPORTED, "wide scan: features must be SAF_F16 or SAF_BF16");
  if (feat_dim != 256 && feat_dim != 512)
    return fail(SAF_E_UNSUPPORTED, "wide scan (fused epilogues): feat_dim must be 256 or 512 (got %d)", feat_dim);
  if ((!feats && n_rows > 0) || !text || n_rows < 0 || n_text <= 0 || feat_stride < feat_dim || text_stride < feat_dim)
    return fail(SAF_E_INVALID, "wide scan: bad arguments");
  if (((uintptr_t)feats & 15) || (feat_stride % 8) != 0) return fail(SAF_E_INVALID, "wide scan: feature rows must be 16-byte aligned");
  int n_bg = 0, n_out_cols = n_text;
  switch (epilogue) {
    case SAF_QW_SCORES:
      if ((!out && n_rows > 0) || out_stride < n_text) return fail(SAF_E_INVALID, "wide scan: SCORES needs out [n_rows, >= n_text]");
      break;
    case SAF_QW_VS_BACKGROUND:
      n_bg = n_background;
      n_out_cols = n_text - n_bg;
      if (n_bg < 1 || n_bg > kWTile || n_out_cols < 1)
        return fail(SAF_E_INVALID, "wide scan: VS_BACKGROUND needs 1..32 background rows followed by at least one target");
      if ((!out && n_rows > 0) || out_stride < n_out_cols) return fail(SAF_E_INVALID, "wide scan: VS_BACKGROUND needs out [n_rows, >= n_text - n_background]");
      if (!(scale > 0.0f)) return fail(SAF_E_INVALID, "wide scan: VS_BACKGROUND needs a positive scale");
      break;
    case SAF_QW_ROW_ARGMAX:
      if ((!out_index || !out_value) && n_rows > 0) return fail(SAF_E_INVALID, "wide scan: ROW_ARGMAX needs out_index and out_value [n_rows]");
      break;
    case SAF_QW_QUERY_MAX:
      if (!out_value || !out_row) return fail(SAF_E_INVALID, "wide scan: QUERY_MAX needs out_value and out_row [n_text]");
      if (row_offset < 0 || row_offset + n_rows > (int64_t)0xffffffffll) return fail(SAF_E_UNSUPPORTED, "wide scan: QUERY_MAX rows must index below 2^32");
      break;
    default: return fail(SAF_E_INVALID, "wide scan: bad epilogue %d", epilogue);
  }
  const size_t need = wide2_ws_bytes(n_text, feat_dim, epilogue, n_bg);
  if (!workspace || ((uintptr_t)workspace & 255) || workspace_bytes < need)
    return fail(SAF_E_WORKSPACE, "wide scan: workspace needs %zu bytes, 256-byte aligned", need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int cols = wide2_cols(n_text, epilogue, n_bg);
  const int Qpad = (cols + kWTile - 1) / kWTile * kWTile;
  uint16_t* text16 = static_cast<uint16_t*>(workspace);
  unsigned long long* qkeys = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(workspace) +
                                                                   (((size_t)Qpad * feat_dim * 2 + 255) & ~(size_t)255));
  const int items = Qpad * feat_dim;
  // ROW_ARGMAX compares raw dot products (a row's scale / norm must not flip their order): a negative scale goes into the text
  const int negate = epilogue == SAF_QW_ROW_ARGMAX && scale < 0.0f ? 1 : 0;
  if (negate) scale = -scale;
  if (feat_dtype == SAF_BF16)
    hipLaunchKernelGGL(text_tiles_kernel<SAF_BF16>, dim3((items + 255) / 256), dim3(256), 0, s, text, n_text, text_stride,
                       feat_dim, Qpad, n_bg, text16, epilogue == SAF_QW_QUERY_MAX ? qkeys : nullptr, negate);
  else
    hipLaunchKernelGGL(text_tiles_kernel<SAF_F16>, dim3((items + 255) / 256), dim3(256), 0, s, text, n_text, text_stride,
                       feat_dim, Qpad, n_bg, text16, epilogue == SAF_QW_QUERY_MAX ? qkeys : nullptr, negate);
  int rc = check_launch("text_tiles_kernel");
  if (rc) return rc;
  if (n_rows > 0) {
    Wide2Args wa;
    wa.feats = static_cast<const uint16_t*>(feats);
    wa.n_rows = n_rows; wa.fstride = feat_stride; wa.text16 = text16; wa.Q = cols; wa.Qpad = Qpad; wa.scale = scale;
    wa.normalize = normalize; wa.n_bg = n_bg; wa.flags = flags; wa.out = out; wa.ostride = out_stride;
    wa.out_index = out_index; wa.out_value = out_value; wa.qkeys = qkeys; wa.row_offset = row_offset;
    wa.safe_wait = getenv("SAF_W2_SAFE_WAIT") && getenv("SAF_W2_SAFE_WAIT")[0] == '1' ? 1 : 0;
    if (feat_dtype == SAF_BF16)
      rc = feat_dim == 512 ? launch_wide2_epi<SAF_BF16, 32>(epilogue, out_dtype, wa, s) : launch_wide2_epi<SAF_BF16, 16>(epilogue, out_dtype, wa, s);
    else
      rc = feat_dim == 512 ? launch_wide2_epi<SAF_F16, 32>(epilogue, out_dtype, wa, s) : launch_wide2_epi<SAF_F16, 16>(epilogue, out_dtype, wa, s);
    if (rc) return rc;
  }
  if (epilogue == SAF_QW_QUERY_MAX) {
    hipLaunchKernelGGL(qkeys_decode_kernel, dim3((n_text + 255) / 256), dim3(256), 0, s, qkeys, n_text, out_value, out_row,
                       (int64_t)0);
    return check_launch("qkeys_decode_kernel");
  }
  return SAF_OK;
}

#ifdef SAF_W2_STAMP
int saf_debug_w2_stamps(unsigned long long* host_out) {  // [8 waves][8]: cycles per segment, [6] = steps
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(saf::g_w2_stamp), sizeof(unsigned long long) * 64) == hipSuccess ? SAF_OK : SAF_E_HIP;
}
#endif

}  // extern "C"
