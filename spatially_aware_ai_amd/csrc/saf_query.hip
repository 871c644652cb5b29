// saf_query.hip -- CLIP-text query scan over fused feature rows on gfx950.
//
// Replaces Clip.run_query (reference clipfusion.py:899-904), Clip.clip_feature_surgery (:906-934,
// redundant_feats=None branch; the other branch is SAF_Q_SCORES against T - r) and the in-place
// row normalisation + nan_to_num of InSituManager.clip_text_query (clip_seem_fusion.py:507-511).
//
// v1 kernel: one wave per feature row.  The row is read once from HBM (the only HBM traffic:
// N*D*4 bytes in, N*L*4 out), normalised in registers/LDS, dotted against every text embedding
// (text matrix served from L1/L2), and the epilogue (softmax / surgery / last column) is fused so
// that the [N,L,D] blow-up of the reference (clipfusion.py:924-929) never exists.  HBM-bound for
// the reference's L <= ~64; the Q=1000 MFMA formulation is listed as next work in DESIGN.md.
#include <math.h>

#include "saf_common.h"
#include "saf_host.h"

namespace saf {
namespace {

constexpr int kQThreads = 256;
constexpr int kQWaves = kQThreads / 64;
enum { EPI_WEIGHTS = 3 };  // internal: surgery weights from row 0

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
  return x;
}

// feature element -> fp32 (FT: saf_dtype).  bf16 / fp16 volumes halve the scan's HBM bytes.
template <int FT>
__device__ __forceinline__ float load_feat(const void* __restrict__ base, int64_t i) {
  if (FT == SAF_BF16) return __builtin_bit_cast(float, (uint32_t) static_cast<const uint16_t*>(base)[i] << 16);
  if (FT == SAF_F16) return (float)static_cast<const _Float16*>(base)[i];
  return static_cast<const float*>(base)[i];
}

template <int EPI, int FT>
__global__ __launch_bounds__(kQThreads) void query_kernel(const void* __restrict__ feats, int64_t n_rows,
                                                           int64_t fstride, int D, const float* __restrict__ text,
                                                           int L, int64_t tstride, float scale, int normalize,
                                                           float* __restrict__ wts, float* __restrict__ out,
                                                           float* __restrict__ out_last) {
  extern __shared__ float s_mem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Dp = (D + 3) & ~3, Lp = (L + 3) & ~3;
  float* row = s_mem + (size_t)wave * (Dp + Lp);
  float* sc = row + Dp;
  // every wave of the block runs the same number of iterations so the barriers are uniform
  for (int64_t r0 = (int64_t)blockIdx.x * kQWaves; r0 < n_rows; r0 += (int64_t)gridDim.x * kQWaves) {
    const int64_t r = r0 + wave;
    const bool active = r < n_rows;
    if (active) {
      float ss = 0.f;
      for (int c = lane; c < D; c += 64) {
        const float x = load_feat<FT>(feats, r * fstride + c);
        row[c] = x;
        ss += x * x;
      }
      if (normalize) {
        // clip_feat /= clip_feat.norm(dim=-1, keepdim=True); nan_to_num   clip_seem_fusion.py:508-511
        float norm = sqrtf(wave_sum(ss));
        // SAF_NORM_L2_CLAMP: feat_norm.clamp_min_(0.1)   eval_scannet_segmentation.py:549-551, hypersim_eval.py:50-51
        if (normalize == SAF_NORM_L2_CLAMP) norm = norm < 0.1f ? 0.1f : norm;
        for (int c = lane; c < D; c += 64) {
          float q = row[c] / norm;
          if (normalize == SAF_NORM_L2) {
            if (q != q) q = 0.f;
            if (__builtin_isinf(q)) q = q > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
          }
          row[c] = q;
        }
      }
      for (int l = 0; l < L; ++l) {
        const float* t = text + (int64_t)l * tstride;
        float acc = 0.f;
        for (int c = lane; c < D; c += 64) acc = __builtin_fmaf(row[c], t[c], acc);
        acc = wave_sum(acc);
        if (lane == 0) sc[l] = acc;
      }
    }
    __syncthreads();  // sc[] written by lane 0, read by all lanes below
    if (active) {
      if (EPI == SAF_Q_SCORES) {
        for (int l = lane; l < L; l += 64) {
          const float val = scale * sc[l];
          if (out) out[r * L + l] = val;
          if (out_last && l == L - 1) out_last[r] = val;
        }
      } else if (EPI == SAF_Q_SOFTMAX) {
        // relevance = (100 * img_feats @ text.T).softmax(-1)            clipfusion.py:902-903
        float m = -INFINITY;
        for (int l = lane; l < L; l += 64) m = fmaxf(m, scale * sc[l]);
        m = wave_max(m);
        float sum = 0.f;
        for (int l = lane; l < L; l += 64) sum += expf(scale * sc[l] - m);
        sum = wave_sum(sum);
        for (int l = lane; l < L; l += 64) {
          const float val = expf(scale * sc[l] - m) / sum;
          if (out) out[r * L + l] = val;
          if (out_last && l == L - 1) out_last[r] = val;
        }
      } else if (EPI == SAF_Q_SURGERY) {
        // feats = F*T*w ; similarity = sum_c(feats - mean_t feats) = S*w - mean_t(S*w)   :924-932
        float part = 0.f;
        for (int l = lane; l < L; l += 64) part += sc[l] * wts[l];
        const float mean = wave_sum(part) / (float)L;
        for (int l = lane; l < L; l += 64) {
          const float val = sc[l] * wts[l] - mean;
          if (out) out[r * L + l] = val;
          if (out_last && l == L - 1) out_last[r] = val;
        }
      } else {
        // prob = softmax(2 * F[0] @ T.t()) ; w = prob / prob.mean()     clipfusion.py:913-915
        float m = -INFINITY;
        for (int l = lane; l < L; l += 64) m = fmaxf(m, 2.0f * sc[l]);
        m = wave_max(m);
        float sum = 0.f;
        for (int l = lane; l < L; l += 64) sum += expf(2.0f * sc[l] - m);
        sum = wave_sum(sum);
        float psum = 0.f;
        for (int l = lane; l < L; l += 64) psum += expf(2.0f * sc[l] - m) / sum;
        const float pmean = wave_sum(psum) / (float)L;
        for (int l = lane; l < L; l += 64) wts[l] = (expf(2.0f * sc[l] - m) / sum) / pmean;
      }
    }
    __syncthreads();  // sc[]/row[] are reused by the next iteration
  }
}

template <int EPI, int FT>
int launch_t(const void* feats, int64_t n_rows, int64_t fstride, int D, const float* text, int L, int64_t tstride,
           float scale, int normalize, float* wts, float* out, float* out_last, hipStream_t s) {
  const size_t shmem = (size_t)kQWaves * (((D + 3) & ~3) + ((L + 3) & ~3)) * sizeof(float);
  if (shmem > 150 * 1024) return fail(SAF_E_UNSUPPORTED, "feat_dim + n_text too large for the v1 scan (%zu B LDS)", shmem);
  auto fn = query_kernel<EPI, FT>;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)shmem);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  int64_t blocks = (n_rows + kQWaves - 1) / kQWaves;
  const int64_t cap = (int64_t)device_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kQThreads), shmem, s, feats, n_rows, fstride, D, text, L,
                     tstride, scale, normalize, wts, out, out_last);
  return check_launch("query_kernel");
}

// ------------------------------------------------------------------------------------------
// MFMA scan (feat_dim % 8 == 0, n_text <= 64): exact-fp32 matrix cores.
//
// A wave owns 32 feature rows at a time.  v_mfma_f32_32x32x2_f32 takes A[row = lane & 31][k = lane >> 5]
// and B[k = lane >> 5][col = lane & 31], one value per lane; the k order inside a dot product is
// free, so lane half h takes k = 8c + 4h + j (j = 0..3) of every 8-wide K chunk c: each lane then
// reads 16 contiguous bytes of ITS row straight from HBM (no LDS staging of the features, every
// byte read once), and 16 contiguous bytes of ITS text row from LDS.  Text rows are padded to
// D + 4 floats: (D/4 + 1) is odd, so the 16 lanes of a ds_read_b128 group hit 16 distinct 4-bank
// slots.  Row norms come from the same registers (sum of squares of what the lane loaded, the two
// halves added), and the epilogue works on the accumulator layout (col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)): row-wise max / sum are 5-step xor shuffles inside
// each 32-lane half.  The result of an MFMA chain is bit-for-bit an fmaf chain in k order.
// Bytes: N*D*s in, N*L*4 out -- HBM-bound for L <= 32, about balanced at L = 64.
// ------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FT>
__device__ __forceinline__ float4 load_feat4(const void* __restrict__ base, int64_t i) {
  if (FT == SAF_F32) return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + i);
  const uint2 w = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(base) + i);
  if (FT == SAF_BF16)
    return make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 a = __builtin_bit_cast(h2, w.x), b = __builtin_bit_cast(h2, w.y);
  return make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
}

#ifndef SAF_QM_GROUP
#define SAF_QM_GROUP 8
#endif
#ifndef SAF_QM_SLOTS
#define SAF_QM_SLOTS 2  // register groups of the row ring (one is multiplied, the others are in flight)
#endif
#ifndef SAF_QM_NT
#define SAF_QM_NT 0     // 1: the rows as non-temporal loads (they are read once)
#endif
#ifndef SAF_QM_ABL
#define SAF_QM_ABL 0    // development (WRONG results): 1 = no row loads (constants), 2 = no MFMAs
#endif
constexpr int kQGroup = SAF_QM_GROUP;  // K chunks (of 8 floats) per prefetch group: 64 floats of every row
constexpr int kQSlots = SAF_QM_SLOTS;

template <int FT>
__device__ __forceinline__ float4 load_row4(const void* __restrict__ base, int64_t i) {  // the scan's row stream (knobs above)
#if SAF_QM_ABL & 1
  return make_float4((float)(i & 7), 1.0f, 0.5f, 0.25f);
#else
  if (SAF_QM_NT && FT == SAF_F32) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(static_cast<const float*>(base) + i));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return load_feat4<FT>(base, i);
#endif
}

// Reductions over the 32 lanes of a half without the LDS (round 6; rounds 1-5: five ds_bpermute steps each, 16 to 32 chains per
// tile): quads, eights and sixteens by DPP inside the adds, the two 16-lane rows of the half by one v_permlane16_swap -- swapping a
// value's odd rows with its own even rows leaves (row 0, row 0, row 2, row 2) and (row 1, row 1, row 3, row 3).
template <int CTRL>
__device__ __forceinline__ float dpp_lane(float x) {  // quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float half_max(float x) {  // over the 32 lanes of this half
  x = fmaxf(x, dpp_lane<0xB1>(x));
  x = fmaxf(x, dpp_lane<0x4E>(x));
  x = fmaxf(x, dpp_lane<0x141>(x));
  x = fmaxf(x, dpp_lane<0x140>(x));
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x), __builtin_bit_cast(uint32_t, x), false, false);
  return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
  x += dpp_lane<0xB1>(x);
  x += dpp_lane<0x4E>(x);
  x += dpp_lane<0x141>(x);
  x += dpp_lane<0x140>(x);
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, x), __builtin_bit_cast(uint32_t, x), false, false);
  return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
}
// a value and its partner's 32 lanes away: v_permlane32_swap of a register with itself leaves (lower, lower) and (upper, upper)
__device__ __forceinline__ float other_half_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, x), __builtin_bit_cast(uint32_t, x), false, false);
  return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}
__device__ __forceinline__ float other_half_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, x), __builtin_bit_cast(uint32_t, x), false, false);
  return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
}
// lane `src` (a constant below 28) for the lower half of the wave, lane `src + 4` for the upper: the accumulator layout's row of
// register i in either half -- two v_readlane and a select instead of a ds_bpermute
template <typename T>
__device__ __forceinline__ T row_of_half(T v, int src, int h) {
  const int lo = __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src), hi = __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src + 4);
  return __builtin_bit_cast(T, h ? hi : lo);
}

// ---- epilogue on the MFMA accumulator layout: this lane holds column n = m (+32 t) of 16 rows.  `inv`: the factor of row m
// (1 / norm); CS (the split scan): the row's and the columns' power-of-two scales 2^rexp, 2^cexp undone by one v_ldexp_f32,
// which cannot leave fp32's range half-way the way two multiplications can.
__device__ __forceinline__ float row_inverse(float ss, int normalize) {
  // clip_feat /= norm ; nan_to_num: an all-zero row gives zeros       clip_seem_fusion.py:508-511
  // SAF_NORM_L2_CLAMP: norm.clamp_min(0.1)                            eval_scannet_segmentation.py:549-551
  return normalize == SAF_NORM_L2_CLAMP ? 1.0f / fmaxf(sqrtf(ss), 0.1f) : (normalize ? (ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f) : 1.0f);
}

template <int EPI, int TILES, bool CS>
__device__ __forceinline__ void scan_epilogue(const f32x16 (&acc)[TILES], float inv, int rexp, const int (&cexp)[TILES], int64_t tile,
                                              int64_t n_rows, int L, float scale, const float (&wl)[TILES],
                                              float* __restrict__ out, float* __restrict__ out_last, int64_t out_stride,
                                              int out_col0, float* __restrict__ stage = nullptr) {
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  // `stage` (this wave's 8 x L floats of LDS, or null): the [N, L] matrix leaves as whole 16-byte pieces of 8 rows at a time --
  // registers 4j .. 4j + 3 hold rows 8j .. 8j + 7 of the tile, 32 L contiguous bytes of the output -- instead of a row's 32
  // columns per store, which for L = 63 begin at every 4-byte phase of a line (raw scores, L = 63: 4.21 -> 4.02 ms per 2^23 rows;
  // softmax / surgery: no change, their stores hide in the epilogue).  Full tiles only.
  const bool staged = stage != nullptr && (tile + 1) * 32 <= n_rows;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int mi = (i & 3) + 8 * (i >> 2) + 4 * h;  // row of accumulator register i in this half
    const float inv_i = row_of_half(inv, (i & 3) + 8 * (i >> 2), h);
    const int rexp_i = CS ? row_of_half(rexp, (i & 3) + 8 * (i >> 2), h) : 0;
    const int64_t r = tile * 32 + mi;
    float val[TILES];
    bool ok[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      ok[t] = (m + 32 * t) < L;
      val[t] = acc[t][i] * inv_i;  // cosine score S[r][n]
      if (CS) val[t] = ldexpf(val[t], -(rexp_i + cexp[t]));
    }
    if (EPI == SAF_Q_SOFTMAX) {
      // relevance = (100 * img_feats @ text.T).softmax(-1)            clipfusion.py:902-903
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < TILES; ++t) mx = fmaxf(mx, ok[t] ? scale * val[t] : -INFINITY);
      mx = half_max(mx);
      float e[TILES], sum = 0.0f;
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        e[t] = ok[t] ? expf(scale * val[t] - mx) : 0.0f;
        sum += e[t];
      }
      sum = half_sum(sum);
#pragma unroll
      for (int t = 0; t < TILES; ++t) val[t] = e[t] / sum;
    } else if (EPI == SAF_Q_SURGERY) {
      // similarity = S*w - mean_t(S*w)                                 clipfusion.py:924-932
      float part = 0.0f;
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        val[t] = ok[t] ? val[t] * wl[t] : 0.0f;
        part += val[t];
      }
      const float mean = half_sum(part) / (float)L;
#pragma unroll
      for (int t = 0; t < TILES; ++t) val[t] -= mean;
    } else {
#pragma unroll
      for (int t = 0; t < TILES; ++t) val[t] = val[t] * scale * wl[t];  // (weights: a block of a wide surgery scan; else 1)
    }
    if (r < n_rows) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        if (ok[t]) {
          if (staged) stage[((i & 3) + 4 * h) * L + m + 32 * t] = val[t];
          else if (out) out[r * out_stride + out_col0 + m + 32 * t] = val[t];
          if (out_last && m + 32 * t == L - 1) out_last[r] = val[t];
        }
      }
    }
    if (staged && (i & 3) == 3) {  // (wave-uniform)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float4* dst = reinterpret_cast<float4*>(out + (tile * 32 + 8 * (i >> 2)) * (int64_t)L);
      for (int q = lane; q < 2 * L; q += 64) dst[q] = reinterpret_cast<const float4*>(stage)[q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// TH threads per workgroup: 256, or 512 where the text tiles leave room for only ONE workgroup per CU (two 32-row tiles at D = 512:
// 132 KB) -- four waves would be one per SIMD, each alone with its loads, LDS reads and MFMA chain: the L = 63 surgery scan over
// the 256^3 x 512 fp32 volume 14.8 -> 12.0 ms (tools/q_threads_ab.sh); 1024 threads spill (128 registers) and are slower (12.6).
template <int EPI, int FT, int TILES, int TH>
__global__ __launch_bounds__(TH) void query_mfma_kernel(const void* __restrict__ feats, int64_t n_rows,
                                                                int64_t fstride, int D, const float* __restrict__ text,
                                                                int L, int64_t tstride, float scale, int normalize,
                                                                const float* __restrict__ wts, float* __restrict__ out,
                                                                float* __restrict__ out_last, int64_t out_stride, int out_col0,
                                                                int /*stage_on: the split scan's*/) {
  // (out_stride / out_col0: where this launch's L columns lie in the output row -- L and 0, or one 64-label block of a wider row)
  extern __shared__ __attribute__((aligned(16))) float s_text[];  // [TILES*32][D + 4], rows >= L are zero
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tstr = D + 4;
  for (int i = tid; i < TILES * 32 * D; i += TH) {
    const int n = i / D, k = i - n * D;
    s_text[n * tstr + k] = n < L ? text[(int64_t)n * tstride + k] : 0.0f;
  }
  __syncthreads();
  const int m = lane & 31, h = lane >> 5;
  float wl[TILES];  // this lane's columns' weights (surgery; 1 without): read once, not per row (a load in the epilogue waits for
#pragma unroll      // every row load in flight before it)
  for (int t = 0; t < TILES; ++t) wl[t] = wts && m + 32 * t < L ? wts[m + 32 * t] : 1.0f;
  const int n_groups = D / (8 * kQGroup);  // full prefetch groups; the remainder is handled chunk by chunk
  const int64_t n_tiles = (n_rows + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * (TH / 64) + wave; tile < n_tiles; tile += (int64_t)gridDim.x * (TH / 64)) {
    int64_t row = tile * 32 + m;
    if (row >= n_rows) row = n_rows - 1;  // padded lanes recompute the last row, never stored
    const int64_t base = row * fstride + 4 * h;
    const float* tb = s_text + m * tstr + 4 * h;
    f32x16 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    float ss = 0.0f;
    // The wave's rows stream through a RING of kQSlots register groups (a group = kQGroup chunks of 8 floats of every row: 8 KB per
    // wave): kQSlots - 1 groups are in flight while one is multiplied.  Round 6: one group ahead (rounds 1-5) is 1.7 us of MFMAs --
    // less than the latency of an HBM request under this load --, and the L = 63 scan ran at matrix time + memory time.
    float4 ring[kQSlots][kQGroup];
#pragma unroll
    for (int sl = 0; sl + 1 < kQSlots; ++sl) {
      if (sl < n_groups) {
#pragma unroll
        for (int q = 0; q < kQGroup; ++q) ring[sl][q] = load_row4<FT>(feats, base + 8 * (sl * kQGroup + q));
      }
    }
    for (int g0 = 0; g0 < n_groups; g0 += kQSlots) {
#pragma unroll
      for (int sl = 0; sl < kQSlots; ++sl) {
        const int g = g0 + sl;
        if (g >= n_groups) break;
        if (g + kQSlots - 1 < n_groups) {
#pragma unroll
          for (int q = 0; q < kQGroup; ++q)
            ring[(sl + kQSlots - 1) % kQSlots][q] = load_row4<FT>(feats, base + 8 * ((g + kQSlots - 1) * kQGroup + q));
        }
#pragma unroll
        for (int q = 0; q < kQGroup; ++q) {
          const int k0 = 8 * (g * kQGroup + q);
          const float4 a = ring[sl][q];
          ss = __builtin_fmaf(a.x, a.x, ss);
          ss = __builtin_fmaf(a.y, a.y, ss);
          ss = __builtin_fmaf(a.z, a.z, ss);
          ss = __builtin_fmaf(a.w, a.w, ss);
#if !(SAF_QM_ABL & 2)
#pragma unroll
          for (int t = 0; t < TILES; ++t) {
            const float4 b = *reinterpret_cast<const float4*>(tb + t * 32 * tstr + k0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
          }
#endif
        }
      }
    }
    for (int k0 = 8 * kQGroup * n_groups; k0 < D; k0 += 8) {  // D not a multiple of 64
      const float4 a = load_feat4<FT>(feats, base + k0);
      ss = __builtin_fmaf(a.x, a.x, ss);
      ss = __builtin_fmaf(a.y, a.y, ss);
      ss = __builtin_fmaf(a.z, a.z, ss);
      ss = __builtin_fmaf(a.w, a.w, ss);
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        const float4 b = *reinterpret_cast<const float4*>(tb + t * 32 * tstr + k0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[t], 0, 0, 0);
      }
    }
    ss += __shfl_xor(ss, 32);  // both halves of row m
    const int none[TILES] = {};
    scan_epilogue<EPI, TILES, false>(acc, row_inverse(ss, normalize), 0, none, tile, n_rows, L, scale, wl, out, out_last, out_stride, out_col0);
  }
}

// ------------------------------------------------------------------------------------------
// Split scan (round 6; feat_dim % 16 == 0, n_text <= 64): the same scan with every fp32 operand cut into two fp16 pieces,
// x * 2^e = hi + lo (hi = fp16(x 2^e), lo = fp16(x 2^e - hi): 22 significant bits between them), and the dot products taken as
// hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation -- 3 x 32 cycles per 16 k and 32 x 32 outputs where
// v_mfma_f32_32x32x2_f32 takes 8 x 64: the L = 63 scan over an fp32 volume stops being bound by the fp32 matrix rate (9.65 ms of
// MFMAs at 256^3 x 512) and becomes what it should be, a stream of the volume.  What is dropped is lo.lo and the
// pieces' rounding: <= 3 x 2^-22 of sum_k |a_k b_k| per score -- the size of the differences between two fp32 summation orders
// of 512 terms -- against the 1e-4 the scores are held to.
//   * Text rows: scaled per label by the power of two that brings the label's largest magnitude into [2^13, 2^14), cut once into
//     LDS in the order the lanes read them (a lane's 8 + 8 halfs of a k-step are 32 contiguous bytes), the scale undone per
//     column in the epilogue.
//   * Feature rows: a row's scale follows its running maximum like the running maximum of an online softmax: it is set from the
//     first 64 features that are not all zero (largest magnitude into [2^13, 2^14)) and, should a later group of 64 exceed 2^15
//     under it, lowered, the row's accumulators multiplied by the same power of two (exact).  No value overflows fp16, none
//     that matters falls below its normal range (a feature 2^-16 of the row's largest still has its hi piece whole), and a row
//     of any magnitude fp32 holds is scanned like any other.  Row norms come from the unscaled fp32 values as before.
// Same loads, same register ring, same epilogue as the fp32 MFMA scan above (SAF_Q_SPLIT=0 selects that one).  Measured at
// 256^3 x 512 fp32 (profiles/r06/split_scan_ab.txt): L = 63 surgery 10.95 -> 7.6 ms (round 5: 12.2), L = 5 softmax 6.9 -> 5.9 (reductions without the LDS).
// ------------------------------------------------------------------------------------------
#ifndef SAF_QS_ABL
#define SAF_QS_ABL 0
#endif
#ifndef SAF_QS_STAGE
#define SAF_QS_STAGE 1  // 0: the score matrix stored a row's 32 columns at a time (A/B)
#endif
#ifndef SAF_QS_CHAIN
#define SAF_QS_CHAIN 1  // 0: a fresh ring per tile, the next group's loads behind a condition (rounds 1-6: every wait a vmcnt(0..7))
#endif
typedef _Float16 h8x __attribute__((ext_vector_type(8)));
typedef _Float16 h2x __attribute__((ext_vector_type(2)));

// the exponent e that brings a largest magnitude `mx` (finite, > 0; denormals too) into [2^13, 2^14) as mx * 2^e
__device__ __forceinline__ int split_exponent(float mx) {
  int ex;
  (void)frexpf(mx, &ex);  // mx = f * 2^ex, 0.5 <= f < 1
  return 14 - ex;
}
__device__ __forceinline__ void split2(float y0, float y1, uint32_t& hi, uint32_t& lo) {
  const h2x hv = {(_Float16)y0, (_Float16)y1};
  const h2x lv = {(_Float16)(y0 - (float)hv.x), (_Float16)(y1 - (float)hv.y)};
  hi = __builtin_bit_cast(uint32_t, hv);
  lo = __builtin_bit_cast(uint32_t, lv);
}

template <int EPI, int FT, int TILES, int TH>
__global__ __launch_bounds__(TH, 2) void query_split_kernel(const void* __restrict__ feats, int64_t n_rows, int64_t fstride, int D,
                                                         const float* __restrict__ text, int L, int64_t tstride, float scale,
                                                         int normalize, const float* __restrict__ wts, float* __restrict__ out,
                                                         float* __restrict__ out_last, int64_t out_stride, int out_col0,
                                                         int stage_on) {
  // [TILES*32] label rows of D/16 k-steps x {half 0, half 1} x {hi, lo} x 8 halfs (+16 bytes: the rows' reads fall on different
  // banks), then the labels' scale exponents
  extern __shared__ __attribute__((aligned(16))) unsigned char s_split[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rs_bytes = 4 * D + 16;
  int* s_ce = reinterpret_cast<int*>(s_split + (size_t)TILES * 32 * rs_bytes);
  float* stage = stage_on ? reinterpret_cast<float*>(s_ce + TILES * 32) + (size_t)wave * (TILES * 256) : nullptr;  // 8 rows x <= 32 TILES floats
  for (int n = wave; n < TILES * 32; n += TH / 64) {
    float mx = 0.0f;
    if (n < L)
      for (int k = lane; k < D; k += 64) mx = fmaxf(mx, fabsf(text[(int64_t)n * tstride + k]));
    mx = wave_max(mx);
    const int be = mx > 0.0f && mx < INFINITY ? split_exponent(mx) : 0;
    if (lane == 0) s_ce[n] = be;
    for (int k = lane; k < D; k += 64) {
      const float y = n < L ? ldexpf(text[(int64_t)n * tstride + k], be) : 0.0f;
      const _Float16 hi = (_Float16)y, lo = (_Float16)(y - (float)hi);
      const int p = k >> 4, u = (k >> 3) & 1, hh = (k >> 2) & 1, j = k & 3;
      _Float16* dst = reinterpret_cast<_Float16*>(s_split + (size_t)n * rs_bytes + p * 64 + hh * 32) + 4 * u + j;
      dst[0] = hi;
      dst[8] = lo;
    }
  }
  __syncthreads();
  const int m = lane & 31, h = lane >> 5;
  float wl[TILES];  // this lane's columns' weights (surgery; 1 without): read once, not per row (a load in the epilogue waits for
#pragma unroll      // every row load in flight before it)
  for (int t = 0; t < TILES; ++t) wl[t] = wts && m + 32 * t < L ? wts[m + 32 * t] : 1.0f;
  int ce[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) ce[t] = s_ce[m + 32 * t];
  const int n_groups = D / (8 * kQGroup);  // full prefetch groups; the remainder is handled pair by pair
  const int64_t n_tiles = (n_rows + 31) / 32;
  const unsigned char* tb = s_split + (size_t)m * rs_bytes + h * 32;
  // The ring, made to work (round 6, late): with the next group's loads behind a condition (`is there a next group?`) the compiler's
  // wait-count pass cannot know how many loads are younger than the group it is about to multiply and waits for ALL of them --
  // vmcnt(7) .. vmcnt(0) in every build of rounds 1-6, the "ring" one group deep and nothing in flight under the MFMAs (which is
  // why three slots, or four, never changed anything).  Where a row is a whole number of PAIRS of groups (feat_dim a multiple of
  // 128) the two slots are loaded by unconditional code -- behind a tile's last group comes the wave's NEXT tile's first (behind
  // the last tile: a row nobody uses) -- and the waits count 8 younger loads: vmcnt(15) .. vmcnt(8).
  const bool chain = SAF_QS_CHAIN && kQSlots == 2 && n_groups >= 2 && n_groups % 2 == 0;
  const int64_t tile0 = (int64_t)blockIdx.x * (TH / 64) + wave, tile_step = (int64_t)gridDim.x * (TH / 64);
  auto row_base = [&](int64_t tile) {
    int64_t row = tile * 32 + m;
    if (row >= n_rows) row = n_rows - 1;  // padded lanes recompute the last row, never stored
    return row * fstride + 4 * h;
  };
  float4 ring[kQSlots][kQGroup];
  if (chain && tile0 < n_tiles) {
    const int64_t base = row_base(tile0);
#pragma unroll
    for (int sl = 0; sl + 1 < kQSlots; ++sl)
#pragma unroll
      for (int q = 0; q < kQGroup; ++q) ring[sl][q] = load_row4<FT>(feats, base + 8 * (sl * kQGroup + q));
  }
  for (int64_t tile = tile0; tile < n_tiles; tile += tile_step) {
    const int64_t base = row_base(tile);
    const int64_t base_next = row_base(tile + tile_step < n_tiles ? tile + tile_step : tile);  // (behind the last tile: loads nobody uses)
    f32x16 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    float ss = 0.0f;
    int re = 0;         // the row's scale is 2^re (applied by v_ldexp_f32: any exponent, denormal features included)
    bool seen = false;  // a feature of the row was not zero
    // a group's largest magnitude `gm` (this half's part) against the row's scale, before the group is cut
    auto fit = [&](float gm) {
      gm = other_half_max(gm);
      const bool need = ldexpf(gm, re) >= 32768.0f || (!seen && gm > 0.0f);
      if (__builtin_amdgcn_ballot_w64(need)) {
        const int ne = need ? (gm < INFINITY ? split_exponent(gm) : 0) : re;
        const float fac = need && seen ? ldexpf(1.0f, ne - re) : 1.0f;  // (0 once the row has grown by more than fp32's range)
        re = ne;
        seen = seen || gm > 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float f = __shfl(fac, (i & 3) + 8 * (i >> 2) + 4 * h);
#pragma unroll
          for (int t = 0; t < TILES; ++t) acc[t][i] *= f;
        }
      }
    };
    // one k-step: chunks 2p and 2p + 1 of the lane's row (k = 16p + 8u + 4h + j)
    auto step = [&](const float4& a0, const float4& a1, int p) {
      ss = __builtin_fmaf(a0.x, a0.x, ss);
      ss = __builtin_fmaf(a0.y, a0.y, ss);
      ss = __builtin_fmaf(a0.z, a0.z, ss);
      ss = __builtin_fmaf(a0.w, a0.w, ss);
      ss = __builtin_fmaf(a1.x, a1.x, ss);
      ss = __builtin_fmaf(a1.y, a1.y, ss);
      ss = __builtin_fmaf(a1.z, a1.z, ss);
      ss = __builtin_fmaf(a1.w, a1.w, ss);
      uint4 hi, lo;
#if SAF_QS_ABL & 2  // (development, WRONG results: the features' bits as they are)
      hi = make_uint4(__builtin_bit_cast(uint32_t, a0.x), __builtin_bit_cast(uint32_t, a0.y), __builtin_bit_cast(uint32_t, a0.z), __builtin_bit_cast(uint32_t, a0.w));
      lo = make_uint4(__builtin_bit_cast(uint32_t, a1.x), __builtin_bit_cast(uint32_t, a1.y), __builtin_bit_cast(uint32_t, a1.z), __builtin_bit_cast(uint32_t, a1.w));
#else
      split2(ldexpf(a0.x, re), ldexpf(a0.y, re), hi.x, lo.x);
      split2(ldexpf(a0.z, re), ldexpf(a0.w, re), hi.y, lo.y);
      split2(ldexpf(a1.x, re), ldexpf(a1.y, re), hi.z, lo.z);
      split2(ldexpf(a1.z, re), ldexpf(a1.w, re), hi.w, lo.w);
#endif
      const h8x ah = __builtin_bit_cast(h8x, hi), al = __builtin_bit_cast(h8x, lo);
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        const uint4* bp = reinterpret_cast<const uint4*>(tb + (size_t)t * 32 * rs_bytes + p * 64);
        const uint4 bc[2] = {bp[0], bp[1]};
        const h8x bh = __builtin_bit_cast(h8x, bc[0]), bl = __builtin_bit_cast(h8x, bc[1]);
#if SAF_QS_ABL & 1  // (development, WRONG results: no MFMAs, the pieces kept alive by four integer operations)
        acc[t][0] += __builtin_bit_cast(float, (hi.x ^ lo.y ^ bc[0].x) & 0x3fffffffu);
        acc[t][1] += __builtin_bit_cast(float, (hi.z ^ lo.w ^ bc[1].x) & 0x3fffffffu);
        acc[t][2] += __builtin_bit_cast(float, (hi.y ^ lo.x ^ hi.w ^ lo.z) & 0x3fffffffu);
#else
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
#endif
      }
    };
    auto multiply = [&](const float4 (&grp)[kQGroup], int g) {  // one group of a row: its scale, then its k-steps
      float gm = 0.0f;
#pragma unroll
      for (int q = 0; q < kQGroup; ++q) {
        const float4 a = grp[q];
        gm = fmaxf(fmaxf(gm, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
      }
      fit(gm);
#pragma unroll
      for (int q = 0; q < kQGroup; q += 2) step(grp[q], grp[q + 1], (g * kQGroup + q) >> 1);
    };
    if (chain) {
      for (int g = 0; g < n_groups; g += 2) {
#pragma unroll
        for (int q = 0; q < kQGroup; ++q) ring[1 % kQSlots][q] = load_row4<FT>(feats, base + 8 * ((g + 1) * kQGroup + q));
        multiply(ring[0], g);
        const int64_t from = g + 2 < n_groups ? base + 8 * ((g + 2) * kQGroup) : base_next;
#pragma unroll
        for (int q = 0; q < kQGroup; ++q) ring[0][q] = load_row4<FT>(feats, from + 8 * q);
        multiply(ring[1 % kQSlots], g + 1);
      }
    } else {
#pragma unroll
      for (int sl = 0; sl + 1 < kQSlots; ++sl) {
        if (sl < n_groups) {
#pragma unroll
          for (int q = 0; q < kQGroup; ++q) ring[sl][q] = load_row4<FT>(feats, base + 8 * (sl * kQGroup + q));
        }
      }
      for (int g0 = 0; g0 < n_groups; g0 += kQSlots) {
#pragma unroll
        for (int sl = 0; sl < kQSlots; ++sl) {
          const int g = g0 + sl;
          if (g >= n_groups) break;
          const int gl = g + kQSlots - 1;
          if (gl < n_groups) {
#pragma unroll
            for (int q = 0; q < kQGroup; ++q) ring[(sl + kQSlots - 1) % kQSlots][q] = load_row4<FT>(feats, base + 8 * (gl * kQGroup + q));
          }
          multiply(ring[sl], g);
        }
      }
    }
    for (int k0 = 8 * kQGroup * n_groups; k0 < D; k0 += 16) {  // D not a multiple of 64
      const float4 a0 = load_feat4<FT>(feats, base + k0), a1 = load_feat4<FT>(feats, base + k0 + 8);
      fit(fmaxf(fmaxf(fmaxf(fabsf(a0.x), fabsf(a0.y)), fmaxf(fabsf(a0.z), fabsf(a0.w))),
               fmaxf(fmaxf(fabsf(a1.x), fabsf(a1.y)), fmaxf(fabsf(a1.z), fabsf(a1.w)))));
      step(a0, a1, k0 >> 4);
    }
    ss = other_half_sum(ss);  // both halves of row m
    scan_epilogue<EPI, TILES, true>(acc, row_inverse(ss, normalize), re, ce, tile, n_rows, L, scale, wl, out, out_last,
                                    out_stride, out_col0, stage);
  }
}

// ------------------------------------------------------------------------------------------
// The split scan over a 16-BIT volume (bf16: BASELINE config 3's; fp16).  Through the kernel above a 16-bit row is widened to fp32
// and cut again -- the same 45 vector instructions per k-step for half the bytes, and 8-byte loads: L = 63 over a 256^3 x 512 bf16
// volume took 6.9 ms where the fp32 volume takes 7.6.  Here a lane's 8 values of a k-step ARE one 16-byte load in the matrix
// instruction's own layout (k = 16 p + 8 h + j: the labels' pieces lie in LDS in that order), and a feature needs ONE fp16 piece:
//   * fp16: the value itself -- no conversion at all, the row norm from v_fma_mix_f32 (v_dot2_f32_f16 was 3 % off: not used);
//   * bf16: its 8 significant bits fit fp16's 11, so `fp16(x * 2^e)` is exact wherever it is normal -- under the running
//     power-of-two scale of the kernel above (largest magnitude of the row so far in [2^13, 2^14)) that is every feature down to
//     2^-27 of the row's largest; below that fp16's denormal spacing leaves an error of 2^-38 of the row's largest.
// Two MFMAs per 16 k and 32 labels (a.hi, a.lo of the LABEL's two pieces); the score's error is the labels' cut alone, 2^-22.
// ------------------------------------------------------------------------------------------
constexpr int kQGroup16 = 8;  // k-steps (16 bytes of every row each) per prefetch group: 128 features

template <int EPI, int FT, int TILES, int TH>
__global__ __launch_bounds__(TH, 2) void query_split16_kernel(const void* __restrict__ feats, int64_t n_rows, int64_t fstride, int D,
                                                           const float* __restrict__ text, int L, int64_t tstride, float scale,
                                                           int normalize, const float* __restrict__ wts, float* __restrict__ out,
                                                           float* __restrict__ out_last, int64_t out_stride, int out_col0,
                                                           int stage_on) {
  static_assert(FT == SAF_F16 || FT == SAF_BF16, "16-bit volumes");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_split[];  // as query_split_kernel's, the pieces in plain k order
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rs_bytes = 4 * D + 16;
  int* s_ce = reinterpret_cast<int*>(s_split + (size_t)TILES * 32 * rs_bytes);
  float* stage = stage_on ? reinterpret_cast<float*>(s_ce + TILES * 32) + (size_t)wave * (TILES * 256) : nullptr;
  for (int n = wave; n < TILES * 32; n += TH / 64) {
    float mx = 0.0f;
    if (n < L)
      for (int k = lane; k < D; k += 64) mx = fmaxf(mx, fabsf(text[(int64_t)n * tstride + k]));
    mx = wave_max(mx);
    const int be = mx > 0.0f && mx < INFINITY ? split_exponent(mx) : 0;
    if (lane == 0) s_ce[n] = be;
    for (int k = lane; k < D; k += 64) {
      const float y = n < L ? ldexpf(text[(int64_t)n * tstride + k], be) : 0.0f;
      const _Float16 hi = (_Float16)y, lo = (_Float16)(y - (float)hi);
      _Float16* dst = reinterpret_cast<_Float16*>(s_split + (size_t)n * rs_bytes + (k >> 4) * 64 + ((k >> 3) & 1) * 32) + (k & 7);
      dst[0] = hi;
      dst[8] = lo;
    }
  }
  __syncthreads();
  const int m = lane & 31, h = lane >> 5;
  float wl[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) wl[t] = wts && m + 32 * t < L ? wts[m + 32 * t] : 1.0f;
  int ce[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) ce[t] = s_ce[m + 32 * t];
  const int n_steps = D >> 4, n_groups = n_steps / kQGroup16;
  const int64_t n_tiles = (n_rows + 31) / 32;
  const unsigned char* tb = s_split + (size_t)m * rs_bytes + h * 32;
  const uint16_t* f16 = static_cast<const uint16_t*>(feats);
  typedef unsigned int q16_u4 __attribute__((ext_vector_type(4)));
  const bool chain = SAF_QS_CHAIN && kQSlots == 2 && n_groups >= 2 && n_groups % 2 == 0;
  const int64_t tile0 = (int64_t)blockIdx.x * (TH / 64) + wave, tile_step = (int64_t)gridDim.x * (TH / 64);
  q16_u4 ring[kQSlots][kQGroup16];
  if (chain && tile0 < n_tiles) {
    int64_t row = tile0 * 32 + m;
    if (row >= n_rows) row = n_rows - 1;
#pragma unroll
    for (int q = 0; q < kQGroup16; ++q) ring[0][q] = *reinterpret_cast<const q16_u4*>(f16 + row * fstride + 8 * h + 16 * q);
  }
  for (int64_t tile = tile0; tile < n_tiles; tile += tile_step) {
    int64_t row = tile * 32 + m;
    if (row >= n_rows) row = n_rows - 1;  // padded lanes recompute the last row, never stored
    const uint16_t* rp = f16 + row * fstride + 8 * h;
    f32x16 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    float ss = 0.0f;
    int re = 0;         // bf16: the row's scale is 2^re, following its running maximum (query_split_kernel); fp16: none
    bool seen = false;
    auto fit = [&](float gm) {
      gm = other_half_max(gm);
      const bool need = ldexpf(gm, re) >= 32768.0f || (!seen && gm > 0.0f);
      if (__builtin_amdgcn_ballot_w64(need)) {
        const int ne = need ? (gm < INFINITY ? split_exponent(gm) : 0) : re;
        const float fac = need && seen ? ldexpf(1.0f, ne - re) : 1.0f;
        re = ne;
        seen = seen || gm > 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float f = __shfl(fac, (i & 3) + 8 * (i >> 2) + 4 * h);
#pragma unroll
          for (int t = 0; t < TILES; ++t) acc[t][i] *= f;
        }
      }
    };
    auto widen = [](uint32_t w, float& x0, float& x1) {  // two bf16 -> fp32
      x0 = __builtin_bit_cast(float, w << 16);
      x1 = __builtin_bit_cast(float, w & 0xffff0000u);
    };
    // (bf16) the largest magnitude of a group, in the integer domain: below the sign bit a bf16's bits order its magnitude, so the
    // group's maximum is 32 v_and + v_pk_max_u16 and ONE conversion -- widening the values here as well kept a second copy of the
    // group alive until its k-steps (the optimiser shares the widening): 105 spilled registers at two label tiles
    auto group_max_bits = [&](const q16_u4& a, uint32_t mx) {
      typedef unsigned short us2 __attribute__((ext_vector_type(2)));
      const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const us2 u = __builtin_bit_cast(us2, w[j] & 0x7fff7fffu), v = __builtin_bit_cast(us2, mx);
        const us2 r = {u.x > v.x ? u.x : v.x, u.y > v.y ? u.y : v.y};
        mx = __builtin_bit_cast(uint32_t, r);
      }
      return mx;
    };
    auto bits_max = [](uint32_t mx) {  // the larger half as fp32 (a NaN among the values: the row is NaN whatever its scale)
      const uint32_t lo = mx & 0xffffu, hi = mx >> 16;
      return __builtin_bit_cast(float, (lo > hi ? lo : hi) << 16);
    };
    auto step = [&](const q16_u4& a, int p) {
      q16_u4 hi;
      if (FT == SAF_F16) {
        hi = a;
        // (the words by name: `a[j]` under the unrolled loop read word 0 four times here -- the row norms of fp16 volumes were off
        //  by a few percent until the parity tests said so)
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const h2x v = __builtin_bit_cast(h2x, w[j]);  // (v_fma_mix_f32: fp16 operands, fp32 product and sum)
          ss = __builtin_fmaf((float)v.x, (float)v.x, ss);
          ss = __builtin_fmaf((float)v.y, (float)v.y, ss);
        }
      } else {
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x0, x1;
          widen(w[j], x0, x1);
          ss = __builtin_fmaf(x0, x0, ss);
          ss = __builtin_fmaf(x1, x1, ss);
          const h2x v = {(_Float16)ldexpf(x0, re), (_Float16)ldexpf(x1, re)};
          hi[j] = __builtin_bit_cast(uint32_t, v);
        }
      }
      const h8x ah = __builtin_bit_cast(h8x, hi);
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        const q16_u4* bp = reinterpret_cast<const q16_u4*>(tb + (size_t)t * 32 * rs_bytes + p * 64);
        const h8x bh = __builtin_bit_cast(h8x, bp[0]), bl = __builtin_bit_cast(h8x, bp[1]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
      }
    };
    auto multiply = [&](const q16_u4 (&grp)[kQGroup16], int g) {
      if (FT == SAF_BF16) {
        uint32_t mx = 0u;
#pragma unroll
        for (int q = 0; q < kQGroup16; ++q) mx = group_max_bits(grp[q], mx);
        fit(bits_max(mx));
      }
#pragma unroll
      for (int q = 0; q < kQGroup16; ++q) step(grp[q], g * kQGroup16 + q);
    };
    if (chain) {  // (see query_split_kernel: unconditional loads, counted waits)
      int64_t nrow = (tile + tile_step < n_tiles ? tile + tile_step : tile) * 32 + m;
      if (nrow >= n_rows) nrow = n_rows - 1;
      const uint16_t* rpn = f16 + nrow * fstride + 8 * h;
      for (int g = 0; g < n_groups; g += 2) {
#pragma unroll
        for (int q = 0; q < kQGroup16; ++q) ring[1 % kQSlots][q] = *reinterpret_cast<const q16_u4*>(rp + 16 * ((g + 1) * kQGroup16 + q));
        multiply(ring[0], g);
        const uint16_t* from = g + 2 < n_groups ? rp + 16 * ((g + 2) * kQGroup16) : rpn;
#pragma unroll
        for (int q = 0; q < kQGroup16; ++q) ring[0][q] = *reinterpret_cast<const q16_u4*>(from + 16 * q);
        multiply(ring[1 % kQSlots], g + 1);
      }
    } else {
#pragma unroll
      for (int sl = 0; sl + 1 < kQSlots; ++sl) {
        if (sl < n_groups) {
#pragma unroll
          for (int q = 0; q < kQGroup16; ++q) ring[sl][q] = *reinterpret_cast<const q16_u4*>(rp + 16 * (sl * kQGroup16 + q));
        }
      }
      for (int g0 = 0; g0 < n_groups; g0 += kQSlots) {
#pragma unroll
        for (int sl = 0; sl < kQSlots; ++sl) {
          const int g = g0 + sl;
          if (g >= n_groups) break;
          const int gl = g + kQSlots - 1;
          if (gl < n_groups) {
#pragma unroll
            for (int q = 0; q < kQGroup16; ++q)
              ring[(sl + kQSlots - 1) % kQSlots][q] = *reinterpret_cast<const q16_u4*>(rp + 16 * (gl * kQGroup16 + q));
          }
          multiply(ring[sl], g);
        }
      }
    }
    for (int p = n_groups * kQGroup16; p < n_steps; ++p) {  // D not a multiple of 128
      const q16_u4 a = *reinterpret_cast<const q16_u4*>(rp + 16 * p);
      if (FT == SAF_BF16) fit(bits_max(group_max_bits(a, 0u)));
      step(a, p);
    }
    ss = other_half_sum(ss);  // both halves of row m
    scan_epilogue<EPI, TILES, true>(acc, row_inverse(ss, normalize), re, ce, tile, n_rows, L, scale, wl, out, out_last,
                                    out_stride, out_col0, stage);
  }
}

template <int EPI, int FT, int TILES, int TH, int SPLIT>  // SPLIT: 0 the exact-fp32 scan, 1 the split scan, 2 its 16-bit-volume form
int launch_mfma_th(const void* feats, int64_t n_rows, int64_t fstride, int D, const float* text, int L, int64_t tstride,
                   float scale, int normalize, const float* wts, float* out, float* out_last, int per_cu, size_t shmem, hipStream_t s,
                   int64_t out_stride, int out_col0) {
  auto fn = SPLIT == 2 ? query_split16_kernel<EPI, (FT == SAF_F32 ? SAF_F16 : FT), TILES, TH>
            : SPLIT ? query_split_kernel<EPI, FT, TILES, TH> : query_mfma_kernel<EPI, FT, TILES, TH>;
  constexpr int kWavesTh = TH / 64;
  int64_t blocks = ((n_rows + 31) / 32 + kWavesTh - 1) / kWavesTh;
  const int64_t cap = (int64_t)device_cus() * (per_cu < 1 ? 1 : per_cu);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  // the split scan's staged stores: a whole [N, L] matrix on a 16-byte boundary, and 8 x 32 TILES floats of LDS per wave to spare
  const size_t stage_bytes = (size_t)kWavesTh * TILES * 256 * sizeof(float);
  const bool stage = SPLIT && SAF_QS_STAGE && out && (out_stride <= 0 || out_stride == L) && out_col0 == 0 && ((uintptr_t)out & 15) == 0 &&
                     shmem + stage_bytes <= 159 * 1024 && (160 * 1024) / (shmem + stage_bytes + 256) >= (size_t)(per_cu < 1 ? 1 : per_cu);
  if (stage) shmem += stage_bytes;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)shmem);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(TH), shmem, s, feats, n_rows, fstride, D, text, L,
                     tstride, scale, normalize, wts, out, out_last, out_stride > 0 ? out_stride : (int64_t)L, out_col0, stage ? 1 : 0);
  return check_launch(SPLIT == 2 ? "query_split16_kernel" : SPLIT ? "query_split_kernel" : "query_mfma_kernel");
}

// SAF_Q_SPLIT (read per call): 0 = the fp32 MFMA scan for every shape (development / A-B); default: the split scan where it applies
inline bool split_wanted(int D) {
  const char* e = getenv("SAF_Q_SPLIT");
  return D % 16 == 0 && !(e && atoi(e) == 0);
}

template <int EPI, int FT, int TILES>
int launch_mfma_t(const void* feats, int64_t n_rows, int64_t fstride, int D, const float* text, int L, int64_t tstride,
                  float scale, int normalize, const float* wts, float* out, float* out_last, hipStream_t s, int64_t out_stride, int out_col0) {
  const bool split = split_wanted(D);
  // (the split scan's label rows take the bytes of the fp32 rows -- two fp16 pieces per value -- plus a scale per label)
  const size_t shmem = (size_t)TILES * 32 * (D + 4) * sizeof(float) + (split ? (size_t)TILES * 32 * sizeof(float) : 0);
  const int per_cu = (int)((160 * 1024) / (shmem + 256)) > 2 ? 2 : (int)((160 * 1024) / (shmem + 256));
  // SAF_Q_THREADS (read per call; development): 256 or 512 threads per workgroup whatever the tiles leave room for
  const char* th_env = getenv("SAF_Q_THREADS");
  const bool wide = th_env ? atoi(th_env) == 512 : per_cu <= 1;
#define SAF_Q_GO(TH, SP) launch_mfma_th<EPI, FT, TILES, TH, SP>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out, out_last, per_cu, shmem, s, out_stride, out_col0)
  // a 16-bit volume whose rows lie on 16-byte boundaries: one load per k-step in the matrix instruction's layout, one piece per feature
  const char* e16 = getenv("SAF_Q_SPLIT16");  // (0: through the fp32 form, development / A-B)
  if (split && FT != SAF_F32 && fstride % 8 == 0 && !(e16 && atoi(e16) == 0)) return wide ? SAF_Q_GO(512, 2) : SAF_Q_GO(256, 2);
  if (split) return wide ? SAF_Q_GO(512, 1) : SAF_Q_GO(256, 1);
  return wide ? SAF_Q_GO(512, 0) : SAF_Q_GO(256, 0);
#undef SAF_Q_GO
}

// true if the MFMA scan can take this shape
inline bool mfma_ok(int ft, int64_t fstride, int D, int L, const void* feats) {
  const int esz = ft == SAF_F32 ? 4 : 2;
  if (D % 8 != 0 || L > 64) return false;  // (L > 64 with an [N, L] output: launch_mfma_blocks, 64 labels at a time)
  if (((uintptr_t)feats & 15) || (fstride * esz) % (4 * esz) != 0 || (fstride % 4) != 0) return false;
  const size_t shmem = (size_t)(L > 32 ? 2 : 1) * 32 * (D + 5) * sizeof(float);  // (+ the split scan's scale per label)
  return shmem <= 150 * 1024;
}

template <int EPI, int FT>
int launch_mfma_f(const void* feats, int64_t n_rows, int64_t fstride, int D, const float* text, int L, int64_t tstride,
                  float scale, int normalize, const float* wts, float* out, float* out_last, hipStream_t s, int64_t out_stride, int out_col0) {
  return L > 32 ? launch_mfma_t<EPI, FT, 2>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out,
                                            out_last, s, out_stride, out_col0)
                : launch_mfma_t<EPI, FT, 1>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out,
                                            out_last, s, out_stride, out_col0);
}

template <int EPI>
int launch_mfma(int ft, const void* feats, int64_t n_rows, int64_t fstride, int D, const float* text, int L,
                int64_t tstride, float scale, int normalize, const float* wts, float* out, float* out_last,
                hipStream_t s, int64_t out_stride = 0, int out_col0 = 0) {
  switch (ft) {
    case SAF_BF16:
      return launch_mfma_f<EPI, SAF_BF16>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out, out_last, s, out_stride, out_col0);
    case SAF_F16:
      return launch_mfma_f<EPI, SAF_F16>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out, out_last, s, out_stride, out_col0);
    default:
      return launch_mfma_f<EPI, SAF_F32>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out, out_last, s, out_stride, out_col0);
  }
}

// More than 64 labels (round 6; the reference's control set grows by one label per new query text, clip_seem_fusion.py:496-505):
// the MFMA scan once per block of 64 labels, each writing its block of the row's scaled (surgery: weighted) scores, then one pass
// over the [N, L] scores for what needs the whole row -- softmax's maximum and denominator, surgery's mean.  (Rounds 1-5 fell
// back to the one-wave-per-row kernel.)
template <int EPI>
__global__ __launch_bounds__(256) void finish_rows_kernel(float* __restrict__ out, int64_t n_rows, int L, float* __restrict__ out_last) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < n_rows; r += n_waves) {
    float* row = out + r * L;
    if (EPI == SAF_Q_SOFTMAX) {
      float mx = -INFINITY;
      for (int c = lane; c < L; c += 64) mx = fmaxf(mx, row[c]);
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float sum = 0.0f;
      for (int c = lane; c < L; c += 64) sum += expf(row[c] - mx);
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      for (int c = lane; c < L; c += 64) {
        const float v = expf(row[c] - mx) / sum;
        row[c] = v;
        if (out_last && c == L - 1) out_last[r] = v;
      }
    } else {  // surgery: minus the row's mean over the labels
      float sum = 0.0f;
      for (int c = lane; c < L; c += 64) sum += row[c];
      for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float mean = sum / (float)L;
      for (int c = lane; c < L; c += 64) {
        const float v = row[c] - mean;
        row[c] = v;
        if (out_last && c == L - 1) out_last[r] = v;
      }
    }
  }
}

template <int EPI>
int launch_mfma_blocks(int ft, const void* feats, int64_t n_rows, int64_t fstride, int D, const float* text, int L, int64_t tstride,
                       float scale, int normalize, const float* wts, float* out, float* out_last, hipStream_t s, int bw) {
  // bw: labels per block -- 64, or 32 where two 32-label tiles of this width do not fit the LDS (feat_dim 768 / 1024: round 6)
  for (int c0 = 0; c0 < L; c0 += bw) {
    const int lb = L - c0 < bw ? L - c0 : bw;
    int rc = launch_mfma<SAF_Q_SCORES>(ft, feats, n_rows, fstride, D, text + (int64_t)c0 * tstride, lb, tstride, scale, normalize,
                                       wts ? wts + c0 : nullptr, out, nullptr, s, (int64_t)L, c0);
    if (rc) return rc;
  }
  if (EPI == SAF_Q_SCORES) {
    if (out_last) return fail(SAF_E_UNSUPPORTED, "query scan: out_last of a raw-score scan over more than 64 labels");
    return SAF_OK;
  }
  int64_t blocks = (n_rows + 3) / 4;
  const int64_t cap = (int64_t)device_cus() * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(finish_rows_kernel<EPI>, dim3((unsigned)blocks), dim3(256), 0, s, out, n_rows, L, out_last);
  return check_launch("finish_rows_kernel");
}

template <int EPI>
int launch(int ft, const void* feats, int64_t n_rows, int64_t fstride, int D, const float* text, int L, int64_t tstride,
           float scale, int normalize, float* wts, float* out, float* out_last, hipStream_t s) {
  switch (ft) {
    case SAF_BF16:
      return launch_t<EPI, SAF_BF16>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out, out_last, s);
    case SAF_F16:
      return launch_t<EPI, SAF_F16>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out, out_last, s);
    default:
      return launch_t<EPI, SAF_F32>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out, out_last, s);
  }
}

}  // namespace
}  // namespace saf

using namespace saf;

extern "C" {

size_t saf_query_workspace_bytes(int32_t n_text, int32_t epilogue) {
  return epilogue == SAF_Q_SURGERY && n_text > 0 ? (((size_t)n_text * sizeof(float) + 255) & ~(size_t)255) : 0;
}

int saf_query_scan(const void* feats, int32_t feat_dtype, int64_t n_rows, int64_t feat_stride, int32_t feat_dim,
                   const float* text, int32_t n_text, int64_t text_stride, int32_t epilogue, float scale,
                   int32_t normalize, float* out, float* out_last, void* workspace, size_t workspace_bytes,
                   void* stream) {
  if (feat_dtype != SAF_F32 && feat_dtype != SAF_BF16 && feat_dtype != SAF_F16)
    return fail(SAF_E_UNSUPPORTED, "query scan: unknown feature dtype %d", feat_dtype);
  if (!feats || !text || n_rows < 0 || feat_dim <= 0 || n_text <= 0 || feat_stride < feat_dim || text_stride < feat_dim)
    return fail(SAF_E_INVALID, "query scan: bad arguments");
  if (!out && !out_last) return fail(SAF_E_INVALID, "query scan: no output buffer");
  if (n_rows == 0) return SAF_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const void* f = feats;
  const int ft = feat_dtype;
  const bool mfma = mfma_ok(ft, feat_stride, feat_dim, n_text, feats);
  // more than 64 labels: block by block on the MFMA scan where it takes a 64-label block of this shape and the caller wants the
  // whole [N, L] matrix (the blocks' scores need somewhere to meet)
  // ... and where feat_dim is too wide for two label tiles in LDS (768, 1024 channels: OpenCLIP's larger towers) 32 at a time -- one
  // pass over the volume per block still beats the one-wave-per-row kernel by far
  const int bw = mfma || out == nullptr ? 0
                 : (n_text > 64 && mfma_ok(ft, feat_stride, feat_dim, 64, feats)) ? 64
                 : (n_text > 32 && mfma_ok(ft, feat_stride, feat_dim, 32, feats)) ? 32 : 0;
  const bool blocks = bw != 0;
  switch (epilogue) {
    case SAF_Q_SCORES:
      if (blocks)
        return launch_mfma_blocks<SAF_Q_SCORES>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale, normalize,
                                                nullptr, out, out_last, s, bw);
      if (mfma)
        return launch_mfma<SAF_Q_SCORES>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale,
                                         normalize, nullptr, out, out_last, s);
      return launch<SAF_Q_SCORES>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale, normalize,
                                  nullptr, out, out_last, s);
    case SAF_Q_SOFTMAX:
      if (blocks)
        return launch_mfma_blocks<SAF_Q_SOFTMAX>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale, normalize,
                                                 nullptr, out, out_last, s, bw);
      if (mfma)
        return launch_mfma<SAF_Q_SOFTMAX>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale,
                                          normalize, nullptr, out, out_last, s);
      return launch<SAF_Q_SOFTMAX>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale, normalize,
                                   nullptr, out, out_last, s);
    case SAF_Q_SURGERY: {
      if (!workspace || workspace_bytes < saf_query_workspace_bytes(n_text, epilogue))
        return fail(SAF_E_WORKSPACE, "query scan: surgery needs %zu bytes of workspace",
                    saf_query_workspace_bytes(n_text, epilogue));
      float* wts = static_cast<float*>(workspace);
      int rc = launch<EPI_WEIGHTS>(ft, f, 1, feat_stride, feat_dim, text, n_text, text_stride, 1.0f, normalize, wts,
                                   nullptr, nullptr, s);
      if (rc) return rc;
      if (blocks)
        return launch_mfma_blocks<SAF_Q_SURGERY>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, 1.0f, normalize,
                                                 wts, out, out_last, s, bw);
      if (mfma)
        return launch_mfma<SAF_Q_SURGERY>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, 1.0f,
                                          normalize, wts, out, out_last, s);
      return launch<SAF_Q_SURGERY>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, 1.0f, normalize, wts,
                                   out, out_last, s);
    }
    default:
      return fail(SAF_E_INVALID, "query scan: unknown epilogue %d", epilogue);
  }
}

}  // extern "C"
