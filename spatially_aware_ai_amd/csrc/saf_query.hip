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

__device__ __forceinline__ float half_max(float x) {  // over the 32 lanes of this half
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
  return x;
}
__device__ __forceinline__ float half_sum(float x) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}



// TH threads per workgroup: 256, or 512 where the text tiles leave room for only ONE workgroup per CU (two 32-row tiles at D = 512:
// 132 KB) -- four waves would be one per SIMD, each alone with its loads, LDS reads and MFMA chain: the L = 63 surgery scan over
// the 256^3 x 512 fp32 volume 14.8 -> 12.0 ms (tools/q_threads_ab.sh); 1024 threads spill (128 registers) and are slower (12.6).
template <int EPI, int FT, int TILES, int TH>
__global__ __launch_bounds__(TH) void query_mfma_kernel(const void* __restrict__ feats, int64_t n_rows,
                                                                int64_t fstride, int D, const float* __restrict__ text,
                                                                int L, int64_t tstride, float scale, int normalize,
                                                                const float* __restrict__ wts, float* __restrict__ out,
                                                                float* __restrict__ out_last, int64_t out_stride, int out_col0) {
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
    // ---- epilogue on the accumulator layout: this lane holds column n = m (+32 t) of 16 rows
    ss += __shfl_xor(ss, 32);  // both halves of row m
    // clip_feat /= norm ; nan_to_num: an all-zero row gives zeros       clip_seem_fusion.py:508-511
    // SAF_NORM_L2_CLAMP: norm.clamp_min(0.1)                            eval_scannet_segmentation.py:549-551
    const float inv = normalize == SAF_NORM_L2_CLAMP ? 1.0f / fmaxf(sqrtf(ss), 0.1f)
                                                     : (normalize ? (ss > 0.0f ? 1.0f / sqrtf(ss) : 0.0f) : 1.0f);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int mi = (i & 3) + 8 * (i >> 2) + 4 * h;  // row of accumulator register i in this half
      const float inv_i = __shfl(inv, mi);
      const int64_t r = tile * 32 + mi;
      float val[TILES];
      bool ok[TILES];
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        ok[t] = (m + 32 * t) < L;
        val[t] = acc[t][i] * inv_i;  // cosine score S[r][n]
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
          val[t] = ok[t] ? val[t] * wts[m + 32 * t] : 0.0f;
          part += val[t];
        }
        const float mean = half_sum(part) / (float)L;
#pragma unroll
        for (int t = 0; t < TILES; ++t) val[t] -= mean;
      } else {
#pragma unroll
        for (int t = 0; t < TILES; ++t) val[t] = wts && ok[t] ? val[t] * scale * wts[m + 32 * t] : val[t] * scale;  // (wts: a block of a wide surgery scan)
      }
      if (r < n_rows) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          if (ok[t]) {
            if (out) out[r * out_stride + out_col0 + m + 32 * t] = val[t];
            if (out_last && m + 32 * t == L - 1) out_last[r] = val[t];
          }
        }
      }
    }
  }
}

template <int EPI, int FT, int TILES, int TH>
int launch_mfma_th(const void* feats, int64_t n_rows, int64_t fstride, int D, const float* text, int L, int64_t tstride,
                   float scale, int normalize, const float* wts, float* out, float* out_last, int per_cu, size_t shmem, hipStream_t s,
                   int64_t out_stride, int out_col0) {
  auto fn = query_mfma_kernel<EPI, FT, TILES, TH>;
  if (shmem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)shmem);
    if (e != hipSuccess) return fail(SAF_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  constexpr int kWavesTh = TH / 64;
  int64_t blocks = ((n_rows + 31) / 32 + kWavesTh - 1) / kWavesTh;
  const int64_t cap = (int64_t)device_cus() * (per_cu < 1 ? 1 : per_cu);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(TH), shmem, s, feats, n_rows, fstride, D, text, L,
                     tstride, scale, normalize, wts, out, out_last, out_stride > 0 ? out_stride : (int64_t)L, out_col0);
  return check_launch("query_mfma_kernel");
}

template <int EPI, int FT, int TILES>
int launch_mfma_t(const void* feats, int64_t n_rows, int64_t fstride, int D, const float* text, int L, int64_t tstride,
                  float scale, int normalize, const float* wts, float* out, float* out_last, hipStream_t s, int64_t out_stride, int out_col0) {
  const size_t shmem = (size_t)TILES * 32 * (D + 4) * sizeof(float);
  const int per_cu = (int)((160 * 1024) / (shmem + 256)) > 2 ? 2 : (int)((160 * 1024) / (shmem + 256));
  // SAF_Q_THREADS (read per call; development): 256 or 512 threads per workgroup whatever the tiles leave room for
  const char* th_env = getenv("SAF_Q_THREADS");
  const bool wide = th_env ? atoi(th_env) == 512 : per_cu <= 1;
  return wide ? launch_mfma_th<EPI, FT, TILES, 512>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out, out_last, per_cu, shmem, s, out_stride, out_col0)
              : launch_mfma_th<EPI, FT, TILES, 256>(feats, n_rows, fstride, D, text, L, tstride, scale, normalize, wts, out, out_last, per_cu, shmem, s, out_stride, out_col0);
}

// true if the MFMA scan can take this shape
inline bool mfma_ok(int ft, int64_t fstride, int D, int L, const void* feats) {
  const int esz = ft == SAF_F32 ? 4 : 2;
  if (D % 8 != 0 || L > 64) return false;  // (L > 64 with an [N, L] output: launch_mfma_blocks, 64 labels at a time)
  if (((uintptr_t)feats & 15) || (fstride * esz) % (4 * esz) != 0 || (fstride % 4) != 0) return false;
  const size_t shmem = (size_t)(L > 32 ? 2 : 1) * 32 * (D + 4) * sizeof(float);
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
                       float scale, int normalize, const float* wts, float* out, float* out_last, hipStream_t s) {
  for (int c0 = 0; c0 < L; c0 += 64) {
    const int lb = L - c0 < 64 ? L - c0 : 64;
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
  const bool blocks = !mfma && n_text > 64 && out != nullptr && mfma_ok(ft, feat_stride, feat_dim, 64, feats);
  switch (epilogue) {
    case SAF_Q_SCORES:
      if (blocks)
        return launch_mfma_blocks<SAF_Q_SCORES>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale, normalize,
                                                nullptr, out, out_last, s);
      if (mfma)
        return launch_mfma<SAF_Q_SCORES>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale,
                                         normalize, nullptr, out, out_last, s);
      return launch<SAF_Q_SCORES>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale, normalize,
                                  nullptr, out, out_last, s);
    case SAF_Q_SOFTMAX:
      if (blocks)
        return launch_mfma_blocks<SAF_Q_SOFTMAX>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale, normalize,
                                                 nullptr, out, out_last, s);
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
                                                 wts, out, out_last, s);
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
