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
        const float norm = sqrtf(wave_sum(ss));
        for (int c = lane; c < D; c += 64) {
          float q = row[c] / norm;
          if (q != q) q = 0.f;
          if (__builtin_isinf(q)) q = q > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
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
  switch (epilogue) {
    case SAF_Q_SCORES:
      return launch<SAF_Q_SCORES>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, scale, normalize,
                                  nullptr, out, out_last, s);
    case SAF_Q_SOFTMAX:
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
      return launch<SAF_Q_SURGERY>(ft, f, n_rows, feat_stride, feat_dim, text, n_text, text_stride, 1.0f, normalize, wts,
                                   out, out_last, s);
    }
    default:
      return fail(SAF_E_INVALID, "query scan: unknown epilogue %d", epilogue);
  }
}

}  // extern "C"
