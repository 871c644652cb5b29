// saf_dwconv.hip -- the 7 x 7 depthwise convolution of a ConvNeXt block (the panoptic encoder of BASELINE config 3:
// kMaX-DeepLab's ConvNeXt-L, handy_utils.py:29-161 via detectron2) for channels-last activations.
//
// Why here: the backbones stay PyTorch-ROCm modules, but for `groups == channels` 7 x 7 convolutions on
// channels-last bf16 / fp16 / fp32 tensors MIOpen falls to `naive_conv_*`: 44-48 ms of a 47 ms panoptic forward, 46 % of all
// kernel time of the default bench (profiles/r03/kernel_stats.csv).  The op is tiny -- 17 GFLOP and 36 layers of at most
// 30 MB per 1281 x 960 frame -- and HBM / L2 bound: every activation is read once from L2 per kernel row it is used in.
//
// Layout: x, y [N, H, W, C] (C contiguous), w [7, 7, C] f32 (re-laid once by the host from PyTorch's [C, 1, 7, 7]), bias [C]
// f32 or NULL.  A lane owns 8 consecutive channels (16 bytes of bf16 / fp16; 32 of f32) of kTX consecutive output pixels of
// one row: per kernel row it loads the kTX + 6 input vectors once (a sliding window in registers), 7 weight vectors, and
// issues 7 x kTX x 8 fp32 FMAs.  Zero padding (3 pixels): a tap outside the image is not loaded.
#include "saf_common.h"
#include "saf_host.h"
#include "../../include/saf.h"

namespace saf {
namespace {

constexpr int kTX = 4;  // output pixels per lane (along W)

template <int FT>
struct Vec8 {
  float v[8];
};
template <int FT>
__device__ __forceinline__ Vec8<FT> ld8(const void* base, int64_t vec_index) {
  Vec8<FT> r;
  if (FT == SAF_F32) {
    const float4* p = reinterpret_cast<const float4*>(base) + 2 * vec_index;
    const float4 a = p[0], b = p[1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  } else {
    const uint4 u = reinterpret_cast<const uint4*>(base)[vec_index];
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (FT == SAF_BF16) {
        r.v[2 * j] = bf16_lo(w[j]);
        r.v[2 * j + 1] = bf16_hi(w[j]);
      } else {
        r.v[2 * j] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[j] & 0xffffu));
        r.v[2 * j + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[j] >> 16));
      }
    }
  }
  return r;
}
template <int FT>
__device__ __forceinline__ void st8(void* base, int64_t vec_index, const float (&v)[8]) {
  if (FT == SAF_F32) {
    float4* p = reinterpret_cast<float4*>(base) + 2 * vec_index;
    p[0] = make_float4(v[0], v[1], v[2], v[3]);
    p[1] = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (FT == SAF_BF16) {
        w[j] = pack_bf16(v[2 * j], v[2 * j + 1]);
      } else {
        w[j] = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)v[2 * j]) |
               ((uint32_t)__builtin_bit_cast(unsigned short, (_Float16)v[2 * j + 1]) << 16);
      }
    }
    reinterpret_cast<uint4*>(base)[vec_index] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// grid: x = channel vectors x strips of a row (flattened), y = rows, z = images
template <int FT>
__global__ __launch_bounds__(256) void dwconv7_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, void* __restrict__ y, int H, int W, int C8,
                                                      int strips) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= C8 * strips) return;
  const int cv = t % C8, sx = t / C8;  // (neighbouring lanes: neighbouring channel vectors of the same pixels -- coalesced)
  const int oy = blockIdx.y, n = blockIdx.z, ox0 = sx * kTX;
  float acc[kTX][8];
#pragma unroll
  for (int i = 0; i < kTX; ++i)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[i][c] = bias ? bias[cv * 8 + c] : 0.0f;
  const int64_t img = (int64_t)n * H * W;
  for (int ky = 0; ky < 7; ++ky) {
    const int iy = oy + ky - 3;
    if (iy < 0 || iy >= H) continue;  // (uniform over the workgroup: one output row)
    Vec8<FT> in[kTX + 6];
#pragma unroll
    for (int j = 0; j < kTX + 6; ++j) {
      const int ix = ox0 + j - 3;
      if (ix >= 0 && ix < W) {
        in[j] = ld8<FT>(x, (img + (int64_t)iy * W + ix) * C8 + cv);
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) in[j].v[c] = 0.0f;
      }
    }
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const Vec8<SAF_F32> wk = ld8<SAF_F32>(w, (int64_t)(ky * 7 + kx) * C8 + cv);
#pragma unroll
      for (int i = 0; i < kTX; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[i][c] = __builtin_fmaf(in[i + kx].v[c], wk.v[c], acc[i][c]);
    }
  }
#pragma unroll
  for (int i = 0; i < kTX; ++i)
    if (ox0 + i < W) st8<FT>(y, (img + (int64_t)oy * W + ox0 + i) * C8 + cv, acc[i]);
}

}  // namespace
}  // namespace saf

using namespace saf;

int saf_dwconv7x7_nhwc(const void* x, const float* w_kkc, const float* bias, void* y, int32_t batch, int32_t height,
                                  int32_t width, int32_t channels, int32_t dtype, void* stream) {
  if (!x || !w_kkc || !y || batch <= 0 || height <= 0 || width <= 0 || channels <= 0)
    return fail(SAF_E_INVALID, "dwconv7x7: bad arguments");
  if (channels % 8 != 0) return fail(SAF_E_UNSUPPORTED, "dwconv7x7: channels must be a multiple of 8 (got %d)", channels);
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_kkc) & 15) return fail(SAF_E_INVALID, "dwconv7x7: buffers must be 16-byte aligned");
  if (height > 65535 || batch > 65535) return fail(SAF_E_UNSUPPORTED, "dwconv7x7: more than 65535 rows or images");
  const int C8 = channels / 8, strips = (width + kTX - 1) / kTX;
  const dim3 grid((unsigned)((C8 * strips + 255) / 256), (unsigned)height, (unsigned)batch);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case SAF_F32: hipLaunchKernelGGL(dwconv7_kernel<SAF_F32>, grid, dim3(256), 0, s, x, w_kkc, bias, y, height, width, C8, strips); break;
    case SAF_BF16: hipLaunchKernelGGL(dwconv7_kernel<SAF_BF16>, grid, dim3(256), 0, s, x, w_kkc, bias, y, height, width, C8, strips); break;
    case SAF_F16: hipLaunchKernelGGL(dwconv7_kernel<SAF_F16>, grid, dim3(256), 0, s, x, w_kkc, bias, y, height, width, C8, strips); break;
    default: return fail(SAF_E_INVALID, "dwconv7x7: bad dtype %d", dtype);
  }
  return check_launch("dwconv7_kernel");
}
