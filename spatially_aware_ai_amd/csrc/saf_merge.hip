// saf_merge.hip -- the device side of the frame-sharded merge's PACKED route (new
// capability, nothing to mirror in the reference).  A real scan touches a thin shell of the grid, so after the all-reduce of
// `weight` (every rank then knows the union of the touched rows) only touched rows travel: the touched rows of part k of a
// piece go to rank k (all_to_all with uneven splits, issued by distributed.py), which adds the world contributions in rank
// order.  Round 5 did the three passes around the collective with PyTorch (nonzero + index_select, sum over a stacked view,
// index_copy_) and one host sync per slab; here:
//   saf_merge_scan_touched   pos[i] = touched rows among [0, i) (one hipCUB exclusive scan over the mask) + the positions at
//                            the plan's part boundaries copied to the host asynchronously -- the split sizes of the collective;
//   saf_merge_pack_rows      touched rows of a row range -> a dense buffer (the send buffer), no index list;
//   saf_merge_add_packed     the world received contributions of this rank's touched rows, added in rank order, straight
//                            into place.
// All three are streaming passes (HBM-bound; 2 KiB rows at D = 512 f32).
#include <hipcub/hipcub.hpp>

#include "saf_host.h"

namespace saf {
namespace {

struct TouchedFlag {  // item i of the scan: 1 if row i is touched; item n (one past the end) is 0
  const int32_t* w;
  int64_t n;
  __host__ __device__ int operator()(int64_t i) const { return i < n && w[i] > 0 ? 1 : 0; }
};
using FlagIter = hipcub::TransformInputIterator<int, TouchedFlag, hipcub::CountingInputIterator<int64_t>>;

__global__ __launch_bounds__(256) void gather_offsets_kernel(const int32_t* __restrict__ pos, const int64_t* __restrict__ bounds,
                                                             int n, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = pos[bounds[i]];
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 is a struct around a union: a native vector stays one 16-byte access)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// A wave takes 64 consecutive rows at a time: lane l reads weight[r0 + l], the ballot names the touched ones, and the wave
// copies them one after the other, all lanes on one row (a row of 2 KiB is two 16-byte accesses per lane).
__global__ __launch_bounds__(256) void pack_rows_kernel(const unsigned char* __restrict__ src, int64_t row_bytes,
                                                        const int32_t* __restrict__ weight, const int32_t* __restrict__ pos,
                                                        int64_t first, int64_t n_rows, unsigned char* __restrict__ packed, int vec) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  const int32_t p0 = pos[first];
  for (int64_t r0 = wave * 64; r0 < n_rows; r0 += n_waves * 64) {
    const int64_t r = first + r0 + lane;
    const bool t = r0 + lane < n_rows && weight[r] > 0;
    const int32_t p = t ? pos[r] - p0 : 0;
    unsigned long long m = __ballot(t);
    while (m) {
      const int l = __builtin_ctzll(m);
      m &= m - 1;
      const unsigned char* s = src + (first + r0 + l) * row_bytes;
      unsigned char* d = packed + (int64_t)__shfl(p, l) * row_bytes;
      if (vec) {
        for (int64_t o = (int64_t)lane * 16; o < row_bytes; o += 64 * 16)
          *reinterpret_cast<u32x4*>(d + o) = *reinterpret_cast<const u32x4*>(s + o);
      } else {
        for (int64_t o = (int64_t)lane * 4; o < row_bytes; o += 64 * 4)
          *reinterpret_cast<uint32_t*>(d + o) = *reinterpret_cast<const uint32_t*>(s + o);
      }
    }
  }
}

// dst[r] = recv[0][p] + recv[1][p] + ... + recv[world - 1][p] (left to right: rank order, the same on every run) for the
// touched rows r of [first, first + n_rows), p = pos[r] - pos[first]; recv is [world][mine] rows.
template <typename V4, typename S>
__global__ __launch_bounds__(256) void add_packed_kernel(unsigned char* __restrict__ dst, int64_t row_bytes,
                                                         const int32_t* __restrict__ weight, const int32_t* __restrict__ pos,
                                                         int64_t first, int64_t n_rows, const unsigned char* __restrict__ recv,
                                                         int64_t mine, int world, int vec) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  const int32_t p0 = pos[first];
  const int64_t part = mine * row_bytes;
  for (int64_t r0 = wave * 64; r0 < n_rows; r0 += n_waves * 64) {
    const int64_t r = first + r0 + lane;
    const bool t = r0 + lane < n_rows && weight[r] > 0;
    const int32_t p = t ? pos[r] - p0 : 0;
    unsigned long long m = __ballot(t);
    while (m) {
      const int l = __builtin_ctzll(m);
      m &= m - 1;
      unsigned char* d = dst + (first + r0 + l) * row_bytes;
      const unsigned char* s = recv + (int64_t)__shfl(p, l) * row_bytes;
      if (vec) {
        for (int64_t o = (int64_t)lane * 16; o < row_bytes; o += 64 * 16) {
          V4 acc = *reinterpret_cast<const V4*>(s + o);
          int k = 1;
          for (; k + 3 < world; k += 4) {  // four contributions in flight; added left to right all the same
            const V4 a = *reinterpret_cast<const V4*>(s + (k + 0) * part + o), b = *reinterpret_cast<const V4*>(s + (k + 1) * part + o);
            const V4 c = *reinterpret_cast<const V4*>(s + (k + 2) * part + o), e = *reinterpret_cast<const V4*>(s + (k + 3) * part + o);
            acc = (((acc + a) + b) + c) + e;
          }
          for (; k < world; ++k) acc = acc + *reinterpret_cast<const V4*>(s + k * part + o);
          *reinterpret_cast<V4*>(d + o) = acc;
        }
      } else {
        for (int64_t o = (int64_t)lane * 4; o < row_bytes; o += 64 * 4) {
          S acc = *reinterpret_cast<const S*>(s + o);
          for (int k = 1; k < world; ++k) acc = acc + *reinterpret_cast<const S*>(s + k * part + o);
          *reinterpret_cast<S*>(d + o) = acc;
        }
      }
    }
  }
}

size_t scan_tmp_bytes(int64_t n_rows) {
  size_t b = 0;
  TouchedFlag f{nullptr, n_rows};
  FlagIter it(hipcub::CountingInputIterator<int64_t>(0), f);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, it, (int32_t*)nullptr, (int)(n_rows + 1));
  return (b + 255) & ~(size_t)255;
}

int grid_for(int64_t n_rows) {
  const int64_t want = (n_rows + 255) / 256;  // 64 rows per wave and step, 4 waves per workgroup
  const int64_t cap = (int64_t)device_cus() * 16;
  return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

}  // namespace
}  // namespace saf

using namespace saf;

extern "C" {

size_t saf_merge_scan_workspace_bytes(int64_t n_rows, int32_t n_bounds) {
  if (n_rows < 0 || n_rows >= (int64_t)1 << 31 || n_bounds < 0) return 0;
  return scan_tmp_bytes(n_rows) + (((size_t)n_bounds * sizeof(int32_t) + 255) & ~(size_t)255);
}

int saf_merge_scan_touched(const int32_t* weight, int64_t n_rows, int32_t* pos, const int64_t* bounds, int32_t n_bounds,
                           int32_t* offs_host, void* workspace, size_t workspace_bytes, void* stream) {
  if (!weight || !pos || n_rows <= 0 || n_rows >= (int64_t)1 << 31 || n_bounds < 0 || (n_bounds > 0 && (!bounds || !offs_host)) || !workspace)
    return fail(SAF_E_INVALID, "merge_scan_touched: bad arguments");
  if (workspace_bytes < saf_merge_scan_workspace_bytes(n_rows, n_bounds)) return fail(SAF_E_WORKSPACE, "merge_scan_touched: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  size_t tb = scan_tmp_bytes(n_rows);
  TouchedFlag f{weight, n_rows};
  FlagIter it(hipcub::CountingInputIterator<int64_t>(0), f);
  if (hipcub::DeviceScan::ExclusiveSum(workspace, tb, it, pos, (int)(n_rows + 1), s) != hipSuccess)
    return fail(SAF_E_HIP, "merge_scan_touched: the scan failed");
  if (n_bounds > 0) {
    int32_t* offs_dev = reinterpret_cast<int32_t*>(static_cast<unsigned char*>(workspace) + scan_tmp_bytes(n_rows));
    hipLaunchKernelGGL(gather_offsets_kernel, dim3((n_bounds + 255) / 256), dim3(256), 0, s, pos, bounds, (int)n_bounds, offs_dev);
    int r = check_launch("gather_offsets_kernel");
    if (r) return r;
    // (asynchronous when offs_host is pinned: the caller synchronises once, before it reads the split sizes)
    if (hipMemcpyAsync(offs_host, offs_dev, (size_t)n_bounds * sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess)
      return fail(SAF_E_HIP, "merge_scan_touched: copy of the offsets failed");
  }
  return SAF_OK;
}

int saf_merge_pack_rows(const void* src, int64_t row_bytes, const int32_t* weight, const int32_t* pos, int64_t first,
                        int64_t n_rows, void* packed, void* stream) {
  if (!src || !weight || !pos || !packed || row_bytes <= 0 || row_bytes % 4 != 0 || first < 0 || n_rows < 0)
    return fail(SAF_E_INVALID, "merge_pack_rows: bad arguments");
  if (n_rows == 0) return SAF_OK;
  const int vec = row_bytes % 16 == 0 && (((uintptr_t)src | (uintptr_t)packed) & 15) == 0;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(grid_for(n_rows)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned char*>(src), row_bytes, weight, pos, first, n_rows, static_cast<unsigned char*>(packed), vec);
  return check_launch("pack_rows_kernel");
}

int saf_merge_add_packed(void* dst, int64_t row_bytes, int32_t is_float, const int32_t* weight, const int32_t* pos, int64_t first,
                         int64_t n_rows, const void* recv, int64_t mine, int32_t world, void* stream) {
  if (!dst || !weight || !pos || row_bytes <= 0 || row_bytes % 4 != 0 || first < 0 || n_rows < 0 || mine < 0 || world < 1 || (mine > 0 && !recv))
    return fail(SAF_E_INVALID, "merge_add_packed: bad arguments");
  if (n_rows == 0 || mine == 0) return SAF_OK;
  const int vec = row_bytes % 16 == 0 && (((uintptr_t)dst | (uintptr_t)recv) & 15) == 0 && (mine * row_bytes) % 16 == 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (is_float)
    hipLaunchKernelGGL((add_packed_kernel<f32x4, float>), dim3(grid_for(n_rows)), dim3(256), 0, s, static_cast<unsigned char*>(dst), row_bytes,
                       weight, pos, first, n_rows, static_cast<const unsigned char*>(recv), mine, (int)world, vec);
  else
    hipLaunchKernelGGL((add_packed_kernel<i32x4, int32_t>), dim3(grid_for(n_rows)), dim3(256), 0, s, static_cast<unsigned char*>(dst), row_bytes,
                       weight, pos, first, n_rows, static_cast<const unsigned char*>(recv), mine, (int)world, vec);
  return check_launch("add_packed_kernel");
}

}  // extern "C"
