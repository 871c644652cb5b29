// seg_rmw_bench.hip -- development probe (not part of the library): how fast does HBM read-modify-write go when the
// touched feature rows (2 KiB each, ~60 % of a contiguous range) are visited (A) as whole rows, one wave per row, or
// (B) in 8 passes over 256-byte channel slabs, 16 lanes per row segment?  Decides whether a channel-slab row kernel
// (map taps LDS-resident per slab) can keep the row traffic at the whole-row rate.
// Build: hipcc -O3 --offload-arch=gfx950 tools/seg_rmw_bench.hip -o gpurun_out/seg_rmw_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_nt(const float4* p) {
  const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_nt(float4* p, float4 x) {
  v4f_t v;
  v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
  __builtin_nontemporal_store(v, reinterpret_cast<v4f_t*>(p));
}

// (A) one wave per row, U rows in flight per wave: lane l moves float4 l and 64 + l of the 128-float4 row
template <int U>
__global__ __launch_bounds__(256) void rows_kernel(float4* __restrict__ feat, const uint32_t* __restrict__ rows, uint32_t n) {
  const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, nw = (gridDim.x * 256) >> 6;
  for (uint32_t i = wave * U; i < n; i += nw * U) {
    float4 a[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t r = rows[min(i + u, n - 1)];
      a[u][0] = ld_nt(feat + (size_t)r * 128 + lane);
      a[u][1] = ld_nt(feat + (size_t)r * 128 + 64 + lane);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i + u < n) {
        const uint32_t r = rows[i + u];
        a[u][0].x += 1.0f; a[u][1].y += 1.0f;
        st_nt(feat + (size_t)r * 128 + lane, a[u][0]);
        st_nt(feat + (size_t)r * 128 + 64 + lane, a[u][1]);
      }
    }
  }
}

// (B) slab pass: SEG float4 per row segment (16 -> 256 B, 32 -> 512 B); 64 / SEG rows per wave instruction, U in flight
template <int SEG, int U>
__global__ __launch_bounds__(256) void slab_kernel(float4* __restrict__ feat, const uint32_t* __restrict__ rows, uint32_t n,
                                                   int slab) {
  constexpr int RPW = 64 / SEG;
  const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, nw = (gridDim.x * 256) >> 6;
  const uint32_t sub = lane / SEG, l = lane % SEG;
  for (uint32_t i = wave * U * RPW; i < n; i += nw * U * RPW) {
    float4 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t r = rows[min(i + u * RPW + sub, n - 1)];
      a[u] = ld_nt(feat + (size_t)r * 128 + slab * SEG + l);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i + u * RPW + sub < n) {
        const uint32_t r = rows[i + u * RPW + sub];
        a[u].x += 1.0f;
        st_nt(feat + (size_t)r * 128 + slab * SEG + l, a[u]);
      }
    }
  }
}

int main(int argc, char** argv) {
  const size_t n_rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : (size_t)6 << 20;  // 6 M rows = 12.9 GB
  float4* feat;
  CK(hipMalloc(&feat, n_rows * 2048));
  CK(hipMemset(feat, 0, n_rows * 2048));
  std::vector<uint32_t> h;
  uint64_t s = 88172645463325252ull;
  for (size_t r = 0; r < n_rows; ++r) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    if ((s >> 11) % 100 < 60) h.push_back((uint32_t)r);
  }
  const uint32_t n = (uint32_t)h.size();
  uint32_t* rows;
  CK(hipMalloc(&rows, n * 4));
  CK(hipMemcpy(rows, h.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)n * 4096.0;
  auto time = [&](const char* name, auto launch) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 3; ++it) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %6.2f TB/s (read+write of %u touched rows)\n", name, ms / 3, bytes / (ms / 3 * 1e-3) / 1e12, n);
  };
  for (int grid : {1024, 2048, 4096}) {
    printf("grid %d\n", grid);
    time("whole rows, 4 in flight per wave", [&] { hipLaunchKernelGGL(rows_kernel<4>, dim3(grid), dim3(256), 0, 0, feat, rows, n); });
    time("whole rows, 8 in flight per wave", [&] { hipLaunchKernelGGL(rows_kernel<8>, dim3(grid), dim3(256), 0, 0, feat, rows, n); });
    time("8 slabs of 256 B, 4x4 segments in flight", [&] { for (int sl = 0; sl < 8; ++sl) hipLaunchKernelGGL((slab_kernel<16, 4>), dim3(grid), dim3(256), 0, 0, feat, rows, n, sl); });
    time("8 slabs of 256 B, 8x4 segments in flight", [&] { for (int sl = 0; sl < 8; ++sl) hipLaunchKernelGGL((slab_kernel<16, 8>), dim3(grid), dim3(256), 0, 0, feat, rows, n, sl); });
    time("4 slabs of 512 B, 8x2 segments in flight", [&] { for (int sl = 0; sl < 4; ++sl) hipLaunchKernelGGL((slab_kernel<32, 8>), dim3(grid), dim3(256), 0, 0, feat, rows, n, sl); });
    time("2 slabs of 1 KiB, 8 segments in flight", [&] { for (int sl = 0; sl < 2; ++sl) hipLaunchKernelGGL((slab_kernel<64, 8>), dim3(grid), dim3(256), 0, 0, feat, rows, n, sl); });
  }
  return 0;
}
