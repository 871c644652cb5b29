// lds_atomic_bench.hip -- development probe: throughput of LDS accumulate forms on gfx950 (per CU, all CUs busy):
// ds_add_f32 (no return), ds_add_u32, ds_read_b32 + v_add + ds_write_b32, ds_read_b128 + 4 adds + ds_write_b128.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_bench.hip -o gpurun_out/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kRows = 128;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const int* rows, int iters) {
  __shared__ float acc[kRows * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < kRows * 64; i += 256) acc[i] = 0.f;
  __syncthreads();
  float s = 1.0f + lane;
  int r = rows[tid >> 6];
  for (int it = 0; it < iters; ++it) {
    r = (r * 5 + 1) & (kRows - 1);  // uniform per wave
    const int ru = __builtin_amdgcn_readfirstlane(r);
    if (MODE == 0) {
      __hip_atomic_fetch_add(&acc[ru * 64 + lane], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 1) {
      __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(&acc[ru * 64 + lane]), (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 2) {
      volatile float* p = &acc[ru * 64 + lane];
      *p = *p + s;
    } else if (MODE == 3) {  // 16 lanes x float4 = one 64-channel row per quarter wave: 4 rows per instruction
      volatile float4* p = reinterpret_cast<volatile float4*>(&acc[((ru + (lane >> 4)) & (kRows - 1)) * 64 + (lane & 15) * 4]);
      float4 v;
      v.x = p->x; v.y = p->y; v.z = p->z; v.w = p->w;
      p->x = v.x + s; p->y = v.y + s; p->z = v.z + s; p->w = v.w + s;
    } else if (MODE == 4) {  // returning float atomic
      s += 1e-9f * __hip_atomic_fetch_add(&acc[ru * 64 + lane], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = acc[5] + s;
}

int main() {
  float* out; int* rows;
  CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&rows, 64));
  int h[4] = {1, 7, 19, 33};
  CK(hipMemcpy(rows, h, 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  const char* names[5] = {"ds_add_f32 (no return)", "ds_add_u32 (no return)", "ds_read_b32 + add + ds_write_b32", "ds_read_b128 + 4 add + ds_write_b128 (4 rows)", "ds_add_rtn_f32"};
  for (int wgs : {256, 512, 1024}) {
    for (int m = 0; m < 5; ++m) {
      auto launch = [&] {
        switch (m) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, out, rows, iters); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, out, rows, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, out, rows, iters); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, out, rows, iters); break;
          default: hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(256), 0, 0, out, rows, iters); break;
        }
      };
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double per_cu = (double)wgs / 256.0 * 4.0 * iters;  // wave-instructions per CU
      printf("wgs %4d  %-48s %8.3f ms  %7.1f ns per wave-op per CU  (%.1f cycles at 2.4 GHz)\n", wgs, names[m], ms, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4);
    }
  }
  return 0;
}
