// hit_loop_bench.hip -- development probe: what the brick walk's per-hit loop costs on gfx950, piece by piece.
// One "hit" = 5 v_readlane (4 bilinear weights + the row), 8 packed multiply-adds over 4 channels per lane x 4 taps,
// 4 v_cvt_rpi, 4 ds_add_u32 into an LDS accumulator [row][4][64].  Variants leave pieces out.
// Build: hipcc -O3 --offload-arch=gfx950 tools/hit_loop_bench.hip -o /tmp/hit_loop_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int cvt_rpi(float x) { int q; asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(x)); return q; }
__device__ __forceinline__ float rl(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }

// MODE bits: 1 = weights by readlane (else loop-invariant SGPRs), 2 = the multiply-adds, 4 = LDS atomic adds,
//            8 = plain LDS stores instead of atomics, 16 = one ds_add per hit instead of four
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, const float* in, int hits_per_group, int groups) {
  extern __shared__ int acc[];  // 64 rows x 256 words
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 256; i += THREADS) acc[i] = 0;
  __syncthreads();
  float t[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) t[a][c] = in[(a * 4 + c) * 64 + lane];
  const float w0 = in[lane] * 1e3f, w1 = in[64 + lane] * 1e3f, w2 = in[128 + lane] * 1e3f, w3 = in[192 + lane] * 1e3f;
  const int rowbase = ((lane * 7) & 63) * 256 + lane;
  float sink = 0.f;
  for (int g = 0; g < groups; ++g) {
    const int l0 = (g * 5) & 31;
    auto hit = [&](int l, float (&s)[4], int& idx) {
      float a0, a1, a2, a3;
      if (MODE & 1) {
        a0 = rl(w0, l); a1 = rl(w1, l); a2 = rl(w2, l); a3 = rl(w3, l);
        idx = __builtin_amdgcn_readlane(rowbase, l) - l + lane;
      } else {
        a0 = rl(w0, 3); a1 = rl(w1, 3); a2 = rl(w2, 3); a3 = rl(w3, 3);
        idx = __builtin_amdgcn_readlane(rowbase, 5) + ((l & 31) << 8) - 5 + lane;
        idx &= 64 * 256 - 1;
      }
      if (MODE & 2) {
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          v2f s2 = (v2f){t[0][c], t[0][c + 1]} * (v2f){a0, a0};
          s2 = __builtin_elementwise_fma((v2f){t[1][c], t[1][c + 1]}, (v2f){a1, a1}, s2);
          s2 = __builtin_elementwise_fma((v2f){t[2][c], t[2][c + 1]}, (v2f){a2, a2}, s2);
          s2 = __builtin_elementwise_fma((v2f){t[3][c], t[3][c + 1]}, (v2f){a3, a3}, s2);
          s[c] = s2.x; s[c + 1] = s2.y;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = a0 + t[0][c];
      }
    };
    auto add = [&](const float (&s)[4], int idx) {
      if (MODE & 4) {
#pragma unroll
        for (int c = 0; c < ((MODE & 16) ? 1 : 4); ++c)
          __hip_atomic_fetch_add(&acc[idx + c * 64], cvt_rpi(s[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE & 16) sink += s[1] + s[2] + s[3];
      } else if (MODE & 8) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[idx + c * 64] = cvt_rpi(s[c]);
      } else {
        sink += s[0] + s[1] + s[2] + s[3];
      }
    };
    int l = l0;
    if (MODE & 32) {  // two hits per iteration, each with its own registers
      for (; l + 1 < l0 + hits_per_group; l += 2) {
        float sa[4], sb[4];
        int ia, ib;
        hit(l, sa, ia);
        hit(l + 1, sb, ib);
        add(sa, ia);
        add(sb, ib);
      }
    }
    for (; l < l0 + hits_per_group; ++l) {
      float s[4];
      int idx;
      hit(l, s, idx);
      add(s, idx);
    }
  }
  __syncthreads();
  if (lane == 0) out[blockIdx.x * (THREADS / 64) + (tid >> 6)] = sink + (float)acc[tid];
}

template <int MODE, int THREADS>
float run(int wgs, float* out, const float* in, int hpg, int groups) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto fn = k<MODE, THREADS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(fn, dim3(wgs), dim3(THREADS), 65536, 0, out, in, hpg, groups); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(fn, dim3(wgs), dim3(THREADS), 65536, 0, out, in, hpg, groups);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  float *out, *in;
  CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&in, 4096 * 4));
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (float)((i * 37) % 101);
  CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  const int groups = 20000;
  const char* names[] = {"full: readlanes + packed fma + cvt + 4 ds_add_u32", "no readlanes (invariant weights)", "no multiply-adds",
                         "no LDS (register sink)", "plain ds_write_b32 instead of the atomics", "one ds_add per hit", "readlanes only", "full, two hits per iteration (own registers each)"};
  for (int hpg : {2, 16}) {
    for (int cfg = 0; cfg < 3; ++cfg) {  // waves per SIMD: 1 (256 thr x 1 WG), 2 (256 x 2 WG), 2 (512 x 1 WG)
      const int wgs = cfg == 1 ? 512 : 256;
      printf("---- %d hits per group; %s\n", hpg, cfg == 0 ? "1 workgroup of 4 waves per CU (1 wave per SIMD)" : cfg == 1 ? "2 workgroups of 4 waves per CU (2 per SIMD)" : "1 workgroup of 8 waves per CU (2 per SIMD)");
      for (int v = 0; v < 8; ++v) {
        float ms;
#define RUN(M) (cfg == 2 ? run<M, 512>(wgs, out, in, hpg, groups) : run<M, 256>(wgs, out, in, hpg, groups))
        switch (v) {
          case 0: ms = RUN(7); break;
          case 1: ms = RUN(6); break;
          case 2: ms = RUN(5); break;
          case 3: ms = RUN(3); break;
          case 4: ms = RUN(11); break;
          case 5: ms = RUN(23); break;
          case 6: ms = RUN(1); break;
          default: ms = RUN(39); break;
        }
        const double hits = (double)groups * hpg;  // per wave
        printf("  %-52s %8.3f ms  %7.1f cycles per hit and wave (2.4 GHz)\n", names[v], ms, ms * 1e-3 * 2.4e9 / hits);
      }
    }
  }
  return 0;
}
