// Development probe: gfx950's v_cvt_pk_bf16_f32 (what `(__bf16)x` compiles to) against the software round-to-nearest-even
// of saf_common.h, over special values and 2^32 / 61 strided bit patterns.  hipcc --offload-arch=gfx950 -O2 -o t this.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __host__ inline uint32_t sw(uint32_t u) {
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__global__ void k(uint32_t start, uint32_t stride, uint32_t n, unsigned long long* bad, uint32_t* first) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t u = start + i * stride;
  float f = __builtin_bit_cast(float, u);
  __bf16 b = (__bf16)f;
  uint32_t hw = __builtin_bit_cast(unsigned short, b);
  uint32_t s = sw(u) & 0xffffu;
  if (hw != s) {
    unsigned long long c = atomicAdd(bad, 1ull);
    if (c < 8) { first[2 * c] = u; first[2 * c + 1] = hw; }
  }
}
int main() {
  unsigned long long* bad; uint32_t* first;
  (void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 64 * 4); (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 256);
  const uint32_t n = 0xffffffffu / 61u;
  k<<<(n + 255) / 256, 256>>>(0u, 61u, n, bad, first);
  // every bit pattern of NaN / inf neighbourhoods and of the denormals' top end
  k<<<(0x01000000u + 255) / 256, 256>>>(0x7f000000u, 1u, 0x01000000u, bad, first);
  k<<<(0x01000000u + 255) / 256, 256>>>(0xff000000u, 1u, 0x00ffffffu, bad, first);
  k<<<(0x01000000u + 255) / 256, 256>>>(0x00000000u, 1u, 0x01000000u, bad, first);
  unsigned long long hb; uint32_t hf[16];
  (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hf, first, 64, hipMemcpyDeviceToHost);
  printf("mismatches: %llu\n", hb);
  for (int i = 0; i < 8 && i < (int)hb; ++i) printf("  f32 bits %08x: hw %04x sw %04x\n", hf[2 * i], hf[2 * i + 1], sw(hf[2 * i]) & 0xffff);
  return 0;
}
