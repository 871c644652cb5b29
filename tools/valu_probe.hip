// valu_probe.hip -- development probe: issue cost of single instructions on gfx950, one wave per SIMD, measured with
// s_memtime around 64 copies of the instruction (dependent chain / independent), 2000 repetitions.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float seed) {
  __shared__ int lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0;
  __syncthreads();
  float a = seed + lane, b = seed * 0.5f, c = 1.0f, d = 2.0f, e = 3.f, f = 4.f, g = 5.f, h = 6.f;
  int addr = (threadIdx.x & 255) * 4, iv = lane;
  unsigned long long lv = (unsigned long long)lane;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u qv = {(unsigned)lane, 1u, 2u, 3u};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 2000; ++it) {
    if (MODE == 0) asm volatile(REP64("v_fma_f32 %0, %0, %1, %0\n") : "+v"(a) : "v"(b));                      // dependent fma
    if (MODE == 1) asm volatile(REP16("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n") : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(f));  // 4 independent chains
    if (MODE == 6) asm volatile(REP64("v_readlane_b32 s20, %0, 3\n v_add_f32 %1, s20, %1\n") : : "v"(a), "v"(c) : "s20");
    if (MODE == 7) asm volatile(REP64("v_cvt_rpi_i32_f32 %0, %1\n") : "=v"(iv) : "v"(a));
    if (MODE == 8) asm volatile(REP64("ds_add_u32 %0, %1\n") : : "v"(addr), "v"(iv) : "memory");
    if (MODE == 10) asm volatile(REP16("ds_add_u64 %0, %1\n ds_add_u64 %0, %1 offset:2048\n ds_add_u64 %0, %1 offset:4096\n ds_add_u64 %0, %1 offset:6144\n") : : "v"(addr * 2), "v"(lv) : "memory");
    if (MODE == 11) asm volatile(REP16("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %1 offset:2048\n ds_write_b32 %0, %1 offset:3072\n") : : "v"(addr), "v"(iv) : "memory");
    if (MODE == 12) asm volatile(REP16("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:4096\n ds_write_b64 %0, %1 offset:6144\n") : : "v"(addr * 2), "v"(lv) : "memory");
    if (MODE == 13) asm volatile(REP16("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:8192\n ds_write_b128 %0, %1 offset:12288\n") : : "v"(addr * 4), "v"(qv) : "memory");
    if (MODE == 14) asm volatile(REP16("ds_read_b128 v[40:43], %0\n ds_read_b128 v[44:47], %0 offset:4096\n ds_read_b128 v[48:51], %0 offset:8192\n ds_read_b128 v[52:55], %0 offset:12288\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr * 4) : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    if (MODE == 15) asm volatile(REP16("ds_read_b64 v[40:41], %0\n ds_read_b64 v[44:45], %0 offset:2048\n ds_read_b64 v[48:49], %0 offset:4096\n ds_read_b64 v[52:53], %0 offset:6144\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr * 2) : "memory", "v40", "v41", "v44", "v45", "v48", "v49", "v52", "v53");
    if (MODE == 16) asm volatile(REP16("ds_read_b32 v40, %0\n ds_read_b32 v44, %0 offset:1024\n ds_read_b32 v48, %0 offset:2048\n ds_read_b32 v52, %0 offset:3072\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(addr) : "memory", "v40", "v44", "v48", "v52");
    if (MODE == 9) asm volatile(REP16("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:1024\n ds_add_u32 %0, %1 offset:2048\n ds_add_u32 %0, %1 offset:3072\n") : : "v"(addr), "v"(iv) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (a + c + d + e + (float)iv == 12345.678f) out[0] = 1;
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v2f_t __attribute__((ext_vector_type(2)));
#define HITBODY "s_add_i32 s22, s22, 1\n s_and_b32 s22, s22, 31\n v_readlane_b32 s20, %0, s22\n v_readlane_b32 s24, %1, s22\n v_readlane_b32 s26, %2, s22\n v_readlane_b32 s28, %3, s22\n v_readlane_b32 s23, %4, s22\n v_pk_mul_f32 %5, %7, s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 %6, %8, s[20:21] op_sel_hi:[1,0]\n v_pk_fma_f32 %5, %7, s[24:25], %5 op_sel_hi:[1,0,1]\n v_lshl_add_u32 %9, s23, 2, %10\n v_pk_fma_f32 %5, %8, s[26:27], %5 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %6, %7, s[24:25], %6 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %7, s[28:29], %5 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %6, %8, s[26:27], %6 op_sel_hi:[1,0,1]\n v_cvt_rpi_i32_f32 %11, %0\n ds_add_u32 %9, %11\n v_pk_fma_f32 %6, %7, s[28:29], %6 op_sel_hi:[1,0,1]\n v_cvt_rpi_i32_f32 %11, %1\n ds_add_u32 %9, %11 offset:256\n ds_add_u32 %9, %11 offset:512\n ds_add_u32 %9, %11 offset:768\n v_add_u32 %10, -4, %10\n"

template <int MODE>
__global__ __launch_bounds__(256) void khit(unsigned long long* out, float seed) {
  __shared__ int lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0;
  __syncthreads();
  float w0 = seed + lane, w1 = seed * 2 + lane, w2 = seed * 3 + lane, w3 = seed * 4 + lane;
  int rb = ((lane * 5) & 15) * 256 + lane;
  v2f_t acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f}, ta = {seed, seed * 0.5f}, tb = {seed * 0.25f, 1.f};
  int idx = 0, base = lane * 4 + 64, q = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 2000; ++it) {
    base = lane * 4 + 64;  // 16 hits take it down to lane * 4: every address stays inside the 32 KB
    asm volatile("s_mov_b32 s22, 0\n" REP16(HITBODY)
                 : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(rb), "+v"(acc0), "+v"(acc1), "+v"(ta), "+v"(tb), "+v"(idx), "+v"(base), "+v"(q)
                 :
                 : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc", "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
__global__ __launch_bounds__(256) void kpk(unsigned long long* out, float seed) {
  const int lane = threadIdx.x & 63;
  v2f a = {seed + lane, seed}, b = {0.5f, 0.25f}, c = {1.f, 2.f}, d = {3.f, 4.f}, e = {5.f, 6.f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 2000; ++it) {
    if (MODE == 0) asm volatile(REP64("v_pk_fma_f32 %0, %0, %1, %0\n") : "+v"(a) : "v"(b));                    // dependent packed fma
    if (MODE == 1) asm volatile(REP16("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n") : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(b));
    if (MODE == 2) asm volatile("s_mov_b32 s20, 0x3e400000\n s_mov_b32 s21, 0x3e400000\n" REP16("v_pk_fma_f32 %0, %4, s[20:21], %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %4, s[20:21], %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %4, s[20:21], %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %4, s[20:21], %3 op_sel_hi:[1,0,1]\n") : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b) : "s20", "s21");
    if (MODE == 3) asm volatile(REP16("v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %4, %5\n v_pk_mul_f32 %2, %4, %5\n v_pk_mul_f32 %3, %4, %5\n") : "=v"(a), "=v"(c), "=v"(d), "=v"(e) : "v"(b), "v"(b));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (a.x + c.x + d.x + e.x == 12345.678f) out[0] = 1;
}


// a loop of N copies of one v_fma per iteration: what a taken backward branch costs
template <int N>
__global__ __launch_bounds__(1024) void kloop(unsigned long long* out, float seed, int iters) {
  const int lane = threadIdx.x & 63;
  float a = seed + lane, b = seed;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (N == 1) asm volatile("v_fma_f32 %0, %0, %1, %0\n" : "+v"(a) : "v"(b));
    if (N == 4) asm volatile(REP4("v_fma_f32 %0, %0, %1, %0\n") : "+v"(a) : "v"(b));
    if (N == 16) asm volatile(REP16("v_fma_f32 %0, %0, %1, %0\n") : "+v"(a) : "v"(b));
    if (N == 32) asm volatile(REP16("v_fma_f32 %0, %0, %1, %0\n") REP16("v_fma_f32 %0, %0, %1, %0\n") : "+v"(a) : "v"(b));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (a == 12345.678f) out[0] = 1;
}

int main() {
  unsigned long long* out; CK(hipMalloc(&out, 1 << 16));
  unsigned long long h[4];
  auto report = [&](const char* name, int n_instr) {
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    CK(hipMemset(out, 0, sizeof(h)));
    printf("%-64s %7.2f counter ticks per instruction   (raw %llu %llu %llu %llu)\n", name, (double)h[0] / (2000.0 * n_instr), h[0], h[1], h[2], h[3]);
  };
  CK(hipMemset(out, 0, sizeof(h)));
  // one workgroup of 4 waves: one wave per SIMD of one CU
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("v_fma_f32, dependent chain", 64);
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("v_fma_f32, 4 independent chains", 64);
  hipLaunchKernelGGL(kpk<0>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("v_pk_fma_f32, dependent chain", 64);
  hipLaunchKernelGGL(kpk<1>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("v_pk_fma_f32, 4 independent chains", 64);
  hipLaunchKernelGGL(kpk<2>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("v_pk_fma_f32 with an SGPR pair (op_sel_hi broadcast), 4 chains", 64);
  hipLaunchKernelGGL(kpk<3>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("v_pk_mul_f32, independent", 64);
  hipLaunchKernelGGL(k<6>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("v_readlane_b32 -> SGPR -> v_add_f32 (pairs; per instruction)", 128);
  hipLaunchKernelGGL(k<7>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("v_cvt_rpi_i32_f32, independent", 64);
  hipLaunchKernelGGL(k<8>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("ds_add_u32, same address", 64);
  hipLaunchKernelGGL(k<9>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("ds_add_u32, four addresses in turn", 64);
  hipLaunchKernelGGL(k<10>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("ds_add_u64, four addresses in turn", 64);
  hipLaunchKernelGGL(k<11>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("ds_write_b32", 64);
  hipLaunchKernelGGL(k<12>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("ds_write_b64", 64);
  hipLaunchKernelGGL(k<13>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("ds_write_b128", 64);
  hipLaunchKernelGGL(k<16>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("ds_read_b32", 64);
  hipLaunchKernelGGL(k<15>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("ds_read_b64", 64);
  hipLaunchKernelGGL(k<14>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("ds_read_b128 (4 waves on the CU)", 64);
  hipLaunchKernelGGL(k<14>, dim3(2), dim3(256), 0, 0, out, 1.5f); report("ds_read_b128 (two workgroups of 4 waves: same CU or not)", 64);
  hipLaunchKernelGGL(k<9>, dim3(1), dim3(64), 0, 0, out, 1.5f); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost)); printf("%-64s %7.2f counter ticks per instruction\n", "ds_add_u32, ONE wave on the CU", (double)h[0] / (2000.0 * 64));
  hipLaunchKernelGGL(k<10>, dim3(1), dim3(64), 0, 0, out, 1.5f); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost)); printf("%-64s %7.2f counter ticks per instruction\n", "ds_add_u64, ONE wave on the CU", (double)h[0] / (2000.0 * 64));
  hipLaunchKernelGGL(khit<0>, dim3(1), dim3(256), 0, 0, out, 1.5f); report("the walk's per-hit body, straight line (23 instructions; per HIT)", 16);
  for (int wg : {1, 2, 4}) {  // waves per SIMD
    printf("loops (%d wave(s) per SIMD): ticks per ITERATION\n", wg);
    hipLaunchKernelGGL(kloop<1>, dim3(1), dim3(256 * wg > 1024 ? 1024 : 256 * wg), 0, 0, out, 1.5f, 32000); report("  1 v_fma per iteration (x 16 for the table's divisor)", 16);
    hipLaunchKernelGGL(kloop<4>, dim3(1), dim3(256 * wg), 0, 0, out, 1.5f, 32000); report("  4 v_fma per iteration", 16);
    hipLaunchKernelGGL(kloop<16>, dim3(1), dim3(256 * wg), 0, 0, out, 1.5f, 32000); report("  16 v_fma per iteration", 16);
    hipLaunchKernelGGL(kloop<32>, dim3(1), dim3(256 * wg), 0, 0, out, 1.5f, 32000); report("  32 v_fma per iteration", 16);
  }
  // the counter's rate: s_memtime ticks per microsecond
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, 1.5f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  printf("counter: %.1f ticks per microsecond of kernel time (launch overhead included in the time)\n", (double)h[1] / (ms * 1e3));
  return 0;
}
