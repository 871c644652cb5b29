// gather_probe.hip -- what the L2 -> L1 path of an MI355X delivers for the row kernel's OWN access shape
// (VERDICT round 3, item 1a): rows of ROWB bytes (2 KiB = the 512 fp32 channels of one map position) gathered from a
// table that fits the XCDs' L2s, `buffer_load_dwordx4` (64 lanes x 16 B = one 1 KiB piece per wave instruction), NB
// pieces in flight per wave, 8 / 12 / 16 waves per CU, results discarded.  The ceiling quoted in DESIGN.md section 4.6 so
// far was borrowed from MI355X_MICROARCH.md (1152-byte rows, one 256-thread workgroup per CU: 16.8-18.8 TB/s, "lower
// bound"); this measures it for the shape fuse_window_kernel has.
//
//   hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o gpurun_out/gather_probe && gpurun_out/gather_probe
//
// Output: one line per (table bytes, waves per CU, pieces in flight): TB/s chip-wide of gathered bytes.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));         \
      exit(1);                                                                                  \
    }                                                                                           \
  } while (0)

typedef unsigned int v4u __attribute__((vector_size(16)));

// One wave gathers `iters` batches of NB pieces.  A ROW is PPR consecutive pieces (PPR = 2: 2 KiB); a batch is NB / PPR
// random rows (wave-uniform row index from a scalar LCG, as the row kernel's tap rows are wave-uniform).
// MODE 0: buffer loads through a descriptor (the row kernel's taps); MODE 1: global_load_dwordx4; MODE 2: nontemporal global.
template <int NB, int PPR, int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ table, uint32_t n_rows, uint32_t table_bytes,
                                                     int iters, uint32_t seed, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave_id = blockIdx.x * 4u + (threadIdx.x >> 6);
  uint32_t s = seed ^ (wave_id * 2654435761u);
  s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(table), 0, (int)table_bytes, 0x00020000);
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    float4 t[NB];
#pragma unroll
    for (int r = 0; r < NB / PPR; ++r) {
      s = s * 1664525u + 1013904223u;
      const uint32_t row = (uint32_t)(((uint64_t)(s >> 4) * n_rows) >> 28);  // uniform in [0, n_rows)
#pragma unroll
      for (int p = 0; p < PPR; ++p) {
        const uint32_t off = (row * (uint32_t)PPR + (uint32_t)p) * 1024u + (uint32_t)lane * 16u;
        if (MODE == 0) {
          const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
          t[r * PPR + p] = __builtin_bit_cast(float4, v);
        } else if (MODE == 1) {
          t[r * PPR + p] = table[off / 16u];
        } else {
          typedef float v4f __attribute__((vector_size(16)));
          const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(table) + off / 16u);
          t[r * PPR + p] = __builtin_bit_cast(float4, v);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) asm volatile("" ::"v"(t[k].x), "v"(t[k].y), "v"(t[k].z), "v"(t[k].w));
    acc += t[0].x;
  }
  if (acc == 12345.678f) sink[0] = acc;  // never true: keeps the loop alive without a store
}

// MIX: what the row kernel's L1 path really carries -- per batch NB pieces of L2-resident rows (the map taps) and, every
// `every`-th batch, NR pieces of rows from a table far larger than every cache, each read once (the old feature rows: HBM
// latency, nontemporal).  Do the slow requests hold up the fast ones?  `sink` also receives NW pieces of stores per such
// batch when stores != 0 (the rows going back).
template <int NB, int NR>
__global__ __launch_bounds__(256) void mix_kernel(const float4* __restrict__ table, uint32_t n_rows, uint32_t table_bytes,
                                                  const float4* __restrict__ big, uint32_t big_rows, float4* __restrict__ out,
                                                  int iters, int every, int stores, uint32_t seed) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave_id = blockIdx.x * 4u + (threadIdx.x >> 6);
  uint32_t s = seed ^ (wave_id * 2654435761u);
  s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(table), 0, (int)table_bytes, 0x00020000);
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
    float4 r[NR];
    const bool slow = every > 0 && it % every == 0;
    if (slow) {
      s = s * 1664525u + 1013904223u;
      const uint32_t row = (uint32_t)(((uint64_t)(s >> 4) * big_rows) >> 28);
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(big) + (size_t)row * (NR * 64), 0, NR * 1024, 0x00020000);
#pragma unroll
      for (int p = 0; p < NR; ++p) r[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, lane * 16 + p * 1024, 0, 2));
    }
    float4 t[NB];
#pragma unroll
    for (int k = 0; k < NB / 2; ++k) {
      s = s * 1664525u + 1013904223u;
      const uint32_t row = (uint32_t)(((uint64_t)(s >> 4) * n_rows) >> 28);
#pragma unroll
      for (int p = 0; p < 2; ++p)
        t[2 * k + p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((row * 2u + p) * 1024u + lane * 16u), 0, 0));
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) asm volatile("" ::"v"(t[k].x), "v"(t[k].y), "v"(t[k].z), "v"(t[k].w));
    acc += t[0].x;
    if (slow) {
#pragma unroll
      for (int p = 0; p < NR; ++p) asm volatile("" ::"v"(r[p].x), "v"(r[p].y), "v"(r[p].z), "v"(r[p].w));
      if (stores) {
        s = s * 1664525u + 1013904223u;
        const uint32_t row = (uint32_t)(((uint64_t)(s >> 4) * big_rows) >> 28);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)row * (NR * 64), 0, NR * 1024, 0x00020000);
#pragma unroll
        for (int p = 0; p < NR; ++p)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, r[p]), rr, lane * 16 + p * 1024, 0, 2);
      }
    }
  }
  if (acc == 12345.678f) out[0] = make_float4(acc, 0, 0, 0);
}


// MOVERS (round 5; DESIGN.md section 9.1's last structural idea for the row path): keep the HBM rows OUT of the tap CUs' memory
// pipeline.  m CUs of every XCD ("movers") stream 2 KiB rows from HBM into a ring that stays in their XCD's L2 and write
// finished rows of the ring back to HBM; the other CUs ("tappers") gather L2-resident rows only -- per batch 16 pieces of map
// rows from the 2 MB table and, at the row kernel's mix (one row per 1.4 batches), one 2 KiB row of the ring read and one
// written.  A workgroup cannot choose its CU, so it learns where it runs (HW_ID / XCC_ID) and the first m CUs of an XCD to
// register become its movers; every workgroup of a CU takes the CU's role.  The tappers run a fixed number of batches; the
// movers run until every tapper has finished (and at most `max_rows` rows: every wave reaches its exit).
struct MoverCtl {
  unsigned int role[8][256];   // per (XCD, CU key): 0 unknown, 1 mover, 2 tapper
  unsigned int cus_seen[8];    // CUs registered per XCD
  unsigned int wgs_registered, tapper_wgs, tapper_wgs_done;
  unsigned long long mover_rows, tapper_batches, t_first_done, t_last_done;
};
template <int R>
__global__ __launch_bounds__(256) void movers_kernel(const float4* __restrict__ table, uint32_t table_bytes, const float4* __restrict__ big,
                                                     float4* __restrict__ out, uint32_t big_rows, float4* __restrict__ rings,
                                                     uint32_t ring_rows, MoverCtl* __restrict__ ctl, int m, int iters, int max_rows,
                                                     uint32_t seed) {
  __shared__ unsigned int s_role;
  const int lane = threadIdx.x & 63;
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;
  const uint32_t key = (hw >> 8) & 255u;
  if (threadIdx.x == 0) {
    unsigned int r = atomicCAS(&ctl->role[xcc][key], 0u, 3u);  // 3: being decided by this workgroup
    if (r == 0u) {
      const unsigned int n = atomicAdd(&ctl->cus_seen[xcc], 1u);
      r = n < (unsigned int)m ? 1u : 2u;
      atomicExch(&ctl->role[xcc][key], r);
    } else {
      while (r == 3u) r = atomicAdd(&ctl->role[xcc][key], 0u);
    }
    s_role = r;
    if (r == 2u) atomicAdd(&ctl->tapper_wgs, 1u);
    __threadfence();
    atomicAdd(&ctl->wgs_registered, 1u);
  }
  __syncthreads();
  const bool mover = s_role == 1u;
  const uint32_t wave_id = blockIdx.x * 4u + (threadIdx.x >> 6);
  uint32_t s = seed ^ (wave_id * 2654435761u);
  s = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
  float4* ring = rings + (size_t)xcc * ring_rows * 128;  // this XCD's ring: ring_rows rows of 2 KiB
  if (mover) {
    unsigned long long rows = 0;
    for (int it = 0; it < max_rows; ++it) {
      if ((it & 15) == 0) {
        const unsigned int reg = atomicAdd(&ctl->wgs_registered, 0u), tw = atomicAdd(&ctl->tapper_wgs, 0u), td = atomicAdd(&ctl->tapper_wgs_done, 0u);
        if (__builtin_amdgcn_readfirstlane((int)(reg == gridDim.x && td >= tw))) break;
      }
      // R rows per iteration: HBM -> ring, ring -> HBM (2 R pieces in flight each way)
      float4 a[2 * R], b[2 * R];
      uint32_t src[R], slot[R];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        s = s * 1664525u + 1013904223u;
        src[k] = (uint32_t)(((uint64_t)(s >> 4) * big_rows) >> 28);
        s = s * 1664525u + 1013904223u;
        slot[k] = (uint32_t)(((uint64_t)(s >> 4) * ring_rows) >> 28);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(big) + (size_t)src[k] * 128, 0, 2048, 0x00020000);
        a[2 * k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, lane * 16, 0, 2));
        a[2 * k + 1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, lane * 16 + 1024, 0, 2));
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(ring + (size_t)slot[k] * 128, 0, 2048, 0x00020000);
        b[2 * k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rg, lane * 16, 0, 1));
        b[2 * k + 1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rg, lane * 16 + 1024, 0, 1));
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(ring + (size_t)((slot[k] + 7u) % ring_rows) * 128, 0, 2048, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, a[2 * k]), rg, lane * 16, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, a[2 * k + 1]), rg, lane * 16 + 1024, 0, 0);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)src[k] * 128, 0, 2048, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, b[2 * k]), ro, lane * 16, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, b[2 * k + 1]), ro, lane * 16 + 1024, 0, 2);
      }
      rows += R;
    }
    if (lane == 0) atomicAdd(&ctl->mover_rows, rows);
    return;
  }
  // ---- tapper: the row kernel's mix, every request an L2 hit
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(table), 0, (int)table_bytes, 0x00020000);
  const uint32_t n_rows = table_bytes / 2048u;
  float acc = 0.0f;
  uint32_t credit = 0;  // in units of 1 / 14 row: 10 per batch -> one ring row read and one written per 1.4 batches
  for (int it = 0; it < iters; ++it) {
    float4 t[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s = s * 1664525u + 1013904223u;
      const uint32_t row = (uint32_t)(((uint64_t)(s >> 4) * n_rows) >> 28);
#pragma unroll
      for (int p = 0; p < 2; ++p)
        t[2 * k + p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((row * 2u + p) * 1024u + lane * 16u), 0, 0));
    }
    credit += 10u;
    const bool row_now = credit >= 14u;
    float4 r0, r1;
    uint32_t slot = 0;
    if (row_now) {
      credit -= 14u;
      s = s * 1664525u + 1013904223u;
      slot = (uint32_t)(((uint64_t)(s >> 4) * ring_rows) >> 28);
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(ring + (size_t)slot * 128, 0, 2048, 0x00020000);
      r0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rg, lane * 16, 0, 1));  // sc0: past the L1, from the L2
      r1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rg, lane * 16 + 1024, 0, 1));
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(t[k].x), "v"(t[k].y), "v"(t[k].z), "v"(t[k].w));
    acc += t[0].x;
    if (row_now) {
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(ring + (size_t)((slot + 3u) % ring_rows) * 128, 0, 2048, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, r0), rg, lane * 16, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, r1), rg, lane * 16 + 1024, 0, 0);
    }
  }
  if (acc == 12345.678f) out[0] = make_float4(acc, 0, 0, 0);
  if (lane == 0) atomicAdd(&ctl->tapper_batches, (unsigned long long)iters);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    atomicMin(&ctl->t_first_done, now);
    atomicMax(&ctl->t_last_done, now);
    __threadfence();
    atomicAdd(&ctl->tapper_wgs_done, 1u);
  }
}

using Fn = void (*)(const float4*, uint32_t, uint32_t, int, uint32_t, float*);

struct Variant {
  const char* name;
  Fn fn;
  int nb, ppr;
};

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# device %s, %d CUs, clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
  const size_t max_table = 64u << 20;
  float4* table;
  float* sink;
  CK(hipMalloc(&table, max_table));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(table, 0, max_table));
  if (argc > 1 && argv[1][0] == 'v') {  // "movers"
    const size_t big_bytes = 8ull << 30;
    float4 *big, *out, *rings;
    MoverCtl* ctl;
    const uint32_t ring_rows = 512;  // 1 MB per XCD: with the 2 MB table well inside a 4 MB L2
    CK(hipMalloc(&big, big_bytes));
    CK(hipMalloc(&out, big_bytes));
    CK(hipMalloc(&rings, (size_t)8 * ring_rows * 2048));
    CK(hipMalloc(&ctl, sizeof(MoverCtl)));
    CK(hipMemset(big, 0, big_bytes));
    CK(hipMemset(rings, 0, (size_t)8 * ring_rows * 2048));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t tb = 2u << 20, big_rows = (uint32_t)(big_bytes / 2048u);
    printf("# movers: m CUs per XCD move 2 KiB rows HBM -> L2 ring -> HBM; the others gather 16 pieces of 2 MB-table rows per batch + one ring row read\n"
           "# and one written per 1.4 batches (all L2).  8 waves per CU.  Target: tappers >= 24 TB/s with movers >= 3.2 TB/s of HBM traffic.\n");
    for (int rr : {4, 8, 12})
    for (int m : {0, 2, 4, 6, 8, 12}) {
      if (m == 0 && rr != 4) continue;
      const int grid = cus * 2, iters = 2048;
      float best = 1e30f;
      MoverCtl h_best{};
      for (int rep = 0; rep < 4; ++rep) {  // (the first one warms up)
        MoverCtl z{};
        z.t_first_done = ~0ull;
        CK(hipMemcpy(ctl, &z, sizeof(z), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, 0));
        auto kfn = rr == 4 ? movers_kernel<4> : (rr == 8 ? movers_kernel<8> : movers_kernel<12>);
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), 0, 0, table, tb, big, out, big_rows, rings, ring_rows, ctl, m, iters, 1 << 20, 7u + rep);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) {
          best = ms;
          CK(hipMemcpy(&h_best, ctl, sizeof(h_best), hipMemcpyDeviceToHost));
        }
      }
      unsigned int movers_cus = 0, tap_cus = 0;
      for (int x = 0; x < 8; ++x)
        for (int k = 0; k < 256; ++k) { movers_cus += h_best.role[x][k] == 1u; tap_cus += h_best.role[x][k] == 2u; }
      const double tap_gb = (double)h_best.tapper_batches * (16 + 2.0 / 1.4) * 1024.0 / 1e9;  // gathered: taps + ring rows read
      const double ring_w_gb = (double)h_best.tapper_batches * (2.0 / 1.4) * 1024.0 / 1e9;
      const double hbm_gb = (double)h_best.mover_rows * 4096.0 / 1e9;  // 2 KiB read + 2 KiB written per row
      printf("rows in flight per mover wave %2d (x 8 waves per CU)  m = %2d of 32 per XCD (%3u mover CUs, %3u tap CUs seen, %u tap workgroups): %7.3f ms  tappers gather %6.2f TB/s (+ %5.2f TB/s ring writes)  movers %5.2f TB/s of HBM traffic (%.2f M rows)  tappers finished within %.3f ms of each other\n",
             rr, m, movers_cus, tap_cus, h_best.tapper_wgs, best, tap_gb / best, ring_w_gb / best, hbm_gb / best, h_best.mover_rows / 1e6,
             (double)(h_best.t_last_done - h_best.t_first_done) / 1e5);
      fflush(stdout);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'm') {  // the mixed stream
    // "mix" = an 8 GB table (HBM); "mix64" etc. = a table of that many MB, read over and over: after the first pass it lives in
    // the 256 MB Infinity Cache -- what the slow rows would cost if something had brought them there ahead of time
    const size_t big_bytes = argv[1][3] ? (size_t)atoi(argv[1] + 3) << 20 : 8ull << 30;
    float4 *big, *out;
    CK(hipMalloc(&big, big_bytes));
    CK(hipMalloc(&out, big_bytes));
    CK(hipMemset(big, 0, big_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t tb = 2u << 20, n_rows = tb / 2048u, big_rows = (uint32_t)(big_bytes / 2048u);
    printf("# per batch 16 pieces of 2 KiB rows from a 2 MB table (L2) + every k-th batch one 2 KiB row of a %zu MB table (nt)\n", big_bytes >> 20);
    for (int wpc : {8, 16}) {
      for (int every : {0, 4, 2, 1}) {
        for (int stores : {0, 1}) {
          if (every == 0 && stores) continue;
          const int grid = cus * wpc / 4;
          const int iters = 2048;
          hipLaunchKernelGGL((mix_kernel<16, 2>), dim3(grid), dim3(256), 0, 0, table, n_rows, tb, big, big_rows, out, iters / 4, every, stores, 1u);
          CK(hipDeviceSynchronize());
          float best = 1e30f;
          for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((mix_kernel<16, 2>), dim3(grid), dim3(256), 0, 0, table, n_rows, tb, big, big_rows, out, iters, every, stores, 7u + rep);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
          }
          const double waves = (double)grid * 4;
          const double fast_gb = waves * iters * 16 * 1024.0 / 1e9;
          const double slow_gb = every ? waves * (iters / every) * 2 * 1024.0 / 1e9 : 0.0;
          printf("waves/CU %2d  slow row every %2d batches%s: %7.3f ms  L2 rows %6.2f TB/s  + HBM rows %5.2f TB/s read%s  (slow share of requests %4.1f %%)\n",
                 wpc, every, stores ? " + stored back" : "              ", best, fast_gb / best, slow_gb / best,
                 stores ? " and as much written" : "", 100.0 * slow_gb / (slow_gb + fast_gb + 1e-30));
          fflush(stdout);
        }
      }
    }
    return 0;
  }
  const Variant vars[] = {
      {"buffer_load x 8 pieces (2 KiB rows)", gather_kernel<8, 2, 0>, 8, 2},
      {"buffer_load x 16 pieces (2 KiB rows: the row kernel's 2 tap groups)", gather_kernel<16, 2, 0>, 16, 2},
      {"buffer_load x 32 pieces (2 KiB rows)", gather_kernel<32, 2, 0>, 32, 2},
      {"global_load x 16 pieces (2 KiB rows)", gather_kernel<16, 2, 1>, 16, 2},
      {"global_load nt x 16 pieces (2 KiB rows)", gather_kernel<16, 2, 2>, 16, 2},
      {"buffer_load x 16 pieces (1 KiB rows)", gather_kernel<16, 1, 0>, 16, 1},
      {"buffer_load x 16 pieces (4 KiB rows)", gather_kernel<16, 4, 0>, 16, 4},
  };
  const size_t tables[] = {256u << 10, 1u << 20, 2u << 20, 3u << 20, 4u << 20, 9437184u /* 128 maps of 36 x 2 KiB */, 32u << 20};
  const int wpcs[] = {4, 8, 12, 16};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (const Variant& v : vars) {
    for (size_t tb : tables) {
      for (int wpc : wpcs) {
        const uint32_t row_bytes = 1024u * (uint32_t)v.ppr;
        const uint32_t n_rows = (uint32_t)(tb / row_bytes);
        const int grid = cus * wpc / 4;
        // ~ 24 GB gathered per launch
        const double bytes_per_iter = (double)grid * 4 * v.nb * 1024.0;
        int iters = (int)(24e9 / bytes_per_iter);
        if (iters < 4) iters = 4;
        hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, table, n_rows, (uint32_t)tb, iters / 4, 1u, sink);  // warm
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0, 0));
          hipLaunchKernelGGL(v.fn, dim3(grid), dim3(256), 0, 0, table, n_rows, (uint32_t)tb, iters, 7u + rep, sink);
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        const double gb = bytes_per_iter * iters / 1e9;
        printf("%-70s table %8.2f MB  waves/CU %2d  %7.3f ms  %6.2f TB/s  (%5.1f GB/s per CU)\n", v.name, tb / 1048576.0, wpc, best,
               gb / best, gb / best * 1000.0 / cus);
        fflush(stdout);
      }
    }
  }
  return 0;
}
