// saf_common.h -- shared device helpers for the gfx950 fusion kernels.
//
// Numerics contract: everything that decides WHICH voxels a frame touches (projection, grid
// normalisation, nearest-pixel depth test) is written one IEEE fp32 operation at a time in the
// order the reference's PyTorch CPU path executes it (clipfusion.py:647-679), with the two BLAS
// 3x3 products in the accumulation order measured against the reference.
// The translation unit is compiled with -ffp-contract=off; FMAs appear only where written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/saf.h"

#pragma clang fp contract(off)

namespace saf {

constexpr int kWave = 64;
constexpr int kNumLists = 16;        // independent compact lists (one counter each)
constexpr int kSweepThreads = 256;
constexpr int kSweepPerThread = 16;  // voxels per thread -> 4096 voxels per sweep block
constexpr int kSweepChunk = kSweepThreads * kSweepPerThread;
constexpr int kFuseThreads = 512;
constexpr int kAxisLds = 1024;       // axis tables staged in LDS by the sweep when nx+ny+nz fits (4 KiB)

// Streamed volume rows are touched once per frame and the volume (GBs) dwarfs every cache:
// non-temporal accesses bypass the 32 KiB vector L1, whose line count otherwise caps the misses a
// CU can keep in flight (measured: TCP pending-stall 81 %, ~11 KB in flight per CU with plain loads).
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
  const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream(float4* p, const float4 x) {
  v4f_t v;
  v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
  __builtin_nontemporal_store(v, reinterpret_cast<v4f_t*>(p));
}

// bf16 <-> f32 exactly as PyTorch does it: widening is a shift, narrowing rounds to nearest even
// (NaN stays NaN).  The oracle uses the same integer arithmetic.
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// f32 -> bf16, round to nearest even, NaN quieted ((u >> 16) | 0x40): gfx950 has the conversion in hardware
// (v_cvt_pk_bf16_f32: two values per instruction), bit for bit the software sequence
//   (u + 0x7fff + ((u >> 16) & 1)) >> 16
// on every f32 pattern tried (tools/bf16_cvt_test.hip: all NaNs, infinities, the denormal range, 70 M strided patterns).
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  return (uint32_t)__builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// n / d for n < 2^31 by multiply-shift (exact; see saf_fuse.hip make_fastdiv).
struct FastDiv {
  uint32_t mul, shift, d, pad;
};
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return (uint32_t)(((uint64_t)n * f.mul) >> f.shift);
}

// Camera of one frame, read once per thread from device memory (uniform -> scalar loads).
struct Cam {
  float r00, r01, r02, r10, r11, r12, r20, r21, r22;  // pose[:3,:3] (cam->world)
  float tx, ty, tz;
  float k00, k01, k02, k10, k11, k12, k20, k21, k22;
  float fw, fh;    // (float)W, (float)H
  float rfw, rfh;  // RN(1/W), RN(1/H)
  float sfx, sfy;  // W/2, H/2
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ pose, const float* __restrict__ K, int width,
                                        int height) {
  Cam c;
  c.r00 = pose[0]; c.r01 = pose[1]; c.r02 = pose[2];  c.tx = pose[3];
  c.r10 = pose[4]; c.r11 = pose[5]; c.r12 = pose[6];  c.ty = pose[7];
  c.r20 = pose[8]; c.r21 = pose[9]; c.r22 = pose[10]; c.tz = pose[11];
  c.k00 = K[0]; c.k01 = K[1]; c.k02 = K[2];
  c.k10 = K[3]; c.k11 = K[4]; c.k12 = K[5];
  c.k20 = K[6]; c.k21 = K[7]; c.k22 = K[8];
  c.fw = (float)width;
  c.fh = (float)height;
  c.rfw = 1.0f / c.fw;  // IEEE divisions: correctly rounded reciprocals
  c.rfh = 1.0f / c.fh;
  c.sfx = c.fw / 2.0f;
  c.sfy = c.fh / 2.0f;
  return c;
}

// R^T (x - t): products rounded, added as (p0 + p2) + p1 (the order of the reference's BLAS call
// for this operand layout); K @ xyz_cam: k-ascending FMA chain.
__device__ __forceinline__ float dot3_rt(float a0, float a1, float a2, float b0, float b1, float b2) {
  float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
  return (p0 + p2) + p1;
}
__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
  float acc = a0 * b0;
  acc = __builtin_fmaf(a1, b1, acc);
  acc = __builtin_fmaf(a2, b2, acc);
  return acc;
}

// grid_sample un-normalisation, align_corners=False: (g + 1) * (size / 2) - 0.5
__device__ __forceinline__ float unnormalize(float g, float half_size) { return (g + 1.0f) * half_size - 0.5f; }

struct Proj {
  float gx, gy, z;
};

// uvz = K @ (R^T (x - t))   (clipfusion.py:648-653): homogeneous pixel coordinates, no division yet.
struct Uvz {
  float u, v, z;
};
__device__ __forceinline__ Uvz project_uvz(const Cam& c, float xw, float yw, float zw) {
  float dx = xw - c.tx, dy = yw - c.ty, dz = zw - c.tz;
  float cx = dot3_rt(c.r00, c.r10, c.r20, dx, dy, dz);
  float cy = dot3_rt(c.r01, c.r11, c.r21, dx, dy, dz);
  float cz = dot3_rt(c.r02, c.r12, c.r22, dx, dy, dz);
  Uvz r;
  r.u = dot3(c.k00, c.k01, c.k02, cx, cy, cz);
  r.v = dot3(c.k10, c.k11, c.k12, cx, cy, cz);
  r.z = dot3(c.k20, c.k21, c.k22, cx, cy, cz);
  return r;
}

// a / b for a divisor that is uniform over the launch, given y = RN(1/b) (one IEEE division per
// thread): q0 = RN(a*y) is within 2 ulp, one FMA refinement makes it faithful, and for a faithful
// quotient with the exact residual the final FMA rounds a/b correctly (Markstein).  Five full-rate
// instructions instead of the ~11-instruction IEEE expansion, no branches.  Bit-identical to IEEE
// division for every finite a in the normal range (tools/divtest.c: all 2^32 dividends x 23 divisors,
// 400 M random pairs).  NOT identical for a = +-inf (gives NaN) or results in the denormal range;
// callers use it only where those cases cannot change a decision (see the call sites).
__device__ __forceinline__ float div_by_uniform(float a, float b, float y) {
  const float q0 = a * y;
  const float r0 = __builtin_fmaf(-b, q0, a);
  const float q1 = __builtin_fmaf(r0, y, q0);
  const float r1 = __builtin_fmaf(-b, q1, a);
  return __builtin_fmaf(r1, y, q1);
}

// clipfusion.py:654-659: uv = uvz[:2] / z ; grid = ((uv + 0.5) / [W, H]) * 2 - 1.  finish_from_uv: everything behind the
// two IEEE divisions (qu = u / z, qv = v / z).
__device__ __forceinline__ Proj finish_from_uv(const Cam& c, float qu, float qv, float z) {
  Proj p;
  float gx = qu + 0.5f, gy = qv + 0.5f;
  gx = div_by_uniform(gx, c.fw, c.rfw);
  gy = div_by_uniform(gy, c.fh, c.rfh);
  gx = gx * 2.0f;
  gy = gy * 2.0f;
  p.gx = gx - 1.0f;
  p.gy = gy - 1.0f;
  p.z = z;
  return p;
}
__device__ __forceinline__ Proj finish_projection(const Cam& c, const Uvz& h) {
  Proj p;
  float gx = h.u / h.z, gy = h.v / h.z;
  gx = gx + 0.5f;
  gy = gy + 0.5f;
  // (uv + 0.5) / [W, H].  For every lane that can be in view (z > 0, finite inputs) the dividend is
  // finite and either 0 or >= 2^-25 in magnitude, where div_by_uniform is exact; on the other lanes
  // (z <= 0, inf, NaN) any result fails |grid| <= 1 or z > 0 exactly as the IEEE quotient does.
  gx = div_by_uniform(gx, c.fw, c.rfw);
  gy = div_by_uniform(gy, c.fh, c.rfh);
  gx = gx * 2.0f;
  gy = gy * 2.0f;
  p.gx = gx - 1.0f;
  p.gy = gy - 1.0f;
  p.z = h.z;
  return p;
}

// clipfusion.py:647-659: voxel centre -> normalised image coordinates + camera depth.
__device__ __forceinline__ Proj project(const Cam& c, float xw, float yw, float zw) {
  return finish_projection(c, project_uvz(c, xw, yw, zw));
}

// Nearest-neighbour grid_sample tap (zeros padding): pixel offset or -1.
__device__ __forceinline__ int nearest_index(float gx, float gy, const Cam& c, int width) {
  float xn = __builtin_rintf(unnormalize(gx, c.sfx));
  float yn = __builtin_rintf(unnormalize(gy, c.sfy));
  bool inb = (xn > -1.0f) && (xn < c.fw) && (yn > -1.0f) && (yn < c.fh);
  return inb ? (int)yn * width + (int)xn : -1;
}

// Bilinear grid_sample taps (ATen GridSamplerKernel.cpp ApplyGridSample<Bilinear, zeros>).
struct Bilin {
  int x0, y0;
  float nw, ne, sw, se;
};
__device__ __forceinline__ Bilin bilinear_setup(float gx, float gy, float half_w, float half_h) {
  float x = unnormalize(gx, half_w), y = unnormalize(gy, half_h);
  float xw = __builtin_floorf(x), yn = __builtin_floorf(y);
  float wx = x - xw, ex = 1.0f - wx;
  float ny = y - yn, sy = 1.0f - ny;
  Bilin b;
  b.nw = sy * ex;
  b.ne = sy * wx;
  b.sw = ny * ex;
  b.se = ny * wx;
  b.x0 = (int)xw;
  b.y0 = (int)yn;
  return b;
}

}  // namespace saf
