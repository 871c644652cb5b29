// saf_misc.hip -- the small kernels either side of the fuse loop:
//   * backproject_lattice : per-frame body of backproject_pcd (reference clipfusion.py:541-565)
//   * merge_finalize / mean_to_sum : sums <-> means around the cross-rank reduction
//   * label_argmax : argmax-with-empty-check of the label histogram (clip_seem_fusion.py:315-325)
#include "saf_common.h"
#include "saf_host.h"

#pragma clang fp contract(off)

namespace saf {
namespace {

// xyz_cam = pix_vec * depth ; world = R @ xyz_cam + t      clipfusion.py:553-558
__global__ void backproject_kernel(const float* __restrict__ depth, int width, const float* __restrict__ pose,
                                   const float* __restrict__ Kinv, const int* __restrict__ u_idx, int nu,
                                   const int* __restrict__ v_idx, int nv, float max_depth, float* __restrict__ xyz,
                                   uint8_t* __restrict__ valid) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= nu * nv) return;
  const int j = o / nu, i = o - j * nu;
  const int u = u_idx[i], v = v_idx[j];
  const float d = depth[(int64_t)v * width + u];
  const float fu = (float)u, fv = (float)v;
  // get_pix_vecs: K^-1 @ [u, v, 1]^T                      clipfusion.py:497-507
  const float rx = dot3(Kinv[0], Kinv[1], Kinv[2], fu, fv, 1.0f);
  const float ry = dot3(Kinv[3], Kinv[4], Kinv[5], fu, fv, 1.0f);
  const float rz = dot3(Kinv[6], Kinv[7], Kinv[8], fu, fv, 1.0f);
  const float cx = rx * d, cy = ry * d, cz = rz * d;
  xyz[(int64_t)o * 3 + 0] = dot3(pose[0], pose[1], pose[2], cx, cy, cz) + pose[3];
  xyz[(int64_t)o * 3 + 1] = dot3(pose[4], pose[5], pose[6], cx, cy, cz) + pose[7];
  xyz[(int64_t)o * 3 + 2] = dot3(pose[8], pose[9], pose[10], cx, cy, cz) + pose[11];
  // valid = ~isnan(depth) & (depth > 0) & (depth < max_depth)   :551
  valid[o] = (uint8_t)((d == d) && (d > 0.0f) && (d < max_depth));
}

// feature rows: x /= w (TO_MEAN) or x *= w; 16-byte accesses when D % 4 == 0
template <bool TO_MEAN, int VEC>
__global__ void scale_rows_kernel(float* __restrict__ feat, const int* __restrict__ weight, int64_t first,
                                  int64_t count, int DV) {
  const int64_t total = count * DV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = first + i / DV;
    const int w = weight[n];
    if (TO_MEAN && w <= 0) continue;
    const float fw = (float)w;
    if (VEC == 4) {
      float4* p = reinterpret_cast<float4*>(feat) + first * DV + i;
      float4 x = *p;
      if (TO_MEAN) { x.x = x.x / fw; x.y = x.y / fw; x.z = x.z / fw; x.w = x.w / fw; }
      else         { x.x = x.x * fw; x.y = x.y * fw; x.z = x.z * fw; x.w = x.w * fw; }
      *p = x;
    } else {
      float* p = feat + first * DV + i;
      *p = TO_MEAN ? *p / fw : *p * fw;
    }
  }
}

template <bool TO_MEAN>
__global__ void scale_scalars_kernel(float* __restrict__ rgb, float* __restrict__ tsdf, const int* __restrict__ weight,
                                     const int* __restrict__ tsdf_w, int64_t first, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = first + i;
    const int w = weight[n], wt = tsdf_w[n];
    if (TO_MEAN) {
      if (w > 0) {
        const float fw = (float)w;
        rgb[n * 3 + 0] = rgb[n * 3 + 0] / fw;
        rgb[n * 3 + 1] = rgb[n * 3 + 1] / fw;
        rgb[n * 3 + 2] = rgb[n * 3 + 2] / fw;
      }
      if (wt > 0) tsdf[n] = tsdf[n] / (float)wt;
    } else {
      const float fw = (float)w;
      rgb[n * 3 + 0] = rgb[n * 3 + 0] * fw;
      rgb[n * 3 + 1] = rgb[n * 3 + 1] * fw;
      rgb[n * 3 + 2] = rgb[n * 3 + 2] * fw;
      tsdf[n] = tsdf[n] * (float)wt;
    }
  }
}

template <bool TO_MEAN>
int scale_volume(const saf_volume* vol, int64_t first, int64_t count, hipStream_t s) {
  if (!vol || !vol->clip_feat || !vol->weight || !vol->tsdf_weight || !vol->rgb || !vol->tsdf)
    return fail(SAF_E_INVALID, "merge: volume has a NULL buffer");
  if (vol->feat_dtype != SAF_F32) return fail(SAF_E_UNSUPPORTED, "merge: only SAF_F32");
  const int64_t N = n_voxels(vol);
  if (first < 0 || count < 0 || first + count > N) return fail(SAF_E_INVALID, "merge: bad voxel range");
  if (count == 0) return SAF_OK;
  const int D = vol->feat_dim;
  const int cap = device_cus() * 8;
  float* feat = static_cast<float*>(vol->clip_feat);
  if (D % 4 == 0 && ((uintptr_t)feat & 15) == 0) {
    const int64_t total = count * (D / 4);
    const int blocks = (int)((total + 255) / 256 < cap ? (total + 255) / 256 : cap);
    hipLaunchKernelGGL((scale_rows_kernel<TO_MEAN, 4>), dim3(blocks), dim3(256), 0, s, feat, vol->weight, first, count,
                       D / 4);
  } else {
    const int64_t total = count * D;
    const int blocks = (int)((total + 255) / 256 < cap ? (total + 255) / 256 : cap);
    hipLaunchKernelGGL((scale_rows_kernel<TO_MEAN, 1>), dim3(blocks), dim3(256), 0, s, feat, vol->weight, first, count,
                       D);
  }
  int rc = check_launch("scale_rows_kernel");
  if (rc) return rc;
  const int blocks = (int)((count + 255) / 256 < cap ? (count + 255) / 256 : cap);
  hipLaunchKernelGGL(scale_scalars_kernel<TO_MEAN>, dim3(blocks), dim3(256), 0, s, vol->rgb, vol->tsdf, vol->weight,
                     vol->tsdf_weight, first, count);
  return check_launch("scale_scalars_kernel");
}

// one wave per voxel row of the histogram; first maximum wins (torch.argmax on CPU), empty row -> -1
__global__ __launch_bounds__(256) void label_argmax_kernel(const int* __restrict__ labels, int64_t n_vox, int C,
                                                            int* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t n = wave; n < n_vox; n += n_waves) {
    const int* r = labels + n * C;
    unsigned long long best = 0ull;
    int any = 0;
    for (int c = lane; c < C; c += 64) {
      const int val = r[c];
      any |= (val != 0);
      // order by (value, -index): bias the value to unsigned, store ~index in the low word
      const unsigned long long key = ((unsigned long long)((uint32_t)val ^ 0x80000000u) << 32) | (uint32_t)(~(uint32_t)c);
      best = key > best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = __shfl_xor(best, o);
      best = other > best ? other : best;
      any |= __shfl_xor(any, o);
    }
    if (lane == 0) out[n] = any ? (int)(~(uint32_t)(best & 0xffffffffull)) : -1;
  }
}

// ------------------------------------------------------------------------------------------
// extract_mesh's vertex sampling: 3-D grid_sample at marching-cubes vertices.
// One wave per vertex: the 8 corner rows (D contiguous values each) are gathered with all lanes.
// Arithmetic follows ATen's scalar 3-D grid sampler (grid_sampler_3d_cpu_impl): un-normalise
// ((g + 1) * size - 1) / 2, corner weights as products of distances to the opposite corner,
// corners summed in the order tnw, tne, tsw, tse, bnw, bne, bsw, bse, out-of-volume corners skipped.
// ------------------------------------------------------------------------------------------
struct Axis3 {
  float g;   // normalised grid coordinate
  float f;   // un-normalised source index
  int i0;    // floor
  float w0, w1;  // weights of corner i0 and i0 + 1
};
__device__ __forceinline__ Axis3 axis_setup(float v_index, int size) {
  Axis3 a;
  // grid = (verts + 0.5) / nvox * 2 - 1 : numpy f32 + 0.5, then Tensor.__rtruediv__ = reciprocal(nvox) * x
  const float r = 1.0f / (float)size;
  float g = (v_index + 0.5f) * r;
  g = g * 2.0f;
  g = g - 1.0f;
  a.g = g;
  a.f = ((g + 1.0f) * (float)size - 1.0f) / 2.0f;
  const float fl = __builtin_floorf(a.f);
  a.i0 = (int)fl;
  a.w0 = (fl + 1.0f) - a.f;  // distance to the far corner  (ix_bse - ix)
  a.w1 = a.f - fl;           // distance to the near corner (ix - ix_tnw)
  return a;
}

template <bool BF16>
__global__ __launch_bounds__(256) void sample_vertices_kernel(
    int nx, int ny, int nz, int D, const void* __restrict__ feat, const float* __restrict__ rgb,
    const float* __restrict__ verts, int64_t n_verts, float* __restrict__ out_feat, float* __restrict__ out_rgb,
    const int* __restrict__ obj_idx, float* __restrict__ out_obj, const float* __restrict__ seg_color,
    float* __restrict__ out_seg) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t vtx = wave; vtx < n_verts; vtx += n_waves) {
    // the reference swaps the grid to (z, y, x) order because grid_sample's x runs along the LAST
    // volume axis (clipfusion.py:742): axis X of the sampler = nz, Y = ny, Z = nx.
    const Axis3 ax = axis_setup(verts[vtx * 3 + 2], nz);
    const Axis3 ay = axis_setup(verts[vtx * 3 + 1], ny);
    const Axis3 az = axis_setup(verts[vtx * 3 + 0], nx);
    int64_t row[8];
    float w[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {  // c = (dz << 2) | (dy << 1) | dx : tnw, tne, tsw, tse, bnw, ...
      const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
      const int x = ax.i0 + dx, y = ay.i0 + dy, z = az.i0 + dz;  // x along nz, y along ny, z along nx
      const bool in = x >= 0 && x < nz && y >= 0 && y < ny && z >= 0 && z < nx;
      row[c] = in ? ((int64_t)z * ny + y) * nz + x : -1;
      w[c] = (dx ? ax.w1 : ax.w0) * (dy ? ay.w1 : ay.w0) * (dz ? az.w1 : az.w0);
    }
    for (int ch = lane; ch < D; ch += 64) {
      float acc = 0.0f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (row[c] >= 0) {
          const float val = BF16 ? __builtin_bit_cast(float, (uint32_t) static_cast<const uint16_t*>(feat)[row[c] * D + ch] << 16)
                                 : static_cast<const float*>(feat)[row[c] * D + ch];
          acc += val * w[c];
        }
      }
      out_feat[vtx * D + ch] = acc;
    }
    if (lane < 3) {
      float acc = 0.0f;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (row[c] >= 0) acc += rgb[row[c] * 3 + lane] * w[c];
      out_rgb[vtx * 3 + lane] = fminf(fmaxf(acc, 0.0f), 1.0f);  // .clamp(0, 1)
    }
    if (obj_idx || seg_color) {
      // mode="nearest": nearbyint of the un-normalised index, zeros outside
      const float xn = __builtin_rintf(ax.f), yn = __builtin_rintf(ay.f), zn = __builtin_rintf(az.f);
      const bool in = xn >= 0.0f && xn < (float)nz && yn >= 0.0f && yn < (float)ny && zn >= 0.0f && zn < (float)nx;
      const int64_t n = in ? ((int64_t)zn * ny + (int64_t)yn) * nz + (int64_t)xn : -1;
      if (obj_idx && lane == 0) out_obj[vtx] = n >= 0 ? (float)obj_idx[n] : 0.0f;
      if (seg_color && lane < 3) out_seg[vtx * 3 + lane] = n >= 0 ? fminf(fmaxf(seg_color[n * 3 + lane], 0.0f), 1.0f) : 0.0f;
    }
  }
}


// One launch copies a frame's inputs into a slot of the staging ring behind integrate() (depth, rgb, pose, K, feature map,
// label map): six tiny copy launches per frame would be most of the host's work per call and fill the stream's queue.
// Round 6: a flat grid sized to the bytes (a workgroup = 256 lanes x 4 x 16 bytes of ONE segment) instead of 450 x 6 workgroups
// of 4-byte copies, most of which found nothing to do: the ring's staging runs BESIDE the window's row kernel and the next
// window's classification, and 2 700 workgroups per frame took the wave slots the classification lives in (its launches beside
// 96 staging kernels: 7.2 ms each, without: 5.1 -- profiles/r06/api_steady_state.txt).
struct StageArgs {
  const float* src[6];
  float* dst[6];
  int n[6];
  int first_block[7];  // segment k owns workgroups [first_block[k], first_block[k + 1])
  // the feature map may be a permuted view ([D, npy, npx] element strides); everything else is contiguous
  int64_t fs0, fs1, fs2;
  int f1, f2;
};
constexpr int kStageUnroll = 4;  // 16-byte pieces per lane
__global__ __launch_bounds__(256) void stage_frame_kernel(StageArgs a) {
  int seg = 0;
#pragma unroll
  for (int k = 1; k < 6; ++k) seg += (int)blockIdx.x >= a.first_block[k];
  const int n = a.n[seg];
  const float* __restrict__ src = a.src[seg];
  float* __restrict__ dst = a.dst[seg];
  const int blk = blockIdx.x - a.first_block[seg];
  if (seg == 4) {  // (a strided gather; 4 elements per lane and workgroup step as for the copies)
    for (int i = blk * 256 * 4 * kStageUnroll + threadIdx.x, e = min(n, (blk + 1) * 256 * 4 * kStageUnroll); i < e; i += 256) {
      const int x = i % a.f2, y = (i / a.f2) % a.f1, c = i / (a.f2 * a.f1);
      dst[i] = src[(int64_t)c * a.fs0 + (int64_t)y * a.fs1 + (int64_t)x * a.fs2];
    }
    return;
  }
  const int i0 = blk * 256 * 4 * kStageUnroll;  // this workgroup's first element
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 && i0 + 256 * 4 * kStageUnroll <= n) {
    typedef float st_f4 __attribute__((ext_vector_type(4)));
    const st_f4* s4 = reinterpret_cast<const st_f4*>(src + i0);
    st_f4* d4 = reinterpret_cast<st_f4*>(dst + i0);
    st_f4 v[kStageUnroll];
#pragma unroll
    for (int u = 0; u < kStageUnroll; ++u) v[u] = s4[u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < kStageUnroll; ++u) d4[u * 256 + threadIdx.x] = v[u];
  } else {  // a segment's last workgroup, pose / K, or rows that do not lie on 16-byte boundaries
    for (int i = i0 + threadIdx.x, e = min(n, i0 + 256 * 4 * kStageUnroll); i < e; i += 256) dst[i] = src[i];
  }
}

// Rows of voxels that were never written (weight == 0) are zero by contract (include/saf.h, saf_volume).  A volume that
// is recycled for a new scan need not be cleared up front -- the windowed fuse path never reads such rows -- as long as
// the rows that are STILL unwritten are zeroed before anyone else looks: this kernel.  A wave checks 64 weights at a
// time and writes zeros only where needed (after a 512-frame scan: almost nowhere).
// Round 5: on a coherent scene most rows ARE unwritten (83 % at 256^3: 28 GB of zeros, a fifth of the job).  The zeros leave as
// ONE 16-byte store per lane -- a native vector type: HIP's `uint4` (a struct around a union) assigned from make_uint4 compiled
// to narrower stores and the kernel wrote at half the rate of a memset (3.4 against 6.8 TB/s; now 5.4-5.6: tools/probe_clear_rate.py;
// rows per wave and iteration, nontemporal stores, the grid size up to one workgroup per 256 rows, and whole runs of 64 or 256
// unwritten rows written linearly by the wave / the workgroup without the per-row bookkeeping were measured and change nothing).
// `masks` (may be NULL): the hit masks of a window whose row kernel may be running beside this kernel -- n_planes planes of one
// word per voxel; a voxel with a bit set is that kernel's to write (it does not read the old row of a weight-0 voxel either).
template <int ESZ>
__global__ __launch_bounds__(256) void clear_unwritten_kernel(void* __restrict__ feat, const int* __restrict__ weight, int64_t first,
                                                              int64_t count, int row_bytes, const uint32_t* __restrict__ masks,
                                                              size_t mask_plane, int n_planes) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  typedef unsigned int clr_v4u __attribute__((ext_vector_type(4)));
  const clr_v4u z4 = {0u, 0u, 0u, 0u};
  for (int64_t i0 = wave * 64; i0 < count; i0 += n_waves * 64) {
    const int64_t n = first + i0 + lane;
    bool mine = i0 + lane < count && weight[n] == 0;
    if (masks && mine) {
      uint32_t any = 0u;
      for (int p = 0; p < n_planes; ++p) any |= masks[(size_t)p * mask_plane + (size_t)(i0 + lane)];
      mine = any == 0u;
    }
    unsigned long long zero = __ballot(mine);
    while (zero) {
      const int l = __ffsll((long long)zero) - 1;
      zero &= zero - 1ull;
      unsigned char* row = static_cast<unsigned char*>(feat) + (first + i0 + l) * (int64_t)row_bytes;
      if ((row_bytes & 15) == 0) {
        for (int o = lane * 16; o < row_bytes; o += 64 * 16) *reinterpret_cast<clr_v4u*>(row + o) = z4;
      } else {
        for (int o = lane * ESZ; o < row_bytes; o += 64 * ESZ) {
          if (ESZ == 4) *reinterpret_cast<uint32_t*>(row + o) = 0u;
          else *reinterpret_cast<uint16_t*>(row + o) = 0;
        }
      }
    }
  }
}

}  // namespace

// (declared in saf_fuse_dev.h: the windowed path launches it beside its last row kernel)
int clear_rows(void* feat, const int* weight, int64_t first, int64_t n_rows, int esz, int row_bytes, const uint32_t* masks,
               size_t mask_plane, int n_planes, hipStream_t s) {
  if (n_rows <= 0) return SAF_OK;
  // Beside a row kernel (masks given) half the grid.  Measured (profiles/r05/clear_beside.txt): the two kernels do run side by side
  // -- with 2 or 4 workgroups per CU the row kernel beside the clear takes 5 ms longer on the coherent scene, which is what the
  // clear takes alone: the zeros leave through the same per-CU address / store path the row kernel is bound by -- so folding the
  // clear into the call is worth 0.2-0.4 ms of 42.7 there and nothing on depth A; 8 per CU was the best of 2 / 4 / 8 / 16.
  const char* bpc_env = getenv("SAF_CLEAR_WGS");
  const int per_cu = bpc_env && atoi(bpc_env) > 0 ? atoi(bpc_env) : (masks ? 8 : 16);
  if ((uintptr_t)feat & 15) return fail(SAF_E_INVALID, "clear_unwritten_rows: clip_feat must be 16-byte aligned");
  int64_t blocks = (n_rows + 255) / 256;
  const int64_t cap = (int64_t)device_cus() * per_cu;
  if (blocks > cap) blocks = cap;
  if (esz == 4)
    hipLaunchKernelGGL(clear_unwritten_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, feat, weight, first, n_rows, row_bytes, masks,
                       mask_plane, n_planes);
  else
    hipLaunchKernelGGL(clear_unwritten_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, feat, weight, first, n_rows, row_bytes, masks,
                       mask_plane, n_planes);
  return check_launch("clear_unwritten_kernel");
}

namespace {

// ---- the tiled CLIP front-end in one pass (clipfusion.py:789-823) ----
// normalize_img (:783-784), Unfold into overlapping p x p tiles at stride s (:797-804) and the bilinear resize of every
// tile to 224 x 224 (:821-823, align_corners = False) fused: an output pixel reads its four source pixels straight from
// the frame (any strides: channel-last as the loaders yield it, or planar), normalises them and blends --
// ATen's arithmetic: src = scale * (dst + 0.5) - 0.5 clamped at 0, i1 = min(i0 + 1, p - 1),
// out = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11).  The [tiles, 3, 224, 224] batch is written once, in the
// dtype the ViT consumes (fp32, or bf16 / fp16 under autocast) instead of three fp32 round trips through HBM.
struct TileArgs {
  const float* rgb;
  int B, H, W, p, s, npy, npx, out;
  int64_t sb, sc, sy, sx;
  float mean[3], stdv[3], scale;
};
template <int OT>
__global__ __launch_bounds__(256) void clip_tiles_kernel(TileArgs a, void* __restrict__ dst) {
  const int ox = blockIdx.x * 256 + threadIdx.x;  // x fastest: coalesced planar stores
  const int oy = blockIdx.y, tile = blockIdx.z;
  if (ox >= a.out) return;
  const int px = tile % a.npx, py = (tile / a.npx) % a.npy, b = tile / (a.npx * a.npy);
  float fy = a.scale * ((float)oy + 0.5f) - 0.5f, fx = a.scale * ((float)ox + 0.5f) - 0.5f;
  fy = fy < 0.0f ? 0.0f : fy;
  fx = fx < 0.0f ? 0.0f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < a.p - 1 ? 1 : 0), x1 = x0 + (x0 < a.p - 1 ? 1 : 0);
  const float h1 = fy - (float)y0, h0 = 1.0f - h1, w1 = fx - (float)x0, w0 = 1.0f - w1;
  const float* base = a.rgb + (int64_t)b * a.sb + (int64_t)(py * a.s) * a.sy + (int64_t)(px * a.s) * a.sx;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* pc = base + (int64_t)c * a.sc;
    const float v00 = (pc[(int64_t)y0 * a.sy + (int64_t)x0 * a.sx] - a.mean[c]) / a.stdv[c];
    const float v01 = (pc[(int64_t)y0 * a.sy + (int64_t)x1 * a.sx] - a.mean[c]) / a.stdv[c];
    const float v10 = (pc[(int64_t)y1 * a.sy + (int64_t)x0 * a.sx] - a.mean[c]) / a.stdv[c];
    const float v11 = (pc[(int64_t)y1 * a.sy + (int64_t)x1 * a.sx] - a.mean[c]) / a.stdv[c];
    const float v = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
    const int64_t o = (((int64_t)tile * 3 + c) * a.out + oy) * a.out + ox;
    if (OT == SAF_F32) {
      static_cast<float*>(dst)[o] = v;
    } else if (OT == SAF_BF16) {
      static_cast<uint16_t*>(dst)[o] = (uint16_t)f32_to_bf16_bits(v);
    } else {
      const _Float16 hv = (_Float16)v;
      static_cast<uint16_t*>(dst)[o] = __builtin_bit_cast(uint16_t, hv);
    }
  }
}

}  // namespace
}  // namespace saf

using namespace saf;

extern "C" {

int saf_backproject_lattice(const float* depth, int32_t height, int32_t width, const float* pose, const float* Kinv,
                            const int32_t* u_idx, int32_t nu, const int32_t* v_idx, int32_t nv, float max_depth,
                            float* xyz, uint8_t* valid, void* stream) {
  if (!depth || !pose || !Kinv || !u_idx || !v_idx || !xyz || !valid || height <= 0 || width <= 0 || nu <= 0 || nv <= 0)
    return fail(SAF_E_INVALID, "backproject: bad arguments");
  const int total = nu * nv;
  hipLaunchKernelGGL(backproject_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     depth, width, pose, Kinv, u_idx, nu, v_idx, nv, max_depth, xyz, valid);
  return check_launch("backproject_kernel");
}

int saf_merge_finalize(const saf_volume* vol, int64_t first_voxel, int64_t count, void* stream) {
  return scale_volume<true>(vol, first_voxel, count, static_cast<hipStream_t>(stream));
}

int saf_mean_to_sum(const saf_volume* vol, int64_t first_voxel, int64_t count, void* stream) {
  return scale_volume<false>(vol, first_voxel, count, static_cast<hipStream_t>(stream));
}

int saf_sample_vertices(const saf_volume* vol, const float* verts_index, int64_t n_verts, float* out_feat,
                        float* out_rgb, const int32_t* obj_idx, float* out_obj, const float* seg_color, float* out_seg,
                        void* stream) {
  if (!vol || !vol->clip_feat || !vol->rgb || !verts_index || !out_feat || !out_rgb || n_verts < 0 ||
      (obj_idx && !out_obj) || (seg_color && !out_seg))
    return fail(SAF_E_INVALID, "sample_vertices: bad arguments");
  if (vol->feat_dtype != SAF_F32 && vol->feat_dtype != SAF_BF16)
    return fail(SAF_E_UNSUPPORTED, "sample_vertices: feature dtype %d", vol->feat_dtype);
  if (n_verts == 0) return SAF_OK;
  int64_t blocks = (n_verts + 3) / 4;
  const int64_t cap = (int64_t)device_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vol->feat_dtype == SAF_BF16)
    hipLaunchKernelGGL(sample_vertices_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, vol->nx, vol->ny, vol->nz,
                       vol->feat_dim, vol->clip_feat, vol->rgb, verts_index, n_verts, out_feat, out_rgb, obj_idx, out_obj,
                       seg_color, out_seg);
  else
    hipLaunchKernelGGL(sample_vertices_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, vol->nx, vol->ny, vol->nz,
                       vol->feat_dim, vol->clip_feat, vol->rgb, verts_index, n_verts, out_feat, out_rgb, obj_idx, out_obj,
                       seg_color, out_seg);
  return check_launch("sample_vertices_kernel");
}

int saf_label_argmax(const int32_t* labels_one_hot, int64_t n_vox, int32_t n_classes, int32_t* out, void* stream) {
  if (!labels_one_hot || !out || n_vox < 0 || n_classes <= 0) return fail(SAF_E_INVALID, "label_argmax: bad arguments");
  if (n_vox == 0) return SAF_OK;
  int64_t blocks = (n_vox + 3) / 4;
  const int64_t cap = (int64_t)device_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(label_argmax_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     labels_one_hot, n_vox, n_classes, out);
  return check_launch("label_argmax_kernel");
}

int saf_clip_tiles(const float* rgb, int32_t batch, int32_t height, int32_t width, int64_t stride_b, int64_t stride_c,
                   int64_t stride_y, int64_t stride_x, int32_t patch, int32_t stride, int32_t out_size,
                   const float* mean3, const float* std3, void* out, int32_t out_dtype, void* stream) {
  if (!rgb || !out || !mean3 || !std3 || batch <= 0 || height <= 0 || width <= 0 || patch <= 0 || stride <= 0 || out_size <= 0)
    return fail(SAF_E_INVALID, "clip_tiles: bad arguments");
  if (patch > height || patch > width || (height - patch) % stride != 0 || (width - patch) % stride != 0)
    return fail(SAF_E_INVALID, "clip_tiles: (H - patch) and (W - patch) must be non-negative multiples of the stride "
                               "(the reference asserts the same, clipfusion.py:792-793)");
  TileArgs a;
  a.rgb = rgb; a.B = batch; a.H = height; a.W = width; a.p = patch; a.s = stride; a.out = out_size;
  a.npy = 1 + (height - patch) / stride;
  a.npx = 1 + (width - patch) / stride;
  a.sb = stride_b; a.sc = stride_c; a.sy = stride_y; a.sx = stride_x;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }
  a.scale = (float)patch / (float)out_size;  // ATen area_pixel_compute_scale, align_corners = false
  const int64_t tiles = (int64_t)batch * a.npy * a.npx;
  if (tiles > 65535 || out_size > 65535) return fail(SAF_E_UNSUPPORTED, "clip_tiles: more than 65535 tiles in one call");
  const dim3 grid((out_size + 255) / 256, out_size, (unsigned)tiles);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (out_dtype) {
    case SAF_F32: hipLaunchKernelGGL(clip_tiles_kernel<SAF_F32>, grid, dim3(256), 0, s, a, out); break;
    case SAF_BF16: hipLaunchKernelGGL(clip_tiles_kernel<SAF_BF16>, grid, dim3(256), 0, s, a, out); break;
    case SAF_F16: hipLaunchKernelGGL(clip_tiles_kernel<SAF_F16>, grid, dim3(256), 0, s, a, out); break;
    default: return fail(SAF_E_INVALID, "clip_tiles: bad out_dtype %d", out_dtype);
  }
  return check_launch("clip_tiles_kernel");
}

int saf_clear_unwritten_rows(const saf_volume* vol, int64_t first_voxel, int64_t n_rows, void* stream) {
  if (!vol || !vol->clip_feat || !vol->weight) return fail(SAF_E_INVALID, "clear_unwritten_rows: volume has a NULL buffer");
  const int64_t N = n_voxels(vol);
  if (first_voxel < 0 || n_rows < 0 || first_voxel + n_rows > N) return fail(SAF_E_INVALID, "clear_unwritten_rows: bad voxel range");
  const int esz = vol->feat_dtype == SAF_F32 ? 4 : 2;
  return clear_rows(vol->clip_feat, vol->weight, first_voxel, n_rows, esz, vol->feat_dim * esz, nullptr, 0, 0, static_cast<hipStream_t>(stream));
}

int saf_stage_frame(const saf_frame* src, int32_t feat_channels, int64_t feat_stride_c, int64_t feat_stride_y,
                    int64_t feat_stride_x, const saf_frame* dst, void* stream) {
  // (depth / rgb / label map may be NULL on BOTH sides: images the caller lends the queue instead of having them copied)
  if (!src || !dst || !src->pose || !src->K || !dst->pose || !dst->K || !src->depth != !dst->depth || !src->rgb != !dst->rgb ||
      src->height <= 0 || src->width <= 0 || (src->label_map && !dst->label_map) ||
      (src->feat_map && (!dst->feat_map || feat_channels <= 0 || src->npy <= 0 || src->npx <= 0)))
    return fail(SAF_E_INVALID, "stage_frame: bad arguments");
  StageArgs a;
  const int hw = src->height * src->width;
  const float* s[6] = {src->depth, src->rgb, src->pose, src->K, src->feat_map, src->label_map};
  float* d[6] = {const_cast<float*>(dst->depth), const_cast<float*>(dst->rgb), const_cast<float*>(dst->pose),
                 const_cast<float*>(dst->K), const_cast<float*>(dst->feat_map), const_cast<float*>(dst->label_map)};
  const int n[6] = {hw, 3 * hw, 16, 9, src->feat_map ? feat_channels * src->npy * src->npx : 0, src->label_map ? hw : 0};
  for (int k = 0; k < 6; ++k) { a.src[k] = s[k]; a.dst[k] = d[k]; a.n[k] = n[k]; }
  a.fs0 = feat_stride_c; a.fs1 = feat_stride_y; a.fs2 = feat_stride_x; a.f1 = src->npy; a.f2 = src->npx;
  a.first_block[0] = 0;
  for (int k = 0; k < 6; ++k) {
    const int per = 256 * 4 * kStageUnroll;
    a.first_block[k + 1] = a.first_block[k] + (s[k] && n[k] > 0 ? (n[k] + per - 1) / per : 0);
  }
  if (a.first_block[6] <= 0) return SAF_OK;
  hipLaunchKernelGGL(stage_frame_kernel, dim3(a.first_block[6]), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch("stage_frame_kernel");
}

}  // extern "C"
