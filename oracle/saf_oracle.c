/*
 * saf_oracle.c -- CPU restatement of the reference's fusion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (spatially_aware_ai_amd/) may import, link
 * or call this file; it exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check / time the HIP path against an independent scalar implementation.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks every function below against
 * golden vectors produced by the reference's own Python code (oracle/gen_golden.py imports
 * /root/reference in the build container; fixtures in tests/golden/).
 *
 * Each function cites the reference lines it restates.  Arithmetic is plain IEEE fp32, one
 * rounding per operation as PyTorch's elementwise kernels do; the two 3x3 products that the
 * reference hands to BLAS (clipfusion.py:648-652) are restated in the accumulation order MKL
 * produces for them single-threaded -- R^T(x-t): rounded products added as (p0+p2)+p1;
 * K@xyz_cam: k-ascending FMA chain -- found by bit-for-bit comparison with torch.bmm and
 * confirmed on 577k valid / 2.7M tsdf decisions of the config-1 golden (zero differences).
 * Build with -ffp-contract=off so the compiler adds no contractions of its own.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/saf.h"

#ifndef SAF_DOT_VARIANT
#define SAF_DOT_VARIANT 0
#endif

/* First 3x3 product, R^T (x - t): both operands reach BLAS as transposed views and MKL's
 * kernel for that layout adds the rounded products as (p0 + p2) + p1, no FMA. */
static inline float dot3_rt(float a0, float a1, float a2, float b0, float b1, float b2) {
#if SAF_DOT_VARIANT == 0
  float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
  return (p0 + p2) + p1;
#else
  float acc = a0 * b0;
  acc = fmaf(a1, b1, acc);
  acc = fmaf(a2, b2, acc);
  return acc;
#endif
}

/* a0*b0 + a1*b1 + a2*b2 as a k-ascending FMA chain (K @ xyz_cam, and the products of
 * backproject_pcd). */
static inline float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
#if SAF_DOT_VARIANT == 0
  float acc = a0 * b0;
  acc = fmaf(a1, b1, acc);
  acc = fmaf(a2, b2, acc);
  return acc;
#elif SAF_DOT_VARIANT == 1
  return (a0 * b0 + a1 * b1) + a2 * b2;
#elif SAF_DOT_VARIANT == 2
  float acc = fmaf(a0, b0, 0.0f);
  acc = fmaf(a1, b1, acc);
  acc = fmaf(a2, b2, acc);
  return acc;
#else
  float acc = a2 * b2;
  acc = fmaf(a1, b1, acc);
  acc = fmaf(a0, b0, acc);
  return acc;
#endif
}

/* bf16 feature volumes (extension; the reference is fp32-only): widening is a shift, narrowing
 * rounds to nearest even with NaN kept quiet -- the same integer arithmetic as the HIP kernels. */
static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float feat_get(const saf_volume* v, int64_t i) {
  return v->feat_dtype == SAF_BF16 ? bf16_to_f32(((const uint16_t*)v->clip_feat)[i]) : ((const float*)v->clip_feat)[i];
}
static inline void feat_set(const saf_volume* v, int64_t i, float x) {
  if (v->feat_dtype == SAF_BF16)
    ((uint16_t*)v->clip_feat)[i] = f32_to_bf16(x);
  else
    ((float*)v->clip_feat)[i] = x;
}

static int g_threads = 1;
void saf_oracle_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int saf_oracle_get_threads(void) { return g_threads; }

/* grid_sample's un-normalisation for align_corners=False (ATen GridSamplerKernel.cpp,
 * ComputeLocationBase<align_corners=false>::unnormalize): (g+1)*(size/2) - 0.5 */
static inline float unnormalize(float g, int size) {
  float sf = (float)size / 2.0f;
  return (g + 1.0f) * sf - 0.5f;
}

typedef struct {
  float gx, gy, z, sdf;
  int in_view; /* _valid */
  int valid, tsdf_valid;
} voxel_class;

/* clipfusion.py:647-679 (= clip_seem_fusion.py:697-728) for one voxel of one frame. */
static inline voxel_class classify(const saf_volume* v, const saf_frame* f, int ix, int iy, int iz) {
  voxel_class c;
  const float* P = f->pose;
  const float* K = f->K;
  /* self.xyz_world[None] - poses[:, None, :3, 3] */
  float dx = v->axis_x[ix] - P[3];
  float dy = v->axis_y[iy] - P[7];
  float dz = v->axis_z[iz] - P[11];
  /* poses[:, :3, :3].transpose(1, 2) @ (...)^T : cam_i = sum_k R[k][i] d_k */
  float cx = dot3_rt(P[0], P[4], P[8], dx, dy, dz);
  float cy = dot3_rt(P[1], P[5], P[9], dx, dy, dz);
  float cz = dot3_rt(P[2], P[6], P[10], dx, dy, dz);
  /* uvz = K @ xyz_cam */
  float u = dot3(K[0], K[1], K[2], cx, cy, cz);
  float w = dot3(K[3], K[4], K[5], cx, cy, cz);
  float z = dot3(K[6], K[7], K[8], cx, cy, cz);
  /* uv = uvz[:, :2] / z ; grid = uv + 0.5 ; grid /= [W,H] ; grid *= 2 ; grid -= 1 */
  float gx = u / z;
  float gy = w / z;
  gx = gx + 0.5f;
  gy = gy + 0.5f;
  gx = gx / (float)f->width;
  gy = gy / (float)f->height;
  gx = gx * 2.0f;
  gy = gy * 2.0f;
  gx = gx - 1.0f;
  gy = gy - 1.0f;
  /* depth = grid_sample(depth, grid, nearest, align_corners=False), zeros padding */
  float fx = unnormalize(gx, f->width);
  float fy = unnormalize(gy, f->height);
  float xn = nearbyintf(fx);
  float yn = nearbyintf(fy);
  float depth = 0.0f;
  if (xn > -1.0f && xn < (float)f->width && yn > -1.0f && yn < (float)f->height)
    depth = f->depth[(int64_t)yn * f->width + (int64_t)xn];
  float sdf = (depth - z) / v->trunc;
  c.in_view = (fabsf(gx) <= 1.0f) && (fabsf(gy) <= 1.0f) && (z > 0.0f);
  c.valid = c.in_view && (fabsf(sdf) <= 1.0f);
  c.tsdf_valid = c.in_view && (sdf > -1.0f);
  c.gx = gx;
  c.gy = gy;
  c.z = z;
  c.sdf = sdf;
  return c;
}

/* zero-padded fetch of channel-last / channel-first images */
static inline float img_at(const float* img, int h, int w, int y, int x, int64_t sy, int64_t sx) {
  if (x < 0 || x >= w || y < 0 || y >= h) return 0.0f;
  return img[(int64_t)y * sy + (int64_t)x * sx];
}

typedef struct {
  int x0, y0;
  float nw, ne, sw, se;
} bilin;

/* ApplyGridSample<..., GridSamplerInterpolation::Bilinear, zeros> weights */
static inline bilin bilinear_setup(float gx, float gy, int w, int h) {
  bilin b;
  float x = unnormalize(gx, w);
  float y = unnormalize(gy, h);
  float xw = floorf(x), yn = floorf(y);
  float wx = x - xw, ex = 1.0f - wx;
  float ny = y - yn, sy = 1.0f - ny;
  b.nw = sy * ex;
  b.ne = sy * wx;
  b.sw = ny * ex;
  b.se = ny * wx;
  b.x0 = (int)xw;
  b.y0 = (int)yn;
  return b;
}

static inline float bilinear_fetch(const float* img, int h, int w, int64_t sy, int64_t sx, const bilin* b) {
  float nw = img_at(img, h, w, b->y0, b->x0, sy, sx);
  float ne = img_at(img, h, w, b->y0, b->x0 + 1, sy, sx);
  float sw = img_at(img, h, w, b->y0 + 1, b->x0, sy, sx);
  float se = img_at(img, h, w, b->y0 + 1, b->x0 + 1, sy, sx);
  return ((nw * b->nw + ne * b->ne) + sw * b->sw) + se * b->se;
}

static inline float nearest_fetch(const float* img, int h, int w, int64_t sy, int64_t sx, float gx, float gy) {
  float xn = nearbyintf(unnormalize(gx, w));
  float yn = nearbyintf(unnormalize(gy, h));
  if (xn > -1.0f && xn < (float)w && yn > -1.0f && yn < (float)h)
    return img[(int64_t)yn * sy + (int64_t)xn * sx];
  return 0.0f;
}

/*
 * One frame, batch element i of integrate(): clipfusion.py:647-721 / clip_seem_fusion.py:697-822.
 * HOST pointers everywhere.  stats (host, may be NULL) as in saf.h.
 */
int saf_oracle_fuse_frame(const saf_volume* v, const saf_frame* f, uint64_t* stats) {
  if (!v || !f || (v->feat_dtype != SAF_F32 && v->feat_dtype != SAF_BF16)) return SAF_E_INVALID;
  const int D = v->feat_dim;
  const int P = f->npy * f->npx;
  uint64_t nv = 0, nt = 0, dropped = 0;
  /* voxels are independent within a frame: x-slabs are spread over the host threads (only the
   * cpu_baseline leg of bench.py raises the thread count above 1) */
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : nv, nt, dropped) num_threads(g_threads)
  for (int ix = 0; ix < v->nx; ++ix)
    for (int iy = 0; iy < v->ny; ++iy)
      for (int iz = 0; iz < v->nz; ++iz) {
        const int64_t n = ((int64_t)ix * v->ny + iy) * v->nz + iz;
        voxel_class c = classify(v, f, ix, iy, iz);
        if (c.tsdf_valid) {
          /* clipfusion.py:681-695 with B=1 */
          float t = c.sdf < -1.0f ? -1.0f : (c.sdf > 1.0f ? 1.0f : c.sdf);
          int32_t w0 = v->tsdf_weight[n], w1 = w0 + 1;
          if (v->accum_mode == SAF_SUM) {
            v->tsdf[n] = v->tsdf[n] + t;
          } else {
            float a = (float)w1;
            float b = (float)w0 / (float)w1;
            v->tsdf[n] = t / a + v->tsdf[n] * b;
          }
          v->tsdf_weight[n] = w1;
          ++nt;
        }
        if (!c.valid) continue;
        ++nv;
        /* clipfusion.py:715-717 */
        int32_t w0 = v->weight[n], w1 = w0 + 1;
        float a = 1.0f / (float)w1;
        float b = (float)w0 * a;
        /* rgb: nearest (clipfusion.py:701-706) or bilinear (clip_seem_fusion.py:793-798) */
        bilin bi = bilinear_setup(c.gx, c.gy, f->width, f->height);
        for (int ch = 0; ch < 3; ++ch) {
          float s = f->rgb_bilinear
                        ? bilinear_fetch(f->rgb + ch, f->height, f->width, (int64_t)f->width * 3, 3, &bi)
                        : nearest_fetch(f->rgb + ch, f->height, f->width, (int64_t)f->width * 3, 3, c.gx, c.gy);
          float* dst = &v->rgb[n * 3 + ch];
          *dst = (v->accum_mode == SAF_SUM) ? (*dst + s) : (s * a + *dst * b);
        }
        /* clip features: bilinear from the low-res map (clipfusion.py:708-713, :720) */
        bilin bf = bilinear_setup(c.gx, c.gy, f->npx, f->npy);
        for (int ch = 0; ch < D; ++ch) {
          float s = bilinear_fetch(f->feat_map + (int64_t)ch * P, f->npy, f->npx, f->npx, 1, &bf);
          float old = feat_get(v, n * D + ch);
          feat_set(v, n * D + ch, (v->accum_mode == SAF_SUM) ? (old + s) : (s * a + old * b));
        }
        v->weight[n] = w1;
        /* label histogram (clip_seem_fusion.py:786-791, :820-822) */
        if (f->label_map && v->labels_one_hot && v->n_classes > 0) {
          float lf = nearest_fetch(f->label_map, f->height, f->width, f->width, 1, c.gx, c.gy);
          int64_t l = (int64_t)lf; /* .to(torch.long) truncates */
          if (l >= 0 && l < v->n_classes)
            v->labels_one_hot[n * v->n_classes + l] += 1;
          else
            ++dropped;
        }
      }
  if (stats) {
    stats[0] += nv;
    stats[1] += nt;
    stats[2] += 1;
    stats[3] += dropped;
  }
  return SAF_OK;
}

/* Debug/test helper: per-voxel masks of one frame. bit0 = valid, bit1 = tsdf_valid, bit2 = _valid. */
int saf_oracle_classify(const saf_volume* v, const saf_frame* f, uint8_t* mask, float* grid_xy) {
  for (int ix = 0; ix < v->nx; ++ix)
    for (int iy = 0; iy < v->ny; ++iy)
      for (int iz = 0; iz < v->nz; ++iz) {
        const int64_t n = ((int64_t)ix * v->ny + iy) * v->nz + iz;
        voxel_class c = classify(v, f, ix, iy, iz);
        mask[n] = (uint8_t)(c.valid | (c.tsdf_valid << 1) | (c.in_view << 2));
        if (grid_xy) {
          grid_xy[2 * n] = c.gx;
          grid_xy[2 * n + 1] = c.gy;
        }
      }
  return SAF_OK;
}

/*
 * backproject_pcd's per-frame body (clipfusion.py:541-565): rays K^-1 [u,v,1]^T
 * (get_pix_vecs :497-507), xyz_cam = ray*depth, world = R xyz_cam + t.
 * `K^-1 @ uv^T` and `R @ xyz_cam^T` go through BLAS in the reference; restated as dot3.
 */
int saf_oracle_backproject_lattice(const float* depth, int32_t height, int32_t width, const float* pose,
                                   const float* Kinv, const int32_t* u_idx, int32_t nu,
                                   const int32_t* v_idx, int32_t nv, float max_depth, float* xyz,
                                   uint8_t* valid) {
  for (int j = 0; j < nv; ++j)
    for (int i = 0; i < nu; ++i) {
      int u = u_idx[i], vv = v_idx[j];
      int64_t o = (int64_t)j * nu + i;
      float d = depth[(int64_t)vv * width + u];
      float fu = (float)u, fv = (float)vv;
      float rx = dot3(Kinv[0], Kinv[1], Kinv[2], fu, fv, 1.0f);
      float ry = dot3(Kinv[3], Kinv[4], Kinv[5], fu, fv, 1.0f);
      float rz = dot3(Kinv[6], Kinv[7], Kinv[8], fu, fv, 1.0f);
      float cx = rx * d, cy = ry * d, cz = rz * d;
      xyz[o * 3 + 0] = dot3(pose[0], pose[1], pose[2], cx, cy, cz) + pose[3];
      xyz[o * 3 + 1] = dot3(pose[4], pose[5], pose[6], cx, cy, cz) + pose[7];
      xyz[o * 3 + 2] = dot3(pose[8], pose[9], pose[10], cx, cy, cz) + pose[11];
      valid[o] = (uint8_t)(!isnan(d) && d > 0.0f && d < max_depth);
    }
  return SAF_OK;
}

/*
 * Text-query scan (Clip.run_query clipfusion.py:899-904; Clip.clip_feature_surgery :906-934;
 * row normalisation + nan_to_num clip_seem_fusion.py:507-511).  Dot products accumulate in
 * double and round once: the reference's BLAS order is unspecified and the result is compared
 * at 1e-4 relative, not bitwise.
 */
int saf_oracle_query_scan(const float* feats, int64_t n_rows, int64_t feat_stride, int32_t D,
                          const float* text, int32_t n_text, int64_t text_stride, int32_t epilogue,
                          float scale, int32_t normalize, float* out, float* out_last) {
  double* s = (double*)malloc(sizeof(double) * (size_t)n_text);
  double* wt = (double*)malloc(sizeof(double) * (size_t)n_text);
  float* row = (float*)malloc(sizeof(float) * (size_t)D);
  if (!s || !wt || !row) return SAF_E_INVALID;
  for (int64_t n = -1; n < n_rows; ++n) {
    /* pass n == -1 computes the surgery weights from row 0 (clipfusion.py:913-915) */
    if (n == -1 && epilogue != SAF_Q_SURGERY) continue;
    int64_t r = n < 0 ? 0 : n;
    const float* f = feats + r * feat_stride;
    if (normalize) {
      double nn = 0;
      for (int c = 0; c < D; ++c) nn += (double)f[c] * f[c];
      float norm = (float)sqrt(nn);
      /* normalize == 2: feat_norm.clamp_min_(0.1), eval_scannet_segmentation.py:549-551, hypersim_eval.py:50-51 */
      if (normalize == 2 && norm < 0.1f) norm = 0.1f;
      for (int c = 0; c < D; ++c) {
        float q = f[c] / norm;
        if (normalize == 1) {
          if (isnan(q)) q = 0.0f;
          if (isinf(q)) q = q > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
        }
        row[c] = q;
      }
    } else {
      memcpy(row, f, sizeof(float) * (size_t)D);
    }
    for (int l = 0; l < n_text; ++l) {
      double acc = 0;
      const float* t = text + (int64_t)l * text_stride;
      for (int c = 0; c < D; ++c) acc += (double)row[c] * t[c];
      s[l] = (double)(float)acc;
    }
    if (n == -1) {
      /* prob = softmax(2*S0) ; w = prob / mean(prob) */
      double m = -INFINITY, sum = 0;
      for (int l = 0; l < n_text; ++l) m = fmax(m, 2.0 * s[l]);
      for (int l = 0; l < n_text; ++l) {
        wt[l] = exp(2.0 * s[l] - m);
        sum += wt[l];
      }
      double mean = 0;
      for (int l = 0; l < n_text; ++l) {
        wt[l] /= sum;
        mean += wt[l];
      }
      mean /= n_text;
      for (int l = 0; l < n_text; ++l) wt[l] /= mean;
      continue;
    }
    float* o = out ? out + n * n_text : NULL;
    double last = 0;
    if (epilogue == SAF_Q_SCORES) {
      for (int l = 0; l < n_text; ++l) {
        double val = scale * s[l];
        if (o) o[l] = (float)val;
        last = val;
      }
    } else if (epilogue == SAF_Q_SOFTMAX) {
      double m = -INFINITY, sum = 0;
      for (int l = 0; l < n_text; ++l) m = fmax(m, (double)scale * s[l]);
      for (int l = 0; l < n_text; ++l) sum += exp((double)scale * s[l] - m);
      for (int l = 0; l < n_text; ++l) {
        double val = exp((double)scale * s[l] - m) / sum;
        if (o) o[l] = (float)val;
        last = val;
      }
    } else {
      double mean = 0;
      for (int l = 0; l < n_text; ++l) mean += s[l] * wt[l];
      mean /= n_text;
      for (int l = 0; l < n_text; ++l) {
        double val = s[l] * wt[l] - mean;
        if (o) o[l] = (float)val;
        last = val;
      }
    }
    if (out_last) out_last[n] = (float)last;
  }
  free(s);
  free(wt);
  free(row);
  return SAF_OK;
}

/* sums -> means after the cross-rank reduction; the reference has no such step,
 * it is the identity mean = (sum of samples) / (number of samples). */
int saf_oracle_merge_finalize(const saf_volume* v, int64_t first, int64_t count) {
  float* feat = (float*)v->clip_feat;
  for (int64_t n = first; n < first + count; ++n) {
    int32_t w = v->weight[n];
    if (w > 0) {
      float fw = (float)w;
      for (int c = 0; c < v->feat_dim; ++c) feat[n * v->feat_dim + c] /= fw;
      for (int c = 0; c < 3; ++c) v->rgb[n * 3 + c] /= fw;
    }
    int32_t wt = v->tsdf_weight[n];
    if (wt > 0) v->tsdf[n] /= (float)wt;
  }
  return SAF_OK;
}

int saf_oracle_mean_to_sum(const saf_volume* v, int64_t first, int64_t count) {
  float* feat = (float*)v->clip_feat;
  for (int64_t n = first; n < first + count; ++n) {
    float fw = (float)v->weight[n];
    for (int c = 0; c < v->feat_dim; ++c) feat[n * v->feat_dim + c] *= fw;
    for (int c = 0; c < 3; ++c) v->rgb[n * 3 + c] *= fw;
    v->tsdf[n] *= (float)v->tsdf_weight[n];
  }
  return SAF_OK;
}

/* clip_seem_fusion.py:315-325: argmax with first-max tie-break, all-zero row -> -1 */
int saf_oracle_label_argmax(const int32_t* labels, int64_t n_voxels, int32_t n_classes, int32_t* out) {
  for (int64_t n = 0; n < n_voxels; ++n) {
    const int32_t* r = labels + n * n_classes;
    int best = 0, any = 0;
    for (int c = 0; c < n_classes; ++c) {
      if (r[c] != 0) any = 1;
      if (r[c] > r[best]) best = c;
    }
    out[n] = any ? best : -1;
  }
  return SAF_OK;
}

/*
 * Vertex sampling of extract_mesh (clipfusion.py:741-760; clip_seem_fusion.py:843-878):
 * grid = (verts + 0.5) / nvox * 2 - 1 evaluated as (v + 0.5) * fl(1/n) (numpy array / int32 tensor goes
 * through Tensor.__rtruediv__ = reciprocal * x), axes swapped to (z, y, x), then ATen's scalar 3-D
 * grid sampler, align_corners=False, zeros padding: trilinear for clip_feat / rgb (clamped), nearest for
 * the object index and the segmentation colour (clamped).
 */
typedef struct {
  float f, w0, w1;
  int i0;
} axis3;
static inline axis3 axis_setup(float v, int size) {
  axis3 a;
  float r = 1.0f / (float)size;
  float g = (v + 0.5f) * r;
  g = g * 2.0f;
  g = g - 1.0f;
  a.f = ((g + 1.0f) * (float)size - 1.0f) / 2.0f;
  float fl = floorf(a.f);
  a.i0 = (int)fl;
  a.w0 = (fl + 1.0f) - a.f;
  a.w1 = a.f - fl;
  return a;
}

int saf_oracle_sample_vertices(const saf_volume* v, const float* verts, int64_t n_verts, float* out_feat,
                               float* out_rgb, const int32_t* obj_idx, float* out_obj, const float* seg_color,
                               float* out_seg) {
  const int D = v->feat_dim;
  for (int64_t t = 0; t < n_verts; ++t) {
    axis3 ax = axis_setup(verts[t * 3 + 2], v->nz), ay = axis_setup(verts[t * 3 + 1], v->ny),
          az = axis_setup(verts[t * 3 + 0], v->nx);
    int64_t row[8];
    float w[8];
    for (int c = 0; c < 8; ++c) {
      int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
      int x = ax.i0 + dx, y = ay.i0 + dy, z = az.i0 + dz;
      int in = x >= 0 && x < v->nz && y >= 0 && y < v->ny && z >= 0 && z < v->nx;
      row[c] = in ? ((int64_t)z * v->ny + y) * v->nz + x : -1;
      w[c] = (dx ? ax.w1 : ax.w0) * (dy ? ay.w1 : ay.w0) * (dz ? az.w1 : az.w0);
    }
    for (int ch = 0; ch < D; ++ch) {
      float acc = 0.0f;
      for (int c = 0; c < 8; ++c)
        if (row[c] >= 0) acc += feat_get(v, row[c] * D + ch) * w[c];
      out_feat[t * D + ch] = acc;
    }
    for (int ch = 0; ch < 3; ++ch) {
      float acc = 0.0f;
      for (int c = 0; c < 8; ++c)
        if (row[c] >= 0) acc += v->rgb[row[c] * 3 + ch] * w[c];
      out_rgb[t * 3 + ch] = acc < 0.0f ? 0.0f : (acc > 1.0f ? 1.0f : acc);
    }
    if (obj_idx || seg_color) {
      float xn = nearbyintf(ax.f), yn = nearbyintf(ay.f), zn = nearbyintf(az.f);
      int in = xn >= 0.0f && xn < (float)v->nz && yn >= 0.0f && yn < (float)v->ny && zn >= 0.0f && zn < (float)v->nx;
      int64_t n = in ? ((int64_t)zn * v->ny + (int64_t)yn) * v->nz + (int64_t)xn : -1;
      if (obj_idx) out_obj[t] = n >= 0 ? (float)obj_idx[n] : 0.0f;
      if (seg_color)
        for (int ch = 0; ch < 3; ++ch) {
          float s = n >= 0 ? seg_color[n * 3 + ch] : 0.0f;
          out_seg[t * 3 + ch] = s < 0.0f ? 0.0f : (s > 1.0f ? 1.0f : s);
        }
    }
  }
  return SAF_OK;
}

/* ---- flood_fill_3d's object discovery (handy_utils.py:295-480) ----
 * Restated as the reference walks: raster order over (x, y, z) (:368-370); visited voxels are skipped
 * (:375-377); the null class and empty voxels start no object (:383); a depth-first flood fill over the 26
 * neighbours collects the voxels of the same class (:312-344); objects of fewer than `min_voxels` voxels
 * are dropped (:390-392); the k-th accepted object gets voxel_obj_ids = -2 - k (:352-353, :447-450, no
 * trained in-situ model). */
int saf_oracle_label_components(const int32_t* labels, int32_t nx, int32_t ny, int32_t nz, int32_t null_class,
                                int32_t min_voxels, int32_t* obj_ids, int32_t* n_objects, int32_t max_objects,
                                int32_t* obj_first, int32_t* obj_class, int32_t* obj_count) {
  const int64_t n = (int64_t)nx * ny * nz;
  uint8_t* visited = (uint8_t*)calloc((size_t)n, 1); /* the reference's visited_voxels */
  uint8_t* seen = (uint8_t*)calloc((size_t)n, 1);    /* flood_fill's own `visited` set, cleared per fill */
  int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n * 26 + 1));
  int32_t* object_voxels = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* touched = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  if (!visited || !seen || !stack || !object_voxels || !touched) return SAF_E_INVALID;
  for (int64_t i = 0; i < n; ++i) obj_ids[i] = -1;
  int32_t k = 0;
  for (int32_t x = 0; x < nx; ++x)
    for (int32_t y = 0; y < ny; ++y)
      for (int32_t z = 0; z < nz; ++z) {
        const int32_t i = (x * ny + y) * nz + z;
        const int32_t class_id = labels[i];
        if (visited[i]) continue;
        visited[i] = 1;
        if (class_id == null_class || class_id == -1) continue;
        /* flood_fill((x, y, z), class_id) */
        int64_t sp = 0, n_obj = 0, n_touched = 0;
        stack[sp++] = i;
        while (sp) {
          const int32_t cur = stack[--sp];
          if (seen[cur]) continue;
          seen[cur] = 1;
          touched[n_touched++] = cur;
          if (labels[cur] != class_id) continue;
          object_voxels[n_obj++] = cur;
          const int32_t cz = cur % nz, cy = (cur / nz) % ny, cx = cur / (nz * ny);
          for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy)
              for (int dz = -1; dz <= 1; ++dz) {
                if (!dx && !dy && !dz) continue;
                const int32_t xx = cx + dx, yy = cy + dy, zz = cz + dz;
                if (xx < 0 || xx >= nx || yy < 0 || yy >= ny || zz < 0 || zz >= nz) continue;
                const int32_t j = (xx * ny + yy) * nz + zz;
                if (!seen[j]) stack[sp++] = j; /* the reference pushes visited ones too and skips them on pop */
              }
        }
        for (int64_t t = 0; t < n_touched; ++t) seen[touched[t]] = 0;
        for (int64_t t = 0; t < n_obj; ++t) visited[object_voxels[t]] = 1;
        if (n_obj < min_voxels) continue;
        for (int64_t t = 0; t < n_obj; ++t) obj_ids[object_voxels[t]] = -2 - k;
        if (k < max_objects) {
          if (obj_first) obj_first[k] = i;
          if (obj_class) obj_class[k] = class_id;
          if (obj_count) obj_count[k] = (int32_t)n_obj;
        }
        ++k;
      }
  *n_objects = k;
  free(visited); free(seen); free(stack); free(object_voxels); free(touched);
  return SAF_OK;
}
