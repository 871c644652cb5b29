/*
 * saf.h -- C ABI of the MI355X-native fusion hot path ("saf" = spatially-aware fusion).
 *
 * The reference (cy-xu/spatially_aware_AI) has no FFI layer: its boundary for this path is the
 * Python class API.  This header is the drop-in boundary *underneath* that API:
 * plain pointers and sizes, no torch types.  Every entry point names the reference code it
 * replaces.  The Python host classes in spatially_aware_ai_amd/ bind these through ctypes
 * (see INTEGRATION.md for the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - all data pointers are DEVICE pointers (HIP, gfx950) unless a parameter says "host";
 *   - nothing here owns volume memory: the caller (torch) allocates and keeps it alive;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and re-entrant.  Process-wide state is limited to: the thread-local error string, a
 *     per-device pool of the auxiliary stream / events of the per-frame pipeline, and a per-device
 *     asynchronous error latch (saf_poll_async_error);
 *   - return value: 0 = ok, <0 = error (SAF_E_*), text via saf_last_error();
 *   - voxel flat index n = (x*ny + y)*nz + z, the C-order flattening of meshgrid(ij)
 *     (clipfusion.py:617-622); 64-bit offsets are used wherever n*D can exceed 2^31.
 *
 * oracle/saf_oracle.c implements CPU twins (saf_oracle_*) of the compute entry points with the
 * same structs and HOST pointers; they are test infrastructure only.
 */
#ifndef SAF_H
#define SAF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAF_ABI_VERSION 3

enum saf_status {
  SAF_OK = 0,
  SAF_E_INVALID = -1,   /* bad argument (NULL pointer, non-positive size, misaligned buffer) */
  SAF_E_WORKSPACE = -2, /* workspace too small */
  SAF_E_HIP = -3,       /* HIP runtime error (launch failure, ...) */
  SAF_E_UNSUPPORTED = -4
};

enum saf_dtype { SAF_F32 = 0, SAF_BF16 = 1, SAF_F16 = 2 };

/* How per-voxel features are folded in (clipfusion.py:715-721). */
enum saf_accum_mode {
  SAF_RUNNING_MEAN = 0, /* reference semantics: x <- s*(1/w') + x*(w*(1/w')) in that op order */
  SAF_SUM = 1           /* x <- x + s ; used by the frame-sharded multi-GPU path, finalised by
                           saf_merge_finalize after the RCCL reduction */
};

/*
 * The dense fusion volume: the registered buffers of ClipFusion (clipfusion.py:605-625) and
 * ClipSeemFusion (clip_seem_fusion.py:640-672).  `xyz_world` is replaced by three per-axis
 * coordinate tables (xyz_world[n] = (axis_x[x], axis_y[y], axis_z[z]) for any elementwise
 * construction from meshgrid(ij)), so the sweep never reads the 12*N-byte buffer.
 */
typedef struct saf_volume {
  int32_t nx, ny, nz;
  int32_t feat_dim;       /* D = n_clip_feats */
  int32_t n_classes;      /* width of labels_one_hot (143) or 0 when there is no label histogram */
  int32_t feat_dtype;     /* saf_dtype of clip_feat; SAF_F32 is the reference layout */
  int32_t accum_mode;     /* saf_accum_mode */
  float trunc;            /* truncation distance in metres (self.trunc) */
  const float* axis_x;    /* [nx] */
  const float* axis_y;    /* [ny] */
  const float* axis_z;    /* [nz] */
  float* tsdf;            /* [N]       f32 */
  int32_t* tsdf_weight;   /* [N]       i32 */
  int32_t* weight;        /* [N]       i32 */
  float* rgb;             /* [N,3]     f32 */
  void* clip_feat;        /* [N,D]     feat_dtype */
  int32_t* labels_one_hot;/* [N,n_classes] i32 or NULL */
} saf_volume;

/*
 * One posed RGB-D frame plus the backbone outputs that integrate() consumes
 * (clipfusion.py:627-645, clip_seem_fusion.py:676-695, :755-760).
 */
typedef struct saf_frame {
  int32_t height, width;
  const float* depth;     /* [H,W]   metres; 0 = missing */
  const float* rgb;       /* [H,W,3] 0..1, channel-last exactly as the loaders yield it */
  const float* pose;      /* [4,4]   camera->world, row-major (device memory) */
  const float* K;         /* [3,3]   intrinsics, row-major (device memory) */
  const float* feat_map;  /* [Dm,npy,npx] f32, Dm >= D: Clip.img_inference_tiled output for this frame;
                             only the first D channels are used (clipfusion.py:709) */
  int32_t npy, npx;
  const float* label_map; /* [H,W] f32 class ids = pano_seg.float() (clip_seem_fusion.py:760) or NULL */
  int32_t rgb_bilinear;   /* 0: nearest (clipfusion.py:701-706); 1: bilinear (clip_seem_fusion.py:793-798) */
} saf_frame;

/* Counters the fuse kernels add to (device memory, SAF_STATS_WORDS x u64 -- 16 since ABI version 3, 8 before --, caller zeroes
 * them when it wants):
 *  [0] sum of Nv (valid voxels)  [1] sum of Nt (tsdf-valid voxels)  [2] frames fused
 *  [3] labels outside [0,n_classes) that were dropped (the reference raises instead)
 *  [4] fuse workgroups that gave up waiting for their frame's sweep (must stay 0)
 *  [5] feature rows read-modify-written by window kernels (= sum over windows of |union of valid sets|;
 *      0 when the per-frame pipeline ran)  [6] voxels whose TSDF a window's classification updated
 *      (sum over windows of |union of tsdf-valid sets|)  [7] disagreements between the classification's guarded pixel path and
 *      the reference's chain, counted only with SAF_CLS_VERIFY=1 in the environment (a self-check; must stay 0)
 *  [8..12] the windowed path's frame cull (a wave tests its 4 x 4 x 16-voxel brick against the 32 frames of a classification
 *      launch before it classifies any voxel; clipfusion.py:647-679 has no counterpart -- it projects every voxel into every
 *      frame): [8] (brick, frame) pairs tested, and of those dropped because the brick lies [9] behind the camera, [10]
 *      farther than the frame's largest depth + trunc, [11] outside the view frustum, [12] more than trunc behind the largest
 *      depth of the pixels it projects onto (occluded).  A dropped pair has no valid and no tsdf-valid voxel; tests assert that
 *      every reason fires on the cameras they fuzz.  [13..15] reserved (0). */
#define SAF_STATS_WORDS 16
/* frames per window of the windowed path of saf_fuse_frames (stats[5] and [6] count per window); SAF_WIN_FRAMES=64 in the
 * environment selects 64-frame windows */
#define SAF_WINDOW_FRAMES 128

const char* saf_last_error(void);
int saf_abi_version(void);

/* Bytes of device scratch saf_fuse_frame(s) needs for a volume of n_voxels and a feature map of
 * feat_dim x npy x npx (compact list of valid voxels + re-laid-out feature map + counters). */
size_t saf_fuse_workspace_bytes(int64_t n_voxels, int32_t feat_dim, int32_t npy, int32_t npx);
/* The same for ONE volume (its grid, width and feature dtype are known): the brick form's segment pools -- 6.5 GB at
 * 256^3 -- are reserved only when that form would run for it (feat_dim a multiple of 64 that the row kernel does not
 * take, or SAF_WIN_FORM=bricks); the default 512-channel f32 / bf16 volumes end at 0.55 GB.  0 for a bad descriptor. */
size_t saf_fuse_workspace_bytes_for(const saf_volume* vol, int32_t npy, int32_t npx);
/* The same, with room for the frames' depth images (height x width) re-laid-out in 4 x 8-pixel tiles, four windows of 128
 * frames of them (0.63 GB at 640 x 480): with such a workspace the windowed path's classification gathers depth from the
 * tiled copies -- half the cache lines per brick and frame --, with a smaller one from the frames'
 * own row-major images; results are identical either way.  New in round 5; nothing to mirror in the reference.
 * Round 6: for a volume that counts labels (ClipSeemFusion) it also holds ONE window of the frames' rgb and label images packed as
 * {r, g, b, label} pixels in 4 x 2-pixel tiles (0.63 GB at 640 x 480): the row kernel then takes a hit's bilinear colour and its
 * class from four 16-byte gathers instead of thirteen 4-byte ones (clip_seem_fusion.py:786-798); identical results either way. */
size_t saf_fuse_workspace_bytes_for_frames(const saf_volume* vol, int32_t npy, int32_t npx, int32_t height, int32_t width);

/*
 * Fuse ONE frame into the volume: replaces the body of ClipFusion.integrate after the CLIP call
 * (clipfusion.py:647-721) / ClipSeemFusion.integrate (clip_seem_fusion.py:697-822) for batch
 * element i: projection + depth test (a2), TSDF running mean (a3), valid-voxel compaction (a4),
 * feature / rgb / label gather (a5), running-mean fuse (a6), label histogram (a7).
 * `stats` may be NULL.
 */
int saf_fuse_frame(const saf_volume* vol, const saf_frame* frame, void* workspace,
                   size_t workspace_bytes, uint64_t* stats, void* stream);

/* The same for n_frames frames in order (host array of descriptors); one host call, no host
 * synchronisation between frames -- the loop of clipfusion.py:1125-1133.
 * Two device paths.  Which voxels are touched, weights, tsdf, tsdf_weight, rgb and label counts are identical bit for bit
 * on both; so are the feature rows with SAF_WIN_FORM=rows:
 *  - per-frame pipeline: one sweep + one fuse kernel per frame (any shape);
 *  - windowed, voxel-major (16 or more frames of one shape, feat_dim a multiple of 64 up to 8192): per window of
 *    SAF_WINDOW_FRAMES frames one classification launch per 32 frames (projection, depth test, TSDF in registers, one
 *    frame-mask word per voxel; any grid) and one row kernel that reads and writes every touched feature row ONCE per
 *    window.  The row kernel's form (environment SAF_WIN_FORM, read per call):
 *      sums   (default where it applies: f32 volume with feat_dim a multiple of 256, bf16 with a multiple of 512, <= 1024) a
 *             row's samples of the window are summed in registers and the row is blended once, (w0 old + sum) / (w0 + k): the
 *             running mean of clipfusion.py:715-721 with the window's k updates folded into one -- feature values within fp32
 *             rounding of frame-after-frame fusion (bf16: one rounding per window instead of one per hit), reproducible
 *             bit for bit from run to run.  For a bf16 volume this form also keeps the window's feature maps in bf16 (rounded
 *             once, to nearest even, when they are re-laid for the taps): exact when the backbone emitted bf16 features
 *             (BASELINE config 3), otherwise one more rounding at the volume's own precision per tap;
 *      rows   the same widths, hits applied one by one in frame order: feature rows bit-identical to the per-frame pipeline;
 *      bricks every other width (and on request): brick-resident rows, map taps shared, fixed-point sums; same contract as sums.
 *    SAF_WINDOW=0 in the environment forces the per-frame pipeline. */
int saf_fuse_frames(const saf_volume* vol, const saf_frame* frames, int32_t n_frames,
                    void* workspace, size_t workspace_bytes, uint64_t* stats, void* stream);

/* Copy one frame's inputs (depth, rgb, pose, K, feature map, label map -- all f32) from `src` into the buffers `dst`
 * points to, in ONE launch: the host queue behind integrate() keeps the frames of small calls in a staging ring until a
 * window is full.  The source feature map is addressed through element strides (Clip.img_inference_tiled returns a
 * permuted view); feat_channels = channels to copy; everything else is contiguous.  src->feat_map may be NULL (the queue
 * computes the feature maps later, in one backbone batch per flush): nothing is copied for it.  depth, rgb and label_map may be
 * NULL in BOTH `src` and `dst`: images the caller lends the queue where they lie (the frame descriptors of the later fuse call
 * point at them) instead of having them copied. */
int saf_stage_frame(const saf_frame* src, int32_t feat_channels, int64_t feat_stride_c, int64_t feat_stride_y,
                    int64_t feat_stride_x, const saf_frame* dst, void* stream);

/* Which device path saf_fuse_frames would take for this call: 1 = windowed (never reads the feature rows of voxels
 * with weight 0), 0 = per-frame pipeline, -1 = invalid arguments.  Host-only, launches nothing. */
int saf_fuse_path(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, size_t workspace_bytes);

/*
 * Streaming session (new in ABI version 3): the loop of clip_seem_fusion.py:303-313 / clipfusion.py:1125-1133 hands over ONE
 * frame per integrate() call; the host queue behind it (spatially_aware_ai_amd/clipfusion.py) stages the frames and pushes them
 * 32 at a time.  A separate saf_fuse_frames call per window exposes the window's whole classification (nothing of the call runs
 * beside it: 3.7 ms at 256^3) and cannot start before the window's last frame has arrived; a session keeps ONE pipeline:
 *   saf_fuse_session_ok      1 if a session takes these frames for this volume and workspace (what the windowed ROW forms take, on
 *                            two streams: SAF_WIN_OVERLAP != 0), 0 if not (use saf_fuse_frames), -1 for bad arguments.
 *   saf_fuse_session_push    classifies the frames at once -- one launch (one mask plane of the open window) per 32 frames, on the
 *                            session's classification stream -- and, when a window (SAF_WINDOW_FRAMES) is complete, launches its row
 *                            kernel on `stream`: the launches of the following pushes run beside it.  Every push but a window's last
 *                            brings a multiple of 32 frames.  Same volume, workspace, counters and frame shapes until finish.
 *                            The frames' device buffers stay untouched until the stream has passed their window's row kernel.
 *                            `ready_event` (a hipEvent_t, may be NULL): recorded by the caller behind whatever produces these
 *                            frames (their staging), on any stream, and behind the previous finish of this session; the
 *                            classification waits for it INSTEAD of for everything queued on `stream` -- where the previous
 *                            window's row kernel sits, the kernel it is meant to run beside.  NULL: it waits for `stream`.
 *                            `tile_stream` (a hipStream_t, may be NULL): the stream the frames were staged on (the one `ready_event`
 *                            was recorded on: with it the event itself is not waited for); the launches' depth tile maxima (two
 *                            small kernels per 32 frames) run there, behind the staging, instead of in the classification chain.  That stream must be ordered behind the row kernel of the window four windows back (the
 *                            tile region holds four windows: the host queue's staging ring has the same period).
 *   saf_fuse_session_prepare (optional) the depth tile maxima of frames that will be pushed NEXT, in order, computed on `stream` (the one
 *                            they were staged on) a call ahead of their push: a `ready_event` recorded behind it has completed by the
 *                            time the frames are pushed, and the push then queues its launch with no cross-stream wait in front.
 *   saf_fuse_session_finish  launches the row kernel of the window that is still open; behind it (in stream order) the volume
 *                            holds every pushed frame, bit for bit as one saf_fuse_frames call over them leaves it.
 *   saf_fuse_session_abandon drops the open window without fusing its rows (the volume is being reset: its classification has
 *                            already updated the TSDF).  Launches nothing.
 *   saf_fuse_session_pending frames of the open window (0: none).
 * `stream`: the same stream for every call of a session (one of the host queue's own, so that the caller's stream can stage
 * later frames meanwhile).  Not thread-safe per session; sessions are independent of each other.
 */
typedef struct saf_fuse_session saf_fuse_session;
saf_fuse_session* saf_fuse_session_create(void);
int saf_fuse_session_ok(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, size_t workspace_bytes);
int saf_fuse_session_push(saf_fuse_session* session, const saf_volume* vol, const saf_frame* frames, int32_t n_frames,
                          void* workspace, size_t workspace_bytes, uint64_t* stats, void* stream, void* ready_event,
                          void* tile_stream);
int saf_fuse_session_prepare(saf_fuse_session* session, const saf_volume* vol, const saf_frame* frames, int32_t n_frames,
                             void* workspace, size_t workspace_bytes, void* stream);
int saf_fuse_session_finish(saf_fuse_session* session, void* stream);
int saf_fuse_session_abandon(saf_fuse_session* session);
int saf_fuse_session_pending(const saf_fuse_session* session);
void saf_fuse_session_destroy(saf_fuse_session* session);

/*
 * The 7 x 7 depthwise convolution of a ConvNeXt block, channels-last (backbone op of BASELINE config 3's panoptic encoder:
 * kMaX-DeepLab's ConvNeXt-L behind KmaxSegmentationModel.run_on_image, handy_utils.py:29-161; in PyTorch
 * nn.Conv2d(C, C, 7, padding=3, groups=C)).  MIOpen runs these through naive_conv; the op is tiny and L2 bound.
 *   x, y    [batch, H, W, C] of `dtype` (SAF_F32 / SAF_BF16 / SAF_F16), C contiguous and a multiple of 8, 16-byte aligned
 *   w_kkc   [7, 7, C] f32: PyTorch's weight [C, 1, 7, 7] permuted once by the host
 *   bias    [C] f32 or NULL
 * fp32 accumulation, one rounding to `dtype` on the way out; zero padding of 3 pixels.
 */
int saf_dwconv7x7_nhwc(const void* x, const float* w_kkc, const float* bias, void* y, int32_t batch, int32_t height,
                       int32_t width, int32_t channels, int32_t dtype, void* stream);

/* Zero the clip_feat rows of voxels [first, first + n) whose weight is 0.  A volume recycled for a new scan may skip the
 * up-front clear of its 4*D*N feature bytes: zero `weight` (and the other small buffers), fuse through the windowed path
 * only, and call this before anything else reads clip_feat (the Python host does all of that behind reset()). */
int saf_clear_unwritten_rows(const saf_volume* vol, int64_t first_voxel, int64_t n_voxels, void* stream);

/* The per-frame pipeline hands sweep(i) to fuse(i) on the device; a fuse workgroup that waited ~2 s without
 * seeing its sweep gives up (stats[4]) and sets a host-visible latch.  The NEXT saf_fuse_frame(s) call on that
 * device -- or this poll, which launches nothing -- returns SAF_E_HIP once (the volume of the earlier call is
 * incomplete; its stats[4] says so for as long as the volume lives) and clears the latch: a stall that has passed
 * does not disable fusion for the rest of the process. */
int saf_poll_async_error(void);

/*
 * Optional per-kernel timing: a pool of HIP event pairs recorded on the launch stream around each
 * kernel of saf_fuse_frames_profiled (class 0 = prep, 1 = sweep / window classification, 2 = fuse /
 * window row kernel).  Recording is
 * asynchronous; saf_profiler_read must be called after the stream has been synchronised.
 * Used by bench.py for the roofline line; the product path passes NULL.
 */
typedef struct saf_profiler saf_profiler;
saf_profiler* saf_profiler_create(int32_t capacity_pairs);
void saf_profiler_destroy(saf_profiler* p);
void saf_profiler_reset(saf_profiler* p);
/* record only every stride-th frame of a saf_fuse_frames_profiled call (event packets between the
 * kernels cost a few microseconds each; 1 = every frame) */
void saf_profiler_set_stride(saf_profiler* p, int32_t stride);
/* total elapsed ms and number of launches recorded for one kernel class; <0 on error */
int saf_profiler_read(saf_profiler* p, int32_t kernel_class, double* total_ms, int64_t* launches);

/* saf_fuse_frames with event pairs recorded into `profiler` (may be NULL = no recording; pairs
 * beyond the pool's capacity are silently not recorded). */
int saf_fuse_frames_profiled(const saf_volume* vol, const saf_frame* frames, int32_t n_frames,
                             void* workspace, size_t workspace_bytes, uint64_t* stats,
                             saf_profiler* profiler, void* stream);

/* saf_fuse_frames_profiled into a recycled volume (see saf_clear_unwritten_rows), then saf_clear_unwritten_rows over all of
 * it, as ONE call: when it has completed (stream order) every voxel whose weight is 0 has a zero row, whatever the rows held before.  The reference builds a
 * new, zeroed module per scan (clip_seem_fusion.py:291-302); this is that scan's fusion with the 4*D*N-byte clear folded in: in
 * the windowed path the zeros are written beside the last window's row kernel (which leaves most of the HBM bandwidth unused)
 * and only where no frame of the call wrote; any other path zeroes the rows first.  profiler may be NULL.  n_frames == 0: the clear
 * alone. */
int saf_fuse_frames_recycled(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, void* workspace,
                             size_t workspace_bytes, uint64_t* stats, saf_profiler* profiler, void* stream);

/* saf_fuse_frames slab by slab (new capability: the frame-sharded multi-GPU job with its merge pipelined behind the fusion,
 * SURVEY 8e): EVERY frame is fused into x-planes [slab_x0[k], slab_x0[k] + slab_nx[k]) of the volume for k = 0 .. n_slabs - 1
 * in turn -- a slab of x-planes is a contiguous range of the flat voxel index and every decision is that of the full volume's
 * voxels (clipfusion.py:617-622: the same axis table), so the slabs together equal saf_fuse_frames bit for bit; a finished
 * slab is never touched again.  slab_done_events (may be NULL, entries may be NULL): hipEvent_t handles, event k is recorded
 * on `stream` behind the last kernel that writes slab k -- the caller's reduce-scatter of that slab waits for it on its own
 * stream.  One call: the first window of slab k + 1 is classified beside the last row kernel of slab k.  Slabs must not
 * overlap; multiples of 16 x-planes keep the fast unit orders.  profiler may be NULL.  recycled != 0: the volume is a recycled
 * one (saf_fuse_frames_recycled) -- the rows of a slab that are still unwritten are zeroed behind the slab's last row kernel,
 * before its event; voxels outside every slab are not touched. */
int saf_fuse_frames_slabs(const saf_volume* vol, const saf_frame* frames, int32_t n_frames, const int32_t* slab_x0,
                          const int32_t* slab_nx, int32_t n_slabs, void* const* slab_done_events, int32_t recycled,
                          void* workspace, size_t workspace_bytes, uint64_t* stats, saf_profiler* profiler, void* stream);

/*
 * Depth un-projection of a lattice of pixels to world points: the per-frame body of
 * backproject_pcd (clipfusion.py:541-565) with get_pix_vecs (:497-507) folded in.
 *   u_idx[nu], v_idx[nv] : pixel columns / rows of the lattice (device i32)
 *   Kinv                 : [3,3] inverse intrinsics (device f32; the host inverts K)
 * Writes xyz[nv*nu,3] (world), valid[nv*nu] (u8: depth not NaN, >0, <max_depth) in lattice order
 * (v-major, as meshgrid(xy).view(-1) orders it).
 */
int saf_backproject_lattice(const float* depth, int32_t height, int32_t width, const float* pose,
                            const float* Kinv, const int32_t* u_idx, int32_t nu,
                            const int32_t* v_idx, int32_t nv, float max_depth, float* xyz,
                            uint8_t* valid, void* stream);

/* Epilogues of the text-query scan. */
enum saf_query_epilogue {
  SAF_Q_SCORES = 0,  /* out[n,l] = scale * <f_n, t_l>                                   */
  SAF_Q_SOFTMAX = 1, /* Clip.run_query, clipfusion.py:899-904: softmax_l(scale*<f_n,t_l>) */
  SAF_Q_SURGERY = 2  /* Clip.clip_feature_surgery (redundant_feats=None), clipfusion.py:911-932:
                        out[n,l] = S[n,l]*w[l] - mean_l(S[n,l]*w[l]), w from row 0 of feats */
};

/* Row normalisation applied to the features before the dot products. */
enum saf_query_normalize {
  SAF_NORM_NONE = 0,
  SAF_NORM_L2 = 1,       /* f / |f|, NaN -> 0 (clip_seem_fusion.py:507-511; query_mesh.py:24-25) */
  SAF_NORM_L2_CLAMP = 2  /* f / max(|f|, 0.1) (eval_scannet_segmentation.py:549-551, hypersim_eval.py:50-51) */
};

/*
 * Scan n_rows feature rows against n_text text embeddings.
 *   feats      [n_rows, feat_stride] feat_dtype (row-major; first D columns used)
 *   text       [n_text, text_stride] f32 (first D columns used: run_query truncates, :901)
 *   normalize  a saf_query_normalize: SAF_NORM_L2 divides each row by its L2 norm first and maps NaN -> 0
 *              (clip_seem_fusion.py:507-511; query_mesh.py:24-25 without the nan_to_num); SAF_NORM_L2_CLAMP
 *              divides by max(norm, 0.1) as the eval scripts do
 *   out        [n_rows, n_text] f32
 *   out_last   optional [n_rows] f32: only the last column (query_mesh.py:38); out may be NULL then
 *   workspace  device scratch of saf_query_workspace_bytes(n_text, epilogue) bytes (may be NULL if 0)
 * Numerics (ABI 3, round 6): for feat_dim % 16 == 0 and up to 64 labels (or more, 64 / 32 at a time, when `out` is given) the dot
 * products run on fp16 matrix instructions with fp32 accumulation -- fp32 operands cut into two fp16 pieces under power-of-two
 * scales (per label; per feature row, following the row's running maximum), 16-bit features as one piece: a score is within
 * 3 x 2^-22 of sum_k |f_k t_k| of the exact dot product for rows and labels of any magnitude the dtype holds (the reference's scores
 * are compared at 1e-4).  SAF_Q_SPLIT=0 in the environment (read per call): the exact-fp32 matrix instructions instead.
 */
int saf_query_scan(const void* feats, int32_t feat_dtype, int64_t n_rows, int64_t feat_stride,
                   int32_t feat_dim, const float* text, int32_t n_text, int64_t text_stride,
                   int32_t epilogue, float scale, int32_t normalize, float* out, float* out_last,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Device scratch saf_query_scan needs (the surgery weights w[n_text]); 0 for other epilogues. */
size_t saf_query_workspace_bytes(int32_t n_text, int32_t epilogue);

/*
 * Wide scan (BASELINE config 5: hundreds to thousands of text queries over a 16-bit feature
 * volume): scores out[n,q] = scale * <f_n, t_q> (rows optionally L2-normalised first) on the 16-bit
 * matrix cores with fp32 accumulation.  The text embeddings are rounded to the feature dtype.
 *   feats      [n_rows, feat_stride] SAF_F16 or SAF_BF16, feat_dim in {128, 256, 512}
 *   out        [n_rows, out_stride] of out_dtype (SAF_F32, SAF_F16 or SAF_BF16), out_stride >= n_text
 *   workspace  saf_query_wide_workspace_bytes(n_text, feat_dim) bytes of device scratch
 */
size_t saf_query_wide_workspace_bytes(int32_t n_text, int32_t feat_dim);
int saf_query_scan_wide(const void* feats, int32_t feat_dtype, int64_t n_rows, int64_t feat_stride,
                        int32_t feat_dim, const float* text, int32_t n_text, int64_t text_stride,
                        float scale, int32_t normalize, void* out, int32_t out_dtype, int64_t out_stride,
                        void* workspace, size_t workspace_bytes, void* stream);

/*
 * Wide scan with fused epilogues (64 feature rows per wave at one wave per SIMD; feat_dim 256 or 512).  BASELINE
 * config 5's N x Q score matrix (33 GB at 256^3 x 1000 fp16) is twice the volume it is computed from; the
 * reductions the reference's callers apply to it are fused here:
 *   SAF_QW_SCORES         out[n, q] = scale * <f_n, t_q>                         (as saf_query_scan_wide)
 *   SAF_QW_VS_BACKGROUND  the first n_background text rows are shared background prompts, the others targets:
 *                         out[n, t] = softmax(scale * [<f_n, bg_0..>, <f_n, target_t>])[-1] -- query_mesh.py:36-39 and
 *                         hypersim_eval.py:76-81 for every target at once; flags & 1 adds query_mesh.py:39's
 *                         ((r - 0.5) * 2).clamp(0, 1).  out is [n_rows, n_text - n_background].
 *   SAF_QW_ROW_ARGMAX     per row the best query and its score: out_index[n] i32, out_value[n] f32
 *                         (eval_scannet_segmentation.py:553-560: the first label of the argsort); nothing N x Q is written.
 *                         The dot products are compared before the row's scale / norm is applied (one factor per row;
 *                         a negative scale is folded into the text): of equal products the first query wins.
 *   SAF_QW_QUERY_MAX      per query the best row and its score: out_value[q] f32, out_row[q] i64 = row_offset + local
 *                         row (equal scores: the smaller row), -1 / -inf when n_rows = 0; row_offset lets ranks that
 *                         scan voxel shards report global voxel indices.
 * Unused outputs may be NULL.  workspace: saf_query_wide_ex_workspace_bytes(...) bytes, 256-byte aligned.
 */
enum saf_wide_epilogue { SAF_QW_SCORES = 0, SAF_QW_VS_BACKGROUND = 1, SAF_QW_ROW_ARGMAX = 2, SAF_QW_QUERY_MAX = 3 };
size_t saf_query_wide_ex_workspace_bytes(int32_t n_text, int32_t feat_dim, int32_t epilogue, int32_t n_background);
int saf_query_scan_wide_ex(const void* feats, int32_t feat_dtype, int64_t n_rows, int64_t feat_stride,
                           int32_t feat_dim, const float* text, int32_t n_text, int64_t text_stride, float scale,
                           int32_t normalize, int32_t epilogue, int32_t n_background, int32_t flags, void* out,
                           int32_t out_dtype, int64_t out_stride, int32_t* out_index, float* out_value,
                           int64_t* out_row, int64_t row_offset, void* workspace, size_t workspace_bytes, void* stream);

/*
 * After the cross-rank SUM of SAF_SUM-mode volumes: clip_feat <- F/w,
 * rgb <- C/w, tsdf <- T/wt over voxels [first, first+count); rows with zero weight stay zero.
 */
int saf_merge_finalize(const saf_volume* vol, int64_t first_voxel, int64_t n_voxels, void* stream);

/* Inverse of the above for a volume that was fused in SAF_RUNNING_MEAN mode: mean -> sum
 * (x*w), so that it can enter the reduction. */
int saf_mean_to_sum(const saf_volume* vol, int64_t first_voxel, int64_t n_voxels, void* stream);

/*
 * The PACKED route of the frame-sharded merge (new capability, nothing to mirror in the reference): after
 * the all-reduce of `weight` every rank knows which rows ANY rank touched; of a piece whose touched share is small only those
 * rows travel (all_to_all with uneven splits, issued by the host: spatially_aware_ai_amd/distributed.py).  Rows are indexed
 * relative to the first row of the slab / volume being merged; `weight` and `pos` cover the same rows.
 *   saf_merge_scan_touched  pos[i] = number of touched rows (weight > 0) among [0, i), i = 0 .. n_rows (n_rows + 1 words), and
 *                           offs_host[j] = pos[bounds[j]] (bounds: device memory; offs_host: host memory, pinned for an
 *                           asynchronous copy) -- the split sizes of the collective; the caller synchronises the stream once
 *                           before reading them.  workspace: saf_merge_scan_workspace_bytes(n_rows, n_bounds).
 *   saf_merge_pack_rows     the touched rows of [first, first + n_rows) of `src` (rows of row_bytes bytes, a multiple of 4) ->
 *                           packed[pos[r] - pos[first]]: the send buffer, touched rows in ascending order.
 *   saf_merge_add_packed    dst[r] = recv[0][p] + recv[1][p] + ... + recv[world - 1][p] (rank order: deterministic) for the
 *                           touched rows r of [first, first + n_rows), p = pos[r] - pos[first]; recv = [world][mine] rows as
 *                           all_to_all_single leaves them; is_float: f32 rows (else i32).
 */
size_t saf_merge_scan_workspace_bytes(int64_t n_rows, int32_t n_bounds);
int saf_merge_scan_touched(const int32_t* weight, int64_t n_rows, int32_t* pos, const int64_t* bounds, int32_t n_bounds,
                           int32_t* offs_host, void* workspace, size_t workspace_bytes, void* stream);
int saf_merge_pack_rows(const void* src, int64_t row_bytes, const int32_t* weight, const int32_t* pos, int64_t first,
                        int64_t n_rows, void* packed, void* stream);
int saf_merge_add_packed(void* dst, int64_t row_bytes, int32_t is_float, const int32_t* weight, const int32_t* pos, int64_t first,
                         int64_t n_rows, const void* recv, int64_t mine, int32_t world, void* stream);

/*
 * Vertex sampling half of extract_mesh (clipfusion.py:741-760; clip_seem_fusion.py:843-878): for
 * marching-cubes vertices given in voxel-index coordinates, grid = (v + 0.5) * (1/nvox) * 2 - 1 and
 * 3-D grid_sample(align_corners=False, zeros padding) of the volume:
 *   out_feat [V,D] f32   trilinear sample of clip_feat (volume dtype f32 or bf16)
 *   out_rgb  [V,3] f32   trilinear sample of rgb, clamped to [0,1]
 *   out_obj  [V]   f32   nearest sample of obj_idx [N] i32            (optional pair, may be NULL)
 *   out_seg  [V,3] f32   nearest sample of seg_color [N,3] f32, clamped (optional pair, may be NULL)
 */
int saf_sample_vertices(const saf_volume* vol, const float* verts_index, int64_t n_verts, float* out_feat,
                        float* out_rgb, const int32_t* obj_idx, float* out_obj, const float* seg_color,
                        float* out_seg, void* stream);

/*
 * The tiled CLIP front-end in one pass (Clip.img_inference_tiled before the ViT,
 * clipfusion.py:808-823): normalize_img, Unfold into overlapping patch x patch tiles at `stride`, bilinear resize of
 * every tile to out_size x out_size (align_corners = False).
 *   rgb    [batch, 3, H, W] f32 in 0..1 addressed through element strides (channel-last frames as the loaders yield
 *          them: stride_c = 1, stride_x = 3, stride_y = 3 W)
 *   mean3 / std3  HOST pointers to the three channel means / standard deviations (clipfusion.py:774-779)
 *   out    [batch * npy * npx, 3, out_size, out_size] of out_dtype, tile index (b * npy + py) * npx + px as
 *          get_patches orders them (clipfusion.py:800-804)
 */
int saf_clip_tiles(const float* rgb, int32_t batch, int32_t height, int32_t width, int64_t stride_b, int64_t stride_c,
                   int64_t stride_y, int64_t stride_x, int32_t patch, int32_t stride, int32_t out_size,
                   const float* mean3, const float* std3, void* out, int32_t out_dtype, void* stream);

/*
 * Marching cubes on the TSDF, on the device: the mesh half of extract_mesh (clipfusion.py:723-739,
 * clip_seem_fusion.py:824-842) -- un-fused voxels (weight == 0) act as the reference's NaN mask, faces with a vertex on
 * an edge to an un-fused voxel are dropped, unused vertices never exist.  Two calls, because the sizes are results:
 *   count: classifies every cube; counts[0] = vertices, counts[1] = faces (device i64[2]; read them back to allocate);
 *   emit : verts [n_verts, 3] f32 in voxel-index coordinates (x, y, z) -- what skimage returns and the reference feeds
 *          to grid_sample / scales by voxel_size -- ordered by (owner voxel in raster order, axis); faces [n_faces, 3]
 *          i32 ordered by cube in raster order, normals toward positive tsdf.  Both calls take the same workspace
 *          (saf_marching_cubes_workspace_bytes, 256-byte aligned), untouched in between.
 * The triangulation of ambiguous cubes and the vertex order differ from scikit-image's Lewiner variant (not in the
 * build image: parity with it is unpinned); the vertex SET is one vertex per level-crossing grid edge in both.
 */
size_t saf_marching_cubes_workspace_bytes(int64_t n_voxels);
int saf_marching_cubes_count(const float* tsdf, const int32_t* weight, int32_t nx, int32_t ny, int32_t nz, float level,
                             void* workspace, size_t workspace_bytes, int64_t* counts, void* stream);
int saf_marching_cubes_emit(const float* tsdf, const int32_t* weight, int32_t nx, int32_t ny, int32_t nz, float level,
                            void* workspace, size_t workspace_bytes, float* verts, int64_t verts_capacity,
                            int32_t* faces, int64_t faces_capacity, void* stream);

/*
 * On-disk and wire formats of the fused results (host code, any thread).
 *   saf_save_npy   data (device memory when on_device != 0, read after `stream` has drained; else host memory) ->
 *                  NumPy .npy v1.0 at `path`, C order.  dtype_code: 0 f32, 1 bf16 (written as 2-byte records '<V2'),
 *                  2 f16, 3 i32, 4 i64, 5 u8.  Device arrays stream through two pinned 64 MiB buffers (copy of chunk
 *                  k + 1 beside the write of chunk k).  The files of save_files_and_broadcast (clip_seem_fusion.py:563-581).
 *   saf_mesh_json  {"vertices": [[x, y, z], ...], "faces": [[i, j, k], ...], "colors": [[...], ...]} (clip_seem_fusion.py:
 *                  553-559, handy_utils.py:233-239) from HOST arrays; floats are printed as the shortest decimal that
 *                  round-trips the double value of each f32, so a JSON parser returns exactly ndarray.tolist().
 *                  *out is malloc'ed: release it with saf_free.
 *   saf_save_ply   binary little-endian PLY (x y z float, optional red green blue alpha uchar from colours in 0..1,
 *                  faces as `list uchar int`) from HOST arrays: mesh_rgb.ply / mesh_segmentation.ply (:584-600).
 */
int saf_save_npy(const void* data, int32_t on_device, int32_t dtype_code, const int64_t* shape, int32_t ndim,
                 const char* path, void* stream);
int saf_mesh_json(const float* verts, int64_t n_verts, const int32_t* faces, int64_t n_faces, const float* colors,
                  int32_t n_color, char** out, int64_t* out_len);
/* A HOST array [rows, cols] (cols == 0: flat) as the JSON text of ndarray.tolist(), Python's separators; dtype_code 0 f32, 3 i32,
 * 4 i64, 6 f64.  The voxel lists and per-object meshes of scene_knowledge.json (clip_seem_fusion.py:393-417, :603-604). */
int saf_array_json(const void* data, int32_t dtype_code, int64_t rows, int32_t cols, char** out, int64_t* out_len);
void saf_free(void* p);
int saf_save_ply(const char* path, const float* verts, int64_t n_verts, const int32_t* faces, int64_t n_faces,
                 const float* colors, int32_t n_color);

/* Per-voxel argmax of the label histogram with the all-zero row -> -1 rule
 * (clip_seem_fusion.py:315-325).  out[N] i32. */
int saf_label_argmax(const int32_t* labels_one_hot, int64_t n_voxels, int32_t n_classes,
                     int32_t* out, void* stream);

/* ---- the connected-component core of flood_fill_3d (handy_utils.py:295-480) ----
 * labels [nx,ny,nz] i32 class ids (row-major, z fastest), as produced by saf_label_argmax reshaped to the
 * grid (clip_seem_fusion.py:322-338).  Voxels of `null_class` (133 in the reference, handy_utils.py:383)
 * and empty voxels (-1) belong to no object.  Objects are the 26-connected sets of equal class
 * (handy_utils.py:312-344); objects of fewer than `min_voxels` voxels are rejected (3 in the reference,
 * :390).  The k-th accepted object in raster order of its first voxel gets the index -2 - k
 * (handy_utils.py:352-353, :447-450 without a trained in-situ model):
 *   obj_ids   [N] i32   -1 or -2 - k                      (= the reference's voxel_obj_ids)
 *   n_objects [1] i32   number of accepted objects
 *   obj_first / obj_class / obj_count [max_objects] i32 (each may be NULL): first voxel (flat index),
 *   class id and voxel count of object k, for k < max_objects. */
size_t saf_label_components_workspace_bytes(int64_t n_voxels);
int saf_label_components(const int32_t* labels, int32_t nx, int32_t ny, int32_t nz, int32_t null_class,
                         int32_t min_voxels, int32_t* obj_ids, int32_t* n_objects, int32_t max_objects,
                         int32_t* obj_first, int32_t* obj_class, int32_t* obj_count, void* workspace,
                         size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAF_H */
