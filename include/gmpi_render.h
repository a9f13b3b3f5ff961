/*
 * gmpi_render.h -- C ABI of the MI355X-native multiplane-image (MPI) renderer.
 *
 * This is the drop-in boundary for ONE path of apple/ml-gmpi: the gmpi/core renderer
 *   homography()          gmpi/core/mpi.py:26-153     (ray/plane intersection + F.grid_sample)
 *   MPI.forward()         gmpi/core/mpi.py:308-436    (front-to-back over-compositing, depth)
 *   MPIRenderer.render()  gmpi/core/mpi_renderer.py:387-469 (range asserts, [0,1] -> [-1,1])
 * The reference has no native entry point for this path (its only native code is the
 * generator's bias_act/upfirdn2d pybind plugins, gmpi/models/torch_utils/ops/bias_act.cpp:94-97,
 * whose convention -- one POD parameter struct, launch on the caller's stream -- is mirrored
 * here).  Everything below is plain C: pointers, sizes, no torch types.  A Python host binds it
 * with ctypes (ml-gmpi_amd/_lib.py); INTEGRATION.md shows the stub a reference maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates or
 *     frees device memory and never synchronises the stream;
 *   - every entry point returns 0 or a negative GMPI_E_* code; it never throws and never exits
 *     (the reference's `sys.exit(1)` on a ray leaving the last plane, mpi.py:105-128, becomes a
 *     status bit the host turns into the same diagnostics);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - re-entrant; no global state that a result depends on (the only process-wide datum is an atomic launch counter that stamps the view
 *     gate words of GMPI_VARIANT_AUTO's two-kernel launches, so that a workspace never needs clearing).
 */
#ifndef GMPI_RENDER_H
#define GMPI_RENDER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMPI_ABI_VERSION 2

/* storage type of the RGBA volume; arithmetic is always fp32 (mpi_renderer.py:446 `.float()`) */
enum { GMPI_DTYPE_F32 = 0, GMPI_DTYPE_BF16 = 1, GMPI_DTYPE_F16 = 2 };

/* GmpiRenderParams.flags */
enum {
    GMPI_FLAG_ALIGN_CORNERS = 1 << 0,    /* MPI(align_corners=...)            mpi.py:157-159, 86-99    */
    GMPI_FLAG_OUT_PM1 = 1 << 1,          /* write 2*C-1 instead of C          mpi_renderer.py:467      */
    GMPI_FLAG_CHECK_LAST_PLANE = 1 << 2, /* assert_not_out_of_last_plane      mpi.py:381-395, 103-109  */
    GMPI_FLAG_CHECK_RANGE = 1 << 3,      /* rgba/alpha in [0,1] (mpi.py:185-187, mpi_renderer.py:447-449) on AT LEAST the texels the render
                                            SAMPLES (the taps with which some pixel forms its bilinear sample); a kernel may test more:
                                              band kernel (GMPI_VARIANT_BAND, AUTO's large launches): exactly the sampled taps (the landed tap
                                                registers are folded into a running maximum);
                                              tile / strip kernels (LDS, WAVE): every texel of the boxes they stage -- the sampled taps plus
                                                the 16-byte items and box rows around them;
                                              gather kernel: exactly the sampled taps.
                                            STATUS_RGBA_RANGE therefore never reports a texel in [0,1], always reports an out-of-range or
                                            NaN texel that some pixel samples, and MAY miss one that no pixel samples (outside every view's
                                            footprint: 10-16 % of a volume) -- which the reference, testing min/max of the whole tensor,
                                            reports.  The exhaustive test is gmpi_rgba_range_check_launch (one streaming pass); the Python
                                            host runs it with range_check="full" (install()'s default) once per unchanged volume.        */
    GMPI_FLAG_STRICT_ORDER = 1 << 4,     /* one rounding per reference op everywhere (bit-identical to
                                            oracle/mpi_oracle.c); default lets the blend use FMA       */
    GMPI_FLAG_HINT_FRONTAL = 1 << 5,     /* advisory: every view's camera axis (z_dir) is within 0.2 rad of the MPI normal (0,0,1).
                                            Only GMPI_VARIANT_AUTO reads it, and only to choose between kernels that render the
                                            same pixels: the strip kernel's narrow boxes win on small frontal launches, the tile
                                            kernel's shared boxes on tilted ones (profiles/r03_pose_sweep.txt).  Results never
                                            depend on it; without it AUTO assumes a tilted camera.                              */
    GMPI_FLAG_HINT_TILTED = 1 << 6,      /* advisory (round 4), the other end: SOME view's camera axis is more than 0.53 rad off the MPI normal
                                            (the 2-sigma corner of the FFHQ / MetFaces pose range).  The strip kernel's wave-private boxes
                                            overflow there -- config 2 takes 0.25-0.77 ms instead of 0.16 -- while the tile kernel stays at
                                            0.21 (profiles/r04_pose_distribution.txt): AUTO keeps such fp32 launches of 1537-2048 strips (the
                                            window that was measured: config 2) off the strip kernel, and (round 5) 16-bit launches of 256-511
                                            bands of 256 x 8 pixels off the band kernel's two-kernel path (a view it cannot stage costs such a
                                            small launch ~18 us of table kernel and empty launches); launches of up to 512 strips stay with the
                                            strip kernel whatever the hint (its 6-way plane split still wins there).  Like
                                            GMPI_FLAG_HINT_FRONTAL it never changes a result.                                                */
    GMPI_FLAG_GRAD_OVERWRITE = 1 << 7,   /* gmpi_mpi_render_backward_launch only (round 6): the caller does not need what grad_rgba holds.
                                            With the workspace gmpi_render_backward_workspace_bytes asks for, the launch then WRITES every element
                                            of grad_rgba (no zero-fill needed); without the flag it reads, adds and writes back.  On the tile-kernel
                                            path (no workspace, or a launch the gather path does not take) the flag changes nothing: that path only
                                            ever ADDS, and the caller zero-fills as before.                                                      */
    GMPI_FLAG_HINT_OBLIQUE = 1 << 8,     /* advisory (round 6), between the two: SOME view's camera axis is more than 0.35 rad off the MPI normal.  Read (a) like
                                            GMPI_FLAG_HINT_TILTED for 16-bit launches of 256-511 bands of 256 x 8 pixels (kept off the band kernel's two-kernel path), and (b) for
                                            launches whose views SHARE MPIs (views_per_mpi > 1: camera paths over one MPI) and that are large enough (from 1024 bands of 256 x 8 pixels
                                            over a 16-bit volume, 2048 of 128 x 8 over an fp32 one: 8 views of 512^2): without it GMPI_VARIANT_AUTO renders
                                            them with the band kernel -- 9 % (fp32) to 20 % (16-bit volumes) faster than the tile kernel when every view's texel
                                            boxes fit its buffers, which cameras up to 0.35 rad do at 512^2 (and further out on larger images) -- and, when the device finds a view that does not
                                            fit, hands that view's whole group of views to the tile kernel (a table kernel and an empty band launch, ~20 us, for
                                            nothing); with it such launches go to the tile kernel at once, as in rounds 1-5.  GMPI_FLAG_HINT_TILTED implies it.
                                            Never changes a result.                                                                                           */
    GMPI_FLAG_ALL = (1 << 9) - 1         /* every defined bit; any other bit -> GMPI_E_FLAGS            */
};

/* bits of status[0] (OR-accumulated across launches until the caller clears the word) */
enum {
    GMPI_STATUS_OUT_OF_LAST_PLANE = 1u << 0, /* mpi.py:106-109 would have failed                      */
    GMPI_STATUS_RGBA_RANGE = 1u << 1,        /* mpi.py:185-187 / mpi_renderer.py:447-449              */
    GMPI_STATUS_CAMERA_BEHIND_PLANE = 1u << 2, /* mpi.py:70-72 "Camera must be placed closer..."     */
    GMPI_STATUS_BAD_VIEW_INDEX = 1u << 3      /* view_to_mpi[n] outside [0, M): clamped and reported   */
};
#define GMPI_STATUS_WORDS 4

/* kernel selection (GmpiRenderParams.variant) */
enum {
    GMPI_VARIANT_AUTO = 0,
    GMPI_VARIANT_GATHER = 1, /* one pixel per lane, taps straight from global memory (any shape/stride) */
    GMPI_VARIANT_LDS = 2,    /* pixel tiles, texel boxes staged through LDS with 16-byte row loads      */
    GMPI_VARIANT_WAVE = 3,   /* wave-private 32x8 pixel strips, whole RGBA texels (fp32 / fp16) in LDS  */
    GMPI_VARIANT_DMA = 4,    /* RETIRED (round 4): reserved, refused with GMPI_E_VARIANT; gmpi_query(7) == 0            */
    GMPI_VARIANT_BAND = 5    /* 256 x 8 (bf16, fp16) / 128 x 8 (fp32) pixel bands, LDS-DMA loader; needs the workspace         */
};

enum {
    GMPI_OK = 0,
    GMPI_E_NULL = -1,        /* required pointer is NULL                      */
    GMPI_E_SHAPE = -2,       /* non-positive / inconsistent extent; N > 65535 views for the gather kernel
                                or the backward (split the batch)             */
    GMPI_E_DTYPE = -3,       /* unknown rgba_dtype                            */
    GMPI_E_STRIDE = -4,      /* innermost rgba stride != 1 or negative stride */
    GMPI_E_ABI = -5,         /* struct_size does not match this library       */
    GMPI_E_VARIANT = -6,     /* requested kernel variant cannot run this shape */
    GMPI_E_FLAGS = -7,       /* a bit outside GMPI_FLAG_ALL is set            */
    GMPI_E_LAUNCH = -100     /* -100 - hipError_t of the failed launch        */
};

/*
 * One render call = MPI.forward (mpi.py:308-436) [+ the epilogue of MPIRenderer.render].
 *
 *   N views, M multiplane images, D planes per MPI (plane 0 nearest, mpi.py:413).
 *   View n samples MPI  view_to_mpi[n]  (or n / views_per_mpi when view_to_mpi is NULL); this
 *   replaces the reference's expand+cat of the volume per view (mpi.py:331-346) -- the volume is
 *   never replicated.
 */
typedef struct GmpiRenderParams {
    uint32_t struct_size; /* = sizeof(GmpiRenderParams) */
    uint32_t flags;       /* GMPI_FLAG_*                */
    int32_t variant;      /* GMPI_VARIANT_*             */
    int32_t rgba_dtype;   /* GMPI_DTYPE_*               */

    int32_t N, M, D;         /* views, MPIs, planes                                   */
    int32_t Ht, Wt;          /* texture height/width (texels)                          */
    int32_t H, W;            /* rendered image height/width (pixels)                   */
    int32_t views_per_mpi;   /* used when view_to_mpi == NULL (>=1)                    */

    const void *rgba;        /* [M, D, 4, Ht, Wt] planar RGBA in [0,1]                 */
    int64_t rgba_stride[5];  /* element strides; [4] must be 1; 0 allowed on [0] (expand) */
    const int32_t *view_to_mpi; /* [N] or NULL                                          */
    const float *dhw;        /* [M, D, 3] (distance, height, width), contiguous         */
    const float *ray_dir;    /* [N, 3, H, W] unit ray directions, contiguous.  The LDS-staged variants (LDS, WAVE, BAND and therefore AUTO)
                                stage, per pixel tile and plane, the texel box spanned by the tile's four corner pixels: that contains every tap
                                of the tile iff the field is a pinhole camera's (straight pixel lines map to straight lines on every plane), which
                                is what Camera.generate_rays (camera.py:182-211) / gmpi_generate_rays_launch produce.  NaN / infinite rays are
                                handled (NaN pixels, no false status bit); an arbitrary smooth field needs GMPI_VARIANT_GATHER.               */
    const float *eye_pos;    /* [N, 3]                                                  */
    const float *z_dir;      /* [N, 3] optical axis                                     */

    float *rgb_out;          /* [N, 3, H, W]   colour, [0,1] (or [-1,1] with OUT_PM1)   */
    float *depth_out;        /* [N, 1, H, W]   expected depth (mpi.py:434)              */
    float *transmittance_out;/* [N, 1, H, W] or NULL: prod_k (1-a_k+1e-10) -- the cumprod
                                element the reference slices off at mpi.py:423            */
    uint32_t *status;        /* [GMPI_STATUS_WORDS] or NULL; word 0 is OR-ed with GMPI_STATUS_* */

    void *workspace;         /* device scratch owned by the caller, 256-byte aligned, or NULL.  GMPI_VARIANT_BAND keeps its
                                per-plane geometry table there (gmpi_render_workspace_bytes() says how much this call
                                wants); without it GMPI_VARIANT_AUTO uses the kernels that need none.  Contents are
                                scratch: nothing is carried from one call to the next, it need not be cleared, and calls
                                in flight at the same time (different streams) must not share one.                  */
    uint64_t workspace_bytes;
} GmpiRenderParams;

/* Bytes of workspace gmpi_mpi_render_launch can make use of for these parameters (0 when no kernel wants any). */
uint64_t gmpi_render_workspace_bytes(const GmpiRenderParams *params);

/* Enqueue the fused render on `stream`.  Replaces MPI.forward (mpi.py:308-436).
 * GMPI_VARIANT_AUTO picks the kernel from the launch shape; for large launches over bf16 / fp32 volumes with a workspace it enqueues
 * the band kernel and the tile kernel together and shares out the views on the device: the band kernel takes every view whose
 * texel boxes fit its staging buffers (mildly tilted cameras), the tile kernel the others.  An explicit variant that cannot
 * take the parameters returns GMPI_E_VARIANT. */
int gmpi_mpi_render_launch(const GmpiRenderParams *params, void *stream);

/*
 * Gradient of the render w.r.t. the RGBA volume -- what autograd computes when the reference's G-step
 * back-propagates through MPIRenderer.render (gmpi/train.py:740-779; the sampling grid carries no gradient,
 * mpi.py:65).  `params` are the forward's parameters; rgb_out / depth_out may be NULL; transmittance_out, when not
 * NULL, must still hold what the forward wrote (the sweep runs back to front from it; without it every pixel first
 * re-walks the alpha channel).  variant GATHER selects the one-pixel-per-lane kernel (16 global atomics per
 * pixel*plane), anything else the tile kernel that stages the scatter in LDS.  grad_rgb [N,3,H,W] is the gradient
 * w.r.t. the colour the forward wrote (the OUT_PM1 factor 2 is applied inside when that flag is set); grad_depth
 * [N,1,H,W] or NULL; grad_rgba [M,D,4,Ht,Wt] fp32 with the given element strides (innermost 1) is ACCUMULATED into
 * (atomicAdd) -- the caller zero-fills it.
 * Round 6: when params->workspace holds at least gmpi_render_backward_workspace_bytes(params) bytes (256-byte aligned; N D H W 16 bytes for the
 * sample gradients of every pixel and plane), the launch runs WITHOUT atomics: a pixel pass writes the sample gradients, a texel pass gathers them
 * through each plane's homography and writes every cell of grad_rgba once (deterministic; += without GMPI_FLAG_GRAD_OVERWRITE, = with it: then no
 * zero-fill is needed).  align_corners = True, uniform views_per_mpi (no view_to_mpi); other launches take the tile kernels whatever the workspace.
 * `ray_dir` must be a pinhole ray field (straight pixel lines map to straight lines on every plane -- what `Camera.generate_rays` /
 * gmpi_generate_rays_launch produce) for the tile kernels' texel boxes and for the gather's candidate windows; GMPI_VARIANT_GATHER (one pixel per lane,
 * 16 atomics per pixel and plane: the cross-check) makes no such assumption and takes no workspace.
 */
int gmpi_mpi_render_backward_launch(const GmpiRenderParams *params, const float *grad_rgb, const float *grad_depth,
                                    float *grad_rgba, const int64_t *grad_rgba_stride, void *stream);
/* Bytes of caller-owned scratch with which the backward runs without atomics (0: this launch takes the tile kernels). */
uint64_t gmpi_render_backward_workspace_bytes(const GmpiRenderParams *params);

/*
 * Diagnostics for a tripped GMPI_STATUS_OUT_OF_LAST_PLANE: min_u, max_u, min_v, max_v of the
 * normalised grid on the LAST plane per view (what mpi.py:106-109 print).  uv_minmax: [N,4] float.
 * Uses N, M, D, H, W, flags(ALIGN_CORNERS), view_to_mpi/views_per_mpi, dhw, ray_dir, eye_pos.
 */
int gmpi_last_plane_uv_minmax_launch(const GmpiRenderParams *params, float *uv_minmax, void *stream);

/*
 * Exhaustive range check over `count` contiguous elements (the reference's two full min/max
 * passes, mpi.py:185-187 and mpi_renderer.py:447-449): ORs GMPI_STATUS_RGBA_RANGE into status[0]
 * when any value is outside [0,1] or NaN.
 */
int gmpi_rgba_range_check_launch(const void *rgba, int32_t rgba_dtype, int64_t count, uint32_t *status,
                                 void *stream);

/*
 * Per-view epilogue of the reference's drivers (render_video.py:118-126, prepare_fake_data.py):
 *   img8  [N,H,W,3] = uint8( ((rgb_pm1 + 1) / 2) * 255 )           (C truncation, as numpy astype)
 *   dep8  [N,H,W,1] = uint8( clip((depth - near) / (far - near), 0, 1) * 255 )
 * rgb is the OUT_PM1 colour [N,3,H,W]; either output may be NULL.  near/far are the Python floats
 * `ray_start`/`ray_end`; as in numpy the subtraction uses float32(near) and the division uses
 * float32(far - near) (difference taken in double).
 */
int gmpi_frames_to_uint8_launch(const float *rgb_pm1, const float *depth, int32_t N, int32_t H, int32_t W,
                                double depth_near, double depth_far, uint8_t *img8, uint8_t *dep8, void *stream);

/*
 * World-space rays of N pinhole views (gmpi/core/camera.py:182-211 `_generate_rays_torch`, called per view
 * from mpi_renderer.py:320-335): for view n with camera-to-world matrix c2w[n] (row-major 4x4, fp32) and the
 * camera-frame unit directions unit_dirs [3, H*W] (camera.py:98-118),
 *     ray_dir[n,c,p] = fma(R[c][2], d2[p], fma(R[c][1], d1[p], R[c][0] * d0[p]))      R = c2w[n][:3,:3]
 *     eye_pos[n]     = c2w[n][:3,3]            z_dir[n] = R[:,2]
 * The FMA order is the one of the reference's CPU `torch.matmul` (3x3 @ 3xN sgemm): it reproduces the rays of
 * the CPU reference bit for bit (tests/test_host_geometry.py, tests/test_hip_parity.py), which the reference's
 * own GPU path (rocBLAS) does not.  All pointers are device pointers.
 */
int gmpi_generate_rays_launch(const float *c2w, const float *unit_dirs, int32_t N, int32_t H, int32_t W,
                              float *ray_dir, float *eye_pos, float *z_dir, void *stream);

/*
 * Expected depth of the UN-WARPED multiplane image (gmpi/core/light_renderer.py:82-100 `LightRenderer.compute_depth`,
 * the same cumprod weights as mpi.py:421-423 with the identity warp):
 *     depth[b,y,x] = sum_k a_k * T_k * plane_ds[k],   T_0 = 1,  T_{k+1} = T_k * ((1 - a_k) + 1e-10)
 * alpha: [B, D, 1, H, W] view (typically channel 3 of the RGBA volume): element strides for B, D and rows are given,
 * the innermost stride is 1.  depth_out [B,1,H,W]; transmittance_out [B,1,H,W] or NULL.  Streaming kernel: one read
 * of the alpha planes.
 */
int gmpi_alpha_depth_launch(const void *alpha, int32_t alpha_dtype, int64_t stride_b, int64_t stride_d, int64_t stride_row,
                            const float *plane_ds, int32_t B, int32_t D, int32_t H, int32_t W, float *depth_out,
                            float *transmittance_out, void *stream);

/*
 * The rest of the reference's shading augmentation (gmpi/core/light_renderer.py `LightRenderer.render`), after
 * gmpi_alpha_depth_launch:
 *  - gmpi_light_blur_launch: torchvision GaussianBlur of the depth images [B,H,W] (light_renderer.py:51-55,109):
 *    reflect padding, 2-D kernel = outer product of `kernel1d` [ksize] (ksize odd, ksize/2 < H, W).
 *  - gmpi_light_shading_launch: point cloud of the blurred depth along the last plane's texel rays (compute_pcl
 *    :102-120; xyz_last [H,W,3] contiguous), normals from the four neighbour cross products with replicate padding
 *    (get_normal :57-80), Lambert term against light_dir [B,3] (unit vectors), shading[b,y,x] = ka + kd * max(-n.l, 0).
 *  - gmpi_light_apply_launch: out[b,k,0:3] = clip(rgba[b,k,0:3] * shading[b], 0, 1), out[b,k,3] = rgba[b,k,3]
 *    (:193-198).  rgba [B,D,4,H,W] with element strides (innermost 1), any storage dtype; out fp32 contiguous.
 */
int gmpi_light_blur_launch(const float *depth, float *blurred, int32_t B, int32_t H, int32_t W, const float *kernel1d,
                           int32_t ksize, void *stream);
int gmpi_light_shading_launch(const float *depth_blurred, const float *xyz_last, const float *light_dir, float ka, float kd,
                              int32_t B, int32_t H, int32_t W, float *shading, void *stream);
int gmpi_light_apply_launch(const void *rgba, int32_t rgba_dtype, const int64_t *rgba_stride, const float *shading, float *out,
                            int32_t B, int32_t D, int32_t H, int32_t W, void *stream);

/*
 * Backward of the two volume-sized ops of the augmentation (the reference runs it inside the G-step with autograd,
 * train.py:535-541):
 *  - gmpi_light_apply_backward_launch: grad_out [B,D,4,H,W] fp32 contiguous -> grad_rgba [B,D,4,H,W] fp32 contiguous
 *    (rgb: grad*shading where 0 <= rgb*shading <= 1, alpha: passed through) and grad_shading [B,H,W].
 *  - gmpi_alpha_depth_backward_launch: gradient of gmpi_alpha_depth_launch w.r.t. alpha, ADDED to grad_alpha (an fp32
 *    [B,D,1,H,W] view with the given element strides, e.g. channel 3 of grad_rgba).  `transmittance` is the forward's
 *    transmittance_out or NULL.
 */
int gmpi_light_apply_backward_launch(const void *rgba, int32_t rgba_dtype, const int64_t *rgba_stride, const float *shading,
                                     const float *grad_out, float *grad_rgba, float *grad_shading, int32_t B, int32_t D,
                                     int32_t H, int32_t W, void *stream);
int gmpi_alpha_depth_backward_launch(const void *alpha, int32_t alpha_dtype, int64_t stride_b, int64_t stride_d,
                                     int64_t stride_row, const float *plane_ds, const float *transmittance,
                                     const float *grad_depth, float *grad_alpha, int64_t gstride_b, int64_t gstride_d,
                                     int64_t gstride_row, int32_t B, int32_t D, int32_t H, int32_t W, void *stream);

/*
 * Self-test of the default mode's division: the coordinate chain (mpi.py:76, 89-90) divides through correctly rounded
 * reciprocals hoisted out of the plane loop (q0 = n*r, e = fma(-d, q0, n), q = fma(e, r, q0) with r = RN(1/d)); this entry
 * compares that against the IEEE division on `pairs` pseudo-random operand pairs (classes: 0 zdiff/ray_z, 1 x/(w/2),
 * 2 full-range significands with exponents within +-20, 3 divisors 2^k (2 - 2^-23)) and ADDS the number of pairs whose
 * quotients differ in any bit to mismatches[class] (device memory, 4 x uint64, zero-filled by the caller).
 */
int gmpi_selftest_division_launch(uint64_t pairs, uint32_t seed, uint64_t *mismatches, void *stream);

/*
 * Diagnostic: one read-only pass over `bytes` of device memory with the fastest read pattern measured on gfx950 (non-temporal dword loads,
 * profiles/r03_calibration.txt).  bench.py times it over the render's own volume and quotes the rate as `roofline.stream_read_gbs`: what a
 * kernel that did nothing but read the volume would reach on this box.  `sink` (4 bytes, may be NULL) only keeps the loads alive.
 */
int gmpi_stream_probe_launch(const void *buf, uint64_t bytes, uint32_t *sink, void *stream);

/* what: 0 ABI version, 1 sizeof(GmpiRenderParams), 2 target arch number (950), 3 LDS bytes the
 * LDS variant uses per workgroup, 4 pixel-tile width, 5 pixel-tile height, 6 whether
 * GMPI_VARIANT_WAVE is built in, 7 GMPI_VARIANT_DMA (retired: 0), 8 GMPI_VARIANT_BAND, 9 the number of 256 x 8 pixel bands from which
 * GMPI_VARIANT_AUTO uses the band kernel on bf16 volumes, 10 the number of 128 x 8 pixel bands on fp32 volumes.  Unknown -> -1.                                        */
int gmpi_query(int32_t what);

const char *gmpi_version_string(void);

#ifdef __cplusplus
}
#endif
#endif /* GMPI_RENDER_H */
