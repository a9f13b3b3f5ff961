/*
 * mpi_oracle.c -- CPU restatement of the GMPI multiplane-image render path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP renderer in
 * ml-gmpi_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it (through oracle/oracle.py).  The product path never calls into it and fails loudly when
 * the HIP library is missing.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors of its own (SURVEY.md
 * section 4), so this restatement is pinned against outputs of the reference itself, generated in
 * the build container by oracle/make_golden.py (imports /root/reference/gmpi/core, CPU fp32) and
 * committed under tests/golden/.  tests/test_oracle_golden.py re-checks every fixture.
 *
 * What it restates (reference file:line, all under /root/reference/gmpi/core/):
 *   mpi.py:74-79    z_diff = d - eye_z ; scale = z_diff / ray_z ; xyz = eye + ray * scale
 *   mpi.py:86-99    u = 2x/w, v = 2y/h ; align_corners=False narrows in-range u,v by 0.95
 *   mpi.py:136-142  F.grid_sample(bilinear, zeros, align_corners)  -> third party: PyTorch
 *                   aten::grid_sampler_2d (environment.yml pins pytorch 1.9.1; the formula is
 *                   unchanged in the installed torch 2.10: ATen/native/GridSampler.h:27-35
 *                   unnormalize, and the nw/ne/sw/se accumulation order of GridSampler.cpp)
 *   mpi.py:149-151  dist2depth = <ray, z_dir> ; depth = scale * dist2depth ; disp = 1/depth
 *   mpi.py:411      depth = 1/disp
 *   mpi.py:421-423  alphas_shifted = [1, 1-a+1e-10] ; weights = a * cumprod(alphas_shifted)[:-1]
 *   mpi.py:430,434  color = sum_k w*rgb ; depth = sum_k w*depth_k
 *   mpi.py:103-109  min/max of u,v on the last plane (the assert_not_out_of_last_plane check)
 *   mpi.py:70-72    every plane distance >= eye_z of the first view
 *   mpi.py:185-187, mpi_renderer.py:447-449  range checks on alpha / rgba
 *   light_renderer.py:82-100  LightRenderer.compute_depth (gmpi_oracle_alpha_depth)
 *
 * Arithmetic contract: IEEE-754 binary32, one rounding per written operation, no FMA contraction
 * (build with -ffp-contract=off, no -ffast-math; x86-64 SSE2 float math has no excess precision).
 *
 * Layouts (all row-major, contiguous):
 *   rgba        [M, D, 4, Ht, Wt] float   plane 0 nearest
 *   view_to_mpi [N] int32                 which MPI a view samples (replaces expand+cat, mpi.py:331-346)
 *   dhw         [M, D, 3] float           (distance, height, width)
 *   ray_dir     [N, 3, H, W] float ; eye [N,3] ; zdir [N,3]
 *   color       [N, 3, H, W] float in [0,1] ; depth [N,1,H,W] ; transmittance [N,1,H,W] (optional)
 *   uv_minmax   [N, 4] float (min_u, max_u, min_v, max_v on the last plane) (optional)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define NARROW_SCALE 0.95f /* mpi.py:23 */

typedef struct {
    float ix, iy;
} gmpi_oracle_coord;

/* mpi.py:74-99 + GridSampler.h:27-35.  Every line is one rounding. */
static inline gmpi_oracle_coord plane_coord(float d, float ph, float pw, float ex, float ey, float ez,
                                            float rx, float ry, float rz, int Ht, int Wt,
                                            int align_corners, float *u_out, float *v_out, float *s_out) {
    float zdiff = d - ez;
    float s = zdiff / rz;
    float tx = rx * s;
    float ty = ry * s;
    float x = ex + tx;
    float y = ey + ty;
    float x2 = 2.0f * x;
    float y2 = 2.0f * y;
    float u = x2 / pw;
    float v = y2 / ph;
    gmpi_oracle_coord c;
    if (align_corners) {
        float u1 = u + 1.0f;
        float v1 = v + 1.0f;
        c.ix = u1 * ((float)(Wt - 1) * 0.5f);
        c.iy = v1 * ((float)(Ht - 1) * 0.5f);
    } else {
        if (v >= -1.0f && v <= 1.0f) v = v * NARROW_SCALE;
        if (u >= -1.0f && u <= 1.0f) u = u * NARROW_SCALE;
        float u1 = u + 1.0f;
        float v1 = v + 1.0f;
        float ux = u1 * (float)Wt;
        float vy = v1 * (float)Ht;
        float uxm = ux - 1.0f;
        float vym = vy - 1.0f;
        c.ix = uxm * 0.5f;
        c.iy = vym * 0.5f;
    }
    *u_out = u;
    *v_out = v;
    *s_out = s;
    return c;
}

static inline float texel(const float *chan, int Ht, int Wt, long yi, long xi) {
    if (xi < 0 || yi < 0 || xi > (long)Wt - 1 || yi > (long)Ht - 1) return 0.0f;
    return chan[(size_t)yi * (size_t)Wt + (size_t)xi];
}

/*
 * Returns 0 on success.  status (optional, 4 x uint32) accumulates:
 *   status[0] bit0: a ray leaves the last plane (|u|>1 or |v|>1 or NaN there)   mpi.py:106-109
 *             bit1: an rgba value that was SAMPLED lies outside [0,1] or is NaN  (touched texels only)
 *             bit2: a plane distance is smaller than eye_z of view 0             mpi.py:70-72
 */
int gmpi_oracle_render(const float *rgba, const int32_t *view_to_mpi, const float *dhw, const float *ray_dir,
                       const float *eye, const float *zdir, int N, int M, int D, int Ht, int Wt, int H, int W,
                       int align_corners, float *color, float *depth, float *transmittance, float *uv_minmax,
                       uint32_t *status) {
    if (N < 0 || M <= 0 || D <= 0 || Ht <= 0 || Wt <= 0 || H <= 0 || W <= 0) return -1;
    const size_t HW = (size_t)H * (size_t)W;
    const size_t THW = (size_t)Ht * (size_t)Wt;
    uint32_t flags = 0;
    for (int n = 0; n < N; ++n) {
        const int m = view_to_mpi ? view_to_mpi[n] : n;
        if (m < 0 || m >= M) return -2;
        const float ex = eye[3 * n + 0], ey = eye[3 * n + 1], ez = eye[3 * n + 2];
        const float zx = zdir[3 * n + 0], zy = zdir[3 * n + 1], zz = zdir[3 * n + 2];
        const float *rd = ray_dir + (size_t)n * 3 * HW;
        const float *pd = dhw + (size_t)m * D * 3;
        const float ez0 = eye[2];
        for (int k = 0; k < D; ++k)
            if (!(pd[3 * k] >= ez0)) flags |= 4u;
        float mnu = INFINITY, mxu = -INFINITY, mnv = INFINITY, mxv = -INFINITY;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(| : flags) reduction(min : mnu, mnv) reduction(max : mxu, mxv)
#endif
        for (int r = 0; r < H; ++r) {
            for (int c = 0; c < W; ++c) {
                const size_t p = (size_t)r * W + c;
                const float rx = rd[p], ry = rd[HW + p], rz = rd[2 * HW + p];
                /* einsum("nchw,nc->nhw") mpi.py:149, left to right */
                float dot = rx * zx;
                dot = dot + ry * zy;
                dot = dot + rz * zz;
                float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f, Z = 0.0f;
                for (int k = 0; k < D; ++k) {
                    float u, v, s;
                    gmpi_oracle_coord q = plane_coord(pd[3 * k], pd[3 * k + 1], pd[3 * k + 2], ex, ey, ez, rx, ry,
                                                      rz, Ht, Wt, align_corners, &u, &v, &s);
                    if (k == D - 1) {
                        if (!(u >= -1.0f && u <= 1.0f && v >= -1.0f && v <= 1.0f)) flags |= 1u;
                        if (u < mnu) mnu = u;
                        if (u > mxu) mxu = u;
                        if (v < mnv) mnv = v;
                        if (v > mxv) mxv = v;
                    }
                    /* bilinear, zeros padding */
                    float fx0 = floorf(q.ix), fy0 = floorf(q.iy);
                    float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
                    float wx1 = q.ix - fx0, wx0 = fx1 - q.ix;
                    float wy1 = q.iy - fy0, wy0 = fy1 - q.iy;
                    float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
                    /* clamp before the integer cast so that huge / NaN coordinates stay out of range */
                    long x0, y0;
                    if (!(fx0 >= -2.0f && fx0 <= (float)Wt)) x0 = -2; else x0 = (long)fx0;
                    if (!(fy0 >= -2.0f && fy0 <= (float)Ht)) y0 = -2; else y0 = (long)fy0;
                    const float *base = rgba + ((size_t)m * D + k) * 4 * THW;
                    float smp[4];
                    for (int ch = 0; ch < 4; ++ch) {
                        const float *chan = base + (size_t)ch * THW;
                        float t_nw = texel(chan, Ht, Wt, y0, x0);
                        float t_ne = texel(chan, Ht, Wt, y0, x0 + 1);
                        float t_sw = texel(chan, Ht, Wt, y0 + 1, x0);
                        float t_se = texel(chan, Ht, Wt, y0 + 1, x0 + 1);
                        if (!(t_nw >= 0.0f && t_nw <= 1.0f && t_ne >= 0.0f && t_ne <= 1.0f && t_sw >= 0.0f &&
                              t_sw <= 1.0f && t_se >= 0.0f && t_se <= 1.0f))
                            flags |= 2u;
                        float acc = t_nw * nw;
                        acc = acc + t_ne * ne;
                        acc = acc + t_sw * sw;
                        acc = acc + t_se * se;
                        smp[ch] = acc;
                    }
                    /* mpi.py:150-151, 411 */
                    float dep = s * dot;
                    float disp = 1.0f / dep;
                    float depk = 1.0f / disp;
                    /* mpi.py:421-434 */
                    float a = smp[3];
                    float wgt = a * T;
                    Cr = Cr + wgt * smp[0];
                    Cg = Cg + wgt * smp[1];
                    Cb = Cb + wgt * smp[2];
                    Z = Z + wgt * depk;
                    float om = 1.0f - a;
                    om = om + 1e-10f;
                    T = T * om;
                }
                color[((size_t)n * 3 + 0) * HW + p] = Cr;
                color[((size_t)n * 3 + 1) * HW + p] = Cg;
                color[((size_t)n * 3 + 2) * HW + p] = Cb;
                depth[(size_t)n * HW + p] = Z;
                if (transmittance) transmittance[(size_t)n * HW + p] = T;
            }
        }
        if (uv_minmax) {
            uv_minmax[4 * n + 0] = mnu;
            uv_minmax[4 * n + 1] = mxu;
            uv_minmax[4 * n + 2] = mnv;
            uv_minmax[4 * n + 3] = mxv;
        }
    }
    if (status) status[0] |= flags;
    return 0;
}

/* Sampling coordinates only (ix, iy per view/plane/pixel), for localising a coordinate mismatch. */
int gmpi_oracle_coords(const int32_t *view_to_mpi, const float *dhw, const float *ray_dir, const float *eye, int N,
                       int M, int D, int Ht, int Wt, int H, int W, int align_corners, float *ix_out,
                       float *iy_out) {
    const size_t HW = (size_t)H * (size_t)W;
    for (int n = 0; n < N; ++n) {
        const int m = view_to_mpi ? view_to_mpi[n] : n;
        if (m < 0 || m >= M) return -2;
        const float *rd = ray_dir + (size_t)n * 3 * HW;
        const float *pd = dhw + (size_t)m * D * 3;
        for (int k = 0; k < D; ++k)
            for (size_t p = 0; p < HW; ++p) {
                float u, v, s;
                gmpi_oracle_coord q =
                    plane_coord(pd[3 * k], pd[3 * k + 1], pd[3 * k + 2], eye[3 * n], eye[3 * n + 1], eye[3 * n + 2],
                                rd[p], rd[HW + p], rd[2 * HW + p], Ht, Wt, align_corners, &u, &v, &s);
                ix_out[((size_t)n * D + k) * HW + p] = q.ix;
                iy_out[((size_t)n * D + k) * HW + p] = q.iy;
            }
    }
    return 0;
}

/* Full-volume range check (mpi.py:185-187 alpha, mpi_renderer.py:447-449 rgba): returns bit1 semantics. */
uint32_t gmpi_oracle_range_check(const float *rgba, size_t count) {
    uint32_t bad = 0;
    for (size_t i = 0; i < count; ++i)
        if (!(rgba[i] >= 0.0f && rgba[i] <= 1.0f)) bad = 2u;
    return bad;
}

/*
 * light_renderer.py:82-100 LightRenderer.compute_depth: alphas_shifted = [1, 1-a+1e-10]; weights = a * cumprod[:-1];
 * depth = sum_k weights_k * plane_ds_k   (un-warped MPI).  alpha [B,D,H,W] contiguous.
 */
int gmpi_oracle_alpha_depth(const float *alpha, const float *plane_ds, int B, int D, int H, int W, float *depth,
                            float *transmittance) {
    const size_t HW = (size_t)H * (size_t)W;
    for (int b = 0; b < B; ++b)
        for (size_t p = 0; p < HW; ++p) {
            float T = 1.0f, Z = 0.0f;
            for (int k = 0; k < D; ++k) {
                const float a = alpha[((size_t)b * D + k) * HW + p];
                const float w = a * T;
                const float wd = w * plane_ds[k];
                Z = Z + wd;
                float om = 1.0f - a;
                om = om + 1e-10f;
                T = T * om;
            }
            depth[(size_t)b * HW + p] = Z;
            if (transmittance) transmittance[(size_t)b * HW + p] = T;
        }
    return 0;
}

int gmpi_oracle_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
