// gmpi_device.hpp -- device-side arithmetic shared by the render kernels (gfx950 / CDNA4 only).
//
// The coordinate chain below must reproduce the reference's fp32 arithmetic EXACTLY
// (gmpi/core/mpi.py:74-99 followed by ATen's grid_sampler unnormalize): a 1-ulp difference in
// ray/plane intersection moves the sample position by up to 4e-4 texel at 1024^2 and the output by
// up to 5e-4 on white-noise textures (hard part 1).  Hence:
//   * this translation unit is compiled with -ffp-contract=off (and the pragma below): a*b+c is
//     v_mul_f32 + v_add_f32, never v_fma_f32, unless __builtin_fmaf is written out;
//   * a/b is the correctly rounded IEEE division (hipcc default
//     -fhip-fp32-correctly-rounded-divide-sqrt; never -ffast-math);
//   * fp32 denormals are on (gfx9 default).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace gmpi {

constexpr float kNarrowScale = 0.95f;  // mpi.py:23 ALIGN_CORNERS_FALSE_NARROW_SCALE

// Kernel-side view of GmpiRenderParams (include/gmpi_render.h), passed by value.
struct KParams {
    const void* rgba;
    const int32_t* view_to_mpi;
    const float* dhw;
    const float* ray_dir;
    const float* eye_pos;
    const float* z_dir;
    float* rgb_out;
    float* depth_out;
    float* T_out;
    uint32_t* status;
    int64_t s_mpi, s_plane, s_chan, s_row;  // element strides of rgba [M,D,4,Ht,Wt] (col stride 1)
    int32_t N, M, D, Ht, Wt, H, W, views_per_mpi;
    uint32_t flags;
    void* ws;           // caller's workspace (GmpiRenderParams.workspace): the band kernel's geometry table
    uint64_t ws_bytes;
    // View gate (GMPI_VARIANT_AUTO's two-kernel launches, gmpi_abi.hip): the band kernel's table kernel stamps gate[n] = gate_gen for every view
    // with a box that does not fit the band kernel's staging buffers; a render kernel takes view n iff (gate[n] == gate_gen) == (gate_sense != 0).
    // Whatever the workspace held before, exactly one of the two kernels renders each view.  gate == nullptr: every view.
    uint32_t* gate;
    uint32_t gate_gen, gate_sense;
    int32_t band_cols;  // band kernel: band columns per XCD window (render_band.hip band_pos; set by launch_band)
    int32_t band_rot, band_split;   // band kernel: per-view rotation of the XCD <-> run assignment, pieces per run (xcd_item_per_group)
    int32_t band_tail;              // band kernel: the last band_tail bands of every XCD's last run are handed out by tickets (render_band.hip)
};

// blockIdx -> work item (pixel tile / band), "per view group" form: XCD x = blockIdx % 8 (workgroups are dealt round-robin to the 8 XCDs)
// gets a contiguous run of the items of EVERY group of views that share an MPI, so that row-major neighbours meet in one L2 AND the XCDs walk
// the views together -- a launch that only renders some of the views (gate) still fills every XCD.  Returns n_items for "no item".
// `rot`: group g hands XCD x the run that XCD (x + g rot) % 8 would get in group 0.  The dispatcher deals workgroups to the XCDs round-robin and NO work moves
// between XCDs afterwards: a launch ends when the most loaded XCD is done.  With rot = 0 every XCD renders the same region of every view; regions differ in
// cost (the camera's keystone makes the texel boxes of some band rows taller: a third DMA pass), and what is expensive in one view tends to be expensive in the
// next: rotating the assignment per view averages that out (round 6, render_band.hip; measured per XCD with s_memtime stamps: profiles/r06_band_order.txt).
// `split` (1, 2, 4 or 8): an XCD's run of a group is cut into `split` pieces taken from regions 8 / split apart (piece s from region + s 8 / split): the XCD then
// renders pieces of opposite parts of the view -- whose costs complement each other -- already within ONE view.
__device__ __forceinline__ int xcd_item_per_group(int block, int group_items, int n_items, int rot = 0, int split = 1) {
    const int per_xcd = (group_items + 7) / 8, n_groups = (n_items + group_items - 1) / group_items;
    const int jb = block / 8, grp = jb / per_xcd, rr = jb - grp * per_xcd;
    const int piece = split > 1 ? rr * split / per_xcd : 0;
    const int in_group = ((block + grp * rot + piece * (8 / max(split, 1))) % 8) * per_xcd + rr;
    const int item = grp * group_items + in_group;
    return (grp >= n_groups || in_group >= group_items || item >= n_items) ? n_items : item;
}
inline __host__ unsigned xcd_grid_per_group(int group_items, int n_items) {
    return static_cast<unsigned>(((group_items + 7) / 8) * 8 * ((n_items + group_items - 1) / group_items));
}

__device__ __forceinline__ bool view_gated_out(const KParams& p, int n) {
    return p.gate != nullptr && ((p.gate[n] == p.gate_gen) != (p.gate_sense != 0u));
}

// MPI sampled by view n: view_to_mpi[n], or n / views_per_mpi.  An index outside [0, M) would address dhw and the volume
// out of bounds: it is clamped and reported (status bit 8 = GMPI_STATUS_BAD_VIEW_INDEX) instead.
__device__ __forceinline__ int view_mpi(const KParams& p, int n, uint32_t& bad) {
    int m = p.view_to_mpi ? p.view_to_mpi[n] : n / p.views_per_mpi;
    if (m < 0 || m >= p.M) {
        bad |= 8u;
        m = min(max(m, 0), p.M - 1);
    }
    return m;
}

// ---- texel fetch: storage type -> fp32 (exact upcast, mpi_renderer.py:446) -----------------------
struct bf16_t { uint16_t bits; };
struct f16_t { _Float16 v; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return __uint_as_float(static_cast<uint32_t>(v.bits) << 16); }
__device__ __forceinline__ float to_f32(f16_t v) { return static_cast<float>(v.v); }

// ---- ray/plane intersection -> unnormalised texture coordinates --------------------------------
// One rounding per line, in the reference's order.
//   zdiff = d - eye_z                (mpi.py:74)   computed by the caller (wave-uniform)
//   s     = zdiff / ray_z            (mpi.py:76)
//   x     = eye_x + ray_x * s        (mpi.py:79)
//   u     = (2 x) / w                (mpi.py:90)   2*x is exact
//   AC:   ix = (u + 1) * ((Wt-1)/2)  == ((u+1)/2)*(Wt-1) as a real number -> same single rounding
//   !AC:  u *= 0.95 if -1<=u<=1 (mpi.py:98-99); ix = ((u+1)*Wt - 1)/2   (GridSampler.h:27-35)
template <bool AC>
__device__ __forceinline__ void plane_coord(float zdiff, float ph, float pw, float ex, float ey, float rx, float ry,
                                            float rz, float cx, float cy, float& ix, float& iy, float& s, float& u,
                                            float& v) {
    s = zdiff / rz;
    const float tx = rx * s;
    const float ty = ry * s;
    const float x = ex + tx;
    const float y = ey + ty;
    const float x2 = 2.0f * x;
    const float y2 = 2.0f * y;
    u = x2 / pw;
    v = y2 / ph;
    if (AC) {
        const float u1 = u + 1.0f;
        const float v1 = v + 1.0f;
        ix = u1 * cx;  // cx = (Wt-1)/2
        iy = v1 * cy;  // cy = (Ht-1)/2
    } else {
        if (v >= -1.0f && v <= 1.0f) v = v * kNarrowScale;
        if (u >= -1.0f && u <= 1.0f) u = u * kNarrowScale;
        const float u1 = u + 1.0f;
        const float v1 = v + 1.0f;
        const float ux = u1 * cx;  // cx = Wt
        const float vy = v1 * cy;  // cy = Ht
        const float uxm = ux - 1.0f;
        const float vym = vy - 1.0f;
        ix = uxm * 0.5f;
        iy = vym * 0.5f;
    }
}

// ---- division through a correctly rounded reciprocal ------------------------------------------------
// q = n / d, correctly rounded, from r = RN(1/d):   q0 = RN(n*r);  e = n - d*q0 (exact, one FMA);
// q = RN(q0 + e*r).  This is Markstein's final-correction step ("IA-64 and Elementary Functions",
// thm. 8.2 / Cornea-Harrison-Tang): with r the CORRECTLY ROUNDED reciprocal and q0 within one ulp of
// n/d, the corrected quotient is the correctly rounded one (for normal-range operands and results;
// the single exceptional significand pattern d = 2^k*(2-2^-23) needs n*r to be faithful, which it
// is here).  r is an IEEE division itself (1.0f / d) but a loop-invariant one: per pixel for ray_z,
// per plane for the plane extents.  `gmpi_selftest_division_launch` (C ABI, gmpi_abi.hip) compares this
// against the hardware-correct `/` on 2^32 operand pairs drawn from the renderer's ranges and on the edge
// pattern (tests/test_hip_parity.py::test_division_through_reciprocal_is_exact runs it), and
// tests/test_hip_parity.py::test_default_mode_samples_the_strict_texels checks the texel indices of the
// default mode against the strict-order mode (compiler division) on fuzzed poses.
__device__ __forceinline__ float div_by_recip(float n, float d, float r) {
    const float q0 = n * r;
    const float e = __builtin_fmaf(-d, q0, n);
    return __builtin_fmaf(e, r, q0);
}

// Same chain as plane_coord(), with the three divisions taken through reciprocals:
//   rrz = RN(1/ray_z) (per pixel), rw = RN(1/hw), rh = RN(1/hh) with hw = w/2, hh = h/2 (per plane).
//   (2x)/w == x/(w/2) exactly (power-of-two scaling), so u = x/hw.
template <bool AC>
__device__ __forceinline__ void plane_coord_recip(float zdiff, float hw, float hh, float rw, float rh, float ex, float ey,
                                                  float rx, float ry, float rz, float rrz, float cx, float cy, float& ix,
                                                  float& iy, float& s) {
    s = div_by_recip(zdiff, rz, rrz);
    const float tx = rx * s;
    const float ty = ry * s;
    const float x = ex + tx;
    const float y = ey + ty;
    float u = div_by_recip(x, hw, rw);
    float v = div_by_recip(y, hh, rh);
    if (AC) {
        const float u1 = u + 1.0f;
        const float v1 = v + 1.0f;
        ix = u1 * cx;
        iy = v1 * cy;
    } else {
        if (v >= -1.0f && v <= 1.0f) v = v * kNarrowScale;
        if (u >= -1.0f && u <= 1.0f) u = u * kNarrowScale;
        const float u1 = u + 1.0f;
        const float v1 = v + 1.0f;
        const float ux = u1 * cx;
        const float vy = v1 * cy;
        const float uxm = ux - 1.0f;
        const float vym = vy - 1.0f;
        ix = uxm * 0.5f;
        iy = vym * 0.5f;
    }
}

// Bilinear footprint: integer corner (clamped so that NaN / huge coordinates stay out of range) and
// the four weights nw, ne, sw, se in ATen's order.
struct Footprint {
    int x0, y0;
    float nw, ne, sw, se;
};

__device__ __forceinline__ Footprint footprint(float ix, float iy, int Ht, int Wt) {
    Footprint f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
    const float wx1 = ix - fx0, wx0 = fx1 - ix;
    const float wy1 = iy - fy0, wy0 = fy1 - iy;
    f.nw = wx0 * wy0;
    f.ne = wx1 * wy0;
    f.sw = wx0 * wy1;
    f.se = wx1 * wy1;
    f.x0 = (fx0 >= -2.0f && fx0 <= static_cast<float>(Wt)) ? static_cast<int>(fx0) : -2;
    f.y0 = (fy0 >= -2.0f && fy0 <= static_cast<float>(Ht)) ? static_cast<int>(fy0) : -2;
    return f;
}

// Composite state per pixel (mpi.py:421-434): T = running cumprod, C/Z the weighted sums.
struct Accum {
    float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f, z = 0.0f;
};

// STRICT: exactly the oracle's op sequence.  Otherwise the blend uses FMA (fewer roundings; the
// difference is below 1e-6, everything after ix/iy is benign -- SURVEY.md section 7).
template <bool STRICT>
__device__ __forceinline__ float bilerp(float t_nw, float t_ne, float t_sw, float t_se, const Footprint& f) {
    if (STRICT) {
        float acc = t_nw * f.nw;
        acc = acc + t_ne * f.ne;
        acc = acc + t_sw * f.sw;
        acc = acc + t_se * f.se;
        return acc;
    } else {
        float acc = t_nw * f.nw;
        acc = __builtin_fmaf(t_ne, f.ne, acc);
        acc = __builtin_fmaf(t_sw, f.sw, acc);
        acc = __builtin_fmaf(t_se, f.se, acc);
        return acc;
    }
}

template <bool STRICT>
__device__ __forceinline__ void blend(Accum& A, float r, float g, float b, float a, float s, float dot) {
    if (STRICT) {
        const float dep = s * dot;          // mpi.py:150
        const float disp = 1.0f / dep;      // mpi.py:151
        const float depk = 1.0f / disp;     // mpi.py:411
        const float w = a * A.T;            // mpi.py:423
        A.r = A.r + w * r;                  // mpi.py:430
        A.g = A.g + w * g;
        A.b = A.b + w * b;
        A.z = A.z + w * depk;               // mpi.py:434
        float om = 1.0f - a;                // mpi.py:421
        om = om + 1e-10f;
        A.T = A.T * om;
    } else {
        // depth_k = s*dot (1/(1/x) == x to within an ulp); dot is constant per pixel, so sum w*s and scale once
        // at the end (finish_depth) -- depth tolerance is 1e-5, this moves it by a few 1e-7
        const float w = a * A.T;
        A.r = __builtin_fmaf(w, r, A.r);
        A.g = __builtin_fmaf(w, g, A.g);
        A.b = __builtin_fmaf(w, b, A.b);
        A.z = __builtin_fmaf(w, s, A.z);
        // (T - a*T would save an instruction but cancels behind nearly opaque planes: 1 - a is exact, a*T is not)
        float om = 1.0f - a;
        om = om + 1e-10f;
        A.T = A.T * om;
    }
}

template <bool STRICT>
__device__ __forceinline__ float finish_depth(const Accum& A, float dot) {
    return STRICT ? A.z : A.z * dot;
}

__device__ __forceinline__ bool in_unit(float v) { return v >= 0.0f && v <= 1.0f; }

// Direct-from-global bilinear sample of the 4 channels of one plane (zeros padding): clamp the
// address, zero the weight of a tap that lies outside the texture.  `pl` points at channel 0.
template <typename TexT, bool STRICT>
__device__ __forceinline__ void gather_sample(const TexT* __restrict__ pl, int64_t s_chan, int64_t s_row, int Ht, int Wt,
                                              float ix, float iy, bool check_range, uint32_t& bad, float (&smp)[4]) {
    Footprint f = footprint(ix, iy, Ht, Wt);
    const bool x0in = f.x0 >= 0 && f.x0 <= Wt - 1, x1in = f.x0 >= -1 && f.x0 <= Wt - 2;
    const bool y0in = f.y0 >= 0 && f.y0 <= Ht - 1, y1in = f.y0 >= -1 && f.y0 <= Ht - 2;
    if (!(x0in && y0in)) f.nw = 0.0f;
    if (!(x1in && y0in)) f.ne = 0.0f;
    if (!(x0in && y1in)) f.sw = 0.0f;
    if (!(x1in && y1in)) f.se = 0.0f;
    const int xa = min(max(f.x0, 0), Wt - 1), xb = min(max(f.x0 + 1, 0), Wt - 1);
    const int ya = min(max(f.y0, 0), Ht - 1), yb = min(max(f.y0 + 1, 0), Ht - 1);
    const int64_t oa = static_cast<int64_t>(ya) * s_row, ob = static_cast<int64_t>(yb) * s_row;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const TexT* __restrict__ ch = pl + c * s_chan;
        const float t_nw = to_f32(ch[oa + xa]);
        const float t_ne = to_f32(ch[oa + xb]);
        const float t_sw = to_f32(ch[ob + xa]);
        const float t_se = to_f32(ch[ob + xb]);
        if (check_range && !(in_unit(t_nw) && in_unit(t_ne) && in_unit(t_sw) && in_unit(t_se))) bad |= 2u;
        smp[c] = bilerp<STRICT>(t_nw, t_ne, t_sw, t_se, f);
    }
}

// floor(x) as an integer in one instruction (v_cvt_flr_i32_f32); NaN -> 0, saturating
__device__ __forceinline__ int floor_to_int(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// OR a per-lane flag word into status[0] with at most one atomic per wave.
__device__ __forceinline__ void report_status(uint32_t* status, uint32_t bad) {
    if (status == nullptr) return;
    if (__any(bad != 0u)) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o);
        if ((threadIdx.x + threadIdx.y * blockDim.x) % 64 == 0) atomicOr(status, bad);
    }
}

}  // namespace gmpi
