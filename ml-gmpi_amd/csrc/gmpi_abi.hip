// gmpi_abi.hip -- the extern "C" entry points declared in include/gmpi_render.h, plus the small
// auxiliary kernels (diagnostics, exhaustive range check, uint8 frame epilogue).
#include "gmpi_device.hpp"

#include "../../include/gmpi_render.h"

#include <atomic>
#include <cstdlib>

namespace gmpi {

hipError_t launch_gather(const KParams& p, int dtype, hipStream_t stream);  // render_gather.hip
hipError_t launch_lds(const KParams& p, int dtype, int tune, hipStream_t stream);  // render_lds.hip
hipError_t launch_backward(const KParams& p, int dtype, const float* g_rgb, const float* g_depth, float* g_rgba,
                           const int64_t* gstride, bool tiles, hipStream_t stream);  // render_backward.hip
bool lds_variant_supports(const KParams& p, int dtype);                     // render_lds.hip
int lds_variant_query(int what);                                            // render_lds.hip
hipError_t launch_wave(const KParams& p, int dtype, int tune, hipStream_t stream);  // render_wave.hip
bool wave_variant_supports(const KParams& p, int dtype);                    // render_wave.hip
hipError_t launch_band(const KParams& p, int dtype, int tune, hipStream_t stream);  // render_band.hip
bool band_variant_supports(const KParams& p, int dtype);                    // render_band.hip
uint64_t band_workspace_bytes(const KParams& p, int dtype);                 // render_band.hip
uint32_t* band_gate_words(const KParams& p, int dtype);                     // render_band.hip
int band_pixels_wide(int dtype);                                            // render_band.hip
uint64_t backward_gather_workspace_bytes(const KParams& p);                 // render_backward_gather.hip

// ---- min/max of the normalised grid on the last plane (mpi.py:103-109 diagnostics) --------------
template <bool AC>
__global__ __launch_bounds__(256) void last_plane_uv_kernel(const KParams p, float* __restrict__ uv) {
    const int n = blockIdx.x;
    uint32_t bad_index = 0;  // (the forward reports a bad view index; here it is only clamped)
    const int m = view_mpi(p, n, bad_index);
    const float* dhw = p.dhw + (static_cast<int64_t>(m) * p.D + (p.D - 1)) * 3;
    const float d = dhw[0], ph = dhw[1], pw = dhw[2];
    const float ex = p.eye_pos[3 * n + 0], ey = p.eye_pos[3 * n + 1], ez = p.eye_pos[3 * n + 2];
    const float zdiff = d - ez;
    const int64_t HW = static_cast<int64_t>(p.H) * p.W;
    const float* rd = p.ray_dir + static_cast<int64_t>(n) * 3 * HW;
    const float inf = __builtin_inff();
    float mnu = inf, mxu = -inf, mnv = inf, mxv = -inf;
    bool nan = false;
    for (int64_t i = threadIdx.x; i < HW; i += blockDim.x) {
        float ix, iy, s, u, v;
        plane_coord<AC>(zdiff, ph, pw, ex, ey, rd[i], rd[HW + i], rd[2 * HW + i], 1.0f, 1.0f, ix, iy, s, u, v);
        nan |= (u != u) || (v != v);
        mnu = fminf(mnu, u), mxu = fmaxf(mxu, u), mnv = fminf(mnv, v), mxv = fmaxf(mxv, v);
    }
    __shared__ float red[4][256];
    __shared__ int red_nan;
    if (threadIdx.x == 0) red_nan = 0;
    red[0][threadIdx.x] = mnu, red[1][threadIdx.x] = mxu, red[2][threadIdx.x] = mnv, red[3][threadIdx.x] = mxv;
    __syncthreads();
    if (nan) atomicOr(&red_nan, 1);
    for (int o = 128; o > 0; o >>= 1) {
        if (static_cast<int>(threadIdx.x) < o) {
            red[0][threadIdx.x] = fminf(red[0][threadIdx.x], red[0][threadIdx.x + o]);
            red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + o]);
            red[2][threadIdx.x] = fminf(red[2][threadIdx.x], red[2][threadIdx.x + o]);
            red[3][threadIdx.x] = fmaxf(red[3][threadIdx.x], red[3][threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        const float q = __builtin_nanf("");
        uv[4 * n + threadIdx.x] = red_nan ? q : red[threadIdx.x][0];  // torch.min/max propagate NaN
    }
}

// ---- exhaustive [0,1] check (the reference's full min/max passes) ----------------------------------
template <typename T>
__global__ __launch_bounds__(256) void range_check_kernel(const T* __restrict__ v, int64_t count, uint32_t* status) {
    bool bad = false;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride)
        bad |= !in_unit(to_f32(v[i]));
    report_status(status, bad ? 2u : 0u);
}

// 16-byte vectorised body for contiguous, aligned volumes
template <typename T, int PER>
__global__ __launch_bounds__(256) void range_check_vec_kernel(const uint4* __restrict__ v, int64_t nvec,
                                                              uint32_t* status) {
    // A pure streaming read: non-temporal loads (7.0-7.4 TB/s against 6.3 for cached loads, profiles/r03_calibration.txt), four 16-byte
    // loads per lane in flight.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* __restrict__ vv = reinterpret_cast<const u32x4*>(v);
    bool bad = false;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    auto test = [&](const u32x4& q) {
        const T* e = reinterpret_cast<const T*>(&q);
#pragma unroll
        for (int j = 0; j < PER; ++j) bad |= !in_unit(to_f32(e[j]));
    };
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        const u32x4 q0 = __builtin_nontemporal_load(vv + i), q1 = __builtin_nontemporal_load(vv + i + stride);
        const u32x4 q2 = __builtin_nontemporal_load(vv + i + 2 * stride), q3 = __builtin_nontemporal_load(vv + i + 3 * stride);
        test(q0), test(q1), test(q2), test(q3);
    }
    for (; i < nvec; i += stride) test(__builtin_nontemporal_load(vv + i));
    report_status(status, bad ? 2u : 0u);
}

// ---- diagnostic: a pure streaming read (the ceiling bench.py quotes next to the render kernel's rate) -------------------------------
// Non-temporal 4-byte loads, 32 per lane in flight, 2 workgroups per CU: the fastest read pattern of profiles/r03_calibration.txt
// (7.0-7.4 TB/s on MI355X; cached loads and 16-byte loads stream at 6.3-7.1).
__global__ __launch_bounds__(256) void stream_probe_kernel(const uint32_t* __restrict__ v, int64_t nwords, uint32_t* sink) {
    uint32_t acc = 0;
    const int64_t trip = static_cast<int64_t>(256) * 32, stride = static_cast<int64_t>(gridDim.x) * trip;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * trip + threadIdx.x; i + 31 * 256 < nwords; i += stride) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc |= __builtin_nontemporal_load(v + i + u * 256);
    }
    if (acc == 0x9e3779b9u && sink != nullptr) sink[0] = acc;  // (keeps the loads alive; a volume of [0, 1] values never has this OR)
}

// ---- driver epilogue: float frames -> uint8 (render_video.py:118-126) -------------------------------
__global__ __launch_bounds__(256) void frames_to_uint8_kernel(const float* __restrict__ rgb, const float* __restrict__ dep,
                                                              int64_t HW, int64_t total, float dnear, float span,
                                                              uint8_t* __restrict__ img8, uint8_t* __restrict__ dep8) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;  // over N*H*W
    if (i >= total) return;
    const int64_t n = i / HW, q = i - n * HW;
    if (img8) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = rgb[(n * 3 + c) * HW + q];
            v = v + 1.0f;    // img = (img + 1) / 2.0
            v = v / 2.0f;
            v = v * 255.0f;  // (img * 255).astype(np.uint8): C truncation toward zero
            // astype(uint8) of an out-of-range float is UB in numpy; compositing keeps v in [0,255], clamp for safety
            v = fminf(fmaxf(v, 0.0f), 255.0f);
            img8[i * 3 + c] = static_cast<uint8_t>(static_cast<int>(v));
        }
    }
    if (dep8) {
        float v = dep[i];
        v = v - dnear;
        v = v / span;
        v = fminf(fmaxf(v, 0.0f), 1.0f);  // np.clip
        v = v * 255.0f;
        dep8[i] = static_cast<uint8_t>(static_cast<int>(v));
    }
}

// ---- rays of N views: R @ unit_dirs with the CPU sgemm's FMA order (camera.py:189-211) -----------------------
__global__ __launch_bounds__(256) void generate_rays_kernel(const float* __restrict__ c2w, const float* __restrict__ dirs,
                                                            int64_t HW, float* __restrict__ ray, float* __restrict__ eye,
                                                            float* __restrict__ zdir) {
    const int n = blockIdx.y;
    const float* m = c2w + 16 * n;
    const float r00 = m[0], r01 = m[1], r02 = m[2], r10 = m[4], r11 = m[5], r12 = m[6], r20 = m[8], r21 = m[9], r22 = m[10];
    if (blockIdx.x == 0 && threadIdx.x < 3) {
        eye[3 * n + threadIdx.x] = m[4 * threadIdx.x + 3];
        zdir[3 * n + threadIdx.x] = m[4 * threadIdx.x + 2];
    }
    const int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float d0 = dirs[p], d1 = dirs[HW + p], d2 = dirs[2 * HW + p];
    float* o = ray + static_cast<int64_t>(n) * 3 * HW + p;
    o[0] = __builtin_fmaf(r02, d2, __builtin_fmaf(r01, d1, r00 * d0));
    o[HW] = __builtin_fmaf(r12, d2, __builtin_fmaf(r11, d1, r10 * d0));
    o[2 * HW] = __builtin_fmaf(r22, d2, __builtin_fmaf(r21, d1, r20 * d0));
}

// ---- LightRenderer.compute_depth (light_renderer.py:82-100): composite plane depths with the un-warped alphas ----
template <typename T>
__global__ __launch_bounds__(256) void alpha_depth_kernel(const T* __restrict__ alpha, int64_t sb, int64_t sd, int64_t sr,
                                                          const float* __restrict__ ds, int D, int H, int W,
                                                          float* __restrict__ depth, float* __restrict__ tout) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const T* a = alpha + b * sb + y * sr + x;
    float Tr = 1.0f, Z = 0.0f;
#pragma unroll 4
    for (int k = 0; k < D; ++k) {
        const float al = to_f32(a[k * sd]);
        const float w = al * Tr;           // weights = alpha * cumprod[:-1]
        const float wd = w * ds[k];
        Z = Z + wd;                        // torch.sum(weights * plane_ds, dim=1)
        float om = 1.0f - al;
        om = om + 1e-10f;
        Tr = Tr * om;
    }
    const int64_t o = (static_cast<int64_t>(b) * H + y) * W + x;
    depth[o] = Z;
    if (tout) tout[o] = Tr;
}

// ---- self-test of div_by_recip (gmpi_device.hpp): q = n / d through RN(1/d) against the IEEE division --------------
// Operand classes: 0 = zdiff / ray_z (plane distance minus eye height over the z component of a unit ray), 1 = x / (w/2)
// (in-plane position over a plane half extent), 2 = full-range significands and exponents within +-20, 3 = divisors of the
// one significand pattern the correction step's proof treats separately, d = 2^k (2 - 2^-23).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // lowbias32
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float unit_float(uint32_t h) { return __uint_as_float(0x3f800000u | (h >> 9)) - 1.0f; }  // [0, 1)
__global__ __launch_bounds__(256) void selftest_division_kernel(uint64_t pairs, uint32_t seed, unsigned long long* mism) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned int bad[4] = {0, 0, 0, 0};
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        const uint32_t lo = static_cast<uint32_t>(i), hi = static_cast<uint32_t>(i >> 32);
        // (the seed enters b non-bijectively: with a = mix32(lo ^ seed) alone two seeds would test permutations of the SAME operand pairs)
        const uint32_t a = mix32(lo ^ seed), b = mix32(a + mix32(seed + 0x5bd1e995u) * 0x9e3779b9u + hi * 0x85ebca6bu + 0x85ebca6bu), c = mix32(b ^ 0xc2b2ae35u);
        const int cls = static_cast<int>(c & 3u);
        float n, d;
        if (cls == 0) {
            n = (unit_float(a) - 0.5f) * 4.0f;             // zdiff in [-2, 2)
            d = 0.35f + 0.65f * unit_float(b);             // ray_z of a unit ray inside a < 70 degree cone
            if (c & 4u) d = -d;
        } else if (cls == 1) {
            n = (unit_float(a) - 0.5f) * 2.0f;             // x in [-1, 1)
            d = 0.02f + 2.0f * unit_float(b);              // half extent of a plane
        } else if (cls == 2) {
            n = __uint_as_float((a & 0x807fffffu) | ((107u + (a >> 23) % 41u) << 23));   // exponent in [-20, 20]
            d = __uint_as_float((b & 0x807fffffu) | ((107u + (b >> 23) % 41u) << 23));
        } else {
            n = __uint_as_float((a & 0x807fffffu) | ((107u + (a >> 23) % 41u) << 23));
            d = __uint_as_float(0x007fffffu | ((107u + (b >> 23) % 41u) << 23) | (b & 0x80000000u));  // 2^k (2 - 2^-23)
        }
        const float r = 1.0f / d;
        const float q = div_by_recip(n, d, r);
        const float q_ref = n / d;
        if (__float_as_uint(q) != __float_as_uint(q_ref) && !(q == 0.0f && q_ref == 0.0f)) bad[cls]++;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned int v = bad[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(mism + k, static_cast<unsigned long long>(v));
    }
}

static int to_kparams(const GmpiRenderParams* q, KParams& p, bool need_outputs, bool need_volume = false) {
    if (q == nullptr) return GMPI_E_NULL;
    if (q->struct_size != sizeof(GmpiRenderParams)) return GMPI_E_ABI;
    if (q->flags & ~static_cast<uint32_t>(GMPI_FLAG_ALL)) return GMPI_E_FLAGS;  // undefined bits never reach a kernel
    if (q->N < 0 || q->M <= 0 || q->D <= 0 || q->Ht <= 0 || q->Wt <= 0 || q->H <= 0 || q->W <= 0) return GMPI_E_SHAPE;
    if (q->view_to_mpi == nullptr) {
        if (q->views_per_mpi < 1) return GMPI_E_SHAPE;
        if (q->N > static_cast<int64_t>(q->M) * q->views_per_mpi) return GMPI_E_SHAPE;
    }
    if (q->dhw == nullptr || q->ray_dir == nullptr || q->eye_pos == nullptr) return GMPI_E_NULL;
    if (need_outputs && (q->rgb_out == nullptr || q->depth_out == nullptr)) return GMPI_E_NULL;
    if (need_outputs || need_volume) {
        if (q->rgba == nullptr || q->z_dir == nullptr) return GMPI_E_NULL;
        if (q->rgba_dtype < GMPI_DTYPE_F32 || q->rgba_dtype > GMPI_DTYPE_F16) return GMPI_E_DTYPE;
        if (q->rgba_stride[4] != 1) return GMPI_E_STRIDE;
        for (int i = 0; i < 4; ++i)
            if (q->rgba_stride[i] < 0) return GMPI_E_STRIDE;
        if (q->rgba_stride[3] < q->Wt || q->rgba_stride[2] == 0 || q->rgba_stride[1] == 0) return GMPI_E_STRIDE;
    }
    p.rgba = q->rgba;
    p.view_to_mpi = q->view_to_mpi;
    p.dhw = q->dhw;
    p.ray_dir = q->ray_dir;
    p.eye_pos = q->eye_pos;
    p.z_dir = q->z_dir;
    p.rgb_out = q->rgb_out;
    p.depth_out = q->depth_out;
    p.T_out = q->transmittance_out;
    p.status = q->status;
    p.s_mpi = q->rgba_stride[0];
    p.s_plane = q->rgba_stride[1];
    p.s_chan = q->rgba_stride[2];
    p.s_row = q->rgba_stride[3];
    p.N = q->N, p.M = q->M, p.D = q->D, p.Ht = q->Ht, p.Wt = q->Wt, p.H = q->H, p.W = q->W;
    p.views_per_mpi = q->view_to_mpi ? 1 : q->views_per_mpi;
    p.flags = q->flags;
    p.ws = q->workspace;
    p.ws_bytes = q->workspace != nullptr ? q->workspace_bytes : 0;
    p.gate = nullptr, p.gate_gen = 0, p.gate_sense = 0;
    p.band_cols = 1, p.band_rot = 0, p.band_split = 1, p.band_tail = 0;
    return GMPI_OK;
}

// bf16 / fp16: bands (of 256 x 8 pixels) from which AUTO takes the band kernel.  Round 5: 256 = one workgroup on every CU (rounds 3-4: 512 = two).
// At 256 bands (two views of 512^2, eight of 256^2) the band kernel takes 0.115-0.125 ms where the strip / tile kernels take 0.140-0.146, whatever the
// frontal hint says; at 128 bands it loses (0.122-0.133 against 0.078): profiles/r05_small_launches.txt.
constexpr int64_t kAutoBandMin = 256;
constexpr int64_t kAutoBandMinF32 = 1024;  // fp32: bands of 128 x 8 pixels (at 512 -- config 2 -- the strip kernel is as fast: profiles/r03_band_variants.txt)
// The 256-band threshold was measured in DEFAULT mode with frontal-to-moderate cameras (profiles/r05_small_launches.txt).  Two kinds of 16-bit launches keep
// rounds 3-4's threshold of 512 bands: strict-order launches (never measured below it: the strip kernel keeps its 2^18..2^19-pixel range) and launches whose
// caller says that some camera is tilted beyond 0.53 rad (a view the band kernel cannot stage then costs a table kernel and an empty band launch on top).
constexpr int64_t kAutoBandMinUnmeasured = 512;
constexpr int64_t kAutoBandMinShared = 1024, kAutoBandMinSharedF32 = 2048;   // views that share MPIs (config 4's launch, the size that was measured)

// does GMPI_VARIANT_AUTO consider the band kernel for this launch (given a workspace and the band kernel's alignment preconditions)?
static int64_t band_count(const KParams& p, int dtype) {
    const int bw = band_pixels_wide(dtype);
    return static_cast<int64_t>(p.N) * ((p.W + bw - 1) / bw) * ((p.H + 7) / 8);
}
static bool auto_takes_band(const KParams& p, int dtype) {
    if (dtype != GMPI_DTYPE_BF16 && dtype != GMPI_DTYPE_F32 && dtype != GMPI_DTYPE_F16) return false;
    // Views that share one MPI (views_per_mpi > 1: the video paths).  Rounds 3-5 kept them on the tile kernel after a measurement -- config 4 through the band
    // kernel + gate: 1.14 ms against 0.545 -- that was an artefact of the gated tile kernel's loop stride (kGatedGrid, render_lds.hip).  Both kernels interleave
    // such views per band / tile position and read the volume from HBM about once per group; with every view fitting, the band kernel is faster: 8 views of one
    // 512^2 x 96 MPI, yaw within +-0.25 rad: 0.521 -> 0.474 ms (fp32), 0.427 -> 0.340 (bf16); +-0.35: 0.528 -> 0.495, 0.427 -> 0.346.  A group with a view that
    // does NOT fit goes to the tile kernel as a whole (the table kernel gates every view of the group: two small tile launches next to a thinned band launch
    // lose, 0.78 against 0.545) -- which a caller that knows its cameras spares the launch by saying so (GMPI_FLAG_HINT_OBLIQUE / _TILTED).
    if (p.view_to_mpi == nullptr && p.views_per_mpi > 1) {
        if ((p.flags & (GMPI_FLAG_HINT_OBLIQUE | GMPI_FLAG_HINT_TILTED)) != 0) return false;
#ifdef GMPI_TUNE
        static const int env_shared = [] { const char* e = getenv("GMPI_TUNE_SHARED"); return e ? atoi(e) : 1; }();   // (0: the old routing, for A/B runs)
        if (env_shared == 0) return false;
#endif
    }
    const int bw = band_pixels_wide(dtype);
    const int64_t cols = (p.W + bw - 1) / bw, rows = (p.H + 7) / 8;
    // (an image that fills less than 3/4 of its bands -- narrower than a band, a ragged last column -- wastes the idle lanes' issue slots:
    //  the tile kernel's 32 x 16 tiles fit such images better)
    if (static_cast<int64_t>(p.W) * 4 < cols * bw * 3 || static_cast<int64_t>(p.H) * 4 < rows * 8 * 3) return false;
    int64_t need = dtype == GMPI_DTYPE_F32 ? kAutoBandMinF32 : kAutoBandMin;
    // (views that share MPIs: only launches of the size that was measured, config 4's -- 8 views of 512^2: 1024 bands of 256 x 8 pixels / 2048 of 128 x 8.  Small
    //  shared launches belong to the tile kernel, whose interleaved tiles keep the one volume in the L2s: 8 views of 256^2 or 2 of 512^2 over a bf16 volume, 256 bands:
    //  0.118 ms against the band kernel's 0.129 (frontal) / 0.143 (0.3 rad of yaw))
    if (p.view_to_mpi == nullptr && p.views_per_mpi > 1) need = dtype == GMPI_DTYPE_F32 ? kAutoBandMinSharedF32 : kAutoBandMinShared;
    // (round 6: GMPI_FLAG_HINT_OBLIQUE counts like _TILTED here -- a camera beyond 0.35 rad is one the band kernel cannot stage at these sizes: 2 views of 512^2 / 8 of
    //  256^2 at 0.45 rad of yaw went through the table kernel, an empty band launch and the gated tile launch, 0.168-0.174 ms against the tile kernel's 0.141-0.145)
    if (dtype != GMPI_DTYPE_F32 && (p.flags & (GMPI_FLAG_STRICT_ORDER | GMPI_FLAG_HINT_TILTED | GMPI_FLAG_HINT_OBLIQUE)) != 0) need = kAutoBandMinUnmeasured;
    return band_count(p, dtype) >= need;
}

static int hip_rc(hipError_t e) { return e == hipSuccess ? GMPI_OK : GMPI_E_LAUNCH - static_cast<int>(e); }

}  // namespace gmpi

using namespace gmpi;

extern "C" {

int gmpi_mpi_render_launch(const GmpiRenderParams* params, void* stream) {
    if (params != nullptr && params->struct_size == sizeof(GmpiRenderParams) && params->N == 0) return GMPI_OK;  // no views
    KParams p;
    const int rc = to_kparams(params, p, true);
    if (rc != GMPI_OK) return rc;
    if (p.N == 0) return GMPI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int variant = params->variant;
    if (variant == GMPI_VARIANT_AUTO) {
        // Measured on MI355X at steady clocks (profiles/r02_variants.txt, "small launches"): the tile kernel wins on large
        // launches (configs 3-5: 32 waves per CU hide its latencies, its shared 32x16 boxes tolerate tilted cameras); the strip
        // kernel wins where the launch under-fills the chip: up to 512 strips of 32x8 pixels (its waves split the planes 6-way),
        // 16-bit volumes up to 1024 strips (3-way), and fp32 volumes between 1537 and 2048 strips (config 2, 8 views of 256^2:
        // the tile kernel needs a second round of workgroups there, 3 x 53 KB of LDS per CU).  Strict-order mode has no plane
        // split: there only launches of 2^18 .. 2^19 pixels go to the strip kernel.
        const int64_t pixels = static_cast<int64_t>(p.N) * p.H * p.W;
        const int64_t strips = static_cast<int64_t>(p.N) * ((p.W + 31) / 32) * ((p.H + 7) / 8);
        const bool strict = (p.flags & GMPI_FLAG_STRICT_ORDER) != 0;
        const bool lds_ok = lds_variant_supports(p, params->rgba_dtype), wave_ok = wave_variant_supports(p, params->rgba_dtype);
        // 16-bit volumes between 513 and 2048 strips: the strip kernel wins by 4-7 % under a frontal camera, the tile kernel by 3-10 % (and
        // more above 1024 strips) from 0.3 rad of yaw on (profiles/r03_pose_sweep.txt) -- the caller's GMPI_FLAG_HINT_FRONTAL decides;
        // without it: tilted.
        const bool frontal = (p.flags & GMPI_FLAG_HINT_FRONTAL) != 0;
        // ... and beyond 0.53 rad of tilt (GMPI_FLAG_HINT_TILTED) the strip kernel's wave-private boxes overflow into half strips and the direct
        // gather: config 2 takes 0.25-0.77 ms instead of 0.16 where the tile kernel stays at 0.21 (profiles/r04_pose_distribution.txt)
        const bool tilted = (p.flags & GMPI_FLAG_HINT_TILTED) != 0;
        // ... and views that SHARE MPIs (round 6): the tile kernel interleaves them per tile, the volume comes from HBM once per group -- 16-bit volumes, 8 views of
        // 256^2 or 2 of 512^2 (2048 strips) under a frontal camera: 0.118 ms against the strip kernel's 0.145 (fp32 volumes: the strip kernel stays ahead, 0.130 / 0.152)
        const bool shared = p.view_to_mpi == nullptr && p.views_per_mpi > 1;
        const bool small = strict ? (pixels <= (int64_t(1) << 19) && pixels > (int64_t(1) << 18))
                         : params->rgba_dtype == GMPI_DTYPE_F32 ? (strips <= 512 || (strips > 1536 && strips <= 2048 && !tilted))
                                                                : (strips <= 512 || (strips <= 2048 && frontal && !shared));
        variant = (wave_ok && (small || !lds_ok)) ? GMPI_VARIANT_WAVE : lds_ok ? GMPI_VARIANT_LDS : GMPI_VARIANT_GATHER;
        // (a launch the band kernel takes -- below -- is not "small", whatever the hints say: 16-bit volumes reach the band threshold at exactly the
        //  2048 strips up to which a frontal hint would pick the strip kernel)
        // (auto_takes_band keeps 16-bit launches of 256-511 bands off the band path in strict-order mode and when the caller says that some camera is
        //  tilted beyond 0.53 rad: a view the band kernel cannot stage goes to the tile kernel behind a table kernel and an empty band launch, ~18 us
        //  that such a small launch feels -- two views of 512^2 at 0.45 rad of yaw: 0.158 ms against the tile kernel's 0.140; the frontal ones 0.120 against 0.143)
        const bool band_path = lds_ok && auto_takes_band(p, params->rgba_dtype) && band_variant_supports(p, params->rgba_dtype);
        if (band_path) variant = GMPI_VARIANT_LDS;
        // Large launches over bf16 / fp32 volumes, when the caller lends a workspace: the band kernel (256 x 8 / 128 x 8 pixel bands, LDS-DMA;
        // 0.81 ms on BASELINE config 3 where the tile kernel takes 1.02, 1.21 against 1.33 with an fp32 volume) -- for the views it can stage.  Whether a view's texel boxes fit the band
        // kernel's buffers depends on the camera (tilt shears the boxes) and is only known on the device, so AUTO launches BOTH kernels and
        // lets the band kernel's table kernel share out the views through a gate word per view (KParams::gate): views with a box that does
        // not fit fall to the tile kernel, the others' tile workgroups exit at once (an empty second launch costs a few microseconds).
        if (band_path) {
            static std::atomic<uint32_t> gate_counter{0x6d2b79f5u};
            uint32_t gen = gate_counter.fetch_add(1u, std::memory_order_relaxed);
            KParams pb = p;
            pb.gate = band_gate_words(p, params->rgba_dtype), pb.gate_gen = gen, pb.gate_sense = 0u;
            const hipError_t e = launch_band(pb, params->rgba_dtype, 0, st);
            if (e != hipSuccess) return hip_rc(e);
            pb.gate_sense = 1u;
            return hip_rc(launch_lds(pb, params->rgba_dtype, 0, st));
        }
    }
    if (variant == GMPI_VARIANT_GATHER) {
        if (p.N > 65535) return GMPI_E_SHAPE;  // the gather kernel puts the view index in grid.z
        return hip_rc(launch_gather(p, params->rgba_dtype, st));
    }
    int tune = 0;
#ifdef GMPI_TUNE  // profiling builds only (make EXTRA=-DGMPI_TUNE): experiment knobs from the environment
    static const int env_tune = [] { const char* e = getenv("GMPI_TUNE_WAVE"); return e ? atoi(e) : 0; }();
    tune = env_tune;
#endif
    if (variant == GMPI_VARIANT_WAVE) {
        if (!wave_variant_supports(p, params->rgba_dtype)) return GMPI_E_VARIANT;
        return hip_rc(launch_wave(p, params->rgba_dtype, tune, st));
    }
    if (variant == GMPI_VARIANT_LDS) {
        if (!lds_variant_supports(p, params->rgba_dtype)) return GMPI_E_VARIANT;
        return hip_rc(launch_lds(p, params->rgba_dtype, tune, st));
    }
    if (variant == GMPI_VARIANT_BAND) {
        if (!band_variant_supports(p, params->rgba_dtype)) return GMPI_E_VARIANT;
        return hip_rc(launch_band(p, params->rgba_dtype, tune, st));
    }
    // (GMPI_VARIANT_DMA -- round 3's LDS-DMA loader inside the 32 x 16 tile decomposition, the A/B that separated the loader from the
    //  decomposition -- was retired in round 4: the band kernel is that loader's home.  The value stays reserved and is refused.)
    return GMPI_E_VARIANT;
}

uint64_t gmpi_render_workspace_bytes(const GmpiRenderParams* params) {
    KParams p;
    if (to_kparams(params, p, true) != GMPI_OK || p.N == 0) return 0;
    if (params->variant == GMPI_VARIANT_BAND) return band_workspace_bytes(p, params->rgba_dtype);
    if (params->variant == GMPI_VARIANT_AUTO && auto_takes_band(p, params->rgba_dtype)) return band_workspace_bytes(p, params->rgba_dtype);
    return 0;
}

uint64_t gmpi_render_backward_workspace_bytes(const GmpiRenderParams* params) {
    KParams p;
    if (to_kparams(params, p, false, true) != GMPI_OK || p.N == 0) return 0;
    if (params->variant == GMPI_VARIANT_GATHER) return 0;   // the all-atomic cross-check kernel
    return backward_gather_workspace_bytes(p);
}

int gmpi_mpi_render_backward_launch(const GmpiRenderParams* params, const float* grad_rgb, const float* grad_depth,
                                    float* grad_rgba, const int64_t* grad_rgba_stride, void* stream) {
    if (params != nullptr && params->struct_size == sizeof(GmpiRenderParams) && params->N == 0) return GMPI_OK;
    KParams p;
    const int rc = to_kparams(params, p, false, true);
    if (rc != GMPI_OK) return rc;
    if (grad_rgb == nullptr || grad_rgba == nullptr || grad_rgba_stride == nullptr) return GMPI_E_NULL;
    if (p.N > 65535) return GMPI_E_SHAPE;  // both backward kernels put the view index in grid.y / grid.z: split the batch
    if (grad_rgba_stride[4] != 1) return GMPI_E_STRIDE;
    for (int i = 0; i < 4; ++i)
        if (grad_rgba_stride[i] <= 0 && !(i == 0 && params->M == 1)) return GMPI_E_STRIDE;
    return hip_rc(launch_backward(p, params->rgba_dtype, grad_rgb, grad_depth, grad_rgba, grad_rgba_stride,
                                  params->variant != GMPI_VARIANT_GATHER, static_cast<hipStream_t>(stream)));
}

int gmpi_last_plane_uv_minmax_launch(const GmpiRenderParams* params, float* uv_minmax, void* stream) {
    KParams p;
    const int rc = to_kparams(params, p, false);
    if (rc != GMPI_OK) return rc;
    if (uv_minmax == nullptr) return GMPI_E_NULL;
    if (p.N == 0) return GMPI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p.flags & GMPI_FLAG_ALIGN_CORNERS) hipLaunchKernelGGL(last_plane_uv_kernel<true>, dim3(p.N), dim3(256), 0, st, p, uv_minmax);
    else hipLaunchKernelGGL(last_plane_uv_kernel<false>, dim3(p.N), dim3(256), 0, st, p, uv_minmax);
    return hip_rc(hipGetLastError());
}

int gmpi_rgba_range_check_launch(const void* rgba, int32_t rgba_dtype, int64_t count, uint32_t* status, void* stream) {
    if (rgba == nullptr || status == nullptr) return GMPI_E_NULL;
    if (count < 0) return GMPI_E_SHAPE;
    if (rgba_dtype < GMPI_DTYPE_F32 || rgba_dtype > GMPI_DTYPE_F16) return GMPI_E_DTYPE;
    if (count == 0) return GMPI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int esz = rgba_dtype == GMPI_DTYPE_F32 ? 4 : 2;
    const int per = 16 / esz;
    const bool aligned = (reinterpret_cast<uintptr_t>(rgba) % 16) == 0;
    const int64_t nvec = aligned ? count / per : 0;
    const int64_t tail = count - nvec * per;
    if (nvec > 0) {
        const int blocks = static_cast<int>(std::min<int64_t>((nvec + 255) / 256, 256 * 8));
        const uint4* v = static_cast<const uint4*>(rgba);
        if (rgba_dtype == GMPI_DTYPE_F32) hipLaunchKernelGGL((range_check_vec_kernel<float, 4>), dim3(blocks), dim3(256), 0, st, v, nvec, status);
        else if (rgba_dtype == GMPI_DTYPE_BF16) hipLaunchKernelGGL((range_check_vec_kernel<bf16_t, 8>), dim3(blocks), dim3(256), 0, st, v, nvec, status);
        else hipLaunchKernelGGL((range_check_vec_kernel<f16_t, 8>), dim3(blocks), dim3(256), 0, st, v, nvec, status);
    }
    if (tail > 0) {
        const char* base = static_cast<const char*>(rgba) + nvec * 16;
        const int blocks = static_cast<int>(std::min<int64_t>((tail + 255) / 256, 256 * 8));
        if (rgba_dtype == GMPI_DTYPE_F32) hipLaunchKernelGGL(range_check_kernel<float>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float*>(base), tail, status);
        else if (rgba_dtype == GMPI_DTYPE_BF16) hipLaunchKernelGGL(range_check_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(base), tail, status);
        else hipLaunchKernelGGL(range_check_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const f16_t*>(base), tail, status);
    }
    return hip_rc(hipGetLastError());
}

int gmpi_frames_to_uint8_launch(const float* rgb_pm1, const float* depth, int32_t N, int32_t H, int32_t W, double depth_near,
                                double depth_far, uint8_t* img8, uint8_t* dep8, void* stream) {
    if (N < 0 || H <= 0 || W <= 0) return GMPI_E_SHAPE;
    if ((img8 && !rgb_pm1) || (dep8 && !depth)) return GMPI_E_NULL;
    if (N == 0 || (!img8 && !dep8)) return GMPI_OK;
    const int64_t HW = static_cast<int64_t>(H) * W, total = HW * N;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(frames_to_uint8_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       rgb_pm1, depth, HW, total, static_cast<float>(depth_near), static_cast<float>(depth_far - depth_near), img8,
                       dep8);
    return hip_rc(hipGetLastError());
}

int gmpi_generate_rays_launch(const float* c2w, const float* unit_dirs, int32_t N, int32_t H, int32_t W, float* ray_dir,
                              float* eye_pos, float* z_dir, void* stream) {
    if (N < 0 || H <= 0 || W <= 0) return GMPI_E_SHAPE;
    if (N == 0) return GMPI_OK;
    if (!c2w || !unit_dirs || !ray_dir || !eye_pos || !z_dir) return GMPI_E_NULL;
    const int64_t HW = static_cast<int64_t>(H) * W;
    const dim3 grid(static_cast<unsigned>((HW + 255) / 256), static_cast<unsigned>(N));
    hipLaunchKernelGGL(generate_rays_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), c2w, unit_dirs, HW, ray_dir,
                       eye_pos, z_dir);
    return hip_rc(hipGetLastError());
}

int gmpi_alpha_depth_launch(const void* alpha, int32_t alpha_dtype, int64_t stride_b, int64_t stride_d, int64_t stride_row,
                            const float* plane_ds, int32_t B, int32_t D, int32_t H, int32_t W, float* depth_out,
                            float* transmittance_out, void* stream) {
    if (B < 0 || D <= 0 || H <= 0 || W <= 0) return GMPI_E_SHAPE;
    if (B == 0) return GMPI_OK;
    if (!alpha || !plane_ds || !depth_out) return GMPI_E_NULL;
    if (alpha_dtype < GMPI_DTYPE_F32 || alpha_dtype > GMPI_DTYPE_F16) return GMPI_E_DTYPE;
    if (stride_b < 0 || stride_d <= 0 || stride_row < W) return GMPI_E_STRIDE;
    const dim3 grid((W + 255) / 256, H, B), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (alpha_dtype == GMPI_DTYPE_F32)
        hipLaunchKernelGGL(alpha_depth_kernel<float>, grid, block, 0, st, static_cast<const float*>(alpha), stride_b, stride_d, stride_row, plane_ds, D, H, W, depth_out, transmittance_out);
    else if (alpha_dtype == GMPI_DTYPE_BF16)
        hipLaunchKernelGGL(alpha_depth_kernel<bf16_t>, grid, block, 0, st, static_cast<const bf16_t*>(alpha), stride_b, stride_d, stride_row, plane_ds, D, H, W, depth_out, transmittance_out);
    else
        hipLaunchKernelGGL(alpha_depth_kernel<f16_t>, grid, block, 0, st, static_cast<const f16_t*>(alpha), stride_b, stride_d, stride_row, plane_ds, D, H, W, depth_out, transmittance_out);
    return hip_rc(hipGetLastError());
}

int gmpi_selftest_division_launch(uint64_t pairs, uint32_t seed, uint64_t* mismatches, void* stream) {
    if (mismatches == nullptr) return GMPI_E_NULL;
    if (pairs == 0) return GMPI_OK;
    const uint64_t want = (pairs + 255) / 256;
    const unsigned blocks = static_cast<unsigned>(want < 256u * 64u ? want : 256u * 64u);
    hipLaunchKernelGGL(selftest_division_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pairs, seed,
                       reinterpret_cast<unsigned long long*>(mismatches));
    return hip_rc(hipGetLastError());
}

int gmpi_stream_probe_launch(const void* buf, uint64_t bytes, uint32_t* sink, void* stream) {
    if (buf == nullptr) return GMPI_E_NULL;
    if (reinterpret_cast<uintptr_t>(buf) % 4 != 0) return GMPI_E_STRIDE;
    if (bytes < 4) return GMPI_OK;
    hipLaunchKernelGGL(stream_probe_kernel, dim3(256 * 2), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint32_t*>(buf),
                       static_cast<int64_t>(bytes / 4), sink);
    return hip_rc(hipGetLastError());
}

int gmpi_query(int32_t what) {
    switch (what) {
        case 0: return GMPI_ABI_VERSION;
        case 1: return static_cast<int>(sizeof(GmpiRenderParams));
        case 2: return 950;
        case 3: case 4: case 5: return lds_variant_query(what);
        case 6: return 1;  // GMPI_VARIANT_WAVE is built in
        case 7: return 0;  // GMPI_VARIANT_DMA: retired in round 4 (reserved value, refused with GMPI_E_VARIANT)
        case 8: return 1;  // GMPI_VARIANT_BAND is built in
        case 9: return static_cast<int>(kAutoBandMin);
        case 10: return static_cast<int>(kAutoBandMinF32);
        case 11: return 1;  // the atomics-free backward (pixel pass + texel gather) is built in
        default: return -1;
    }
}

const char* gmpi_version_string(void) { return "ml-gmpi_amd 0.3 (gfx950, ABI 2)"; }

}  // extern "C"
