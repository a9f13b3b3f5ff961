// light_kernels.hip -- the reference's shading augmentation (gmpi/core/light_renderer.py `LightRenderer.render`)
// after `compute_depth` (alpha_depth_kernel in gmpi_abi.hip):
//   gaussian_blur_kernel  = torchvision.transforms.GaussianBlur on the depth image (light_renderer.py:51-55, 109),
//   light_shading_kernel  = point cloud from the last plane's texel coordinates (compute_pcl :102-120), normals from
//                           the four neighbour cross products (get_normal :57-80), Lambert term and ka + kd*diffuse
//                           (:163-190) -> one shading factor per texel,
//   light_apply_kernel    = clip(rgb * shading, 0, 1), alpha passed through, over the whole volume (:193-198).
// The first two work on B*H*W images (tiny); the third streams the RGBA volume once (read + write).
#include "gmpi_device.hpp"

#include "../../include/gmpi_render.h"

namespace gmpi {

constexpr float kLightEps = 1e-8f;  // EPS of light_renderer.py:8

__device__ __forceinline__ int reflect_index(int i, int n) {  // padding_mode="reflect": -1 -> 1, n -> n-2
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// out[b,y,x] = sum_{dy,dx} in[b, refl(y+dy-r), refl(x+dx-r)] * (k1d[dy] * k1d[dx])   (kernel2d = k1d^T k1d as torchvision)
__global__ __launch_bounds__(256) void gaussian_blur_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                            const float* __restrict__ k1d, int ksize) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const float* __restrict__ img = in + static_cast<int64_t>(blockIdx.z) * H * W;
    const int r = ksize / 2;
    float acc = 0.0f;
    for (int dy = 0; dy < ksize; ++dy) {
        const float* __restrict__ row = img + static_cast<int64_t>(reflect_index(y + dy - r, H)) * W;
        const float wy = k1d[dy];
        for (int dx = 0; dx < ksize; ++dx) acc += row[reflect_index(x + dx - r, W)] * (wy * k1d[dx]);
    }
    out[(static_cast<int64_t>(blockIdx.z) * H + y) * W + x] = acc;
}

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__global__ __launch_bounds__(256) void light_shading_kernel(const float* __restrict__ depth, const float* __restrict__ xyz_last,
                                                            const float* __restrict__ light_dir, float ka, float kd, int H, int W,
                                                            float* __restrict__ shading) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= W) return;
    const float* __restrict__ d = depth + static_cast<int64_t>(b) * H * W;
    // normals exist for the interior and are replicate-padded to the border (light_renderer.py:73)
    const int yc = min(max(y, 1), H - 2), xc = min(max(x, 1), W - 2);
    auto pcl = [&](int yy, int xx) {
        const float* __restrict__ p = xyz_last + (static_cast<int64_t>(yy) * W + xx) * 3;
        const float scale = d[static_cast<int64_t>(yy) * W + xx] / (p[2] + kLightEps);
        return V3{p[0] * scale, p[1] * scale, p[2] * scale};
    };
    const V3 c = pcl(yc, xc), up = pcl(yc - 1, xc), down = pcl(yc + 1, xc), left = pcl(yc, xc - 1), right = pcl(yc, xc + 1);
    V3 n = cross(up - c, left - c) + cross(left - c, down - c);
    n = n + cross(down - c, right - c);
    n = n + cross(right - c, up - c);
    float len2 = n.x * n.x;
    len2 = len2 + n.y * n.y;
    len2 = len2 + n.z * n.z;
    const float len = sqrtf(len2) + kLightEps;
    n = {n.x / len, n.y / len, n.z / len};
    const float* __restrict__ l = light_dir + 3 * b;
    float dotv = n.x * l[0];
    dotv = dotv + n.y * l[1];
    dotv = dotv + n.z * l[2];
    const float diffuse = fmaxf(-1.0f * dotv, 0.0f);
    shading[(static_cast<int64_t>(b) * H + y) * W + x] = ka + diffuse * kd;
}

// VEC consecutive texels of a row per thread (4: 16-byte loads for fp32 / 8-byte for 16-bit storage, 16-byte stores)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void light_apply_kernel(const T* __restrict__ rgba, int64_t sb, int64_t sd, int64_t sc, int64_t sr,
                                                          const float* __restrict__ shading, float* __restrict__ out, int D, int H,
                                                          int W) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * VEC, y = blockIdx.y;
    if (x >= W) return;
    const int b = blockIdx.z / D, k = blockIdx.z - b * D;
    struct alignas(sizeof(T) * VEC) TV { T v[VEC]; };
    struct alignas(sizeof(float) * VEC) FV { float v[VEC]; };
    const FV s = *reinterpret_cast<const FV*>(shading + (static_cast<int64_t>(b) * H + y) * W + x);
    const T* __restrict__ src = rgba + b * sb + k * sd + static_cast<int64_t>(y) * sr + x;
    float* __restrict__ dst = out + ((static_cast<int64_t>(blockIdx.z) * 4) * H + y) * W + x;
    const int64_t plane = static_cast<int64_t>(H) * W;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const TV t = *reinterpret_cast<const TV*>(src + c * sc);
        FV o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o.v[i] = c < 3 ? fminf(fmaxf(to_f32(t.v[i]) * s.v[i], 0.0f), 1.0f) : to_f32(t.v[i]);
        *reinterpret_cast<FV*>(dst + c * plane) = o;
    }
}

template <typename T>
static void launch_light_apply(const void* rgba, const int64_t* st, const float* shading, float* out, int B, int D, int H, int W,
                               hipStream_t stream) {
    const T* src = static_cast<const T*>(rgba);
    const bool vec = W % 4 == 0 && st[0] % 4 == 0 && st[1] % 4 == 0 && st[2] % 4 == 0 && st[3] % 4 == 0 &&
                     reinterpret_cast<uintptr_t>(rgba) % (4 * sizeof(T)) == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(shading) % 16 == 0;
    if (vec) hipLaunchKernelGGL((light_apply_kernel<T, 4>), dim3((W / 4 + 255) / 256, H, B * D), dim3(256), 0, stream, src, st[0], st[1], st[2], st[3], shading, out, D, H, W);
    else hipLaunchKernelGGL((light_apply_kernel<T, 1>), dim3((W + 255) / 256, H, B * D), dim3(256), 0, stream, src, st[0], st[1], st[2], st[3], shading, out, D, H, W);
}

// ---- backward of the augmentation (the reference applies it inside the G-step, train.py:535-541, 703-709) -------------
// Only the two volume-sized ops need kernels; the B*H*W middle (blur, point cloud, normals, Lambert) is differentiated by
// torch autograd in light.py.
//
// (1) out = clip(rgb * s, 0, 1) | alpha:   g_rgb = g_out * s * [0 < rgb*s < 1],  g_alpha = g_out_alpha (more is added by (2)),
//     g_s[b,y,x] = sum_{k,c} g_out * rgb * [0 < rgb*s < 1].  One pixel column per thread, planes in a loop (coalesced in x).
//     torch.clip passes the gradient at the bounds themselves (min <= x <= max), so the mask is closed.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void light_apply_backward_kernel(const T* __restrict__ rgba, int64_t sb, int64_t sd, int64_t sc,
                                                                   int64_t sr, const float* __restrict__ shading,
                                                                   const float* __restrict__ g_out, float* __restrict__ g_rgba,
                                                                   float* __restrict__ g_shading, int D, int H, int W) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * VEC, y = blockIdx.y, b = blockIdx.z;
    if (x >= W) return;
    struct alignas(sizeof(T) * VEC) TV { T v[VEC]; };
    struct alignas(sizeof(float) * VEC) FV { float v[VEC]; };
    const int64_t plane = static_cast<int64_t>(H) * W, pix = static_cast<int64_t>(y) * W + x;
    const FV s = *reinterpret_cast<const FV*>(shading + b * plane + pix);
    FV gs;
#pragma unroll
    for (int i = 0; i < VEC; ++i) gs.v[i] = 0.0f;
    for (int k = 0; k < D; ++k) {
        const T* __restrict__ src = rgba + b * sb + k * sd + static_cast<int64_t>(y) * sr + x;
        const int64_t o = (static_cast<int64_t>(b) * D + k) * 4 * plane + pix;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const TV v = *reinterpret_cast<const TV*>(src + c * sc);
            const FV g = *reinterpret_cast<const FV*>(g_out + o + c * plane);
            FV r;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float vf = to_f32(v.v[i]), t = vf * s.v[i];
                const bool pass = t >= 0.0f && t <= 1.0f;
                r.v[i] = pass ? g.v[i] * s.v[i] : 0.0f;
                gs.v[i] += pass ? g.v[i] * vf : 0.0f;
            }
            *reinterpret_cast<FV*>(g_rgba + o + c * plane) = r;
        }
        *reinterpret_cast<FV*>(g_rgba + o + 3 * plane) = *reinterpret_cast<const FV*>(g_out + o + 3 * plane);
    }
    *reinterpret_cast<FV*>(g_shading + b * plane + pix) = gs;
}

// (2) depth = sum_k a_k T_k d_k (compute_depth): dL/da_k = g (T_k d_k - S_k / om_k), S_k = sum_{j>k} a_j T_j d_j, back to
//     front from the forward's final transmittance (same scheme as render_backward.hip: no cancellation behind opaque
//     planes; T carried as mantissa x 2^exponent).  ADDS into g_alpha (the alpha channel of the volume gradient).
template <typename T>
__global__ __launch_bounds__(256) void alpha_depth_backward_kernel(const T* __restrict__ alpha, int64_t sb, int64_t sd, int64_t sr,
                                                                   const float* __restrict__ ds, const float* __restrict__ t_final,
                                                                   const float* __restrict__ g_depth, float* __restrict__ g_alpha,
                                                                   int64_t gb, int64_t gd, int64_t gr, int D, int H, int W) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= W) return;
    const T* __restrict__ a = alpha + b * sb + static_cast<int64_t>(y) * sr + x;
    float* __restrict__ ga = g_alpha + b * gb + static_cast<int64_t>(y) * gr + x;
    const int64_t o = (static_cast<int64_t>(b) * H + y) * W + x;
    const float g = g_depth[o];
    float tm = t_final ? t_final[o] : 0.0f;
    int te = 0;
    if (!(tm >= 1e-30f)) {  // missing or underflowed: rebuild front to back in the extended representation
        tm = 1.0f;
        for (int k = 0; k < D; ++k) {
            tm *= (1.0f - to_f32(a[k * sd])) + 1e-10f;
            te += __builtin_amdgcn_frexp_expf(tm);
            tm = __builtin_amdgcn_frexp_mantf(tm);
        }
    }
    float S = 0.0f;
    for (int k = D - 1; k >= 0; --k) {
        const float al = to_f32(a[k * sd]);
        const float om = (1.0f - al) + 1e-10f;
        tm = tm / om;
        te += __builtin_amdgcn_frexp_expf(tm);
        tm = __builtin_amdgcn_frexp_mantf(tm);
        const float Tk = __builtin_amdgcn_ldexpf(tm, te);
        const float q = g * ds[k];
        ga[k * gd] += Tk * q - S / om;
        S += al * Tk * q;
    }
}

static int rc_of(hipError_t e) { return e == hipSuccess ? GMPI_OK : GMPI_E_LAUNCH - static_cast<int>(e); }

}  // namespace gmpi

using namespace gmpi;

extern "C" {

int gmpi_light_blur_launch(const float* depth, float* blurred, int32_t B, int32_t H, int32_t W, const float* kernel1d,
                           int32_t ksize, void* stream) {
    if (B < 0 || H <= 0 || W <= 0 || ksize <= 0 || (ksize & 1) == 0) return GMPI_E_SHAPE;
    if (ksize / 2 >= H || ksize / 2 >= W) return GMPI_E_SHAPE;  // reflect padding needs pad < size
    if (B == 0) return GMPI_OK;
    if (!depth || !blurred || !kernel1d) return GMPI_E_NULL;
    hipLaunchKernelGGL(gaussian_blur_kernel, dim3((W + 255) / 256, H, B), dim3(256), 0, static_cast<hipStream_t>(stream), depth,
                       blurred, H, W, kernel1d, ksize);
    return rc_of(hipGetLastError());
}

int gmpi_light_shading_launch(const float* depth_blurred, const float* xyz_last, const float* light_dir, float ka, float kd,
                              int32_t B, int32_t H, int32_t W, float* shading, void* stream) {
    if (B < 0 || H < 3 || W < 3) return GMPI_E_SHAPE;
    if (B == 0) return GMPI_OK;
    if (!depth_blurred || !xyz_last || !light_dir || !shading) return GMPI_E_NULL;
    hipLaunchKernelGGL(light_shading_kernel, dim3((W + 255) / 256, H, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       depth_blurred, xyz_last, light_dir, ka, kd, H, W, shading);
    return rc_of(hipGetLastError());
}

int gmpi_light_apply_launch(const void* rgba, int32_t rgba_dtype, const int64_t* rgba_stride, const float* shading, float* out,
                            int32_t B, int32_t D, int32_t H, int32_t W, void* stream) {
    if (B < 0 || D <= 0 || H <= 0 || W <= 0 || static_cast<int64_t>(B) * D > 65535) return GMPI_E_SHAPE;
    if (B == 0) return GMPI_OK;
    if (!rgba || !rgba_stride || !shading || !out) return GMPI_E_NULL;
    if (rgba_dtype < GMPI_DTYPE_F32 || rgba_dtype > GMPI_DTYPE_F16) return GMPI_E_DTYPE;
    if (rgba_stride[4] != 1 || rgba_stride[3] < W || rgba_stride[2] <= 0 || rgba_stride[1] <= 0 || rgba_stride[0] < 0) return GMPI_E_STRIDE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (rgba_dtype == GMPI_DTYPE_F32) launch_light_apply<float>(rgba, rgba_stride, shading, out, B, D, H, W, st);
    else if (rgba_dtype == GMPI_DTYPE_BF16) launch_light_apply<bf16_t>(rgba, rgba_stride, shading, out, B, D, H, W, st);
    else launch_light_apply<f16_t>(rgba, rgba_stride, shading, out, B, D, H, W, st);
    return rc_of(hipGetLastError());
}

int gmpi_light_apply_backward_launch(const void* rgba, int32_t rgba_dtype, const int64_t* rgba_stride, const float* shading,
                                     const float* grad_out, float* grad_rgba, float* grad_shading, int32_t B, int32_t D, int32_t H,
                                     int32_t W, void* stream) {
    if (B < 0 || D <= 0 || H <= 0 || W <= 0) return GMPI_E_SHAPE;
    if (B == 0) return GMPI_OK;
    if (!rgba || !rgba_stride || !shading || !grad_out || !grad_rgba || !grad_shading) return GMPI_E_NULL;
    if (rgba_dtype < GMPI_DTYPE_F32 || rgba_dtype > GMPI_DTYPE_F16) return GMPI_E_DTYPE;
    if (rgba_stride[4] != 1 || rgba_stride[3] < W || rgba_stride[2] <= 0 || rgba_stride[1] <= 0 || rgba_stride[0] < 0) return GMPI_E_STRIDE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t sb = rgba_stride[0], sd = rgba_stride[1], sc = rgba_stride[2], sr = rgba_stride[3];
    const int es = rgba_dtype == GMPI_DTYPE_F32 ? 4 : 2;
    const bool vec = W % 4 == 0 && sb % 4 == 0 && sd % 4 == 0 && sc % 4 == 0 && sr % 4 == 0 && reinterpret_cast<uintptr_t>(rgba) % (4 * es) == 0 &&
                     reinterpret_cast<uintptr_t>(shading) % 16 == 0 && reinterpret_cast<uintptr_t>(grad_out) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(grad_rgba) % 16 == 0 && reinterpret_cast<uintptr_t>(grad_shading) % 16 == 0;
    const dim3 block(256), grid(((vec ? W / 4 : W) + 255) / 256, H, B);
#define GMPI_LAB(T, V) hipLaunchKernelGGL((light_apply_backward_kernel<T, V>), grid, block, 0, st, static_cast<const T*>(rgba), sb, sd, sc, sr, shading, grad_out, grad_rgba, grad_shading, D, H, W)
    if (rgba_dtype == GMPI_DTYPE_F32) { if (vec) GMPI_LAB(float, 4); else GMPI_LAB(float, 1); }
    else if (rgba_dtype == GMPI_DTYPE_BF16) { if (vec) GMPI_LAB(bf16_t, 4); else GMPI_LAB(bf16_t, 1); }
    else { if (vec) GMPI_LAB(f16_t, 4); else GMPI_LAB(f16_t, 1); }
#undef GMPI_LAB
    return rc_of(hipGetLastError());
}

int gmpi_alpha_depth_backward_launch(const void* alpha, int32_t alpha_dtype, int64_t stride_b, int64_t stride_d, int64_t stride_row,
                                     const float* plane_ds, const float* transmittance, const float* grad_depth, float* grad_alpha,
                                     int64_t gstride_b, int64_t gstride_d, int64_t gstride_row, int32_t B, int32_t D, int32_t H,
                                     int32_t W, void* stream) {
    if (B < 0 || D <= 0 || H <= 0 || W <= 0) return GMPI_E_SHAPE;
    if (B == 0) return GMPI_OK;
    if (!alpha || !plane_ds || !grad_depth || !grad_alpha) return GMPI_E_NULL;
    if (alpha_dtype < GMPI_DTYPE_F32 || alpha_dtype > GMPI_DTYPE_F16) return GMPI_E_DTYPE;
    if (stride_b < 0 || stride_d <= 0 || stride_row < W || gstride_b <= 0 || gstride_d <= 0 || gstride_row < W) return GMPI_E_STRIDE;
    const dim3 grid((W + 255) / 256, H, B), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (alpha_dtype == GMPI_DTYPE_F32) hipLaunchKernelGGL(alpha_depth_backward_kernel<float>, grid, block, 0, st, static_cast<const float*>(alpha), stride_b, stride_d, stride_row, plane_ds, transmittance, grad_depth, grad_alpha, gstride_b, gstride_d, gstride_row, D, H, W);
    else if (alpha_dtype == GMPI_DTYPE_BF16) hipLaunchKernelGGL(alpha_depth_backward_kernel<bf16_t>, grid, block, 0, st, static_cast<const bf16_t*>(alpha), stride_b, stride_d, stride_row, plane_ds, transmittance, grad_depth, grad_alpha, gstride_b, gstride_d, gstride_row, D, H, W);
    else hipLaunchKernelGGL(alpha_depth_backward_kernel<f16_t>, grid, block, 0, st, static_cast<const f16_t*>(alpha), stride_b, stride_d, stride_row, plane_ds, transmittance, grad_depth, grad_alpha, gstride_b, gstride_d, gstride_row, D, H, W);
    return rc_of(hipGetLastError());
}

}  // extern "C"
