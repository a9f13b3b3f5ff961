// r3_probe -- round 3: semantics and rates the LDS-DMA tile kernel (render_dma.hip) is designed around.
//   r3_probe sem              LDS-DMA out-of-range lanes / inactive lanes, ds_read_u16_d16_hi low half, v_cvt_u32_f32 edge cases
//   r3_probe valu             issue cost of the instructions the new compositor / loader use (ns per wave64 instruction and SIMD)
//   r3_probe taps             a pixel's 16 taps from LDS, four ways (u16_d16_hi x16 | read2_b64 x2 + unpack | read2_b32 x8 | b128 x4)
//   r3_probe stream [mode]    read-only streaming rate of 3.2 GB: 0 dwordx4, 1 dwordx4 nt, 2 LDS-DMA b128, 3 dword, 4 dwordx2
// Build: hipcc --offload-arch=gfx950 -O3 r3_probe.hip -o bin/r3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void;

// ---------------------------------------------------------------- semantics
__global__ void sem_kernel(const uint32_t* src, int src_bytes, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 64 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4 * 64 * 4; i += 64) lds[i] = 0xAAAAAAAAu;
    __syncthreads();
    // (a) num_records = 2^31, odd lanes get the explicit out-of-range offset 0x80000000
    const __amdgpu_buffer_rsrc_t big = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, static_cast<int>(0x80000000u), 0x00020000);
    uint32_t off = (lane & 1) ? 0x80000000u : lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(big, (lds_void*)&lds[0], 16, off, 0, 0, 0);
    // (b) num_records = src_bytes: lanes past the end
    const __amdgpu_buffer_rsrc_t small = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, src_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(small, (lds_void*)&lds[256], 16, lane * 16u, 0, 0, 0);
    // (c) inactive lanes (exec mask): lanes >= 32 do not execute the load
    if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(big, (lds_void*)&lds[512], 16, lane * 16u, 0, 0, 0);
    // (d) "negative" offset: base in the middle of the buffer, voffset = -64 as unsigned
    const __amdgpu_buffer_rsrc_t mid = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src + 64), 0, src_bytes - 256, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(mid, (lds_void*)&lds[768], 16, lane * 16u - 64u, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
    // d16_hi: low half of the destination pre-set to 0xBEEF
    uint32_t t = 0x0000BEEFu, a = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint32_t*)lds)) + lane * 2u;
    asm volatile("ds_read_u16_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(t) : "v"(a));
    out[1024 + lane] = t;
    uint32_t u = 0xBEEF0000u;
    asm volatile("ds_read_u16_d16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(u) : "v"(a));
    out[1088 + lane] = u;
    // v_cvt_u32_f32 edge cases
    const float vals[8] = {-1.0f, -0.0f, 0.5f, 3.99f, 4294967296.0f, __builtin_nanf(""), __builtin_inff(), -__builtin_inff()};
    if (lane < 8) {
        uint32_t r;
        float v = vals[lane];
        asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
        out[1152 + lane] = r;
    }
}

static void run_sem() {
    const int words = 64 * 4 + 64;  // 320 words: lanes 0..63 x 16 B + a little more
    std::vector<uint32_t> h(words);
    for (int i = 0; i < words; ++i) h[i] = 0x10000u + i;
    uint32_t *src, *out;
    CK(hipMalloc(&src, words * 4)); CK(hipMalloc(&out, 2048 * 4));
    CK(hipMemcpy(src, h.data(), words * 4, hipMemcpyHostToDevice));
    CK(hipMemset(out, 0, 2048 * 4));
    sem_kernel<<<1, 64>>>(src, 40 * 16, out);  // lanes >= 40 are past num_records in case (b)
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> o(2048);
    CK(hipMemcpy(o.data(), out, 2048 * 4, hipMemcpyDeviceToHost));
    printf("(a) explicit OOB offset: lane0 %08x (want 00010000) lane1 %08x %08x %08x %08x (zeros = OOB writes zeros; aaaaaaaa = skipped) lane2 %08x\n", o[0], o[4], o[5], o[6], o[7], o[8]);
    printf("(b) past num_records:    lane39 %08x lane40 %08x lane63 %08x\n", o[256 + 39 * 4], o[256 + 40 * 4], o[256 + 63 * 4]);
    printf("(c) inactive lanes:      lane31 %08x lane32 %08x (aaaaaaaa = untouched)\n", o[512 + 31 * 4], o[512 + 32 * 4]);
    printf("(d) negative voffset:    lane0 %08x lane3 %08x lane4 %08x (want lane4 = %08x; lanes 0-3 before the base)\n", o[768], o[768 + 12], o[768 + 16], 0x10000u + 64);
    printf("d16_hi with low half 0xBEEF: lane0 %08x lane1 %08x  | d16 (lo) with high half 0xBEEF: %08x\n", o[1024], o[1025], o[1088]);
    printf("v_cvt_u32_f32: -1 -> %u, -0 -> %u, .5 -> %u, 3.99 -> %u, 2^32 -> %u, nan -> %u, inf -> %u, -inf -> %u\n", o[1152], o[1153], o[1154], o[1155], o[1156], o[1157], o[1158], o[1159]);
    hipFree(src); hipFree(out);
}

// ---------------------------------------------------------------- VALU rates
#define REP16(x) x x x x x x x x x x x x x x x x
#define OP8(T) T("%0") T("%1") T("%2") T("%3") T("%4") T("%5") T("%6") T("%7")
template <int MODE>
__global__ void valu_kernel(float* out, int iters) {
    float a0 = threadIdx.x * 0.37f, a1 = a0 + 1.3f, a2 = a0 + 2.1f, a3 = a0 + 3.7f, a4 = a0 + 4.2f, a5 = a0 + 5.9f, a6 = a0 + 6.4f, a7 = a0 + 7.8f;
    const float c = 1.0001f, d = 0.5f;
    for (int i = 0; i < iters; ++i) {
#define ASM8(body) REP16(asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d) : "vcc");)
        if (MODE == 0) { ASM8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n") }
        if (MODE == 1) { ASM8("v_floor_f32 %0, %1\n v_floor_f32 %1, %2\n v_floor_f32 %2, %3\n v_floor_f32 %3, %4\n v_floor_f32 %4, %5\n v_floor_f32 %5, %6\n v_floor_f32 %6, %7\n v_floor_f32 %7, %0\n") }
        if (MODE == 2) { ASM8("v_fract_f32 %0, %1\n v_fract_f32 %1, %2\n v_fract_f32 %2, %3\n v_fract_f32 %3, %4\n v_fract_f32 %4, %5\n v_fract_f32 %5, %6\n v_fract_f32 %6, %7\n v_fract_f32 %7, %0\n") }
        if (MODE == 3) { ASM8("v_cvt_u32_f32 %0, %1\n v_cvt_u32_f32 %1, %2\n v_cvt_u32_f32 %2, %3\n v_cvt_u32_f32 %3, %4\n v_cvt_u32_f32 %4, %5\n v_cvt_u32_f32 %5, %6\n v_cvt_u32_f32 %6, %7\n v_cvt_u32_f32 %7, %0\n") }
        if (MODE == 4) { ASM8("v_cvt_flr_i32_f32 %0, %1\n v_cvt_flr_i32_f32 %1, %2\n v_cvt_flr_i32_f32 %2, %3\n v_cvt_flr_i32_f32 %3, %4\n v_cvt_flr_i32_f32 %4, %5\n v_cvt_flr_i32_f32 %5, %6\n v_cvt_flr_i32_f32 %6, %7\n v_cvt_flr_i32_f32 %7, %0\n") }
        if (MODE == 5) { ASM8("v_sub_f32 %0, %1, %8\n v_sub_f32 %1, %2, %8\n v_sub_f32 %2, %3, %8\n v_sub_f32 %3, %4, %8\n v_sub_f32 %4, %5, %8\n v_sub_f32 %5, %6, %8\n v_sub_f32 %6, %7, %8\n v_sub_f32 %7, %0, %8\n") }
        if (MODE == 6) { ASM8("v_cndmask_b32 %0, %1, %8, vcc\n v_cndmask_b32 %1, %2, %8, vcc\n v_cndmask_b32 %2, %3, %8, vcc\n v_cndmask_b32 %3, %4, %8, vcc\n v_cndmask_b32 %4, %5, %8, vcc\n v_cndmask_b32 %5, %6, %8, vcc\n v_cndmask_b32 %6, %7, %8, vcc\n v_cndmask_b32 %7, %0, %8, vcc\n") }
        if (MODE == 7) { ASM8("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8\n") }
        if (MODE == 8) { ASM8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n") }
        if (MODE == 9) { ASM8("v_lshl_add_u32 %0, %1, 3, %8\n v_lshl_add_u32 %1, %2, 3, %8\n v_lshl_add_u32 %2, %3, 3, %8\n v_lshl_add_u32 %3, %4, 3, %8\n v_lshl_add_u32 %4, %5, 3, %8\n v_lshl_add_u32 %5, %6, 3, %8\n v_lshl_add_u32 %6, %7, 3, %8\n v_lshl_add_u32 %7, %0, 3, %8\n") }
        if (MODE == 10) { ASM8("v_mad_u32_u24 %0, %1, %8, %9\n v_mad_u32_u24 %1, %2, %8, %9\n v_mad_u32_u24 %2, %3, %8, %9\n v_mad_u32_u24 %3, %4, %8, %9\n v_mad_u32_u24 %4, %5, %8, %9\n v_mad_u32_u24 %5, %6, %8, %9\n v_mad_u32_u24 %6, %7, %8, %9\n v_mad_u32_u24 %7, %0, %8, %9\n") }
        if (MODE == 11) { ASM8("v_med3_f32 %0, %1, %8, %9\n v_med3_f32 %1, %2, %8, %9\n v_med3_f32 %2, %3, %8, %9\n v_med3_f32 %3, %4, %8, %9\n v_med3_f32 %4, %5, %8, %9\n v_med3_f32 %5, %6, %8, %9\n v_med3_f32 %6, %7, %8, %9\n v_med3_f32 %7, %0, %8, %9\n") }
        if (MODE == 12) { ASM8("v_mul_f32 %0, %1, %8\n v_mul_f32 %1, %2, %8\n v_mul_f32 %2, %3, %8\n v_mul_f32 %3, %4, %8\n v_mul_f32 %4, %5, %8\n v_mul_f32 %5, %6, %8\n v_mul_f32 %6, %7, %8\n v_mul_f32 %7, %0, %8\n") }
        if (MODE == 13) { ASM8("v_and_b32 %0, 0xffff0000, %1\n v_and_b32 %1, 0xffff0000, %2\n v_and_b32 %2, 0xffff0000, %3\n v_and_b32 %3, 0xffff0000, %4\n v_and_b32 %4, 0xffff0000, %5\n v_and_b32 %5, 0xffff0000, %6\n v_and_b32 %6, 0xffff0000, %7\n v_and_b32 %7, 0xffff0000, %0\n") }
        if (MODE == 14) { ASM8("v_max_f32 %0, %1, %8\n v_max_f32 %1, %2, %8\n v_max_f32 %2, %3, %8\n v_max_f32 %3, %4, %8\n v_max_f32 %4, %5, %8\n v_max_f32 %5, %6, %8\n v_max_f32 %6, %7, %8\n v_max_f32 %7, %0, %8\n") }
        if (MODE == 16) { ASM8("v_max3_u16 %0, %0, %1, %1 op_sel:[0,0,1,0]\n v_max3_u16 %1, %1, %2, %2 op_sel:[0,0,1,0]\n v_max3_u16 %2, %2, %3, %3 op_sel:[0,0,1,0]\n v_max3_u16 %3, %3, %4, %4 op_sel:[0,0,1,0]\n v_max3_u16 %4, %4, %5, %5 op_sel:[0,0,1,0]\n v_max3_u16 %5, %5, %6, %6 op_sel:[0,0,1,0]\n v_max3_u16 %6, %6, %7, %7 op_sel:[0,0,1,0]\n v_max3_u16 %7, %7, %0, %0 op_sel:[0,0,1,0]\n") }
        if (MODE == 17) { ASM8("v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %1, %1, %2\n v_pk_max_u16 %2, %2, %3\n v_pk_max_u16 %3, %3, %4\n v_pk_max_u16 %4, %4, %5\n v_pk_max_u16 %5, %5, %6\n v_pk_max_u16 %6, %6, %7\n v_pk_max_u16 %7, %7, %0\n") }
        if (MODE == 18) { ASM8("v_max3_u32 %0, %0, %1, %8\n v_max3_u32 %1, %1, %2, %8\n v_max3_u32 %2, %2, %3, %8\n v_max3_u32 %3, %3, %4, %8\n v_max3_u32 %4, %4, %5, %8\n v_max3_u32 %5, %5, %6, %8\n v_max3_u32 %6, %6, %7, %8\n v_max3_u32 %7, %7, %0, %8\n") }
        if (MODE == 19) { ASM8("v_max_u32 %0, %0, %1\n v_max_u32 %1, %1, %2\n v_max_u32 %2, %2, %3\n v_max_u32 %3, %3, %4\n v_max_u32 %4, %4, %5\n v_max_u32 %5, %5, %6\n v_max_u32 %6, %6, %7\n v_max_u32 %7, %7, %0\n") }
        if (MODE == 20) { ASM8("v_or_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_or_b32 %3, %3, %4\n v_or_b32 %4, %4, %5\n v_or_b32 %5, %5, %6\n v_or_b32 %6, %6, %7\n v_or_b32 %7, %7, %0\n") }
        if (MODE == 15) { ASM8("v_cvt_f32_u32 %0, %1\n v_cvt_f32_u32 %1, %2\n v_cvt_f32_u32 %2, %3\n v_cvt_f32_u32 %3, %4\n v_cvt_f32_u32 %4, %5\n v_cvt_f32_u32 %5, %6\n v_cvt_f32_u32 %6, %7\n v_cvt_f32_u32 %7, %0\n") }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// packed fp32 (VOP3P on 64-bit register pairs): two results per lane and instruction
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void valu_pk_kernel(float* out, int iters) {
    f32x2 a0 = {threadIdx.x * 0.37f, 1.0f}, a1 = a0 + 1.3f, a2 = a0 + 2.1f, a3 = a0 + 3.7f, a4 = a0 + 4.2f, a5 = a0 + 5.9f, a6 = a0 + 6.4f, a7 = a0 + 7.8f;
    const f32x2 c = {1.0001f, 0.9999f}, d = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
#define ASM8P(body) REP16(asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        if (MODE == 0) { ASM8P("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n") }
        if (MODE == 1) { ASM8P("v_pk_mul_f32 %0, %1, %8\n v_pk_mul_f32 %1, %2, %8\n v_pk_mul_f32 %2, %3, %8\n v_pk_mul_f32 %3, %4, %8\n v_pk_mul_f32 %4, %5, %8\n v_pk_mul_f32 %5, %6, %8\n v_pk_mul_f32 %6, %7, %8\n v_pk_mul_f32 %7, %0, %8\n") }
        if (MODE == 2) { ASM8P("v_pk_add_f32 %0, %1, %8\n v_pk_add_f32 %1, %2, %8\n v_pk_add_f32 %2, %3, %8\n v_pk_add_f32 %3, %4, %8\n v_pk_add_f32 %4, %5, %8\n v_pk_add_f32 %5, %6, %8\n v_pk_add_f32 %6, %7, %8\n v_pk_add_f32 %7, %0, %8\n") }
    }
    const f32x2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r.x + r.y;
}
template <int MODE> static void run_valu_pk(const char* name, int wps) {
    float* out; const int block = 64 * 4 * wps; CK(hipMalloc(&out, 256 * block * 4));
    const int iters = 1500; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    valu_pk_kernel<MODE><<<256, block>>>(out, 10);
    hipEventRecord(e0); valu_pk_kernel<MODE><<<256, block>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %-22s waves/SIMD=%d  %.3f ns per wave-instr per SIMD (two fp32 results per lane)\n", name, wps, ms * 1e6 / ((double)iters * 128 * wps));
    hipFree(out);
}
template <int MODE> static void run_valu(const char* name, int wps) {
    float* out; const int block = 64 * 4 * wps; CK(hipMalloc(&out, 256 * block * 4));
    const int iters = 1500; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    valu_kernel<MODE><<<256, block>>>(out, 10);
    hipEventRecord(e0); valu_kernel<MODE><<<256, block>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %-22s waves/SIMD=%d  %.3f ns per wave-instr per SIMD\n", name, wps, ms * 1e6 / ((double)iters * 128 * wps));
    hipFree(out);
}
static void ramp() { float* out; CK(hipMalloc(&out, 256 * 1024 * 4)); for (int i = 0; i < 60; ++i) valu_kernel<0><<<256, 1024>>>(out, 2000); CK(hipDeviceSynchronize()); hipFree(out); }

// ---------------------------------------------------------------- tap patterns
// 512-thread workgroups (32 x 16 pixels), 4 per CU; LDS image of a 56 x 27 texel box.  Every iteration = one plane of one pixel:
// a base address from (lx, ly) (frontal view: lx = x + jitter, ly = y), 16 taps, 16 FMA that consume them, EXTRA independent fp32 ops.
template <int MODE, int EXTRA>
__global__ __launch_bounds__(512) void tap_kernel(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[28672];  // 28 KB: 5 workgroups per CU would fit; we launch 4 per CU
    for (int i = threadIdx.x; i < 28672 / 4; i += 512) reinterpret_cast<uint32_t*>(lds)[i] = 0x3f003e80u + i;
    __syncthreads();
    const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
    const uint32_t base0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds));
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, e0 = x, e1 = y, e2 = 1.f, e3 = 2.f;
    const float w0 = 0.25f, w1 = 0.26f, w2 = 0.24f, w3 = 0.25f;
    uint32_t t[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) t[j] = 0;
    for (int i = 0; i < iters; ++i) {
        const int jit = (i * 7) & 7;
        if (MODE == 0) {  // raw bf16 planar [row][ch][x], pitch 112 B per (row, channel) line: 16 x ds_read_u16_d16_hi, no unpack
            uint32_t a = base0 + (y + (jit >> 2)) * 448u + (x + jit) * 2u;
            asm volatile(
                "ds_read_u16_d16_hi %0, %16\n ds_read_u16_d16_hi %1, %16 offset:2\n ds_read_u16_d16_hi %2, %16 offset:448\n ds_read_u16_d16_hi %3, %16 offset:450\n"
                "ds_read_u16_d16_hi %4, %16 offset:112\n ds_read_u16_d16_hi %5, %16 offset:114\n ds_read_u16_d16_hi %6, %16 offset:560\n ds_read_u16_d16_hi %7, %16 offset:562\n"
                "ds_read_u16_d16_hi %8, %16 offset:224\n ds_read_u16_d16_hi %9, %16 offset:226\n ds_read_u16_d16_hi %10, %16 offset:672\n ds_read_u16_d16_hi %11, %16 offset:674\n"
                "ds_read_u16_d16_hi %12, %16 offset:336\n ds_read_u16_d16_hi %13, %16 offset:338\n ds_read_u16_d16_hi %14, %16 offset:784\n ds_read_u16_d16_hi %15, %16 offset:786\n"
                : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]),
                  "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15])
                : "v"(a));
        } else if (MODE == 1) {  // interleaved bf16 texels [row][x][RGBA] 8 B per texel, pitch 56 texels: 2 x ds_read2_b64 + 16 unpack
            uint32_t a = base0 + (y + (jit >> 2)) * 448u + (x + jit) * 8u;
            asm volatile("ds_read2_b64 %0, %2 offset1:1\n ds_read2_b64 %1, %2 offset0:56 offset1:57\n"
                         : "=v"(*reinterpret_cast<__attribute__((ext_vector_type(4))) uint32_t*>(&t[0])), "=v"(*reinterpret_cast<__attribute__((ext_vector_type(4))) uint32_t*>(&t[4])) : "v"(a));
        } else if (MODE == 2) {  // fp32 planar [row][ch][x], pitch 56 floats: 8 x ds_read2_b32 (two row bases: the offset fields are 8 bits of dwords)
            uint32_t a = base0 + (y + (jit >> 2)) * 896u + (x + jit) * 4u, b = a + 896u;
            typedef __attribute__((ext_vector_type(2))) uint32_t u2;
            asm volatile("ds_read2_b32 %0, %8 offset1:1\n ds_read2_b32 %1, %8 offset0:56 offset1:57\n ds_read2_b32 %2, %8 offset0:112 offset1:113\n ds_read2_b32 %3, %8 offset0:168 offset1:169\n"
                         "ds_read2_b32 %4, %9 offset1:1\n ds_read2_b32 %5, %9 offset0:56 offset1:57\n ds_read2_b32 %6, %9 offset0:112 offset1:113\n ds_read2_b32 %7, %9 offset0:168 offset1:169\n"
                         : "=v"(*reinterpret_cast<u2*>(&t[0])), "=v"(*reinterpret_cast<u2*>(&t[2])), "=v"(*reinterpret_cast<u2*>(&t[4])), "=v"(*reinterpret_cast<u2*>(&t[6])),
                           "=v"(*reinterpret_cast<u2*>(&t[8])), "=v"(*reinterpret_cast<u2*>(&t[10])), "=v"(*reinterpret_cast<u2*>(&t[12])), "=v"(*reinterpret_cast<u2*>(&t[14])) : "v"(a), "v"(b));
        } else {  // fp32 RGBA texels 16 B, pitch 56 texels (buffer 24 KB): 4 x ds_read_b128
            uint32_t a = base0 + (y + (jit >> 2)) * 896u + (x + jit) * 16u;
            typedef __attribute__((ext_vector_type(4))) uint32_t u4;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:896\n ds_read_b128 %3, %4 offset:912\n"
                         : "=v"(*reinterpret_cast<u4*>(&t[0])), "=v"(*reinterpret_cast<u4*>(&t[4])), "=v"(*reinterpret_cast<u4*>(&t[8])), "=v"(*reinterpret_cast<u4*>(&t[12])) : "v"(a));
        }
        // independent fp32 work of the pixel (coordinate chain, weights, blend) while the taps are in flight
#pragma unroll
        for (int j = 0; j < EXTRA / 4; ++j) {
            e0 = __builtin_fmaf(e0, 1.0001f, 0.5f); e1 = __builtin_fmaf(e1, 0.9999f, 0.25f); e2 = e2 * 1.0001f; e3 = e3 + 0.125f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]), "+v"(t[10]),
                     "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]));
        float f[16];
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[2 * j] = __uint_as_float(t[j] << 16), f[2 * j + 1] = __uint_as_float(t[j] & 0xffff0000u);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) f[j] = __uint_as_float(t[j]);
        }
        acc0 = __builtin_fmaf(f[0], w0, acc0); acc0 = __builtin_fmaf(f[1], w1, acc0); acc0 = __builtin_fmaf(f[2], w2, acc0); acc0 = __builtin_fmaf(f[3], w3, acc0);
        acc1 = __builtin_fmaf(f[4], w0, acc1); acc1 = __builtin_fmaf(f[5], w1, acc1); acc1 = __builtin_fmaf(f[6], w2, acc1); acc1 = __builtin_fmaf(f[7], w3, acc1);
        acc2 = __builtin_fmaf(f[8], w0, acc2); acc2 = __builtin_fmaf(f[9], w1, acc2); acc2 = __builtin_fmaf(f[10], w2, acc2); acc2 = __builtin_fmaf(f[11], w3, acc2);
        acc3 = __builtin_fmaf(f[12], w0, acc3); acc3 = __builtin_fmaf(f[13], w1, acc3); acc3 = __builtin_fmaf(f[14], w2, acc3); acc3 = __builtin_fmaf(f[15], w3, acc3);
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc0 + acc1 + acc2 + acc3 + e0 + e1 + e2 + e3;
}
template <int MODE, int EXTRA> static void run_taps(const char* name) {
    float* out; CK(hipMalloc(&out, 1024 * 512 * 4));
    const int iters = 3000; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    tap_kernel<MODE, EXTRA><<<1024, 512>>>(out, 10);
    hipEventRecord(e0); tap_kernel<MODE, EXTRA><<<1024, 512>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 1024 workgroups x 8 waves over 1024 SIMDs: 8 waves per SIMD, each `iters` wave*planes
    printf("  %-44s extra fp32 %2d: %.1f ns per wave*plane per SIMD  (config 3 = 6144 wave*planes per SIMD: %.3f ms)\n", name, EXTRA, ms * 1e6 / (iters * 8.0), ms / (iters * 8.0) * 6144);
    hipFree(out);
}

// ---------------------------------------------------------------- streaming read
__global__ void fill_random(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = i + 0x9e3779b97f4a7c15ull;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        p[i] = static_cast<uint32_t>(z >> 32) & 0x3f7f3f7fu;  // two bf16 values in [0, 1)
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ v, size_t nvec, uint32_t* sink) {
    __shared__ __attribute__((aligned(16))) uint4 lds[8 * 256];
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    for (size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; i + 7 * 256 < nvec; i += stride) {
        if (MODE == 2) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(v + (i - threadIdx.x)), 0, 8 * 256 * 16, 0x00020000);
            const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)&lds[u * 256 + w * 64], 16, (u * 256 + threadIdx.x) * 16u, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += lds[threadIdx.x].x;
        } else if (MODE == 3) {  // dword loads: the same bytes with 4-byte accesses (8 x 4 per lane and trip)
            const uint32_t* p = reinterpret_cast<const uint32_t*>(v + (i - threadIdx.x));
#pragma unroll
            for (int u = 0; u < 32; ++u) acc |= __builtin_nontemporal_load(p + u * 256 + threadIdx.x);
        } else if (MODE == 5) {  // plain (cached) dword loads: the shape of the bf16 tile loader of render_lds.hip
            const uint32_t* p = reinterpret_cast<const uint32_t*>(v + (i - threadIdx.x));
#pragma unroll
            for (int u = 0; u < 32; ++u) acc |= p[u * 256 + threadIdx.x];
        } else if (MODE == 4) {
            const uint2* p = reinterpret_cast<const uint2*>(v + (i - threadIdx.x));
#pragma unroll
            for (int u = 0; u < 16; ++u) { const uint2 q = p[u * 256 + threadIdx.x]; acc |= q.x ^ q.y; }
        } else {
            uint4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { typedef __attribute__((ext_vector_type(4))) uint32_t u4; const u4 r = MODE == 1 ? __builtin_nontemporal_load(reinterpret_cast<const u4*>(v + i + u * 256)) : *reinterpret_cast<const u4*>(v + i + u * 256); q[u] = make_uint4(r.x, r.y, r.z, r.w); }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc |= q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int MODE> static void run_stream(const char* name, const uint4* v, size_t nvec, uint32_t* sink, int blocks_per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) stream_kernel<MODE><<<256 * blocks_per_cu, 256>>>(v, nvec, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) stream_kernel<MODE><<<256 * blocks_per_cu, 256>>>(v, nvec, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("  %-18s blocks/CU=%d  %.3f ms  %.2f TB/s\n", name, blocks_per_cu, ms, nvec * 16.0 / ms / 1e9);
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "sem";
    if (!strcmp(what, "sem")) { run_sem(); return 0; }
    ramp();
    if (!strcmp(what, "valu")) {
        if (argc > 2) {  // the candidates of the band kernel's range-check fold
            run_valu_pk<0>("v_pk_fma_f32", 4); run_valu_pk<1>("v_pk_mul_f32", 4); run_valu_pk<2>("v_pk_add_f32", 4);
            run_valu<16>("v_max3_u16 op_sel", 4); run_valu<17>("v_pk_max_u16", 4); run_valu<18>("v_max3_u32", 4); run_valu<19>("v_max_u32", 4); run_valu<20>("v_or_b32", 4); run_valu<0>("v_fma_f32", 4);
            return 0;
        }
        for (int w : {4}) {
            run_valu<0>("v_fma_f32", w); run_valu<12>("v_mul_f32", w); run_valu<5>("v_sub_f32", w); run_valu<14>("v_max_f32", w); run_valu<1>("v_floor_f32", w); run_valu<2>("v_fract_f32", w);
            run_valu<3>("v_cvt_u32_f32", w); run_valu<15>("v_cvt_f32_u32", w); run_valu<4>("v_cvt_flr_i32_f32", w); run_valu<11>("v_med3_f32", w); run_valu<6>("v_cndmask_b32", w);
            run_valu<7>("v_cmp_lt_u32", w); run_valu<8>("v_mov_b32", w); run_valu<13>("v_and_b32", w); run_valu<9>("v_lshl_add_u32", w); run_valu<10>("v_mad_u32_u24", w);
        }
    } else if (!strcmp(what, "taps")) {
        run_taps<0, 0>("bf16 planar, 16 x ds_read_u16_d16_hi"); run_taps<1, 0>("bf16 texels, 2 x ds_read2_b64 + 16 unpack"); run_taps<2, 0>("fp32 planar, 8 x ds_read2_b32"); run_taps<3, 0>("fp32 texels, 4 x ds_read_b128");
        run_taps<0, 36>("bf16 planar, 16 x ds_read_u16_d16_hi"); run_taps<1, 36>("bf16 texels, 2 x ds_read2_b64 + 16 unpack"); run_taps<2, 36>("fp32 planar, 8 x ds_read2_b32"); run_taps<3, 36>("fp32 texels, 4 x ds_read_b128");
    } else if (!strcmp(what, "stream")) {
        const size_t bytes = (size_t)3221225472u;  // the bf16 volume of config 3
        uint4* v; uint32_t* sink; CK(hipMalloc(&v, bytes)); CK(hipMalloc(&sink, 16));
        fill_random<<<4096, 256>>>(reinterpret_cast<uint32_t*>(v), bytes / 4);  // (bf16 values in [0, 1): constant data would flatter the rate -- DVFS is data dependent)
        CK(hipDeviceSynchronize());
        const size_t nvec = bytes / 16;
        const int mode = argc > 2 ? atoi(argv[2]) : -1;
        const int only_b = argc > 3 ? atoi(argv[3]) : 0;
        for (int b : {2, 4, 8}) {
            if (only_b && b != only_b) continue;
            if (mode < 0 || mode == 0) run_stream<0>("dwordx4", v, nvec, sink, b);
            if (mode < 0 || mode == 1) run_stream<1>("dwordx4 nt", v, nvec, sink, b);
            if (mode < 0 || mode == 2) run_stream<2>("LDS-DMA b128", v, nvec, sink, b);
            if (mode < 0 || mode == 3) run_stream<3>("dword nt", v, nvec, sink, b);
            if (mode < 0 || mode == 4) run_stream<4>("dwordx2", v, nvec, sink, b);
            if (mode < 0 || mode == 5) run_stream<5>("dword", v, nvec, sink, b);
        }
    }
    return 0;
}
