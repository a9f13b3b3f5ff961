// kbench -- torch-free timing / cross-check harness for the render kernels, through the C ABI (include/gmpi_render.h).
//   kbench <lib.so> <set: gpurun_in/kb_<set>.bin> <bf16|f16|f32> <variant[:s][,variant...]> [reps] [views] [planes]
// variant = auto | gather | lds | wave ; ":s" = strict-order mode.  The first variant is the reference of the cross-check
// (max |difference| of colour and depth against it).  Camera tensors come from tools/kbench_dump.py (bench.py's poses).
// Build: hipcc --offload-arch=gfx950 -O2 -I include tools/kbench.cpp -o tools/ubench/bin/kbench -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "gmpi_render.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ inline uint32_t hash32(uint64_t i) {
    uint64_t z = i + 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
// uniform [0,1) values in the storage type; alpha of the last plane = 1 (background_alpha_full)
template <int DT> __global__ void fill(void* vol, size_t n, int D, size_t plane_elems, size_t chan_elems) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = (hash32(i) >> 8) * (1.0f / 16777216.0f);
        const size_t k = (i / plane_elems) % D, c = (i % plane_elems) / chan_elems;
        if (k == (size_t)D - 1 && c == 3) v = 1.0f;
        if (DT == 0) ((float*)vol)[i] = v;
        else if (DT == 1) { uint32_t b = __float_as_uint(v); b += 0x7fffu + ((b >> 16) & 1u); ((uint16_t*)vol)[i] = (uint16_t)(b >> 16); }
        else ((_Float16*)vol)[i] = (_Float16)v;
    }
}

int main(int argc, char** argv) {
    if (argc < 5) { printf("usage: kbench lib set dtype variants [reps]\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
    auto launch = (int (*)(const GmpiRenderParams*, void*))dlsym(h, "gmpi_mpi_render_launch");
    const std::string set = argv[2], dts = argv[3];
    const int dt = dts == "f32" ? 0 : dts == "bf16" ? 1 : 2;
    const int reps = argc > 5 ? atoi(argv[5]) : 10;
    FILE* f = fopen(("gpurun_in/kb_" + set + ".bin").c_str(), "rb");
    if (!f) { printf("no input set %s\n", set.c_str()); return 1; }
    int hdr[3];
    if (fread(hdr, 4, 3, f) != 3) return 1;
    const int krep = getenv("KB_REPEAT") ? atoi(getenv("KB_REPEAT")) : 1;   // KB_REPEAT=k: the set's views k times over (more rounds of workgroups per launch, the same cameras)
    const int N = ((argc > 6 && atoi(argv[6]) > 0 && atoi(argv[6]) < hdr[0]) ? atoi(argv[6]) : hdr[0]) * krep, S = hdr[1], Dfile = hdr[2];  // [views]: the first n of the set
    const int D = (argc > 7 && atoi(argv[7]) > 0 && atoi(argv[7]) < Dfile) ? atoi(argv[7]) : Dfile;  // [planes]: the nearest d of the set
    float focal;
    std::vector<float> dhw1(Dfile * 3), c2w(hdr[0] * 16), eye(N * 3), zd(N * 3), ray((size_t)N * 3 * S * S);
    if (fread(&focal, 4, 1, f) != 1 || fread(dhw1.data(), 4, dhw1.size(), f) != dhw1.size() || fread(c2w.data(), 4, c2w.size(), f) != c2w.size()) { printf("short file\n"); return 1; }
    fclose(f);
    for (int n = 0; n < N; ++n) {  // camera.py:98-118, 182-211: K^-1 [x + .5, y + .5, 1] normalised (float64), cast, rotated
        const float* M = &c2w[(n % (N / krep)) * 16];
        for (int c = 0; c < 3; ++c) eye[n * 3 + c] = M[c * 4 + 3], zd[n * 3 + c] = M[c * 4 + 2];
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x) {
                const double dx = (x + 0.5 - S / 2.0) / focal, dy = (y + 0.5 - S / 2.0) / focal, inv = 1.0 / sqrt(dx * dx + dy * dy + 1.0);
                const float d0 = (float)(dx * inv), d1 = (float)(dy * inv), d2 = (float)inv;
                for (int c = 0; c < 3; ++c) ray[((size_t)(n * 3 + c) * S + y) * S + x] = fmaf(M[c * 4 + 2], d2, fmaf(M[c * 4 + 1], d1, M[c * 4 + 0] * d0));
            }
    }
    std::vector<float> dhw((size_t)N * D * 3);
    for (int n = 0; n < N; ++n) memcpy(&dhw[(size_t)n * D * 3], dhw1.data(), D * 3 * 4);
    const size_t es = dt == 0 ? 4 : 2, nvol = (size_t)N * D * 4 * S * S, npix = (size_t)N * S * S;
    void* vol; float *d_dhw, *d_eye, *d_zd, *d_ray, *d_rgb, *d_dep; uint32_t* d_st;
    CK(hipMalloc(&vol, nvol * es)); CK(hipMalloc(&d_dhw, dhw.size() * 4)); CK(hipMalloc(&d_eye, eye.size() * 4)); CK(hipMalloc(&d_zd, zd.size() * 4));
    const size_t st_bytes = getenv("KB_STAMPS") ? (64 + 3 * 65536) * 4 : 256;   // KB_STAMPS=1 (with GMPI_TUNE_WAVE=16384 on a profiling build): room for per-workgroup time stamps
    CK(hipMalloc(&d_ray, ray.size() * 4)); CK(hipMalloc(&d_rgb, npix * 3 * 4)); CK(hipMalloc(&d_dep, npix * 4)); CK(hipMalloc(&d_st, st_bytes)); CK(hipMemset(d_st, 0, st_bytes));
    CK(hipMemcpy(d_dhw, dhw.data(), dhw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_eye, eye.data(), eye.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_zd, zd.data(), zd.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ray, ray.data(), ray.size() * 4, hipMemcpyHostToDevice));
    const size_t chan = (size_t)S * S, plane = 4 * chan;
    if (dt == 0) fill<0><<<4096, 256>>>(vol, nvol, D, plane, chan); else if (dt == 1) fill<1><<<4096, 256>>>(vol, nvol, D, plane, chan); else fill<2><<<4096, 256>>>(vol, nvol, D, plane, chan);
    CK(hipDeviceSynchronize());

    GmpiRenderParams p; memset(&p, 0, sizeof p);
    const int vpm = getenv("KB_VPM") ? atoi(getenv("KB_VPM")) : 1;   // KB_VPM=8: the views share MPIs in groups of 8 (config 4: a camera path over ONE MPI)
    p.struct_size = sizeof p; p.rgba_dtype = dt; p.N = N; p.M = (N + vpm - 1) / vpm; p.D = D; p.Ht = p.Wt = p.H = p.W = S; p.views_per_mpi = vpm;
    p.rgba = vol; p.rgba_stride[0] = (int64_t)D * plane; p.rgba_stride[1] = plane; p.rgba_stride[2] = chan; p.rgba_stride[3] = S; p.rgba_stride[4] = 1;
    p.dhw = d_dhw; p.ray_dir = d_ray; p.eye_pos = d_eye; p.z_dir = d_zd; p.rgb_out = d_rgb; p.depth_out = d_dep; p.status = d_st;
    if (getenv("KB_PLANE_STRIDE0")) p.rgba_stride[1] = 0;  // (ablation: every plane reads plane 0 -- the volume becomes cache resident, the byte counts stay)
    {  // workspace for the kernels that want one (GMPI_VARIANT_BAND)
        auto wsb = (uint64_t (*)(const GmpiRenderParams*))dlsym(h, "gmpi_render_workspace_bytes");
        p.flags = GMPI_FLAG_ALIGN_CORNERS;
        p.variant = GMPI_VARIANT_BAND;  // (what the band kernel wants: AUTO asks for the same or for nothing)
        const uint64_t need = wsb ? wsb(&p) : 0;
        p.variant = GMPI_VARIANT_AUTO;
        if (need) { CK(hipMalloc(&p.workspace, need)); p.workspace_bytes = need; printf("workspace %.1f MB\n", need / 1e6); }
    }
    std::vector<float> ref_rgb, ref_dep, rgb(npix * 3), dep(npix);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::string vs = argv[4];
    size_t pos = 0;
    while (pos <= vs.size()) {
        size_t c = vs.find(',', pos); if (c == std::string::npos) c = vs.size();
        std::string v = vs.substr(pos, c - pos); pos = c + 1;
        bool strict = false;
        if (v.size() > 2 && v.substr(v.size() - 2) == ":s") strict = true, v = v.substr(0, v.size() - 2);
        p.variant = v == "gather" ? 1 : v == "lds" ? 2 : v == "wave" ? 3 : v == "dma" ? 4 : v == "band" ? 5 : 0;
        p.flags = GMPI_FLAG_ALIGN_CORNERS | GMPI_FLAG_OUT_PM1 | GMPI_FLAG_CHECK_LAST_PLANE | GMPI_FLAG_CHECK_RANGE | (strict ? GMPI_FLAG_STRICT_ORDER : 0);
        if (getenv("KB_NOCHECK")) p.flags &= ~static_cast<unsigned>(GMPI_FLAG_CHECK_RANGE);  // (A/B: what the fused [0,1] test costs)
        {  // the hint a host that knows the poses gives (MPIRenderer.render does): every camera axis within 0.2 rad of the MPI normal
            bool frontal = true;
            for (int n = 0; n < N; ++n) frontal = frontal && zd[n * 3 + 2] >= 0.98006658f;  // cos(0.2)
            if (frontal) p.flags |= GMPI_FLAG_HINT_FRONTAL;
            bool tilted = false;
            for (int n = 0; n < N; ++n) tilted = tilted || zd[n * 3 + 2] < 0.86280707f;  // cos(0.53)
            if (tilted) p.flags |= GMPI_FLAG_HINT_TILTED;
            bool oblique = false;   // (KB_NOHINT=1: a C host that does not know its cameras -- the device decides, group by group)
            for (int n = 0; n < N; ++n) oblique = oblique || zd[n * 3 + 2] < 0.93937271f;  // cos(0.35)
            if (oblique && !getenv("KB_NOHINT")) p.flags |= GMPI_FLAG_HINT_OBLIQUE;
        }
        CK(hipMemset(d_st, 0, 256)); CK(hipMemset(d_rgb, 0xff, npix * 12)); CK(hipMemset(d_dep, 0xff, npix * 4));
        int rc = launch(&p, nullptr);
        if (rc != 0) { printf("%-8s rc=%d\n", v.c_str(), rc); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(rgb.data(), d_rgb, npix * 12, hipMemcpyDeviceToHost)); CK(hipMemcpy(dep.data(), d_dep, npix * 4, hipMemcpyDeviceToHost));
        uint32_t st[64]; CK(hipMemcpy(st, d_st, 256, hipMemcpyDeviceToHost));
        if (st[16]) { printf("  prof (cycles of one wave): total %u | barrier %u burst %u check %u issue %u fg %u px0 %u px1 %u other %u\n", st[16], st[8], st[9], st[10], st[11], st[12], st[13], st[14], st[15]); }
        float best = 1e9, sum = 0;
        {  // warm-up: ~0.25 s of launches (a GPU coming from idle needs ~0.1 s to reach its busy clocks)
            hipEvent_t w0, w1; CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1));
            float acc = 0;
            while (acc < 250.f) {
                CK(hipEventRecord(w0)); for (int i = 0; i < 16; ++i) launch(&p, nullptr); CK(hipEventRecord(w1)); CK(hipEventSynchronize(w1));
                float ms; CK(hipEventElapsedTime(&ms, w0, w1)); acc += ms;
            }
        }
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(e0)); launch(&p, nullptr); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
        }
        if (getenv("KB_STAMPS")) {   // per XCD: when its workgroups of the LAST launch started and ended (s_memtime: 100 MHz)
            std::vector<uint32_t> stv(64 + 3 * 65536);
            CK(hipMemcpy(stv.data(), d_st, stv.size() * 4, hipMemcpyDeviceToHost));
            uint32_t t0 = 0xffffffffu, t1 = 0; int nwg = 0;
            for (int b = 0; b < 65536; ++b) { const uint32_t a = stv[64 + 3 * b], e = stv[65 + 3 * b]; if (e == 0) continue; ++nwg; if (a < t0) t0 = a; if (e > t1) t1 = e; }
            printf("  stamps: %d workgroups, launch span %.1f us\n", nwg, (t1 - t0) * 0.01);
            for (int x = 0; x < 8; ++x) {
                uint32_t last = 0, first = 0xffffffffu; int cnt = 0, by_block = 0; double busy = 0;
                for (int b = 0; b < 65536; ++b) { const uint32_t a = stv[64 + 3 * b], e = stv[65 + 3 * b]; if (e == 0 || (int)stv[66 + 3 * b] != x) continue; ++cnt; by_block += (b % 8 == x); if (e > last) last = e; if (a < first) first = a; busy += (e - a) * 0.01; }
                if (!cnt) continue;
                // (the stamps of one XCC share an epoch: its own span = first start ... last end; how full its 64 slots were over that span; how many slots still ran 2 / 5 / 10 % before its end)
                const double span = (last - first) * 0.01; int run2 = 0, run5 = 0, run10 = 0;
                for (int b = 0; b < 65536; ++b) { const uint32_t a = stv[64 + 3 * b], e = stv[65 + 3 * b]; if (e == 0 || (int)stv[66 + 3 * b] != x) continue;
                    const double ta = (a - first) * 0.01, te = (e - first) * 0.01; run2 += ta <= 0.98 * span && te > 0.98 * span; run5 += ta <= 0.95 * span && te > 0.95 * span; run10 += ta <= 0.9 * span && te > 0.9 * span; }
                printf("    XCC %d: %4d workgroups (%4d with blockIdx %% 8 == XCC), span %8.1f, mean workgroup life %7.1f, sum of lives / 64 slots %8.1f = %.3f of the span; slots busy at 90 / 95 / 98 %% of the span: %d %d %d\n", x, cnt, by_block, span, busy / cnt, busy / 64, busy / 64 / span, run10, run5, run2);
            }
        }
        double dc = 0, dd = 0; size_t nan = 0;
        if (ref_rgb.empty()) ref_rgb = rgb, ref_dep = dep;
        for (size_t i = 0; i < rgb.size(); ++i) { double d = fabs((double)rgb[i] - ref_rgb[i]); if (!(d == d)) ++nan; else if (d > dc) dc = d; }
        for (size_t i = 0; i < dep.size(); ++i) { double d = fabs((double)dep[i] - ref_dep[i]); if (!(d == d)) ++nan; else if (d > dd) dd = d; }
        const double gb = (double)nvol * es + npix * 12.0 + npix * 16.0;
        printf("%-6s %-5s %-7s%s mean %.4f ms  best %.4f ms  (%.3f of 8 TB/s)  status %u,%u,%u,%u  vs first: colour %.2e depth %.2e nan %zu\n", set.c_str(), dts.c_str(),
               v.c_str(), strict ? ":s" : "  ", sum / reps, best, gb / (sum / reps * 1e-3) / 8e12, st[0], st[1], st[2], st[3], dc, dd, nan);
        fflush(stdout);
    }
    return 0;
}
