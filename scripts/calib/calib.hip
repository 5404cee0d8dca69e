// calib.hip -- calibration micro-kernels for the roofline line of bench.py (MI355X_MICROARCH.md, HBM section:
// "Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//
// Every kernel moves a byte count known on the host with the access shape of sr_align_blk_kernel's row traffic:
// 8 B per lane, 512 contiguous bytes per wave instruction, at scattered 512-B chunks of a buffer much larger
// than the 256 MiB Infinity Cache.  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
// passes); scripts/calib/collect.py divides counter bytes by known bytes -> correction factors.
// The VALU kernel measures integer VALU wave-instructions per cycle and SIMD with s_memtime stamps (the
// denominator of roofline.valu_issue_frac).
//
//   hipcc --offload-arch=gfx950 -O3 -o calib calib.hip && ./calib out.json
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <typename T> using GP = T __attribute__((address_space(1))) *;

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

// scattered 512-B chunks, 8 B per lane (the alignment kernel's ld4<int16_t> / st4<int16_t>)
template <bool NT>
__global__ void __launch_bounds__(256) calib_rd8_scatter(const u2 *buf, uint64_t nchunks, int iters, uint64_t *sink) {
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    GP<const u2> p = (GP<const u2>)buf;
    u2 acc = {0u, 0u};
    for (int i = 0; i < iters; i++) {
        const uint64_t c = mix(wave * 1000003ULL + (uint64_t)i) % nchunks;
        const u2 v = NT ? __builtin_nontemporal_load(p + c * 64 + lane) : p[c * 64 + lane];
        acc.x ^= v.x; acc.y += v.y;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc.x;
}
template <bool NT>
__global__ void __launch_bounds__(256) calib_wr8_scatter(u2 *buf, uint64_t nchunks, int iters) {
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    GP<u2> p = (GP<u2>)buf;
    for (int i = 0; i < iters; i++) {
        const uint64_t c = mix(wave * 1000003ULL + (uint64_t)i) % nchunks;
        const u2 v = {(uint32_t)i, (uint32_t)lane};
        if (NT) __builtin_nontemporal_store(v, p + c * 64 + lane); else p[c * 64 + lane] = v;
    }
}
// the alignment kernel's mix: per iteration R scattered chunk loads and W scattered chunk stores (8 B per lane);
// the loads of an iteration are issued together, their sum is consumed, then the stores -- like a tile
template <int R, int W>
__global__ void __launch_bounds__(256) calib_mix8_scatter(u2 *buf, uint64_t nchunks, int iters, uint64_t *sink) {
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    GP<u2> p = (GP<u2>)buf;
    u2 acc = {0u, 0u};
    for (int i = 0; i < iters; i++) {
        u2 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = p[(mix(wave * 1000003ULL + (uint64_t)i * 16 + r) % nchunks) * 64 + lane];
#pragma unroll
        for (int r = 0; r < R; r++) { acc.x ^= v[r].x; acc.y += v[r].y; }
#pragma unroll
        for (int w = 0; w < W; w++) {
            const u2 o = {acc.x + (uint32_t)w, acc.y};
            p[(mix(wave * 1000003ULL + (uint64_t)i * 16 + 8 + w) % nchunks) * 64 + lane] = o;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc.x;
}
// streaming 16 B per lane (the guide's calibrated shape: FETCH_SIZE reads 1/2, WRITE_SIZE exact)
__global__ void __launch_bounds__(256) calib_rd16_stream(const u4 *buf, uint64_t n16, uint64_t *sink) {
    GP<const u4> p = (GP<const u4>)buf;
    u4 acc = {0u, 0u, 0u, 0u};
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
        const u4 v = p[i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u && acc.z == 1u && acc.w == 2u) sink[0] = acc.x;
}
__global__ void __launch_bounds__(256) calib_wr16_stream(u4 *buf, uint64_t n16) {
    GP<u4> p = (GP<u4>)buf;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
        const u4 v = {(uint32_t)i, 1u, 2u, 3u}; p[i] = v;
    }
}
__global__ void __launch_bounds__(256) calib_rd8_stream(const u2 *buf, uint64_t n8, uint64_t *sink) {
    GP<const u2> p = (GP<const u2>)buf;
    u2 acc = {0u, 0u};
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * 256) {
        const u2 v = p[i]; acc.x ^= v.x; acc.y += v.y;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc.x;
}

// integer VALU issue rate: 8 independent chains of v_max_i32 / v_add_u32 / v_cndmask-like selects
__global__ void __launch_bounds__(256) calib_valu(int iters, int seed, uint64_t *cycles, int *sink) {
    int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 ^ 5, a3 = a0 + 7, a4 = a0 * 5, a5 = a0 ^ 11, a6 = a0 + 13, a7 = a0 * 7;
    const int b = seed * 17 + 1;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            a0 = max(a0 + b, a4); a1 = max(a1 + b, a5); a2 = max(a2 + b, a6); a3 = max(a3 + b, a7);
            a4 = min(a4 ^ b, a0); a5 = min(a5 ^ b, a1); a6 = min(a6 ^ b, a2); a7 = min(a7 ^ b, a3);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[(uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x7fffffff) sink[0] = a0;
}

static float run_timed(hipStream_t st, void (*launch)(hipStream_t)) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, st)); launch(st); CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return ms;
}

int main(int argc, char **argv) {
    const char *out = argc > 1 ? argv[1] : "calib.json";
    const uint64_t bytes = 4ULL << 30;                       // 4 GiB >> 256 MiB Infinity Cache
    void *buf; uint64_t *sink; uint64_t *cyc;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&sink, 64)); CHK(hipMalloc(&cyc, 8 * 65536));
    CHK(hipMemset(buf, 1, bytes));
    hipStream_t st; CHK(hipStreamCreate(&st));
    const uint64_t nchunks = bytes / 512;
    const int wgs = 256 * 8, iters = 2048;                   // 8192 waves x 2048 chunks x 512 B = 8 GiB
    const uint64_t scat_bytes = (uint64_t)wgs * 4 * iters * 512;
    FILE *f = fopen(out, "w");
    if (!f) { perror(out); return 1; }
    fprintf(f, "{\n");
    static void *g_buf; static uint64_t *g_sink, *g_cyc; static uint64_t g_nchunks, g_bytes; static int g_wgs, g_iters;
    g_buf = buf; g_sink = sink; g_cyc = cyc; g_nchunks = nchunks; g_bytes = bytes; g_wgs = wgs; g_iters = iters;
    struct K { const char *name; uint64_t bytes; void (*fn)(hipStream_t); };
    K ks[] = {
        {"calib_rd8_scatter<false>", scat_bytes, [](hipStream_t s) { hipLaunchKernelGGL(calib_rd8_scatter<false>, dim3(g_wgs), dim3(256), 0, s, (const u2 *)g_buf, g_nchunks, g_iters, g_sink); }},
        {"calib_rd8_scatter<true>", scat_bytes, [](hipStream_t s) { hipLaunchKernelGGL(calib_rd8_scatter<true>, dim3(g_wgs), dim3(256), 0, s, (const u2 *)g_buf, g_nchunks, g_iters, g_sink); }},
        {"calib_wr8_scatter<false>", scat_bytes, [](hipStream_t s) { hipLaunchKernelGGL(calib_wr8_scatter<false>, dim3(g_wgs), dim3(256), 0, s, (u2 *)g_buf, g_nchunks, g_iters); }},
        {"calib_wr8_scatter<true>", scat_bytes, [](hipStream_t s) { hipLaunchKernelGGL(calib_wr8_scatter<true>, dim3(g_wgs), dim3(256), 0, s, (u2 *)g_buf, g_nchunks, g_iters); }},
        {"calib_mix8_scatter<3, 2>", (uint64_t)g_wgs * 4 * (g_iters / 4) * 512 * 5, [](hipStream_t s) { hipLaunchKernelGGL((calib_mix8_scatter<3, 2>), dim3(g_wgs), dim3(256), 0, s, (u2 *)g_buf, g_nchunks, g_iters / 4, g_sink); }},
        {"calib_mix8_scatter<6, 4>", (uint64_t)g_wgs * 4 * (g_iters / 8) * 512 * 10, [](hipStream_t s) { hipLaunchKernelGGL((calib_mix8_scatter<6, 4>), dim3(g_wgs), dim3(256), 0, s, (u2 *)g_buf, g_nchunks, g_iters / 8, g_sink); }},
        {"calib_mix8_scatter<8, 1>", (uint64_t)g_wgs * 4 * (g_iters / 8) * 512 * 9, [](hipStream_t s) { hipLaunchKernelGGL((calib_mix8_scatter<8, 1>), dim3(g_wgs), dim3(256), 0, s, (u2 *)g_buf, g_nchunks, g_iters / 8, g_sink); }},
        {"calib_mix8_scatter<1, 4>", (uint64_t)g_wgs * 4 * (g_iters / 4) * 512 * 5, [](hipStream_t s) { hipLaunchKernelGGL((calib_mix8_scatter<1, 4>), dim3(g_wgs), dim3(256), 0, s, (u2 *)g_buf, g_nchunks, g_iters / 4, g_sink); }},
        {"calib_rd16_stream", bytes, [](hipStream_t s) { hipLaunchKernelGGL(calib_rd16_stream, dim3(256 * 16), dim3(256), 0, s, (const u4 *)g_buf, g_bytes / 16, g_sink); }},
        {"calib_wr16_stream", bytes, [](hipStream_t s) { hipLaunchKernelGGL(calib_wr16_stream, dim3(256 * 16), dim3(256), 0, s, (u4 *)g_buf, g_bytes / 16); }},
        {"calib_rd8_stream", bytes, [](hipStream_t s) { hipLaunchKernelGGL(calib_rd8_stream, dim3(256 * 16), dim3(256), 0, s, (const u2 *)g_buf, g_bytes / 8, g_sink); }},
    };
    fprintf(f, " \"hbm\": [\n");
    const int nk = (int)(sizeof(ks) / sizeof(ks[0]));
    for (int i = 0; i < nk; i++) {
        run_timed(st, ks[i].fn);                                   // warm-up (also a profiled dispatch: same bytes)
        const float ms = run_timed(st, ks[i].fn);
        fprintf(f, "  {\"kernel\": \"%s\", \"bytes\": %llu, \"ms\": %.4f, \"GBps\": %.1f}%s\n", ks[i].name,
                (unsigned long long)ks[i].bytes, ms, (double)ks[i].bytes / ms / 1e6, i + 1 < nk ? "," : "");
    }
    fprintf(f, " ],\n \"valu\": [\n");
    // VALU: waves per SIMD = 1, 2, 4, 8 (256-thread WGs put one wave on each SIMD; k WGs per CU)
    const int vit = 4096;
    const int wps[] = {1, 2, 4, 8};
    for (int w = 0; w < 4; w++) {
        const int nwg = 256 * wps[w];
        CHK(hipMemsetAsync(cyc, 0, 8 * 65536, st));
        hipLaunchKernelGGL(calib_valu, dim3(nwg), dim3(256), 0, st, vit, 3, cyc, (int *)sink);
        hipLaunchKernelGGL(calib_valu, dim3(nwg), dim3(256), 0, st, vit, 3, cyc, (int *)sink);
        CHK(hipStreamSynchronize(st));
        std::vector<uint64_t> h((size_t)nwg * 4);
        CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0; for (uint64_t v : h) sum += (double)v;
        const double mean = sum / h.size();
        const double insts = (double)vit * 8 * 16;                  // 16 VALU per unrolled body (add/xor + max/min)
        // per SIMD: wps waves each issuing `insts` in `mean` cycles (if all resident together)
        fprintf(f, "  {\"waves_per_simd\": %d, \"insts_per_wave\": %.0f, \"mean_cycles\": %.0f, "
                   "\"valu_insts_per_cycle_per_simd\": %.4f}%s\n", wps[w], insts, mean, wps[w] * insts / mean, w < 3 ? "," : "");
    }
    fprintf(f, " ]\n}\n");
    fclose(f);
    return 0;
}
