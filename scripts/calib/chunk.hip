// chunk.hip -- does row locality pay?  Mixed 8-B-per-lane traffic like sr_align_blk_kernel's tiles (10 row loads + 6 row
// stores per iteration and wave, 512 B per wave instruction) with the rows of an iteration either scattered over the
// buffer (today's [level][component][diagonal] rows: consecutive levels are 10+ KB apart) or adjacent (a level-minor
// layout: the rows a tile reads / writes are consecutive 512-B pieces of one region).
//   hipcc --offload-arch=gfx950 -O3 -o chunk chunk.hip && ./chunk
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
template <typename T> using GP = T __attribute__((address_space(1))) *;
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL; x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
// RUN: rows of one iteration that are adjacent (1 = all scattered, R = one run of R rows)
template <int R, int W, int RUN>
__global__ void __launch_bounds__(256) k(u2 *buf, uint64_t nchunks, int iters, uint64_t *sink) {
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    GP<u2> p = (GP<u2>)buf;
    u2 acc = {0u, 0u};
    for (int i = 0; i < iters; i++) {
        u2 v[R > 0 ? R : 1];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint64_t c = (mix(wave * 1000003ULL + (uint64_t)i * 64 + r / RUN) % (nchunks - RUN)) + r % RUN;
            v[r] = p[c * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < R; r++) { acc.x ^= v[r].x; acc.y += v[r].y; }
#pragma unroll
        for (int w = 0; w < W; w++) {
            const uint64_t c = (mix(wave * 1000003ULL + (uint64_t)i * 64 + 32 + w / RUN) % (nchunks - RUN)) + w % RUN;
            const u2 o = {acc.x + (uint32_t)w, acc.y};
            p[c * 64 + lane] = o;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc.x;
}
template <int R, int W, int RUN>
static void run(u2 *buf, uint64_t nchunks, uint64_t *sink, const char *name) {
    const int iters = 400, grid = 256 * 4;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<R, W, RUN>), dim3(grid), dim3(256), 0, 0, buf, nchunks, 20, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<R, W, RUN>), dim3(grid), dim3(256), 0, 0, buf, nchunks, iters, sink);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * 4 * iters * (R + W) * 512.0;
    printf("{\"shape\": \"%s\", \"loads\": %d, \"stores\": %d, \"adjacent_rows\": %d, \"GBps\": %.1f}\n", name, R, W, RUN, bytes / ms / 1e6);
}
int main() {
    const uint64_t bytes = 4ULL << 30, nchunks = bytes / 512;
    u2 *buf; uint64_t *sink;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&sink, 8)); CHK(hipMemset(buf, 1, bytes));
    run<10, 6, 1>(buf, nchunks, sink, "scattered rows");
    run<10, 6, 2>(buf, nchunks, sink, "runs of 2 rows");
    run<10, 6, 5>(buf, nchunks, sink, "runs of 5 rows");
    run<10, 6, 10>(buf, nchunks, sink, "runs of 10 rows");
    run<20, 12, 1>(buf, nchunks, sink, "scattered rows, 2x in flight");
    run<20, 12, 10>(buf, nchunks, sink, "runs of 10 rows, 2x in flight");
    run<16, 0, 1>(buf, nchunks, sink, "loads only, scattered");
    run<16, 0, 16>(buf, nchunks, sink, "loads only, runs of 16");
    run<0, 16, 1>(buf, nchunks, sink, "stores only, scattered");
    run<0, 16, 16>(buf, nchunks, sink, "stores only, runs of 16");
    return 0;
}
