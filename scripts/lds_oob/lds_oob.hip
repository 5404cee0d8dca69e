// lds_oob.hip -- what an LDS read outside the LDS does on gfx950.
// Every workgroup fills its dynamic LDS with a pattern, then each lane reads ds_read2_b32 pairs at (a) valid addresses,
// (b) the address under test, (c) valid addresses again, for several rounds with the LDS re-filled in between
// (a workgroup that aligns several pairs one after the other).  Reports per address class: the values a bad read returns,
// whether valid reads issued after it (same wave, later waves, later rounds) still return the pattern.
//   hipcc --offload-arch=gfx950 -O2 -o lds_oob lds_oob.hip && ./lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
extern __shared__ uint32_t lds[];
typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) W2;
__device__ __forceinline__ W2 rd2(uint32_t byte_addr) {
    return *(const W2 __attribute__((address_space(3))) *)(uintptr_t)byte_addr;
}
struct Res { uint32_t bad_lo, bad_hi, good_before_err, good_after_err, later_round_err, nonzero_bad; };
__global__ void __launch_bounds__(256) probe(uint32_t words, uint32_t bad_addr, uint32_t lane_stride, int rounds, Res *out) {
    const int tid = threadIdx.x;
    uint32_t gb = 0, ga = 0, lr = 0, nz = 0, blo = 0, bhi = 0;
    for (int r = 0; r < rounds; r++) {
        const uint32_t salt = 0x9e3779b9u * (uint32_t)(r + 1) + blockIdx.x;
        for (uint32_t i = tid; i < words; i += 256) lds[i] = i * 2654435761u + salt;
        __syncthreads();
        uint32_t base = (uint32_t)(uintptr_t)(uint32_t __attribute__((address_space(3))) *)lds;   // byte address of the allocation
        // (a) valid reads
        for (int k = 0; k < 8; k++) {
            const uint32_t w = (tid * 37u + k * 101u) % (words - 1);
            const W2 v = rd2(base + 4 * w);
            if (v.x != w * 2654435761u + salt || v.y != (w + 1) * 2654435761u + salt) gb++;
        }
        // (b) the read under test (only in round 0, only the waves selected by lane_stride != 0)
        if (r == 0) {
            const W2 b = rd2(bad_addr + lane_stride * (uint32_t)tid);
            blo = b.x; bhi = b.y;
            if (b.x | b.y) nz++;
        }
        // (c) valid reads issued right after it
        for (int k = 0; k < 8; k++) {
            const uint32_t w = (tid * 53u + k * 7u + 3u) % (words - 1);
            const W2 v = rd2(base + 4 * w);
            if (v.x != w * 2654435761u + salt || v.y != (w + 1) * 2654435761u + salt) { if (r == 0) ga++; else lr++; }
        }
        __syncthreads();
    }
    Res *o = out + (size_t)blockIdx.x * 256 + tid;
    o->bad_lo = blo; o->bad_hi = bhi; o->good_before_err = gb; o->good_after_err = ga; o->later_round_err = lr; o->nonzero_bad = nz;
}
int main() {
    const uint32_t words = 5 * 1024 / 4 + 4096;              // ~21 KB per workgroup (4 per CU like the alignment kernel)
    const int nwg = 1024, rounds = 4;
    Res *d; hipMalloc(&d, sizeof(Res) * nwg * 256);
    std::vector<Res> h(nwg * 256);
    struct Case { const char *name; uint32_t addr, stride; } cases[] = {
        {"inside own allocation", 64, 8},
        {"beyond the allocation, inside 64 KB", 48 * 1024, 8},
        {"inside the 160 KB LDS (100 KB)", 100 * 1024, 8},
        {"just below 160 KB", 160 * 1024 - 8 * 256, 8},
        {"at 160 KB", 160 * 1024, 8},
        {"256 KB", 256 * 1024, 8},
        {"1 MB", 1u << 20, 8},
        {"16 MB", 1u << 24, 8},
        {"2 GB", 1u << 31, 8},
        {"4 GB - 4 KB (negative index)", 0xfffff000u, 8},
        {"4 GB - 4 (second dword wraps)", 0xfffffffcu, 0},
    };
    for (auto &c : cases) {
        hipMemset(d, 0, sizeof(Res) * nwg * 256);
        hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), words * 4, 0, words, c.addr, c.stride, rounds, d);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(h.data(), d, sizeof(Res) * nwg * 256, hipMemcpyDeviceToHost);
        unsigned long long gb = 0, ga = 0, lr = 0, nz = 0;
        for (auto &r : h) { gb += r.good_before_err; ga += r.good_after_err; lr += r.later_round_err; nz += r.nonzero_bad; }
        printf("%-40s addr 0x%08x: sync %s | bad read non-zero in %llu lanes (first lane got %08x %08x) | valid reads wrong: before %llu, after (same round) %llu, later rounds %llu\n",
               c.name, c.addr, hipGetErrorString(e), nz, h[0].bad_lo, h[0].bad_hi, gb, ga, lr);
    }
    return 0;
}
