// sr_align_blk.hip -- translation unit of the score-blocked, wave-tiled biWFA kernel (see sr_align_blk.inc)
#include "sr_dev_common.h"
#include "sr_uf_dev.h"
#define SR_BLK_TU 1
#undef SR_BFS_MAXACT
#ifdef SR_BLK_WAVE
// One-wave workgroups (-DSR_BLK_WAVE): a pair is aligned start to finish by ONE wave, 16 pairs per CU.  No wave ever
// waits for another one at a workgroup barrier, in a serial control section or for the slowest tile of a pass; the
// SIMDs interleave 4 independent pairs each.  4 segments (8 aligners) per pass keep the static LDS near 3 KB.
#define SR_BFS_MAXACT 4
#define K_P2 4
#define SR_BLK_SUB wave1
#else
// 16 segments (32 aligners) per pass keep the static LDS of a workgroup below 20 KB
#define SR_BFS_MAXACT 16
#define SR_BLK_SUB wg4
#endif
#define BFS_MAK_SLOTS SR_BLK_MAK_SLOTS
// The backtrace of a base case is inlined into the blocked kernel (round 4).  As a real call (it was the kernel's only one) it
// handed its op count back through `int *n_out`, a stack slot the callee writes with a flat store and the caller reads
// back with scratch_load, from a kernel that lives on ~100 spilled VGPRs and ~450 SGPRs spilled into VGPR lanes.  In
// some instances of the kernel -- which ones changes with register allocation, i.e. with unrelated edits and with debug
// prints -- the caller then continued with a stale count (a CIGAR short of its first run) and wrong uniform values (a
// mismatch penalty of 0 in the score of the CIGAR), or left the private aperture ("Memory access fault", round 3's
// "unexplained" faults and this round's first runs of the 512-thread 32-bit instance).  Same source, SR_BT_ATTR =
// __noinline__: wrong / faulting; __forceinline__: right (profiles/r04_backtrace_call.log).  DESIGN.md section 4.1.
#ifndef SR_BT_ATTR
#define SR_BT_ATTR __forceinline__
#endif
namespace SR_NS { namespace SR_BLK_SUB {
#include "sr_align_bfs.inc"
#ifndef SR_BLK_MIN_WAVES
#define SR_BLK_MIN_WAVES 4
#endif
#include "sr_align_blk.inc"
} }  // namespaces
using namespace SR_NS::SR_BLK_SUB;

template <typename OT, int NT, bool TWO, bool PROF = false>
static int launch_blk3(const SrAlignArgs *a, int nwg, size_t lds_bytes, hipStream_t st) {
    if (lds_bytes > 16 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)sr_align_blk_kernel<OT, NT, TWO, 5, 2, 1, PROF>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((sr_align_blk_kernel<OT, NT, TWO, 5, 2, 1, PROF>), dim3(nwg), dim3(NT), lds_bytes, st, *a);
    return (int)hipGetLastError();
}
// exact-penalty instance: mismatch 5, o1 + e1 = 10 (the reference's default -S 0,5,8,2,24,1 and the one-piece 0,5,8,2):
// blocks of 10 levels (sr_align_blk.inc blk_tile)
template <typename OT, int NT, bool TWO, bool PROF = false, typename RT = OT>
static int launch_blk10(const SrAlignArgs *a, int nwg, size_t lds_bytes, hipStream_t st) {
    if (lds_bytes > 16 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)sr_align_blk_kernel<OT, NT, TWO, 10, 2, 1, PROF, 5, 10, RT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((sr_align_blk_kernel<OT, NT, TWO, 10, 2, 1, PROF, 5, 10, RT>), dim3(nwg), dim3(NT), lds_bytes, st, *a);
    return (int)hipGetLastError();
}
// penalty sets this build has a blocked instance for (host side asks before choosing impl 2): levels per block
#if SR_SYMBITS == 2 && !defined(SR_BLK_WAVE)
extern "C" int srk_align_blk_max_levels(void) { return KB_MAX; }
#ifndef SR_BUILD_TAG
#define SR_BUILD_TAG "default"
#endif
extern "C" const char *srk_align_blk_build_tag(void) { return SR_BUILD_TAG; }          // A/B builds name themselves (workspace report)
extern "C" int srk_align_blk_supports(const SrPen *pen, const SrPen *ori) {
    if (ori->two || ori->e1 != 1 || ori->scope + 2 > BFS_MAK_SLOTS) return 0;
    if (pen->e1 != 2 || (pen->two && pen->e2 != 1)) return 0;
    // (second gap piece a multiple of five levels back: the tile reads M[s - o2 - e2] as two runs of five adjacent rows)
    if (pen->x == 5 && pen->o1 + pen->e1 == 10 && (!pen->two || (pen->o2 + pen->e2 >= 10 && (pen->o2 + pen->e2) % 5 == 0)) &&
        (2 * pen->scope + 2 * 10 + 2 + 9) / 10 * 10 <= BFS_MAK_SLOTS) return 10;
    const int B = 5;
    if (pen->x < B || pen->o1 + pen->e1 < B) return 0;
    if (pen->two && pen->o2 + pen->e2 < B) return 0;
    if (pen->scope + B + 1 > BFS_MAK_SLOTS) return 0;
    return B;
}
#endif
#ifdef SR_BLK_WAVE
extern "C" int SRK_NAME(srk_align_blkw)(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const bool two = a->pen.two != 0;
    if (a->kblock == 10 && nthreads == 128 && off16)          // two-wave workgroups, 8 per CU (same lean LDS state)
        return two ? launch_blk10<int16_t, 128, true>(a, nwg, lds_bytes, st) : launch_blk10<int16_t, 128, false>(a, nwg, lds_bytes, st);
    if (a->kblock == 10) {
        if (off16) return two ? launch_blk10<int16_t, 64, true>(a, nwg, lds_bytes, st) : launch_blk10<int16_t, 64, false>(a, nwg, lds_bytes, st);
        return two ? launch_blk10<int32_t, 64, true>(a, nwg, lds_bytes, st) : launch_blk10<int32_t, 64, false>(a, nwg, lds_bytes, st);
    }
    if (off16) return two ? launch_blk3<int16_t, 64, true>(a, nwg, lds_bytes, st) : launch_blk3<int16_t, 64, false>(a, nwg, lds_bytes, st);
    return two ? launch_blk3<int32_t, 64, true>(a, nwg, lds_bytes, st) : launch_blk3<int32_t, 64, false>(a, nwg, lds_bytes, st);
}
#else
extern "C" int SRK_NAME(srk_align_blk)(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const bool two = a->pen.two != 0;
#ifdef SR_BLK_ONLY_C5     // experiment builds: the C5 instance only (32-bit search, 16-bit ring, 512 threads)
    if (a->kblock == 10 && !off16 && two && a->ring_u16 && nthreads == 512) return launch_blk10<int32_t, 512, true, false, uint16_t>(a, nwg, lds_bytes, st);
    return -1;
}
#elif defined(SR_BLK_ONLY_PROD)   // experiment builds (seconds instead of minutes, one kernel in the disassembly): the C2 / C4 production instance only
    if (a->kblock == 10 && off16 && two && nthreads == 256 && !a->profile_ticks) return launch_blk10<int16_t, 256, true>(a, nwg, lds_bytes, st);
#ifdef SR_NT192                   // (experiment: three-wave workgroups, five per CU)
    if (a->kblock == 10 && off16 && two && nthreads == 192 && !a->profile_ticks) return launch_blk10<int16_t, 192, true>(a, nwg, lds_bytes, st);
#endif
    return -1;
}
#else
    if (a->kblock == 10) {
        if (off16) {
#if SR_SYMBITS == 2
#ifdef SR_PROF_WIDE   // (A/B builds only: tick counters for the 1024-thread shape)
            if (nthreads == 1024 && two && a->profile_ticks) return launch_blk10<int16_t, 1024, true, true>(a, nwg, lds_bytes, st);
#endif
            if (nthreads == 1024 && two) return launch_blk10<int16_t, 1024, true>(a, nwg, lds_bytes, st);    // fewer pairs than CUs
#endif
            if (nthreads >= 512) return two ? launch_blk10<int16_t, 512, true>(a, nwg, lds_bytes, st) : launch_blk10<int16_t, 512, false>(a, nwg, lds_bytes, st);
#if SR_SYMBITS == 2
            // the instrumented instance (tick counters [6..15], SR_PROFILE_TICKS=1): default shape only
            if (a->profile_ticks && two) return launch_blk10<int16_t, 256, true, true>(a, nwg, lds_bytes, st);
#endif
            return two ? launch_blk10<int16_t, 256, true>(a, nwg, lds_bytes, st) : launch_blk10<int16_t, 256, false>(a, nwg, lds_bytes, st);
        }
        // 32-bit searches with the ring stored as uint16 (offset + 8192): longest sequence < 57 k (host: ring_u16)
        if (a->ring_u16 && nthreads == 512) return two ? launch_blk10<int32_t, 512, true, false, uint16_t>(a, nwg, lds_bytes, st)
                                                       : launch_blk10<int32_t, 512, false, false, uint16_t>(a, nwg, lds_bytes, st);
        if (a->ring_u16) return two ? launch_blk10<int32_t, 256, true, false, uint16_t>(a, nwg, lds_bytes, st)
                                    : launch_blk10<int32_t, 256, false, false, uint16_t>(a, nwg, lds_bytes, st);
        return two ? launch_blk10<int32_t, 256, true>(a, nwg, lds_bytes, st) : launch_blk10<int32_t, 256, false>(a, nwg, lds_bytes, st);
    }
    if (off16) {
        if (nthreads == 128) return two ? launch_blk3<int16_t, 128, true>(a, nwg, lds_bytes, st) : launch_blk3<int16_t, 128, false>(a, nwg, lds_bytes, st);
        if (nthreads == 512) return two ? launch_blk3<int16_t, 512, true>(a, nwg, lds_bytes, st) : launch_blk3<int16_t, 512, false>(a, nwg, lds_bytes, st);
        return two ? launch_blk3<int16_t, 256, true>(a, nwg, lds_bytes, st) : launch_blk3<int16_t, 256, false>(a, nwg, lds_bytes, st);
    }
    return two ? launch_blk3<int32_t, 256, true>(a, nwg, lds_bytes, st) : launch_blk3<int32_t, 256, false>(a, nwg, lds_bytes, st);
}
#endif   // SR_BLK_ONLY_PROD
#endif
