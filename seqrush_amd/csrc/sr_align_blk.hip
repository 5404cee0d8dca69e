// sr_align_blk.hip -- translation unit of the score-blocked, wave-tiled biWFA kernel (see sr_align_blk.inc)
#include "sr_dev_common.h"
#define SR_BLK_TU 1
// 16 segments (32 aligners) per pass keep the static LDS of a workgroup below 16 KB: 8 workgroups per CU
#undef SR_BFS_MAXACT
#define SR_BFS_MAXACT 16
#define BFS_MAK_SLOTS SR_BLK_MAK_SLOTS
namespace SR_NS {
#include "sr_align_bfs.inc"
#ifndef SR_BLK_MIN_WAVES
#define SR_BLK_MIN_WAVES 4
#endif
#include "sr_align_blk.inc"
}  // namespace
using namespace SR_NS;

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
// penalty sets this build has a blocked instance for (host side asks before choosing impl 2)
#if SR_SYMBITS == 2
extern "C" int srk_align_blk_supports(const SrPen *pen, const SrPen *ori) {
    const int B = 5;
    if (pen->x < B || pen->o1 + pen->e1 < B || pen->e1 != 2) return 0;
    if (pen->two && (pen->o2 + pen->e2 < B || pen->e2 != 1)) return 0;
    if (ori->two || ori->e1 != 1) return 0;
    if (pen->scope + B + 1 > BFS_MAK_SLOTS || ori->scope + 2 > BFS_MAK_SLOTS) return 0;
    return B;
}
#endif
extern "C" int SRK_NAME(srk_align_blk)(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (off16) {
        if (nthreads == 128) return a->pen.two ? launch_blk3<int16_t, 128, true>(a, nwg, lds_bytes, st) : launch_blk3<int16_t, 128, false>(a, nwg, lds_bytes, st);
        if (nthreads == 512) return a->pen.two ? launch_blk3<int16_t, 512, true>(a, nwg, lds_bytes, st) : launch_blk3<int16_t, 512, false>(a, nwg, lds_bytes, st);
#if SR_SYMBITS == 2
        // the instrumented instance (tick counters [6..15], SR_PROFILE_TICKS=1): default shape only
        if (a->profile_ticks && a->pen.two) return launch_blk3<int16_t, 256, true, true>(a, nwg, lds_bytes, st);
#endif
        return a->pen.two ? launch_blk3<int16_t, 256, true>(a, nwg, lds_bytes, st) : launch_blk3<int16_t, 256, false>(a, nwg, lds_bytes, st);
    }
    return a->pen.two ? launch_blk3<int32_t, 256, true>(a, nwg, lds_bytes, st) : launch_blk3<int32_t, 256, false>(a, nwg, lds_bytes, st);
}
