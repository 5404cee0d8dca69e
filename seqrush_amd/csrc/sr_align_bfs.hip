// sr_align_bfs.hip -- translation unit of the level-synchronous biWFA kernel (see sr_align_bfs.inc)
#include "sr_dev_common.h"
#ifdef SR_BFS_WIDE
// -DSR_BFS_WIDE: the instance for penalty sets whose ring is deeper than 32 levels (scope + 1 up to 128; e.g. a second
// gap piece that opens at 40): 128 slots of per-level maxima per aligner, 8 segments (16 aligners) per pass so that the
// static LDS stays near 22 KB.  Round 3: replaces sr_align_kernel (sr_align_v3.hip), which was the only kernel for such
// penalties and was built for 2-bit buffers only.
#define BFS_MAK_SLOTS 128
#undef SR_BFS_MAXACT
#define SR_BFS_MAXACT 8
#define SR_BFS_SUB wide
#define SRK_BFS_ENTRY SRK_NAME(srk_align_bfs_wide)
#else
#define SR_BFS_SUB lvl
#define SRK_BFS_ENTRY SRK_NAME(srk_align_bfs)
#endif
namespace SR_NS { namespace SR_BFS_SUB {
#include "sr_align_bfs.inc"
} }  // namespaces
using namespace SR_NS::SR_BFS_SUB;

template <typename OT, int NT, bool TWO>
static int launch_bfs3(const SrAlignArgs *a, int nwg, size_t lds_bytes, hipStream_t st) {
    if (lds_bytes > 32 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)sr_align_bfs_kernel<OT, NT, TWO>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((sr_align_bfs_kernel<OT, NT, TWO>), dim3(nwg), dim3(NT), lds_bytes, st, *a);
    return (int)hipGetLastError();
}
template <typename OT, int NT>
static int launch_bfs(const SrAlignArgs *a, int nwg, size_t lds_bytes, hipStream_t st) {
    return a->pen.two ? launch_bfs3<OT, NT, true>(a, nwg, lds_bytes, st)
                      : launch_bfs3<OT, NT, false>(a, nwg, lds_bytes, st);
}
extern "C" int SRK_BFS_ENTRY(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (off16) {
        if (nthreads == 128) return launch_bfs<int16_t, 128>(a, nwg, lds_bytes, st);
        if (nthreads == 512) return launch_bfs<int16_t, 512>(a, nwg, lds_bytes, st);
        return launch_bfs<int16_t, 256>(a, nwg, lds_bytes, st);
    }
    if (nthreads == 128) return launch_bfs<int32_t, 128>(a, nwg, lds_bytes, st);
    if (nthreads == 512) return launch_bfs<int32_t, 512>(a, nwg, lds_bytes, st);
    return launch_bfs<int32_t, 256>(a, nwg, lds_bytes, st);
}
