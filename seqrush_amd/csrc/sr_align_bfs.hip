// sr_align_bfs.hip -- translation unit of the level-synchronous biWFA kernel (see sr_align_bfs.inc)
#include "sr_dev_common.h"
namespace SR_NS {
#include "sr_align_bfs.inc"
}  // namespace
using namespace SR_NS;

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
extern "C" int SRK_NAME(srk_align_bfs)(const SrAlignArgs *a, int nwg, size_t lds_bytes, int off16, int nthreads, void *stream) {
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
