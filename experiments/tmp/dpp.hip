#include <hip/hip_runtime.h>
__global__ void k(int *o, const int *i) {
    int x = i[threadIdx.x];
    int l = __builtin_amdgcn_update_dpp(-1, x, 0x138, 0xf, 0xf, false);
    int r = __builtin_amdgcn_update_dpp(-1, x, 0x130, 0xf, 0xf, false);
    o[threadIdx.x] = l * 3 + r;
}
