// Stand-alone reproducer: two four-wave doublings inlined back to back (see ec_coop.h).  extern "C" dd_run(out_host[64*36]).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "ec_coop.h"
using namespace reef;
__global__ void __launch_bounds__(256) k2(u32 *out) {
    __shared__ CoopLds L;
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    affine g;
    g.x = fe_canon<0>(fe_neg<0, 2>(fe_one<0>()));
    g.y = fe_canon<0>(fe_dbl<0>(fe_one<0>()));
    xyzz a = xyzz_dbl<0>(xyzz_from_affine<0>(g));
    for (int k = 0; k < (lane & 7); ++k) a = xyzz_madd<0>(a, g);
    xyzz r = xyzz_dbl_coop<0>(a, L, role, lane);
    r = xyzz_dbl_coop<0>(r, L, role, lane);
    if (role == 0) {
        const xyzz_mem m = xyzz_to_mem(r);
        for (int k = 0; k < 36; ++k) out[lane * 36 + k] = m.w[k];
    }
}
extern "C" int dd_run(unsigned *out_host) {
    u32 *d = nullptr;
    if (hipMalloc(&d, 64 * 36 * 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(k2, dim3(1), dim3(256), 0, 0, d);
    if (hipMemcpy(out_host, d, 64 * 36 * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    hipFree(d);
    return 0;
}
