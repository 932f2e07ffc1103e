// Two four-wave doublings inlined back to back against the one-wave doubling applied twice, in the SAME kernel:
// dd_run(out_host[64*36]) returns 1 word per lane: 0 = equal, else a bit mask of the differing coordinates (x=1, y=2, zz=4, zzz=8).
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
    const xyzz e = xyzz_dbl<0>(xyzz_dbl<0>(a));
    // projective comparison: x1*zz2 == x2*zz1, y1*zzz2 == y2*zzz1
    u32 m = 0;
    if (!fe_is_zero<0>(fe_sub<0, 2>(fe_mul<0>(r.x, e.zz), fe_mul<0>(e.x, r.zz)))) m |= 1;
    if (!fe_is_zero<0>(fe_sub<0, 2>(fe_mul<0>(r.y, e.zzz), fe_mul<0>(e.y, r.zzz)))) m |= 2;
    if (!fe_is_zero<0>(fe_sub<0, 2>(fe_mul<0>(fe_sqr<0>(r.zz), r.zz), fe_sqr<0>(r.zzz)))) m |= 4;       // zz^3 == zzz^2
    out[role * 64 + lane] = m;                      // every wave reports: they must all hold the same (right) point
}
extern "C" int dd_run(unsigned *out_host) {
    u32 *d = nullptr;
    if (hipMalloc(&d, 64 * 36 * 4) != hipSuccess) return 1;
    (void)hipMemset(d, 0, 64 * 36 * 4);
    hipLaunchKernelGGL(k2, dim3(1), dim3(256), 0, 0, d);
    if (hipMemcpy(out_host, d, 64 * 36 * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    (void)hipFree(d);
    return 0;
}
