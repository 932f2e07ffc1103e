// Vesta instantiation of the MSM engine (coordinates in Fq, scalars in Fp).
#define REEF_CURVE 1
#include "msm_kernels.inc"
#include "sumcheck_kernels.inc"
#include "mle_kernels.inc"
#include "merkle_kernels.inc"
#include "keygen_kernels.inc"
#include "engine.inc"
namespace reef {
const CurveVTable *vesta_vtable() {
    static const CurveVTable vt = make_vtable<1>();
    return &vt;
}
}
