// Pallas instantiation of the MSM engine (coordinates in Fp, scalars in Fq).
#define REEF_CURVE 0
#include "msm_kernels.inc"
#include "sumcheck_kernels.inc"
#include "mle_kernels.inc"
#include "merkle_kernels.inc"
#include "keygen_kernels.inc"
#include "engine.inc"
namespace reef {
const CurveVTable *pallas_vtable() {
    static const CurveVTable vt = make_vtable<0>();
    return &vt;
}
}
