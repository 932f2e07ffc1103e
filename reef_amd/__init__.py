"""reef_amd -- MI355X (gfx950) backend for the Pasta-curve MSM hot path of eniac/Reef.

Only what that path needs lives here:
  csrc/        hand-written HIP kernels + the C ABI (libreef_msm.so, include/reef_msm.h)
  _ffi.py      ctypes binding of the C ABI
  msm.py       resident keys, MSMs, row commitments, folds, normalisation (buffer marshalling)
  provider.py  host-side mirror of the nova-snark provider interface Reef calls
  sumcheck.py  sum-check vector kernels of nlookup witness generation (row N2)
  mle.py       bound rows / evaluation of the document polynomial at proof end (row N3)
  distributed.py  one-process-per-GPU sharding of large MSMs (RCCL all-gather of partial sums)
"""
__all__ = ["msm", "provider", "distributed", "sumcheck", "mle"]
