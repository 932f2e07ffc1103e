import ctypes, glob, os, sys
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
for p in sorted(glob.glob(os.path.join(here, "sc_*.so"))):
    lib = ctypes.CDLL(p)
    out = np.zeros(64 * 36, dtype=np.uint32)
    assert lib.dd_run(out.ctypes.data_as(ctypes.c_void_p)) == 0
    per_role = [sorted(set(out[64 * r:64 * r + 64].tolist())) for r in range(4)]
    print(os.path.basename(p), "OK" if not out[:256].any() else "BAD masks per wave %s" % per_role, flush=True)
