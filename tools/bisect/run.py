import ctypes, glob, os, sys
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
def run(path):
    lib = ctypes.CDLL(path)
    out = np.zeros(64 * 36, dtype=np.uint32)
    rc = lib.dd_run(out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    return out
ref = run(os.path.join(here, "dd_ref.so"))
for p in sorted(glob.glob(os.path.join(here, "dd_*.so")), key=lambda s: (len(s), s)):
    if p.endswith("dd_ref.so"): continue
    o = run(p)
    print(os.path.basename(p), "OK" if (o == ref).all() else "BAD", flush=True)
