#!/usr/bin/env python3
"""Row N4: wall time of the Poseidon Merkle commitment (stand-in constants) for documents of 2^16 .. 2^26 symbols,
hashes per second, and the oracle's pure-Python rate beside it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
from oracle import merkle_oracle as M
from reef_amd import _ffi, merkle
from reef_amd.sumcheck import ints_to_array
lib = _ffi.load()
p = M.standin_params()
rc = ints_to_array(p.rc); mds = ints_to_array([x for r in p.mds for x in r])
pp = merkle.PoseidonParams(5, p.rf, p.rp, 0, rc.ctypes.data, mds.ctypes.data, (ctypes.c_uint64 * 4)(p.tag_leaf & (2**64 - 1), p.tag_leaf >> 64, 0, 0),
                           (ctypes.c_uint64 * 4)(p.tag_node & (2**64 - 1), p.tag_node >> 64, 0, 0))
for logn in [int(a) for a in sys.argv[1:]] or [16, 20, 24, 26]:
    n = 1 << logn
    doc = np.random.default_rng(1).integers(0, 131, size=n, dtype=np.uint32)
    root = np.zeros((1, 4), dtype=np.uint64)
    for rep in range(2):
        t0 = time.perf_counter()
        rcode = lib.reef_merkle_commit(0, ctypes.byref(pp), doc.ctypes.data, n, 0, False, None, 0, root.ctypes.data)
        dt = time.perf_counter() - t0
    assert rcode == 0
    h = merkle.nodes(n)
    print(f"2^{logn} symbols: {h} hashes in {dt * 1e3:.2f} ms = {h / dt / 1e6:.1f} M hashes/s (document uploaded, root returned)", flush=True)
# the same tree in blocks over the visible devices (reef_merkle_commit_devices; ordinals repeat when the box has fewer GPUs than members)
from reef_amd import msm
vis = msm.device_count()
for logn in (20, 24, 27):
    n = 1 << logn
    doc = np.random.default_rng(1).integers(0, 131, size=n, dtype=np.uint32)
    root1 = np.zeros((1, 4), dtype=np.uint64)
    assert lib.reef_merkle_commit(0, ctypes.byref(pp), doc.ctypes.data, n, 0, False, None, 0, root1.ctypes.data) == 0
    for members in (2, 4, 8):
        devs = (ctypes.c_int * members)(*[i % vis for i in range(members)])
        blocks = ctypes.c_uint32(0)
        root = np.zeros((1, 4), dtype=np.uint64)
        for rep in range(2):
            t0 = time.perf_counter()
            rcode = lib.reef_merkle_commit_devices(0, ctypes.byref(pp), doc.ctypes.data, n, False, devs, members, None, root.ctypes.data, ctypes.byref(blocks))
            dt = time.perf_counter() - t0
        assert rcode == 0 and (root == root1).all()
        print(f"2^{logn} symbols in {blocks.value} blocks over {members} members on {min(vis, members)} device(s): {dt * 1e3:.2f} ms (root equal to one device's)", flush=True)
t0 = time.perf_counter(); M.commit(list(range(512)), p); dt = time.perf_counter() - t0
print(f"oracle (pure Python): {merkle.nodes(512) / dt:.0f} hashes/s on one host core")
