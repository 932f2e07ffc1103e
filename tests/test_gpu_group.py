"""Device groups (include/reef_msm.h section 5): the multi-GPU split behind the C ABI, in one process -- what a Rust prover can call
(Reef is ONE process: src/backend/main.rs:82, src/backend/framework.rs:81-166; the MSM call sites the group serves:
framework.rs:668-721, src/backend/commitment.rs:187).  A test box has one GPU, so `devices[]` repeats ordinal 0 (the header allows
it): every member, every exchange and every split runs; only the peer copies between DIFFERENT devices do not (members of one
device write their partial sums in place).  Every result is compared with the C oracle (oracle/pasta_ref.c)."""
import ctypes
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CURVE = {"pallas": 0, "vesta": 1}


@pytest.fixture(scope="module")
def key(cref):
    out = {}
    for cid in (0, 1):
        n = 5003                                            # not divisible by 2, 3 or 8
        bases = cref.gen_bases_ap(cid, 401 + cid, 7, n)
        bases[n // 2] = 0                                   # an identity base
        out[cid] = bases
    return out


@pytest.mark.parametrize("members", [1, 2, 3, 8])
@pytest.mark.parametrize("split", ["windows", "points"])
@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_group_msm_equals_the_oracle(name, split, members, gpu_lib, cref, key):
    """One MSM over a group of 1 / 2 / 3 / 8 members on device 0, both splits, both curves, the three exchanges (RCCL: a one-rank communicator, every member but
    member 0 sends its partial sum to its own rank); host and device
    scalars; prefixes of the key (n < key length, n smaller than the member count, n = 0); both scalar conventions."""
    from reef_amd import msm
    cid = CURVE[name]
    bases = key[cid]
    n = bases.shape[0]
    sp = msm.SPLIT_WINDOWS if split == "windows" else msm.SPLIT_POINTS
    sc = cref.gen_scalars(cid, 55 + members, n, kind=0)
    sc_w = cref.gen_scalars(cid, 56, n, kind=1)
    canon = cref.gen_scalars(cid, 55 + members, n, kind=0, mont=False)
    want = cref.compress(cid, cref.msm_pippenger(cid, bases, sc, threads=4))
    want_w = cref.compress(cid, cref.msm_pippenger(cid, bases, sc_w, threads=4))
    for exchange in (msm.EXCHANGE_PEER, msm.EXCHANGE_HOST, msm.EXCHANGE_RCCL):
        with msm.MsmGroup(cid, bases, [0] * members, split=sp, exchange=exchange) as g:
            info = g.info()
            assert info["members"] == members and info["distinct_devices"] == 1 and info["split"] == split
            assert info["exchange"] == {msm.EXCHANGE_PEER: "peer", msm.EXCHANGE_HOST: "host-staged", msm.EXCHANGE_RCCL: "rccl"}[exchange]
            if split == "points":
                assert sum(info["key_points"]) == n and max(info["key_points"]) - min(info["key_points"]) <= 1
            else:
                assert info["key_points"] == [n] + [0] * (members - 1)        # one resident copy per DEVICE
            assert msm.compress(cid, g.msm(sc)) == want
            assert msm.compress(cid, g.msm(sc_w)) == want_w
            assert msm.compress(cid, g.msm(canon, is_mont=False)) == want
            dsc = msm.DeviceBuffer.from_host(sc)
            assert msm.compress(cid, g.msm(dsc, n)) == want
            for m in (0, 1, members - 1 if members > 1 else 2, 700, n - 1):
                exp = cref.compress(cid, cref.msm_pippenger(cid, bases[:m].copy(), sc[:m].copy(), threads=2)) if m else bytes(32)
                assert msm.compress(cid, g.msm(sc[:m].copy() if m else np.zeros((0, 4), np.uint64), m)) == exp, (m, exchange)
            assert msm.compress(cid, g.msm(sc)) == want                       # and whole again after the prefixes


@pytest.mark.parametrize("groups", [0, 4])
def test_group_on_keys_that_are_not_pre_shifted(groups, gpu_lib, cref, key):
    """The group takes the key options of reef_msm_ctx_create: plain keys (a bucket group per window) and partial precompute."""
    from reef_amd import msm
    cid = 0
    bases = key[cid]
    sc = cref.gen_scalars(cid, 91, bases.shape[0], kind=0)
    want = cref.compress(cid, cref.msm_pippenger(cid, bases, sc, threads=4))
    for sp in (msm.SPLIT_WINDOWS, msm.SPLIT_POINTS):
        with msm.MsmGroup(cid, bases, [0, 0, 0], split=sp, bucket_groups=groups) as g:
            assert msm.compress(cid, g.msm(sc)) == want


@pytest.mark.parametrize("members", [2, 3, 8])
def test_group_rows_dealt_out_whole(members, gpu_lib, cref, key):
    """HyraxPC::commit over a group (commitment.rs:187): rows in contiguous blocks, a row count that the member count does not
    divide, fewer rows than members, blinds; full-width rows, symbol-sized rows as field elements and as bytes; rows == 1 with a
    blind is split by window with the blind term on member 0 only."""
    from reef_amd import msm
    cid = 0
    bases = key[cid]
    h = cref.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
    with msm.MsmGroup(cid, bases, [0] * members, split=msm.SPLIT_WINDOWS) as g:
        for rows, row_len, bound in ((7, 600, 0), (members - 1, 900, 0), (37, 512, 7), (1300, 300, 131)):
            kind = 2 if bound else 0
            sc = cref.gen_scalars(cid, 4242 + rows, rows * row_len, kind=kind, small_bound=bound)
            bl = cref.gen_scalars(cid, 8, rows)
            kb = bases[:row_len].copy()
            exp = cref.compress(cid, cref.row_msm(cid, kb, sc, rows, row_len, h=h, blinds=bl, threads=8))
            exp_nb = cref.compress(cid, cref.row_msm(cid, kb, sc, rows, row_len, threads=8))
            assert msm.compress(cid, g.msm_rows(sc, rows, row_len, blinds=bl, h=h)) == exp, (rows, row_len, bound)
            assert msm.compress(cid, g.msm_rows(sc, rows, row_len)) == exp_nb
            dsc, dbl, dh = msm.DeviceBuffer.from_host(sc), msm.DeviceBuffer.from_host(bl), msm.DeviceBuffer.from_host(h)
            assert msm.compress(cid, g.msm_rows(dsc, rows, row_len, blinds=dbl, h=dh)) == exp
            if bound:
                canon = cref.gen_scalars(cid, 4242 + rows, rows * row_len, kind=2, small_bound=bound, mont=False)
                sym = np.ascontiguousarray(canon[:, 0].astype(np.uint8))
                bits = max(1, (bound - 1).bit_length())
                assert msm.compress(cid, g.msm_rows_symbols(sym, rows, row_len, bits, blinds=bl, h=h)) == exp
                assert msm.compress(cid, g.msm_rows_symbols(sym[:row_len].copy(), 1, row_len, bits, blinds=bl[:1].copy(), h=h)) == exp[:32]
        # CE::commit with a blind: one row, split by window
        n = bases.shape[0]
        v = cref.gen_scalars(cid, 77, n)
        b = cref.gen_scalars(cid, 78, 1)
        exp1 = cref.compress(cid, cref.row_msm(cid, bases, v, 1, n, h=h, blinds=b, threads=4))
        assert msm.compress(cid, g.msm_rows(v, 1, n, blinds=b, h=h)) == exp1
        assert msm.compress(cid, g.msm_rows(v, 1, n)) == cref.compress(cid, cref.msm_pippenger(cid, bases, v, threads=4))
        assert msm.compress(cid, g.msm(v)) == cref.compress(cid, cref.msm_pippenger(cid, bases, v, threads=4))     # the split ctx is untouched by rows
    with msm.MsmGroup(cid, bases, [0, 0], split=msm.SPLIT_POINTS) as g:
        with pytest.raises(msm.ReefError) as e:
            g.msm_rows(cref.gen_scalars(cid, 1, 20), 2, 10)
        assert e.value.status == 1 and "REEF_SPLIT_WINDOWS" in str(e.value)


def test_group_rccl_exchange_runs_in_a_process_of_its_own_and_fails_loudly_without_the_library(cref, tmp_path):
    """REEF_EXCHANGE_RCCL in a fresh process (RCCL is opened once per process): one commitment with a blind and one MSM over three
    members, both splits, against the C oracle; and with REEF_RCCL_LIB pointing nowhere the group's creation fails with the
    loader's message (status 2) -- no fallback to another exchange."""
    import os
    import subprocess
    import sys
    cid, n = 0, 3000
    bases = cref.gen_bases_ap(cid, 321, 7, n)
    sc = cref.gen_scalars(cid, 17, n)
    b = cref.gen_scalars(cid, 18, 1)
    h = cref.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
    want = cref.compress(cid, cref.msm_pippenger(cid, bases, sc, threads=4)).hex()
    want_b = cref.compress(cid, cref.row_msm(cid, bases, sc, 1, n, h=h, blinds=b, threads=4)).hex()
    np.savez(tmp_path / "in.npz", bases=bases, sc=sc, b=b, h=h)
    code = f"""
import numpy as np, sys
from reef_amd import msm
d = np.load(r'{tmp_path / "in.npz"}')
try:
    for sp in (msm.SPLIT_WINDOWS, msm.SPLIT_POINTS):
        with msm.MsmGroup(0, d['bases'], [0, 0, 0], split=sp, exchange=msm.EXCHANGE_RCCL) as g:
            assert g.info()['exchange'] == 'rccl'
            for rep in range(3):
                print('msm', msm.compress(0, g.msm(d['sc'])).hex())
            if sp == msm.SPLIT_WINDOWS:
                print('blind', msm.compress(0, g.msm_rows(d['sc'], 1, {n}, blinds=d['b'], h=d['h'])).hex())
except msm.ReefError as e:
    print('error', e.status, str(e))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.split("\n")
    assert [l for l in lines if l.startswith("msm ")] == ["msm " + want] * 6, r.stdout + r.stderr[-1500:]
    assert [l for l in lines if l.startswith("blind ")] == ["blind " + want_b], r.stdout
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(env, REEF_RCCL_LIB="/nonexistent/librccl.so"), cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("error 2 ") and "REEF_EXCHANGE_RCCL" in r.stdout and "/nonexistent/librccl.so" in r.stdout, r.stdout


def test_group_at_the_headline_size_dlog_property(gpu_lib):
    """configs[1]'s 2^20-point key over groups of 2 and 8 members: both splits against the discrete-log closed form (bases
    B_i = (k0 + i d) G, so sum s_i B_i = (sum s_i (k0 + i d)) G) and against one context."""
    from oracle.pasta_oracle import CURVES
    from reef_amd import msm
    cid, n, k0, d = 0, 1 << 20, 12345, 7
    C = CURVES["pallas"]
    dbases = msm.gen_bases(cid, k0, d, n, device=True)
    sc = msm.gen_scalars(cid, 0x5EEF, n, kind=0, mont=False)
    cols = [sc[:, j].astype(object) for j in range(4)]
    idx = np.arange(n, dtype=object)
    total = sum((int(np.sum(cols[j])) * k0 + int(np.sum(cols[j] * idx)) * d) << (64 * j) for j in range(4)) % C.order
    want = C.compress(C.mul(total, C.gen))
    for members, sp in ((2, msm.SPLIT_WINDOWS), (8, msm.SPLIT_WINDOWS), (8, msm.SPLIT_POINTS), (3, msm.SPLIT_POINTS)):
        with msm.MsmGroup(cid, dbases, [0] * members, n, split=sp) as g:
            assert msm.compress(cid, g.msm(sc, is_mont=False)) == want, (members, sp)


def test_groups_from_several_threads_and_bad_arguments(gpu_lib, cref, key):
    """A group serialises its calls; two groups work side by side; argument errors are REEF_ERR_ARG with a message."""
    from reef_amd import _ffi, msm
    cid = 0
    bases = key[cid]
    n = bases.shape[0]
    scs = [cref.gen_scalars(cid, 300 + j, n, kind=j % 2) for j in range(4)]
    want = [cref.compress(cid, cref.msm_pippenger(cid, bases, s, threads=4)) for s in scs]
    errs = []
    with msm.MsmGroup(cid, bases, [0, 0, 0], split=msm.SPLIT_WINDOWS) as ga, msm.MsmGroup(cid, bases, [0, 0], split=msm.SPLIT_POINTS) as gb:
        def work(t):
            try:
                for rep in range(6):
                    j = (t + rep) % 4
                    g = ga if (t + rep) % 2 else gb
                    assert msm.compress(cid, g.msm(scs[j])) == want[j], (t, rep)
            except BaseException as e:
                errs.append(repr(e))
        ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs[:2]
        with pytest.raises(msm.ReefError) as e:
            ga.msm(np.zeros((n + 1, 4), np.uint64))
        assert e.value.status == 1
    lib = _ffi.load()
    h = ctypes.c_void_p()
    devs = (ctypes.c_int * 2)(0, lib.reef_device_count())                          # the second ordinal does not exist
    assert lib.reef_msm_group_create(ctypes.byref(h), 0, bases.ctypes.data, n, 0, None, devs, 2, None) == 1
    assert b"visible" in lib.reef_last_error()


def test_group_rows_at_the_baseline_document_shape_dlog_property(gpu_lib):
    """BASELINE configs[3]: the Hyrax commitment of a 16 MiB DNA document (4096 rows of 8192 three-bit symbols, commitment.rs:173-187) with the
    rows dealt out over 8 / 3 members, from host bytes: rows of every member against the discrete-log closed form (generators in arithmetic
    progression), and the whole result against one context."""
    from oracle.pasta_oracle import CURVES
    from reef_amd import msm
    C = CURVES["pallas"]
    rows, row_len, bits, k0, d = 4096, 8192, 3, 0xFEED, 3
    doc = np.random.default_rng(0xD0C).integers(0, 7, size=(rows, row_len), dtype=np.uint8)
    flat = np.ascontiguousarray(doc.reshape(-1))
    gens = msm.gen_bases("pallas", k0, d, row_len, device=True)
    with msm.MsmContext("pallas", gens, row_len) as one:
        ref = msm.compress("pallas", one.msm_rows_symbols(flat, rows, row_len, bits))
    for members in (8, 3):
        with msm.MsmGroup("pallas", gens, [0] * members, row_len, split=msm.SPLIT_WINDOWS) as g:
            got = g.msm_rows_symbols(flat, rows, row_len, bits)
        assert msm.compress("pallas", got) == ref, members
        for r in (0, rows // members - 1, rows // members, rows // 2 + 1, rows - 1):         # the edges of the members' blocks
            dl = sum(int(v) * (k0 + j * d) for j, v in enumerate(doc[r].tolist())) % C.order
            assert msm.compress("pallas", got[r].copy()) == C.compress(C.mul(dl, C.gen)), (members, r)


@pytest.mark.parametrize("members", [2, 3, 8])
@pytest.mark.parametrize("exchange", ["peer", "rccl"])
def test_window_split_scalars_fanned_out_from_member_zero(members, exchange, gpu_lib, cref, key):
    """Round 6: REEF_SCALARS_FANOUT -- the host scalars of a windows group are uploaded ONCE, to devices[0], and every other member
    fetches them from there with a peer copy on its own stream after the upload's event (a member on devices[0] too: the calls a node runs).  Same points
    as the default (every member uploads for itself) and as the oracle; prefixes; repeated calls reuse the staging buffers; device scalars and points
    groups ignore the option."""
    from reef_amd import msm
    cid = 0
    bases = key[cid]
    n = bases.shape[0]
    scs = [cref.gen_scalars(cid, 810 + j, n, kind=j % 2) for j in range(3)]
    want = [cref.compress(cid, cref.msm_pippenger(cid, bases, s, threads=4)) for s in scs]
    ex = msm.EXCHANGE_PEER if exchange == "peer" else msm.EXCHANGE_RCCL
    with msm.MsmGroup(cid, bases, [0] * members, split=msm.SPLIT_WINDOWS, exchange=ex, scalars=msm.SCALARS_FANOUT) as g:
        for rep in range(2):
            for j, s in enumerate(scs):
                assert msm.compress(cid, g.msm(s)) == want[j], (rep, j)
        for m in (1, members, 999):
            assert msm.compress(cid, g.msm(scs[0][:m].copy(), m)) == cref.compress(cid, cref.msm_pippenger(cid, bases[:m].copy(), scs[0][:m].copy(), threads=2)), m
        assert msm.compress(cid, g.msm(np.zeros((0, 4), np.uint64), 0)) == bytes(32)
        dsc = msm.DeviceBuffer.from_host(scs[1])
        assert msm.compress(cid, g.msm(dsc, n)) == want[1]
    with msm.MsmGroup(cid, bases, [0] * members, split=msm.SPLIT_POINTS, scalars=msm.SCALARS_FANOUT) as g:
        assert msm.compress(cid, g.msm(scs[2])) == want[2]


@pytest.mark.parametrize("split", ["windows", "points"])
def test_group_call_says_where_its_time_went(split, gpu_lib, cref, key):
    """reef_msm_group_enable_timing / _last_timing: the phases of the last split call -- distribution, the members' streams, the combine -- are
    reported, are consistent with each other, and the call computes the same point with timing on as off."""
    from reef_amd import msm
    cid = 1
    bases = key[cid]
    n = bases.shape[0]
    sc = cref.gen_scalars(cid, 99, n)
    want = cref.compress(cid, cref.msm_pippenger(cid, bases, sc, threads=4))
    sp = msm.SPLIT_WINDOWS if split == "windows" else msm.SPLIT_POINTS
    for scal_mode in (msm.SCALARS_EACH, msm.SCALARS_FANOUT):
        with msm.MsmGroup(cid, bases, [0, 0, 0], split=sp, scalars=scal_mode) as g:
            with pytest.raises(msm.ReefError):
                g.last_timing()                                  # nothing timed yet
            assert msm.compress(cid, g.msm(sc)) == want
            g.enable_timing(True)
            for _ in range(2):
                assert msm.compress(cid, g.msm(sc)) == want
            t = g.last_timing()
            assert len(t["member_issue_ms"]) == 3 and len(t["member_stream_ms"]) == 3
            assert all(v > 0 for v in t["member_stream_ms"]) and all(v > 0 for v in t["member_issue_ms"])
            assert 0 < t["distribute_ms"] <= t["members_done_ms"] <= t["total_ms"] and t["combine_ms"] > 0
            assert t["total_ms"] < 50 and max(t["member_issue_ms"]) <= t["distribute_ms"] + 0.05
            g.enable_timing(False)
            assert msm.compress(cid, g.msm(sc)) == want
