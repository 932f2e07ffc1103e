"""Row N3: bound rows / evaluation of the document polynomial (through the C ABI) against
oracle/mle_oracle.py, which tests/test_mle_oracle.py pins to the reference's `mle_partial` known
answers and to the line-by-line restatement of verifier_mle_eval.  Bit-exact (integers mod q/p)."""
import itertools

import numpy as np
import pytest

from oracle import mle_oracle
from oracle.pasta_oracle import P, Q, SplitMix64

pytestmark = pytest.mark.gpu

MOD = {"pallas": Q, "vesta": P}     # scalar field of the curve


def rand_fe(rng, mod):
    return (rng.next() << 192 | rng.next() << 128 | rng.next() << 64 | rng.next()) % mod


def test_reference_kat_on_gpu(gpu_lib):
    """`mle_partial` (r1cs.rs:2517-2578) inputs: boolean points read the table back."""
    from reef_amd import mle
    table = [1, 3, 8, 2, 9, 5, 13, 4]
    for x in itertools.product((0, 1), repeat=3):
        e = x[0] * 4 + x[1] * 2 + x[2]
        assert mle.evaluate("pallas", table, list(x)) == table[e]
        assert mle.evaluate("pallas", np.array(table, dtype=np.uint8), list(x)) == table[e]
        assert mle.verifier_mle_eval(table, list(x)) == table[e]            # under the reference's own name (r1cs_helper.rs:637)


@pytest.mark.parametrize("name", ["pallas", "vesta"])
@pytest.mark.parametrize("m,left,n_short", [(1, 0, 0), (1, 1, 0), (4, 2, 3), (9, 4, 0), (9, 9, 5), (9, 0, 1), (13, 6, 100), (14, 3, 0)])
def test_field_tables_vs_oracle(name, m, left, n_short, gpu_lib):
    """32-byte entries, canonical and Montgomery forms, ragged (zero-padded) lengths."""
    from reef_amd import mle
    from reef_amd.sumcheck import array_to_ints, ints_to_array
    mod = MOD[name]
    rng = SplitMix64(1000 + m * 17 + left)
    n = (1 << m) - n_short
    z = [rand_fe(rng, mod) for _ in range(n)]
    point = [rand_fe(rng, mod) for _ in range(m)]
    lz_ref, ev_ref = mle_oracle.bound_rows(z, point, left, mod)
    lz, ev = mle.bound_rows(name, z, point, left)
    assert ev == ev_ref
    assert lz == lz_ref
    # Montgomery (pasta ABI) form in, Montgomery form out
    R = 1 << 256
    zm = ints_to_array([v * R % mod for v in z])
    pm = ints_to_array([v * R % mod for v in point])
    lzm, evm = mle.bound_rows_raw(name, zm, pm, left, is_mont=True)
    assert array_to_ints(evm)[0] == ev_ref * R % mod
    assert array_to_ints(lzm) == [v * R % mod for v in lz_ref]


@pytest.mark.parametrize("dtype,bound", [(np.uint8, 131), (np.uint8, 7), (np.uint16, 259), (np.uint32, 1 << 32), (np.uint16, 1 << 16)])
@pytest.mark.parametrize("m,left", [(12, 6), (16, 8), (15, 2)])
def test_symbol_tables_vs_oracle(dtype, bound, m, left, gpu_lib):
    """Document symbols (framework.rs:978-1011: < |alphabet| + 3) as compact unsigned integers."""
    from reef_amd import mle
    from reef_amd.sumcheck import array_to_ints, ints_to_array
    rng = SplitMix64(31 * m + left + bound % 1000)
    n = (1 << m) - 37
    z = np.array([rng.next() % bound for _ in range(n)], dtype=dtype)
    if bound >= 1 << 16:
        z[::3] = bound - 1                     # worst case for the limb accumulators
    point = [rand_fe(rng, Q) for _ in range(m)]
    lz_ref, ev_ref = mle_oracle.bound_rows([int(v) for v in z], point, left, Q)
    lz, ev = mle.bound_rows("pallas", z, point, left)
    assert (lz, ev) == (lz_ref, ev_ref)
    R = 1 << 256
    lzm, evm = mle.bound_rows_raw("pallas", z, ints_to_array([v * R % Q for v in point]), left, is_mont=True)
    assert array_to_ints(evm)[0] == ev_ref * R % Q and array_to_ints(lzm) == [v * R % Q for v in lz_ref]


def test_blind_combination_and_empty(gpu_lib):
    """sum_i L_i * blind_i is the one-column case; an empty table evaluates to zero."""
    from reef_amd import mle
    rng = SplitMix64(5)
    blinds = [rand_fe(rng, Q) for _ in range(64)]
    point = [rand_fe(rng, Q) for _ in range(6)]
    L = mle_oracle.eq_evals(point, Q)
    lz, ev = mle.bound_rows("pallas", blinds, point, 6)
    assert lz == [sum(a * b for a, b in zip(L, blinds)) % Q] and ev == lz[0]
    assert mle.bound_rows("pallas", [], point, 3) == ([0] * 8, 0)
    assert mle.bound_rows("pallas", [7], [], 0) == ([7], 7)          # zero variables: the constant


@pytest.mark.parametrize("m,left,dtype,bound", [(21, 10, np.uint8, 131), (25, 12, np.uint8, 7), (27, 13, np.uint16, 55000)])
def test_baseline_size_properties(m, left, dtype, bound, gpu_lib):
    """BASELINE.json configs[2]/[3]/[4] sizes (1 MiB ASCII: 2^21; 16 MiB DNA: 2^25; 64 MiB UTF-8: 2^27 two-byte
    symbols), device resident.
    Sampled columns of LZ against the oracle's definition, <LZ, R> against eval, a boolean left
    half reads a document row back, a boolean point reads a symbol back."""
    from reef_amd import mle, msm
    from reef_amd.sumcheck import array_to_ints, ints_to_array
    n = (1 << m) - 12345
    rng = np.random.default_rng(m)
    z = rng.integers(0, bound, size=n, dtype=dtype)
    dz = msm.DeviceBuffer.from_host(z.view(np.uint8))
    sm = SplitMix64(99)
    point = [rand_fe(sm, Q) for _ in range(m)]
    cols = 1 << (m - left)
    lz, ev = mle.bound_rows_raw("pallas", dz, ints_to_array(point), left, is_mont=False, n=n, elem_bytes=z.itemsize)
    lz, ev = array_to_ints(lz), array_to_ints(ev)[0]
    L = mle_oracle.eq_evals(point[:left], Q)
    Rv = mle_oracle.eq_evals(point[left:], Q)
    zp = np.zeros(1 << m, dtype=dtype)
    zp[:n] = z
    zmat = zp.reshape(1 << left, cols)
    for j in (0, 1, cols // 2 + 3, cols - 1):
        assert lz[j] == sum(int(v) * L[i] for i, v in enumerate(zmat[:, j]) if v) % Q
    assert ev == sum(a * b for a, b in zip(lz, Rv)) % Q
    row = 0b1011 % (1 << left)
    bpt = [(row >> (left - 1 - k)) & 1 for k in range(left)] + point[left:]
    lz_b, _ = mle.bound_rows_raw("pallas", dz, ints_to_array(bpt), left, is_mont=False, n=n, elem_bytes=z.itemsize)
    assert array_to_ints(lz_b) == [int(v) for v in zmat[row]]
    idx = 987654 % n
    ipt = [(idx >> (m - 1 - k)) & 1 for k in range(m)]
    _, ev_b = mle.bound_rows_raw("pallas", dz, ints_to_array(ipt), left, is_mont=False, n=n, elem_bytes=z.itemsize, want_rows=False)
    assert array_to_ints(ev_b)[0] == int(z[idx])


def test_error_paths(gpu_lib):
    from reef_amd import mle, msm
    with pytest.raises(msm.ReefError):
        mle.bound_rows("pallas", [1, 2, 3], [5], 0)            # n > 2^num_vars
    with pytest.raises(msm.ReefError):
        mle.bound_rows("pallas", [1, 2], [5], 2)               # left_vars > num_vars
