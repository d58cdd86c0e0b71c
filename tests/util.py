"""Helpers for the parity tests: seeded Arrow-shaped inputs and comparisons.

Parity bar (BASELINE.json north_star): bit-exact for integer / index / bitmap results; f64/f32
arithmetic, trig and sums within RTOL = 1e-6 relative of the oracle.
"""
import numpy as np

from rust_dataframe_amd import _abi as A

RTOL = 1e-6


def rand_values(rng, dtype, n, kind="plain"):
    npdt = A.NP_OF[dtype]
    if dtype in (A.F32, A.F64):
        if kind == "unit":
            v = rng.uniform(-1.0, 1.0, n)
        elif kind == "pos":
            v = rng.uniform(0.01, 50.0, n)
        elif kind == "special" and n >= 8:
            v = rng.uniform(-100.0, 100.0, n)
            v[:8] = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-300, -1e300, 1.0]
            rng.shuffle(v)
        else:
            v = rng.uniform(-100.0, 100.0, n)
        return v.astype(npdt)
    info = np.iinfo(npdt)
    if kind == "extreme":
        v = rng.integers(info.min, info.max, n, dtype=npdt, endpoint=True)
        if n >= 2:
            v[0], v[1] = info.min, info.max
        return v
    lo, hi = max(info.min, -1000), min(info.max, 1000)
    return rng.integers(lo, hi, n, endpoint=True).astype(npdt)


def make_chunks(rng, dtype, lens, null_frac=0.0, offset=0, kind="plain", nonzero=False):
    """A column: one HostArray per chunk length."""
    out = []
    for n in lens:
        v = rand_values(rng, dtype, n, kind)
        if nonzero:
            v[v == 0] = 1
        valid = None
        if null_frac > 0:
            valid = rng.uniform(size=n) >= null_frac
        elif null_frac < 0:  # all null
            valid = np.zeros(n, dtype=bool)
        out.append(A.HostArray.from_numpy(v, valid=valid, offset=offset, dtype=dtype, rng=rng))
    return out


def assert_arrays_match(got, exp, exact=None, what=""):
    """One output chunk vs the oracle's: length, null count, validity bits, values at valid slots."""
    assert got.dtype == exp.dtype, what
    assert got.length == exp.length, f"{what}: length {got.length} != {exp.length}"
    assert got.null_count == exp.null_count, f"{what}: null_count {got.null_count} != {exp.null_count}"
    gm, em = got.valid_mask(), exp.valid_mask()
    assert np.array_equal(gm, em), f"{what}: validity bitmaps differ"
    gv, ev = got.to_numpy()[em], exp.to_numpy()[em]
    if exact is None:
        exact = got.dtype not in (A.F32, A.F64)
    if exact:
        if got.dtype in (A.F32, A.F64):
            assert np.array_equal(gv.view(np.uint64 if got.dtype == A.F64 else np.uint32),
                                  ev.view(np.uint64 if got.dtype == A.F64 else np.uint32)), f"{what}: values differ bitwise"
        else:
            assert np.array_equal(gv, ev), f"{what}: values differ"
    else:
        np.testing.assert_allclose(gv, ev, rtol=RTOL, atol=0, equal_nan=True, err_msg=what)


def assert_chunks_match(got, exp, exact=None, what=""):
    assert len(got) == len(exp), what
    for i, (g, e) in enumerate(zip(got, exp)):
        assert_arrays_match(g, e, exact, f"{what} chunk {i}")


def assert_scalar_close(got, exp, dtype, what=""):
    if exp is None or got is None:
        assert got is None and exp is None, f"{what}: {got} vs {exp}"
        return
    if dtype in (A.F32, A.F64):
        if np.isnan(exp):
            assert np.isnan(got), what
        else:
            assert abs(got - exp) <= RTOL * max(abs(exp), 1e-300) or got == exp, f"{what}: {got} vs {exp}"
    else:
        assert got == exp, f"{what}: {got} vs {exp}"
