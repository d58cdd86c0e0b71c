"""eval_lean_kernel (rdf_eval_lean.hip): the interpreter's branch-free kernel for aggregate programs over 8-byte columns.
Held to the oracle like every other device path, and to eval_kernel bit for bit, any NaN counting as NaN (the two must be interchangeable).
[Evaluate::calculate src/evaluation.rs:97-323, BooleanFilter::eval_to_array src/expression.rs:766-861,
 AggregateFunctions src/functions/aggregate.rs:12-93]"""
import os
import sys

import numpy as np
import pytest

from rust_dataframe_amd import _abi as A

from util import make_chunks

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def interp():
    """the product with the specialised kernels and the run-time compiler off: every program is interpreted"""
    from rust_dataframe_amd import lib
    api = lib.api()
    if lib.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible")
    lib.set_option("spec", 0)
    lib.set_option("fast_filter", 0)
    lib.set_option("jit", 0)
    lib.set_option("interp_lean", 1)
    yield api, lib
    lib.set_option("spec", 1)
    lib.set_option("fast_filter", 1)
    lib.set_option("jit", 1)
    lib.set_option("interp_lean", 1)


def _programs(e):
    a, b, k, u = e.col(0), e.col(1), e.col(2), e.col(3)
    gt = e.op("gt", a, e.scalar(0.5))
    return {
        "filter_sum": dict(values=[e.op("add", a, e.scalar(0.0))], filt=gt),
        "filter_and_or_not": dict(values=[a, b], filt=e.op("or", e.op("and", gt, e.op("lt", b, e.scalar(0.25))), e.op("not", e.op("ge", a, b)))),
        "literal_on_the_left": dict(values=[e.op("subtract", e.scalar(1.0), a), e.op("divide", e.scalar(2.0), e.op("add", e.op("multiply", a, a), e.scalar(1.0)))],
                                    filt=e.op("lt", e.scalar(0.25), b)),
        "bushy_with_temporaries": dict(values=[e.op("multiply", e.op("add", a, b), e.op("subtract", a, e.op("multiply", b, e.scalar(3.0))))], filt=-1),
        "integer_sums": dict(values=[e.op("add", e.op("multiply", k, k), e.scalar(1, A.I64)), e.op("subtract", e.scalar(7, A.U64), u), k, u],
                             filt=e.op("ne", k, e.scalar(0, A.I64))),
        "casts_to_f64": dict(values=[e.op("add", e.cast(k, A.F64), a), e.op("multiply", e.cast(u, A.F64), b)], filt=e.op("gt", u, k)),
        "no_filter_four_values": dict(values=[a, b, k, u], filt=-1),
    }


LAYOUTS = [([1], 0.0, 0), ([255, 257, 1024, 1], 0.1, 3), ([1024] * 20 + [576], 0.05, 0), ([300_001], 0.0, 1), ([70_000, 5], 0.9, 2)]


def _check_aggregates(got, exp, what):
    for v, (g, x) in enumerate(zip(got, exp)):
        w = f"{what} value {v}"
        assert g.count == x.count and g.is_some == x.is_some and g.dtype == x.dtype, w
        if not x.is_some:
            continue
        if g.dtype in (A.F32, A.F64):   # the sum's order differs from the oracle's row order: 1e-6 relative, as for every fused aggregate
            assert abs(g.sum - x.sum) <= 1e-6 * max(abs(x.sum), 1e-9) + 1e-9, f"sum {w}: {g.sum} vs {x.sum}"
            assert g.min == x.min and g.max == x.max, w
        else:
            assert (g.sum, g.min, g.max) == (x.sum, x.min, x.max), w


@pytest.mark.parametrize("layout", LAYOUTS)
def test_lean_kernel_against_the_oracle(interp, ora, layout):
    api, lib = interp
    lens, nf, off = layout
    rng = np.random.default_rng(606)
    cols = [make_chunks(rng, A.F64, lens, nf, off, "unit"), make_chunks(rng, A.F64, lens, nf, off, "unit"),
            make_chunks(rng, A.I64, lens, nf, off, "plain"), make_chunks(rng, A.U64, lens, nf, off, "plain")]
    e = A.Expr()
    for name, p in _programs(e).items():
        exp = ora.pipeline(e, cols, p["values"], p["filt"])
        forms = []
        for mode in (1, 2):   # two tiles per trip of the step loop where the launcher picks that; one tile forced
            lib.set_option("interp_lean", mode)
            got = api.pipeline(e, cols, p["values"], p["filt"])
            lib.set_option("interp_lean", 1)
            assert lib.last_kernel() == "eval_kernel<AGG, lean>", f"{name} ran on {lib.last_kernel()}"
            _check_aggregates(got, exp, f"{name} lens={lens[:3]} nf={nf} interp_lean={mode}")
            forms.append(got)
        # a lane's rows meet its running aggregates in the same order either way: the two forms agree to the bit
        for g, x in zip(*forms):
            assert (g.count, g.is_some) == (x.count, x.is_some), name
            if g.is_some:
                assert (g.min, g.max) == (x.min, x.max) and (g.sum == x.sum or (g.sum != g.sum and x.sum != x.sum)), name


def test_lean_kernel_and_general_kernel_give_the_same_bits(interp):
    """Random trees over f64 / i64 / u64 columns with NaN, +-0, infinities, NULLs, ragged batches at odd offsets
    (tools/lean_ab.py): every rdf_agg_result field equal as a bit pattern (a NaN equals any NaN: which one an operation hands on
    depends on the compiled operand order), errors (divide by zero) equal too."""
    api, lib = interp
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lean_ab
    r = lean_ab.run(api, 150, 2026)
    assert r["mismatches"] == 0, r["examples"]
    assert r["took_the_lean_kernel"] >= r["programs"] // 2, r


def test_programs_outside_the_lean_class_take_the_general_kernel(interp, ora):
    api, lib = interp
    rng = np.random.default_rng(5)
    lens = [1024, 300]
    f = [make_chunks(rng, A.F64, lens, 0.0, 0, "unit") for _ in range(5)]
    i32 = make_chunks(rng, A.I32, lens, 0.0, 0, "plain")
    e = A.Expr()
    cases = {
        "a libm function": ([f[0]], [e.op("sin", e.col(0))], -1),
        "a 4-byte column": ([f[0], i32], [e.col(0)], e.op("gt", e.col(1), e.scalar(0, A.I32))),
        "five columns": (f, [e.op("add", e.col(0), e.col(4))], e.op("gt", e.col(1), e.op("add", e.col(2), e.col(3)))),
        "a narrowing cast": ([f[0]], [e.cast(e.op("multiply", e.col(0), e.scalar(100.0)), A.I64)], -1),
    }
    for name, (cols, values, filt) in cases.items():
        exp = ora.pipeline(e, cols, values, filt)
        got = api.pipeline(e, cols, values, filt)
        assert lib.last_kernel() == "eval_kernel<AGG>", f"{name} ran on {lib.last_kernel()}"
        assert got[0].count == exp[0].count, name
    lib.set_option("interp_lean", 0)
    api.pipeline(e, [f[0]], [e.col(0)], e.op("gt", e.col(0), e.scalar(0.5)))
    assert lib.last_kernel() == "eval_kernel<AGG>", lib.last_kernel()


def test_divide_by_zero_at_a_live_slot_is_an_error_on_the_lean_kernel_too(interp):
    """arrow's math_divide (Evaluate::calculate's divide, src/evaluation.rs:120-128): a zero divisor where both sides are valid
    fails the call; at a NULL slot it does not."""
    api, lib = interp
    x = np.array([1.0, 2.0, 3.0, 4.0] * 300)
    d = np.array([1.0, 0.0, 2.0, 4.0] * 300)
    e = A.Expr()
    q = e.op("divide", e.col(0), e.col(1))
    with pytest.raises(Exception) as ei:
        api.pipeline(e, [[A.HostArray.from_numpy(x)], [A.HostArray.from_numpy(d)]], [q], -1)
    assert lib.last_kernel() == "eval_kernel<AGG, lean>", lib.last_kernel()
    msg_lean = str(ei.value)
    lib.set_option("interp_lean", 0)
    with pytest.raises(Exception) as ei:
        api.pipeline(e, [[A.HostArray.from_numpy(x)], [A.HostArray.from_numpy(d)]], [q], -1)
    assert str(ei.value) == msg_lean
    lib.set_option("interp_lean", 1)
    valid = d != 0.0
    got = api.pipeline(e, [[A.HostArray.from_numpy(x)], [A.HostArray.from_numpy(d, valid=valid)]], [q], -1)[0]
    assert lib.last_kernel() == "eval_kernel<AGG, lean>", lib.last_kernel()
    assert got.count == int(valid.sum()) and got.sum == pytest.approx(float((x[valid] / d[valid]).sum()), rel=1e-12)


# ------------------------------------------------------------------------------------------------ SINK_STORE
# Evaluate::calculate's own shape (src/evaluation.rs:97-323): a computed column per value expression, NULL where an input is NULL;
# BooleanFilter::eval_to_array (src/expression.rs:766-861): a predicate's Boolean array.

def _store_programs(e):
    a, b, k, u = e.col(0), e.col(1), e.col(2), e.col(3)
    return {
        "fma": ([e.op("add", e.op("multiply", a, b), e.scalar(0.25))], [A.F64]),
        "two_columns": ([e.op("subtract", e.scalar(1.0), a), e.op("divide", a, e.op("add", e.op("multiply", b, b), e.scalar(1.0)))], [A.F64, A.F64]),
        "integers": ([e.op("add", e.op("multiply", k, k), e.scalar(3, A.I64)), e.op("multiply", u, e.scalar(5, A.U64))], [A.I64, A.U64]),
        "cast_and_add": ([e.op("add", e.cast(k, A.F64), a)], [A.F64]),
        "predicate": ([e.op("and", e.op("gt", a, e.scalar(0.5)), e.op("not", e.op("lt", b, a))), e.op("ne", k, e.scalar(0, A.I64))], [A.BOOL, A.BOOL]),
        "predicate_and_value": ([e.op("ge", u, k), e.op("multiply", a, e.scalar(2.0))], [A.BOOL, A.F64]),
        "bushy_with_temporaries": ([e.op("multiply", e.op("add", a, b), e.op("subtract", a, e.op("multiply", b, e.scalar(3.0))))], [A.F64]),
    }


def _assert_same_chunks(got, exp, what):
    """two results of the SAME computation: lengths, NULL counts, validity bits and the values at valid slots as bit patterns (NaN = NaN)"""
    assert len(got) == len(exp), what
    for i, (g, x) in enumerate(zip(got, exp)):
        w = f"{what} chunk {i}"
        assert (g.dtype, g.length, g.null_count) == (x.dtype, x.length, x.null_count), f"{w}: {(g.dtype, g.length, g.null_count)} vs {(x.dtype, x.length, x.null_count)}"
        gm, xm = g.valid_mask(), x.valid_mask()
        assert np.array_equal(gm, xm), f"{w}: validity bitmaps differ"
        gv, xv = g.to_numpy()[xm], x.to_numpy()[xm]
        if g.dtype == A.F64:
            nan = np.isnan(xv)
            assert np.array_equal(np.isnan(gv), nan), w
            assert np.array_equal(gv[~nan].view(np.uint64), xv[~nan].view(np.uint64)), w
        else:
            assert np.array_equal(gv, xv), w


@pytest.mark.parametrize("layout", LAYOUTS + [([4096 + 17], 0.3, 1)])
@pytest.mark.parametrize("with_validity", [True, False])
def test_lean_store_against_the_oracle_and_the_general_kernel(interp, ora, layout, with_validity):
    from util import assert_chunks_match
    api, lib = interp
    lens, nf, off = layout
    if not with_validity and nf > 0:
        pytest.skip("NULLs in the inputs need an output bitmap")
    rng = np.random.default_rng(707)
    cols = [make_chunks(rng, A.F64, lens, nf, off, "unit"), make_chunks(rng, A.F64, lens, nf, off, "unit"),
            make_chunks(rng, A.I64, lens, nf, off, "plain"), make_chunks(rng, A.U64, lens, nf, off, "plain")]
    e = A.Expr()
    for name, (vals, dts) in _store_programs(e).items():
        mk = lambda: [[A.HostArray.empty_out(dt, n, with_validity) for n in lens] for dt in dts]
        exp = ora.pipeline(e, cols, vals, -1, A.SINK_STORE, mk())
        lib.set_option("interp_lean", 1)
        got = api.pipeline(e, cols, vals, -1, A.SINK_STORE, mk())
        assert lib.last_kernel() == "eval_kernel<STORE, lean>", f"{name} ran on {lib.last_kernel()}"
        lib.set_option("interp_lean", 0)
        gen = api.pipeline(e, cols, vals, -1, A.SINK_STORE, mk())
        assert lib.last_kernel() == "eval_kernel<STORE>", lib.last_kernel()
        lib.set_option("interp_lean", 1)
        for v in range(len(vals)):
            assert_chunks_match(got[v], exp[v], exact=True, what=f"{name} value {v} lens={lens[:3]} (oracle)")
            _assert_same_chunks(got[v], gen[v], f"{name} value {v} lens={lens[:3]} (general kernel)")


def test_lean_store_random_programs_same_as_general_kernel(interp):
    """random trees (tools/lean_ab.py's generator) stored as columns: the two kernels' outputs chunk by chunk"""
    api, lib = interp
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lean_ab
    rng = np.random.default_rng(808)
    ran = took = 0
    for pi in range(60):
        ncols = int(rng.integers(1, 5))
        dts = [int(rng.choice([A.F64, A.F64, A.I64, A.U64])) for _ in range(ncols)]
        lens = [int(rng.choice([1, 3, 255, 256, 257, 1024, 1025, 4096 + 17])) for _ in range(int(rng.choice([1, 2, 4])))]
        nf = float(rng.choice([0.0, 0.1, 0.9]))
        off = int(rng.integers(0, 4))
        cols = [make_chunks(rng, dt, lens, nf, off, "unit" if dt == A.F64 else "plain") for dt in dts]
        e = A.Expr()
        try:
            vals, odts = [], []
            for _ in range(int(rng.integers(1, 4))):
                icols = [c for c in range(ncols) if dts[c] != A.F64]
                want = str(rng.choice(["f", "b"] + (["i"] if icols else [])))
                node = lean_ab.random_tree(rng, e, dts, int(rng.integers(0, 3)), want)
                vals.append(node)
                odts.append(want)
        except Exception:
            continue
        outs = []
        for mode in (0, 1):
            lib.set_option("interp_lean", mode)
            # the output type of an integer tree is its columns' type: ask the library by trying the candidates
            res = None
            for cand in ([A.I64, A.U64] if "i" in odts else [None]):
                mk = [[A.HostArray.empty_out(A.F64 if w == "f" else A.BOOL if w == "b" else cand, n, True) for n in lens] for w in odts]
                try:
                    res = ("ok", api.pipeline(e, cols, vals, -1, A.SINK_STORE, mk), lib.last_kernel())
                    break
                except Exception as ex:
                    res = ("err", str(ex)[:60], lib.last_kernel())
            outs.append(res)
        lib.set_option("interp_lean", 1)
        assert outs[0][0] == outs[1][0], (pi, outs[0][:2], outs[1][:2])
        if outs[0][0] == "err":
            assert outs[0][1] == outs[1][1], (pi, outs[0][1], outs[1][1])
            continue
        ran += 1
        took += "lean" in outs[1][2]
        for v in range(len(vals)):
            _assert_same_chunks(outs[1][1][v], outs[0][1][v], f"program {pi} value {v} dtypes {dts} lens {lens}")
    assert ran >= 20 and took >= ran // 2, (ran, took)
