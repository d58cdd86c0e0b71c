"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Mirrors the reference's own test style (in-file unit tests per kernel) with two-sided
checks: integers/bitmaps bit-exact, floats within util.RTOL = 1e-6 relative.
"""
import os

import numpy as np
import pytest

from rust_dataframe_amd import _abi as A

from util import rand_values, assert_arrays_match, assert_chunks_match, assert_scalar_close, make_chunks

pytestmark = pytest.mark.gpu

NUMERIC = [A.I8, A.I16, A.I32, A.I64, A.U8, A.U16, A.U32, A.U64, A.F32, A.F64]
LAYOUTS = [  # (chunk lengths, null fraction, offset)
    ([5], 0.0, 0),
    ([1000], 0.0, 0),
    ([1024, 1024, 576], 0.0, 0),          # reader batches (src/dataframe.rs:352)
    ([4097], 0.1, 0),
    ([700, 0, 3000], 0.1, 13),            # empty chunk, sliced arrays (non-zero offset)
    ([2500], -1.0, 5),                    # all null
]


@pytest.mark.parametrize("dtype", NUMERIC)
@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide"])
def test_binary_arithmetic(gpu, ora, dtype, op):
    rng = np.random.default_rng(100 + dtype)
    for lens, nf, off in LAYOUTS:
        kind = "extreme" if dtype <= A.U64 and op != "divide" else "plain"
        a = make_chunks(rng, dtype, lens, nf, off, kind)
        b = make_chunks(rng, dtype, lens, nf, off, kind, nonzero=(op == "divide"))
        exp = ora.binary(op, a, b)
        got = gpu.binary(op, a, b)
        # add/sub/mul/div are single IEEE operations: bit-exact even for floats
        assert_chunks_match(got, exp, exact=True, what=f"{op} dtype={dtype} lens={lens}")


@pytest.mark.parametrize("dtype", [A.F32, A.F64])
@pytest.mark.parametrize("op", ["atan2", "hypot", "log"])
def test_binary_float_math(gpu, ora, dtype, op):
    rng = np.random.default_rng(7)
    for lens, nf, off in LAYOUTS[:5]:
        a = make_chunks(rng, dtype, lens, nf, off, "pos")
        b = make_chunks(rng, dtype, lens, nf, off, "pos")
        assert_chunks_match(gpu.binary(op, a, b), ora.binary(op, a, b), exact=False, what=f"{op} {dtype}")


UNARY_KIND = {"acos": "unit", "asin": "unit", "log10": "pos", "log2": "pos", "sqrt": "pos", "cosh": "unit", "sinh": "unit",
              "exp": "unit", "expm1": "unit"}
EXACT_UNARY = {"abs", "ceil", "floor", "round", "sqrt"}  # IEEE-exact operations


@pytest.mark.parametrize("dtype", [A.F32, A.F64])
@pytest.mark.parametrize("op", A.UNARY_OPS)
def test_unary_float(gpu, ora, dtype, op):
    rng = np.random.default_rng(11)
    for lens, nf, off in LAYOUTS:
        a = make_chunks(rng, dtype, lens, nf, off, UNARY_KIND.get(op, "plain"))
        assert_chunks_match(gpu.unary(op, a), ora.unary(op, a), exact=op in EXACT_UNARY, what=f"{op} {dtype} {lens}")


def test_unary_large_arguments_and_specials(gpu, ora):
    """sin/cos/tan far outside [-pi, pi] (Payne-Hanek territory), NaN/inf/±0 inputs."""
    v = np.array([1e6, -1e9, 1e15, 1e22, 3.0e300, -7.5e-310, 0.0, -0.0, np.nan, np.inf, -np.inf, 0.5, 1e-8, 710.0, -745.0])
    a = [A.HostArray.from_numpy(v)]
    for op in ["sin", "cos", "tan", "cot", "sec", "csc", "exp", "tanh", "atan", "cbrt", "floor", "round", "abs"]:
        g, e = gpu.unary(op, a)[0], ora.unary(op, a)[0]
        with np.errstate(all="ignore"):
            np.testing.assert_allclose(g.to_numpy(), e.to_numpy(), rtol=1e-6, atol=0, equal_nan=True, err_msg=op)
    # round is half away from zero (num::Float::round)
    r = gpu.unary("round", [A.HostArray.from_numpy(np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5]))])[0].to_numpy()
    assert r.tolist() == [1.0, 2.0, 3.0, -1.0, -2.0, -3.0]


def test_unary_f32_trig_reduction_range(gpu, ora):
    """f32 sin / cos / sec / csc run the library's own routine (f64 argument reduction + an f32 polynomial) below 1e9 and the
    device libm above: arguments next to multiples of pi/2 (where the relative accuracy of the reduced argument decides), large
    ones on both sides of the hand-over, tiny ones, -0.0 (sign kept), inf / NaN."""
    near = []
    for k in [1, 2, 3, 7, 100, 1001, 31416, 1_000_003, 12_345_678, 200_000_001]:
        for h in (0.0, 0.5):
            c = np.float32((k + h) * np.pi)
            near += [c, np.nextafter(c, np.float32(np.inf)), np.nextafter(c, np.float32(-np.inf)), -c]
    v = np.array(near + [1e6, -3.3e7, 9.9e8, 1.0e9, 1.1e9, -2.5e9, 3e20, 1e38, 1e-5, -1e-5, 1e-20, -1e-30, 1e-45, 0.0, -0.0, 0.5, -1.5707964,
                         np.nan, np.inf, -np.inf], dtype=np.float32)
    a = [A.HostArray.from_numpy(v)]
    for op in ["sin", "cos", "sec", "csc", "tan", "cot"]:
        g, e = gpu.unary(op, a)[0], ora.unary(op, a)[0]
        with np.errstate(all="ignore"):
            np.testing.assert_allclose(g.to_numpy(), e.to_numpy(), rtol=1e-6, atol=0, equal_nan=True, err_msg=op)
    s = gpu.unary("sin", [A.HostArray.from_numpy(np.array([-0.0, 0.0], dtype=np.float32))])[0].to_numpy()
    assert np.signbit(s[0]) and not np.signbit(s[1])


@pytest.mark.parametrize("dtype", [A.I8, A.I16, A.I32, A.I64])
def test_abs_signed_int(gpu, ora, dtype):
    rng = np.random.default_rng(3)
    for lens, nf, off in LAYOUTS:
        a = make_chunks(rng, dtype, lens, nf, off, "extreme")
        assert_chunks_match(gpu.unary("abs", a), ora.unary("abs", a), what=f"abs {dtype}")


CAST_TYPES = NUMERIC + [A.BOOL]


@pytest.mark.parametrize("src", CAST_TYPES)
def test_cast_matrix(gpu, ora, src):
    rng = np.random.default_rng(50 + src)
    for dst in CAST_TYPES:
        for lens, nf, off in [([1000], 0.0, 0), ([700, 0, 3000], 0.1, 13)]:
            if src == A.BOOL:
                a = [A.HostArray.from_numpy(rng.integers(0, 2, n).astype(bool), valid=(rng.uniform(size=n) >= nf) if nf else None,
                                            offset=off, dtype=A.BOOL, rng=rng) for n in lens]
            else:
                a = make_chunks(rng, src, lens, nf, off, "special" if src in (A.F32, A.F64) else "extreme")
                if src in (A.F32, A.F64):  # add magnitudes that saturate every integer width
                    for ch in a:
                        if ch.length >= 16:
                            ch.values[ch.offset + 8:ch.offset + 16] = [300.7, -300.7, 7e4, -7e4, 5e9, -5e9, 3e19, -3e19]
            assert_chunks_match(gpu.cast(a, dst), ora.cast(a, dst), exact=True, what=f"cast {src}->{dst}")


def test_cast_nulls_where_unrepresentable(gpu, ora, request):
    """The edge values of every lossy direction through rdf_cast, through a fused aggregate of a cast, and through a stored
    fused expression holding a cast (specialised kernels and the evaluator): NULL exactly where the oracle says so."""
    nan, inf = float("nan"), float("inf")
    cases = [(A.I64, A.U8, [-1, 0, 255, 256, 300]), (A.I32, A.I8, [-129, -128, 127, 128]), (A.U32, A.I32, [2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1]),
             (A.U64, A.I64, [2 ** 63 - 1, 2 ** 63, 2 ** 64 - 1]), (A.I8, A.U64, [-1, 5]), (A.I64, A.I32, [-2 ** 63, -7, 2 ** 31]),
             (A.F64, A.I64, [nan, inf, -inf, 1e20, -1e20, 2.0 ** 63, -(2.0 ** 63), 9.2e18, -0.9, 3.99]), (A.F64, A.U8, [-0.5, -1.0, 255.9, 256.0, nan]),
             (A.F64, A.U32, [4294967295.9, 4294967296.0, -0.99]), (A.F64, A.I32, [2147483647.9, 2147483648.0, -2147483648.9, -2147483649.0]),
             (A.F32, A.I32, [2.0 ** 31, -(2.0 ** 31), 1.5e9]), (A.F32, A.U64, [1.8446742e19, 2.0 ** 64, -1.0]), (A.F64, A.U64, [1.8e19, 2.0 ** 64, -1.0, nan]),
             (A.F64, A.I16, [32767.5, 32768.0, -32768.5, -32769.0]), (A.U16, A.I8, [127, 128]), (A.I16, A.U16, [-1, 65535 // 2])]
    rng = np.random.default_rng(3)
    for src, dst, vals in cases:
        base = np.array(vals, dtype=A.NP_OF[src])
        # long enough for full vector tiles, with the edge values scattered among ordinary ones
        body = rand_values(rng, src, 5000, "special" if src in (A.F32, A.F64) else "extreme")
        body[rng.integers(0, 5000, 10 * len(base))] = np.tile(base, 10)
        for arr in (A.HostArray.from_numpy(base), A.HostArray.from_numpy(body), A.HostArray.from_numpy(body, valid=rng.uniform(size=5000) > 0.2, offset=3, rng=rng)):
            assert_chunks_match(gpu.cast([arr], dst), ora.cast([arr], dst), exact=True, what=f"cast {src}->{dst}")
            e = A.Expr()
            c = e.cast(e.col(0), dst)
            g, o = gpu.pipeline(e, [[arr]], [c])[0], ora.pipeline(e, [[arr]], [c])[0]
            assert (g.count, g.is_some) == (o.count, o.is_some), f"count of cast {src}->{dst}"
            if o.is_some:
                assert (g.min, g.max) == (o.min, o.max) and (g.sum == o.sum or abs(g.sum - o.sum) <= 1e-9 * abs(o.sum)), f"aggregates of cast {src}->{dst}"


@pytest.mark.parametrize("dtype", [A.I32, A.I64])
def test_hour(gpu, ora, dtype):
    """ScalarFunctions::hour over the storage of Time32 / Time64 / Date / Timestamp chunks, every time unit; and the
    hour opcodes inside a fused pipeline (count of the rows after noon, sum of the hours)."""
    rng = np.random.default_rng(900 + dtype)
    for unit in (A.TIME_SECOND, A.TIME_MILLISECOND, A.TIME_MICROSECOND, A.TIME_NANOSECOND, A.TIME_DAY):
        for lens, nf, off in LAYOUTS:
            a = make_chunks(rng, dtype, lens, nf, off, "extreme")
            assert_chunks_match(gpu.hour(a, unit), ora.hour(a, unit), exact=True, what=f"hour unit={unit} dtype={dtype} lens={lens}")
    for name in ("hour_s", "hour_ms", "hour_us", "hour_ns"):
        a = make_chunks(rng, dtype, [1024, 1024, 576], 0.1, 3, "extreme")
        e = A.Expr()
        h = e.op(name, e.col(0))
        pred = e.op("gt", h, e.scalar(12))
        for api_res in zip(gpu.pipeline(e, [a], [h], pred), ora.pipeline(e, [a], [h], pred)):
            g, o = api_res
            assert (g.sum, g.min, g.max, g.count, g.is_some) == (o.sum, o.min, o.max, o.count, o.is_some), name
    with pytest.raises(A.RdfError) as ei:
        gpu.hour(make_chunks(rng, A.F64, [10], 0.0, 0), A.TIME_SECOND)
    assert ei.value.status == A.RDF_COMPUTE_ERROR
    with pytest.raises(A.RdfError) as ei:
        gpu.hour(make_chunks(rng, A.I64, [10], 0.0, 0), 7)
    assert ei.value.status == A.RDF_INVALID_ARGUMENT


@pytest.mark.parametrize("dtype", NUMERIC)
def test_aggregates(gpu, ora, dtype):
    rng = np.random.default_rng(200 + dtype)
    for lens, nf, off in LAYOUTS + [([50_000, 70_001], 0.05, 3)]:
        a = make_chunks(rng, dtype, lens, nf, off, "plain" if dtype in (A.F32, A.F64) else "extreme")
        what = f"dtype={dtype} lens={lens} nf={nf}"
        if dtype == A.F32:
            # the reference sums f32 in f32 (sequential left fold, aggregate.rs:82-93); the device folds in f64
            # and rounds once.  Both are within the f32 fold's own error bound n * 2^-24 * sum|x| of the exact sum.
            mag = sum(float(np.abs(ch.to_numpy()[ch.valid_mask()].astype(np.float64)).sum()) for ch in a)
            n = sum(ch.length for ch in a)
            assert abs(gpu.sum(a) - ora.sum(a)) <= max(1e-6 * abs(ora.sum(a)), n * 2.0 ** -24 * mag), "sum " + what
        else:
            assert_scalar_close(gpu.sum(a), ora.sum(a), dtype, "sum " + what)
        assert_scalar_close(gpu.min(a), ora.min(a), dtype, "min " + what)
        assert_scalar_close(gpu.max(a), ora.max(a), dtype, "max " + what)
        assert gpu.count(a) == ora.count(a), "count " + what
        g, e = gpu.avg(a), ora.avg(a)
        assert (g is None) == (e is None), what
        if e is not None:
            assert abs(g - e) <= 1e-6 * max(abs(e), 1e-12) + 1e-9, f"avg {what}: {g} vs {e}"
        # unknown null_count: count must read the bitmap
        for ch in a:
            ch._unknown_nc = True
        assert gpu.count(a) == ora.count(a), "count(bitmap) " + what


def test_aggregates_nan_and_reference_known_answers(gpu, ora):
    # NaN never wins min/max unless everything is NaN (documented divergence)
    a = [A.HostArray.from_numpy(np.array([np.nan, 3.0, -2.0, np.nan]))]
    assert gpu.min(a) == -2.0 and gpu.max(a) == 3.0
    assert np.isnan(gpu.min([A.HostArray.from_numpy(np.array([np.nan, np.nan]))]))
    # src/functions/aggregate.rs:123-146
    assert gpu.count([A.HostArray.from_numpy(np.array([5, 6, 7, 8, 9], dtype=np.int32))]) == 5
    a = A.HostArray.from_numpy(np.arange(0, 5, dtype=np.int32))
    b = A.HostArray.from_numpy(np.arange(5, 10, dtype=np.int32))
    assert gpu.avg([a, b]) == 4.5
    d = A.HostArray.from_numpy(np.array([0, 0, 1, 0, 2, 3, 4], dtype=np.int32), valid=[1, 0, 1, 0, 1, 1, 1])
    assert gpu.avg([d, b]) == 4.5
    # a leading empty / all-NULL chunk: the reference's merge divides 0 by 0 there (aggregate.rs:56-59) — not copied
    empty = A.HostArray.from_numpy(np.zeros(0, dtype=np.int32))
    nulls = A.HostArray.from_numpy(np.array([7, 8, 9], dtype=np.int32), valid=[0, 0, 0])
    assert gpu.avg([empty, b]) == 7.0 == ora.avg([empty, b]) and gpu.avg([nulls, b, empty]) == 7.0 and gpu.avg([nulls, empty]) is None


def _pred_cases(e, ncols):
    c0, c1 = e.col(0), e.col(1 % ncols)
    gt = e.op("gt", c0, e.scalar(0.5))
    le = e.op("le", c0, c1)
    return {
        "gt_scalar": gt,
        "scalar_lt_col": e.op("lt", e.scalar(10, A.I64), c0),
        "le_cols": le,
        "eq": e.op("eq", c0, c1),
        "ne": e.op("ne", c0, e.scalar(3.0)),
        "ge": e.op("ge", c0, e.scalar(-7, A.I32)),
        "not": e.op("not", gt),
        "and": e.op("and", gt, le),
        "or": e.op("or", e.op("not", le), e.op("lt", c0, e.scalar(-50.0))),
        "not_numeric": e.op("not", c1),
        "null_scalar": e.op("or", gt, e.scalar(None)),
        "deep": e.op("and", e.op("or", e.op("gt", e.op("add", c0, c0), e.op("multiply", c1, c1)), gt),
                     e.op("not", e.op("and", le, e.op("ne", c1, e.scalar(0.0))))),
    }


@pytest.mark.parametrize("dtypes", [(A.F64, A.F64), (A.I64, A.I64), (A.F32, A.F32), (A.I32, A.I32), (A.U8, A.U8)])
def test_predicate_eval_to_array(gpu, ora, dtypes):
    rng = np.random.default_rng(31)
    for lens, nf, off in LAYOUTS:
        cols = [make_chunks(rng, dt, lens, nf, off, "plain") for dt in dtypes]
        e = A.Expr()
        for name, root in _pred_cases(e, len(cols)).items():
            exp = ora.predicate(e, root, cols)
            got = gpu.predicate(e, root, cols)
            assert_chunks_match(got, exp, what=f"predicate {name} {dtypes} {lens}")
            for g in got:  # the value bit of a null slot is 0
                assert not np.any(g.to_numpy() & ~g.valid_mask())


def test_predicate_mixed_types_compare_in_f64(gpu, ora):
    """Comparisons cast both sides to Float64 (src/expression.rs:844-845), so i64 beyond 2^53 compares lossy (B6)."""
    big = np.array([2 ** 53, 2 ** 53 + 1, -(2 ** 53) - 1, 5], dtype=np.int64)
    x = [A.HostArray.from_numpy(big)]
    y = [A.HostArray.from_numpy(np.array([2.0 ** 53, 2.0 ** 53, -(2.0 ** 53), 5.5]))]
    e = A.Expr()
    for op in ["eq", "gt", "le"]:
        root = e.op(op, e.col(0), e.col(1))
        assert_chunks_match(gpu.predicate(e, root, [x, y]), ora.predicate(e, root, [x, y]), what=op)
    got = gpu.predicate(e, e.op("eq", e.col(0), e.col(1)), [x, y])[0].to_numpy()
    assert got.tolist() == [True, True, True, False]


@pytest.mark.parametrize("dtype", NUMERIC)
def test_filter(gpu, ora, dtype):
    rng = np.random.default_rng(400 + dtype)
    for lens, nf, off in LAYOUTS + [([2048], 0.0, 0), ([2049, 4095, 1], 0.2, 7), ([100_000], 0.1, 1)]:
        for sel in (0.5, 0.02, 1.0, 0.0):
            col = make_chunks(rng, dtype, lens, nf, off, "extreme" if dtype <= A.U64 else "special")
            mask = [A.HostArray.from_numpy(rng.uniform(size=n) < sel, valid=(rng.uniform(size=n) > 0.1) if nf else None,
                                           offset=(off * 3) % 11, dtype=A.BOOL, rng=rng) for n in lens]
            assert gpu.filter_count(mask) == ora.filter_count(mask)
            assert_chunks_match(gpu.filter(col, mask), ora.filter(col, mask), exact=True, what=f"filter {dtype} {lens} sel={sel}")


def test_filter_columns_one_pass(gpu, ora):
    """DataFrame::filter (src/dataframe.rs:178-189): every column compacted by one mask."""
    rng = np.random.default_rng(9)
    lens = [1024, 1024, 1024, 333]
    cols = [make_chunks(rng, dt, lens, nf, off) for dt, nf, off in [(A.F64, 0.1, 0), (A.I64, 0.0, 5), (A.I32, 0.3, 2), (A.U8, 0.0, 0), (A.F32, 0.5, 9), (A.I16, 0.1, 1)]]
    e = A.Expr()
    root = e.op("gt", e.col(0), e.scalar(0.0))
    mask = ora.predicate(e, root, cols)
    assert_chunks_match(gpu.predicate(e, root, cols), mask, what="mask")
    got, exp = gpu.filter_columns(cols, mask), ora.filter_columns(cols, mask)
    for k in range(len(cols)):
        assert_chunks_match(got[k], exp[k], exact=True, what=f"column {k}")


@pytest.mark.parametrize("lens", [[9000], [5000, 4097], [1024] * 9 + [100]])
def test_filter_wide_frames_every_kernel_variant(gpu, ora, lens):
    """11 columns (two launches of <= 8 columns) of mixed and of equal element sizes through the three compaction paths:
    one long chunk (descriptors in the kernel arguments), long chunks (4096-row tiles), reader batches (1024-row tiles)."""
    rng = np.random.default_rng(4000 + len(lens))
    for dts in ([A.F64, A.I64, A.I32, A.U8, A.F32, A.I16, A.U64, A.F64, A.I8, A.U16, A.I64], [A.F64] * 6 + [A.I64] * 5):
        cols = [make_chunks(rng, dt, lens, 0.2 if k % 3 == 0 else 0.0, k % 4) for k, dt in enumerate(dts)]
        e = A.Expr()
        root = e.op("gt", e.col(0), e.scalar(0.1))
        mask = ora.predicate(e, root, cols)
        got, exp = gpu.filter_columns(cols, mask), ora.filter_columns(cols, mask)
        for k in range(len(cols)):
            assert_chunks_match(got[k], exp[k], exact=True, what=f"lens={lens} column {k} dtype={dts[k]}")
        from rust_dataframe_amd import lib
        lib.set_option("filter_one", 0)   # the table-driven kernels on the same input
        try:
            got = gpu.filter_columns(cols, mask)
        finally:
            lib.set_option("filter_one", 1)
        for k in range(len(cols)):
            assert_chunks_match(got[k], exp[k], exact=True, what=f"table-driven, lens={lens} column {k}")


@pytest.mark.parametrize("dtype", [A.F64, A.I64, A.F32, A.I32, A.I16, A.U8])
def test_filter_wave_granular_and_first_generation(gpu, ora, dtype):
    """Both compaction generations (rdf_set_option("filter_gen", 2 | 1)) on the shapes that separate their code paths:
    dense tiles (vector loads + 16-byte stores), sparse tiles (kept rows only), tiles that start at odd output positions,
    chunk slices whose values are not 16-byte aligned, the reader's 1024-row batches (two wave tiles per chunk), chunks
    shorter than a tile, empty chunks, all-kept / none-kept masks, NULLs in mask and column."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(700 + dtype)
    try:
        for lens, nf, off in [([5000], 0.0, 0), ([1024] * 7 + [333], 0.15, 0), ([4096, 0, 777, 2048], 0.1, 1), ([100, 300, 50, 1, 0, 600], 0.2, 5), ([20000], 0.05, 3)]:
            col = make_chunks(rng, dtype, lens, nf, off, "extreme" if dtype <= A.U64 else "special")
            for sel in (0.9, 0.5, 0.05, 1.0, 0.0):
                mask = [A.HostArray.from_numpy(rng.uniform(size=n) < sel, valid=(rng.uniform(size=n) > 0.1) if nf else None,
                                               offset=(off * 3) % 11, dtype=A.BOOL, rng=rng) for n in lens]
                exp = ora.filter(col, mask)
                for gen in (2, 3, 1):   # default (LDS-DMA tiles where eligible), register-staged wave tiles, first generation
                    lib.set_option("filter_gen", gen)
                    assert gpu.filter_count(mask) == ora.filter_count(mask), f"count gen={gen}"
                    assert_chunks_match(gpu.filter(col, mask), exp, exact=True, what=f"filter gen={gen} {dtype} {lens} sel={sel}")
    finally:
        lib.set_option("filter_gen", 2)


@pytest.mark.parametrize("dtype", NUMERIC)
@pytest.mark.parametrize("idx_dtype", [A.U32, A.U64])
def test_take(gpu, ora, dtype, idx_dtype):
    rng = np.random.default_rng(600 + dtype)
    for lens, nf, off in LAYOUTS[:5] + [([1024] * 37 + [99], 0.1, 3)]:
        col = make_chunks(rng, dtype, lens, nf, off, "extreme" if dtype <= A.U64 else "special")
        total = sum(lens)
        for n_idx, idx_nf, idx_off in [(1, 0.0, 0), (777, 0.0, 0), (5000, 0.2, 9)]:
            idx = A.HostArray.from_numpy(rng.integers(0, total, n_idx).astype(A.NP_OF[idx_dtype]),
                                         valid=(rng.uniform(size=n_idx) >= idx_nf) if idx_nf else None, offset=idx_off, rng=rng)
            assert_arrays_match(gpu.take(col, idx), ora.take(col, idx), exact=True, what=f"take {dtype} {lens}")


def test_take_reference_sort_case(gpu):
    """test_sort (src/dataframe.rs:963-1003): lexsort indices [5,3,4,0,1,2] applied by Column::take."""
    a = A.HostArray.from_numpy(np.array([1, 1, 0, 3, 3, 4], dtype=np.int32), valid=[1, 1, 0, 1, 1, 1])
    b = A.HostArray.from_numpy(np.array([9, 5, 6, 7, 4, 8], dtype=np.uint8))
    idx = A.HostArray.from_numpy(np.array([5, 4, 3, 1, 0, 2], dtype=np.uint32))
    assert gpu.take([a], idx).to_pylist() == [4, 3, 3, 1, 1, None]
    assert gpu.take([b], idx).to_pylist() == [8, 4, 7, 5, 9, 6]


def _pipeline_programs(e):
    a, b, c, k = e.col(0), e.col(1), e.col(2), e.col(3)
    fma = e.op("add", e.op("multiply", a, b), c)
    return {
        "filter_sum": dict(values=[a], filt=e.op("gt", a, e.scalar(0.5))),
        "filter_other": dict(values=[b, k], filt=e.op("le", a, e.scalar(0.25))),
        "c3_fused": dict(values=[fma, k], filt=-1),
        "c1_sin_add_scalar": dict(values=[e.op("sin", e.op("add", a, e.scalar(1.0)))], filt=-1),
        "bushy": dict(values=[e.op("multiply", e.op("add", a, b), e.op("subtract", c, e.op("divide", a, e.op("add", e.op("abs", b), e.scalar(1.0)))))], filt=e.op("ne", k, e.scalar(0, A.I64))),
        "casts": dict(values=[e.op("add", e.cast(k, A.F64), a), e.cast(e.op("multiply", a, e.scalar(100.0)), A.I32)], filt=e.op("and", e.op("gt", k, e.scalar(-100, A.I64)), e.op("lt", b, c))),
        "four_values": dict(values=[a, b, c, k], filt=e.op("or", e.op("lt", a, e.scalar(-0.9)), e.op("gt", a, e.scalar(0.9)))),
    }


@pytest.mark.parametrize("layout", LAYOUTS + [([1024] * 50 + [576], 0.05, 0), ([300_000], 0.0, 2)])
def test_pipeline_fused_vs_unfused_oracle(gpu, ora, layout):
    """The fused batch loop against the oracle's step-by-step (materialising) evaluation."""
    lens, nf, off = layout
    rng = np.random.default_rng(77)
    cols = [make_chunks(rng, A.F64, lens, nf, off, "unit") for _ in range(3)] + [make_chunks(rng, A.I64, lens, nf, off, "plain")]
    e = A.Expr()
    for name, p in _pipeline_programs(e).items():
        exp = ora.pipeline(e, cols, p["values"], p["filt"])
        got = gpu.pipeline(e, cols, p["values"], p["filt"])
        for v, (g, x) in enumerate(zip(got, exp)):
            what = f"{name} value {v} lens={lens[:3]} nf={nf}"
            assert g.count == x.count and g.is_some == x.is_some and g.dtype == x.dtype, what
            if not x.is_some:
                continue
            if g.dtype in (A.F32, A.F64):
                assert abs(g.sum - x.sum) <= 1e-6 * max(abs(x.sum), 1e-9) + 1e-9, f"sum {what}: {g.sum} vs {x.sum}"
                assert g.min == pytest.approx(x.min, rel=1e-6) and g.max == pytest.approx(x.max, rel=1e-6), what
            else:
                assert (g.sum, g.min, g.max) == (x.sum, x.min, x.max), what


def test_pipeline_store_sink(gpu, ora):
    rng = np.random.default_rng(5)
    lens, nf, off = [1024, 1024, 100], 0.1, 3
    cols = [make_chunks(rng, A.F64, lens, nf, off, "unit") for _ in range(3)] + [make_chunks(rng, A.I64, lens, nf, off)]
    e = A.Expr()
    progs = _pipeline_programs(e)
    for name in ["c3_fused", "c1_sin_add_scalar", "casts"]:
        vals = progs[name]["values"]
        dts = {"c3_fused": [A.F64, A.I64], "c1_sin_add_scalar": [A.F64], "casts": [A.F64, A.I32]}[name]
        mk = lambda: [[A.HostArray.empty_out(dt, n, True) for n in lens] for dt in dts]
        got = gpu.pipeline(e, cols, vals, -1, A.SINK_STORE, mk())
        exp = ora.pipeline(e, cols, vals, -1, A.SINK_STORE, mk())
        for v in range(len(vals)):
            assert_chunks_match(got[v], exp[v], exact=(name != "c1_sin_add_scalar"), what=f"{name} value {v}")


def test_error_values_match_reference_semantics(gpu, ora):
    a = [A.HostArray.from_numpy(np.array([1.0, 2.0, 3.0]))]
    b = [A.HostArray.from_numpy(np.array([1.0, 0.0]))]
    for api in (gpu, ora):
        with pytest.raises(A.RdfError) as ei:  # compute::add length check -> ComputeError
            api.binary("add", a, b)
        assert ei.value.status == A.RDF_COMPUTE_ERROR
        z = [A.HostArray.from_numpy(np.array([1.0, 0.0, 2.0]))]
        with pytest.raises(A.RdfError) as ei:  # zero divisor at a valid slot -> DivideByZero (floats too)
            api.binary("divide", a, z)
        assert ei.value.status == A.RDF_DIVIDE_BY_ZERO
        zn = [A.HostArray.from_numpy(np.array([1.0, 0.0, 2.0]), valid=[1, 0, 1])]
        assert api.binary("divide", a, zn)[0].to_pylist() == [1.0, None, 1.5]  # ...but not at a null slot
        zi = [A.HostArray.from_numpy(np.array([4, 0, 2], dtype=np.int32))]
        with pytest.raises(A.RdfError) as ei:
            api.binary("divide", zi, zi)
        assert ei.value.status == A.RDF_DIVIDE_BY_ZERO
        with pytest.raises(A.RdfError) as ei:  # sin on integers: T::Native: Float
            api.unary("sin", zi)
        assert ei.value.status == A.RDF_INVALID_ARGUMENT
        with pytest.raises(A.RdfError) as ei:  # abs on unsigned: T::Native: Signed
            api.unary("abs", [A.HostArray.from_numpy(np.array([1], dtype=np.uint32))])
        assert ei.value.status == A.RDF_INVALID_ARGUMENT
        with pytest.raises(A.RdfError) as ei:  # take out of bounds
            api.take(a, A.HostArray.from_numpy(np.array([0, 3], dtype=np.uint32)))
        assert ei.value.status == A.RDF_COMPUTE_ERROR
        with pytest.raises(A.RdfError) as ei:  # filter length mismatch
            api.filter(a, [A.HostArray.from_numpy(np.array([True, False]), dtype=A.BOOL)])
        assert ei.value.status == A.RDF_COMPUTE_ERROR
        e = A.Expr()
        with pytest.raises(A.RdfError) as ei:  # add(f64, i32) without the Cast AddOperation inserts
            api.pipeline(e, [a, zi], [e.op("add", e.col(0), e.col(1))])
        assert ei.value.status == A.RDF_INVALID_ARGUMENT


def _sorted_groups(k, s, c):
    kv, km = k.to_numpy(), k.valid_mask()
    order = np.lexsort((kv, ~km))  # valid keys ascending, the NULL-key group last
    return kv[order], km[order], s.to_numpy()[order], c.to_numpy()[order]


@pytest.mark.parametrize("key_dtype", [A.I64, A.I32, A.U32, A.U8])
@pytest.mark.parametrize("val_dtype", [A.F64, A.I64, A.F32, None])
def test_groupby_sum(gpu, ora, key_dtype, val_dtype):
    """GROUP BY key -> sum, count: no reference execution exists (src/evaluation.rs:73), SQL semantics via the oracle."""
    rng = np.random.default_rng(900 + key_dtype)
    for lens, nf, off, ngroups in [([5], 0.0, 0, 3), ([1024, 1024, 576], 0.0, 0, 50), ([700, 0, 3000], 0.15, 13, 200), ([40_000], 0.05, 3, 5000)]:
        hi = min(ngroups, np.iinfo(A.NP_OF[key_dtype]).max)
        keys = []
        for n in lens:
            kv = rng.integers(0, hi, n).astype(A.NP_OF[key_dtype])
            if key_dtype == A.I64 and n > 2:
                kv[0], kv[1] = np.iinfo(np.int64).min, np.iinfo(np.int64).max  # the table's free marker is a legal key
            keys.append(A.HostArray.from_numpy(kv, valid=(rng.uniform(size=n) >= nf) if nf else None, offset=off, rng=rng))
        vals = make_chunks(rng, val_dtype, lens, nf, off) if val_dtype is not None else None
        exp = _sorted_groups(*ora.groupby_sum(keys, vals, ngroups + 8))
        got = _sorted_groups(*gpu.groupby_sum(keys, vals, ngroups + 8))
        what = f"keys={key_dtype} vals={val_dtype} lens={lens}"
        assert np.array_equal(got[1], exp[1]) and np.array_equal(got[0][exp[1]], exp[0][exp[1]]), "keys " + what
        assert np.array_equal(got[3], exp[3]), "counts " + what
        if val_dtype in (A.F64, A.F32):
            np.testing.assert_allclose(got[2], exp[2], rtol=1e-6, atol=1e-9, err_msg="sums " + what)
        else:
            assert np.array_equal(got[2], exp[2]), "sums " + what
    with pytest.raises(A.RdfError) as ei:  # more distinct keys than promised
        gpu.groupby_sum([A.HostArray.from_numpy(np.arange(5000, dtype=np.int64))], None, 100)
    assert ei.value.status == A.RDF_MEMORY_ERROR


def test_specialised_kernels_are_the_ones_that_run(gpu, request):
    """In 'spec' mode the catalog shapes must hit a specialised kernel (no silent fall-back to the interpreter);
    in 'interp' mode the general evaluator must run.  One aligned chunk each."""
    from rust_dataframe_amd import lib
    spec_mode = request.node.callspec.params["gpu"] == "spec"
    rng = np.random.default_rng(1)
    n = 5000
    e = A.Expr()
    c0, c1 = e.col(0), e.col(1)
    cases = []
    for dt in (A.F64, A.I64, A.U64, A.F32, A.I32, A.U32):
        a, b = make_chunks(rng, dt, [n], 0.1, 0, nonzero=True), make_chunks(rng, dt, [n], 0.0, 0, nonzero=True)
        cases.append((f"add {dt}", lambda a=a, b=b: gpu.binary("add", a, b)))
        cases.append((f"sum {dt}", lambda a=a: gpu.sum(a)))
    for dt in (A.F64, A.F32):
        a = make_chunks(rng, dt, [n], 0.1, 0, "unit")
        cases.append((f"sin {dt}", lambda a=a: gpu.unary("sin", a)))
        cases.append((f"mask {dt}", lambda a=a: gpu.predicate(e, e.op("gt", c0, e.scalar(0.25)), [a])))
        cases.append((f"filter->sum {dt}", lambda a=a: gpu.pipeline(e, [a], [c0], e.op("le", c0, e.scalar(0.5)))))
    x, y, z = (make_chunks(rng, A.F64, [n], 0.0, 0, "unit") for _ in range(3))
    k = make_chunks(rng, A.I64, [n], 0.0, 0)
    fma = e.op("add", e.op("multiply", c0, c1), e.col(2))
    cases.append(("C3", lambda: gpu.pipeline(e, [x, y, z, k], [fma, e.col(3)])))
    cases.append(("C1", lambda: gpu.pipeline(e, [x], [e.op("sin", e.op("add", c0, e.scalar(1.0)))])))
    # fused Calculate chains outside the exact catalog: shape-level kernels for every 8- / 4- / 2-byte type, up to three levels
    for dt in (A.F64, A.I64, A.U64, A.F32, A.I32, A.U32, A.I16, A.U16):
        cols = [make_chunks(rng, dt, [n], 0.0, 0, nonzero=True) for _ in range(3)]
        ks = e.scalar(2.0 if dt in (A.F64, A.F32) else 2, dt)
        two = e.op("subtract", e.op("multiply", c0, ks), c1)
        three = e.op("multiply", e.op("multiply", c0, e.op("subtract", ks, c1)), e.op("add", ks, e.col(2)))
        cases.append((f"two-level {dt}", lambda cols=cols, two=two: gpu.pipeline(e, cols, [two])))
        cases.append((f"three-level {dt}", lambda cols=cols, three=three: gpu.pipeline(e, cols, [three])))
        cases.append((f"three-level behind a filter {dt}", lambda cols=cols, three=three: gpu.pipeline(e, cols, [three], e.op("gt", c0, e.scalar(0.0)))))
    for name, call in cases:
        call()
        kern = lib.last_kernel()
        assert kern.startswith("spec_kernel<" if spec_mode else "eval_kernel<"), f"{name}: ran {kern}"


def test_pinned_frame_pipeline(gpu, ora, request):
    """rdf_frame_pin / rdf_pipeline_frame (and the _frame forms of rdf_predicate / rdf_group_pipeline): the same results as
    the plain calls over the same device-resident columns — both sinks, specialised and interpreted programs, several programs on one handle (cached tables per tile size and column
    order), nullable and unaligned chunk layouts, one chunk; host buffers and released handles are refused."""
    import ctypes as C
    import torch
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(404)
    e = A.Expr()
    c0, c1, c2 = e.col(0), e.col(1), e.col(2)
    programs = [([c0], e.op("gt", c0, e.scalar(0.1))),                                     # the headline shape
                ([e.op("add", e.op("multiply", c0, c1), c2)], -1),                          # C3's value
                ([e.op("multiply", c1, e.op("subtract", e.scalar(1.0), c2))], e.op("le", c0, e.scalar(0.5))),
                ([e.op("atan2", c0, c1)], -1)]                                              # interpreter
    for lens, off, nf in [([1024] * 7 + [500], 0, 0.0), ([4096, 0, 1000, 64], 0, 0.2), ([3000, 2000], 3, 0.1), ([5000], 0, 0.0)]:
        host = [make_chunks(rng, A.F64, lens, nf, off, "unit", nonzero=True) for _ in range(3)]
        keep, dev = [], []
        for col in host:   # the same chunks in device memory (values + bitmaps keep their element / bit offsets)
            dcol = []
            for ch in col:
                vt = torch.from_numpy(np.frombuffer(ch.values.tobytes() + b"\0" * 64, dtype=np.uint8).copy()).cuda()
                bt = None
                if ch.validity is not None:
                    bt = torch.from_numpy(np.frombuffer(ch.validity.tobytes() + b"\0" * 64, dtype=np.uint8).copy()).cuda()
                keep += [vt, bt]
                dcol.append(A.DeviceArray(vt.data_ptr(), bt.data_ptr() if bt is not None else None, ch.offset, ch.length, ch.dtype, -1))
            dev.append(dcol)
        torch.cuda.synchronize()
        with A.PinnedFrame(gpu, dev) as frame:
            for values, pred in programs:
                exp = ora.pipeline(e, host, values, pred)[0]
                got = gpu.pipeline(e, frame, values, pred)[0]
                plain = gpu.pipeline(e, dev, values, pred)[0]
                what = f"lens={lens} off={off}"
                assert got.count == exp.count == plain.count, what
                assert abs(got.sum - exp.sum) <= 1e-9 * max(1.0, abs(exp.sum)) and abs(got.sum - plain.sum) <= 1e-9 * max(1.0, abs(exp.sum)), what
                if exp.count:
                    assert (got.min, got.max) == (plain.min, plain.max), what
            # SINK_STORE through the handle
            v = e.op("add", e.op("multiply", c0, c1), c2)
            outs_e = [[A.HostArray.empty_out(A.F64, n, True) for n in lens]]
            ora.pipeline(e, host, [v], -1, A.SINK_STORE, outs_e)
            bufs = [(torch.zeros(n * 8 + 64, dtype=torch.uint8, device="cuda"), torch.zeros(n // 8 + 72, dtype=torch.uint8, device="cuda")) for n in lens]
            outs_g = [[A.DeviceArray(vb.data_ptr(), bb.data_ptr(), 0, n, A.F64, 0, keep=(vb, bb), capacity=n) for (vb, bb), n in zip(bufs, lens)]]
            torch.cuda.synchronize()
            gpu.pipeline(e, frame, [v], -1, A.SINK_STORE, outs_g)
            lib.synchronize()
            for (vb, bb), o, ee in zip(bufs, outs_g[0], outs_e[0]):
                assert o.length == ee.length and o.null_count == ee.null_count
                gv = vb.cpu().numpy()[:ee.length * 8].view(np.float64)
                m = ee.valid_mask()
                gm = np.unpackbits(bb.cpu().numpy()[:(ee.length + 7) // 8], bitorder="little")[:ee.length].astype(bool)
                assert np.array_equal(gm, m) and np.array_equal(gv[m], ee.to_numpy()[m])
            # rdf_predicate_frame: the mask of a comparison, device outputs
            pr = e.op("and", e.op("gt", c0, e.scalar(0.3)), e.op("lt", c1, c2))
            exp_m = ora.predicate(e, pr, host)
            mb = [(torch.zeros(n // 8 + 72, dtype=torch.uint8, device="cuda"), torch.zeros(n // 8 + 72, dtype=torch.uint8, device="cuda")) for n in lens]
            outs_m = [A.DeviceArray(vb.data_ptr(), bb.data_ptr(), 0, n, A.BOOL, 0, keep=(vb, bb), capacity=n) for (vb, bb), n in zip(mb, lens)]
            torch.cuda.synchronize()
            gpu.predicate(e, pr, frame, outs_m)
            lib.synchronize()
            for (vb, bb), o, ee in zip(mb, outs_m, exp_m):
                assert o.length == ee.length and o.null_count == ee.null_count
                m = ee.valid_mask()
                gv = np.unpackbits(vb.cpu().numpy()[:(ee.length + 7) // 8], bitorder="little")[:ee.length].astype(bool)
                gm = np.unpackbits(bb.cpu().numpy()[:(ee.length + 7) // 8], bitorder="little")[:ee.length].astype(bool)
                assert np.array_equal(gm, m) and np.array_equal(gv[m], ee.to_numpy()[m])
            # rdf_group_pipeline_frame: grouped sums (a key derived from column 0, filtered) — specialised and interpreted
            gk = e.cast(e.op("multiply", e.op("abs", c0), e.scalar(7.0)), A.I64)   # |x| <= 1 -> groups 0..7
            for values, pred in [([e.op("multiply", c1, c2), c1], e.op("le", c2, e.scalar(0.8))), ([e.op("atan2", c1, c2)], -1)]:
                exp_g = ora.group_pipeline(e, host, values, gk, 9, pred)
                got_g = gpu.group_pipeline(e, frame, values, gk, 9, pred)
                plain_g = gpu.group_pipeline(e, dev, values, gk, 9, pred)
                assert got_g[1] == exp_g[1] == plain_g[1]
                for rv_g, rv_e, rv_p in zip(got_g[0], exp_g[0], plain_g[0]):
                    for (sg, cg), (se, ce), (sp, cp) in zip(rv_g, rv_e, rv_p):
                        assert cg == ce == cp
                        assert abs(sg - se) <= 1e-9 * max(1.0, abs(se)) and abs(sg - sp) <= 1e-9 * max(1.0, abs(se))
    with pytest.raises(A.RdfError):       # host buffers are staged per call: nothing to pin
        A.PinnedFrame(gpu, [make_chunks(rng, A.F64, [100], 0.0, 0)])


def test_validity_window_load_paths_agree(gpu, ora):
    """The specialised kernels fetch a wave's validity words either with scalar loads (default) or with one vector load by
    lanes 0..NW and readlane (rdf_set_option("vec_bitmap", 1)): both give the oracle's aggregates and bitmaps on aligned,
    offset and many-chunk layouts."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(77)
    e = A.Expr()
    c0, c1 = e.col(0), e.col(1)
    pred = e.op("gt", c0, e.scalar(0.1))
    try:
        for lens, off in [([8192], 0), ([5000, 0, 3000, 1024], 0), ([1024] * 9 + [700], 0), ([4096, 1000], 16)]:
            x = make_chunks(rng, A.F64, lens, 0.2, off, "unit")
            y = make_chunks(rng, A.F64, lens, 0.1, off, "unit")
            exp = ora.pipeline(e, [x, y], [c1], pred)[0]
            exp_add = ora.binary("add", x, y)
            for vb in (0, 1):
                lib.set_option("vec_bitmap", vb)
                got = gpu.pipeline(e, [x, y], [c1], pred)[0]
                assert got.count == exp.count and abs(got.sum - exp.sum) <= 1e-9 * max(1.0, abs(exp.sum)), f"vec_bitmap={vb} lens={lens}"
                assert_chunks_match(gpu.binary("add", x, y), exp_add, exact=True, what=f"vec_bitmap={vb} lens={lens}")
    finally:
        lib.set_option("vec_bitmap", 0)


def test_worker_threads_own_their_contexts(gpu, ora):
    """Every entry point is re-entrant with per-thread state (the reference runs its kernels on rayon workers,
    src/functions/scalar.rs:28,99): calls from short-lived threads give the oracle's results, an error in one thread
    leaves the others alone, and the contexts of exited threads are released (the main thread keeps working)."""
    import threading
    rng = np.random.default_rng(5)
    a = make_chunks(rng, A.F64, [5000, 1200], 0.1, 0)
    b = make_chunks(rng, A.F64, [5000, 1200], 0.0, 0, nonzero=True)
    zeros = [A.HostArray.from_numpy(np.zeros(n)) for n in (5000, 1200)]
    exp = ora.binary("divide", a, b)
    exp_sum = ora.sum(a)
    results, errors = {}, {}

    def work(i):
        try:
            if i == 3:   # a failing call, found out on the device: a zero divisor at a valid slot
                gpu.binary("divide", a, zeros)
            results[i] = (gpu.binary("divide", a, b), gpu.sum(a))
        except Exception as ex:   # noqa: BLE001
            errors[i] = ex

    for _ in range(2):
        threads = [threading.Thread(target=work, args=(i,)) for i in range(6)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert set(errors) == {3} and isinstance(errors[3], A.RdfError) and errors[3].status == A.RDF_DIVIDE_BY_ZERO
        for i, (chunks, total) in results.items():
            assert_chunks_match(chunks, exp, exact=True, what=f"thread {i}")
            assert abs(total - exp_sum) <= 1e-9 * max(1.0, abs(exp_sum))
        results.clear(); errors.clear()
    assert_chunks_match(gpu.binary("divide", a, b), exp, exact=True, what="main thread afterwards")


def test_sort_reference_case(gpu, ora):
    """test_sort (src/dataframe.rs:963-1003): a desc, b asc, nulls last -> a = [4,3,3,1,1,null], b = [8,4,7,5,9,6]."""
    a = A.HostArray.from_numpy(np.array([1, 1, 0, 3, 3, 4], dtype=np.int32), valid=[1, 1, 0, 1, 1, 1])
    b = A.HostArray.from_numpy(np.array([9, 5, 6, 7, 4, 8], dtype=np.uint8))
    for api in (gpu, ora):
        idx = api.sort_to_indices([[a], [b]], [True, False])
        assert idx.to_numpy().tolist() == [5, 4, 3, 1, 0, 2]
        assert api.take([a], idx).to_pylist() == [4, 3, 3, 1, 1, None]
        assert api.take([b], idx).to_pylist() == [8, 4, 7, 5, 9, 6]


@pytest.mark.parametrize("dtype", NUMERIC)
def test_sort_to_indices(gpu, ora, dtype):
    rng = np.random.default_rng(1200 + dtype)
    for lens, nf, off in [([7], 0.0, 0), ([2048], 0.0, 0), ([1024, 1024, 576], 0.2, 3), ([700, 0, 5000], 0.1, 13), ([60_000], 0.05, 1)]:
        kind = "special" if dtype in (A.F32, A.F64) else ("extreme" if len(lens) == 1 else "plain")
        k1 = make_chunks(rng, dtype, lens, nf, off, kind)
        k2 = make_chunks(rng, A.I16, lens, 0.0, 0, "plain")
        for ch in k2:  # few distinct values -> many ties on the first key
            ch.values[:] = ch.values % 5
        for desc in ([False, False], [True, False], [False, True]):
            for cols in ([k2, k1], [k1]):
                d = desc[:len(cols)]
                got = gpu.sort_to_indices(cols, d).to_numpy()
                exp = ora.sort_to_indices(cols, d).to_numpy()
                assert np.array_equal(got, exp), f"sort dtype={dtype} lens={lens} desc={d} ncols={len(cols)}"


@pytest.mark.parametrize("dtype", [A.I64, A.F64, A.I32])
def test_sort_digit_passes_with_scanner_blocks(gpu, ora, dtype):
    """rdf_set_option("sort_pipe", 1): the digit passes publish a tile's counts one iteration before its offsets are asked for and
    take the offsets from scanner blocks (os_scatter3_kernel, round 6; measured slower than the look-back kernel and not the default —
    kept as the A/B partner, so it is held to the same oracle): several tiles per block, a ragged last tile, NULL keys (the
    NULLs-last pass), ties, descending, two key columns."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(1300 + dtype)
    try:
        lib.set_option("sort_pipe", 1)
        for lens, nf in [([300_000, 1, 123_457], 0.05), ([2_500_000], 0.0)]:
            kind = "special" if dtype == A.F64 else "extreme"
            k1 = make_chunks(rng, dtype, lens, nf, 3, kind)
            k2 = make_chunks(rng, A.I16, lens, 0.0, 0, "plain")
            for ch in k2:
                ch.values[:] = ch.values % 3
            for cols, d in (([k1], [False]), ([k2, k1], [True, False])):
                got = gpu.sort_to_indices(cols, d).to_numpy()
                exp = ora.sort_to_indices(cols, d).to_numpy()
                assert np.array_equal(got, exp), f"sort_pipe=1 dtype={dtype} lens={lens} desc={d}"
    finally:
        lib.set_option("sort_pipe", 0)


@pytest.mark.parametrize("dtype", [A.I64, A.F64, A.I32, A.U8])
def test_sort_digit_passes_on_super_tiles(gpu, ora, dtype):
    """Round 6: a ticket of the digit passes is K consecutive tiles — counted together, ONE look-back, then ranked and written one
    by one (os_scatter4_kernel; measured no faster than a tile per ticket and not the default — kept as its A/B partner, so it is
    held to the same oracle).  `sort_super` 100 + K takes that form on inputs of any size: K = 2, 3, 8 over 104 and 611 tiles (a ragged last super-tile, a last tile of one row's worth), NULL keys (the NULLs-last
    pass reads its digit through the row index), ties, descending, two key columns, value-bucket passes of doubles; and K = 1 by
    option (the look-back kernel) gives the same indices."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(1400 + dtype)
    try:
        for lens, nf in [([300_000, 1, 123_457], 0.05), ([2_500_000], 0.0)]:
            kind = "special" if dtype == A.F64 else "extreme"
            k1 = make_chunks(rng, dtype, lens, nf, 3, kind)
            k2 = make_chunks(rng, A.I16, lens, 0.0, 0, "plain")
            for ch in k2:
                ch.values[:] = ch.values % 3
            for cols, d in (([k1], [False]), ([k2, k1], [True, False])):
                exp = ora.sort_to_indices(cols, d).to_numpy()
                for k in (102, 103, 108, 1):
                    lib.set_option("sort_super", k)
                    got = gpu.sort_to_indices(cols, d).to_numpy()
                    assert np.array_equal(got, exp), f"sort_super={k} dtype={dtype} lens={lens} desc={d}"
    finally:
        lib.set_option("sort_super", 1)


@pytest.mark.parametrize("ngroups,n", [(20_000, 150_000), (300_000, 700_000), (1_300_000, 2_000_000)])
def test_groupby_partitioned_high_cardinality(gpu, ora, ngroups, n):
    """More than 1024 groups: records are scattered once on 9 hash bits (default) — or radix-sorted in 1-2 passes (the
    earlier variant, kept for very large domains) — and aggregated per partition in LDS; the HBM-atomics table is the third path.
    NULL keys, the free-marker keys of both tables, multi-chunk input with offsets; both value classes."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(ngroups)
    lens = [n // 3, 0, n - n // 3]
    for val_dtype in (A.F64, A.I64, None):
        keys, vals = [], ([] if val_dtype is not None else None)
        for ln in lens:
            kv = rng.integers(-ngroups // 2, ngroups // 2, ln).astype(np.int64)
            if ln > 4:
                kv[0], kv[1] = np.iinfo(np.int64).min, np.iinfo(np.int64).max
                kv[2] = np.int64(-3487469807577879104)  # mix64(key) == 2^64 - 1, the LDS free marker of the partition tables
                kv[3] = np.int64(7406324358081711299)   # the same for the single-pass path's hash (gb_hash)
            keys.append(A.HostArray.from_numpy(kv, valid=rng.uniform(size=ln) >= 0.01, offset=5, rng=rng))
            if val_dtype is not None:
                vals.append(A.HostArray.from_numpy(rng.uniform(-1, 1, ln) if val_dtype == A.F64 else rng.integers(-10 ** 9, 10 ** 9, ln), offset=2, dtype=val_dtype, rng=rng))
        exp = _sorted_groups(*ora.groupby_sum(keys, vals, ngroups + 8))
        for part in (3, 4, 1, 2, 0):  # second generation (auto / scatter path forced), first-generation single-pass and radix-sort partitioning, the HBM table
            lib.set_option("gb_partition", part)
            got = _sorted_groups(*gpu.groupby_sum(keys, vals, ngroups + 8))
            if part in (1, 2):
                assert lib.last_kernel().startswith("gb_aggregate_kernel" if part == 1 else "groupby_partitions_kernel")
            elif part:
                assert lib.last_kernel().startswith("gb2_"), lib.last_kernel()
            assert np.array_equal(got[1], exp[1]) and np.array_equal(got[0][exp[1]], exp[0][exp[1]]), f"keys part={part}"
            assert np.array_equal(got[3], exp[3]), f"counts part={part}"
            if val_dtype == A.F64:
                np.testing.assert_allclose(got[2], exp[2], rtol=1e-6, atol=1e-9)
            else:
                assert np.array_equal(got[2], exp[2])
        lib.set_option("gb_partition", 3)


def _pairs(l, r):
    return sorted(zip(l.to_pylist(), r.to_pylist()), key=lambda p: (p[0] is None, p[0] or 0, p[1] is None, p[1] or 0))


def test_join_reference_fixture(gpu, ora):
    """join_test_j1 / join_test_j2 (sql/postgresql/002.sql) and the row counts the reference's join tests assert
    (src/dataframe.rs:1006-1060): left join b=d -> 9 rows, right join a=d -> 10, inner join a=d -> 4."""
    a = [A.HostArray.from_numpy(np.array([0, 2, 3, 0, 0, 6, 6], dtype=np.int32), valid=[0, 1, 1, 0, 0, 1, 1])]
    b = [A.HostArray.from_numpy(np.array([1, 2, 3, 4, 5, 6, 60], dtype=np.int32))]
    d = [A.HostArray.from_numpy(np.array([1, 2, 3, 4, 4, 4, 5, 6, 7], dtype=np.int32))]
    for api in (gpu, ora):
        l, r = api.equijoin_indices(b, d, "left")
        assert l.length == 9 and r.null_count == 1            # b = 60 has no partner
        l, r = api.equijoin_indices(a, d, "right")
        assert l.length == 10 and l.null_count == 6
        l, r = api.equijoin_indices(a, d, "inner")
        assert l.length == 4 and _pairs(l, r) == [(1, 1), (2, 2), (5, 7), (6, 7)]
        l, r = api.equijoin_indices(a, d, "full")               # true full outer: 4 pairs + 3 NULL-key lefts + 6 unmatched rights
        assert l.length == 13 and l.null_count == 6 and r.null_count == 3


@pytest.mark.parametrize("dtype", [A.I64, A.I32, A.U8, A.F64])
@pytest.mark.parametrize("how", ["left", "right", "inner", "full"])
def test_equijoin_indices(gpu, ora, dtype, how):
    rng = np.random.default_rng(3000 + dtype)
    for (llens, rlens, card, nf) in [([9], [7], 5, 0.0), ([700, 0, 1300], [1024, 500], 300, 0.1), ([3000], [2500], 5000, 0.05)]:
        def side(lens, off):
            out = []
            for n in lens:
                v = rng.integers(0, min(card, 250 if dtype == A.U8 else card), n).astype(A.NP_OF[dtype])
                out.append(A.HostArray.from_numpy(v, valid=(rng.uniform(size=n) >= nf) if nf else None, offset=off, rng=rng))
            return out
        lk, rk = side(llens, 3), side(rlens, 5)
        gl, gr = gpu.equijoin_indices(lk, rk, how)
        el, er = ora.equijoin_indices(lk, rk, how)
        assert gl.length == el.length and gl.null_count == el.null_count and gr.null_count == er.null_count
        assert _pairs(gl, gr) == _pairs(el, er), f"join {how} dtype={dtype}"


def test_shape_specialised_runtime_op_kernels(gpu, ora, request):
    """Fused Calculate chains that are not in the exact catalog run on kernels specialised on the tree SHAPE with
    runtime operators: every arithmetic pair, scalar-first and right-nested operands (swap bit), trig on top, a
    `x CMP c [AND|OR y CMP d]` predicate in front, both sinks; results equal the oracle's unfused evaluation."""
    from rust_dataframe_amd import lib
    spec_mode = request.node.callspec.params["gpu"] == "spec"
    rng = np.random.default_rng(4242)
    lens = [4096, 1000]
    cols = [make_chunks(rng, A.F64, lens, nf, 0, kind="unit", nonzero=True) for nf in (0.0, 0.1, 0.0)]
    e = A.Expr()
    a, b, c = e.col(0), e.col(1), e.col(2)
    k1, k2 = e.scalar(1.5), e.scalar(-0.25)
    values = {}
    for o1 in ("add", "subtract", "multiply", "divide"):
        values[f"cc_{o1}"] = e.op(o1, a, b)
        values[f"ck_{o1}"] = e.op(o1, a, k1)
        values[f"kc_{o1}"] = e.op(o1, k1, a)                                   # scalar first: swap bit
        for o2 in ("add", "multiply", "divide"):
            values[f"ccc_{o1}_{o2}"] = e.op(o1, e.op(o2, a, b), c)
            values[f"c_cc_{o1}_{o2}"] = e.op(o1, c, e.op(o2, a, b))            # right-nested: swap bit on the outer op
            values[f"cck_{o1}_{o2}"] = e.op(o1, e.op(o2, a, b), k2)
            values[f"ckc_{o1}_{o2}"] = e.op(o1, e.op(o2, a, k1), b)
            values[f"ckk_{o1}_{o2}"] = e.op(o1, e.op(o2, k1, a), k2)
    for t in ("sin", "cos", "tan"):
        values[f"T_{t}_cc"] = e.op(t, e.op("subtract", a, b))
        values[f"T_{t}_ck"] = e.op(t, e.op("multiply", a, k1))
    preds = {"none": -1, "cmp": e.op("gt", a, e.scalar(-0.3)), "cmp_swapped": e.op("ge", e.scalar(0.2), b),
             "and": e.op("and", e.op("gt", a, e.scalar(-0.5)), e.op("lt", b, e.scalar(0.6))),
             "or": e.op("or", e.op("le", a, e.scalar(-0.5)), e.op("ne", c, e.scalar(0.0)))}
    hits = 0
    for vn, v in values.items():
        for pn, p in preds.items():
            exp = ora.pipeline(e, cols, [v], p)[0]
            got = gpu.pipeline(e, cols, [v], p)[0]
            k = lib.last_kernel()
            hits += k.startswith("spec_kernel<") 
            if spec_mode:
                assert k.startswith("spec_kernel<") or pn in ("and", "or") and vn.startswith(("ccc", "c_cc")), f"{vn}/{pn} ran on {k}"   # 5 columns do not fit
            assert got.count == exp.count, f"{vn}/{pn}"
            assert abs(got.sum - exp.sum) <= 1e-6 * max(abs(exp.sum), 1.0), f"{vn}/{pn}: {got.sum} vs {exp.sum}"
            if exp.count:
                assert np.isclose(got.min, exp.min, rtol=1e-9, atol=0) and np.isclose(got.max, exp.max, rtol=1e-9, atol=0), f"{vn}/{pn}"
        # SINK_STORE: the value as a new column
        outs_e = [[A.HostArray.empty_out(A.F64, n, True) for n in lens]]
        outs_g = [[A.HostArray.empty_out(A.F64, n, True) for n in lens]]
        ora.pipeline(e, cols, [v], -1, A.SINK_STORE, outs_e)
        gpu.pipeline(e, cols, [v], -1, A.SINK_STORE, outs_g)
        for ge, ee in zip(outs_g[0], outs_e[0]):
            assert_arrays_match(ge, ee, exact=not vn.startswith("T_"), what=vn)
    assert (hits > 0) == spec_mode
    # the combined predicate as a mask
    m = e.op("and", e.op("gt", a, e.scalar(-0.5)), e.op("lt", b, e.scalar(0.6)))
    for ge, ee in zip(gpu.predicate(e, m, cols), ora.predicate(e, m, cols)):
        assert_arrays_match(ge, ee, exact=True, what="mask")
    # division by a zero at a valid slot is still an error on this path
    z = [A.HostArray.from_numpy(np.array([1.0, 0.0, 2.0]))]
    with pytest.raises(A.RdfError) as ei:
        gpu.pipeline(e, [z, z, z], [e.op("add", e.op("divide", a, b), c)], -1)
    assert ei.value.status == A.RDF_DIVIDE_BY_ZERO


def test_kernels_compiled_at_run_time(gpu, ora, request):
    """Program shapes outside the catalogs — four levels of arithmetic, a function inside a chain, casts below the leaves, a
    wrapping integer tree — get spec_kernel<Prog> instantiated for them at run time (hiprtc, rdf_jit.cpp) instead of the
    interpreter: aggregates (plain and behind predicates, up to five distinct columns) and new columns against the oracle, the
    second call served from the process's cache, the interpreter with the switch off, and a program the kernel template cannot
    hold (more than eight literals)."""
    from rust_dataframe_amd import lib
    if request.node.callspec.params["gpu"] != "spec":
        pytest.skip("with the specialised kernels off nothing is compiled")
    rng = np.random.default_rng(20260927)
    lens = [4096, 1000, 0, 2500]
    F = [make_chunks(rng, A.F64, lens, nf, 0, kind="unit", nonzero=True) for nf in (0.0, 0.1, 0.0, 0.05)]
    I32 = make_chunks(rng, A.I32, lens, 0.05, 0, kind="plain", nonzero=True)
    L = [make_chunks(rng, A.I64, lens, nf, 0, kind="plain", nonzero=True) for nf in (0.0, 0.1)]
    cols = F + [I32] + L
    e = A.Expr()
    a, b, c, d, i, l, m = (e.col(k) for k in range(7))
    k1, k2 = e.scalar(1.5), e.scalar(-0.25)
    programs = {
        "four_levels": (e.op("multiply", e.op("subtract", e.op("divide", e.op("add", e.op("multiply", a, b), c), d), k1), a), A.F64),
        "function_inside": (e.op("add", e.op("multiply", e.op("sin", a), b), e.op("sqrt", e.op("abs", c))), A.F64),
        "casts_below": (e.op("multiply", e.op("add", e.cast(i, A.F64), a), e.cast(l, A.F64)), A.F64),
        "integer_tree": (e.op("subtract", e.op("multiply", e.op("add", l, e.scalar(3, A.I64)), l), e.op("multiply", m, e.op("add", m, e.scalar(7, A.I64)))), A.I64),
    }
    preds = {"none": -1, "cmp": e.op("gt", a, e.scalar(-0.3)), "two_columns": e.op("and", e.op("gt", a, e.scalar(-0.6)), e.op("ne", l, e.scalar(3, A.I64)))}
    lib.set_option("jit", 2)      # every call waits for its kernel
    try:
        for vn, (v, dt) in programs.items():
            for pn, p in preds.items():
                exp = ora.pipeline(e, cols, [v], p)[0]
                for attempt in range(2):                        # compiled, then found in the cache
                    got = gpu.pipeline(e, cols, [v], p)[0]
                    assert lib.last_kernel().startswith("spec_kernel<") and lib.last_kernel().endswith("[compiled at run time]"), f"{vn}/{pn} ran on {lib.last_kernel()}"
                    assert got.count == exp.count, f"{vn}/{pn}"
                    if dt == A.F64:
                        assert abs(got.sum - exp.sum) <= 1e-6 * max(abs(exp.sum), 1.0), f"{vn}/{pn}: {got.sum} vs {exp.sum}"
                        if exp.count:
                            assert np.isclose(got.min, exp.min, rtol=1e-9, atol=0) and np.isclose(got.max, exp.max, rtol=1e-9, atol=0), f"{vn}/{pn}"
                    else:
                        assert (got.sum, got.min, got.max) == (exp.sum, exp.min, exp.max), f"{vn}/{pn}"
            outs_e = [[A.HostArray.empty_out(dt, n, True) for n in lens]]
            outs_g = [[A.HostArray.empty_out(dt, n, True) for n in lens]]
            ora.pipeline(e, cols, [v], -1, A.SINK_STORE, outs_e)
            gpu.pipeline(e, cols, [v], -1, A.SINK_STORE, outs_g)
            assert lib.last_kernel().endswith("[compiled at run time]"), f"{vn} store ran on {lib.last_kernel()}"
            for ge, ee in zip(outs_g[0], outs_e[0]):
                assert_arrays_match(ge, ee, exact=vn != "function_inside", what=vn)
        # nine literals do not fit the kernel template (eight): the interpreter answers, as it does for everything with the switch off
        v = a
        for j in range(9):
            v = e.op("add", e.op("multiply", v, e.scalar(1.0 + 0.125 * j)), e.scalar(0.5 + j)) if j < 4 else e.op("subtract", v, e.scalar(0.03125 * (j + 1)))
        exp = ora.pipeline(e, cols, [v], -1)[0]
        got = gpu.pipeline(e, cols, [v], -1)[0]
        assert lib.last_kernel().startswith("eval_kernel<"), lib.last_kernel()
        assert got.count == exp.count and abs(got.sum - exp.sum) <= 1e-6 * max(abs(exp.sum), 1.0)
        lib.set_option("jit", 0)
        v = e.op("multiply", e.op("subtract", e.op("divide", e.op("add", e.op("multiply", b, c), d), a), k2), b)   # a shape not met above
        exp = ora.pipeline(e, cols, [v], -1)[0]
        got = gpu.pipeline(e, cols, [v], -1)[0]
        assert lib.last_kernel().startswith("eval_kernel<"), lib.last_kernel()
        assert got.count == exp.count and abs(got.sum - exp.sum) <= 1e-6 * max(abs(exp.sum), 1.0)
    finally:
        lib.set_option("jit", 0)


def test_kernels_compiled_at_run_time_cached_across_processes(tmp_path):
    """RDF_JIT_CACHE: the first process compiles the program's kernel and leaves the code object in the directory, the second one
    loads it from there (no compiler run) and computes the same aggregate."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        from rust_dataframe_amd import _abi as A, lib
        lib.set_device(0); api = lib.api()
        x = [A.HostArray.from_numpy(np.arange(5000, dtype=np.float64) / 7.0)]
        e = A.Expr(); a = e.col(0)
        v = e.op("add", e.op("multiply", e.op("sqrt", a), a), e.op("divide", e.op("subtract", a, e.scalar(2.0)), e.op("add", a, e.scalar(1.0))))
        r = api.pipeline(e, [x], [v], -1)[0]
        print("RESULT", repr(r.sum), lib.last_kernel())
    """ % root)
    script = script.replace("lib.set_device(0);", "lib.set_device(0); lib.set_option('jit', 2);")
    env = dict(os.environ, RDF_JIT_CACHE=str(tmp_path / "jit"), RDF_DEBUG_JIT="1")
    outs = []
    for _ in range(2):
        p = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(p)
    res = [[l for l in p.stdout.splitlines() if l.startswith("RESULT")][0] for p in outs]
    assert res[0] == res[1] and "[compiled at run time]" in res[0], res
    assert "loading" in outs[0].stderr and " from " not in outs[0].stderr.split("loading")[0]
    assert " from " + str(tmp_path / "jit") in outs[1].stderr, outs[1].stderr[-1500:]
    files = list((tmp_path / "jit").glob("*.hsaco"))
    assert len(files) == 1
    # the default directory ($XDG_CACHE_HOME/rdf_mi355x/jit) when RDF_JIT_CACHE is not set; without the wait ("jit" = 1) a cached
    # code object still answers the FIRST call of a process
    env2 = {k: v for k, v in os.environ.items() if k != "RDF_JIT_CACHE"}
    env2.update(XDG_CACHE_HOME=str(tmp_path / "xdg"), RDF_DEBUG_JIT="1")
    p1 = subprocess.run([sys.executable, "-c", script], env=env2, capture_output=True, text=True, timeout=300)
    p2 = subprocess.run([sys.executable, "-c", script.replace("lib.set_option('jit', 2);", "lib.set_option('jit', 1);")], env=env2, capture_output=True, text=True, timeout=300)
    assert p1.returncode == 0 and p2.returncode == 0, p1.stderr[-1000:] + p2.stderr[-1000:]
    assert len(list((tmp_path / "xdg" / "rdf_mi355x" / "jit").glob("*.hsaco"))) == 1
    assert "[compiled at run time]" in p2.stdout and " from " + str(tmp_path / "xdg") in p2.stderr, p2.stdout + p2.stderr[-1500:]
    # a file whose name matches but whose recorded signature does not (a collision, a stale object) is a miss, not a wrong kernel;
    # a directory others may write is not used at all
    raw = files[0].read_bytes()
    assert raw[:7] == b"RDFJIT1"
    files[0].write_bytes(raw[:12] + bytes([raw[12] ^ 1]) + raw[13:])
    p3 = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert p3.returncode == 0 and [l for l in p3.stdout.splitlines() if l.startswith("RESULT")][0] == res[0]
    assert " from " + str(tmp_path / "jit") not in p3.stderr and "loading" in p3.stderr
    os.chmod(tmp_path / "jit", 0o777)
    p4 = subprocess.run([sys.executable, "-c", script + "print('STATUS', lib.jit_status())"], env=env, capture_output=True, text=True, timeout=300)
    assert p4.returncode == 0 and "writable by group / others" in p4.stdout and " from " not in p4.stderr, p4.stdout + p4.stderr[-800:]


def test_run_time_compiler_in_the_background_then_swapped_in(gpu, ora, request, tmp_path):
    """The default mode ("jit" = 1): thirty program shapes no catalog holds are met for the first time — every first call is answered
    at once by the interpreter while `hipcc` children compile the kernels on helper threads (six at a time); once a kernel is ready
    the same call runs on it.  Every answer, interpreted or compiled, is held to the oracle (integers bit-exact, f64 sums to 1e-6
    relative).  The bounded subset of the RDF_TEST_JIT=1 sweep that runs in the default suite."""
    import time
    from rust_dataframe_amd import lib
    if request.node.callspec.params["gpu"] != "spec":
        pytest.skip("with the specialised kernels off nothing is compiled")
    rng = np.random.default_rng(777)
    lens = [2048, 1500, 0, 700]
    F = [make_chunks(rng, A.F64, lens, nf, 0, kind="unit", nonzero=True) for nf in (0.0, 0.1, 0.0, 0.05)]
    G = [make_chunks(rng, A.F32, lens, nf, 0, kind="unit", nonzero=True) for nf in (0.0, 0.1)]
    L = [make_chunks(rng, A.I64, lens, nf, 0, kind="plain", nonzero=True) for nf in (0.0, 0.1)]
    cols = F + G + L                                  # eight columns: what a fused program may read
    e = A.Expr()
    a, b, c, d, f, g, l, m = (e.col(k) for k in range(8))
    K = [e.scalar(0.5 + 0.25 * j) for j in range(4)]
    progs = []
    for j, (p, q, r, s_) in enumerate([(a, b, c, d), (b, c, d, a), (c, d, a, b), (d, a, b, c)]):
        progs.append((f"four_levels_{j}", e.op("subtract", e.op("multiply", e.op("add", e.op("divide", e.op("subtract", p, q), r), s_), K[j]), p), -1))
        progs.append((f"trig_inside_{j}", e.op("multiply", e.op(["sin", "cos", "tan", "sqrt"][j], e.op("abs", p)), e.op("add", q, r)), -1))
        progs.append((f"behind_filter_{j}", e.op("add", e.op("multiply", e.op("multiply", p, q), e.op("subtract", r, K[j])), s_), e.op("gt", s_, e.scalar(-0.2 * j))))
    for j, (p, q) in enumerate([(l, m), (m, l)]):
        progs.append((f"integer_tree_{j}", e.op("add", e.op("multiply", e.op("subtract", e.op("multiply", p, q), p), e.scalar(3 + j, A.I64)), e.op("multiply", q, q)), -1))
        progs.append((f"casts_below_{j}", e.op("multiply", e.op("add", e.cast(p, A.F64), a), e.op("subtract", e.cast(f, A.F64), e.cast(q, A.F64))), e.op("ne", p, e.scalar(7, A.I64))))
    for j, (p, q) in enumerate([(f, g), (g, f)]):
        progs.append((f"f32_chain_{j}", e.op("add", e.op("multiply", e.op("sin", p), q), e.op("multiply", e.op("sqrt", e.op("abs", q)), e.op("subtract", p, e.scalar(0.25 * (j + 1), A.F32)))), -1))
        progs.append((f"f32_cast_{j}", e.op("multiply", e.op("add", e.cast(p, A.F64), b), e.op("subtract", e.cast(q, A.F64), K[j])), -1))
    ops = ["add", "subtract", "multiply", "divide"]
    for j in range(10):                              # ten more four-level trees, every one a different operator sequence
        o = [ops[(j + t * (1 + j // 4)) % 4] for t in range(4)]
        progs.append((f"ops_{'_'.join(o)}_{j}", e.op(o[3], e.op(o[2], e.op(o[1], e.op(o[0], a, b), c), d), e.op(o[(j + 2) % 4], b, K[j % 4])), e.op("lt", c, e.scalar(0.5)) if j % 2 else -1))
    assert len({n for n, _, _ in progs}) == len(progs) == 30

    def check(name, got, exp):
        assert got.count == exp.count, name
        if exp.dtype == A.I64:
            assert (got.sum, got.min, got.max) == (exp.sum, exp.min, exp.max), name
        else:
            assert abs(got.sum - exp.sum) <= 1e-6 * max(abs(exp.sum), 1.0), f"{name}: {got.sum} vs {exp.sum}"    # 1e-6 relative (north_star)
    before = lib.jit_status()
    assert before.startswith("run-time compiler ready"), before
    lib.set_option("jit", 1)
    try:
        exps, interpreted = {}, 0
        t0 = time.perf_counter()
        for name, v, p in progs:                     # first meeting: nobody waits for a compiler
            exps[name] = ora.pipeline(e, cols, [v], p)[0]
            check(name, gpu.pipeline(e, cols, [v], p)[0], exps[name])
            interpreted += lib.last_kernel().startswith("eval_kernel<")
        first_pass = time.perf_counter() - t0
        # (a code object already in the cache directory loads at once; under RDF_TEST_JIT=1 the tests before this one have compiled
        # some of these shapes into the session's cache, so "first meeting" is not what the count can hold them to there)
        if os.environ.get("RDF_TEST_JIT") != "1":
            assert interpreted >= len(progs) - 2, f"{interpreted} of {len(progs)} first calls interpreted"
        lib.set_option("jit", 2)                     # now wait for each kernel: it was compiled meanwhile, or is finished here
        for name, v, p in progs:
            check(name + " compiled", gpu.pipeline(e, cols, [v], p)[0], exps[name])
            assert lib.last_kernel().endswith("[compiled at run time]"), f"{name} ran on {lib.last_kernel()}"
        lib.set_option("jit", 1)
        for name, v, p in progs[:5]:                 # and the default mode finds them ready
            gpu.pipeline(e, cols, [v], p)
            assert lib.last_kernel().endswith("[compiled at run time]"), name
        st = lib.jit_status()
        import re
        failed = lambda text: int(re.search(r"(\d+) failed", text).group(1))
        # (none of THESE thirty failed to compile; under RDF_TEST_JIT=1 earlier tests have met shapes the kernel template refuses by design)
        assert failed(st) == failed(before) and "0 in progress" in st, (before, st)
        print(f"{len(progs)} shapes: first pass (interpreted) {first_pass:.2f} s, all compiled after {time.perf_counter() - t0:.2f} s; {st}")
    finally:
        lib.set_option("jit", 0)


def test_shape_specialised_kernels_i64_and_mixed_predicates(gpu, ora, request):
    """The runtime-operator kernels for wrapping i64 arithmetic (bit-exact, incl. MIN / -1 and the divide-by-zero
    error), and predicates on a column of the other dtype (an i64 key in front of f64 measures and vice versa)."""
    from rust_dataframe_amd import lib
    spec_mode = request.node.callspec.params["gpu"] == "spec"
    rng = np.random.default_rng(99)
    lens = [3000, 1200]
    K = make_chunks(rng, A.I64, lens, 0.05, 0, kind="extreme", nonzero=True)
    L = make_chunks(rng, A.I64, lens, 0.0, 0, kind="plain", nonzero=True)
    X = make_chunks(rng, A.F64, lens, 0.1, 0, kind="unit", nonzero=True)
    cols = [K, L, X]
    e = A.Expr()
    k, l, x = e.col(0), e.col(1), e.col(2)
    i3, f2 = e.scalar(3, A.I64), e.scalar(2.5)
    values = {"ll": e.op("multiply", k, l), "lk": e.op("subtract", i3, l), "lll": e.op("add", e.op("multiply", k, l), l),
              "llk_div": e.op("divide", e.op("add", k, l), i3), "lkk": e.op("multiply", e.op("add", l, i3), i3),
              "div_ll": e.op("divide", k, l), "x_ck": e.op("multiply", x, f2), "x_T": e.op("sin", e.op("add", x, f2))}
    preds = {"none": -1, "i64_pred": e.op("gt", l, e.scalar(0, A.I64)), "f64_pred": e.op("lt", x, e.scalar(0.25)),
             "and_i64": e.op("and", e.op("ge", l, e.scalar(-500, A.I64)), e.op("ne", k, e.scalar(7, A.I64)))}
    for vn, v in values.items():
        for pn, p in preds.items():
            exp = ora.pipeline(e, cols, [v], p)[0]
            got = gpu.pipeline(e, cols, [v], p)[0]
            if spec_mode and not (vn == "lll" and pn == "and_i64"):   # 2 + 3 column slots do not fit the 4 a kernel reads
                assert lib.last_kernel().startswith("spec_kernel<"), f"{vn}/{pn} ran on {lib.last_kernel()}"
            assert got.count == exp.count, f"{vn}/{pn}"
            if exp.dtype == A.I64:
                assert (got.sum, got.min, got.max) == (exp.sum, exp.min, exp.max), f"{vn}/{pn}"
            else:
                assert abs(got.sum - exp.sum) <= 1e-6 * max(abs(exp.sum), 1.0), f"{vn}/{pn}"
        if vn.startswith("l") or vn == "div_ll":
            outs_e = [[A.HostArray.empty_out(A.I64, n, True) for n in lens]]
            outs_g = [[A.HostArray.empty_out(A.I64, n, True) for n in lens]]
            ora.pipeline(e, cols, [v], -1, A.SINK_STORE, outs_e)
            gpu.pipeline(e, cols, [v], -1, A.SINK_STORE, outs_g)
            for ge, ee in zip(outs_g[0], outs_e[0]):
                assert_arrays_match(ge, ee, exact=True, what=vn)
    z = [A.HostArray.from_numpy(np.array([4, 0, -9], dtype=np.int64))]
    with pytest.raises(A.RdfError) as ei:
        gpu.pipeline(e, [z, z, z], [e.op("add", e.op("divide", k, l), l)], -1)
    assert ei.value.status == A.RDF_DIVIDE_BY_ZERO
    mn = [A.HostArray.from_numpy(np.array([np.iinfo(np.int64).min, 10], dtype=np.int64))]
    m1 = [A.HostArray.from_numpy(np.array([-1, 3], dtype=np.int64))]
    got = gpu.pipeline(e, [mn, m1, m1], [e.op("add", e.op("divide", k, l), e.scalar(0, A.I64))], -1)[0]
    exp = ora.pipeline(e, [mn, m1, m1], [e.op("add", e.op("divide", k, l), e.scalar(0, A.I64))], -1)[0]
    assert (got.sum, got.min, got.max) == (exp.sum, exp.min, exp.max)


@pytest.mark.parametrize("dtype", [A.F64, A.I64, A.U64, A.F32, A.I32, A.U32, A.I16, A.U16])
def test_shape_kernels_three_levels_and_every_wide_type(gpu, ora, request, dtype):
    """Evaluate::calculate's type matrix (src/evaluation.rs:107-293) as fused chains: one- to three-level arithmetic trees
    (left-deep, balanced, a two-level beside a one-level subtree, in either operand order), sin / cos / tan on top for
    the float types, plain and behind `x CMP c`, both sinks, on every 8-, 4- and 2-byte numeric type.  In 'spec' mode they
    must run on a shape-specialised kernel; the results equal the oracle's unfused evaluation (integers bit-exact)."""
    from rust_dataframe_amd import lib
    spec_mode = request.node.callspec.params["gpu"] == "spec"
    rng = np.random.default_rng(500 + dtype)
    lens = [4096, 1500]
    is_float = dtype in (A.F64, A.F32)
    cols = [make_chunks(rng, dtype, lens, nf, 0, kind="unit" if is_float else "plain", nonzero=True) for nf in (0.0, 0.1, 0.0, 0.05)]
    e = A.Expr()
    a, b, c, d = (e.col(i) for i in range(4))
    k1, k2 = (e.scalar(1.5, dtype), e.scalar(-0.25, dtype)) if is_float else (e.scalar(3, dtype), e.scalar(7, dtype))
    ops3 = ("add", "subtract", "multiply")
    values = {}
    for i, o1 in enumerate(("add", "subtract", "multiply", "divide")):
        o2, o3 = ops3[i % 3], ops3[(i + 1) % 3]
        values[f"cc_{o1}"] = e.op(o1, a, b)
        values[f"kc_{o1}"] = e.op(o1, k1, a)
        values[f"ccc_{o1}"] = e.op(o1, e.op(o2, a, b), c)
        values[f"ckk_{o1}"] = e.op(o1, e.op(o2, k1, a), k2)
        # three levels, left-deep
        values[f"cccc_{o1}"] = e.op(o1, e.op(o2, e.op(o3, a, b), c), d)
        values[f"cckc_{o1}"] = e.op(o1, e.op(o2, e.op(o3, a, b), k1), c)
        values[f"ckck_{o1}"] = e.op(o1, e.op(o2, e.op(o3, a, k1), b), k2)
        values[f"ckkc_{o1}"] = e.op(o1, e.op(o2, e.op(o3, k2, a), k1), b)
        values[f"c_ccc_{o1}"] = e.op(o2, d, e.op(o3, e.op(o1, a, b), c))          # right-nested twice: swap bits
        # balanced and (two-level)(one-level); a division keeps a leaf divisor
        values[f"cc_cc_{o2}_{o1}"] = e.op(o2, e.op(o1, a, b), e.op(o3, c, d))
        values[f"ck_cc_{o2}_{o1}"] = e.op(o2, e.op(o1, a, k1), e.op(o3, b, c))
        values[f"ck_ck_{o2}_{o1}"] = e.op(o2, e.op(o1, k1, a), e.op(o3, b, k2))
        values[f"ccc_ck_{o2}_{o1}"] = e.op(o2, e.op(o3, e.op(o1, a, b), c), e.op(o3, d, k1))
        values[f"ck_ckc_{o2}_{o1}"] = e.op(o2, e.op(o3, k2, c), e.op(o3, e.op(o1, a, k1), b))   # the deeper subtree second: swap
    values["q1_charge"] = e.op("multiply", e.op("multiply", a, e.op("subtract", k1, b)), e.op("add", k1, c))   # price * (1 - disc) * (1 + tax)
    if is_float:
        for t in ("sin", "cos", "tan"):
            values[f"T_{t}_c"] = e.op(t, a)
            values[f"T_{t}_ck"] = e.op(t, e.op("multiply", a, k1))
            values[f"T_{t}_cck"] = e.op(t, e.op("add", e.op("multiply", a, b), k2))
    preds = {"none": -1, "cmp": e.op("gt", c, e.scalar(-0.3 if is_float else -200.0))}
    rtol = 1e-6 if dtype == A.F64 else 1e-4
    for vn, v in values.items():
        for pn, p in preds.items():
            exp = ora.pipeline(e, cols, [v], p)[0]
            got = gpu.pipeline(e, cols, [v], p)[0]
            k = lib.last_kernel()
            if spec_mode and not (pn == "cmp" and vn.startswith(("cccc", "c_ccc", "cc_cc", "ccc_ck"))):   # 4 value columns + the predicate's do not fit
                assert k.startswith("spec_kernel<"), f"{vn}/{pn} ran on {k}"
            if not spec_mode:
                assert k.startswith("eval_kernel<"), f"{vn}/{pn} ran on {k}"
            assert got.count == exp.count, f"{vn}/{pn}"
            if is_float:
                assert abs(got.sum - exp.sum) <= rtol * max(abs(exp.sum), 1.0), f"{vn}/{pn}: {got.sum} vs {exp.sum}"
                if exp.count:
                    assert np.isclose(got.min, exp.min, rtol=rtol, atol=0) and np.isclose(got.max, exp.max, rtol=rtol, atol=0), f"{vn}/{pn}"
            else:
                assert (got.sum, got.min, got.max) == (exp.sum, exp.min, exp.max), f"{vn}/{pn}"
        outs_e = [[A.HostArray.empty_out(dtype, n, True) for n in lens]]
        outs_g = [[A.HostArray.empty_out(dtype, n, True) for n in lens]]
        ora.pipeline(e, cols, [v], -1, A.SINK_STORE, outs_e)
        gpu.pipeline(e, cols, [v], -1, A.SINK_STORE, outs_g)
        if spec_mode:
            assert lib.last_kernel().startswith("spec_kernel<"), f"{vn}/store ran on {lib.last_kernel()}"
        for ge, ee in zip(outs_g[0], outs_e[0]):
            assert_arrays_match(ge, ee, exact=not vn.startswith("T_"), what=f"{vn} dtype={dtype}")
    # a zero divisor at a valid slot is an error on these kernels too
    z = [A.HostArray.from_numpy(np.array([4, 0, 9]).astype(A.NP_OF[dtype]))]
    with pytest.raises(A.RdfError) as ei:
        gpu.pipeline(e, [z, z, z, z], [e.op("add", e.op("multiply", e.op("divide", a, b), c), d)], -1)
    assert ei.value.status == A.RDF_DIVIDE_BY_ZERO


@pytest.mark.parametrize("to_dt,from_dt", [(A.I64, A.I32), (A.F64, A.F32), (A.F64, A.I64), (A.I32, A.I64), (A.F32, A.I16), (A.U16, A.I64),
                                           (A.I64, A.F64), (A.U32, A.F32), (A.F64, A.U16), (A.I16, A.U64)])
def test_operands_of_different_types_run_specialised(gpu, ora, request, to_dt, from_dt):
    """The reference plans `a OP b` over columns of different types as cast(b -> a's type) then OP (AddOperation ... DivideOperation,
    src/operation/scalar.rs:47-72), casts on their own (Function::Cast, src/evaluation.rs:296-315) and sin / cos / tan of an integer
    column as cast -> Float64 -> sin (SinOperation, :256-294).  Fused, those are programs whose columns differ in WIDTH: they run on
    the specialised kernels with per-column load widths (spec mode) and equal the oracle's unfused evaluation, lossy casts included
    (a value the target cannot hold is NULL and stays NULL through the arithmetic)."""
    from rust_dataframe_amd import lib
    spec_mode = request.node.callspec.params["gpu"] == "spec"
    rng = np.random.default_rng(900 + 16 * to_dt + from_dt)
    lens = [4096, 1000]
    to_float, from_float = to_dt in (A.F64, A.F32), from_dt in (A.F64, A.F32)
    a = make_chunks(rng, to_dt, lens, 0.1, 0, kind="unit" if to_float else "plain", nonzero=True)
    # the cast operand: mostly small values, a few that the target type cannot hold (lossy pairs) / NaN
    b = []
    for n in lens:
        v = rng.uniform(1.0, 900.0, n) * rng.choice([-1.0, 1.0], n) if from_float else rng.integers(-900 if from_dt in (A.I64, A.I32, A.I16) else 0, 900, n)   # (|v| >= 1: a cast to an integer must not make a zero divisor)
        v = np.asarray(v).astype(A.NP_OF[from_dt])
        v[v == 0] = 1
        big = {A.I64: 2 ** 40, A.U64: 2 ** 40, A.I32: 2 ** 30, A.U32: 2 ** 31 + 5, A.F64: 1e30, A.F32: 1e30, A.I16: 30000, A.U16: 60000}[from_dt]
        v[::97] = big
        if from_float and not to_float:   # NaN -> integer is NULL (between floats a NaN would only test sign / payload conventions)
            v[5::131] = np.nan
        b.append(A.HostArray.from_numpy(v, valid=rng.uniform(size=n) >= 0.05))
    e = A.Expr()
    c0, c1 = e.col(0), e.col(1)
    cb = e.cast(c1, to_dt)
    values = {op: e.op(op, c0, cb) for op in ("add", "subtract", "multiply", "divide")}
    if to_float:
        values["sin_cast"] = e.op("sin", cb)
    for name, v in values.items():
        exp = ora.pipeline(e, [a, b], [v])[0]
        got = gpu.pipeline(e, [a, b], [v])[0]
        if spec_mode:
            assert lib.last_kernel().startswith("spec_kernel<"), f"{name}: ran {lib.last_kernel()}"
        assert got.count == exp.count, name
        if to_float:
            assert abs(got.sum - exp.sum) <= 1e-5 * max(abs(exp.sum), 1.0) or (np.isnan(exp.sum) and np.isnan(got.sum)) or (np.isinf(exp.sum) and got.sum == exp.sum), f"{name}: {got.sum} vs {exp.sum}"
        else:
            assert (got.sum, got.min, got.max) == (exp.sum, exp.min, exp.max), name
        outs_e = [[A.HostArray.empty_out(to_dt, n, True) for n in lens]]
        outs_g = [[A.HostArray.empty_out(to_dt, n, True) for n in lens]]
        ora.pipeline(e, [a, b], [v], -1, A.SINK_STORE, outs_e)
        gpu.pipeline(e, [a, b], [v], -1, A.SINK_STORE, outs_g)
        if spec_mode:
            assert lib.last_kernel().startswith("spec_kernel<"), f"{name}/store: ran {lib.last_kernel()}"
        for ge, ee in zip(outs_g[0], outs_e[0]):
            assert_arrays_match(ge, ee, exact=name != "sin_cast", what=f"{name} {from_dt}->{to_dt}")
    # the cast alone
    assert_chunks_match(gpu.cast(b, to_dt), ora.cast(b, to_dt), exact=True, what=f"cast {from_dt}->{to_dt}")
    if spec_mode:
        assert lib.last_kernel().startswith("spec_kernel<"), f"cast: ran {lib.last_kernel()}"


def test_sort_skips_constant_key_bytes(gpu, ora):
    """Radix passes cover only the bytes of (max key - min key): constant keys (no pass needed at all), a 1-byte range
    inside an i64, sparse byte patterns, negative small ranges (sign extension is not a range), two-valued floats, with
    no / some / only NULLs — always the oracle's order."""
    rng = np.random.default_rng(77)
    n = 10_000
    cases = {
        "all_equal": np.full(n, 123456789, dtype=np.int64),
        "one_byte": rng.integers(0, 200, n).astype(np.int64),
        "bytes_2_and_5": (rng.integers(0, 256, n).astype(np.int64) << 16) | (rng.integers(0, 256, n).astype(np.int64) << 40),
        "negative_small": -rng.integers(0, 1000, n).astype(np.int64),
        "f64_two_values": rng.choice(np.array([1.5, -2.25]), n),
    }
    for name, v in cases.items():
        for valid in (None, rng.uniform(size=n) > 0.3, np.zeros(n, dtype=bool)):
            for desc in (False, True):
                col = [A.HostArray.from_numpy(v[:6000], valid=None if valid is None else valid[:6000]),
                       A.HostArray.from_numpy(v[6000:], valid=None if valid is None else valid[6000:])]
                got = gpu.sort_to_indices([col], [desc]).to_numpy()
                exp = ora.sort_to_indices([col], [desc]).to_numpy()
                assert np.array_equal(got, exp), f"{name} desc={desc} nulls={'none' if valid is None else int((~valid).sum())}"
    # two columns where the more significant one is constant
    c0 = [A.HostArray.from_numpy(np.full(n, 7, dtype=np.int32))]
    c1 = [A.HostArray.from_numpy(rng.integers(-50, 50, n).astype(np.int16))]
    for cols in ([c0, c1], [c1, c0]):
        assert np.array_equal(gpu.sort_to_indices(cols, [False, True]).to_numpy(), ora.sort_to_indices(cols, [False, True]).to_numpy())


@pytest.mark.parametrize("how", ["left", "right", "inner", "full"])
def test_equijoin_bucket_index_edge_cases(gpu, ora, how):
    """The probe goes through a table of the distinct build keys, or (A/B, and for several key columns) through a bucket index on
    the top bits of (key - min key): sparse keys over the whole i64 range
    (many empty buckets, shift > 0), heavy skew (one bucket holds most rows), keys outside the build range, a single
    distinct build key, extreme keys, negative floats and NaN-free f64 keys; pairs equal the oracle's nested loops."""
    rng = np.random.default_rng(9100)
    i64 = np.iinfo(np.int64)
    def mk(v, nf=0.0):
        v = np.asarray(v)
        return [A.HostArray.from_numpy(v, valid=(rng.uniform(size=len(v)) >= nf) if nf else None)]
    cases = []
    sparse = rng.integers(i64.min // 2, i64.max // 2, 1500)
    cases.append((mk(rng.choice(sparse, 2500)), mk(np.concatenate([sparse, sparse[:200]]), 0.05)))           # sparse, duplicates on the build side
    skew = np.where(rng.uniform(size=2000) < 0.9, 42, rng.integers(0, 10 ** 9, 2000))
    cases.append((mk(np.concatenate([np.full(30, 42), rng.integers(0, 10 ** 9, 500)])), mk(skew)))            # one hot key: 30 x 1800 pairs
    cases.append((mk(rng.integers(-100, 300, 3000), 0.1), mk(rng.integers(0, 200, 1000))))                     # probe keys below / above the build range
    cases.append((mk(rng.integers(5, 9, 500)), mk(np.full(40, 7))))                                            # single distinct build key
    cases.append((mk(np.array([i64.min, i64.max, 0, -1, i64.min, 5])), mk(np.array([i64.max, i64.min, i64.min, 7]))))
    cases.append((mk(np.round(rng.uniform(-3, 3, 2000), 1)), mk(np.round(rng.uniform(-3, 3, 700), 1), 0.05)))   # f64 keys incl. -0.0 / 0.0 patterns
    cases.append((mk(rng.integers(0, 50, 100)), mk(np.array([], dtype=np.int64))))                            # empty build side
    from rust_dataframe_amd import lib
    for lk, rk in cases:
        el, er = ora.equijoin_indices(lk, rk, how)
        # the table of distinct build keys — laid out by a scan over the hash-ordered build side (2, default) or claimed slot by slot
        # with compare-and-swap (1, round 3) — and the bucket index over the sorted build keys (0)
        for table in (2, 1, 0):
            lib.set_option("join_table", table)
            try:
                gl, gr = gpu.equijoin_indices(lk, rk, how)
            finally:
                lib.set_option("join_table", 2)
            assert gl.length == el.length and gl.null_count == el.null_count and gr.null_count == er.null_count
            assert _pairs(gl, gr) == _pairs(el, er), f"join {how} table={table}"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["distinct", "duplicates", "few_keys", "few_keys_wide_range", "crowded_slots"])
def test_equijoin_table_placed_by_scan(gpu, shape):
    """Round 5: the table of distinct build keys is laid out without atomics — the build side sorted by key * golden ratio, slot(r) =
    r + prefix-max(home - r) over the distinct keys (three kernels: per-tile aggregates, one block scanning them, placement).  Sizes
    that span thousands of 4096-row tiles (the one-block scan handles several tiles per thread), against the compare-and-swap table
    of round 3: identical index arrays (the probe rows' order and, inside a probe row, the build rows' order do not depend on how the
    table was built), pair counts against numpy.  `crowded_slots`: keys whose hashes pile onto the table's last home slot, far beyond its
    margin — the placement flags it and the call falls back to the round-3 table."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(515)
    if shape == "distinct":
        nb, npb = 6_000_000, 3_000_000
        bk = rng.permutation(nb).astype(np.int64) * 3
        pk = rng.integers(0, 3 * nb, npb).astype(np.int64)
    elif shape == "few_keys":
        # a build side of 3e6 rows over 37 dictionary codes: the table is sized from the key RANGE (64 slots), not from the rows (round 6:
        # 6e6 slots whose empty stretches a handful of lanes had to write one after another); 1 probe row in 100 finds a key
        nb, npb = 3_000_000, 2_000
        bk = rng.integers(0, 37, nb).astype(np.int64)
        pk = rng.integers(0, 3700, npb).astype(np.int64)
    elif shape == "few_keys_wide_range":
        # ... and 37 keys spread over the whole 64-bit range (hashed identifiers): the range says nothing, so the sorted build side is
        # read once more to count its distinct keys and the table is sized from that (round 6)
        nb, npb = 3_000_000, 2_000
        ids = rng.integers(-(1 << 62), 1 << 62, 37).astype(np.int64)
        bk = ids[rng.integers(0, 37, nb)]
        pk = np.concatenate([ids[rng.integers(0, 37, 20)], rng.integers(-(1 << 62), 1 << 62, npb - 20).astype(np.int64)])
    elif shape == "duplicates":
        nb, npb = 5_000_000, 500_000
        bk = rng.integers(0, 1_000_000, nb).astype(np.int64)          # ~5 build rows per key
        pk = rng.integers(-1000, 1_001_000, npb).astype(np.int64)
    else:
        # keys whose hashes are the LARGEST 64-bit numbers: every key's home slot is the table's last one, the cluster runs 300 000 slots
        # past it — beyond the 65 536-slot margin (the scan-placed table does not wrap around)
        inv = pow(0x9E3779B97F4A7C15, -1, 1 << 64)
        nb, npb = 300_000, 100_000
        with np.errstate(over="ignore"):
            bk = ((np.uint64(0xFFFFFFFFFFFFFFFF) - np.arange(nb, dtype=np.uint64)) * np.uint64(inv)).astype(np.int64)
        pk = np.concatenate([bk[: npb // 2], rng.integers(0, 1 << 62, npb - npb // 2).astype(np.int64)])
    L, R = [A.HostArray.from_numpy(pk)], [A.HostArray.from_numpy(bk)]
    uniq, cnt = np.unique(bk, return_counts=True)
    pos = np.searchsorted(uniq, pk)
    hit = (pos < len(uniq)) & (uniq[np.minimum(pos, len(uniq) - 1)] == pk)
    inner_rows = int(cnt[pos[hit]].sum())
    for how, rows in (("inner", inner_rows), ("left", inner_rows + int((~hit).sum()))):
        got = {}
        for table in (2, 1):
            lib.set_option("join_table", table)
            try:
                gl, gr = gpu.equijoin_indices(L, R, how)
            finally:
                lib.set_option("join_table", 2)
            assert gl.length == rows, (shape, how, table)
            got[table] = (gl.to_numpy(), gr.to_numpy(), gr.null_count)
        assert np.array_equal(got[2][0], got[1][0]) and got[2][2] == got[1][2], (shape, how)
        m = got[1][1] if how == "inner" else None
        if how == "inner":
            assert np.array_equal(got[2][1], m), shape
            assert np.array_equal(pk[got[2][0]], bk[got[2][1]]), shape          # every pair joins equal keys


@pytest.mark.parametrize("val_dtype", [A.F64, A.I64, A.F32, None])
def test_groupby_partitioned_value_nulls_and_skew(gpu, ora, val_dtype):
    """The single-pass partitioned GROUP BY with NULL values (a group whose values are all NULL still exists, count 0)
    and with skewed keys: a Zipf-like distribution makes one partition dominate, the skew detector switches to the
    scatter variant that combines equal keys inside a super-tile (cnt travels in the record); also forced on uniform keys."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(31337)
    n, ngroups = 400_000, 50_000
    uniform = rng.integers(0, ngroups, n).astype(np.int64)
    zipf = np.minimum(rng.zipf(1.1, n), ngroups).astype(np.int64) * 7919          # a few very hot keys
    hot = np.where(rng.uniform(size=n) < 0.6, 12345, rng.integers(0, ngroups, n)).astype(np.int64)
    for name, kv, force in (("uniform", uniform, 0), ("uniform_forced_combine", uniform, 3), ("zipf", zipf, 0), ("hot_key", hot, 0)):
        lens = [n // 2, n - n // 2]
        keys, vals, pos = [], ([] if val_dtype is not None else None), 0
        for ln in lens:
            keys.append(A.HostArray.from_numpy(kv[pos:pos + ln], valid=rng.uniform(size=ln) >= 0.01, rng=rng))
            if val_dtype is not None:
                v = rng.uniform(-1, 1, ln) if val_dtype in (A.F64, A.F32) else rng.integers(-10 ** 9, 10 ** 9, ln)
                valid = rng.uniform(size=ln) >= 0.3
                valid[kv[pos:pos + ln] % 11 == 0] = False       # whole groups with only NULL values
                vals.append(A.HostArray.from_numpy(v, valid=valid, dtype=val_dtype, rng=rng))
            pos += ln
        exp = _sorted_groups(*ora.groupby_sum(keys, vals, ngroups + 8))
        lib.set_option("gb_debug", force)
        got = _sorted_groups(*gpu.groupby_sum(keys, vals, ngroups + 8))
        lib.set_option("gb_debug", 0)
        k = lib.last_kernel()
        # uniform keys: the second-generation line-aligned scatter; skewed keys overflow one of its fixed-capacity regions and
        # the call falls back to the first-generation histogram + combining scatter
        if name == "uniform":
            assert k.startswith("gb2_scatter_kernel"), k
        else:
            assert k == "gb_aggregate_kernel(combined)", f"{name}: {k}"
        assert np.array_equal(got[1], exp[1]) and np.array_equal(got[0][exp[1]], exp[0][exp[1]]), f"keys {name}"
        assert np.array_equal(got[3], exp[3]), f"counts {name}"
        if val_dtype in (A.F64, A.F32):
            np.testing.assert_allclose(got[2], exp[2], rtol=1e-6, atol=1e-9, err_msg=name)
        else:
            assert np.array_equal(got[2], exp[2]), f"sums {name}"


@pytest.mark.parametrize("how", ["left", "right", "inner", "full"])
def test_equijoin_multi_column_keys(gpu, ora, how):
    """JoinCriteria with several column pairs: rows match on the whole tuple (mixed dtypes per pair, NULL in any key column
    never matches, duplicates on both sides, tuples that agree on one column only); pairs equal the oracle's nested loops."""
    rng = np.random.default_rng(8800)
    for (llens, rlens, nf) in [([9], [7], 0.0), ([700, 0, 900], [1000, 400], 0.08), ([2500], [2000], 0.02)]:
        def side(lens, seed_off):
            cols = [[], [], []]
            for n in lens:
                cols[0].append(A.HostArray.from_numpy(rng.integers(0, 12, n).astype(np.int64), valid=(rng.uniform(size=n) >= nf) if nf else None, offset=seed_off, rng=rng))
                cols[1].append(A.HostArray.from_numpy(rng.integers(-3, 4, n).astype(np.int16), valid=(rng.uniform(size=n) >= nf) if nf else None, rng=rng))
                cols[2].append(A.HostArray.from_numpy(np.round(rng.uniform(0, 2, n), 0), rng=rng))
            return cols
        L, R = side(llens, 3), side(rlens, 5)
        for nk in (2, 3):
            gl, gr = gpu.equijoin_indices_multi(L[:nk], R[:nk], how)
            el, er = ora.equijoin_indices_multi(L[:nk], R[:nk], how)
            assert gl.length == el.length and gl.null_count == el.null_count and gr.null_count == er.null_count, f"{how} nk={nk}"
            assert _pairs(gl, gr) == _pairs(el, er), f"join {how} nk={nk}"
    # one key column through the multi entry point == the single-key entry point
    a = [[A.HostArray.from_numpy(rng.integers(0, 50, 500).astype(np.int32))]]
    b = [[A.HostArray.from_numpy(rng.integers(0, 50, 300).astype(np.int32))]]
    assert _pairs(*gpu.equijoin_indices_multi(a, b, how)) == _pairs(*gpu.equijoin_indices(a[0], b[0], how))


def test_concurrent_callers(gpu, ora):
    """The reference calls its kernels from rayon workers (src/functions/scalar.rs:28,99): the library is re-entrant with
    per-thread state.  Eight threads run different entry points at once (ctypes releases the GIL inside the calls); every
    result must equal the single-threaded oracle answer."""
    import threading
    rng = np.random.default_rng(2024)
    lens = [30_000, 1000, 17]
    jobs = []
    for t in range(8):
        a = make_chunks(rng, A.F64, lens, 0.1, t % 5, kind="unit", nonzero=True)
        b = make_chunks(rng, A.F64, lens, 0.0, 0, kind="unit", nonzero=True)
        k = make_chunks(rng, A.I64, lens, 0.05, 3, kind="plain")
        jobs.append((a, b, k))
    e = A.Expr()
    x, y = e.col(0), e.col(1)
    pred = e.op("gt", x, e.scalar(0.1))
    val = e.op("add", e.op("multiply", x, y), e.scalar(0.5))
    expected = []
    for a, b, k in jobs:
        expected.append((ora.pipeline(e, [a, b], [val], pred)[0], ora.binary("add", a, b), ora.sort_to_indices([k], [False]).to_numpy(),
                         _sorted_groups(*ora.groupby_sum(k, a, 3000))))
    errors = []

    def worker(i):
        try:
            a, b, k = jobs[i]
            for _ in range(5):
                got = gpu.pipeline(e, [a, b], [val], pred)[0]
                exp = expected[i][0]
                assert got.count == exp.count and abs(got.sum - exp.sum) <= 1e-6 * max(abs(exp.sum), 1.0)
                for g, x_ in zip(gpu.binary("add", a, b), expected[i][1]):
                    assert_arrays_match(g, x_, exact=True, what=f"thread {i}")
                assert np.array_equal(gpu.sort_to_indices([k], [False]).to_numpy(), expected[i][2])
                gg = _sorted_groups(*gpu.groupby_sum(k, a, 3000))
                assert np.array_equal(gg[1], expected[i][3][1]) and np.array_equal(gg[3], expected[i][3][3])
                np.testing.assert_allclose(gg[2], expected[i][3][2], rtol=1e-6, atol=1e-9)
        except Exception as ex:   # surfaced in the main thread
            errors.append((i, repr(ex)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
