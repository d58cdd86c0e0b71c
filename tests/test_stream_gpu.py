"""The streamed batch loop: rdf_pipeline over HOST-resident RecordBatches larger than one slab (rdf_capi_stream.inc).

Evaluate::evaluate walks a frame batch by batch (src/evaluation.rs:66-96); the frames come out of readers in host memory
(src/dataframe.rs:349-407).  Above one slab the library cuts the batch list into slabs, uploads slab k + 1 while the kernel runs
over slab k and folds the slabs' partial aggregates in slab order.  Held here to the oracle's unfused batch loop and to the
library's own one-shot (staged whole) path on the same inputs: counts / integer aggregates / extrema bit-exact, f64 sums within
north_star's 1e-6 relative."""
import ctypes as C

import numpy as np
import pytest

from rust_dataframe_amd import _abi as A
from util import make_chunks

pytestmark = pytest.mark.gpu


def _program():
    e = A.Expr()
    x, y, k = e.col(0), e.col(1), e.col(2)
    return e, [x, e.op("multiply", y, e.scalar(2.0, A.F32)), k], e.op("gt", x, e.scalar(-0.3))


def _check(got, exp, what):
    for v, (g, o) in enumerate(zip(got, exp)):
        assert g.count == o.count and g.is_some == o.is_some and g.dtype == o.dtype, (what, v, g, o)
        if g.dtype == A.F32:
            # the reference sums f32 in f32 (sequential left fold, aggregate.rs:82-93); the device folds in f64 and rounds once:
            # both are within the f32 fold's own error bound n * 2^-24 * sum|x| of the exact sum (|x| <= 2 here)
            assert g.min == o.min and g.max == o.max, (what, v, g, o)
            assert abs(g.sum - o.sum) <= g.count * 2.0 ** -24 * 2.0 * g.count + 1e-6 * abs(o.sum), (what, v, g, o)
        elif g.dtype == A.F64:
            assert g.min == o.min and g.max == o.max, (what, v, g, o)
            assert abs(g.sum - o.sum) <= 1e-6 * max(abs(o.sum), 1e-300), (what, v, g, o)      # 1e-6 relative: f64 sums (BASELINE north_star)
        else:
            assert (g.sum, g.min, g.max) == (o.sum, o.min, o.max), (what, v, g, o)


@pytest.mark.parametrize("lens,off,nf", [([1024] * 200 + [576], 0, 0.0), ([300_000, 0, 1024, 77, 150_000], 0, 0.1), ([65_536] * 6, 5, 0.2), ([1_000_000], 3, 0.05)])
def test_streamed_pipeline_matches_oracle_and_one_shot(gpu, ora, lens, off, nf):
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(9 + len(lens))
    cols = [make_chunks(rng, A.F64, lens, nf, off, "unit"), make_chunks(rng, A.F32, lens, 0.0, off, "unit"), make_chunks(rng, A.I64, lens, nf / 2, off, "extreme")]
    e, values, filt = _program()
    exp = ora.pipeline(e, cols, values, filt)
    try:
        lib.set_option("stream_slab_bytes", -1)
        one = gpu.pipeline(e, cols, values, filt)
        assert lib.stream_stats()[0] == 0
        _check(one, exp, "one shot")
        for slab in (256 << 10, 3 << 20):
            lib.set_option("stream_slab_bytes", slab)
            got = gpu.pipeline(e, cols, values, filt)
            slabs, staged, direct = lib.stream_stats()
            total = sum(lens) * 20
            assert slabs >= max(2, total // slab // 2), (slabs, slab, total)
            assert staged > 0 and direct == 0           # numpy's pageable memory goes through the staging buffer
            _check(got, exp, f"streamed slab={slab}")
            _check(got, one, f"streamed vs one shot slab={slab}")
            assert abs(got[1].sum - one[1].sum) <= 1e-9 * 2.0 * got[1].count, "both device folds run in f64"
    finally:
        lib.set_option("stream_slab_bytes", 0)


def test_streamed_pipeline_reads_page_locked_buffers_in_place(gpu, ora):
    """Column buffers in page-locked memory (rdf_host_alloc; rdf_host_register for memory the caller already holds) leave with one
    asynchronous copy each, straight out of the caller's buffer."""
    from rust_dataframe_amd import lib
    L = lib.load()
    n, chunk = 1_500_000, 250_000
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 1, n)
    k = rng.integers(-10 ** 9, 10 ** 9, n).astype(np.int64)
    p = C.c_void_p(0)
    assert L.rdf_host_alloc(C.byref(p), n * 8) == 0
    px = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n,))
    px[:] = x
    assert L.rdf_host_register(C.c_void_p(k.ctypes.data), k.nbytes) == 0
    try:
        xs = [A.HostArray(px, None, i, min(chunk, n - i), A.F64, 0) for i in range(0, n, chunk)]
        ks = [A.HostArray(k, None, i, min(chunk, n - i), A.I64, 0) for i in range(0, n, chunk)]
        e = A.Expr()
        filt = e.op("gt", e.col(0), e.scalar(0.5))
        exp = ora.pipeline(e, [xs, ks], [e.col(0), e.col(1)], filt)
        lib.set_option("stream_slab_bytes", 8 << 20)
        got = gpu.pipeline(e, [xs, ks], [e.col(0), e.col(1)], filt)
        slabs, staged, direct = lib.stream_stats()
        assert slabs >= 3 and staged < (1 << 20) and direct + staged >= 2 * n * 8, (slabs, staged, direct)    # (a slab's tail piece below 256 KiB is packed)
        _check(got, exp, "page-locked")
    finally:
        lib.set_option("stream_slab_bytes", 0)
        L.rdf_host_unregister(C.c_void_p(k.ctypes.data))
        L.rdf_host_free(p)


def test_streamed_pipeline_errors_are_values(gpu):
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(2)
    lens = [50_000] * 4
    a, b = make_chunks(rng, A.F64, lens, 0.0, 0), make_chunks(rng, A.F64, [50_000, 50_000, 49_999, 50_000], 0.0, 0)
    e = A.Expr()
    try:
        lib.set_option("stream_slab_bytes", 256 << 10)
        with pytest.raises(A.RdfError) as ei:
            gpu.pipeline(e, [a, b], [e.op("add", e.col(0), e.col(1))])
        assert ei.value.status == A.RDF_COMPUTE_ERROR and "differ in length" in ei.value.message
        num, den = make_chunks(rng, A.I32, lens, 0.0, 0), make_chunks(rng, A.I32, lens, 0.0, 0, nonzero=True)
        den[2].values[7] = 0
        with pytest.raises(A.RdfError) as ei:                   # a zero divisor in a later slab: DivideByZero, not a crash
            gpu.pipeline(e, [num, den], [e.op("divide", e.col(0), e.col(1))])
        assert ei.value.status == A.RDF_DIVIDE_BY_ZERO
        # the thread's stream buffers are intact afterwards
        got = gpu.pipeline(e, [a], [e.col(0)])
        assert got[0].count == sum(lens) and lib.stream_stats()[0] >= 2
    finally:
        lib.set_option("stream_slab_bytes", 0)


# ---------------------------------------------------------------------------------------------------------------------------
# round 5: the sinks that materialise.  Evaluate::evaluate returns a DataFrame per step (src/evaluation.rs:66-96,
# src/dataframe.rs:178-189): new columns / masks leave the device slab by slab on a third stream while the next slab is computed on.

@pytest.mark.parametrize("lens,off,nf", [([1024] * 150 + [576], 0, 0.0), ([300_000, 0, 1024, 77, 150_000], 0, 0.1), ([65_536] * 5, 5, 0.2), ([700_001], 3, 0.05)])
def test_streamed_store_sinks_match_oracle_and_one_shot(gpu, ora, lens, off, nf):
    from rust_dataframe_amd import lib
    from util import assert_chunks_match
    rng = np.random.default_rng(21 + len(lens))
    a, b = make_chunks(rng, A.F64, lens, nf, off, "unit"), make_chunks(rng, A.F64, lens, nf / 2, off, "unit")
    k = make_chunks(rng, A.I32, lens, nf, off)
    try:
        lib.set_option("stream_slab_bytes", -1)
        one_add = gpu.binary("add", a, b)
        one_sin = gpu.unary("sin", a)
        one_cast = gpu.cast(k, A.F64)
        assert lib.stream_stats()[0] == 0
        for slab in (256 << 10, 2 << 20):
            lib.set_option("stream_slab_bytes", slab)
            for what, run, one, exp in (("add", lambda: gpu.binary("add", a, b), one_add, ora.binary("add", a, b)),
                                        ("sin", lambda: gpu.unary("sin", a), one_sin, ora.unary("sin", a)),
                                        ("cast", lambda: gpu.cast(k, A.F64), one_cast, ora.cast(k, A.F64))):
                got = run()
                in_bytes = sum(lens) * (16 if what == "add" else 8 if what == "sin" else 4)
                assert lib.stream_stats()[0] >= 2 or in_bytes <= slab, (what, slab, lib.stream_stats())
                assert_chunks_match(got, exp, exact=(what != "sin"), what=f"streamed {what} slab={slab} vs oracle")
                assert_chunks_match(got, one, exact=True, what=f"streamed {what} slab={slab} vs one shot")
            # a fused program with a new column as its sink, and a predicate mask (bit-packed values)
            e = A.Expr()
            y = e.op("sin", e.op("add", e.col(0), e.scalar(1.0)))
            outs = [[A.HostArray.empty_out(A.F64, n, nf > 0) for n in lens]]
            got = gpu.pipeline(e, [a], [y], sink=A.SINK_STORE, outs=outs)[0]
            assert lib.stream_stats()[0] >= 2 or sum(lens) * 8 <= slab
            exp = ora.pipeline(e, [a], [y], sink=A.SINK_STORE, outs=[[A.HostArray.empty_out(A.F64, n, nf > 0) for n in lens]])[0]
            assert_chunks_match(got, exp, exact=False, what=f"sin(x + 1) -> column, slab={slab}")
            p = A.Expr()
            pred = p.op("and", p.op("gt", p.col(0), p.scalar(0.25)), p.op("le", p.col(1), p.scalar(0.75)))
            gm, em = gpu.predicate(p, pred, [a, b]), ora.predicate(p, pred, [a, b])
            assert lib.stream_stats()[0] >= 2 or sum(lens) * 16 <= slab
            assert_chunks_match(gm, em, exact=True, what=f"predicate mask slab={slab}")
    finally:
        lib.set_option("stream_slab_bytes", 0)


def test_streamed_store_writes_page_locked_outputs_in_place(gpu, ora):
    from rust_dataframe_amd import lib
    from util import assert_chunks_match
    L = lib.load()
    n, chunk = 1_200_000, 300_000
    rng = np.random.default_rng(8)
    x = rng.uniform(-1, 1, n)
    y = np.empty(n)
    assert L.rdf_host_register(C.c_void_p(x.ctypes.data), x.nbytes) == 0
    assert L.rdf_host_register(C.c_void_p(y.ctypes.data), y.nbytes) == 0
    try:
        xs = [A.HostArray(x, None, i, min(chunk, n - i), A.F64, 0) for i in range(0, n, chunk)]
        outs = [A.HostArray(y[i:i + chunk], None, 0, min(chunk, n - i), A.F64, 0) for i in range(0, n, chunk)]
        lib.set_option("stream_slab_bytes", 4 << 20)
        got = gpu.unary("cos", xs, outs)
        assert lib.stream_stats()[0] >= 2
        exp = ora.unary("cos", xs)
        assert_chunks_match(got, exp, exact=False, what="cos into registered memory")
        assert np.allclose(y, np.cos(x), rtol=1e-12)
    finally:
        lib.set_option("stream_slab_bytes", 0)
        L.rdf_host_unregister(C.c_void_p(x.ctypes.data))
        L.rdf_host_unregister(C.c_void_p(y.ctypes.data))


@pytest.mark.parametrize("lens,nf", [([1024] * 120 + [576], 0.0), ([200_000, 0, 1024, 90_000], 0.1)])
def test_streamed_group_pipeline_matches_oracle_and_one_shot(gpu, ora, lens, nf):
    from rust_dataframe_amd import lib
    from test_group_pipeline import q1_columns, q1_program, check_groups, _cols_list
    rng = np.random.default_rng(33)
    cols = q1_columns(rng, lens, nf, 0)
    e, pred, gid, vals = q1_program()
    exp = ora.group_pipeline(e, _cols_list(cols), vals, gid, 6, pred)
    try:
        lib.set_option("stream_slab_bytes", -1)
        one = gpu.group_pipeline(e, _cols_list(cols), vals, gid, 6, pred)
        check_groups(one, exp, "one shot")
        for slab in (256 << 10, 2 << 20):
            lib.set_option("stream_slab_bytes", slab)
            got = gpu.group_pipeline(e, _cols_list(cols), vals, gid, 6, pred)
            assert lib.stream_stats()[0] >= 2
            check_groups(got, exp, f"streamed Q1 slab={slab}")
            check_groups(got, one, f"streamed Q1 vs one shot slab={slab}")
    finally:
        lib.set_option("stream_slab_bytes", 0)


@pytest.mark.parametrize("lens,off,nf", [([1024] * 100 + [576], 0, 0.0), ([300_000, 0, 1024, 77, 150_000], 0, 0.1), ([65_536] * 5, 5, 0.2), ([900_001], 3, 0.05), ([2_000_000], 0, 0.0)])
def test_streamed_filter_pipeline_matches_oracle(gpu, ora, lens, off, nf):
    """rdf_filter_pipeline = DataFrame::filter over host-resident batches in one streamed call: held to the oracle's
    BooleanFilter::eval_to_array + Column::filter per column (src/dataframe.rs:178-189), bit-exact, for slabs that cut the batch
    list in different places (a long batch is cut inside: its kept rows are appended piece by piece, bitmaps bit by bit)."""
    from rust_dataframe_amd import lib
    from util import assert_chunks_match
    rng = np.random.default_rng(51 + len(lens))
    x, y = make_chunks(rng, A.F64, lens, nf, off, "unit"), make_chunks(rng, A.I64, lens, nf / 2, off, "extreme")
    z = make_chunks(rng, A.I32, lens, 0.0, off)
    e = A.Expr()
    preds = {"one term (evaluated inside the compaction kernel)": e.op("gt", e.col(0), e.scalar(0.1)),
             "arithmetic inside (predicate -> mask -> compaction)": e.op("and", e.op("lt", e.op("multiply", e.col(0), e.scalar(2.0)), e.scalar(1.2)), e.op("ge", e.col(2), e.scalar(0, A.I32)))}
    try:
        for what, pred in preds.items():
            mask = ora.predicate(e, pred, [x, y, z])
            exp = ora.filter_columns([x, y, z], mask)
            for slab in (256 << 10, 3 << 20, 0):
                lib.set_option("stream_slab_bytes", slab)
                got = gpu.filter_pipeline(e, pred, [x, y, z])
                if slab:
                    assert lib.stream_stats()[0] >= 2 or sum(lens) * 20 <= slab, (what, slab, lib.stream_stats())
                for k in range(3):
                    assert_chunks_match(got[k], exp[k], exact=True, what=f"{what}: column {k}, slab={slab}")
    finally:
        lib.set_option("stream_slab_bytes", 0)


def test_streamed_filter_pipeline_errors_are_values(gpu):
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(3)
    a, b = make_chunks(rng, A.F64, [40_000] * 3, 0.0, 0), make_chunks(rng, A.F64, [40_000, 39_999, 40_000], 0.0, 0)
    e = A.Expr()
    pred = e.op("gt", e.col(0), e.scalar(0.0))
    with pytest.raises(A.RdfError) as ei:
        gpu.filter_pipeline(e, pred, [a, b])
    assert ei.value.status == A.RDF_COMPUTE_ERROR and "differ in length" in ei.value.message
    with pytest.raises(A.RdfError) as ei:      # a root that is not boolean
        gpu.filter_pipeline(e, e.col(0), [a])
    assert ei.value.status == A.RDF_INVALID_ARGUMENT
    got = gpu.filter_pipeline(e, pred, [a])     # the thread's buffers are intact afterwards
    assert sum(o.length for o in got[0]) == sum(int((c.to_numpy() > 0).sum()) for c in a)


@pytest.mark.parametrize("agg", ["sum", "min", "max", "count"])
@pytest.mark.parametrize("lens,ngroups,nf,vdt", [([1024] * 90 + [576], 300, 0.0, A.F64), ([150_000, 0, 1024, 60_000], 5000, 0.1, A.I64), ([400_001], 70_000, 0.05, A.F64)])
def test_streamed_groupby_matches_oracle_and_one_shot(gpu, ora, agg, lens, ngroups, nf, vdt):
    """rdf_groupby_agg over host-resident batches beyond one slab: every slab is aggregated on the device, the partial groups are
    merged slab by slab.  Held to the oracle and to the one-shot path: groups, counts and integer aggregates exact, f64 sums 1e-6."""
    from rust_dataframe_amd import lib
    from test_groupby_agg import _groups, _assert_same_groups, _key_chunks
    rng = np.random.default_rng(61 + ngroups)
    keys = _key_chunks(rng, A.I64, lens, ngroups, 0.0, 0, lo=-ngroups // 2)
    vals = make_chunks(rng, vdt, lens, nf, 0, "unit" if vdt == A.F64 else "plain")
    cap = ngroups + 8
    exp = _groups(*ora.groupby_agg([keys], vals, agg, cap))
    try:
        lib.set_option("stream_slab_bytes", -1)
        one = _groups(*gpu.groupby_agg([keys], vals, agg, cap))
        _assert_same_groups(one, exp, vdt == A.F64 and agg != "count", f"one shot {agg}")
        for slab in (256 << 10, 2 << 20):
            lib.set_option("stream_slab_bytes", slab)
            got = _groups(*gpu.groupby_agg([keys], vals, agg, cap))
            assert lib.stream_stats()[0] >= 2, (slab, lib.stream_stats())
            _assert_same_groups(got, exp, vdt == A.F64 and agg != "count", f"streamed {agg} slab={slab}")
        with pytest.raises(A.RdfError) as ei:      # more distinct keys than promised: an error value from whichever slab or merge meets it
            gpu.groupby_agg([keys], vals, agg, max(1, len(exp) // 3))
        assert ei.value.status == A.RDF_MEMORY_ERROR
    finally:
        lib.set_option("stream_slab_bytes", 0)
