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
