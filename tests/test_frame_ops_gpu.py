"""Frame-level operators (rdf_filter_frame / rdf_take_columns / rdf_take_frame / rdf_sort_frame / rdf_groupby_agg_frame) against
the oracle's per-column statements of the same reference functions: DataFrame::filter (src/dataframe.rs:178-189) =
BooleanFilter::eval_to_array + Column::filter per column; DataFrame::take / sort (:194-222) = lexsort_to_indices +
Column::take per column; GroupAggregate (SQL semantics, parity unpinned by the reference).  Inputs live in HBM, a frame goes
in and a frame comes out; the test downloads the result batch by batch and holds it to the oracle bit for bit."""
import numpy as np
import pytest

from rust_dataframe_amd import _abi as A
from util import assert_arrays_match, assert_chunks_match, make_chunks

pytestmark = pytest.mark.gpu


def to_device(cols):
    """The same chunks in device memory (values and bitmaps keep their element / bit offsets) -> (device columns, keep-alive)."""
    import torch
    keep, dev = [], []
    for col in cols:
        dcol = []
        for ch in col:
            vt = torch.from_numpy(np.frombuffer(ch.values.tobytes() + b"\0" * 64, dtype=np.uint8).copy()).cuda()
            bt = None
            if ch.validity is not None:
                bt = torch.from_numpy(np.frombuffer(ch.validity.tobytes() + b"\0" * 64, dtype=np.uint8).copy()).cuda()
            keep += [vt, bt]
            dcol.append(A.DeviceArray(vt.data_ptr(), bt.data_ptr() if bt is not None else None, ch.offset, ch.length, ch.dtype, -1, keep=(vt, bt)))
        dev.append(dcol)
    torch.cuda.synchronize()
    return dev, keep


def to_device_contiguous(cols):
    """The same columns as consecutive slices of ONE device buffer each (what DataFrame::from_csv's zero-copy 1024-row batches of
    one parsed column look like): chunk c starts at row r_c of the buffer, addressed as (pointer to row r_c - r_c % 8, offset r_c % 8)."""
    import torch
    keep, dev = [], []
    for col in cols:
        dt = col[0].dtype
        es = np.dtype(A.NP_OF[dt]).itemsize
        vals = np.concatenate([c.to_numpy() for c in col]) if col else np.zeros(0, A.NP_OF[dt])
        nullable = any(c.validity is not None for c in col)
        vt = torch.from_numpy(np.frombuffer(vals.tobytes() + b"\0" * 64, dtype=np.uint8).copy()).cuda()
        bt = None
        if nullable:
            bits = np.concatenate([c.valid_mask() for c in col])
            bt = torch.from_numpy(np.frombuffer(A.pack_bits(bits).tobytes() + b"\0" * 64, dtype=np.uint8).copy()).cuda()
        keep += [vt, bt]
        dcol, r = [], 0
        for c in col:
            a = r - r % 8
            dcol.append(A.DeviceArray(vt.data_ptr() + a * es, bt.data_ptr() + a // 8 if bt is not None else None, r % 8, c.length, dt, -1, keep=(vt, bt)))
            r += c.length
        dev.append(dcol)
    torch.cuda.synchronize()
    return dev, keep


def frame_columns(frame):
    nc, _, _ = frame.info()
    return [frame.column_to_host(c) for c in range(nc)]


def match_unknown_nulls(got, exp, what):
    """Frames report null counts as unknown: compare lengths, bitmaps and values."""
    assert len(got) == len(exp), what
    for i, (g, e) in enumerate(zip(got, exp)):
        g.null_count = e.null_count if g.validity is not None else 0
        if g.validity is None and e.validity is not None:
            assert e.null_count == 0, f"{what} chunk {i}: bitmap missing"
            e = A.HostArray(e.values, None, e.offset, e.length, e.dtype, 0)
        elif g.validity is not None and e.validity is None:
            assert g.valid_mask().all(), f"{what} chunk {i}"
            g = A.HostArray(g.values, None, g.offset, g.length, g.dtype, 0)
        assert_arrays_match(g, e, exact=True, what=f"{what} chunk {i}")


LAYOUTS = [([1024] * 9 + [100], 0, 0.0), ([5000], 0, 0.1), ([4096, 0, 1000, 64], 0, 0.2), ([3000, 2000, 1], 3, 0.1), ([100] * 40, 0, 0.0)]


@pytest.mark.parametrize("lens,off,nf", LAYOUTS)
@pytest.mark.parametrize("dts", [[A.F64, A.I64, A.F32, A.I32], [A.F64, A.I16, A.U8, A.I64, A.F32], [A.F64]])
def test_filter_frame_parity(gpu, ora, lens, off, nf, dts):
    rng = np.random.default_rng(31 + len(lens) + len(dts))
    host = [make_chunks(rng, dt, lens, nf if k % 2 == 0 else 0.0, off, "unit" if dt in (A.F64, A.F32) else "plain") for k, dt in enumerate(dts)]
    dev, keep = to_device(host)
    e = A.Expr()
    for sel, root in [("half", e.op("gt", e.col(0), e.scalar(0.0))), ("few", e.op("gt", e.col(0), e.scalar(0.9))), ("none", e.op("gt", e.col(0), e.scalar(5.0))),
                      ("all", e.op("le", e.col(0), e.scalar(5.0)))]:
        mask = ora.predicate(e, root, host)
        exp = ora.filter_columns(host, mask)
        with A.PinnedFrame(gpu, dev) as frame:
            out = gpu.filter_frame(frame, e, root)
            first = gpu.pipeline(e, out, [e.col(0)])[0]     # the returned frame's host mirrors are fetched on first need
            nc, nch, rows = out.info()
            assert first.count <= rows
            assert (nc, nch) == (len(dts), len(lens)) and rows == sum(x.length for x in exp[0]), sel
            got = frame_columns(out)
            for k in range(len(dts)):
                match_unknown_nulls(got[k], exp[k], f"{sel} lens={lens} column {k}")
            # the returned frame is an ordinary frame: aggregate it, filter it again
            agg = gpu.pipeline(e, out, [e.col(0)])[0]
            ref = ora.pipeline(e, [exp[0]], [e.col(0)])[0]
            assert agg.count == ref.count and abs(agg.sum - ref.sum) <= 1e-9 * max(1.0, abs(ref.sum)), sel
            again = gpu.filter_frame(out, e, e.op("lt", e.col(0), e.scalar(0.95)))
            m2 = ora.predicate(e, e.op("lt", e.col(0), e.scalar(0.95)), exp)
            exp2 = ora.filter_columns(exp, m2)
            got2 = frame_columns(again)
            for k in range(len(dts)):
                match_unknown_nulls(got2[k], exp2[k], f"{sel} twice column {k}")
            again.release()
            out.release()


def test_filter_frame_errors(gpu):
    rng = np.random.default_rng(5)
    dev, keep = to_device([make_chunks(rng, A.F64, [1000, 24], 0.0, 0)])
    e = A.Expr()
    with A.PinnedFrame(gpu, dev) as frame:
        with pytest.raises(A.RdfError):
            gpu.filter_frame(frame, e, e.col(0))               # not boolean
        with pytest.raises(A.RdfError):
            gpu.filter_frame(frame, A.Expr(), 0)               # empty expression


@pytest.mark.parametrize("lens,off,nf", [([5000], 0, 0.0), ([1024] * 5 + [77], 0, 0.2), ([700, 0, 1300, 64, 1], 5, 0.1)])
@pytest.mark.parametrize("idt", [A.U32, A.U64])
def test_take_columns_parity(gpu, ora, lens, off, nf, idt):
    rng = np.random.default_rng(77 + len(lens))
    dts = [A.F64, A.I64, A.I32, A.F32, A.I16, A.U8, A.U64, A.I8, A.U16, A.F64]     # ten columns: two gather launches
    host = [make_chunks(rng, dt, lens, nf if k % 3 == 0 else 0.0, off) for k, dt in enumerate(dts)]
    total = sum(lens)
    for n, inull in [(0, 0.0), (1, 0.0), (63, 0.0), (1000, 0.3), (3 * total + 5, 0.0)]:
        iv = rng.integers(0, total, n).astype(A.NP_OF[idt])
        idx = A.HostArray.from_numpy(iv, valid=(rng.uniform(size=n) >= inull) if inull else None, dtype=idt, offset=3 if n > 10 else 0, rng=rng)
        got = gpu.take_columns(host, idx)
        for k in range(len(dts)):
            exp = ora.take(host[k], idx)
            assert_arrays_match(got[k], exp, exact=True, what=f"take_columns lens={lens} n={n} column {k}")
            assert_arrays_match(gpu.take(host[k], idx), exp, exact=True, what=f"take lens={lens} n={n} column {k}")
    bad = A.HostArray.from_numpy(np.array([0, total], dtype=A.NP_OF[idt]), dtype=idt)
    with pytest.raises(A.RdfError):
        gpu.take_columns(host, bad)


@pytest.fixture(params=["columns", "row-records"])
def take_path(request):
    """Both gather strategies of rdf_take_frame / rdf_sort_frame: column by column, and through interleaved row records."""
    from rust_dataframe_amd import lib
    lib.set_option("take_rows", 0 if request.param == "columns" else 2)
    yield request.param
    lib.set_option("take_rows", 1)


@pytest.mark.parametrize("contiguous", [False, True])
@pytest.mark.parametrize("lens,nf", [([6000], 0.0), ([1024] * 6 + [300], 0.15), ([512, 3000, 17], 0.0)])
def test_take_frame_and_sort_frame(gpu, ora, lens, nf, contiguous, take_path):
    import torch
    rng = np.random.default_rng(13 + len(lens))
    dts = [A.I64, A.F64, A.I32, A.F32]
    host = [make_chunks(rng, dt, lens, nf if k in (0, 1) else 0.0, 0) for k, dt in enumerate(dts)]
    host[2] = [A.HostArray.from_numpy((c.to_numpy() % 7).astype(np.int32), dtype=A.I32) for c in host[2]]     # many ties for the second criterion
    dev, keep = to_device_contiguous(host) if contiguous else to_device(host)
    total = sum(lens)
    with A.PinnedFrame(gpu, dev) as frame:
        iv = rng.integers(0, total, 2 * total + 3).astype(np.uint32)
        idx = A.HostArray.from_numpy(iv, valid=rng.uniform(size=len(iv)) >= 0.1, dtype=A.U32)
        out = gpu.take_frame(frame, idx)
        assert out.info() == (len(dts), 1, len(iv))
        got = frame_columns(out)
        for k in range(len(dts)):
            match_unknown_nulls(got[k], [ora.take(host[k], idx)], f"take_frame column {k}")
        out.release()
        # indices already in HBM
        it = torch.from_numpy(iv.copy()).cuda()
        didx = A.DeviceArray(it.data_ptr(), None, 0, len(iv), A.U32, 0, keep=it)
        out = gpu.take_frame(frame, didx)
        got = frame_columns(out)
        plain = A.HostArray.from_numpy(iv, dtype=A.U32)
        for k in range(len(dts)):
            match_unknown_nulls(got[k], [ora.take(host[k], plain)], f"take_frame (device indices) column {k}")
        out.release()
        with pytest.raises(A.RdfError):
            gpu.take_frame(frame, A.HostArray.from_numpy(np.array([total], dtype=np.uint32), dtype=A.U32))
        # DataFrame::sort: criteria (column 2 ascending, column 0 descending), then every column taken by the order
        for sort_cols, desc in [([2, 0], [False, True]), ([1], [False]), ([0, 1, 2], [True, False, True])]:
            order = ora.sort_to_indices([host[c] for c in sort_cols], desc)
            ib = torch.zeros(total * 4 + 64, dtype=torch.uint8, device="cuda")
            oi = A.DeviceArray(ib.data_ptr(), None, 0, total, A.U32, 0, keep=ib, capacity=total)
            torch.cuda.synchronize()
            sf, oi = gpu.sort_frame(frame, sort_cols, desc, out_indices=oi)
            assert np.array_equal(ib.cpu().numpy()[:total * 4].view(np.uint32), order.to_numpy()), f"sort order {sort_cols}"
            got = frame_columns(sf)
            for k in range(len(dts)):
                match_unknown_nulls(got[k], [ora.take(host[k], order)], f"sort_frame {sort_cols} column {k}")
            # the sorted frame is an ordinary frame
            e = A.Expr()
            a = gpu.pipeline(e, sf, [e.col(1)])[0]
            r = ora.pipeline(e, host, [e.col(1)])[0]
            assert a.count == r.count and abs(a.sum - r.sum) <= 1e-9 * max(1.0, abs(r.sum))
            sf.release()
            only_idx, _ = gpu.sort_frame(frame, sort_cols, desc, out_indices=oi, want_frame=False)
            assert only_idx is None


@pytest.mark.parametrize("contiguous", [False, True])
@pytest.mark.parametrize("lens,knf,vnf", [([20000], 0.0, 0.0), ([1024] * 11 + [5], 0.1, 0.2), ([3000, 0, 4000], 0.0, 0.3)])
def test_groupby_agg_frame(gpu, ora, lens, knf, vnf, contiguous):
    rng = np.random.default_rng(55 + len(lens))
    total = sum(lens)
    def keycol(card, dt, nf):
        return [A.HostArray.from_numpy(rng.integers(-card // 2, card // 2, n).astype(A.NP_OF[dt]), valid=(rng.uniform(size=n) >= nf) if nf else None, dtype=dt) for n in lens]
    host = [keycol(300, A.I64, knf), keycol(5, A.I32, 0.0), make_chunks(rng, A.F64, lens, vnf, 0, "unit"), make_chunks(rng, A.I64, lens, 0.0, 0)]
    dev, keep = to_device_contiguous(host) if contiguous else to_device(host)

    def table(keys, vals, counts, vvalid=None):
        rows = {}
        for i in range(len(counts)):
            rows[tuple(k[i] for k in keys)] = (vals[i] if vvalid is None or vvalid[i] else None, counts[i])
        return rows

    def host_table(ok, ov, oc):
        keys = [[None if not m else v for v, m in zip(o.to_numpy().tolist(), o.valid_mask().tolist())] for o in ok]
        return table(keys, ov.to_numpy().tolist(), oc.to_numpy().tolist(), ov.valid_mask().tolist())

    with A.PinnedFrame(gpu, dev) as frame:
        for key_cols, vcol, agg in [([0], 2, "sum"), ([0], 2, "min"), ([0], 2, "max"), ([0], -1, "count"), ([0], 3, "sum"), ([1, 0], 2, "sum"), ([0, 1], 3, "max")]:
            exp = host_table(*ora.groupby_agg([host[c] for c in key_cols], host[vcol] if vcol >= 0 else None, agg, total + 2))
            out = gpu.groupby_agg_frame(frame, key_cols, vcol, agg, 4096)
            nc, nch, ng = out.info()
            assert (nc, nch) == (len(key_cols) + 2, 1)
            cols = [c[0] for c in frame_columns(out)]
            got = host_table(cols[:len(key_cols)], cols[len(key_cols)], cols[len(key_cols) + 1])
            assert ng == len(exp) == len(got), (key_cols, agg)
            for k, (v, c) in exp.items():
                gv, gc = got[k]
                assert gc == c, (key_cols, agg, k)
                if v is None or gv is None:
                    assert v is None and gv is None, (key_cols, agg, k)
                elif isinstance(v, float):
                    assert abs(gv - v) <= 1e-9 * max(1.0, abs(v)), (key_cols, agg, k)
                else:
                    assert gv == v, (key_cols, agg, k)
            out.release()
        with pytest.raises(A.RdfError):
            gpu.groupby_agg_frame(frame, [0], 2, "sum", 10)       # more than max_groups distinct keys
        with pytest.raises(A.RdfError):
            gpu.groupby_agg_frame(frame, [2], 3, "sum", 100)      # float grouping column


def test_take_frame_wide_frames_through_row_records(gpu, ora, take_path):
    """20 columns of every width (two record groups of <= 15 columns + the validity slot), nullable and not, null indices."""
    rng = np.random.default_rng(99)
    lens = [1024] * 4 + [500]
    dts = [A.F64, A.I64, A.I32, A.F32, A.I16, A.U8, A.U64, A.I8, A.U16, A.U32] * 2
    host = [make_chunks(rng, dt, lens, 0.2 if k % 4 == 0 else 0.0, 0) for k, dt in enumerate(dts)]
    dev, keep = to_device(host)
    total = sum(lens)
    with A.PinnedFrame(gpu, dev) as frame:
        for n, inull in [(total, 0.0), (777, 0.25), (1, 0.0)]:
            iv = rng.permutation(total)[:n].astype(np.uint32) if n == total else rng.integers(0, total, n).astype(np.uint32)
            idx = A.HostArray.from_numpy(iv, valid=(rng.uniform(size=n) >= inull) if inull else None, dtype=A.U32)
            out = gpu.take_frame(frame, idx)
            got = frame_columns(out)
            for k in range(len(dts)):
                match_unknown_nulls(got[k], [ora.take(host[k], idx)], f"{take_path} n={n} column {k}")
            out.release()
        with pytest.raises(A.RdfError):
            gpu.take_frame(frame, A.HostArray.from_numpy(np.array([3, total + 7], dtype=np.uint32), dtype=A.U32))


@pytest.mark.parametrize("sort_gen", [1, 2, 3])
def test_sort_generations(gpu, ora, sort_gen):
    """The three radix-pass implementations behind rdf_sort_to_indices / rdf_sort_frame (rdf_set_option("sort_gen", ..)):
    first generation, static tile ranges, decoupled look-back — many tiles (the look-back crosses tiles and XCDs), full-range
    and narrow keys, ties, NULLs, several criteria, descending."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(1000 + sort_gen)
    lib.set_option("sort_gen", sort_gen)
    try:
        for n, lens in [(70_001, [70_001]), (200_000, [65_536, 1, 100_000, 34_463])]:
            wide = [A.HostArray.from_numpy(rng.integers(-2 ** 62, 2 ** 62, m), valid=rng.uniform(size=m) >= 0.05, dtype=A.I64) for m in lens]
            ties = [A.HostArray.from_numpy(rng.integers(0, 5, m).astype(np.int32), dtype=A.I32) for m in lens]
            f = [A.HostArray.from_numpy(rng.normal(size=m), dtype=A.F64) for m in lens]
            for cols, desc in [([wide], [False]), ([ties, wide], [True, False]), ([f], [True]), ([ties, f, wide], [False, False, True])]:
                got = gpu.sort_to_indices(cols, desc)
                exp = ora.sort_to_indices(cols, desc)
                assert np.array_equal(got.to_numpy(), exp.to_numpy()), (sort_gen, n, desc)
    finally:
        lib.set_option("sort_gen", 3)


def test_sort_top_bits_then_lds_buckets(gpu, ora):
    """Keys that vary in more than 32 bits: stable passes over the top bits, then every bucket sorted in LDS (os_local_kernel).
    Buckets of every LDS size class (a quarter of the rows crowd an eighth of the key range), ties inside buckets (stability),
    NULLs last (they are moved behind the other rows first), descending, a second criterion; keys crowded on a few top-bit
    patterns (eight patterns of integer keys) go through the byte passes instead.  Doubles: their buckets are cut in VALUE
    space (sign and exponent would crowd the key bits' top patterns) and the whole key is compared inside a bucket — uniform
    values ascending, descending and behind another criterion; a column whose values crowd a few value buckets keeps the byte
    passes (so did any column with an infinity, until the buckets were planned from a sample: below); zeros of both signs, NaNs and denormals order as the oracle's.  The
    A/B switch gives the same order."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(777)
    n = 4_000_000
    kv = rng.integers(-2 ** 61, 2 ** 61, n)
    crowd = rng.uniform(size=n) < 0.25
    kv[crowd] = rng.integers(2 ** 58, 2 ** 58 + 2 ** 59, int(crowd.sum()))
    kv[rng.integers(0, n, n // 50)] = kv[rng.integers(0, n, n // 50)]          # ties
    wide = [A.HostArray.from_numpy(kv, valid=rng.uniform(size=n) >= 0.01, dtype=A.I64)]
    ties = [A.HostArray.from_numpy(rng.integers(0, 3, n).astype(np.int8), dtype=A.I8)]
    fv = rng.normal(size=n) * 1e6
    fv[rng.integers(0, n, 2000)] = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 5e-324, -5e-324, 1e300, -1e300])[rng.integers(0, 10, 2000)]
    fl = [A.HostArray.from_numpy(fv, valid=rng.uniform(size=n) >= 0.01, dtype=A.F64)]                 # doubles of one magnitude: VALUE buckets
    fu = rng.uniform(size=n)
    fu[rng.integers(0, n, 4000)] = np.array([0.0, -0.0, 5e-324, -5e-324, 0.5, 0.5])[rng.integers(0, 6, 4000)]
    funi = [A.HostArray.from_numpy(fu, valid=rng.uniform(size=n) >= 0.01, dtype=A.F64)]
    fnorm = [A.HostArray.from_numpy(rng.normal(size=n) * 1e3 + 7.0, dtype=A.F64)]
    fexp = [A.HostArray.from_numpy(rng.exponential(size=n) ** 6, dtype=A.F64)]                       # nearly all of it in a few value buckets: byte passes
    crowded = [A.HostArray.from_numpy((rng.integers(0, 8, n) << 59) + rng.integers(0, 2 ** 40, n), dtype=A.I64)]
    # round 4: the value buckets are planned from a sample of the keys — a handful of far outliers / infinities / NaNs no longer
    # stretch the bucket map (they collect in the end buckets), heavy tails get more bucket bits, a spike of equal values or a
    # column that keeps concentrating (x^6) is recognised BEFORE any pass is spent and keeps the byte passes
    fo = rng.normal(size=n) * 3.0 + 100.0
    fo[rng.integers(0, n, 24)] = np.array([np.inf, -np.inf, np.nan, -np.nan, 1e300, -1e300, 1e12, -4e9])[rng.integers(0, 8, 24)]
    fout = [A.HostArray.from_numpy(fo, valid=rng.uniform(size=n) >= 0.01, dtype=A.F64)]
    uo = rng.uniform(size=n)
    uo[rng.integers(0, n, 10)] = np.array([np.inf, -1e300, 1e18, np.nan, -np.inf])[rng.integers(0, 5, 10)]
    uout = [A.HostArray.from_numpy(uo, dtype=A.F64)]
    flogn = [A.HostArray.from_numpy(rng.lognormal(size=n), dtype=A.F64)]
    sp = rng.uniform(size=n)
    sp[rng.uniform(size=n) < 0.05] = 0.25
    fspike = [A.HostArray.from_numpy(sp, dtype=A.F64)]
    for cols, desc, local in [([wide], [False], True), ([ties, wide], [False, True], True), ([funi], [False], True), ([funi], [True], True), ([fnorm], [False], True), ([ties, funi], [True, False], True),
                              ([fout], [False], True), ([fout], [True], True), ([uout], [False], True), ([ties, uout], [False, True], True), ([flogn], [False], True),
                              ([fl], [False], None), ([fl], [True], None), ([fexp], [False], False), ([fspike], [False], False), ([crowded], [False], False)]:
        exp = ora.sort_to_indices(cols, desc).to_numpy()
        got = gpu.sort_to_indices(cols, desc).to_numpy()
        if local is not None:     # (fl: ~800 non-finite rows of 4e6 — whether its end buckets are trusted with them depends on how many the sample met)
            assert ("os_local_kernel" in lib.last_kernel()) == local, lib.last_kernel()
        assert np.array_equal(got, exp), (desc, local)
        lib.set_option("sort_sample", 0)        # A/B: buckets over [min, max] (round 3)
        try:
            got1 = gpu.sort_to_indices(cols, desc).to_numpy()
        finally:
            lib.set_option("sort_sample", 1)
        assert np.array_equal(got1, exp), (desc, "buckets over [min, max]")
        lib.set_option("sort_msd", 0)
        try:
            got0 = gpu.sort_to_indices(cols, desc).to_numpy()
            assert "os_local_kernel" not in lib.last_kernel()
        finally:
            lib.set_option("sort_msd", 1)
        assert np.array_equal(got0, exp), (desc, "byte passes")


def test_filter_frame_wider_than_a_program(gpu, ora):
    """A frame of 12 columns (lineitem has 16): the predicate reads columns 9 and 2 — a projection onto the columns it reads runs
    the fused predicate, the compaction takes all twelve (two launches of <= 8 columns)."""
    rng = np.random.default_rng(314)
    lens = [1024] * 5 + [333]
    dts = [A.F64, A.I64, A.F64, A.I32, A.F32, A.I16, A.U8, A.I64, A.F64, A.F64, A.U16, A.I8]
    host = [make_chunks(rng, dt, lens, 0.1 if k in (2, 7) else 0.0, 0, "unit" if dt in (A.F64, A.F32) else "plain") for k, dt in enumerate(dts)]
    dev, keep = to_device(host)
    e = A.Expr()
    root = e.op("and", e.op("gt", e.col(9), e.scalar(-0.2)), e.op("lt", e.col(2), e.scalar(0.7)))
    exp = ora.filter_columns(host, ora.predicate(e, root, host))
    with A.PinnedFrame(gpu, dev) as frame:
        for _ in range(2):     # the projection is cached in the frame
            out = gpu.filter_frame(frame, e, root)
            got = frame_columns(out)
            for k in range(len(dts)):
                match_unknown_nulls(got[k], exp[k], f"wide frame column {k}")
            out.release()


@pytest.mark.parametrize("lens,off,nf", [([1024] * 300 + [500], 0, 0.0), ([1024] * 40, 0, 0.15), ([200_000], 0, 0.1), ([70_000, 1024, 3000, 0, 50_000], 0, 0.0),
                                         ([5000, 4000], 5, 0.1)])
def test_filter_frame_one_pass_predicates(gpu, ora, lens, off, nf):
    """`column CMP literal [AND | OR column CMP literal]` predicates run inside the compaction kernel (ffilter_dma_kernel: no mask
    written, counted or read back; tiles of a batch longer than 1024 rows find their offsets by look-back).  Held to the oracle's
    BooleanFilter::eval_to_array + Column::filter per column AND to the three-pass device path, bit for bit: comparisons in f64
    (src/expression.rs:844-845), a NULL on either side of and / or drops the row, NaN compares false."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(404 + len(lens))
    dts = [A.F64, A.I64, A.F32, A.U64, A.I32]
    host = [make_chunks(rng, dt, lens, nf if k in (0, 1, 2) else 0.0, off, "special" if dt == A.F64 and sum(lens) > 5000 else "unit" if dt in (A.F64, A.F32) else "plain")
            for k, dt in enumerate(dts)]
    for ch in host[3]:                                   # UInt64 values beyond 2^63: compared as f64, not as i64
        ch.values[ch.offset:ch.offset + ch.length:3] += np.uint64(2 ** 63)
    dev, keep = to_device(host)
    e = A.Expr()
    preds = {
        "literal first": e.op("lt", e.scalar(0.25), e.col(0)),
        "i64 >= int literal": e.op("ge", e.col(1), e.scalar(17, A.I64)),
        "i64 != float literal": e.op("ne", e.col(1), e.scalar(-3.0)),
        "f32 <= f32 literal": e.op("le", e.col(2), e.scalar(0.1, A.F32)),
        "u64 > 2^63": e.op("gt", e.col(3), e.scalar(float(2 ** 63))),
        "range on one column": e.op("and", e.op("gt", e.col(0), e.scalar(-0.5)), e.op("lt", e.col(0), e.scalar(0.5))),
        "and over two nullable columns": e.op("and", e.op("gt", e.col(0), e.scalar(0.0)), e.op("le", e.col(1), e.scalar(500, A.I32))),
        "or over two nullable columns": e.op("or", e.op("gt", e.col(2), e.scalar(0.9)), e.op("lt", e.col(1), e.scalar(-900, A.I64))),
        "eq keeps almost nothing": e.op("eq", e.col(4), e.scalar(7, A.I32)),
    }
    seen_kernels = set()
    with A.PinnedFrame(gpu, dev) as frame:
        for name, root in preds.items():
            exp = ora.filter_columns(host, ora.predicate(e, root, host))
            try:
                # fused 2: the one-pass kernels whatever the batch lengths; 1: the default choice; 0: three passes.  mixed 1 (round 6): a frame
                # of 8- and 4-byte columns takes the block kernel twice over the same tiles when the predicate's columns are of one width
                # and the batches fill its slots / tiles — the second launch by the mask the first one wrote (2: wherever those forms
                # apply; 1, the default: only where the wave-tile kernel's 1024-row tiles come out partial); 0: the wave-tile kernel
                for fused, mixed in ((2, 2), (2, 0), (1, 2), (1, 1), (0, 2)):
                    lib.set_option("filter_fused", fused)
                    lib.set_option("filter_mixed", mixed)
                    out = gpu.filter_frame(frame, e, root)
                    kern = lib.last_kernel()
                    one_pass = kern == "ffilter_dma_kernel" or kern.startswith("bfilter_kernel x 2")
                    if not mixed:
                        assert kern == "ffilter_dma_kernel", (name, kern)
                    elif fused != 1:
                        assert one_pass == bool(fused), (name, kern)
                    elif max(lens) <= 65536:
                        assert one_pass, (name, kern)
                    seen_kernels.add(kern)
                    nc, nch, rows = out.info()
                    assert (nc, nch) == (len(dts), len(lens)) and rows == sum(x.length for x in exp[0]), (name, fused)
                    got = frame_columns(out)
                    for k in range(len(dts)):
                        match_unknown_nulls(got[k], exp[k], f"{name} fused={fused} lens={lens} column {k}")
                    # the new frame is an ordinary frame: filter it again (its batches sit at the input's stride)
                    again = gpu.filter_frame(out, e, e.op("gt", e.col(4), e.scalar(0, A.I32)))
                    exp2 = ora.filter_columns(exp, ora.predicate(e, e.op("gt", e.col(4), e.scalar(0, A.I32)), exp))
                    got2 = frame_columns(again)
                    for k in range(len(dts)):
                        match_unknown_nulls(got2[k], exp2[k], f"{name} fused={fused} twice column {k}")
                    again.release()
                    out.release()
            finally:
                lib.set_option("filter_fused", 1)
                lib.set_option("filter_mixed", 1)
    if sum(lens) >= len(lens) * 768:                 # (layouts the one-pass paths take at all) the two-launch form ran for some predicate
        assert any(k.startswith("bfilter_kernel x 2") for k in seen_kernels), seen_kernels


@pytest.mark.parametrize("lens,nf", [([3_000_000, 1024, 200_000], 0.0), ([1_500_000, 700_001], 0.1)])
def test_filter_frame_one_pass_on_long_batches(gpu, ora, lens, nf):
    """Batches far longer than a super-tile of 64 tiles: the offsets inside a batch come from the two-level look-back — round 5: the
    rows in front of a super-tile are found once per super-tile by its first tile (`filter_lookback` 3, the default: from the nearest
    super-tiles' tile counts and the older ones' totals; 2: from totals only; any batch length), round 4: by every tile (1).  All
    against the oracle and the three-pass path, bit for bit."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(77)
    dts = [A.F64, A.I64]
    host = [make_chunks(rng, dt, lens, nf, 0, "unit" if dt == A.F64 else "plain") for dt in dts]
    dev, keep = to_device(host)
    e = A.Expr()
    preds = {"one term": e.op("gt", e.col(0), e.scalar(0.3)),
             "two columns": e.op("or", e.op("lt", e.col(0), e.scalar(0.05)), e.op("gt", e.col(1), e.scalar(0, A.I64)))}
    with A.PinnedFrame(gpu, dev) as frame:
        try:
            for name, root in preds.items():
                exp = ora.filter_columns(host, ora.predicate(e, root, host))
                for fused, lookback, block in ((1, 3, 1), (1, 3, 0), (1, 2, 0), (2, 1, 0), (0, 3, 1)):
                    lib.set_option("filter_fused", fused)
                    lib.set_option("filter_lookback", lookback)
                    lib.set_option("filter_block", block)      # round 6: block tiles + scanner wave (the default for batches this long)
                    out = gpu.filter_frame(frame, e, root)
                    want_kernel = "bfilter_kernel" if fused and block else "ffilter_dma_kernel" if fused else None
                    assert want_kernel is None or lib.last_kernel() == want_kernel, (name, fused, lookback, block, lib.last_kernel())
                    assert fused or lib.last_kernel() not in ("bfilter_kernel", "ffilter_dma_kernel")
                    got = frame_columns(out)
                    for k in range(len(dts)):
                        match_unknown_nulls(got[k], exp[k], f"{name} fused={fused} lookback={lookback} block={block} column {k}")
                    out.release()
        finally:
            lib.set_option("filter_fused", 1)
            lib.set_option("filter_lookback", 3)
            lib.set_option("filter_block", 1)


BLOCK_LAYOUTS = [([200_000], 0, 0.1), ([70_000, 1024, 3000, 0, 50_000], 0, 0.0), ([8192 * 3], 0, 0.2), ([8191, 8193, 16384 + 7, 1], 5, 0.1),
                 ([1_500_000, 700_001], 3, 0.05), ([40_000] * 7, 1, 0.0)]


@pytest.mark.parametrize("lens,off,nf", BLOCK_LAYOUTS)
@pytest.mark.parametrize("dts", [[A.F64], [A.I64, A.F64, A.U64], [A.F32, A.I32, A.U32], [A.U32], [A.I64] * 8])
def test_filter_frame_block_tiles(gpu, ora, lens, off, nf, dts):
    """Round 6: DataFrame::filter of long batches on BLOCK tiles held in registers (rdf_bfilter.hip) — counts published one iteration
    ahead, prefixes from the scanner wave.  `filter_block_rows` = 1 sends every layout there: batches that are not multiples of a
    tile, a batch shorter than a tile, an empty batch, slices that are not 16-byte aligned (element-wise loads), validity bitmaps
    at odd bit offsets, 8- and 4-byte columns, up to 8 columns.  Integer columns are compared through the interval of x on which
    `(double)x CMP literal` holds (no conversion on the device): the literals sit where that matters (beyond 2^53, at the type's
    ends, fractional, NaN).  Held to the oracle's eval_to_array + Column::filter per column, bit for bit."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(6006 + len(lens) + len(dts))
    host = []
    for k, dt in enumerate(dts):
        kind = "special" if dt in (A.F64, A.F32) and k == 0 and sum(lens) > 100_000 else "unit" if dt in (A.F64, A.F32) else "extreme" if k % 2 == 0 else "plain"
        host.append(make_chunks(rng, dt, lens, nf if k < 3 else 0.0, off, kind))
    dev, keep = to_device(host)
    e = A.Expr()
    d0 = dts[0]
    is_f = d0 in (A.F64, A.F32)
    wide = d0 in (A.I64, A.U64)
    preds = {}
    if is_f:
        preds["gt"] = e.op("gt", e.col(0), e.scalar(0.25))
        preds["le literal first"] = e.op("ge", e.scalar(-0.5), e.col(0))
        preds["ne"] = e.op("ne", e.col(0), e.scalar(1.0))
        preds["nan literal"] = e.op("lt", e.col(0), e.scalar(float("nan")))
        preds["keeps almost nothing"] = e.op("gt", e.col(0), e.scalar(0.999))
        preds["keeps everything"] = e.op("le", e.col(0), e.scalar(float("inf")))
    else:
        big = float(2 ** 62) if wide else float(2 ** 30)
        preds["gt 0"] = e.op("gt", e.col(0), e.scalar(0, A.I64))
        preds["ge fractional"] = e.op("ge", e.col(0), e.scalar(-0.5))
        preds["lt beyond 2^53"] = e.op("lt", e.col(0), e.scalar(big + 1.0))
        preds["le at the rounding edge"] = e.op("le", e.col(0), e.scalar(float(2 ** 53 + 2) if wide else float(2 ** 24 + 1)))
        preds["eq"] = e.op("eq", e.col(0), e.scalar(float(np.iinfo(A.NP_OF[d0]).max)))
        preds["ne"] = e.op("ne", e.col(0), e.scalar(float(np.iinfo(A.NP_OF[d0]).min)))
        preds["nan literal ne"] = e.op("ne", e.col(0), e.scalar(float("nan")))
        preds["keeps almost nothing"] = e.op("gt", e.col(0), e.scalar(big))
    if len(dts) > 1:
        preds["and over two columns"] = e.op("and", preds[next(iter(preds))], e.op("lt", e.col(1), e.scalar(0.3)))
        preds["or over two columns"] = e.op("or", e.op("gt", e.col(len(dts) - 1), e.scalar(0.9)), e.op("le", e.col(1), e.scalar(-0.7)))
        preds["range on the last column"] = e.op("and", e.op("gt", e.col(len(dts) - 1), e.scalar(-0.5)), e.op("lt", e.col(len(dts) - 1), e.scalar(500.0)))
    with A.PinnedFrame(gpu, dev) as frame:
        try:
            lib.set_option("filter_block_rows", 1)
            for name, root in preds.items():
                exp = ora.filter_columns(host, ora.predicate(e, root, host))
                # offsets from the scanner wave (few long batches), then a block per batch with its own running offset (many batches,
                # none a large share of the frame; `filter_owned` 2 takes that form whatever the layout)
                for owned, kern in ((0, "bfilter_kernel"), (2, "bfilter_kernel (a block per batch)")):
                    lib.set_option("filter_owned", owned)
                    out = gpu.filter_frame(frame, e, root)
                    assert lib.last_kernel() == kern, (name, lib.last_kernel())
                    nc, nch, rows = out.info()
                    assert (nc, nch) == (len(dts), len(lens)) and rows == sum(x.length for x in exp[0]), (name, owned, rows)
                    got = frame_columns(out)
                    for k in range(len(dts)):
                        match_unknown_nulls(got[k], exp[k], f"{name} owned={owned} lens={lens} column {k}")
                    out.release()
        finally:
            lib.set_option("filter_block_rows", 8192)
            lib.set_option("filter_owned", 1)


SHORT_LAYOUTS = [([1024] * 37 + [500], 0, 0.1), ([1024, 1000, 0, 1, 1023, 1024, 512, 777, 1024, 1024, 1024, 1024], 5, 0.1), ([4096] * 5 + [3000], 0, 0.0),
                 ([4000, 3900, 4096, 2100], 3, 0.2), ([8192, 8000, 5000], 0, 0.1), ([3000], 0, 0.1), ([1024] * 300, 0, 0.0), ([2048] * 9 + [0, 2047], 2, 0.0),
                 ([1000] * 40, 0, 0.1), ([1500] * 21 + [7], 0, 0.0), ([5000] * 6, 0, 0.1), ([1001, 999, 1000, 3, 1000, 1000], 1, 0.0)]


def block_form_expected(lens, ncols, frame=True):
    """The host's choice (filter_frame_fused / rdf_filter_columns) -> "short" | "long" | None: slots filled to 0.9 take the short form,
    batches that average half a tile and fill their tiles to 0.5 (one column) / 0.65 the long forms, half-full slots the short form."""
    n, total = len(lens), sum(lens)
    if total == 0 or (frame and total < n * 768):
        return None
    tr = 8192 if ncols == 1 else 4096
    mean = -(-total // n)
    if mean >= 8192:
        return "long"
    wr, fill = tr // 8, 0.0
    for sh in range(4):
        if max(lens) <= wr << sh:
            fill = total / (n * (wr << sh))
            break
    if fill >= 0.9:
        return "short"
    if mean * 2 >= tr:
        ntl = sum(-(-x // tr) for x in lens)
        if total / (ntl * tr) >= (0.5 if ncols == 1 else 0.65):
            return "long"
    return "short" if fill >= 0.5 else None


def short_mode_expected(lens, ncols, frame=True):
    return block_form_expected(lens, ncols, frame) == "short"


@pytest.mark.parametrize("lens,off,nf", SHORT_LAYOUTS)
@pytest.mark.parametrize("dts", [[A.F64], [A.I64, A.F64, A.U64], [A.F32, A.I32, A.U32], [A.U32], [A.I64] * 8])
def test_filter_frame_short_batches_block_kernel(gpu, ora, lens, off, nf, dts):
    """Round 6: DataFrame::filter of batches no longer than a block tile — the readers' 1024-row RecordBatches, batches of a few
    thousand rows — on the block kernel's short-batch mode (rdf_bfilter.hip, SHORT): a batch takes 1 / 2 / 4 / 8 waves of one block,
    its kept rows start its own output, nothing is waited for.  Full, ragged, empty and one-row batches, a frame of ONE batch, tiles
    whose last slots lie past the frame's end, slices that are not 16-byte aligned, validity bitmaps at odd bit offsets, 8- and 4-byte
    columns, 1 / 3 / 8 columns, one- and two-term predicates.  Held to the oracle's eval_to_array + Column::filter per column AND to
    the wave-tile kernel (`filter_short` 0), bit for bit."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(6300 + len(lens) + len(dts))
    host = []
    for k, dt in enumerate(dts):
        kind = "unit" if dt in (A.F64, A.F32) else "extreme" if k % 2 == 0 else "plain"
        host.append(make_chunks(rng, dt, lens, nf if k < 3 else 0.0, off, kind))
    dev, keep = to_device(host)
    e = A.Expr()
    is_f = dts[0] in (A.F64, A.F32)
    preds = {}
    if is_f:
        preds["gt"] = e.op("gt", e.col(0), e.scalar(0.25))
        preds["keeps almost nothing"] = e.op("gt", e.col(0), e.scalar(0.999))
        preds["keeps everything"] = e.op("le", e.col(0), e.scalar(float("inf")))
    else:
        preds["gt 0"] = e.op("gt", e.col(0), e.scalar(0, A.I64))
        preds["ge fractional"] = e.op("ge", e.col(0), e.scalar(-0.5))
        preds["ne"] = e.op("ne", e.col(0), e.scalar(float(np.iinfo(A.NP_OF[dts[0]]).min)))
    if len(dts) > 1:
        preds["and over two columns"] = e.op("and", preds[next(iter(preds))], e.op("lt", e.col(1), e.scalar(0.3)))
        preds["or over two columns"] = e.op("or", e.op("gt", e.col(len(dts) - 1), e.scalar(0.9)), e.op("le", e.col(1), e.scalar(-0.7)))
    form = block_form_expected(lens, len(dts))
    expect_short = form == "short"
    with A.PinnedFrame(gpu, dev) as frame:
        try:
            for name, root in preds.items():
                exp = ora.filter_columns(host, ora.predicate(e, root, host))
                for short in (1, 0):
                    lib.set_option("filter_short", short)
                    out = gpu.filter_frame(frame, e, root)
                    if short and expect_short:
                        assert lib.last_kernel() == "bfilter_kernel (short batches)", (name, lib.last_kernel())
                    elif form == "long":
                        assert lib.last_kernel().startswith("bfilter_kernel") and "short" not in lib.last_kernel(), (name, lib.last_kernel())
                    else:
                        assert lib.last_kernel() != "bfilter_kernel (short batches)", (name, lib.last_kernel())
                    nc, nch, rows = out.info()
                    assert (nc, nch) == (len(dts), len(lens)) and rows == sum(x.length for x in exp[0]), (name, short, rows)
                    got = frame_columns(out)
                    for k in range(len(dts)):
                        match_unknown_nulls(got[k], exp[k], f"{name} short={short} lens={lens} column {k}")
                    out.release()
        finally:
            lib.set_option("filter_short", 1)


@pytest.mark.parametrize("lens,off,nf", BLOCK_LAYOUTS)
@pytest.mark.parametrize("dts", [[A.F64], [A.I64, A.F64], [A.F32, A.I32, A.U32], [A.I64] * 11])
def test_filter_columns_device_block_tiles(gpu, ora, lens, off, nf, dts):
    """Column::filter (src/table.rs:97-107,213-215) with the mask GIVEN, device-resident, outputs that can hold every row: one pass on
    block tiles (rdf_bfilter.hip, mask form) — no count pass, lengths and null counts come back from the kernel.  Nullable masks at
    odd bit offsets, nullable columns, 11 columns (a launch of 8 and one of 3), 4-byte columns; a second call with outputs sized by
    rdf_filter_count (a two-phase caller) takes the same one pass — the kernel knows every output's capacity — and must give the same
    bytes; a third with an output one row too small fails with MemoryError and writes nothing past a buffer."""
    import torch
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(6100 + len(lens) + len(dts))
    host = [make_chunks(rng, dt, lens, nf if k % 2 == 0 else 0.0, (off + k) % 7, "extreme" if dt not in (A.F64, A.F32) else "unit") for k, dt in enumerate(dts)]
    for sel in (0.5, 0.01, 1.0):
        mask = [A.HostArray.from_numpy(rng.uniform(size=n) < sel, valid=(rng.uniform(size=n) > 0.1) if nf else None, offset=(off * 3) % 11, dtype=A.BOOL, rng=rng)
                for n in lens]
        exp = ora.filter_columns(host, mask)
        dev, keep = to_device(host)
        dmask, keep2 = to_device([mask])

        def outputs(rows):
            bufs, outs = [], []
            for k, dt in enumerate(dts):
                es = np.dtype(A.NP_OF[dt]).itemsize
                col = []
                for c, n in enumerate(rows):
                    # every other chunk's values start one element past a 16-byte boundary: the stores' scalar head / tail
                    skew = es if (k + c) % 2 else 0
                    vb0 = torch.full((n * es + 96,), 0xAB, dtype=torch.uint8, device="cuda")
                    vb = vb0[skew:]
                    bb = torch.full(((n + 63) // 64 * 8 + 64,), 0xAB, dtype=torch.uint8, device="cuda") if host[k][c].validity is not None else None
                    bufs.append((vb, bb))
                    col.append(A.DeviceArray(vb0.data_ptr() + skew, bb.data_ptr() if bb is not None else None, 0, 0, dt, 0, keep=(vb0, bb), capacity=n))
                outs.append(col)
            torch.cuda.synchronize()
            return bufs, outs

        def check(bufs, outs, what):
            lib.synchronize()
            i = 0
            for k, dt in enumerate(dts):
                es = np.dtype(A.NP_OF[dt]).itemsize
                for c in range(len(lens)):
                    ee, o, (vb, bb) = exp[k][c], outs[k][c], bufs[i]
                    i += 1
                    assert o.length == ee.length and o.null_count == ee.null_count, (what, k, c, o.length, ee.length, o.null_count, ee.null_count)
                    gv = vb.cpu().numpy()[:ee.length * es].view(A.NP_OF[dt])
                    m = ee.valid_mask()
                    if bb is not None:
                        gm = np.unpackbits(bb.cpu().numpy()[:(ee.length + 7) // 8], bitorder="little")[:ee.length].astype(bool)
                        assert np.array_equal(gm, m), (what, k, c)
                    assert np.array_equal(gv[m].view(np.uint8), ee.to_numpy()[m].view(np.uint8)), (what, k, c)

        try:
            lib.set_option("filter_block_rows", 1)
            lib.set_option("filter_owned", 2)               # a block per batch, its own running offset
            bufs, outs = outputs(lens)
            gpu.filter_columns(dev, dmask[0], outs)
            assert lib.last_kernel() == "bfilter_kernel (a block per batch)", lib.last_kernel()
            check(bufs, outs, f"one pass, a block per batch, sel={sel}")
            lib.set_option("filter_owned", 0)               # tiles by ticket, offsets from the scanner wave
            bufs, outs = outputs(lens)
            gpu.filter_columns(dev, dmask[0], outs)
            assert lib.last_kernel() == "bfilter_kernel", lib.last_kernel()
            check(bufs, outs, f"one pass sel={sel}")
            counts = gpu.filter_count(dmask[0])
            assert counts == [x.length for x in exp[0]]
            if any(c < n for c, n in zip(counts, lens)):
                bufs, outs = outputs(counts)               # a two-phase caller's outputs, sized by rdf_filter_count: the same one pass, every output's capacity known to the kernel
                gpu.filter_columns(dev, dmask[0], outs)
                assert lib.last_kernel() == "bfilter_kernel", lib.last_kernel()
                check(bufs, outs, f"counted sel={sel}")
                short = [max(c - 1, 0) if i == len(counts) - 1 or c > 5 else c for i, c in enumerate(counts)]
                if short != counts:                        # an output one row too small: the chunk is not written, the call fails
                    bufs, outs = outputs(short)
                    with pytest.raises(A.RdfError) as ei:
                        gpu.filter_columns(dev, dmask[0], outs)
                    assert ei.value.status == A.RDF_MEMORY_ERROR, ei.value
                    lib.synchronize()
                    i = 0
                    for k, dt in enumerate(dts):             # nothing was written past any output's capacity
                        es = np.dtype(A.NP_OF[dt]).itemsize
                        for c in range(len(lens)):
                            vb, _ = bufs[i]
                            i += 1
                            guard = vb.cpu().numpy()[short[c] * es: short[c] * es + 32]
                            assert (guard == 0xAB).all(), (k, c)
        finally:
            lib.set_option("filter_block_rows", 8192)
            lib.set_option("filter_owned", 1)


@pytest.mark.parametrize("lens,off,nf", [([1024] * 40 + [500], 0, 0.1), ([1024, 0, 1024, 777, 1024], 3, 0.0), ([1000] * 9, 0, 0.2),
                                         ([4096] * 5 + [3000], 0, 0.1), ([2048] * 9 + [0, 2047], 2, 0.0), ([8192, 8000, 5000], 0, 0.1), ([3000], 1, 0.1)])
@pytest.mark.parametrize("dts", [[A.F64], [A.I64, A.F64, A.U64], [A.F32, A.I32], [A.I64] * 11])
def test_filter_columns_device_one_pass_reader_batches(gpu, ora, lens, off, nf, dts):
    """Column::filter over the readers' 1024-row batches and over chunks of a few thousand rows, device-resident, outputs that can
    hold every row: no count pass, no scan — a chunk's kept rows start its output.  Round 6, first the wave-tile LDS-DMA kernel
    (a chunk is ONE wave tile; `filter_short` 0), then the block kernel's short-batch mode (a chunk on 1 / 2 / 4 / 8 waves of a block,
    chunks of up to a block tile; the default).  Same oracle, same bytes from both and from the counted path."""
    import torch
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(6200 + len(lens) + len(dts))
    host = [make_chunks(rng, dt, lens, nf if k % 2 == 0 else 0.0, (off + k) % 7, "extreme" if dt not in (A.F64, A.F32) else "unit") for k, dt in enumerate(dts)]
    for sel in (0.5, 0.02, 1.0):
        mask = [A.HostArray.from_numpy(rng.uniform(size=n) < sel, valid=(rng.uniform(size=n) > 0.1) if nf else None, offset=(off * 3) % 11, dtype=A.BOOL, rng=rng)
                for n in lens]
        exp = ora.filter_columns(host, mask)
        dev, keep = to_device(host)
        dmask, keep2 = to_device([mask])
        bufs, outs = [], []
        for k, dt in enumerate(dts):
            es = np.dtype(A.NP_OF[dt]).itemsize
            col = []
            for c, n in enumerate(lens):
                vb = torch.full((n * es + 64,), 0xAB, dtype=torch.uint8, device="cuda")
                bb = torch.full(((n + 63) // 64 * 8 + 64,), 0xAB, dtype=torch.uint8, device="cuda") if host[k][c].validity is not None else None
                bufs.append((vb, bb))
                col.append(A.DeviceArray(vb.data_ptr(), bb.data_ptr() if bb is not None else None, 0, 0, dt, 0, keep=(vb, bb), capacity=n))
            outs.append(col)
        torch.cuda.synchronize()
        try:
            for short in (1, 0):                          # the block kernel's short-batch mode (the default), the wave-tile LDS-DMA kernel
                lib.set_option("filter_short", short)
                for vb, bb in bufs:
                    vb.fill_(0xAB)
                    if bb is not None:
                        bb.fill_(0xAB)
                torch.cuda.synchronize()
                gpu.filter_columns(dev, dmask[0], outs)
                form = block_form_expected(lens, len(dts), frame=False)
                if short and form == "short":
                    assert lib.last_kernel() == "bfilter_kernel (short batches)", lib.last_kernel()
                elif form == "long":
                    assert lib.last_kernel().startswith("bfilter_kernel") and "short" not in lib.last_kernel(), lib.last_kernel()
                elif max(lens) <= 1024 and sum(lens) >= len(lens) * 768:
                    assert lib.last_kernel() == "fcompact_dma_kernel (one pass)", lib.last_kernel()
                else:
                    assert lib.last_kernel() != "bfilter_kernel (short batches)", lib.last_kernel()
                lib.synchronize()
                i = 0
                for k, dt in enumerate(dts):
                    es = np.dtype(A.NP_OF[dt]).itemsize
                    for c in range(len(lens)):
                        ee, o, (vb, bb) = exp[k][c], outs[k][c], bufs[i]
                        i += 1
                        assert o.length == ee.length and o.null_count == ee.null_count, (short, sel, k, c, o.length, ee.length, o.null_count, ee.null_count)
                        gv = vb.cpu().numpy()[:ee.length * es].view(A.NP_OF[dt])
                        m = ee.valid_mask()
                        if bb is not None:
                            gm = np.unpackbits(bb.cpu().numpy()[:(ee.length + 7) // 8], bitorder="little")[:ee.length].astype(bool)
                            assert np.array_equal(gm, m), (short, sel, k, c)
                        assert np.array_equal(gv[m].view(np.uint8), ee.to_numpy()[m].view(np.uint8)), (short, sel, k, c)
                        assert (vb.cpu().numpy()[ee.length * es + 16:] == 0xAB).all(), ("written past the kept rows", short, sel, k, c)
        finally:
            lib.set_option("filter_short", 1)


def test_filter_block_tiles_stuck_prefix_fails_the_call_not_the_gpu(gpu, ora):
    """The block-tile kernel's tiles wait for a word another wave writes.  If that word never comes (here: `filter_block` 3 keeps
    the scanner wave idle) every wait gives up after a few seconds, the call returns a device error, and the next call works."""
    from rust_dataframe_amd import lib
    rng = np.random.default_rng(99)
    host = [make_chunks(rng, A.F64, [100_000], 0.0, 0, "unit")]
    dev, keep = to_device(host)
    e = A.Expr()
    root = e.op("gt", e.col(0), e.scalar(0.0))
    with A.PinnedFrame(gpu, dev) as frame:
        try:
            lib.set_option("filter_block", 3)
            with pytest.raises(A.RdfError) as ei:
                gpu.filter_frame(frame, e, root)
            assert ei.value.status == A.RDF_DEVICE_ERROR and "no progress" in str(ei.value)
        finally:
            lib.set_option("filter_block", 1)
        out = gpu.filter_frame(frame, e, root)
        exp = ora.filter_columns(host, ora.predicate(e, root, host))
        match_unknown_nulls(frame_columns(out)[0], exp[0], "after the stuck call")
        out.release()
