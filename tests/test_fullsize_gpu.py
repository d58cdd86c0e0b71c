"""BASELINE.json's full sizes on the GPU, checked through size-independent properties (the oracle only sees a
sample prefix): partition-of-unity / linearity of the fused filter->sum, fused == materialised, idempotence of
filter, shard sums == whole, bit-exact min/max between the fused and the stored a*b+c, sortedness after sort,
group sums adding up to the column sum."""
import pytest

from rust_dataframe_amd import _abi as A

pytestmark = pytest.mark.gpu

N = 1_000_000_000


def _dev(n, col, dtype, lo, hi, seed=42):
    import torch
    from rust_dataframe_amd import lib
    if dtype == A.F64:
        t = torch.empty(n, dtype=torch.float64, device="cuda")
        lib.fill_uniform_f64(t.data_ptr(), n, seed, col, 0, lo, hi)
    else:
        t = torch.empty(n, dtype=torch.int64, device="cuda")
        lib.fill_uniform_i64(t.data_ptr(), n, seed, col, 0, lo, hi)
    return t


def _arr(t, dtype, n, first=0):
    return A.DeviceArray(t.data_ptr() + first * 8, None, 0, n, dtype, 0, keep=t)


def _out(dtype, n, es=8):
    import torch
    pad = (n + 63) // 64 * 64
    v = torch.empty(pad * es if es else pad // 8 + 8, dtype=torch.uint8, device="cuda")
    return A.DeviceArray(v.data_ptr(), None, 0, n, dtype, 0, keep=v)


def test_headline_1e9_properties(gpu, ora):
    import torch
    x = _dev(N, 0, A.F64, 0.0, 1.0)
    X = _arr(x, A.F64, N)
    e = A.Expr()
    c = e.col(0)
    gt, le = e.op("gt", c, e.scalar(0.5)), e.op("le", c, e.scalar(0.5))
    hi = gpu.pipeline(e, [[X]], [c], gt)[0]
    lo = gpu.pipeline(e, [[X]], [c], le)[0]
    whole = gpu.pipeline(e, [[X]], [c])[0]
    # partition of unity: the two filters split the rows exactly and the sums add up
    assert hi.count + lo.count == N == whole.count
    assert abs((hi.sum + lo.sum) - whole.sum) <= 1e-9 * whole.sum
    assert hi.min > 0.5 >= lo.max and whole.min == lo.min and whole.max == hi.max
    assert abs(hi.count / N - 0.5) < 1e-3 and abs(whole.sum / N - 0.5) < 1e-4      # U[0,1)
    # shards == whole (the multi-GPU partitioning, on one device): 8 row ranges, folded in rank order
    from rust_dataframe_amd import sharding
    parts = []
    for r in range(8):
        b, en = sharding.shard_rows(N, 8, r)
        parts.append(gpu.pipeline(e, [[_arr(x, A.F64, en - b, b)]], [c], gt)[0])
    assert sum(p.count for p in parts) == hi.count
    assert abs(sum(p.sum for p in parts) - hi.sum) <= 1e-12 * hi.sum
    assert max(p.max for p in parts) == hi.max and min(p.min for p in parts) == hi.min
    # chunked (1M-row RecordBatches) == one chunk: bit-exact count, 1e-12 sum
    chunks = [_arr(x, A.F64, min(1 << 20, N - i), i) for i in range(0, N, 1 << 20)]
    ch = gpu.pipeline(e, [chunks], [c], gt)[0]
    assert ch.count == hi.count and abs(ch.sum - hi.sum) <= 1e-12 * hi.sum
    # the oracle on a 2e6-row prefix agrees
    n0 = 2_000_000
    host = A.HostArray.from_numpy(x[:n0].cpu().numpy())
    g0 = gpu.pipeline(e, [[_arr(x, A.F64, n0)]], [c], gt)[0]
    o0 = ora.pipeline(e, [[host]], [c], gt)[0]
    assert g0.count == o0.count and abs(g0.sum - o0.sum) <= 1e-6 * o0.sum
    # fused == materialised: predicate -> mask, Column::filter, then sum; filtering again is idempotent
    n1 = 200_000_000
    X1 = _arr(x, A.F64, n1)
    mask = _out(A.BOOL, n1, 0)
    gpu.predicate(e, gt, [[X1]], [mask])
    kept = gpu.filter_count([mask])[0]
    fused = gpu.pipeline(e, [[X1]], [c], gt)[0]
    assert kept == fused.count
    out = _out(A.F64, kept)
    gpu.filter([X1], [mask], [out])
    assert out.length == kept
    mat = gpu.pipeline(e, [[out]], [c])[0]
    assert mat.count == kept and mat.min == fused.min and mat.max == fused.max
    assert abs(mat.sum - fused.sum) <= 1e-9 * fused.sum
    again = gpu.pipeline(e, [[out]], [c], gt)[0]
    assert again.count == kept                      # everything already satisfies the predicate
    del x, out, mask
    torch.cuda.empty_cache()


def test_c3_1e9_properties(gpu):
    """Config C3: 1e9 rows x 4 columns (32 GB): fused a*b+c -> min/max/count and the key's min/max/count in one
    pass; the stored y (two roundings, no FMA contraction) has bit-identical extrema."""
    import torch
    a, b, cc = (_dev(N, i, A.F64, -1.0, 1.0) for i in range(3))
    k = _dev(N, 3, A.I64, -2 ** 31, 2 ** 31)
    cols = [[_arr(t, dt, N)] for t, dt in ((a, A.F64), (b, A.F64), (cc, A.F64), (k, A.I64))]
    e = A.Expr()
    fma = e.op("add", e.op("multiply", e.col(0), e.col(1)), e.col(2))
    y, kk = gpu.pipeline(e, cols, [fma, e.col(3)])
    assert y.count == N == kk.count
    assert -2.0 <= y.min < -1.9 and 1.9 < y.max <= 2.0 and abs(y.sum / N) < 1e-3
    assert -2 ** 31 <= kk.min < -2 ** 31 + 100 and 2 ** 31 - 100 < kk.max < 2 ** 31
    out = _out(A.F64, N)
    gpu.pipeline(e, cols[:3], [fma], -1, A.SINK_STORE, [[out]])
    ys = gpu.pipeline(e, [[out]], [e.col(0)])[0]
    assert ys.min == y.min and ys.max == y.max and ys.count == N
    assert abs(ys.sum - y.sum) <= 1e-9 * N          # |y| <= 2: absolute bound on the reassociated sum
    k2 = gpu.pipeline(e, [cols[3]], [e.col(0)])[0]
    assert (k2.sum, k2.min, k2.max) == (kk.sum, kk.min, kk.max)   # integers: bit-exact
    del a, b, cc, k, out
    torch.cuda.empty_cache()


def test_trig_identities_1e9(gpu):
    """The kernels' own sine / cosine at full size, through identities that need no oracle: sin^2 + cos^2 == 1 per row
    (min / max within a few ulp of 1, so the sum is the row count), sin(-x) == -sin(x) (odd: the two sums cancel
    exactly), and the stored sin(x) aggregates to what the fused kernel folded (same per-row arithmetic on both paths,
    rows-at-once or not)."""
    import torch
    x = _dev(N, 0, A.F64, -1000.0, 1000.0)
    cols = [[_arr(x, A.F64, N)]]
    e = A.Expr()
    s, c = e.op("sin", e.col(0)), e.op("cos", e.col(0))
    one = gpu.pipeline(e, cols, [e.op("add", e.op("multiply", s, s), e.op("multiply", c, c))])[0]
    assert one.count == N and abs(one.min - 1.0) < 1e-15 and abs(one.max - 1.0) < 1e-15 and abs(one.sum - N) <= 1e-9 * N
    sp = gpu.pipeline(e, cols, [s])[0]
    sn = gpu.pipeline(e, cols, [e.op("sin", e.op("subtract", e.scalar(0.0), e.col(0)))])[0]
    assert sp.count == sn.count == N and sn.sum == -sp.sum and sn.min == -sp.max and sn.max == -sp.min
    assert -1.0 <= sp.min < -0.999999 and 0.999999 < sp.max <= 1.0
    out = _out(A.F64, N)
    gpu.pipeline(e, cols, [s], -1, A.SINK_STORE, [[out]])
    ss = gpu.pipeline(e, [[out]], [e.col(0)])[0]
    assert (ss.min, ss.max, ss.count) == (sp.min, sp.max, N) and ss.sum == sp.sum     # same values, same fold order
    del x, out
    torch.cuda.empty_cache()


def test_sort_and_groupby_large_properties(gpu):
    import torch
    n = 50_000_000
    k = _dev(n, 5, A.I64, -10 ** 12, 10 ** 12)
    K = _arr(k, A.I64, n)
    idx = _out(A.U32, n, 4)
    gpu.sort_to_indices([[K]], [False], idx)
    srt = _out(A.I64, n)
    gpu.take([K], idx, srt)
    # sortedness: s[i] <= s[i+1] everywhere (predicate over two shifted views); a permutation keeps sum/min/max
    e = A.Expr()
    le = e.op("le", e.col(0), e.col(1))
    ok = gpu.pipeline(e, [[_arr(srt.keep, A.I64, n - 1)], [_arr(srt.keep, A.I64, n - 1, 1)]], [e.col(0)], le)[0]
    assert ok.count == n - 1
    s0, s1 = gpu.pipeline(e, [[K]], [e.col(0)])[0], gpu.pipeline(e, [[srt]], [e.col(0)])[0]
    assert (s0.sum, s0.min, s0.max, s0.count) == (s1.sum, s1.min, s1.max, s1.count)
    # GROUP BY: group sums add up to the column sum, counts to n, every key appears once
    g = _dev(n, 6, A.I64, 0, 1_000_000)
    v = _dev(n, 7, A.F64, 0.0, 1.0)
    outs = (_out(A.I64, 1_000_002), _out(A.F64, 1_000_002), _out(A.I64, 1_000_002))
    gk, gs, gc = gpu.groupby_sum([_arr(g, A.I64, n)], [_arr(v, A.F64, n)], 1_000_000, outs)
    ng = gk.length
    assert 999_000 < ng <= 1_000_000
    tot = gpu.pipeline(e, [[_arr(v, A.F64, n)]], [e.col(0)])[0]
    sums = gpu.pipeline(e, [[A.DeviceArray(gs.values_ptr, None, 0, ng, A.F64, 0)]], [e.col(0)])[0]
    cnts = gpu.pipeline(e, [[A.DeviceArray(gc.values_ptr, None, 0, ng, A.I64, 0)]], [e.col(0)])[0]
    assert cnts.sum == n and abs(sums.sum - tot.sum) <= 1e-9 * tot.sum
    KG = A.DeviceArray(gk.values_ptr, None, 0, ng, A.I64, 0)
    kidx = _out(A.U32, ng, 4)
    gpu.sort_to_indices([[KG]], [False], kidx)
    ks = _out(A.I64, ng)
    gpu.take([KG], kidx, ks)
    lt = e.op("lt", e.col(0), e.col(1))
    strict = gpu.pipeline(e, [[_arr(ks.keep, A.I64, ng - 1)], [_arr(ks.keep, A.I64, ng - 1, 1)]], [e.col(0)], lt)[0]
    assert strict.count == ng - 1    # strictly increasing: no key appears twice
    torch.cuda.empty_cache()


def test_c1_csv_shape_1e6_rows_977_batches(gpu, ora):
    """Config C1 (the reference's own CPU-runnable case): 1 000 000 f64 rows read in 1024-row batches (977 chunks, the
    last one 576 rows), y = sin(x + 1.0), sum(y) — handed over as HOST buffers like the Rust shim would, so the
    small-chunk staging path (one packed H2D copy) is what runs.  Oracle on the full input; numpy as a second anchor."""
    import numpy as np
    n = 1_000_000
    x = np.random.default_rng(42).uniform(0.0, 1.0, n)
    chunks = [A.HostArray.from_numpy(x[i:i + 1024]) for i in range(0, n, 1024)]
    assert len(chunks) == 977 and chunks[-1].length == 576
    e = A.Expr()
    y = e.op("sin", e.op("add", e.col(0), e.scalar(1.0)))
    g = gpu.pipeline(e, [chunks], [y])[0]
    o = ora.pipeline(e, [chunks], [y])[0]
    ref = float(np.sin(x + 1.0).sum())
    assert g.count == o.count == n
    assert abs(g.sum - o.sum) <= 1e-6 * abs(o.sum) and abs(g.sum - ref) <= 1e-9 * abs(ref)
    assert abs(g.min - o.min) <= 1e-12 and abs(g.max - o.max) <= 1e-12
    # unfused, like Evaluate::calculate twice: add -> new column, sin -> new column (chunking preserved), then sum
    s1 = gpu.binary("add", chunks, [A.HostArray.from_numpy(np.full(c.length, 1.0)) for c in chunks])
    s2 = gpu.unary("sin", s1)
    assert [c.length for c in s2] == [c.length for c in chunks]
    tot = gpu.sum(s2)
    assert abs(tot - ref) <= 1e-9 * abs(ref)


def _bench_workload(name, rows):
    """bench.py's own --workload builder, step and check (so the benchmark's generator and self-checks are what is tested)."""
    import torch
    import bench
    from rust_dataframe_amd import lib, sharding
    dev = torch.device("cuda", 0)
    step, alg_bytes, desc, check = bench.WORKLOADS[name](torch, lib, lib.api(), A, sharding, dev, dev, 0, rows, 0, rows, {})
    return step, check


def test_groupby_1e9_1e6keys_properties(gpu):
    """Config C4 at full single-GPU size: 1e9 rows, i64 keys = hash(row) mod 1e6, f64 values.  Size-independent
    properties (no key twice, every count positive, counts add up to the rows, group sums to the column sum, exactly
    1e6 groups, keys inside the domain) + the oracle on a 2e6-row prefix, through bench.py's own workload."""
    import torch
    from rust_dataframe_amd import lib
    step, check = _bench_workload("c4", N)
    res = step()
    assert res["groups"] == 1_000_000
    assert lib.last_kernel().startswith("gb2_scatter_kernel"), lib.last_kernel()
    chk = check(res, 1)
    assert chk["self_check"] is True and chk["parity_on_sample"] is True and chk["groups_total"] == 1_000_000, chk
    del step, check
    torch.cuda.empty_cache()


def test_q1_6e8_properties(gpu):
    """Config C5 at SF100 size (6e8 lineitem rows): per-group averages inside the columns' domains, the filter's
    selectivity, groups of equal weight, discounted price <= price <= charge / 0.9; oracle parity on a 2e6-row prefix."""
    import torch
    step, check = _bench_workload("q1", 600_000_000)
    res = step()
    chk = check(res, 1)
    assert chk["self_check"] is True and chk["parity_on_sample"] is True, (chk, res)
    assert all(1.0 <= s / c <= 50.0 for s, c in zip(res["sum_qty"], res["count_star"]))
    del step, check
    torch.cuda.empty_cache()


def test_c3_bench_workload_checks(gpu):
    import torch
    step, check = _bench_workload("c3", 200_000_000)
    res = step()
    chk = check(res, 1)
    assert chk["self_check"] is True and chk["parity_on_sample"] is True, (chk, res)
    del step, check
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------------
# round 5: the FRAME operators at full size, on the reference readers' batches (src/dataframe.rs:352: 1024 rows) — 1e9 rows are
# 976 563 RecordBatches, i.e. ~1e6 tiles behind the one-pass filter's ticketed look-back and ~1e6 descriptors per column in every
# table a kernel builds.  Size-independent properties, the fused path against the per-column path, the oracle on a prefix.

def _batched_frame(api, cols, rows, chunk_rows, keep):
    """cols: [(torch tensor, rdf dtype)] -> a frame pinned from a numpy-built descriptor table (no Python loop per batch)."""
    import ctypes as C
    import numpy as np
    rec = np.dtype([("values", "u8"), ("validity", "u8"), ("offset", "i8"), ("length", "i8"), ("null_count", "i8"), ("dtype", "i4"), ("mem", "i4")])
    assert rec.itemsize == C.sizeof(A.rdf_array)
    starts = np.arange(0, rows, chunk_rows, dtype=np.int64)
    nch = len(starts)
    tab = np.zeros(len(cols) * nch, dtype=rec)
    for k, (t, dt) in enumerate(cols):
        v = tab[k * nch:(k + 1) * nch]
        v["values"] = t.data_ptr() + starts * 8
        v["length"] = np.minimum(chunk_rows, rows - starts)
        v["dtype"] = dt
        v["mem"] = A.MEM_DEVICE
    h = C.c_void_p(0)
    fn = api._fn("frame_pin")
    fn.restype = C.c_int
    api._check(fn(C.c_void_p(tab.ctypes.data), C.c_int32(len(cols)), C.c_int64(nch), C.byref(h)))
    f = A.Frame(api, h)
    f.keep = keep
    return f, nch


def _same_aggs(got, want, what):
    for v, (g, w) in enumerate(zip(got, want)):
        assert (g.count, g.min, g.max, g.dtype) == (w.count, w.min, w.max, w.dtype), (what, v, g, w)
        if g.dtype == A.F64:
            assert abs(g.sum - w.sum) <= 1e-9 * max(abs(w.sum), 1.0), (what, v, g, w)     # different fold orders of ~5e8 doubles
        else:
            assert g.sum == w.sum, (what, v, g, w)


@pytest.mark.parametrize("ncols", [1, 4])
def test_filter_frame_1e9_rows_in_1024_row_batches(gpu, ora, ncols):
    import numpy as np
    import torch
    from rust_dataframe_amd import lib
    from test_frame_ops_gpu import match_unknown_nulls
    x = _dev(N, 0, A.F64, -1.0, 1.0)
    ts = [(x, A.F64)]
    if ncols == 4:
        ts += [(_dev(N, 3, A.I64, -2 ** 31, 2 ** 31), A.I64), (_dev(N, 1, A.F64, -1.0, 1.0), A.F64), (_dev(N, 2, A.F64, -1.0, 1.0), A.F64)]
    lib.synchronize()
    fr, nch = _batched_frame(gpu, ts, N, 1024, ts)
    assert nch == 976_563
    e = A.Expr()
    vals = [e.col(k) for k in range(ncols)]
    gt, le = e.op("gt", e.col(0), e.scalar(0.0)), e.op("le", e.col(0), e.scalar(0.0))
    want = gpu.pipeline(e, fr, vals, gt)                      # the per-column path: fused filter -> aggregates, nothing materialised
    n_le = gpu.pipeline(e, fr, [e.col(0)], le)[0].count
    try:
        for fused, what in ((1, "one pass (predicate inside the compaction kernel)"), (0, "predicate -> mask, count, compact")):
            lib.set_option("filter_fused", fused)
            out = gpu.filter_frame(fr, e, gt)
            nc, nb, rows = out.info()
            assert (nc, nb) == (ncols, nch) and rows == want[0].count and rows + n_le == N, (what, rows, want[0].count, n_le)
            _same_aggs(gpu.pipeline(e, out, vals), want, what)           # every column of the result against the per-column path
            assert gpu.pipeline(e, out, [e.col(0)], le)[0].count == 0
            again = gpu.filter_frame(out, e, gt)                         # idempotent
            assert again.info() == (ncols, nch, rows)
            _same_aggs(gpu.pipeline(e, again, vals), want, what + ", applied twice")
            again.release()
            # batch boundaries are kept: the lengths of a few batches against a count over the same rows
            for b in (0, nch // 2, nch - 1):
                a = out.column(0)[b]
                nb_rows = min(1024, N - b * 1024)
                cnt = gpu.pipeline(e, [[A.DeviceArray(x.data_ptr() + 8 * b * 1024, None, 0, nb_rows, A.F64, 0)]], [e.col(0)], gt)[0].count
                assert a.length == cnt, (what, b, a.length, cnt)
            out.release()
        # the oracle on a prefix of 2 000 batches, both forms
        m = 2000 * 1024
        pf, _ = _batched_frame(gpu, ts, m, 1024, ts)
        host = []
        for t, dt in ts:
            h = torch.empty(m, dtype=t.dtype)
            lib.load().rdf_copy_d2h(h.data_ptr(), t.data_ptr(), m * 8)
            hv = h.numpy()
            host.append([A.HostArray(hv, None, i, 1024, dt, 0) for i in range(0, m, 1024)])
        mask = ora.predicate(e, gt, host)
        exp = ora.filter_columns(host, mask)
        for fused in (1, 0):
            lib.set_option("filter_fused", fused)
            po = gpu.filter_frame(pf, e, gt)
            for k in range(ncols):
                match_unknown_nulls(po.column_to_host(k), exp[k], f"prefix, fused={fused}, column {k}")
            po.release()
        pf.release()
    finally:
        lib.set_option("filter_fused", 1)
        fr.release()
    del ts, x
    torch.cuda.empty_cache()


def test_take_frame_1e9_rows_in_1024_row_batches(gpu):
    import torch
    from rust_dataframe_amd import lib
    x, k = _dev(N, 0, A.F64, -1.0, 1.0), _dev(N, 3, A.I64, -2 ** 31, 2 ** 31)
    lib.synchronize()
    fr, nch = _batched_frame(gpu, [(x, A.F64), (k, A.I64)], N, 1024, (x, k))
    e = A.Expr()
    vals = [e.col(0), e.col(1)]
    try:
        # consecutive rows (recognised as a sequential list): the first quarter of the frame, bit for bit in every aggregate
        nq = N // 4
        seq = torch.arange(0, nq, dtype=torch.int64, device="cuda").to(torch.uint32)
        torch.cuda.synchronize()
        out = gpu.take_frame(fr, A.DeviceArray(seq.data_ptr(), None, 0, nq, A.U32, 0, keep=seq))
        assert out.info()[2] == nq
        want = gpu.pipeline(e, [[_arr(x, A.F64, nq)], [_arr(k, A.I64, nq)]], vals)
        _same_aggs(gpu.pipeline(e, out, vals), want, "sequential take")
        out.release()
        del seq
        # random rows with repeats: against torch's gather of the same indices (an independent gather)
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        ridx = torch.randint(0, N, (nq,), dtype=torch.int64, device="cuda", generator=g)
        r32 = ridx.to(torch.uint32)
        torch.cuda.synchronize()
        out = gpu.take_frame(fr, A.DeviceArray(r32.data_ptr(), None, 0, nq, A.U32, 0, keep=r32))
        assert out.info()[2] == nq
        got = gpu.pipeline(e, out, vals)
        tx, tk = x[ridx], k[ridx]
        assert got[0].count == nq and got[0].min == tx.min().item() and got[0].max == tx.max().item()
        assert abs(got[0].sum - tx.sum().item()) <= 1e-9 * nq
        assert (got[1].sum, got[1].min, got[1].max) == (tk.sum().item(), tk.min().item(), tk.max().item())
        # the head of the result, bit for bit and in order (whatever batches the operator cut it into)
        a0 = out.column(0)[0]
        m = min(a0.length, 4096)
        h = torch.empty(m, dtype=torch.float64)
        lib.load().rdf_copy_d2h(h.data_ptr(), a0.values_ptr + 8 * a0.offset, m * 8)
        assert torch.equal(h, x[ridx[:m]].cpu())
        out.release()
    finally:
        fr.release()
    del x, k
    torch.cuda.empty_cache()


def test_sort_frame_1e8_rows_in_1024_row_batches(gpu):
    import torch
    from rust_dataframe_amd import lib
    n = 100_000_000
    k, p = _dev(n, 5, A.I64, -10 ** 12, 10 ** 12), _dev(n, 6, A.I64, -2 ** 20, 2 ** 20)
    lib.synchronize()
    fr, nch = _batched_frame(gpu, [(k, A.I64), (p, A.I64)], n, 1024, (k, p))
    e = A.Expr()
    pair = e.op("multiply", e.col(0), e.col(1))              # sum(k * p) wraps, and changes if a row's two values part company
    try:
        before = gpu.pipeline(e, fr, [e.col(0), e.col(1), pair])
        sf, _ = gpu.sort_frame(fr, [0], [False])
        nc, nb, rows = sf.info()
        assert (nc, rows) == (2, n)
        after = gpu.pipeline(e, sf, [e.col(0), e.col(1), pair])
        for b_, a_ in zip(before, after):
            assert (b_.sum, b_.min, b_.max, b_.count) == (a_.sum, a_.min, a_.max, a_.count)      # a permutation of whole rows
        # sortedness over the whole frame (the sorted frame is what DataFrame::take returns: one batch; were it several, they
        # would have to be consecutive in the column's buffer for the shifted views below)
        ck = sf.column(0)
        base = ck[0].values_ptr + 8 * ck[0].offset
        at = 0
        for a_ in ck:
            assert a_.values_ptr + 8 * a_.offset == base + 8 * at
            at += a_.length
        le = e.op("le", e.col(0), e.col(1))
        ok = gpu.pipeline(e, [[A.DeviceArray(base, None, 0, n - 1, A.I64, 0)], [A.DeviceArray(base + 8, None, 0, n - 1, A.I64, 0)]], [e.col(0)], le)[0]
        assert ok.count == n - 1
        sf.release()
    finally:
        fr.release()
    del k, p
    torch.cuda.empty_cache()


def test_groupby_agg_frame_1e9_rows_in_1024_row_batches(gpu):
    import torch
    from rust_dataframe_amd import lib
    ng = 1_000_000
    kk, v = _dev(N, 7, A.I64, 0, ng), _dev(N, 0, A.F64, 0.0, 1.0)
    lib.synchronize()
    fr, nch = _batched_frame(gpu, [(kk, A.I64), (v, A.F64)], N, 1024, (kk, v))
    e = A.Expr()
    try:
        tot = gpu.pipeline(e, fr, [e.col(1)])[0]
        out = gpu.groupby_agg_frame(fr, [0], 1, "sum", ng)
        nc, _, rows = out.info()
        assert nc == 3 and rows == ng                              # 1e9 uniform draws: every one of the 1e6 keys appears
        sq = e.op("multiply", e.col(0), e.col(0))
        keys, sums, counts, keys2 = gpu.pipeline(e, out, [e.col(0), e.col(1), e.col(2), sq])
        assert (keys.min, keys.max, keys.count) == (0, ng - 1, ng)
        assert keys.sum == ng * (ng - 1) // 2 and keys2.sum == (ng - 1) * ng * (2 * ng - 1) // 6      # 1e6 values of [0, 1e6) with these power sums: each key once
        assert counts.sum == N and counts.min >= 1 and abs(counts.sum / ng - N / ng) < 1
        assert abs(sums.sum - tot.sum) <= 1e-9 * tot.sum
        out.release()
    finally:
        fr.release()
    del kk, v
    torch.cuda.empty_cache()


def test_filter_frame_1e9_rows_one_batch(gpu):
    """ONE RecordBatch of 1e9 rows: 244 141 block tiles of 4096 rows (two columns), their prefixes from the scanner wave
    (rdf_bfilter.hip, round 6; round 5: 976 563 wave tiles in 15 259 super-tiles of one look-back chain) — same rows, same order
    as the three-pass path."""
    import torch
    from rust_dataframe_amd import lib
    x, k = _dev(N, 0, A.F64, -1.0, 1.0), _dev(N, 3, A.I64, -2 ** 31, 2 ** 31)
    lib.synchronize()
    fr, nch = _batched_frame(gpu, [(x, A.F64), (k, A.I64)], N, N, (x, k))
    assert nch == 1
    e = A.Expr()
    vals = [e.col(0), e.col(1)]
    gt = e.op("gt", e.col(0), e.scalar(0.25))
    want = gpu.pipeline(e, fr, vals, gt)
    try:
        for fused in (1, 0):
            lib.set_option("filter_fused", fused)
            out = gpu.filter_frame(fr, e, gt)
            assert (lib.last_kernel() == "bfilter_kernel") == bool(fused), lib.last_kernel()      # round 6: block tiles + scanner wave
            assert out.info() == (2, 1, want[0].count)
            _same_aggs(gpu.pipeline(e, out, vals), want, f"one batch, fused={fused}")
            # order is kept: the first kept rows are the first rows of x that pass
            a0 = out.column(0)[0]
            h = torch.empty(4096, dtype=torch.float64)
            lib.load().rdf_copy_d2h(h.data_ptr(), a0.values_ptr + 8 * a0.offset, 4096 * 8)
            head = x[:20000]
            assert torch.equal(h, head[head > 0.25][:4096].cpu())
            out.release()
    finally:
        lib.set_option("filter_fused", 1)
        fr.release()
    del x, k
    torch.cuda.empty_cache()


def test_filter_frame_1e9_rows_in_ragged_long_batches(gpu):
    """1e9 rows in 41 RecordBatches of uneven lengths (3 rows ... 1.2e8 rows, one empty) through the block-tile kernel with the scanner
    wave (rdf_bfilter.hip), next to a row-number column: per batch the kept count equals a filter -> count over that batch alone,
    the row numbers that come out are STRICTLY INCREASING inside every batch (order kept, nothing twice) and stay inside the batch's
    row range, and the aggregates over the whole result equal the fused filter -> aggregate over the input (nothing lost)."""
    import ctypes as C
    import numpy as np
    import torch
    from rust_dataframe_amd import lib
    x = _dev(N, 0, A.F64, -1.0, 1.0)
    k = torch.arange(N, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.synchronize()
    rng = np.random.default_rng(606)
    cuts = np.sort(rng.integers(0, N, 37))
    cuts = np.concatenate([[0, 3, 3], cuts, [N - 8191 - 17, N]]).astype(np.int64)        # a 3-row batch, an empty one, a last one of 8208 rows
    cuts = np.unique(np.concatenate([cuts, [3]]))
    starts, lens = cuts[:-1], np.diff(cuts)
    starts, lens = np.insert(starts, 2, 3), np.insert(lens, 2, 0)                        # the empty batch
    nch = len(starts)
    rec = np.dtype([("values", "u8"), ("validity", "u8"), ("offset", "i8"), ("length", "i8"), ("null_count", "i8"), ("dtype", "i4"), ("mem", "i4")])
    tab = np.zeros(2 * nch, dtype=rec)
    for j, (t, dt) in enumerate(((x, A.F64), (k, A.I64))):
        v = tab[j * nch:(j + 1) * nch]
        v["values"] = t.data_ptr() + starts * 8
        v["length"] = lens
        v["dtype"] = dt
        v["mem"] = A.MEM_DEVICE
    h = C.c_void_p(0)
    fn = gpu._fn("frame_pin")
    fn.restype = C.c_int
    gpu._check(fn(C.c_void_p(tab.ctypes.data), C.c_int32(2), C.c_int64(nch), C.byref(h)))
    fr = A.Frame(gpu, h)
    fr.keep = (x, k)
    e = A.Expr()
    gt = e.op("gt", e.col(0), e.scalar(0.25))
    try:
        want = gpu.pipeline(e, fr, [e.col(0), e.col(1)], gt)
        out = gpu.filter_frame(fr, e, gt)
        assert lib.last_kernel() == "bfilter_kernel", lib.last_kernel()
        assert out.info() == (2, nch, want[0].count)
        _same_aggs(gpu.pipeline(e, out, [e.col(0), e.col(1)]), want, "ragged long batches")
        ox, ok_ = out.column(0), out.column(1)
        e2 = A.Expr()
        inc = e2.op("lt", e2.col(0), e2.col(1))
        for c in range(nch):
            n = ok_[c].length
            alone = gpu.pipeline(e, [[_arr(x, A.F64, int(lens[c]), int(starts[c]))]], [e.col(0)], gt)[0].count if lens[c] else 0
            assert n == alone == ox[c].length, (c, n, alone)
            if n == 0:
                continue
            base = ok_[c].values_ptr + 8 * ok_[c].offset
            rows = gpu.pipeline(e2, [[A.DeviceArray(base, None, 0, n, A.I64, 0)]], [e2.col(0)])[0]
            assert rows.min >= starts[c] and rows.max < starts[c] + lens[c], (c, rows.min, rows.max)
            if n > 1:        # row numbers strictly increasing: k[i] < k[i + 1] for every i
                ordered = gpu.pipeline(e2, [[A.DeviceArray(base, None, 0, n - 1, A.I64, 0)], [A.DeviceArray(base + 8, None, 0, n - 1, A.I64, 0)]], [e2.col(0)], inc)[0]
                assert ordered.count == n - 1, (c, ordered.count, n - 1)
        out.release()
    finally:
        fr.release()
    del x, k
    torch.cuda.empty_cache()
