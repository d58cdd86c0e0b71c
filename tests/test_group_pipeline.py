"""Fused grouped aggregation (rdf_group_pipeline; BASELINE.json config C5, TPC-H Q1 shape).

The reference plans GroupAggregate (Dataset::try_aggregate, src/expression.rs:114-221) but panics on execution
(src/evaluation.rs:73), so the expectations are SQL semantics restated by the oracle (parity unpinned by the
reference) and cross-checked here against a numpy restatement of Q1."""
import numpy as np
import pytest

from rust_dataframe_amd import _abi as A
from util import make_chunks

Q1_CUTOFF = 10471  # 1998-09-02 as days since 1970-01-01 (date32)


def q1_columns(rng, lens, null_frac=0.0, offset=0):
    """Synthetic lineitem per the TPC-H column distributions: 4 f64 measures, 2 i8 dictionary
    codes, date32 ship date."""
    cols = {k: [] for k in ("qty", "price", "disc", "tax", "flag", "status", "ship")}
    for n in lens:
        def arr(v, dt):
            valid = (rng.uniform(size=n) >= null_frac) if null_frac > 0 else None
            return A.HostArray.from_numpy(v, valid=valid, offset=offset, dtype=dt, rng=rng)
        cols["qty"].append(arr(rng.integers(1, 51, n).astype(np.float64), A.F64))
        cols["price"].append(arr(np.round(rng.uniform(900.0, 105000.0, n), 2), A.F64))
        cols["disc"].append(arr(rng.integers(0, 11, n) / 100.0, A.F64))
        cols["tax"].append(arr(rng.integers(0, 9, n) / 100.0, A.F64))
        cols["flag"].append(arr(rng.integers(0, 3, n).astype(np.int8), A.I8))
        cols["status"].append(arr(rng.integers(0, 2, n).astype(np.int8), A.I8))
        cols["ship"].append(arr(rng.integers(8036, 10562, n).astype(np.int32), A.I32))
    return cols


def q1_program():
    """cols: 0 qty 1 price 2 disc 3 tax 4 flag 5 status 6 ship.  group id = flag * 2 + status (6 slots)."""
    e = A.Expr()
    qty, price, disc, tax, flag, status, ship = (e.col(i) for i in range(7))
    pred = e.op("le", ship, e.scalar(Q1_CUTOFF, A.I32))
    gid = e.op("add", e.op("multiply", e.cast(flag, A.I32), e.scalar(2, A.I32)), e.cast(status, A.I32))
    disc_price = e.op("multiply", price, e.op("subtract", e.scalar(1.0), disc))
    charge = e.op("multiply", disc_price, e.op("add", e.scalar(1.0), tax))
    return e, pred, gid, [qty, price, disc_price, charge, disc]


def check_groups(got, exp, what=""):
    (gres, grows), (eres, erows) = got, exp
    assert grows == erows, f"{what}: count(*) per group"
    for v, (gr, er) in enumerate(zip(gres, eres)):
        for g, ((gs, gc), (es, ec)) in enumerate(zip(gr, er)):
            assert gc == ec, f"{what}: value {v} group {g}: count {gc} != {ec}"
            if isinstance(es, float):
                assert abs(gs - es) <= 1e-6 * abs(es) + 1e-12, f"{what}: value {v} group {g}: sum {gs} != {es}"
            else:
                assert gs == es, f"{what}: value {v} group {g}: sum {gs} != {es}"


def numpy_q1(cols):
    """Independent restatement with numpy (pins the oracle on the no-null case)."""
    cat = {k: np.concatenate([a.to_numpy() for a in v]) for k, v in cols.items()}
    keep = cat["ship"] <= Q1_CUTOFF
    gid = cat["flag"].astype(np.int32) * 2 + cat["status"]
    dp = cat["price"] * (1.0 - cat["disc"])
    vals = [cat["qty"], cat["price"], dp, dp * (1.0 + cat["tax"]), cat["disc"]]
    res = [[(float(v[keep & (gid == g)].sum()), int((keep & (gid == g)).sum())) for g in range(6)] + [(0.0, 0)] for v in vals]
    rows = [int((keep & (gid == g)).sum()) for g in range(6)] + [0]
    return res, rows


def _cols_list(cols):
    return [cols[k] for k in ("qty", "price", "disc", "tax", "flag", "status", "ship")]


def test_oracle_q1_matches_numpy(ora):
    rng = np.random.default_rng(71)
    cols = q1_columns(rng, [1024, 1024, 576])
    e, pred, gid, vals = q1_program()
    got = ora.group_pipeline(e, _cols_list(cols), vals, gid, 6, pred)
    check_groups(got, numpy_q1(cols), "oracle vs numpy")


@pytest.mark.gpu
@pytest.mark.parametrize("lens,nf,off", [([5], 0.0, 0), ([1024, 1024, 576], 0.0, 0), ([700, 0, 3000], 0.15, 13), ([200_000], 0.02, 3), ([0], 0.0, 0),
                                         ([4096, 100, 70_001], 0.1, 0)])
def test_q1_parity(gpu, ora, request, lens, nf, off):
    rng = np.random.default_rng(sum(lens) + off)
    cols = q1_columns(rng, lens, nf, off)
    e, pred, gid, vals = q1_program()
    exp = ora.group_pipeline(e, _cols_list(cols), vals, gid, 6, pred)
    got = gpu.group_pipeline(e, _cols_list(cols), vals, gid, 6, pred)
    check_groups(got, exp, f"lens={lens} nulls={nf}")
    if sum(lens) and off == 0:  # the register-accumulator kernel must be the one that ran in 'spec' mode (aligned chunks)
        from rust_dataframe_amd import lib
        want = "gspec_kernel<G6;" if request.node.callspec.params["gpu"] == "spec" else "eval_kernel<GROUP>"
        assert lib.last_kernel().startswith(want), lib.last_kernel()
    if nf == 0.0 and sum(lens):
        check_groups(got, numpy_q1(cols), "gpu vs numpy")


@pytest.mark.gpu
@pytest.mark.parametrize("key_dtype,val_dtype", [(A.I8, A.I64), (A.U8, A.F32), (A.I64, A.F64), (A.U32, A.I32), (A.BOOL, A.F64)])
def test_group_pipeline_types_and_shapes(gpu, ora, request, key_dtype, val_dtype):
    """Integer sums wrap and are bit-exact; float sums 1e-6; the key column itself as group id; 1..8 values."""
    rng = np.random.default_rng(300 + key_dtype * 16 + val_dtype)
    for lens, nf, off, ng in [([3000, 1], 0.1, 7, 2 if key_dtype == A.BOOL else 100), ([50_000], 0.0, 0, 2 if key_dtype == A.BOOL else 17),
                              ([50_000, 77], 0.05, 0, 2 if key_dtype == A.BOOL else 5)]:
        if key_dtype == A.BOOL:
            keys = [A.HostArray.from_numpy(rng.integers(0, 2, n).astype(bool), valid=(rng.uniform(size=n) >= nf) if nf else None, offset=off, rng=rng) for n in lens]
        else:
            keys = [A.HostArray.from_numpy(rng.integers(0, ng, n).astype(A.NP_OF[key_dtype]), valid=(rng.uniform(size=n) >= nf) if nf else None, offset=off, rng=rng) for n in lens]
        vals = make_chunks(rng, val_dtype, lens, nf, off, kind="extreme" if val_dtype == A.I64 else "plain")
        for nv in (1, 3, 8):
            if (ng + 1) * nv > A.MAX_GROUP_SLOTS:
                continue
            e = A.Expr()
            k, v = e.col(0), e.col(1)
            roots = [v] + [e.op("add", v, e.scalar(i, val_dtype)) for i in range(1, nv)]
            exp = ora.group_pipeline(e, [keys, vals], roots, k, ng)
            got = gpu.group_pipeline(e, [keys, vals], roots, k, ng)
            check_groups(got, exp, f"key={key_dtype} val={val_dtype} nv={nv} lens={lens}")
            if ng == 5 and nv == 1 and (key_dtype, val_dtype) in ((A.I8, A.I64), (A.I64, A.F64)) and request.node.callspec.params["gpu"] == "spec":
                from rust_dataframe_amd import lib
                assert lib.last_kernel().startswith("gspec_kernel<G8;P:-;K:c0"), lib.last_kernel()


@pytest.mark.gpu
def test_group_pipeline_errors(gpu, ora):
    rng = np.random.default_rng(5)
    keys = [A.HostArray.from_numpy(np.array([0, 1, 7, 2], dtype=np.int32))]
    vals = [A.HostArray.from_numpy(np.array([1.0, 2.0, 3.0, 4.0]))]
    e = A.Expr()
    k, v = e.col(0), e.col(1)
    for api in (ora, gpu):
        with pytest.raises(A.RdfError) as ei:  # id 7 outside [0, 4)
            api.group_pipeline(e, [keys, vals], [v], k, 4)
        assert ei.value.status == A.RDF_COMPUTE_ERROR
        # ... unless the filter drops that row first
        res, rows = api.group_pipeline(e, [keys, vals], [v], k, 4, e.op("lt", k, e.scalar(5, A.I32)))
        assert rows == [1, 1, 1, 0, 0] and res[0][2] == (4.0, 1)
        with pytest.raises(A.RdfError) as ei:  # negative ids are out of range too
            api.group_pipeline(e, [[A.HostArray.from_numpy(np.array([0, -1, 1, 2], dtype=np.int32))], vals], [v], k, 4)
        assert ei.value.status == A.RDF_COMPUTE_ERROR
        with pytest.raises(A.RdfError) as ei:  # float group id
            api.group_pipeline(e, [keys, vals], [v], v, 4)
        assert ei.value.status == A.RDF_INVALID_ARGUMENT
        with pytest.raises(A.RdfError) as ei:  # domain too large for the fused sink
            api.group_pipeline(e, [keys, vals], [v], k, 5000)
        assert ei.value.status == A.RDF_INVALID_ARGUMENT


@pytest.mark.gpu
def test_grouped_kernels_compiled_at_run_time(gpu, ora, request):
    """A grouped program the catalog of rdf_gspec.hip does not hold (it has the Q1 shape and sums of plain columns per key): the
    group id computed from a float column, three value expressions (a tree, a function inside a chain, an i64 product), behind a
    two-column predicate — gspec_kernel<GProg> is compiled for it at run time (rdf_jit.cpp) instead of the interpreter's grouped
    sink; chunked inputs with NULLs; the second call comes from the process's cache; with the switch off the interpreter answers."""
    from rust_dataframe_amd import lib
    if request.node.callspec.params["gpu"] != "spec":
        pytest.skip("with the specialised kernels off nothing is compiled")
    rng = np.random.default_rng(909)
    lens = [4096, 700, 0, 3000]
    x = make_chunks(rng, A.F64, lens, 0.05, 0, kind="unit")
    y = make_chunks(rng, A.F64, lens, 0.1, 0, kind="unit")
    k = make_chunks(rng, A.I64, lens, 0.0, 0, kind="plain")
    m = make_chunks(rng, A.I32, lens, 0.02, 0, kind="plain")
    cols = [x, y, k, m]
    e = A.Expr()
    cx, cy, ck, cm = (e.col(i) for i in range(4))
    gid = e.cast(e.op("multiply", e.op("abs", cx), e.scalar(4.999)), A.I32)                    # |x| <= 1 -> groups 0..4
    vals = [e.op("add", e.op("multiply", cx, cy), e.op("divide", cy, e.scalar(3.0))),
            e.op("multiply", e.op("sin", cx), e.op("sqrt", e.op("abs", cy))),
            e.op("multiply", ck, e.cast(cm, A.I64))]
    pred = e.op("and", e.op("gt", cx, e.scalar(-0.8)), e.op("ne", cm, e.scalar(7, A.I32)))
    exp = ora.group_pipeline(e, cols, vals, gid, 5, pred)
    lib.set_option("jit", 2)      # the call waits for the compiler
    try:
        for _ in range(2):
            got = gpu.group_pipeline(e, cols, vals, gid, 5, pred)
            assert lib.last_kernel().startswith("gspec_kernel<G6;") and lib.last_kernel().endswith("[compiled at run time]"), lib.last_kernel()
            check_groups(got, exp, "grouped program compiled at run time")
        lib.set_option("jit", 0)
        got = gpu.group_pipeline(e, cols, vals, gid, 5, pred)
        assert lib.last_kernel().startswith("eval_kernel<GROUP>"), lib.last_kernel()
        check_groups(got, exp, "grouped program interpreted")
    finally:
        lib.set_option("jit", 0)
