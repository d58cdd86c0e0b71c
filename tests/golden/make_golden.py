#!/usr/bin/env python3
"""Generates tests/golden/vectors_v1.npz: small seeded input/output vectors for every kernel on the path.

The reference itself cannot run here (2020 nightly Rust + an un-vendored arrow branch), so the
expected outputs come from the CPU oracle (oracle/rdf_oracle.c) and are CROSS-CHECKED against pyarrow.compute —
an independent implementation of the same Arrow kernels the reference calls — before being written.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from rust_dataframe_amd import _abi as A  # noqa: E402

N = 4096
SEED = 20260926


def to_pa(h: A.HostArray):
    v, m = h.to_numpy(), h.valid_mask()
    return pa.array(v, mask=~m)


def main():
    o = oracle.api()
    rng = np.random.default_rng(SEED)
    out = {}
    valid_a = rng.uniform(size=N) >= 0.1
    valid_b = rng.uniform(size=N) >= 0.1
    a = rng.uniform(-1, 1, N)
    b = rng.uniform(-1, 1, N)
    b[b == 0] = 0.5
    c = rng.uniform(-1, 1, N)
    k = rng.integers(-2 ** 31, 2 ** 31, N).astype(np.int64)
    k[:4] = [np.iinfo(np.int64).min, np.iinfo(np.int64).max, 0, -1]
    sp = a.copy()
    sp[:8] = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-300, -1e300, 1e22]
    idx = rng.integers(0, N, 1000).astype(np.uint32)
    idx_valid = rng.uniform(size=1000) >= 0.1
    out.update(a=a, b=b, c=c, k=k, sp=sp, valid_a=valid_a, valid_b=valid_b, idx=idx, idx_valid=idx_valid)
    A_ = A.HostArray.from_numpy(a, valid_a)
    B_ = A.HostArray.from_numpy(b, valid_b)
    C_ = A.HostArray.from_numpy(c)
    K_ = A.HostArray.from_numpy(k, valid_b)
    SP_ = A.HostArray.from_numpy(sp)

    def put(name, h):
        out[name + "_values"] = h.to_numpy().copy()
        out[name + "_valid"] = h.valid_mask().copy()

    # arithmetic (validity = AND), cross-checked with pyarrow
    for op, fn in [("add", pc.add), ("subtract", pc.subtract), ("multiply", pc.multiply), ("divide", pc.divide)]:
        r = o.binary(op, [A_], [B_])[0]
        ref = fn(to_pa(A_), to_pa(B_))
        assert np.array_equal(r.valid_mask(), ~np.asarray(ref.is_null())), op
        m = r.valid_mask()
        assert np.array_equal(r.to_numpy()[m], ref.to_numpy(zero_copy_only=False)[m]), op
        put(op, r)
    ki = o.binary("multiply", [K_], [K_])[0]   # wrapping i64
    put("mul_i64_wrap", ki)
    with np.errstate(over="ignore"):
        assert np.array_equal(ki.to_numpy()[ki.valid_mask()], (k * k)[valid_b])
    # unary (libm)
    for op, npf in [("sin", np.sin), ("cos", np.cos), ("tan", np.tan), ("abs", np.abs), ("sqrt", np.sqrt), ("exp", np.exp),
                    ("floor", np.floor), ("tanh", np.tanh), ("cot", lambda x: 1.0 / np.tan(x)), ("sec", lambda x: 1.0 / np.cos(x)),
                    ("csc", lambda x: 1.0 / np.sin(x))]:
        src = A_ if op != "sqrt" else A.HostArray.from_numpy(np.abs(a), valid_a)
        r = o.unary(op, [src])[0]
        with np.errstate(all="ignore"):
            np.testing.assert_allclose(r.to_numpy()[valid_a], npf(src.to_numpy())[valid_a], rtol=1e-15, atol=0)
        put(op, r)
    for op in ["sin", "cos", "tan", "cot", "sec", "csc"]:
        put(op + "_special", o.unary(op, [SP_])[0])
    # cast i64 -> f64 and f64 -> i32 (saturating `as`)
    put("cast_k_f64", o.cast([K_], A.F64)[0])
    with np.errstate(all="ignore"):
        big = A.HostArray.from_numpy(sp * 1e10)
    out["cast_src_f64"] = big.to_numpy().copy()
    put("cast_f64_i32", o.cast([big], A.I32)[0])
    # aggregates
    for name, h in [("a", A_), ("k", K_)]:
        out[f"sum_{name}"] = np.array(o.sum([h]))
        out[f"min_{name}"] = np.array(o.min([h]))
        out[f"max_{name}"] = np.array(o.max([h]))
        out[f"count_{name}"] = np.array(o.count([h]))
        out[f"avg_{name}"] = np.array(o.avg([h]))
    assert abs(out["sum_a"] - pc.sum(to_pa(A_)).as_py()) < 1e-9
    assert out["min_a"] == pc.min(to_pa(A_)).as_py() and out["max_k"] == pc.max(to_pa(K_)).as_py()
    assert out["count_a"] == pc.count(to_pa(A_)).as_py()
    # predicate + filter + take
    e = A.Expr()
    ca, cb = e.col(0), e.col(1)
    pred = e.op("gt", ca, e.scalar(0.25))
    pred2 = e.op("and", e.op("le", ca, cb), e.op("not", e.op("gt", cb, e.scalar(0.9))))
    for name, root in [("pred_gt", pred), ("pred_and_not", pred2)]:
        m = o.predicate(e, root, [[A_], [B_]])[0]
        put(name, m)
    m = o.predicate(e, pred, [[A_], [B_]])[0]
    ref = pc.greater(to_pa(A_), pa.scalar(0.25))
    assert np.array_equal(m.valid_mask(), ~np.asarray(ref.is_null()))
    assert np.array_equal(m.to_numpy() & m.valid_mask(), np.asarray(ref.fill_null(False)))
    f = o.filter([K_], [m])[0]
    reff = pc.filter(to_pa(K_), ref)   # drops null-mask rows: same as "value bit 0 at null slots"
    assert f.length == len(reff) and np.array_equal(f.valid_mask(), ~np.asarray(reff.is_null()))
    assert np.array_equal(f.to_numpy()[f.valid_mask()], reff.fill_null(0).to_numpy(zero_copy_only=False)[f.valid_mask()])
    put("filter_k_by_pred_gt", f)
    IDX = A.HostArray.from_numpy(idx, idx_valid)
    t = o.take([A_.slice(0, 1000), A_.slice(1000, N - 1000)], IDX)
    reft = pc.take(to_pa(A_), to_pa(IDX))
    assert np.array_equal(t.valid_mask(), ~np.asarray(reft.is_null()))
    assert np.array_equal(t.to_numpy()[t.valid_mask()], reft.to_numpy(zero_copy_only=False)[t.valid_mask()])
    put("take_a", t)
    # fused pipelines (as aggregates)
    r = o.pipeline(e, [[A_], [B_]], [ca], pred)[0]
    out["pipe_filter_sum"] = np.array([r.sum, r.min, r.max, r.count])
    e2 = A.Expr()
    fa, fb, fc, fk = e2.col(0), e2.col(1), e2.col(2), e2.col(3)
    y = e2.op("add", e2.op("multiply", fa, fb), fc)
    rr = o.pipeline(e2, [[A_], [B_], [C_], [K_]], [y, fk])
    out["pipe_c3_y"] = np.array([rr[0].sum, rr[0].min, rr[0].max, rr[0].count])
    out["pipe_c3_k"] = np.array([rr[1].sum, rr[1].min, rr[1].max, rr[1].count], dtype=np.int64)
    ab = (a * b)
    yy = (ab + c)[valid_a & valid_b]
    assert rr[0].min == yy.min() and rr[0].max == yy.max() and rr[0].count == len(yy)
    e3 = A.Expr()
    s1 = e3.op("sin", e3.op("add", e3.col(0), e3.scalar(1.0)))
    r3 = o.pipeline(e3, [[A_]], [s1])[0]
    out["pipe_c1_sin_add"] = np.array([r3.sum, r3.min, r3.max, r3.count])
    np.testing.assert_allclose(r3.sum, np.sin(a + 1.0)[valid_a].sum(), rtol=1e-12)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
