#!/usr/bin/env python3
"""eval_lean_kernel against eval_kernel on the same interpreted aggregate programs (round 6): results must be the same bits.
Random expression trees over f64 / i64 / u64 columns — comparisons, + - * /, AND / OR / NOT, casts to f64, literals on either
side — on data with NaN, +-0, infinities, NULLs, ragged multi-batch columns whose batches start at odd offsets; every program
runs with rdf_set_option("interp_lean", 0) and 1 (specialised kernels and run-time compilation off) and the two
rdf_agg_result lists are compared field by field as bit patterns (a NaN equals any NaN).  Prints one JSON line: programs run, how many took the lean
kernel, mismatches.  --time adds the two kernels' times on filter -> sum at --rows.
Usage (GPU box): python tools/lean_ab.py [--programs 300] [--seed 1] [--time --rows 1000000000]"""
import argparse
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from rust_dataframe_amd import _abi as A  # noqa: E402
from rust_dataframe_amd import lib  # noqa: E402


def bits(x):
    """the value as a bit pattern; every NaN is one value (which NaN an f64 operation hands on — sign, payload — depends on the
    operand order and the source modifiers the compiler picked: the parity tests against the oracle treat NaN the same way)"""
    if isinstance(x, float):
        return b"nan" if x != x else struct.pack("<d", x)
    return int(x)


def same(r0, r1):
    if len(r0) != len(r1):
        return False
    for a, b in zip(r0, r1):
        if (bits(a.sum), bits(a.min), bits(a.max), a.count, a.is_some, a.dtype) != (bits(b.sum), bits(b.min), bits(b.max), b.count, b.is_some, b.dtype):
            return False
    return True


SPECIAL_F = [0.0, -0.0, float("nan"), float("inf"), float("-inf"), 1.0, -1.0, 0.5, 1e308, -1e308, 5e-324]


def make_column(rng, dt, lens, null_frac):
    """-> [DeviceArray per batch]; batches are slices of one buffer at odd element offsets"""
    arrs = []
    for n in lens:
        off = int(rng.integers(0, 5))
        tot = n + off + 8
        if dt == A.F64:
            h = rng.uniform(-2, 2, tot)
            k = rng.random(tot) < 0.05
            h[k] = rng.choice(SPECIAL_F, int(k.sum()))
            t = torch.from_numpy(h).cuda()
        elif dt == A.I64:
            h = rng.integers(-5, 6, tot, dtype=np.int64)
            k = rng.random(tot) < 0.02
            h[k] = rng.choice(np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max, 0, 2 ** 53 + 1], dtype=np.int64), int(k.sum()))
            t = torch.from_numpy(h).cuda()
        else:
            h = rng.integers(0, 7, tot, dtype=np.uint64)
            k = rng.random(tot) < 0.02
            h[k] = rng.choice(np.array([2 ** 64 - 1, 2 ** 63, 0, 2 ** 53 + 1], dtype=np.uint64), int(k.sum()))
            t = torch.from_numpy(h.view(np.int64)).cuda()
        v = None
        if null_frac > 0:
            vb = np.packbits((rng.random(tot + 64) >= null_frac).astype(np.uint8), bitorder="little")
            v = torch.from_numpy(np.concatenate([vb, np.zeros(16, np.uint8)])).cuda()
        arrs.append(A.DeviceArray(t.data_ptr(), v.data_ptr() if v is not None else None, off, n, dt, -1, keep=(t, v)))
    return arrs


def random_tree(rng, e, dts, depth, want):
    """want: 'f' an f64 value, 'i' a 64-bit integer value of column type dts[k], 'b' a predicate; -> node index (or None)"""
    ncols = len(dts)
    fcols = [c for c in range(ncols) if dts[c] == A.F64]
    icols = [c for c in range(ncols) if dts[c] != A.F64]
    if want == "b":
        r = rng.random()
        if depth > 0 and r < 0.3:
            return e.op(str(rng.choice(["and", "or"])), random_tree(rng, e, dts, depth - 1, "b"), random_tree(rng, e, dts, depth - 1, "b"))
        if depth > 0 and r < 0.4:
            return e.op("not", random_tree(rng, e, dts, depth - 1, "b"))
        cmp_ = str(rng.choice(["gt", "ge", "eq", "ne", "lt", "le"]))
        kind = "f" if (fcols and (not icols or rng.random() < 0.6)) else "i"
        lhs = random_tree(rng, e, dts, max(depth - 1, 0), kind)
        rr = rng.random()
        if rr < 0.5:
            rhs = e.scalar(float(rng.choice([0.0, 0.5, -1.0, 2.0, float("nan")])))
        elif rr < 0.6:
            rhs = e.scalar(int(rng.integers(-3, 4)))
        else:
            rhs = random_tree(rng, e, dts, max(depth - 1, 0), kind)
        return e.op(cmp_, lhs, rhs) if rng.random() < 0.7 else e.op(cmp_, rhs, lhs)
    if want == "f":
        if not fcols:
            return e.op("cast", random_tree(rng, e, dts, depth, "i"), dtype=A.F64)
        if depth == 0 or rng.random() < 0.3:
            if icols and rng.random() < 0.2:
                return e.op("cast", e.col(int(rng.choice(icols))), dtype=A.F64)
            return e.col(int(rng.choice(fcols)))
        op = str(rng.choice(["add", "subtract", "multiply", "divide"]))
        a = random_tree(rng, e, dts, depth - 1, "f")
        b = e.scalar(float(rng.choice([0.0, 0.5, -1.5, 3.0]))) if rng.random() < 0.4 else random_tree(rng, e, dts, depth - 1, "f")
        return e.op(op, a, b) if rng.random() < 0.7 else e.op(op, b, a)
    # 'i': one integer column type per tree (the reference's binary kernels take equal types)
    c0 = int(rng.choice(icols))
    same_t = [c for c in icols if dts[c] == dts[c0]]

    def itree(d):
        if d == 0 or rng.random() < 0.4:
            return e.col(int(rng.choice(same_t)))
        op = str(rng.choice(["add", "subtract", "multiply"]))
        a = itree(d - 1)
        b = e.scalar(int(rng.integers(0, 4)), dtype=dts[c0]) if rng.random() < 0.4 else itree(d - 1)
        return e.op(op, a, b) if rng.random() < 0.7 else e.op(op, b, a)
    return itree(depth)


def run(api, programs, seed):
    """-> the summary dict; the caller has switched the specialised kernels and run-time compilation off"""
    rng = np.random.default_rng(seed)
    ran = lean = bad = failed_both = 0
    examples = []
    for pi in range(programs):
        ncols = int(rng.integers(1, 5))
        dts = [int(rng.choice([A.F64, A.F64, A.I64, A.U64])) for _ in range(ncols)]
        nb = int(rng.choice([1, 1, 2, 5]))
        lens = [int(rng.choice([1, 3, 255, 256, 257, 1024, 1025, 4096 + 17, 70001])) for _ in range(nb)]
        cols = [make_column(rng, dt, lens, float(rng.choice([0.0, 0.0, 0.1, 0.9]))) for dt in dts]
        e = A.Expr()
        try:
            nval = int(rng.integers(1, 5))
            values = []
            for _ in range(nval):
                icols = [c for c in range(ncols) if dts[c] != A.F64]
                want = "i" if (icols and rng.random() < 0.3) else "f"
                values.append(random_tree(rng, e, dts, int(rng.integers(0, 4)), want))
            pred = random_tree(rng, e, dts, int(rng.integers(0, 3)), "b") if rng.random() < 0.8 else -1
        except Exception as ex:   # a tree the expression builder refuses (mixed integer types): not a program
            continue
        out = []
        for mode in (0, 1):
            lib.set_option("interp_lean", mode)
            try:
                r = api.pipeline(e, cols, values, pred)
                out.append(("ok", r, lib.last_kernel()))
            except Exception as ex:
                out.append(("err", str(ex)[:80], lib.last_kernel()))
        ran += 1
        took_lean = "lean" in out[1][2]
        lean += took_lean
        if out[0][0] != out[1][0] or (out[0][0] == "ok" and not same(out[0][1], out[1][1])) or (out[0][0] == "err" and out[0][1] != out[1][1]):
            bad += 1
            if len(examples) < 5:
                examples.append({"program": pi, "dtypes": dts, "lens": lens, "general": str(out[0][:2])[:300], "lean": str(out[1][:2])[:300]})
        failed_both += out[0][0] == "err"
    lib.set_option("interp_lean", 1)
    return {"programs": ran, "took_the_lean_kernel": lean, "calls_that_failed_on_both (divide by zero)": failed_both, "mismatches": bad, "examples": examples, "seed": seed}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--programs", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    args = ap.parse_args()
    lib.set_device(0)
    api = lib.api()
    lib.set_option("spec", 0)
    lib.set_option("fast_filter", 0)
    lib.set_option("jit", 0)
    line = run(api, args.programs, args.seed)
    bad = line["mismatches"]
    if args.time:
        import tools.bench_kernels as bk   # noqa
        n = args.rows
        x = bk.dev_f64(n, 0, -1, 1)
        y = bk.dev_f64(n, 1, -1, 1)
        X, Y = bk.arr(x, A.F64, n), bk.arr(y, A.F64, n)
        e = A.Expr()
        cx, cy = e.col(0), e.col(1)
        gt = e.op("gt", cx, e.scalar(0.0))
        add0 = e.op("add", cx, e.scalar(0.0))
        and2 = e.op("and", gt, e.op("lt", cy, e.scalar(0.5)))
        fma = e.op("add", e.op("multiply", cx, cy), e.scalar(1.0))
        progs = {"filter_sum": (8.0, lambda: api.pipeline(e, [[X]], [add0], gt)), "filter_and2_sum": (16.0, lambda: api.pipeline(e, [[X], [Y]], [cx], and2)),
                 "sum_of_x_times_y_plus_1": (16.0, lambda: api.pipeline(e, [[X], [Y]], [fma]))}
        timing = {}
        for name, (bpr, fn) in progs.items():
            for mode, label in ((0, "general"), (2, "lean, one tile per trip"), (1, "lean")):
                lib.set_option("interp_lean", mode)
                wall, kern = bk.timed(fn, 10)
                timing[f"{name}/{label}"] = {"kernel_ms": round(kern * 1e3, 3), "frac_of_8TBps": round(bpr * n / kern / 8e12, 3), "kernel": lib.last_kernel()}
        # SINK_STORE: a computed column (16 bytes per row in and out), one of two columns (24), a predicate's bitmap (8.125)
        o64, obool = bk.out_like(A.F64, n), bk.out_like(A.BOOL, n)
        axpb = e.op("add", e.op("multiply", cx, e.scalar(2.0)), e.scalar(1.0))
        sprogs = {"store_2x_plus_1": (16.0, lambda: api.pipeline(e, [[X]], [axpb], -1, A.SINK_STORE, [[o64]])),
                  "store_x_times_y_plus_1": (24.0, lambda: api.pipeline(e, [[X], [Y]], [fma], -1, A.SINK_STORE, [[o64]])),
                  "store_predicate_x_gt_0_and_y_lt_half": (16.125, lambda: api.pipeline(e, [[X], [Y]], [and2], -1, A.SINK_STORE, [[obool]]))}
        for name, (bpr, fn) in sprogs.items():
            for mode, label in ((0, "general"), (2, "lean, one tile per trip"), (1, "lean")):
                lib.set_option("interp_lean", mode)
                wall, kern = bk.timed(fn, 10)
                timing[f"{name}/{label}"] = {"kernel_ms": round(kern * 1e3, 3), "frac_of_8TBps": round(bpr * n / kern / 8e12, 3), "kernel": lib.last_kernel()}
        lib.set_option("interp_lean", 1)
        line["rows"] = n
        line["timing"] = timing
    print(json.dumps(line))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
