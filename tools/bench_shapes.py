#!/usr/bin/env python3
"""Shape-level specialised kernels (runtime-operator trees, rdf_spec_kernel.hip.h) on HBM-resident columns: achieved
algorithmic GB/s for one- to three-level Calculate chains on every 8- and 4-byte numeric type, both sinks, with the
kernel that ran.  The numbers behind DESIGN.md's "interpreter only for exotic trees" claim (profiles/rNN_shapes.jsonl).
Usage: python tools/bench_shapes.py [--rows N] [--steps K] [--interp]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

from rust_dataframe_amd import _abi as A  # noqa: E402
from rust_dataframe_amd import lib  # noqa: E402
from bench_kernels import timed  # noqa: E402

PEAK = 8000.0
ES = {A.F64: 8, A.I64: 8, A.U64: 8, A.F32: 4, A.I32: 4, A.U32: 4, A.I16: 2, A.U16: 2}
TORCH_OF = {A.F64: torch.float64, A.I64: torch.int64, A.U64: torch.int64, A.F32: torch.float32, A.I32: torch.int32, A.U32: torch.int32, A.I16: torch.int16, A.U16: torch.int16}
NAME = {A.F64: "f64", A.I64: "i64", A.U64: "u64", A.F32: "f32", A.I32: "i32", A.U32: "u32", A.I16: "i16", A.U16: "u16"}


def column(dt, n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if dt in (A.F64, A.F32):
        t = torch.empty(n, dtype=TORCH_OF[dt], device="cuda").uniform_(-1.0, 1.0, generator=g)
    else:
        t = torch.randint(1, 1000, (n,), dtype=TORCH_OF[dt], device="cuda", generator=g)
    return A.DeviceArray(t.data_ptr(), None, 0, n, dt, 0, keep=(t,))


def out(dt, n):
    es = ES[dt]
    v = torch.empty((n + 63) // 64 * 64 * es, dtype=torch.uint8, device="cuda")
    return A.DeviceArray(v.data_ptr(), None, 0, n, dt, 0, keep=(v,))


def beyond_the_catalogs(api, n, steps):
    """Program shapes no catalog holds: four levels of arithmetic, a function inside a chain, casts below the leaves, a four-level
    i64 / f32 tree.  jit = 1: spec_kernel<Prog> compiled for the program at run time (rdf_jit.cpp; the first call's compile time
    is reported on its own); jit = 0: the general evaluator."""
    import time
    for dt in (A.F64, A.I64, A.F32):
        es = ES[dt]
        is_float = dt in (A.F64, A.F32)
        cols = [[column(dt, n, 10 * dt + i)] for i in range(4)]
        i32 = [column(A.I32, n, 777)]
        o = out(dt, n)
        e = A.Expr()
        a, b, c, d, i = (e.col(k) for k in range(5))
        k = e.scalar(1.5 if is_float else 3, dt)
        progs = [("four_levels_4col", e.op("multiply", e.op("subtract", e.op("add", e.op("multiply", a, b), c), d), e.op("add", a, k)), cols, 4 * es)]
        if is_float:
            progs.append(("function_inside_3col", e.op("add", e.op("multiply", e.op("sin", a), b), e.op("sqrt", e.op("abs", c))), cols, 3 * es))
        if dt == A.F64:
            progs.append(("casts_below_2col", e.op("multiply", e.op("add", e.cast(i, A.F64), a), e.op("subtract", a, k)), cols + [i32], es + 4))
        flt = e.op("gt", b, e.scalar(0.0 if is_float else 500.0))
        for name, root, pc, read in progs:
            for jit in (1, 0):
                lib.set_option("jit", 2 if jit else 0)
                t0 = time.perf_counter()
                api.pipeline(e, pc, [root])                 # jit = 1: compiles here
                first = time.perf_counter() - t0
                cases = [("agg", read * n, lambda: api.pipeline(e, pc, [root])),
                         ("store", (read + es) * n, lambda: api.pipeline(e, pc, [root], -1, A.SINK_STORE, [[o]])),
                         ("filter_agg", read * n, lambda: api.pipeline(e, pc, [root], flt))]
                for sink, alg, fn in cases:
                    fn()
                    wall, kern = timed(fn, steps)
                    gbs = alg / kern / 1e9 if kern > 0 else 0.0
                    print(json.dumps({"dtype": NAME[dt], "program": name, "sink": sink, "jit": jit, "rows": n, "alg_bytes": alg, "kernel_ms": round(kern * 1e3, 4),
                                      "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 3), "first_call_s": round(first, 3) if sink == "agg" else None,
                                      "kernel": lib.last_kernel()[:140]}), flush=True)
        if dt == A.F64:   # the grouped sink: 5 groups from a float column, two value expressions, behind a predicate (24 B/row)
            gid = e.cast(e.op("multiply", e.op("abs", a), e.scalar(4.999)), A.I32)
            gvals = [e.op("add", e.op("multiply", a, b), c), e.op("multiply", e.op("sin", a), b)]
            gpred = e.op("gt", c, e.scalar(-0.5))
            for jit in (1, 0):
                lib.set_option("jit", 2 if jit else 0)
                t0 = time.perf_counter()
                api.group_pipeline(e, cols[:3], gvals, gid, 5, gpred)
                first = time.perf_counter() - t0
                fn = lambda: api.group_pipeline(e, cols[:3], gvals, gid, 5, gpred)
                fn()
                wall, kern = timed(fn, steps)
                alg = 3 * es * n
                gbs = alg / kern / 1e9 if kern > 0 else 0.0
                print(json.dumps({"dtype": NAME[dt], "program": "grouped_5_groups_2_values_3col", "sink": "group", "jit": jit, "rows": n, "alg_bytes": alg, "kernel_ms": round(kern * 1e3, 4),
                                  "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 3), "first_call_s": round(first, 3), "kernel": lib.last_kernel()[:140]}), flush=True)
        lib.set_option("jit", 1)
        del cols, o, i32
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=250_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--interp", action="store_true", help="also time the general evaluator on the same programs")
    ap.add_argument("--dtypes", type=str, default="f64,i64,u64,f32,i32,u32,i16,u16")
    ap.add_argument("--programs", type=str, default="", help="comma list of program names (default: all)")
    ap.add_argument("--sinks", type=str, default="agg,store,filter_agg")
    ap.add_argument("--beyond", action="store_true", help="ONLY the programs outside the catalogs: the kernel compiled at run time against the interpreter")
    args = ap.parse_args()
    n = args.rows
    lib.set_device(0)
    api = lib.api()
    if args.beyond:
        beyond_the_catalogs(api, n, args.steps)
        return
    for dt in (A.F64, A.I64, A.U64, A.F32, A.I32, A.U32, A.I16, A.U16):
        if NAME[dt] not in args.dtypes.split(","):
            continue
        es = ES[dt]
        is_float = dt in (A.F64, A.F32)
        cols = [[column(dt, n, 10 * dt + i)] for i in range(3)]
        o = out(dt, n)
        e = A.Expr()
        a, b, c = e.col(0), e.col(1), e.col(2)
        k = e.scalar(1.0 if is_float else 1, dt)
        two = e.op("subtract", e.op("multiply", a, k), b)
        deep = e.op("subtract", e.op("multiply", e.op("add", a, b), c), k)
        q1 = e.op("multiply", e.op("multiply", a, e.op("subtract", k, b)), e.op("add", k, c))
        flt = e.op("gt", a, e.scalar(0.0 if is_float else 500.0))
        progs = [("two_level_2col", two, 2), ("three_level_left_deep_3col", deep, 3), ("three_level_q1_charge_3col", q1, 3)]
        if is_float:
            progs.append(("sin_of_two_level_2col", e.op("sin", e.op("add", e.op("multiply", a, b), k)), 2))
        for spec in ((1, 0) if args.interp else (1,)):
            lib.set_option("spec", spec)
            for name, root, nc in progs:
                if args.programs and name not in args.programs.split(","):
                    continue
                cases = [("agg", es * nc * n, lambda root=root, nc=nc: api.pipeline(e, cols[:nc], [root])),
                         ("store", es * (nc + 1) * n, lambda root=root, nc=nc: api.pipeline(e, cols[:nc], [root], -1, A.SINK_STORE, [[o]])),
                         ("filter_agg", es * nc * n, lambda root=root, nc=nc: api.pipeline(e, cols[:nc], [root], flt))]
                for sink, alg, fn in cases:
                    if sink not in args.sinks.split(","):
                        continue
                    wall, kern = timed(fn, args.steps)
                    gbs = alg / kern / 1e9 if kern > 0 else 0.0
                    print(json.dumps({"dtype": NAME[dt], "program": name, "sink": sink, "rows": n, "alg_bytes": alg, "kernel_ms": round(kern * 1e3, 4),
                                      "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 3), "kernel": lib.last_kernel()[:120]}), flush=True)
        lib.set_option("spec", 1)
        del cols, o
        torch.cuda.empty_cache()
    # operands of different types: a OP cast(b -> a's type), the plan the reference builds for unequal column types
    if "mixed" in args.dtypes.split(",") or args.dtypes == ap.get_default("dtypes"):
        for to_dt, from_dt in ((A.I64, A.I32), (A.F64, A.F32), (A.F64, A.I64), (A.I32, A.I16)):
            a_, b_ = [column(to_dt, n, 501)], [column(from_dt, n, 502)]
            o = out(to_dt, n)
            e = A.Expr()
            prog = e.op("multiply", e.col(0), e.cast(e.col(1), to_dt))
            progs = [("a_times_cast_b", prog, ES[to_dt] + ES[from_dt])]
            if to_dt == A.F64:
                progs.append(("sin_cast_b", e.op("sin", e.cast(e.col(1), to_dt)), None))
            for spec in ((1, 0) if args.interp else (1,)):
                lib.set_option("spec", spec)
                for name, root, rb in progs:
                    for sink in ("agg", "store"):
                        cols2 = [a_, b_]
                        read = rb if rb is not None else ES[from_dt]
                        alg = (read + (ES[to_dt] if sink == "store" else 0)) * n
                        fn = (lambda root=root: api.pipeline(e, cols2, [root])) if sink == "agg" else (lambda root=root: api.pipeline(e, cols2, [root], -1, A.SINK_STORE, [[o]]))
                        wall, kern = timed(fn, args.steps)
                        gbs = alg / kern / 1e9 if kern > 0 else 0.0
                        print(json.dumps({"dtype": f"{NAME[to_dt]}<-{NAME[from_dt]}", "program": name, "sink": sink, "rows": n, "alg_bytes": alg, "kernel_ms": round(kern * 1e3, 4),
                                          "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 3), "kernel": lib.last_kernel()[:120]}), flush=True)
            lib.set_option("spec", 1)
            del a_, b_, o
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
