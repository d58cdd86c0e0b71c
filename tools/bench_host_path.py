#!/usr/bin/env python3
"""The drop-in path with HOST buffers (what the Rust shim hands over): PCIe-inclusive throughput of a few entry points.
Usage: python tools/bench_host_path.py [--rows N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from rust_dataframe_amd import _abi as A  # noqa: E402
from rust_dataframe_amd import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    n = args.rows
    lib.set_device(0)
    api = lib.api()
    rng = np.random.default_rng(1)
    x, y = rng.uniform(size=n), rng.uniform(size=n)
    for chunk in (n, 1 << 20, 1024):
        if chunk == 1024 and n > 10_000_000:
            xs, ys = x[:10_000_000], y[:10_000_000]
        else:
            xs, ys = x, y
        m = len(xs)
        X = [A.HostArray.from_numpy(xs[i:i + chunk]) for i in range(0, m, chunk)]
        Y = [A.HostArray.from_numpy(ys[i:i + chunk]) for i in range(0, m, chunk)]
        e = A.Expr()
        c = e.col(0)
        pred = e.op("gt", c, e.scalar(0.5))
        outs = [A.HostArray.empty_out(A.F64, a.length, False) for a in X]

        def run(name, fn, nbytes):
            fn()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            dt = (time.perf_counter() - t0) / args.steps
            print(json.dumps({"entry": name, "rows": m, "chunk_rows": chunk, "ms": round(dt * 1e3, 2), "rows_per_s": round(m / dt), "host_GBps": round(nbytes / dt / 1e9, 2)}), flush=True)
        run("pipeline filter->sum (8 B/row in)", lambda: api.pipeline(e, [X], [c], pred), 8.0 * m)
        run("binary add (16 B/row in, 8 out)", lambda: api.binary("add", X, Y, outs), 24.0 * m)
        # the same add with the ctypes descriptor arrays built once: the library's own time, without Python marshalling
        import ctypes as C
        ca, cb = A._flat([X], len(X)), A._flat([Y], len(Y))
        co = (A.rdf_out * len(outs))(*[o.out_struct() for o in outs])
        fn = api._fn("binary")
        run("binary add, descriptors prebuilt", lambda: api._check(fn(C.c_int32(A.OP_NAMES["add"]), ca, cb, C.c_int64(len(X)), co)), 24.0 * m)


if __name__ == "__main__":
    main()
