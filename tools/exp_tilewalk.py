#!/usr/bin/env python3
"""Round 5 A/B of the specialised kernels' launch shape on data built once per configuration: resident blocks per CU x tile walk
(plain grid stride / rotated rows / XCD-contiguous) for the headline, the headline with a validity bitmap, the headline over
1024-row batches (pinned frame), C3 (4 columns) and a two-column store; kernel time from the library's HIP events.

    python tools/exp_tilewalk.py [--rows 1000000000] [--steps 20] [--only headline,validity,batches,c3,store,q1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from rust_dataframe_amd import _abi as A  # noqa: E402
from rust_dataframe_amd import lib  # noqa: E402

SEED = 42


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    lib.synchronize()
    lib.kernel_timing_reset(True)
    for _ in range(steps):
        fn()
    lib.synchronize()
    ms, n = lib.kernel_timing_get()
    lib.kernel_timing_reset(False)
    return ms / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--variants", type=str, default="")
    a = ap.parse_args()
    only = set(filter(None, a.only.split(",")))
    n = a.rows
    lib.set_device(0)
    api = lib.api()

    def f64(col, lo=0.0, hi=1.0, rows=n):
        t = torch.empty(rows, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        lib.fill_uniform_f64(t.data_ptr(), rows, SEED, col, 0, lo, hi)
        return t

    def i64(col, lo, hi, rows=n):
        t = torch.empty(rows, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        lib.fill_uniform_i64(t.data_ptr(), rows, SEED, col, 0, lo, hi)
        return t

    # (blocks per CU, tile_rot, xcd_swz, grid_adj); 0 / -1 / -1 / 0 = what run_program picks for the program
    variants = [(0, -1, -1, 0), (3, 0, 0, 0), (4, 0, 0, 0), (5, 0, 0, 0), (6, 0, 0, 0), (8, 0, 0, 0),
                (4, 37, 0, 0), (8, 37, 0, 0), (4, 1, 0, 0), (8, 1, 0, 0), (3, 37, 0, 0),
                (4, 0, 1, 0), (8, 0, 1, 0), (3, 0, 1, 0), (8, 37, 1, 0),
                (4, 0, 0, -1), (8, 0, 0, -1), (8, 0, 0, -3)]
    if a.variants:
        variants = [tuple(int(x) for x in v.split(":")) for v in a.variants.split(",")]

    def opt(k, v):
        try:
            lib.set_option(k, v)
        except Exception:      # a build of an earlier round (RDF_LIB_PATH): no such option
            if v:
                raise

    def sweep(name, alg_bytes, fn, vs=None):
        for b, rot, swz, adj in (vs or variants):
            lib.set_option("spec_blocks_per_cu", b)
            opt("spec_tile_rot", rot)
            opt("spec_xcd_swz", swz)
            opt("spec_grid_adj", adj)
            ms = timed(fn, a.steps)
            print(json.dumps({"exp": "tilewalk", "config": name, "rows": n, "blocks_per_cu": b, "tile_rot": rot, "xcd_swz": swz, "grid_adj": adj,
                              "kernel_ms": round(ms, 4), "frac_of_8TBps": round(alg_bytes / ms / 1e6 / 8000.0, 4), "kernel": lib.last_kernel()}), flush=True)
        for k, v in (("spec_blocks_per_cu", 0), ("spec_tile_rot", -1), ("spec_xcd_swz", -1), ("spec_grid_adj", 0)):
            try:
                lib.set_option(k, v)
            except Exception:
                pass

    e = A.Expr()
    c = e.col(0)
    pred = e.op("gt", c, e.scalar(0.5))
    if not only or only & {"headline", "validity", "batches"}:
        x = f64(0)
        lib.synchronize()
        if not only or "headline" in only:
            fr = A.Prepared([[A.DeviceArray(x.data_ptr(), None, 0, n, A.F64, -1, keep=x)]])
            sweep("headline", 8.0 * n, lambda: api.pipeline(e, fr, [c], pred))
        if not only or "validity" in only:
            v = torch.zeros((n + 63) // 64 * 8 + 64, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            lib.fill_validity(v.data_ptr(), n, SEED, 0, 0, 0.1)
            lib.synchronize()
            fr = A.Prepared([[A.DeviceArray(x.data_ptr(), v.data_ptr(), 0, n, A.F64, -1, keep=(x, v))]])
            sweep("validity_10pct", 8.125 * n, lambda: api.pipeline(e, fr, [c], pred))
            del v
        if not only or "batches" in only:
            col = [A.DeviceArray(x.data_ptr() + 8 * i, None, 0, min(1024, n - i), A.F64, -1, keep=x) for i in range(0, n, 1024)]
            fr = A.PinnedFrame(api, [col])
            sweep("batches_1024", 8.0 * n, lambda: api.pipeline(e, fr, [c], pred))
            del fr, col
        del x
        torch.cuda.empty_cache()
    if not only or "c3" in only:
        ts = [f64(cid, -1.0, 1.0) for cid in range(3)] + [i64(3, -2 ** 31, 2 ** 31)]
        lib.synchronize()
        dts = (A.F64, A.F64, A.F64, A.I64)
        cols = [[A.DeviceArray(t.data_ptr(), None, 0, n, dt, 0, keep=t)] for t, dt in zip(ts, dts)]
        e3 = A.Expr()
        fma = e3.op("add", e3.op("multiply", e3.col(0), e3.col(1)), e3.col(2))
        fr = A.Prepared(cols)
        sweep("c3", 32.0 * n, lambda: api.pipeline(e3, fr, [fma, e3.col(3)]))
        del ts, cols, fr
        torch.cuda.empty_cache()
    if not only or "store" in only:
        xa, xb = f64(0), f64(1)
        pad = (n + 63) // 64 * 64
        ov = torch.empty(pad * 8, dtype=torch.uint8, device="cuda")
        out = A.DeviceArray(ov.data_ptr(), None, 0, n, A.F64, 0, keep=ov)
        lib.synchronize()
        A1 = [A.DeviceArray(xa.data_ptr(), None, 0, n, A.F64, 0, keep=xa)]
        B1 = [A.DeviceArray(xb.data_ptr(), None, 0, n, A.F64, 0, keep=xb)]
        sweep("add_store", 24.0 * n, lambda: api.binary("add", A1, B1, [out]),
              vs=[(0, -1, -1, 0), (7, 0, 0, 0), (8, 0, 0, 0), (5, 0, 0, 0), (8, 37, 0, 0), (8, 0, 1, 0), (7, 0, 1, 0), (8, 0, 0, -1)])
        del xa, xb, ov
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
