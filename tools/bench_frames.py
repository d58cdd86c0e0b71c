#!/usr/bin/env python3
"""Frame-level operators on HBM-resident frames held in the reference readers' 1024-row RecordBatches (src/dataframe.rs:352):
wall time per call against the time of the kernels it launches, and the kernels' algorithmic GB/s against the 8 TB/s peak.
The point of the frame operators is the first ratio: a 1e9-row frame is 976 563 batches, and an entry point that walks a
batch list on the host pays 8-60 x its kernels (profiles/r02_kernels_1e9_microbench.jsonl).

Usage: python tools/bench_frames.py [--rows N] [--chunk-rows 1024] [--steps K] [--only name,...]
Every line carries its own `rows` (the rows the entry actually processed)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from rust_dataframe_amd import _abi as A  # noqa: E402
from rust_dataframe_amd import lib  # noqa: E402

PEAK = 8000.0
RDF_ARRAY = np.dtype([("values", "u8"), ("validity", "u8"), ("offset", "i8"), ("length", "i8"), ("null_count", "i8"), ("dtype", "i4"), ("mem", "i4")])
assert RDF_ARRAY.itemsize == C.sizeof(A.rdf_array)


def descriptors(cols, rows, chunk_rows):
    """cols: [(base pointer, element size, dtype)] -> the flat rdf_array table [col][chunk] without a Python loop per batch."""
    starts = np.arange(0, rows, chunk_rows, dtype=np.int64)
    nch = len(starts)
    tab = np.zeros(len(cols) * nch, dtype=RDF_ARRAY)
    for k, (ptr, es, dt) in enumerate(cols):
        t = tab[k * nch:(k + 1) * nch]
        t["values"] = ptr + starts * es
        t["length"] = np.minimum(chunk_rows, rows - starts)
        t["dtype"] = dt
        t["mem"] = A.MEM_DEVICE
    return tab, nch


class RawFrame(A.Frame):
    def __init__(self, api, tab, ncols, nch, keep):
        h = C.c_void_p(0)
        fn = api._fn("frame_pin")
        fn.restype = C.c_int
        api._check(fn(C.c_void_p(tab.ctypes.data), C.c_int32(ncols), C.c_int64(nch), C.byref(h)))
        super().__init__(api, h)
        self.keep = keep


def fill_f64(n, col, lo=-1.0, hi=1.0):
    t = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    lib.fill_uniform_f64(t.data_ptr(), n, 42, col, 0, lo, hi)
    return t


def fill_i64(n, col, lo, hi):
    t = torch.empty(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.fill_uniform_i64(t.data_ptr(), n, 42, col, 0, lo, hi)
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--chunk-rows", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--sort-rows", type=int, default=100_000_000)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--fused", type=int, default=1, help="rdf_set_option(\"filter_fused\") of the filter_frame_* entries: 2 forces the one-pass kernel on batches of any length")
    ap.add_argument("--lookback", type=int, default=3, help="rdf_set_option(\"filter_lookback\"): 3 = a super-tile's first tile finds the rows in front of it from tile counts + older totals (default), 2 = from totals only, 1 = every tile walks the totals (round 4)")
    ap.add_argument("--block", type=int, default=1, help="rdf_set_option(\"filter_block\"): 1 = long batches on block tiles with a scanner wave (rdf_bfilter.hip, round 6, default), 0 = wave tiles + look-back (round 5)")
    ap.add_argument("--block-rows", type=int, default=0, help="rdf_set_option(\"filter_block_rows\"): the mean batch length from which the block kernel is taken (0: the library's default)")
    ap.add_argument("--short", type=int, default=1, help="rdf_set_option(\"filter_short\"): 1 = batches no longer than a block tile on the block kernel's short-batch mode (round 6, default), 0 = the wave-tile kernel")
    ap.add_argument("--owned", type=int, default=1, help="rdf_set_option(\"filter_owned\"): 1 = many long batches, none a large share of the frame: a block draws whole batches and adds up its own offsets (round 6, default), 0 = tiles by ticket, offsets from the scanner wave")
    args = ap.parse_args()
    n, cr = args.rows, args.chunk_rows
    only = set(filter(None, args.only.split(",")))
    lib.set_device(0)
    api = lib.api()
    lib.set_option("filter_lookback", args.lookback)
    lib.set_option("filter_block", args.block)
    lib.set_option("filter_short", args.short)
    lib.set_option("filter_owned", args.owned)
    if args.block_rows > 0:
        lib.set_option("filter_block_rows", args.block_rows)

    def timed(fn, steps, warmup=2):
        torch.cuda.synchronize()
        for _ in range(warmup):
            fn()
        lib.synchronize()
        lib.kernel_timing_reset(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        lib.synchronize()
        wall = (time.perf_counter() - t0) / steps
        ms, cnt = lib.kernel_timing_get()
        lib.kernel_timing_reset(False)
        return wall, ms * 1e-3 / steps

    def report(name, rows, alg_bytes, fn, **extra):
        if only and not any(name.startswith(o) for o in only):
            return
        wall, kern = timed(fn, args.steps)
        gbs = alg_bytes / kern / 1e9 if kern > 0 else 0.0
        r = {"kernel": name, "rows": rows, "batches": (rows + cr - 1) // cr, "alg_bytes": alg_bytes, "wall_ms": round(wall * 1e3, 3), "kernel_ms": round(kern * 1e3, 3),
             "wall_over_kernel": round(wall / kern, 3) if kern > 0 else None, "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 3), "last_kernel": lib.last_kernel(), **extra}
        print(json.dumps(r), flush=True)

    x, y, z = fill_f64(n, 0), fill_f64(n, 1), fill_f64(n, 2)
    k = fill_i64(n, 3, -2 ** 31, 2 ** 31)
    lib.synchronize()
    cols4 = [(x.data_ptr(), 8, A.F64), (k.data_ptr(), 8, A.I64), (y.data_ptr(), 8, A.F64), (z.data_ptr(), 8, A.F64)]
    e = A.Expr()
    gt = e.op("gt", e.col(0), e.scalar(0.0))
    sel = 0.5

    # ---- DataFrame::filter: predicate + every column compacted, frame in / frame out
    for m in (1, 2, 4):
        tab, nch = descriptors(cols4[:m], n, cr)
        with RawFrame(api, tab, m, nch, (x, k, y, z)) as fr:
            def run():
                out = api.filter_frame(fr, e, gt)
                out.release()
            # algorithmic bytes (materialising filter of M columns): 8 M read + 8 M s written per row
            lib.set_option("filter_fused", args.fused)
            report(f"filter_frame_{m}col", n, (8 + 8 * sel) * m * n, run, selectivity=sel, path="one pass: predicate inside the compaction kernel")
            lib.set_option("filter_fused", 0)
            report(f"filter_frame_{m}col_three_pass", n, (8 + 8 * sel) * m * n, run, selectivity=sel, path="predicate -> mask, count, compact (round 3)")
            lib.set_option("filter_fused", 1)
    # ---- the same over frames of TWO column widths (f64 predicate column + 4-byte columns beside it: a lineitem-shaped frame; round 6:
    #      the block kernel twice over the same tiles, the second launch by the mask the first one wrote) against the wave-tile kernel
    q = (k & 0x7FFFFFFF).to(torch.int32)
    r32 = x.to(torch.float32)
    lib.synchronize()
    for name, colsm in (("mixed_f64_i32", [(x.data_ptr(), 8, A.F64), (q.data_ptr(), 4, A.I32)]),
                        ("mixed_f64_i64_i32_f32", [(x.data_ptr(), 8, A.F64), (k.data_ptr(), 8, A.I64), (q.data_ptr(), 4, A.I32), (r32.data_ptr(), 4, A.F32)])):
        if only and not any(f"filter_frame_{name}".startswith(o) for o in only):
            continue
        tab, nch = descriptors(colsm, n, cr)
        row_bytes = sum(es for _, es, _ in colsm)
        with RawFrame(api, tab, len(colsm), nch, (x, k, q, r32)) as fr:
            def run():
                out = api.filter_frame(fr, e, gt)
                out.release()
            lib.set_option("filter_fused", args.fused)
            for mixed in (2, 0):
                lib.set_option("filter_mixed", mixed)
                report(f"filter_frame_{name}" + ("" if mixed else "_wave_tiles"), n, (1 + sel) * row_bytes * n + (0.25 * n if mixed else 0), run, selectivity=sel,
                       path="block kernel x 2 (predicate's width, then the other width by the mask)" if mixed else "wave-tile kernel (any widths)")
            lib.set_option("filter_mixed", 1)
            lib.set_option("filter_fused", 1)
    # ---- frames of 4-byte columns only (f32 predicate column)
    q2 = (q >> 3).contiguous()
    r2 = (r32 * 0.5).contiguous()
    gt32 = e.op("gt", e.col(0), e.scalar(0.0))
    for m in (1, 2, 4):
        name = f"filter_frame_narrow_{m}col"
        if only and not any(name.startswith(o) for o in only):
            continue
        colsn = [(r32.data_ptr(), 4, A.F32), (q.data_ptr(), 4, A.I32), (r2.data_ptr(), 4, A.F32), (q2.data_ptr(), 4, A.I32)][:m]
        tab, nch = descriptors(colsn, n, cr)
        with RawFrame(api, tab, m, nch, (r32, q, r2, q2)) as fr:
            def run():
                out = api.filter_frame(fr, e, gt32)
                out.release()
            lib.set_option("filter_fused", args.fused)
            report(name, n, (4 + 4 * sel) * m * n, run, selectivity=sel, path="one pass, 4-byte columns")
            lib.set_option("filter_fused", 1)
    del q, r32, q2, r2
    # ---- DataFrame::take: random / sequential indices, every column in one gather pass
    nidx = n // 4
    ridx = torch.randint(0, n, (nidx,), dtype=torch.int64, device="cuda").to(torch.uint32)
    sidx = torch.arange(0, nidx, dtype=torch.int64, device="cuda").to(torch.uint32)
    RI = A.DeviceArray(ridx.data_ptr(), None, 0, nidx, A.U32, 0, keep=ridx)
    SI = A.DeviceArray(sidx.data_ptr(), None, 0, nidx, A.U32, 0, keep=sidx)
    for m in (1, 2, 4):
        tab, nch = descriptors(cols4[:m], n, cr)
        with RawFrame(api, tab, m, nch, (x, k, y, z)) as fr:
            def run_r():
                api.take_frame(fr, RI).release()
            def run_s():
                api.take_frame(fr, SI).release()
            report(f"take_frame_random_{m}col", nidx, (4 + 16 * m) * nidx, run_r, source_rows=n)
            report(f"take_frame_sequential_{m}col", nidx, (4 + 16 * m) * nidx, run_s, source_rows=n)
    # the same column as ONE chunk (no row -> batch lookup)
    for m in (1, 4):
        tab, nch = descriptors(cols4[:m], n, n)
        with RawFrame(api, tab, m, nch, (x, k, y, z)) as fr:
            def run_r():
                api.take_frame(fr, RI).release()
            report(f"take_frame_random_{m}col_one_chunk", nidx, (4 + 16 * m) * nidx, run_r, source_rows=n)
    del ridx, sidx
    # ---- DataFrame::sort: i64 key, then every column taken by the order
    ns = min(n, args.sort_rows)
    for m, cols in ((1, [cols4[1]]), (4, [cols4[1], cols4[0], cols4[2], cols4[3]])):
        tab, nch = descriptors(cols, ns, cr)
        with RawFrame(api, tab, m, nch, (x, k, y, z)) as fr:
            def run():
                sf, _ = api.sort_frame(fr, [0], [False])
                sf.release()
            report(f"sort_frame_i64_key_{m}col", ns, (8.0 + 16.0 * m) * ns, run)
            def run_i():
                api.sort_frame(fr, [0], [False], out_indices=oi, want_frame=False)
            ib = torch.empty(ns * 4 + 64, dtype=torch.uint8, device="cuda")
            oi = A.DeviceArray(ib.data_ptr(), None, 0, ns, A.U32, 0, keep=ib, capacity=ns)
            if m == 1:
                report("sort_frame_i64_key_indices_only", ns, 8.0 * ns, run_i)
    # ---- GroupAggregate: one i64 key column, sum of an f64 column
    for groups in (1_000_000, 1000):
        kk = fill_i64(n, 7, 0, groups)
        lib.synchronize()
        tab, nch = descriptors([(kk.data_ptr(), 8, A.I64), (x.data_ptr(), 8, A.F64)], n, cr)
        with RawFrame(api, tab, 2, nch, (kk, x)) as fr:
            def run():
                api.groupby_agg_frame(fr, [0], 1, "sum", groups).release()
            report(f"groupby_agg_frame_sum_{groups}_groups", n, 16.0 * n, run)
        del kk


if __name__ == "__main__":
    main()
