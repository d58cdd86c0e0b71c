#!/usr/bin/env python3
"""The two-number memory model of DESIGN.md §4.1 fitted and checked on ONE box in ONE process: the bare-stream probes
(rdf_probe_stream: read, copy, two reads + one write) give r and w, then the streaming kernels of the library run on the same
device and their HIP-event times are held against  bytes_read / r + bytes_written / w.  Run it under
`rocprofv3 --kernel-trace --stats` (tools/profile_round.sh does) to get the per-kernel split of the multi-kernel operators
(the hash GROUP BY's scatter and aggregate passes) from the same process.
Usage: python tools/model_check.py [--rows 1000000000] > model.jsonl"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from rust_dataframe_amd import _abi as A  # noqa: E402
from rust_dataframe_amd import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    n = args.rows
    lib.set_device(0)
    api = lib.api()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    z = torch.empty(n, dtype=torch.float64, device="cuda")
    k = torch.empty(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.fill_uniform_f64(x.data_ptr(), n, 42, 0, 0, 0.0, 1.0)
    lib.fill_uniform_f64(y.data_ptr(), n, 42, 1, 0, 0.0, 1.0)
    lib.fill_uniform_i64(k.data_ptr(), n, 42, 7, 0, 0, 1_000_000)
    lib.synchronize()
    probes = {}
    for kind, name in ((0, "read"), (1, "copy"), (2, "two_reads_one_write")):
        g, shape = lib.probe_stream(kind, x.data_ptr(), y.data_ptr(), z.data_ptr(), n * 8, 7)
        probes[name] = g
        print(json.dumps({"probe": name, "GBps_of_bytes_moved": round(g, 1), "shape": shape}), flush=True)
    r = probes["read"]
    w = 1.0 / (2.0 / probes["copy"] - 1.0 / r)
    print(json.dumps({"model": "time = bytes_read / r + bytes_written / w", "r_GBps": round(r, 1), "w_GBps": round(w, 1),
                      "two_reads_one_write_predicted_GBps": round(3.0 / (2.0 / r + 1.0 / w), 1), "two_reads_one_write_measured_GBps": round(probes["two_reads_one_write"], 1)}), flush=True)
    lib.fill_uniform_f64(y.data_ptr(), n, 42, 1, 0, 0.0, 1.0)

    def timed(fn):
        for _ in range(2):
            fn()
        lib.synchronize()
        lib.kernel_timing_reset(True)
        for _ in range(args.steps):
            fn()
        lib.synchronize()
        ms, _ = lib.kernel_timing_get()
        lib.kernel_timing_reset(False)
        return ms / args.steps

    def report(name, rd, wr, fn, note=""):
        ms = timed(fn)
        pred = (rd / r + wr / w) * 1e-6
        print(json.dumps({"kernel": name, "GB_read": rd / 1e9, "GB_written": wr / 1e9, "predicted_ms": round(pred, 3), "measured_ms": round(ms, 3),
                          "measured_over_predicted": round(ms / pred, 3), "last_kernel": lib.last_kernel(), "note": note}), flush=True)

    X = A.DeviceArray(x.data_ptr(), None, 0, n, A.F64, 0, keep=x)
    Y = A.DeviceArray(y.data_ptr(), None, 0, n, A.F64, 0, keep=y)
    K = A.DeviceArray(k.data_ptr(), None, 0, n, A.I64, 0, keep=k)
    e = A.Expr()
    gt = e.op("gt", e.col(0), e.scalar(0.5))
    report("headline filter(x > 0.5) -> sum", 8.0 * n, 0, lambda: api.pipeline(e, [[X]], [e.col(0)], gt))
    e2 = A.Expr()
    add = e2.op("add", e2.col(0), e2.col(1))
    ob = torch.empty(n * 8 + 64, dtype=torch.uint8, device="cuda")
    O = A.DeviceArray(ob.data_ptr(), None, 0, n, A.F64, 0, keep=ob, capacity=n)
    report("add -> new column", 16.0 * n, 8.0 * n, lambda: api.pipeline(e2, [[X], [Y]], [add], -1, A.SINK_STORE, [[O]]))
    # DataFrame::filter on ONE batch (block tiles): frames of 1, 2, 4 columns
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_frames import RawFrame, descriptors  # noqa: E402
    cols4 = [(x.data_ptr(), 8, A.F64), (k.data_ptr(), 8, A.I64), (y.data_ptr(), 8, A.F64), (z.data_ptr(), 8, A.F64)]
    kept = api.pipeline(e, [[X]], [e.col(0)], gt)[0].count
    for m in (1, 2, 4):
        tab, nch = descriptors(cols4[:m], n, n)
        with RawFrame(api, tab, m, nch, (x, k, y, z)) as fr:
            def run():
                out = api.filter_frame(fr, e, gt)
                out.release()
            report(f"DataFrame::filter, one batch, {m} column(s)", 8.0 * m * n, 8.0 * m * kept, run)
    # hash GROUP BY, 1e6 keys (C4): scatter 16 R + 16 W, aggregate 16 R (the split comes from the profiler)
    cap = 1_000_002
    bufs = [torch.empty(cap * 8 + 64, dtype=torch.uint8, device="cuda") for _ in range(3)]
    outs = tuple(A.DeviceArray(b.data_ptr(), None, 0, cap, dt, 0, keep=b) for b, dt in zip(bufs, (A.I64, A.F64, A.I64)))
    report("hash GROUP BY 1e6 keys (C4): scatter + aggregate", 32.0 * n, 16.0 * n, lambda: api.groupby_sum([K], [X], 1_000_000, outs),
           note="scatter 16 R + 16 W then aggregate 16 R")


if __name__ == "__main__":
    main()
