#!/usr/bin/env python3
"""C4 (hash GROUP BY, 1e9 rows x 1e6 keys): what would running scatter -> aggregate per WINDOW of rows buy?

The judge's round-4 proposal: a window of 4-8 M rows leaves 64-128 MB of 16-byte records, which stay in the 256 MiB Infinity
Cache between the scatter pass that writes them and the aggregate pass that reads them.  This measures it with the REAL kernels
before anything is rebuilt around it: the same rdf_groupby_agg call over consecutive row windows of the same columns (the
library's pooled scratch gives every call the same record buffer, i.e. a fixed slab), kernel time summed over the windows
(HIP events around every kernel; host gaps between calls are not counted), against one call over all rows.  A windowed call emits
its groups every time (24 MB written per window) — the price a real windowed design pays for writing / reloading its partition
tables, so the model is faithful to the traffic, only the results are per window.

    python tools/exp_gb_window.py [--rows 1000000000] [--groups 1000000] [--windows 4,8,16,32,64]   (window sizes in M rows)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--groups", type=int, default=1_000_000)
    ap.add_argument("--windows", type=str, default="4,8,16,32,64")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    n, ng = a.rows, a.groups
    lib.set_device(0)
    api = lib.api()
    kk = torch.empty(n, dtype=torch.int64, device="cuda")
    xx = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    lib.fill_uniform_i64(kk.data_ptr(), n, 42, 7, 0, 0, ng)
    lib.fill_uniform_f64(xx.data_ptr(), n, 42, 1, 0, 0.0, 1.0)
    lib.synchronize()

    def out(dt, m):
        t = torch.empty((m + 64) * 8, dtype=torch.uint8, device="cuda")
        return A.DeviceArray(t.data_ptr(), None, 0, m, dt, 0, keep=(t,))
    ok_, os_, oc_ = out(A.I64, ng + 2), out(A.F64, ng + 2), out(A.I64, ng + 2)

    def window(lo, hi):
        m = hi - lo
        return (A.DeviceArray(kk.data_ptr() + 8 * lo, None, 0, m, A.I64, -1, keep=(kk,)),
                A.DeviceArray(xx.data_ptr() + 8 * lo, None, 0, m, A.F64, -1, keep=(xx,)))

    def run(wrows):
        wins = [window(lo, min(n, lo + wrows)) for lo in range(0, n, wrows)]
        best = None
        for _ in range(a.reps + 1):
            lib.synchronize()
            lib.kernel_timing_reset(True)
            for K, X in wins:
                api.groupby_sum([K], [X], ng, (ok_, os_, oc_))
            lib.synchronize()
            ms, cnt = lib.kernel_timing_get()
            lib.kernel_timing_reset(False)
            best = ms if best is None else min(best, ms)
        return best, len(wins)

    whole, _ = run(n)
    print(json.dumps({"exp": "gb_window", "rows": n, "groups": ng, "window_rows": n, "windows": 1, "kernel_ms": round(whole, 3)}), flush=True)
    for w in [int(x) for x in a.windows.split(",") if x]:
        ms, nw = run(w * 1_000_000)
        print(json.dumps({"exp": "gb_window", "rows": n, "groups": ng, "window_rows": w * 1_000_000, "windows": nw, "kernel_ms": round(ms, 3),
                          "vs_whole": round(ms / whole, 3), "record_MB_per_window": round(w * 16.0, 1)}), flush=True)


if __name__ == "__main__":
    main()
