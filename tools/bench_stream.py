"""End-to-end rate of the streamed batch loop: filter(x > 0.5) -> sum over a HOST-resident f64 frame (rdf_pipeline over
RDF_MEM_HOST arrays, rdf_capi_stream.inc), next to the link rate the same box gives a single page-locked copy.

  python tools/bench_stream.py [--gb 16] [--chunk-rows 0] [--hbm-left-gb 0]

--chunk-rows 0: one chunk per column (cut into slab pieces by the library); 1024 = the reference readers' batches (packed
through the staging buffer by host threads).  --hbm-left-gb G: hog HBM first so that only G GB are free — the frame then
does not fit and still runs (two slabs).  Prints one JSON line per variant."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rust_dataframe_amd import _abi as A   # noqa: E402
from rust_dataframe_amd import lib         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=16.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--hbm-left-gb", type=float, default=0.0)
    args = ap.parse_args()
    import torch
    lib.set_device(0)
    api = lib.api()
    L = lib.load()
    n = int(args.gb * 1e9 / 8) // 1024 * 1024
    # page-locked frame, filled on the host
    p = C.c_void_p(0)
    t0 = time.perf_counter()
    assert L.rdf_host_alloc(C.byref(p), n * 8) == 0, lib.load().rdf_last_error()
    x = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n,))
    rng = np.random.default_rng(1)
    step = 1 << 24
    for i in range(0, n, step):
        x[i:i + step] = rng.random(min(step, n - i))
    t_fill = time.perf_counter() - t0
    # link rate: one page-locked copy of 1 GiB
    d = C.c_void_p(0)
    gib = min(n * 8, 1 << 30)
    assert L.rdf_dev_alloc(C.byref(d), gib) == 0
    L.rdf_copy_h2d(d, p, gib)
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        L.rdf_copy_h2d(d, p, gib)
        rates.append(gib / (time.perf_counter() - t0) / 1e9)
    link = max(rates)
    L.rdf_dev_free(d)
    hog = []
    if args.hbm_left_gb > 0:
        free, total = torch.cuda.mem_get_info()
        want = free - int(args.hbm_left_gb * 1e9)
        while want > 0:
            sz = min(want, 32 << 30)
            q = C.c_void_p(0)
            if L.rdf_dev_alloc(C.byref(q), sz) != 0:
                break
            hog.append(q)
            want -= sz
    free_now = torch.cuda.mem_get_info()[0]
    e = A.Expr()
    c = e.col(0)
    pred = e.op("gt", c, e.scalar(0.5))
    exp_count = int((x > 0.5).sum()) if n <= 3_000_000_000 else None

    def run(name, cols, note):
        res = api.pipeline(e, cols, [c], pred)[0]          # warm-up: buffers, code objects
        ts = []
        for _ in range(args.steps):
            t0 = time.perf_counter()
            res = api.pipeline(e, cols, [c], pred)[0]
            ts.append(time.perf_counter() - t0)
        slabs, staged, direct = lib.stream_stats()
        best = min(ts)
        print(json.dumps({"bench": name, "rows": n, "bytes": n * 8, "seconds": best, "GBps_end_to_end": n * 8 / best / 1e9, "link_GBps_1GiB_pinned_copy": link,
                          "frac_of_link": n * 8 / best / 1e9 / link, "slabs": slabs, "bytes_staged": staged, "bytes_direct": direct, "count": res.count,
                          "count_ok": None if exp_count is None else bool(res.count == exp_count), "hbm_free_before_GB": free_now / 1e9, "note": note}), flush=True)
    run("stream_filter_sum_pinned_one_chunk", [[A.HostArray(x, None, 0, n, A.F64, 0)]], "page-locked column, one chunk: direct asynchronous copies per slab piece")
    cr = 1 << 20
    run("stream_filter_sum_pinned_1M_row_chunks", [[A.HostArray(x, None, i, min(cr, n - i), A.F64, 0) for i in range(0, n, cr)]], "page-locked column in 2^20-row RecordBatches")
    y = np.empty(min(n, 1 << 29))
    y[:] = x[:len(y)]
    exp_count = int((y > 0.5).sum())
    n_saved = n
    n = len(y)
    run("stream_filter_sum_pageable_one_chunk_4GiB", [[A.HostArray(y, None, 0, n, A.F64, 0)]], "pageable numpy memory: staged through page-locked buffers by host threads")
    n = n_saved
    for q in hog:
        L.rdf_dev_free(q)
    L.rdf_host_free(p)
    print(json.dumps({"fill_seconds": t_fill}), flush=True)


if __name__ == "__main__":
    main()
