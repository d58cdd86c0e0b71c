"""End-to-end rates of the streamed batch loop for the sinks that MATERIALISE (round 5, rdf_capi_stream.inc): host-resident frame in,
host-resident result out, next to what one page-locked copy reaches in either direction on the same box.

  python tools/bench_stream_sinks.py [--gb 16] [--hbm-left-gb 0] [--only store,filter,q1,groupby]

  store    sin(x + 1) -> new column, host -> host (rdf_pipeline, SINK_STORE): bytes in + bytes out, concurrently
  filter   DataFrame::filter(x > 0.5) over an (f64, i64) frame (rdf_filter_pipeline): kept rows of both columns back on the host
  q1       TPC-H Q1 shape over a host-resident lineitem (rdf_group_pipeline): 38 B/row in, 6 groups out
  groupby  hash GROUP BY key -> sum(val), 1e6 keys (rdf_groupby_agg): partial groups merged slab by slab
Prints one JSON line per sink."""
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
    ap.add_argument("--gb", type=float, default=16.0, help="bytes of the largest input frame")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--hbm-left-gb", type=float, default=0.0)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--probe-only", action="store_true", help="measure the link (each direction alone, both at once) and stop")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    import torch
    lib.set_device(0)
    api = lib.api()
    L = lib.load()

    def pinned(nbytes):
        p = C.c_void_p(0)
        assert L.rdf_host_alloc(C.byref(p), nbytes) == 0, L.rdf_last_error()
        return p

    def view(p, dtype, n):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,))

    # link rates: one page-locked copy of 1 GiB, each direction
    gib = 1 << 30
    hp, d = pinned(gib), C.c_void_p(0)
    assert L.rdf_dev_alloc(C.byref(d), gib) == 0
    up, dn = [], []
    for _ in range(4):
        t0 = time.perf_counter(); L.rdf_copy_h2d(d, hp, gib); up.append(gib / (time.perf_counter() - t0) / 1e9)
        t0 = time.perf_counter(); L.rdf_copy_d2h(hp, d, gib); dn.append(gib / (time.perf_counter() - t0) / 1e9)
    link_up, link_dn = max(up[1:]), max(dn[1:])
    # ... and both directions at once: asynchronous copies of 1 GiB on two streams, six per direction queued back to back, HIP events
    # around each direction's queue (blocking copies issued from two threads take turns — 28 GB/s each way, measured: not the link)
    hp2, d2 = pinned(gib), C.c_void_p(0)
    assert L.rdf_dev_alloc(C.byref(d2), gib) == 0
    # (torch's own page-locked buffers: its non_blocking copies are asynchronous only from memory ITS allocator pinned)
    t_up = torch.empty(gib, dtype=torch.uint8, pin_memory=True)
    t_dn = torch.empty(gib, dtype=torch.uint8, pin_memory=True)
    g_up = torch.empty(gib, dtype=torch.uint8, device="cuda")
    g_dn = torch.empty(gib, dtype=torch.uint8, device="cuda")
    s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
    reps, both, alone_async = 6, [], []
    for trial in range(4):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        with torch.cuda.stream(s_up):
            ev[0].record()
            for _r in range(reps):
                g_up.copy_(t_up, non_blocking=True)
            ev[1].record()
        if trial > 0:                                  # trial 0: the upload alone, as a check of the method against link_up
            with torch.cuda.stream(s_dn):
                ev[2].record()
                for _r in range(reps):
                    t_dn.copy_(g_dn, non_blocking=True)
                ev[3].record()
        torch.cuda.synchronize()
        if trial == 0:
            alone_async.append(reps * gib / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9)
        else:
            both.append((reps * gib / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9, reps * gib / (ev[2].elapsed_time(ev[3]) * 1e-3) / 1e9))
    del g_up, g_dn, t_up, t_dn
    link_both = min(max(b[0] for b in both), max(b[1] for b in both))
    print(json.dumps({"probe": "link", "up_alone_GBps": link_up, "down_alone_GBps": link_dn, "up_alone_async_GBps": alone_async[0],
                      "both_at_once_up_down_GBps": both}), file=sys.stderr, flush=True)
    L.rdf_dev_free(d2)
    L.rdf_dev_free(d); L.rdf_host_free(hp); L.rdf_host_free(hp2)
    if args.probe_only:
        return
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
    rng = np.random.default_rng(1)

    def fill(v, fn):
        step = 1 << 24
        for i in range(0, len(v), step):
            v[i:i + step] = fn(min(step, len(v) - i))

    def report(name, in_bytes, out_bytes, seconds, extra):
        slabs, staged, direct = lib.stream_stats()
        print(json.dumps(dict({"bench": name, "bytes_in": in_bytes, "bytes_out": out_bytes, "seconds": seconds,
                               "GBps_in": in_bytes / seconds / 1e9, "GBps_out": out_bytes / seconds / 1e9,
                               "link_up_GBps": link_up, "link_down_GBps": link_dn, "link_each_way_when_both_run_GBps": link_both, "frac_of_link_in": in_bytes / seconds / 1e9 / link_up,
                               "frac_of_link_out": out_bytes / seconds / 1e9 / link_dn, "slabs": slabs, "bytes_staged": staged, "bytes_direct": direct,
                               "hbm_free_before_GB": free_now / 1e9}, **extra)), flush=True)

    def best_of(fn):
        fn()
        ts = []
        for _ in range(args.steps):
            t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
        return min(ts), r

    if not only or "store" in only:
        n = int(args.gb * 1e9 / 8) // 1024 * 1024
        px, py = pinned(n * 8), pinned(n * 8)
        x, y = view(px, np.float64, n), view(py, np.float64, n)
        fill(x, lambda m: rng.random(m))
        cr = 1 << 20
        cols = [[A.HostArray(x, None, i, min(cr, n - i), A.F64, 0) for i in range(0, n, cr)]]
        outs = [[A.HostArray(y[i:i + cr], None, 0, min(cr, n - i), A.F64, 0) for i in range(0, n, cr)]]
        e = A.Expr()
        prog = e.op("sin", e.op("add", e.col(0), e.scalar(1.0)))
        sec, _ = best_of(lambda: api.pipeline(e, cols, [prog], sink=A.SINK_STORE, outs=outs))
        i0 = rng.integers(0, n - 1000)
        ok = bool(np.allclose(y[i0:i0 + 1000], np.sin(x[i0:i0 + 1000] + 1.0), rtol=1e-12) and np.allclose(y[-1000:], np.sin(x[-1000:] + 1.0), rtol=1e-12))
        report("stream_store_sin_x_plus_1_pinned_1M_row_batches", n * 8, n * 8, sec, {"rows": n, "check": ok, "note": "page-locked input and output columns: direct DMA both ways, H2D of slab k+1 | kernel k | D2H of slab k-1"})
        m = min(n, 1 << 28)
        xs, ys = np.array(x[:m]), np.empty(m)
        colsp = [[A.HostArray(xs, None, 0, m, A.F64, 0)]]
        outsp = [[A.HostArray(ys, None, 0, m, A.F64, 0)]]
        sec, _ = best_of(lambda: api.pipeline(e, colsp, [prog], sink=A.SINK_STORE, outs=outsp))
        report("stream_store_sin_x_plus_1_pageable_2GiB", m * 8, m * 8, sec, {"rows": m, "check": bool(np.allclose(ys[:1000], np.sin(xs[:1000] + 1.0), rtol=1e-12)), "note": "pageable numpy memory in and out: staged through page-locked buffers by host threads, both ways"})
        del xs, ys
        L.rdf_host_free(px); L.rdf_host_free(py)
    if not only or "filter" in only:
        n = int(args.gb * 1e9 / 16) // 1024 * 1024
        px, pk, ox, ok_ = pinned(n * 8), pinned(n * 8), pinned(n * 8), pinned(n * 8)
        x, k = view(px, np.float64, n), view(pk, np.int64, n)
        fill(x, lambda m: rng.random(m))
        fill(k, lambda m: rng.integers(-2 ** 40, 2 ** 40, m))
        yx, yk = view(ox, np.float64, n), view(ok_, np.int64, n)
        n_all = n
        for cr, tag in ((1 << 20, "1M_row_batches"), (1024, "1024_row_batches")):
            n = n_all if cr > 1024 else min(n_all, 1 << 27)          # (the readers' batches: a 2 GiB frame — a million descriptors per column are marshalled once, outside the timed call)
            cols = [[A.HostArray(x, None, i, min(cr, n - i), A.F64, 0) for i in range(0, n, cr)], [A.HostArray(k, None, i, min(cr, n - i), A.I64, 0) for i in range(0, n, cr)]]
            outs = [[A.HostArray(yx[i:i + cr], None, 0, min(cr, n - i), A.F64, 0) for i in range(0, n, cr)], [A.HostArray(yk[i:i + cr], None, 0, min(cr, n - i), A.I64, 0) for i in range(0, n, cr)]]
            e = A.Expr()
            pred = e.op("gt", e.col(0), e.scalar(0.5))
            nch = len(cols[0])
            cc = A._flat(cols, nch)
            flat = [o for col in outs for o in col]
            carr = (A.rdf_out * len(flat))(*[o.out_struct() for o in flat])
            nodes = e.c_array()
            fn = api._fn("filter_pipeline")
            fn.restype = C.c_int

            def call():
                api._check(fn(nodes, C.c_int32(len(e.nodes)), C.c_int32(pred), cc, C.c_int32(2), C.c_int64(nch), carr))
                return carr
            sec, got = best_of(call)
            kept = sum(got[i].length for i in range(nch))
            c0 = x[:cr] > 0.5
            ok = bool(kept == sum(got[nch + i].length for i in range(nch)) and got[0].length == int(c0.sum()) and np.array_equal(yx[:got[0].length], x[:cr][c0]) and np.array_equal(yk[:got[nch].length], k[:cr][c0]))
            report("stream_filter_pipeline_2col_pinned_" + tag, n * 16, kept * 16, sec, {"rows": n, "kept": kept, "check": ok, "note": "DataFrame::filter(x > 0.5) over (f64, i64): predicate + compaction on the device slab by slab, kept rows unpacked into the caller's batches by host threads"})
        n = n_all
        for p in (px, pk, ox, ok_):
            L.rdf_host_free(p)
    if not only or "q1" in only:
        n = int(args.gb * 1e9 / 38) // 1024 * 1024
        ps = [pinned(n * 8) for _ in range(4)] + [pinned(n), pinned(n), pinned(n * 4)]
        qty, price, disc, tax = (view(p, np.float64, n) for p in ps[:4])
        flag, status, ship = view(ps[4], np.int8, n), view(ps[5], np.int8, n), view(ps[6], np.int32, n)
        fill(qty, lambda m: rng.integers(1, 51, m).astype(np.float64)); fill(price, lambda m: rng.uniform(900.0, 105000.0, m))
        fill(disc, lambda m: rng.integers(0, 11, m) / 100.0); fill(tax, lambda m: rng.integers(0, 9, m) / 100.0)
        fill(flag, lambda m: rng.integers(0, 3, m).astype(np.int8)); fill(status, lambda m: rng.integers(0, 2, m).astype(np.int8))
        fill(ship, lambda m: rng.integers(8036, 10562, m).astype(np.int32))
        cr = 1 << 20
        dts = (A.F64, A.F64, A.F64, A.F64, A.I8, A.I8, A.I32)
        cols = [[A.HostArray(v, None, i, min(cr, n - i), dt, 0) for i in range(0, n, cr)] for v, dt in zip((qty, price, disc, tax, flag, status, ship), dts)]
        e = A.Expr()
        c = [e.col(i) for i in range(7)]
        pred = e.op("le", c[6], e.scalar(10471, A.I32))
        gid = e.op("add", e.op("multiply", e.cast(c[4], A.I32), e.scalar(2, A.I32)), e.cast(c[5], A.I32))
        dp = e.op("multiply", c[1], e.op("subtract", e.scalar(1.0), c[2]))
        ch = e.op("multiply", dp, e.op("add", e.scalar(1.0), c[3]))
        sec, (res, rows) = best_of(lambda: api.group_pipeline(e, cols, [c[0], c[1], dp, ch, c[2]], gid, 6, pred))
        keep = ship[:cr] <= 10471
        ok = bool(abs(sum(rows[:6]) / n - (10471 - 8036 + 1) / (10562 - 8036)) < 1e-3)
        report("stream_q1_group_pipeline_pinned_1M_row_batches", n * 38, 0, sec, {"rows": n, "rows_per_s": n / sec, "check": ok, "note": "TPC-H Q1 shape over a host-resident lineitem: 7 columns in, 6 groups out"})
        for p in ps:
            L.rdf_host_free(p)
    if not only or "groupby" in only:
        n = int(args.gb * 1e9 / 16) // 1024 * 1024
        pk, pv = pinned(n * 8), pinned(n * 8)
        k, v = view(pk, np.int64, n), view(pv, np.float64, n)
        ng = 1_000_000
        fill(k, lambda m: rng.integers(0, ng, m)); fill(v, lambda m: rng.random(m))
        cr = 1 << 20
        keys = [A.HostArray(k, None, i, min(cr, n - i), A.I64, 0) for i in range(0, n, cr)]
        vals = [A.HostArray(v, None, i, min(cr, n - i), A.F64, 0) for i in range(0, n, cr)]
        sec, (ok_, ov, oc) = best_of(lambda: api.groupby_agg([keys], vals, "sum", ng))
        tot = float(ov.to_numpy()[:ov.length].sum())
        ok = bool(oc.length == ng and int(oc.to_numpy()[:oc.length].sum()) == n and abs(tot - float(v.sum())) <= 1e-6 * tot)
        report("stream_groupby_sum_1e6_keys_pinned_1M_row_batches", n * 16, ng * 24, sec, {"rows": n, "rows_per_s": n / sec, "groups": int(oc.length), "check": ok, "note": "hash GROUP BY over a host-resident frame: every slab aggregated on the device, partial groups merged slab by slab"})
        L.rdf_host_free(pk); L.rdf_host_free(pv)
    for q in hog:
        L.rdf_dev_free(q)


if __name__ == "__main__":
    main()
