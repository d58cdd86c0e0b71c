#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on HBM-resident columns (RDF_MEM_DEVICE): achieved algorithmic GB/s against the
8 TB/s HBM peak for every kernel family of the path.  Not the driver's bench (that is bench.py); this is the
tool the kernel tuning loop reads.  Usage: python tools/bench_kernels.py [--rows N] [--only name,name]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from rust_dataframe_amd import _abi as A  # noqa: E402
from rust_dataframe_amd import lib  # noqa: E402

PEAK = 8000.0


def dev_f64(n, col, lo=0.0, hi=1.0, seed=42):
    t = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()   # the block may be one torch just freed with work still queued on torch's stream; the fill runs on the library's
    lib.fill_uniform_f64(t.data_ptr(), n, seed, col, 0, lo, hi)
    return t


def dev_i64(n, col, lo=-2 ** 31, hi=2 ** 31, seed=42):
    t = torch.empty(n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib.fill_uniform_i64(t.data_ptr(), n, seed, col, 0, lo, hi)
    return t


def dev_validity(n, col, frac, seed=42):
    t = torch.zeros((n + 63) // 64 * 8 + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    lib.fill_validity(t.data_ptr(), n, seed, col, 0, frac)
    return t


def arr(t, dtype, n, validity=None):
    return A.DeviceArray(t.data_ptr(), validity.data_ptr() if validity is not None else None, 0, n, dtype, -1, keep=(t, validity))


def out_like(dtype, n, with_validity=False):
    es = {A.F64: 8, A.I64: 8, A.U32: 4, A.I32: 4, A.BOOL: 0}[dtype]
    pad = (n + 63) // 64 * 64
    v = torch.empty(pad * es if es else pad // 8 + 8, dtype=torch.uint8, device="cuda")
    b = torch.empty(pad // 8 + 8, dtype=torch.uint8, device="cuda") if with_validity else None
    return A.DeviceArray(v.data_ptr(), b.data_ptr() if b is not None else None, 0, n, dtype, 0, keep=(v, b))


def timed(fn, steps, warmup=2):
    torch.cuda.synchronize()   # inputs made by torch ops live on torch's stream; the library launches on its own
    for _ in range(warmup):
        fn()
    lib.synchronize()
    lib.kernel_timing_reset(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    lib.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms, n = lib.kernel_timing_get()
    lib.kernel_timing_reset(False)
    return wall, (ms / max(n, 1)) * 1e-3 * (n / steps if n else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--no-spec", action="store_true", help="force the general evaluator")
    ap.add_argument("--sort-super", type=int, default=-1, help="rdf_set_option(\"sort_super\"): tiles per ticket of the sort's digit passes (default: the library's, 1 = a tile per ticket; K > 1 = os_scatter4_kernel)")
    args = ap.parse_args()
    n = args.rows
    only = set(filter(None, args.only.split(",")))
    lib.set_device(0)
    api = lib.api()
    if args.sort_super >= 0:
        lib.set_option("sort_super", args.sort_super)
    if args.no_spec:
        lib.set_option("spec", 0)
        lib.set_option("fast_filter", 0)
    results = []

    def report(name, alg_bytes, fn, rows=None):
        """rows = the rows THIS entry processes (take / sort / join / list entries run on fewer rows than --rows)."""
        if only and name not in only:
            return
        wall, kern = timed(fn, args.steps)
        gbs = alg_bytes / kern / 1e9 if kern > 0 else 0.0
        r = {"kernel": name, "rows": n if rows is None else rows, "alg_bytes": alg_bytes, "wall_ms": wall * 1e3, "kernel_ms": kern * 1e3,
             "GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 3)}
        results.append(r)
        print(json.dumps(r), flush=True)

    if not only or "probe" in only:
        # independent denominators: a read-only and a copy probe with stock torch kernels on the same box
        px = torch.empty(n, dtype=torch.float64, device="cuda").uniform_()
        py = torch.empty_like(px)
        def probe(name, fn, nbytes):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.steps
            r = {"kernel": name, "rows": n, "alg_bytes": nbytes, "kernel_ms": ms, "GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / PEAK, 3)}
            results.append(r)
            print(json.dumps(r), flush=True)
        probe("probe_torch_sum_read_only", lambda: px.sum(), 8.0 * n)
        probe("probe_torch_copy", lambda: py.copy_(px), 16.0 * n)
        probe("probe_torch_add_store", lambda: torch.add(px, px, out=py), 16.0 * n)
        del px, py
        torch.cuda.empty_cache()
    x, y, z = dev_f64(n, 0, -1, 1), dev_f64(n, 1, -1, 1), dev_f64(n, 2, -1, 1)
    k = dev_i64(n, 3)
    vx = dev_validity(n, 0, 0.1)
    X, Y, Z, K = arr(x, A.F64, n), arr(y, A.F64, n), arr(z, A.F64, n), arr(k, A.I64, n)
    XV = arr(x, A.F64, n, vx)

    e = A.Expr()
    cx, cy, cz, ck = e.col(0), e.col(1), e.col(2), e.col(3)
    gt = e.op("gt", cx, e.scalar(0.0))
    e_add0 = e.op("add", cx, e.scalar(0.0))   # same bytes as the fast path, but forces the interpreter
    fma = e.op("add", e.op("multiply", cx, cy), cz)
    sin1 = e.op("sin", e.op("add", cx, e.scalar(1.0)))

    report("filter_sum_fast", 8.0 * n, lambda: api.pipeline(e, [[X]], [cx], gt))
    report("filter_sum_fast_validity", 8.125 * n, lambda: api.pipeline(e, [[XV]], [cx], gt))
    report("filter_sum_other_col_fast", 16.0 * n, lambda: api.pipeline(e, [[X], [Y]], [cy], gt))
    report("filter_sum_add0_shape_rt", 8.0 * n, lambda: api.pipeline(e, [[X]], [e_add0], gt))   # shape-specialised, runtime operators
    sin_ab = e.op("sin", e.op("subtract", cx, cy))
    report("sin_a_minus_b_sum_shape_rt", 16.0 * n, lambda: api.pipeline(e, [[X], [Y]], [sin_ab]))
    abc = e.op("multiply", e.op("add", cx, cy), cz)
    report("a_plus_b_times_c_store_shape_rt", 32.0 * n, lambda: api.pipeline(e, [[X], [Y], [Z]], [abc], -1, A.SINK_STORE, [[out_like(A.F64, n)]]))
    and2 = e.op("and", gt, e.op("lt", cy, e.scalar(0.5)))
    report("filter_and2_sum_shape_rt", 16.0 * n, lambda: api.pipeline(e, [[X], [Y]], [cx], and2))
    if not args.no_spec:
        lib.set_option("spec", 0); lib.set_option("fast_filter", 0)
    report("filter_sum_interp", 8.0 * n, lambda: api.pipeline(e, [[X]], [e_add0], gt))            # the general evaluator on the same program
    report("filter_and2_sum_interp", 16.0 * n, lambda: api.pipeline(e, [[X], [Y]], [cx], and2))
    if not args.no_spec:
        lib.set_option("spec", 1); lib.set_option("fast_filter", 1)
    report("sum_interp", 8.0 * n, lambda: api.pipeline(e, [[X]], [cx]))
    report("sum_interp_validity", 8.125 * n, lambda: api.pipeline(e, [[XV]], [cx]))
    report("c3_fma_minmax_4col", 32.0 * n, lambda: api.pipeline(e, [[X], [Y], [Z], [K]], [fma, ck]))
    report("c1_sin_add_scalar_sum", 8.0 * n, lambda: api.pipeline(e, [[X]], [sin1]))
    o1 = out_like(A.F64, n)
    report("add_store", 24.0 * n, lambda: api.binary("add", [X], [Y], [o1]))
    report("fma_store", 32.0 * n, lambda: api.pipeline(e, [[X], [Y], [Z]], [fma], -1, A.SINK_STORE, [[o1]]))
    report("sin_store", 16.0 * n, lambda: api.unary("sin", [X], [o1]))
    ov = out_like(A.F64, n, True)
    report("add_store_validity", 24.25 * n, lambda: api.binary("add", [XV], [Y], [ov]))
    m = out_like(A.BOOL, n)
    report("predicate_to_mask", 8.125 * n, lambda: api.predicate(e, gt, [[X]], [m]))
    api.predicate(e, gt, [[X]], [m])
    cnt = api.filter_count([m])[0]
    sel = cnt / n
    of, ok2 = out_like(A.F64, n), out_like(A.I64, n)
    report("filter_count", n / 8.0, lambda: api.filter_count([m]))
    report("filter_1col", (8 + 8 * sel + 0.25) * n, lambda: api.filter([X], [m], [of]))
    report("filter_2col", (16 + 16 * sel + 0.25) * n, lambda: api.filter_columns([[X], [K]], [m], [[of], [ok2]]))
    lib.set_option("filter_block", 0)     # (the A/B of the count -> scan -> compact kernels: round 6's one-pass block kernel would take all three otherwise)
    for gen in (1, 3, 2):   # A/B: first-generation block tiles (one barrier per tile), register-staged wave tiles, default (LDS-DMA wave tiles)
        lib.set_option("filter_gen", gen)
        report(f"filter_1col_gen{gen}", (8 + 8 * sel + 0.25) * n, lambda: api.filter([X], [m], [of]))
        report(f"filter_2col_gen{gen}", (16 + 16 * sel + 0.25) * n, lambda: api.filter_columns([[X], [K]], [m], [[of], [ok2]]))
    lib.set_option("filter_gen", 2)
    lib.set_option("filter_block", 1)
    # selective filters: 1 row in 16 kept (sparse tiles fetch only the sectors that hold a kept row)
    m16 = out_like(A.BOOL, n)
    api.predicate(e, e.op("gt", cx, e.scalar(0.875)), [[X]], [m16])
    sel16 = api.filter_count([m16])[0] / n
    report("filter_1col_selectivity_1_16", (8 + 8 * sel16 + 0.25) * n, lambda: api.filter([X], [m16], [of]))
    # the same filter over a frame in the reference's 1024-row RecordBatches (the whole column = 976 563 chunks at 1e9 rows)
    nfc = n
    XF = A.PreparedCol([A.DeviceArray(x.data_ptr() + i * 8, None, 0, min(1024, nfc - i), A.F64, 0, keep=x) for i in range(0, nfc, 1024)])
    MF = A.PreparedCol([A.DeviceArray(m.values_ptr + i // 8, None, 0, min(1024, nfc - i), A.BOOL, 0, keep=m) for i in range(0, nfc, 1024)])
    ofb = torch.empty(nfc, dtype=torch.float64, device="cuda")
    OF = [A.DeviceArray(ofb.data_ptr() + i * 8, None, 0, min(1024, nfc - i), A.F64, 0, keep=ofb) for i in range(0, nfc, 1024)]
    report("filter_1col_1024_row_chunks", (8 + 8 * sel + 0.25) * nfc, lambda: api.filter(XF, MF, OF))     # round 6: one pass on the block kernel's short-batch mode (a chunk per wave), no count / scan
    lib.set_option("filter_short", 0)
    report("filter_1col_1024_row_chunks_wave_tiles", (8 + 8 * sel + 0.25) * nfc, lambda: api.filter(XF, MF, OF))     # round 6, earlier: one pass on the wave-tile LDS-DMA kernel
    lib.set_option("filter_short", 1)
    lib.set_option("filter_block", 0)
    for gen in (1, 2, 3):   # 3 = wave-granular without the next-tile look-ahead
        lib.set_option("filter_gen", gen)
        report(f"filter_1col_1024_row_chunks_gen{gen}", (8 + 8 * sel + 0.25) * nfc, lambda: api.filter(XF, MF, OF))
    lib.set_option("filter_gen", 2)
    lib.set_option("filter_block", 1)
    # the same with output chunks sized by rdf_filter_count and packed back to back (what a two-phase caller allocates):
    # the slots above leave every other 4 KB of the output buffer untouched
    import itertools
    cnts = api.filter_count(MF)
    offs = [0] + list(itertools.accumulate((c + 7) // 8 * 8 for c in cnts))
    OFP = [A.DeviceArray(ofb.data_ptr() + o * 8, None, 0, c, A.F64, 0, keep=ofb, capacity=(c + 7) // 8 * 8) for o, c in zip(offs, cnts)]
    report("filter_1col_1024_row_chunks_packed_outputs", (8 + 8 * sel + 0.25) * nfc, lambda: api.filter(XF, MF, OFP))
    del OFP
    lib.set_option("filter_gen", 2)
    del XF, MF, OF
    for cr in (4096, 65536):   # the per-chunk cost: the same filter on longer batches
        XF = A.PreparedCol([A.DeviceArray(x.data_ptr() + i * 8, None, 0, min(cr, nfc - i), A.F64, 0, keep=x) for i in range(0, nfc, cr)])
        MF = A.PreparedCol([A.DeviceArray(m.values_ptr + i // 8, None, 0, min(cr, nfc - i), A.BOOL, 0, keep=m) for i in range(0, nfc, cr)])
        OF = [A.DeviceArray(ofb.data_ptr() + i * 8, None, 0, min(cr, nfc - i), A.F64, 0, keep=ofb) for i in range(0, nfc, cr)]
        report(f"filter_1col_{cr}_row_chunks", (8 + 8 * sel + 0.25) * nfc, lambda: api.filter(XF, MF, OF))
        if cr > 8192:
            lib.set_option("filter_owned", 0)
            report(f"filter_1col_{cr}_row_chunks_scanner_wave", (8 + 8 * sel + 0.25) * nfc, lambda: api.filter(XF, MF, OF))     # tiles by ticket, offsets from the scanner wave (before a block owned whole chunks)
            lib.set_option("filter_owned", 1)
        if cr <= 8192:
            lib.set_option("filter_short", 0)
            report(f"filter_1col_{cr}_row_chunks_three_kernels", (8 + 8 * sel + 0.25) * nfc, lambda: api.filter(XF, MF, OF))     # count, scan, compact (before the short-batch mode)
            lib.set_option("filter_short", 1)
        del XF, MF, OF
    del ofb
    nidx = n // 4
    idx = torch.randint(0, n, (nidx,), dtype=torch.int64, device="cuda").to(torch.uint32 if hasattr(torch, "uint32") else torch.int32)
    I = A.DeviceArray(idx.data_ptr(), None, 0, nidx, A.U32, 0, keep=idx)
    ot = out_like(A.F64, nidx)
    report("take_random_u32", (4 + 8 + 8) * nidx, lambda: api.take([X], I, ot), rows=nidx)
    sidx = torch.arange(0, nidx, dtype=torch.int64, device="cuda").to(torch.uint32)
    S = A.DeviceArray(sidx.data_ptr(), None, 0, nidx, A.U32, 0, keep=sidx)
    report("take_sequential_u32", (4 + 8 + 8) * nidx, lambda: api.take([X], S, ot), rows=nidx)
    # take from / sort of a column in the reference's 1024-row batches (a 1e8-row prefix = 97 657 chunks): every element
    # resolves its chunk on its own (find_chunk_row)
    nsrc = min(n, 100_000_000)
    XS = [A.DeviceArray(x.data_ptr() + i * 8, None, 0, min(1024, nsrc - i), A.F64, 0, keep=x) for i in range(0, nsrc, 1024)]
    nix = nsrc // 4
    idx2 = torch.randint(0, nsrc, (nix,), dtype=torch.int64, device="cuda").to(torch.uint32)
    I2 = A.DeviceArray(idx2.data_ptr(), None, 0, nix, A.U32, 0, keep=idx2)
    ot2 = out_like(A.F64, nix)
    report("take_random_u32_from_1024_row_chunks", (4 + 8 + 8) * nix, lambda: api.take(XS, I2, ot2), rows=nix)
    nks = min(nsrc, 50_000_000)
    KS1 = [A.DeviceArray(k.data_ptr() + i * 8, None, 0, min(1024, nks - i), A.I64, 0, keep=k) for i in range(0, nks, 1024)]
    oi2 = out_like(A.U32, nks)
    report("sort_to_indices_i64_1024_row_chunks", 8.0 * nks, lambda: api.sort_to_indices([KS1], [False], oi2), rows=nks)
    del XS, KS1, idx2
    # chunked device-resident columns (what a frame looks like after a filter): 1M-row chunks
    def chunked(t, dtype, rows=1 << 20):
        return [A.DeviceArray(t.data_ptr() + i * 8, None, 0, min(rows, n - i), dtype, 0, keep=t) for i in range(0, n, rows)]
    XC, YC, ZC, KC = chunked(x, A.F64), chunked(y, A.F64), chunked(z, A.F64), chunked(k, A.I64)
    report("filter_sum_chunked_1M", 8.0 * n, lambda: api.pipeline(e, [XC], [cx], gt))
    report("c3_chunked_1M", 32.0 * n, lambda: api.pipeline(e, [XC, YC, ZC, KC], [fma, ck]))
    oc1 = [out_like(A.F64, a_.length) for a_ in XC]
    report("add_store_chunked_1M", 24.0 * n, lambda: api.binary("add", XC, YC, oc1))
    # DataFrame::sort: radix sort to indices on an i64 key (8 passes) and an i32-range key stored as i64
    ns = min(n, 50_000_000)
    KS = arr(k, A.I64, ns)
    oi = out_like(A.U32, ns)
    report("sort_to_indices_i64", 8.0 * ns, lambda: api.sort_to_indices([[KS]], [False], oi), rows=ns)   # keys in [-2^31, 2^31): 4 varying bytes + sign
    kw = dev_i64(ns, 9, -2 ** 62, 2 ** 62)
    report("sort_to_indices_i64_full_range", 8.0 * ns, lambda: api.sort_to_indices([[arr(kw, A.I64, ns)]], [False], oi), rows=ns)
    lib.set_option("sort_msd", 0)      # A/B: one pass per byte
    report("sort_to_indices_i64_full_range_byte_passes", 8.0 * ns, lambda: api.sort_to_indices([[arr(kw, A.I64, ns)]], [False], oi), rows=ns)
    lib.set_option("sort_msd", 1)
    if n >= 200_000_000:
        ns2 = min(n, 1_000_000_000)
        kw2 = dev_i64(ns2, 9, -2 ** 62, 2 ** 62)
        oi2_ = out_like(A.U32, ns2)
        report("sort_to_indices_i64_full_range_1e9", 8.0 * ns2, lambda: api.sort_to_indices([[arr(kw2, A.I64, ns2)]], [False], oi2_), rows=ns2)
        del kw2, oi2_
    kd = dev_i64(ns, 10, 0, 200)
    report("sort_to_indices_i64_dictionary_codes", 8.0 * ns, lambda: api.sort_to_indices([[arr(kd, A.I64, ns)]], [False], oi), rows=ns)
    xs_ = dev_f64(ns, 13, 0.0, 1.0)
    report("sort_to_indices_f64_uniform", 8.0 * ns, lambda: api.sort_to_indices([[arr(xs_, A.F64, ns)]], [False], oi), rows=ns)
    xn_ = torch.randn(ns, device="cuda", dtype=torch.float64)
    report("sort_to_indices_f64_normal", 8.0 * ns, lambda: api.sort_to_indices([[arr(xn_, A.F64, ns)]], [False], oi), rows=ns)
    lib.set_option("sort_msd", 0)
    report("sort_to_indices_f64_uniform_byte_passes", 8.0 * ns, lambda: api.sort_to_indices([[arr(xs_, A.F64, ns)]], [False], oi), rows=ns)
    lib.set_option("sort_msd", 1)
    # value buckets planned from a sample (round 4) against buckets over [min, max] with ~500 rows each (round 3); a handful of far
    # outliers / infinities / NaNs in the column; a heavy-tailed column
    lib.set_option("sort_sample", 0)
    report("sort_to_indices_f64_uniform_buckets_over_min_max", 8.0 * ns, lambda: api.sort_to_indices([[arr(xs_, A.F64, ns)]], [False], oi), rows=ns)
    report("sort_to_indices_f64_normal_buckets_over_min_max", 8.0 * ns, lambda: api.sort_to_indices([[arr(xn_, A.F64, ns)]], [False], oi), rows=ns)
    lib.set_option("sort_sample", 1)
    xo_ = xn_.clone()
    xo_[torch.randint(0, ns, (12,), device="cuda")] = torch.tensor([float("inf"), float("-inf"), float("nan"), 1e300, -1e300, 1e15] * 2, device="cuda", dtype=torch.float64)
    report("sort_to_indices_f64_normal_with_outliers", 8.0 * ns, lambda: api.sort_to_indices([[arr(xo_, A.F64, ns)]], [False], oi), rows=ns)
    lib.set_option("sort_sample", 0)
    report("sort_to_indices_f64_normal_with_outliers_buckets_over_min_max", 8.0 * ns, lambda: api.sort_to_indices([[arr(xo_, A.F64, ns)]], [False], oi), rows=ns)
    lib.set_option("sort_sample", 1)
    xl_ = torch.exp(xn_)
    report("sort_to_indices_f64_lognormal", 8.0 * ns, lambda: api.sort_to_indices([[arr(xl_, A.F64, ns)]], [False], oi), rows=ns)
    lib.set_option("sort_sample", 0)
    report("sort_to_indices_f64_lognormal_buckets_over_min_max", 8.0 * ns, lambda: api.sort_to_indices([[arr(xl_, A.F64, ns)]], [False], oi), rows=ns)
    lib.set_option("sort_sample", 1)
    del xo_, xl_
    # the reference's own sort case (src/dataframe.rs:963-1003: two criteria, `a` descending then `b` ascending) at size: a = a
    # category column of 1000 values, b = a measure; and an f32 key column (value buckets since round 5, byte passes before)
    ka_ = dev_i64(ns, 11, 0, 1000)
    report("sort_to_indices_2keys_i64_desc_f64_asc", 16.0 * ns, lambda: api.sort_to_indices([[arr(ka_, A.I64, ns)], [arr(xs_, A.F64, ns)]], [True, False], oi), rows=ns)
    report("sort_to_indices_2keys_i64_asc_i64_asc", 16.0 * ns, lambda: api.sort_to_indices([[arr(ka_, A.I64, ns)], [arr(kw, A.I64, ns)]], [False, False], oi), rows=ns)
    xf_ = xs_.to(torch.float32)
    xfn_ = xn_.to(torch.float32)
    torch.cuda.synchronize()
    F32 = lambda t: A.DeviceArray(t.data_ptr(), None, 0, ns, A.F32, -1, keep=t)
    report("sort_to_indices_f32_uniform", 4.0 * ns, lambda: api.sort_to_indices([[F32(xf_)]], [False], oi), rows=ns)
    report("sort_to_indices_f32_normal", 4.0 * ns, lambda: api.sort_to_indices([[F32(xfn_)]], [False], oi), rows=ns)
    lib.set_option("sort_msd", 0)
    report("sort_to_indices_f32_uniform_byte_passes", 4.0 * ns, lambda: api.sort_to_indices([[F32(xf_)]], [False], oi), rows=ns)
    lib.set_option("sort_msd", 1)
    del ka_, xf_, xfn_
    del xs_, xn_
    # ArrayFunctions over a List<f64> column: rows of 10 elements (one row per lane) and of 1000 elements (one row per wave)
    for rl in (10, 1000):
        nm = f"list_rows_of_{rl}"
        if only and not any(k.startswith(nm) for k in only):
            continue
        nv = min(n, 200_000_000) // rl * rl
        nr = nv // rl
        lo_ = (torch.arange(0, nr + 1, device="cuda", dtype=torch.int64) * rl).to(torch.int32)
        lv_ = torch.floor(dev_f64(nv, 13, 0.0, 50.0))
        L = A.DeviceList(lo_.data_ptr(), nr, arr(lv_, A.F64, nv), keep=(lo_, lv_))
        ob, op, om = out_like(A.BOOL, nr, True), out_like(A.I32, nr), out_like(A.F64, nr, True)
        lbytes = 8.0 * nv + 4.0 * nr
        report(nm + "_contains", lbytes, lambda: api.list_contains(L, 7.0, ob), rows=nr)
        report(nm + "_position", lbytes + 4.0 * nr, lambda: api.list_position(L, 7.0, op), rows=nr)
        report(nm + "_max", lbytes + 8.0 * nr, lambda: api.list_extreme(L, True, om), rows=nr)
        oro, orv = out_like(A.I32, nr + 1), out_like(A.F64, nv)
        report(nm + "_remove", 2 * lbytes + 8.0 * nv * 0.98 + 4.0 * nr, lambda: api.list_remove(L, 7.0, (oro, orv)), rows=nr)
        # the set-valued functions: reads in both passes (count, write) + the elements kept; quadratic compare work per row
        osd = (out_like(A.I32, nr + 1), out_like(A.F64, nv))
        kept = api.list_set("distinct", L, outs=osd)[1].length
        report(nm + "_distinct", 2 * lbytes + 8.0 * kept + 4.0 * nr, lambda: api.list_set("distinct", L, outs=osd), rows=nr)
        if rl == 10:
            lv2_ = torch.floor(dev_f64(nv, 14, 0.0, 50.0))
            L2 = A.DeviceList(lo_.data_ptr(), nr, arr(lv2_, A.F64, nv), keep=(lo_, lv2_))
            osu = (out_like(A.I32, nr + 1), out_like(A.F64, 2 * nv))
            kept = api.list_set("union", L, L2, outs=osu)[1].length
            report(nm + "_union", 4 * lbytes + 8.0 * kept + 4.0 * nr, lambda: api.list_set("union", L, L2, outs=osu), rows=nr)
            kept = api.list_set("intersect", L, L2, outs=osd)[1].length
            report(nm + "_intersect", 4 * lbytes + 8.0 * kept + 4.0 * nr, lambda: api.list_set("intersect", L, L2, outs=osd), rows=nr)
            del lv2_, osu
        if rl == 10:
            nsort = min(nv, 50_000_000)
            Ls = A.DeviceList(lo_.data_ptr(), nsort // rl, arr(lv_, A.F64, nsort), keep=(lo_, lv_))
            osv = out_like(A.F64, nsort)
            report(nm + "_sort", 16.0 * nsort, lambda: api.list_sort(Ls, osv), rows=nsort)
        del lo_, lv_
    # DataFrame::join: 1e8 probe rows against 1e7 distinct build keys (inner: every probe row finds exactly one partner)
    if not only or any(k.startswith("join_inner_") for k in only):
        nl_, nr_ = min(n, 100_000_000), 10_000_000
        lk_ = dev_i64(nl_, 12, 0, nr_)
        rk_ = torch.randperm(nr_, device="cuda", dtype=torch.int64)
        jl, jr = out_like(A.U32, nl_, True), out_like(A.U32, nl_, True)
        report("join_inner_1e8_x_1e7", 8.0 * nl_ + 8.0 * nr_ + 8.0 * nl_, lambda: api.equijoin_indices([arr(lk_, A.I64, nl_)], [arr(rk_, A.I64, nr_)], "inner", (jl, jr)), rows=nl_)
        lib.set_option("join_table", 1)      # A/B: the build side sorted by key, table slots claimed by compare-and-swap (round 3)
        report("join_inner_1e8_x_1e7_cas_table", 8.0 * nl_ + 8.0 * nr_ + 8.0 * nl_, lambda: api.equijoin_indices([arr(lk_, A.I64, nl_)], [arr(rk_, A.I64, nr_)], "inner", (jl, jr)), rows=nl_)
        lib.set_option("join_table", 0)      # A/B: bucket index over the sorted build keys
        report("join_inner_1e8_x_1e7_bucket_index", 8.0 * nl_ + 8.0 * nr_ + 8.0 * nl_, lambda: api.equijoin_indices([arr(lk_, A.I64, nl_)], [arr(rk_, A.I64, nr_)], "inner", (jl, jr)), rows=nl_)
        lib.set_option("join_table", 2)
        del lk_, rk_
        # a build side as long as the probe side: 1e8 distinct keys (a 3.2 GB table of 16-byte slots at load 0.5: every probe is a
        # random line from HBM; the case a radix-partitioned join was priced for)
        if n >= 100_000_000:
            nb_ = 100_000_000
            rk8 = torch.randperm(nb_, device="cuda", dtype=torch.int64)
            lk8 = dev_i64(nl_, 12, 0, nb_)
            report("join_inner_1e8_x_1e8", 8.0 * nl_ + 8.0 * nb_ + 8.0 * nl_, lambda: api.equijoin_indices([arr(lk8, A.I64, nl_)], [arr(rk8, A.I64, nb_)], "inner", (jl, jr)), rows=nl_)
            lib.set_option("join_table", 1)
            report("join_inner_1e8_x_1e8_cas_table", 8.0 * nl_ + 8.0 * nb_ + 8.0 * nl_, lambda: api.equijoin_indices([arr(lk8, A.I64, nl_)], [arr(rk8, A.I64, nb_)], "inner", (jl, jr)), rows=nl_)
            lib.set_option("join_table", 2)
            del lk8, rk8
    # hash GROUP BY key -> sum(val): 1e6 groups (config C4's per-GPU leg) and 1e3 groups (contended)
    for ng in (1_000_000, 2_000, 1_000, 100, 8):
        kk = dev_i64(n, 7, 0, ng)
        KK = arr(kk, A.I64, n)
        ok_, os_, oc_ = out_like(A.I64, ng + 2), out_like(A.F64, ng + 2), out_like(A.I64, ng + 2)
        oc2_ = out_like(A.I64, ng + 2)
        report(f"groupby_sum_{ng}_groups", 16.0 * n, lambda: api.groupby_sum([KK], [X], ng, (ok_, os_, oc_)))
        report(f"groupby_max_{ng}_groups", 16.0 * n, lambda: api.groupby_agg([[KK]], [X], "max", ng, ([ok_], os_, oc_)))
        lib.set_option("gb_partition", 1)
        report(f"groupby_sum_{ng}_groups_first_generation", 16.0 * n, lambda: api.groupby_sum([KK], [X], ng, (ok_, os_, oc_)))
        lib.set_option("gb_partition", 3)
        if ng > 2048:
            # skewed keys (SURVEY.md §8d C4 variant): Zipf-like s = 1.1 (inverse-CDF of the continuous power law, clamped) and one hot key
            u = torch.rand(n, device="cuda", dtype=torch.float64).clamp_(min=1e-12)
            kz = torch.clamp(torch.floor(u.pow(-1.0 / 0.1)), max=float(ng - 1)).to(torch.int64)
            report(f"groupby_sum_{ng}_groups_zipf", 16.0 * n, lambda: api.groupby_sum([arr(kz, A.I64, n)], [X], ng, (ok_, os_, oc_)))
            kh = torch.where(torch.rand(n, device="cuda") < 0.3, torch.full((n,), 7, device="cuda", dtype=torch.int64), kk)
            report(f"groupby_sum_{ng}_groups_hot_key_30pct", 16.0 * n, lambda: api.groupby_sum([arr(kh, A.I64, n)], [X], ng, (ok_, os_, oc_)))
            lib.set_option("gb_hot", 0)         # A/B: without the heavy-hitter split (round 3: first-generation path / capacity plan)
            report(f"groupby_sum_{ng}_groups_zipf_no_split", 16.0 * n, lambda: api.groupby_sum([arr(kz, A.I64, n)], [X], ng, (ok_, os_, oc_)))
            report(f"groupby_sum_{ng}_groups_hot_key_30pct_no_split", 16.0 * n, lambda: api.groupby_sum([arr(kh, A.I64, n)], [X], ng, (ok_, os_, oc_)))
            lib.set_option("gb_hot", 1)
            ks = (kk * 6364136223846793005 + 1442695040888963407) ^ (kk << 29)     # the same groups under scattered 64-bit key values
            report(f"groupby_sum_{ng}_groups_scattered_keys", 16.0 * n, lambda: api.groupby_sum([arr(ks, A.I64, n)], [X], ng, (ok_, os_, oc_)))
            lib.set_option("gb_bucket", 1)      # A/B: one key per probe in the aggregate pass's partition tables (round 3)
            report(f"groupby_sum_{ng}_groups_scattered_keys_single_slot_tables", 16.0 * n, lambda: api.groupby_sum([arr(ks, A.I64, n)], [X], ng, (ok_, os_, oc_)))
            lib.set_option("gb_bucket", 4)
            report(f"groupby_sum_{ng}_groups_bucket_tables", 16.0 * n, lambda: api.groupby_sum([KK], [X], ng, (ok_, os_, oc_)))
            lib.set_option("gb_bucket", 0)
            del ks
            report(f"groupby_sum_{ng}_groups_null_values", 16.125 * n, lambda: api.groupby_sum([KK], [XV], ng, (ok_, os_, oc_)))
            del u, kz, kh
            for dbg in ((1, 2, 4, 5, 6, 7, 8) if "ablate" in args.only else ()):   # ablations of the FIRST-generation passes
                lib.set_option("gb_debug", dbg)
                try:
                    report(f"groupby_sum_{ng}_groups_ablate{dbg}", 16.0 * n, lambda: api.groupby_sum([KK], [X], ng, (ok_, os_, oc_)))
                except Exception as ex:   # the ablations produce wrong results by construction
                    print("ablation", dbg, ex)
            lib.set_option("gb_debug", 0)
            report(f"groupby_count_{ng}_groups", 8.0 * n, lambda: api.groupby_sum([KK], None, ng, (ok_, oc2_, oc_)))
            lib.set_option("gb_compact", 0)      # A/B: 16-byte records for COUNT too
            report(f"groupby_count_{ng}_groups_16_byte_records", 8.0 * n, lambda: api.groupby_sum([KK], None, ng, (ok_, oc2_, oc_)))
            lib.set_option("gb_compact", 2)      # A/B: 12-byte records (key word + value) for SUM
            report(f"groupby_sum_{ng}_groups_12_byte_records", 16.0 * n, lambda: api.groupby_sum([KK], [X], ng, (ok_, os_, oc_)))
            lib.set_option("gb_compact", 1)
            lib.set_option("gb_partition", 2)
            report(f"groupby_sum_{ng}_groups_radix_sort", 16.0 * n, lambda: api.groupby_sum([KK], [X], ng, (ok_, os_, oc_)))
            lib.set_option("gb_partition", 0)
            report(f"groupby_sum_{ng}_groups_hbm_atomics", 16.0 * n, lambda: api.groupby_sum([KK], [X], ng, (ok_, os_, oc_)))
            lib.set_option("gb_partition", 3)
    # config C5 (TPC-H Q1 shape): filter(shipdate <= c) -> 5 sums + counts in 6 groups, 38 B/row
    if not only or "q1_grouped" in only:
        qty = torch.randint(1, 51, (n,), device="cuda").to(torch.float64)
        price = dev_f64(n, 11, 900.0, 105000.0)
        disc = torch.randint(0, 11, (n,), device="cuda").to(torch.float64) / 100.0
        tax = torch.randint(0, 9, (n,), device="cuda").to(torch.float64) / 100.0
        flag = torch.randint(0, 3, (n,), dtype=torch.int8, device="cuda")
        status = torch.randint(0, 2, (n,), dtype=torch.int8, device="cuda")
        ship = torch.randint(8036, 10562, (n,), dtype=torch.int32, device="cuda")
        qcols = [[arr(qty, A.F64, n)], [arr(price, A.F64, n)], [arr(disc, A.F64, n)], [arr(tax, A.F64, n)],
                 [arr(flag, A.I8, n)], [arr(status, A.I8, n)], [arr(ship, A.I32, n)]]
        q = A.Expr()
        c_ = [q.col(i) for i in range(7)]
        pred = q.op("le", c_[6], q.scalar(10471, A.I32))
        gid = q.op("add", q.op("multiply", q.cast(c_[4], A.I32), q.scalar(2, A.I32)), q.cast(c_[5], A.I32))
        dp = q.op("multiply", c_[1], q.op("subtract", q.scalar(1.0), c_[2]))
        ch = q.op("multiply", dp, q.op("add", q.scalar(1.0), c_[3]))
        report("q1_grouped", 38.0 * n, lambda: api.group_pipeline(q, qcols, [c_[0], c_[1], dp, ch, c_[2]], gid, 6, pred))
        del qty, price, disc, tax, flag, status, ship
    # A/B in one process: the specialised template kernel vs the dedicated filter_agg_f64 kernel on the headline shape
    for rep in range(3):
        lib.set_option("spec", 1)
        report(f"headline_spec_{rep}", 8.0 * n, lambda: api.pipeline(e, [[X]], [cx], gt))
        lib.set_option("spec", 0)
        report(f"headline_filter_agg_{rep}", 8.0 * n, lambda: api.pipeline(e, [[X]], [cx], gt))
    lib.set_option("spec", 1)
    # A/B: bitmap words through scalar loads (0) vs vector loads + readlane (1)
    for rep in range(2):
        for vb in (0, 1):
            lib.set_option("vec_bitmap", vb)
            report(f"ab_filter_sum_validity_vecbitmap{vb}_{rep}", 8.125 * n, lambda: api.pipeline(e, [[XV]], [cx], gt))
            report(f"ab_sum_validity_vecbitmap{vb}_{rep}", 8.125 * n, lambda: api.pipeline(e, [[XV]], [cx]))
            # the same over chunked columns: 1 Mi-row chunks, and the reference's 1024-row batches on a 5e7-row prefix
            XVC = [A.DeviceArray(x.data_ptr() + i * 8, vx.data_ptr() + i // 8, 0, min(1 << 20, n - i), A.F64, -1, keep=(x, vx)) for i in range(0, n, 1 << 20)]
            report(f"ab_filter_sum_validity_chunked_1M_vecbitmap{vb}_{rep}", 8.125 * n, lambda: api.pipeline(e, [XVC], [cx], gt))
            ns_ = min(n, 50_000_000)
            XVS = [A.DeviceArray(x.data_ptr() + i * 8, vx.data_ptr() + i // 8, 0, min(1024, ns_ - i), A.F64, -1, keep=(x, vx)) for i in range(0, ns_, 1024)]
            report(f"ab_filter_sum_validity_chunked_1024_vecbitmap{vb}_{rep}", 8.125 * ns_, lambda: api.pipeline(e, [XVS], [cx], gt), rows=ns_)
            if vb == 1 and rep == 0:
                # the same call with the ctypes descriptors built once: wall_ms is then the library's own per-call host
                # work for 48 828 chunks (chunk tables, their upload, the launch), without the Python marshalling
                import ctypes as C
                nodes = e.c_array()
                prog = A.rdf_program(C.cast(nodes, C.POINTER(A.rdf_expr_node)), len(e.nodes), gt, 1,
                                     (C.c_int32 * A.MAX_VALUES)(*([cx] + [0] * (A.MAX_VALUES - 1))), A.SINK_AGG)
                cc, aggs, fn = A._flat([XVS], len(XVS)), (A.rdf_agg_result * A.MAX_VALUES)(), api._fn("pipeline")
                report("filter_sum_validity_chunked_1024_descriptors_prebuilt", 8.125 * ns_,
                       lambda: api._check(fn(C.byref(prog), cc, C.c_int32(1), C.c_int64(len(XVS)), None, aggs)), rows=ns_)
    lib.set_option("vec_bitmap", 0)
    return results


if __name__ == "__main__":
    main()
