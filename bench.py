#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: filter(x > 0.5) -> sum over f64 Arrow rows.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run with one rank per GPU.  One "step" = one pass of the fused hot path
(rdf_pipeline, RDF_MEM_DEVICE) over the rank's HBM-resident RecordBatch of synthetic rows.  W untimed
warm-up steps, then exactly K timed steps bracketed by barrier + torch.cuda.synchronize(); the MAX over
ranks is the job time; rank 0 prints ONE JSON line.

  value     = rows all ranks processed / job time            (whole-job rows/s, inputs resident in HBM)
  roofline  = algorithmic bytes per launch (8 B/row) / average duration of the dominant
              kernel (the specialised filter->aggregate kernel, rdf_spec.hip), from hipEvents the library records on ITS stream around
              that launch, against the 8 TB/s HBM3E peak (MI355X_MICROARCH.md); `peak_measured` = a stock read-only
              probe (torch.sum over the same column) timed in the same run, the second denominator SURVEY.md §8d asks for
  cpu_baseline = the oracle's structurally faithful restatement of the reference CPU path (const-array
              materialisation -> f64 casts -> compare -> bitmap -> Column::filter -> sum, one thread,
              2^20-row chunks) on a bounded sample, median of 5 warmed runs, rank 0 at N = 1 only; `c1` = BASELINE
              config 1 (1e6 rows in 1024-row batches, sin(x + 1.0) -> sum, unfused).  Test infrastructure
              used here as the reported baseline only, never as the measured product.

Multi-GPU: row ranges are sharded across ranks (rank r owns rows [r*R, (r+1)*R), weak scaling); the only
exchange of the headline is the tiny combine of per-rank {sum, count} partials (all_gather over RCCL, folded in rank
order for determinism).  `--workload c4` (hash GROUP BY) has the one real exchange: device-resident partial groups
-> all_to_all_single over RCCL -> local merge on device buffers.

Every synthetic column is a counter-based hash of (seed, column id, global row) (rdf_fill_uniform_*), so any prefix can
be re-created on the host: each workload checks its device result on a sample prefix against the oracle
(`parity_on_sample`) and through size-independent invariants on the full result (`self_check`).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
THRESHOLD = 0.5
SEED = 42
CHECK_ROWS = 2_000_000  # prefix every --workload is checked on against the oracle


KERNEL_SOURCES = ("rdf_spec_kernel.hip.h", "rdf_spec.hip", "rdf_common.hip.h", "rdf_expr.hip.h", "rdf_gspec_kernel.hip.h", "rdf_groupby.hip", "rdf_eval.hip")


def kernel_sources_sha():
    """sha256 over the sources of the kernels whose HBM traffic profiles/hbm_traffic.json holds: a traffic figure measured on other
    sources is not reported (tools/collect_profiles.py writes the same stamp next to the figures)."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "rust_dataframe_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _traffic_current(tj):
    """-> (ok, note): the committed traffic figures belong to the kernels of this tree."""
    try:
        stamp, mine = tj.get("kernel_sources_sha"), kernel_sources_sha()
    except Exception as ex_:
        return False, f"kernel sources not readable ({ex_})"
    if stamp != mine:
        return False, f"profiles/hbm_traffic.json was measured on other kernel sources (stamp {stamp}, this tree {mine}): not reported — re-run tools/profile_round.sh"
    return True, None


def _median_time(fn, runs=5, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), r, ts


def cpu_baseline(sample_rows: int):
    """Time the oracle (kind = "port") on rows [0, sample_rows) of the same synthetic column: median of 5 warmed runs."""
    import numpy as np
    from oracle import oracle
    from rust_dataframe_amd import _abi as A
    o = oracle.api()
    x = np.empty(sample_rows, dtype=np.float64)
    o.lib.ora_fill_uniform_f64(x.ctypes.data, sample_rows, SEED, 0, 0, 0.0, 1.0)
    chunk = 1 << 20
    chunks = [A.HostArray(x, None, i, min(chunk, sample_rows - i), A.F64, 0) for i in range(0, sample_rows, chunk)]
    e = A.Expr()
    c = e.col(0)
    pred = e.op("gt", c, e.scalar(THRESHOLD))
    dt, r, ts = _median_time(lambda: o.pipeline(e, [chunks], [c], pred)[0])
    # "all host cores" variant: the same path over disjoint chunk ranges on every core, like rayon over
    # chunks (src/functions/scalar.rs:28); ctypes releases the GIL during the call.  Reported beside, never instead of, `value`.
    import concurrent.futures
    ncores = os.cpu_count() or 1
    parts = [chunks[i::ncores] for i in range(ncores) if chunks[i::ncores]]
    with concurrent.futures.ThreadPoolExecutor(len(parts)) as ex:
        dt_all, rs, _ = _median_time(lambda: list(ex.map(lambda p: o.pipeline(e, [p], [c], pred)[0], parts)), runs=3)
    assert sum(q.count for q in rs) == r.count
    # BASELINE config 1 (the reference's own CPU-runnable case): 1e6 f64 rows in the CSV reader's 1024-row batches
    # (src/dataframe.rs:352), y = sin(x + 1.0) materialised step by step (src/evaluation.rs:66-96), then sum(y)
    n1 = 1_000_000
    c1_chunks = [A.HostArray(x, None, i, min(1024, n1 - i), A.F64, 0) for i in range(0, min(n1, sample_rows), 1024)]
    e1 = A.Expr()
    y = e1.op("sin", e1.op("add", e1.col(0), e1.scalar(1.0)))
    dt1, r1, _ = _median_time(lambda: o.pipeline(e1, [c1_chunks], [y])[0])
    # the all-cores number on a region of at least a second (a 70 ms region swung 7e8 <-> 1.1e9 rows/s between boxes)
    reps = max(1, int(1.0 / max(dt_all, 1e-3)))
    with concurrent.futures.ThreadPoolExecutor(len(parts)) as ex:
        t0 = time.perf_counter()
        for _ in range(reps):
            list(ex.map(lambda p: o.pipeline(e, [p], [c], pred)[0], parts))
        dt_all = (time.perf_counter() - t0) / reps
    return {"value": sample_rows / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"rows [0,{sample_rows}) of the same column, 2^20-row chunks, reference-shaped unfused path "
                      f"(oracle/rdf_oracle.c ora_pipeline), median of 5 warmed runs ({dt:.2f} s each; min {min(ts):.2f}, max {max(ts):.2f})",
            "all_cores": {"value": sample_rows / dt_all, "cores": len(parts), "seconds": dt_all, "passes_timed": reps},
            "_c1_chunks": c1_chunks,
            "c1": {"value": len(c1_chunks) * 1024 / dt1 if dt1 > 0 else 0.0, "unit": "rows/s", "cores": 1, "kind": "port",
                   "sample": f"BASELINE config 1: {len(c1_chunks)} batches of 1024 rows, sin(x + 1.0) -> sum, unfused; median of 5 warmed runs, {dt1 * 1e3:.1f} ms each",
                   "result_sum": r1.sum},
            "_sum": r.sum, "_count": r.count}


# ---------------------------------------------------------------------------------------------------------
# The other BASELINE.json configurations, selectable with --workload (the driver's default run stays the headline).
# Each builder returns (step, algorithmic bytes per launch, description, check) for this rank's HBM-resident shard;
# check(result of the last step) -> dict with `self_check` (invariants of the full-size result) and, on rank 0,
# `parity_on_sample` (device vs oracle on the first CHECK_ROWS rows) and the oracle's rows/s on that sample.

class Gen:
    """Race-free synthetic columns: the library fills on ITS stream (and synchronises it), torch converts on its own;
    torch's stream is drained before the library touches a buffer the caching allocator may have just recycled."""

    def __init__(self, torch, lib, dev):
        self.torch, self.lib, self.dev = torch, lib, dev

    def f64(self, rows, col, first, lo, hi):
        self.torch.cuda.synchronize()
        t = self.torch.empty(rows, dtype=self.torch.float64, device=self.dev)
        self.lib.fill_uniform_f64(t.data_ptr(), rows, SEED, col, first, lo, hi)
        return t

    def i64(self, rows, col, first, lo, hi):
        self.torch.cuda.synchronize()
        t = self.torch.empty(rows, dtype=self.torch.int64, device=self.dev)
        self.lib.fill_uniform_i64(t.data_ptr(), rows, SEED, col, first, lo, hi)
        return t

    def done(self):
        self.torch.cuda.synchronize()
        self.lib.synchronize()


def _host_f64(o, rows, col, first, lo, hi):
    import numpy as np
    x = np.empty(rows, dtype=np.float64)
    o.lib.ora_fill_uniform_f64(x.ctypes.data, rows, SEED, col, first, lo, hi)
    return x


def _host_i64(o, rows, col, first, lo, hi):
    import numpy as np
    x = np.empty(rows, dtype=np.int64)
    o.lib.ora_fill_uniform_i64(x.ctypes.data, rows, SEED, col, first, lo, hi)
    return x


def _close(a, b, rtol=1e-6):
    return a == b or abs(a - b) <= rtol * max(abs(a), abs(b))


def workload_c3(torch, lib, api, A, sharding, dev, comm_dev, rank, rows, first, total, opts):
    """C3: fused a*b+c -> min/max/count and the i64 key's min/max/count, one pass over 4 columns (32 B/row)."""
    g = Gen(torch, lib, dev)
    ts = [g.f64(rows, cid, first, -1.0, 1.0) for cid in range(3)] + [g.i64(rows, 3, first, -2 ** 31, 2 ** 31)]
    g.done()
    dts = (A.F64, A.F64, A.F64, A.I64)
    cols = [[A.DeviceArray(t.data_ptr(), None, 0, rows, dt, 0, keep=t)] for t, dt in zip(ts, dts)]
    e = A.Expr()
    fma = e.op("add", e.op("multiply", e.col(0), e.col(1)), e.col(2))
    roots = [fma, e.col(3)]

    native = opts.get("native")

    def step():
        if native is not None and os.environ.get("RDF_BENCH_FUSED_COMBINE"):
            y, kk = native.pipeline_dist(e, cols, roots)        # kernel + all-gather + fold on the device, one host wait
        else:
            local = api.pipeline(e, cols, roots)
            y, kk = native.agg_combine(local) if native is not None else sharding.all_combine(local, device=comm_dev)
        return {"min_y": y.min, "max_y": y.max, "count_y": y.count, "min_k": kk.min, "max_k": kk.max, "count_k": kk.count}

    def check(res, world):
        out = {"self_check": bool(res["count_y"] == total == res["count_k"] and -2.0 <= res["min_y"] < res["max_y"] <= 2.0
                                  and -2 ** 31 <= res["min_k"] < res["max_k"] < 2 ** 31)}
        if rank == 0:
            from oracle import oracle
            o = oracle.api()
            n = min(CHECK_ROWS, rows)
            sub = [[A.DeviceArray(t.data_ptr(), None, 0, n, dt, 0)] for t, dt in zip(ts, dts)]
            gy, gk = api.pipeline(e, sub, roots)
            hc = [[A.HostArray.from_numpy(_host_f64(o, n, cid, 0, -1.0, 1.0))] for cid in range(3)] + [[A.HostArray.from_numpy(_host_i64(o, n, 3, 0, -2 ** 31, 2 ** 31))]]
            dt_o, (oy, ok), _ = _median_time(lambda: o.pipeline(e, hc, roots), runs=3)
            out["parity_on_sample"] = bool(gy.count == oy.count and gy.min == oy.min and gy.max == oy.max and _close(gy.sum, oy.sum)
                                           and (gk.min, gk.max, gk.count, gk.sum) == (ok.min, ok.max, ok.count, ok.sum))
            out["cpu_baseline"] = {"value": n / dt_o, "unit": "rows/s", "cores": 1, "kind": "port",
                                   "sample": f"rows [0,{n}): multiply -> add materialised, then min/max/count (oracle), median of 3 warmed runs"}
        return out
    return step, 32.0 * rows, f"C3: fused a*b+c -> min/max/count + i64 key min/max/count over {rows:.0e} rows x 4 columns per GPU", check


def workload_c4(torch, lib, api, A, sharding, dev, comm_dev, rank, rows, first, total, opts, ngroups=1_000_000):
    """C4: SELECT key, sum(val) GROUP BY key, 1e6 keys: local hash aggregate, then (N > 1) the all-to-all of partial groups."""
    g = Gen(torch, lib, dev)
    kk = g.i64(rows, 7, first, 0, ngroups)
    v = g.f64(rows, 0, first, 0.0, 1.0)
    g.done()
    K = A.DeviceArray(kk.data_ptr(), None, 0, rows, A.I64, 0, keep=kk)
    V = A.DeviceArray(v.data_ptr(), None, 0, rows, A.F64, 0, keep=v)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    exchange_on = world > 1 or opts.get("force_exchange", False)     # a 1-rank communicator still runs the whole exchange
    cap = ngroups + 2

    def outs3():
        bufs = [torch.empty(cap * 8 + 64, dtype=torch.uint8, device=dev) for _ in range(3)]
        return tuple(A.DeviceArray(b.data_ptr(), None, 0, cap, dt, 0, keep=b) for b, dt in zip(bufs, (A.I64, A.F64, A.I64)))
    outs = outs3()
    native = opts.get("native")           # the library's own communicator (RCCL behind the C ABI); None: the torch.distributed harness
    ex = sharding.GroupExchange(api, lib, torch, dev, comm_dev, cap) if (exchange_on and native is None) else None
    last = {}

    # with about as many groups as rows pre-aggregation cannot shrink the shard: the rows are shuffled instead
    shuffle = exchange_on and (os.environ.get("RDF_C4_SHUFFLE_ROWS") == "1" or sharding.shuffle_rows_pays(rows, ngroups))

    merged = outs3() if native is not None else None

    def step():
        if native is not None and exchange_on:
            # rdf_groupby_agg_dist: local aggregate -> pack -> ncclSend / ncclRecv -> merge, all inside the library
            mk_, ms_, mc_ = native.groupby_agg([K], [V], "sum", ngroups, merged, "rows" if os.environ.get("RDF_C4_SHUFFLE_ROWS") == "1" else "auto")
            last["groups"] = (mk_, ms_, mc_)
            opts.setdefault("exchange_ms_total", [0.0])[0] += float(native.stats.get("exchange_ms", 0.0))
            return {"groups_owned": mk_.length, **native.stats}
        if shuffle:
            ok_, mk2, mc2 = ex.shuffle_rows_and_aggregate(K, V, ngroups)
            last["groups"] = (ok_[0], mk2, mc2)
            return {"groups_owned": ok_[0].length, "exchange": "rows", **ex.stats}
        gk, gs, gc = api.groupby_sum([K], [V], ngroups, outs)
        ng = gk.length
        if not exchange_on:
            last["groups"] = (gk, gs, gc)
            return {"groups": ng}
        # partial groups of this rank -> owners (RCCL all-to-all on device buffers), merged there by a second, tiny group-by
        mk_, ms_, mc_ = ex.exchange_and_merge(gk, gs, gc, ngroups)
        last["groups"] = (mk_, ms_, mc_)
        return {"groups_owned": mk_.length, "exchange": "partial groups", **ex.stats}

    def check(res, world_):
        import numpy as np
        gk, gs, gc = last["groups"]
        n = gk.length
        hk = torch.empty(n, dtype=torch.int64); hs = torch.empty(n, dtype=torch.float64); hc = torch.empty(n, dtype=torch.int64)
        L = lib.load()
        L.rdf_copy_d2h(hk.data_ptr(), gk.values_ptr, n * 8); L.rdf_copy_d2h(hs.data_ptr(), gs.values_ptr, n * 8); L.rdf_copy_d2h(hc.data_ptr(), gc.values_ptr, n * 8)
        hk, hs, hc = hk.numpy(), hs.numpy(), hc.numpy()
        # invariants of the full-size result: no key twice, every count positive, counts add up to the rows and the
        # group sums to the column sum (all ranks' shares together when N > 1)
        e = A.Expr()
        lsum = api.pipeline(e, [[V]], [e.col(0)])
        col_sum = (native.agg_combine(lsum) if native is not None else sharding.all_combine(lsum, device=comm_dev))[0].sum
        tot = [float(hs.sum()), int(hc.sum()), int(n)]
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            t = torch.tensor(tot, dtype=torch.float64, device=comm_dev if comm_dev is not None else "cpu")
            dist.all_reduce(t)
            tot = [t[0].item(), int(t[1].item()), int(t[2].item())]
        ok = (len(np.unique(hk)) == n and int(hc.min(initial=1)) >= 1 and tot[1] == total and _close(tot[0], col_sum, 1e-9)
              and tot[2] <= ngroups and (total < 20 * ngroups or tot[2] == ngroups) and (n == 0 or (0 <= hk.min() and hk.max() < ngroups)))
        out = {"self_check": bool(ok), "groups_total": tot[2], "check_totals": {"sum_of_group_sums": tot[0], "column_sum": col_sum, "sum_of_group_counts": tot[1], "rows": total}}
        if rank == 0:
            from oracle import oracle
            o = oracle.api()
            m = min(CHECK_ROWS, rows)
            Ks, Vs = A.DeviceArray(kk.data_ptr(), None, 0, m, A.I64, 0), A.DeviceArray(v.data_ptr(), None, 0, m, A.F64, 0)
            dk, ds, dc = api.groupby_sum([Ks], [Vs], ngroups, outs3())
            dn = dk.length
            a = [torch.empty(dn, dtype=d) for d in (torch.int64, torch.float64, torch.int64)]
            for t_, src in zip(a, (dk, ds, dc)):
                L.rdf_copy_d2h(t_.data_ptr(), src.values_ptr, dn * 8)
            hkeys, hvals = A.HostArray.from_numpy(_host_i64(o, m, 7, 0, 0, ngroups)), A.HostArray.from_numpy(_host_f64(o, m, 0, 0, 0.0, 1.0))
            dt_o, (ok_, os_, oc_), _ = _median_time(lambda: o.groupby_sum([hkeys], [hvals], ngroups), runs=3)
            o1, o2 = np.argsort(a[0].numpy()), np.argsort(ok_.to_numpy())
            out["parity_on_sample"] = bool(dn == ok_.length and np.array_equal(a[0].numpy()[o1], ok_.to_numpy()[o2])
                                           and np.array_equal(a[2].numpy()[o1], oc_.to_numpy()[o2])
                                           and np.allclose(a[1].numpy()[o1], os_.to_numpy()[o2], rtol=1e-6, atol=0))
            out["cpu_baseline"] = {"value": m / dt_o, "unit": "rows/s", "cores": 1, "kind": "port",
                                   "sample": f"rows [0,{m}): hash GROUP BY key -> sum, count (oracle), median of 3 warmed runs"}
        return out
    return step, 16.0 * rows, f"C4: hash GROUP BY key -> sum(val), {ngroups:.0e} keys over {rows:.0e} rows per GPU", check


def q1_program(A):
    q = A.Expr()
    c = [q.col(i) for i in range(7)]
    pred = q.op("le", c[6], q.scalar(10471, A.I32))
    gid = q.op("add", q.op("multiply", q.cast(c[4], A.I32), q.scalar(2, A.I32)), q.cast(c[5], A.I32))
    dp = q.op("multiply", c[1], q.op("subtract", q.scalar(1.0), c[2]))
    ch = q.op("multiply", dp, q.op("add", q.scalar(1.0), c[3]))
    return q, [c[0], c[1], dp, ch, c[2]], gid, pred


def workload_q1(torch, lib, api, A, sharding, dev, comm_dev, rank, rows, first, total, opts):
    """C5: TPC-H Q1 shape over a synthetic lineitem shard (38 B/row): filter(shipdate <= c) -> 5 sums + counts in 6 groups.
    Columns per the TPC-H distributions: quantity 1..50, extendedprice, discount 0.00..0.10, tax 0.00..0.08
    as f64; returnflag (3) / linestatus (2) dictionary codes as i8; shipdate as date32 days."""
    g = Gen(torch, lib, dev)
    qty = g.i64(rows, 12, first, 1, 51).to(torch.float64)
    price = g.f64(rows, 11, first, 900.0, 105000.0)
    disc = g.i64(rows, 13, first, 0, 11).to(torch.float64) / 100.0
    tax = g.i64(rows, 14, first, 0, 9).to(torch.float64) / 100.0
    flag = g.i64(rows, 15, first, 0, 3).to(torch.int8)
    status = g.i64(rows, 16, first, 0, 2).to(torch.int8)
    ship = g.i64(rows, 17, first, 8036, 10562).to(torch.int32)
    g.done()
    ts = (qty, price, disc, tax, flag, status, ship)
    dts = (A.F64, A.F64, A.F64, A.F64, A.I8, A.I8, A.I32)
    cols = [[A.DeviceArray(t.data_ptr(), None, 0, rows, dt, 0, keep=t)] for t, dt in zip(ts, dts)]
    q, vals, gid, pred = q1_program(A)

    native = opts.get("native")

    def step():
        local = api.group_pipeline(q, cols, vals, gid, 6, pred)
        res, nrows = native.group_combine(local) if native is not None else sharding.all_combine_groups(local, device=comm_dev)
        return {"count_star": nrows[:6], "sum_qty": [r[0] for r in res[0][:6]], "sum_disc": [r[0] for r in res[4][:6]],
                "sum_price": [r[0] for r in res[1][:6]], "sum_disc_price": [r[0] for r in res[2][:6]], "sum_charge": [r[0] for r in res[3][:6]]}

    def check(res, world):
        cs = res["count_star"]
        sel = (10471 - 8036 + 1) / (10562 - 8036)
        ok = abs(sum(cs) / total - sel) < 1e-3
        for gi in range(6):
            c_ = max(cs[gi], 1)
            ok = ok and 1.0 <= res["sum_qty"][gi] / c_ <= 50.0 and 0.0 <= res["sum_disc"][gi] / c_ <= 0.10
            ok = ok and 900.0 <= res["sum_price"][gi] / c_ <= 105000.0
            ok = ok and 0.9 * res["sum_price"][gi] <= res["sum_disc_price"][gi] <= res["sum_price"][gi] <= res["sum_charge"][gi] / 0.9 <= 1.2 * res["sum_price"][gi]
            ok = ok and abs(cs[gi] / max(sum(cs), 1) - 1 / 6) < 1e-2
        out = {"self_check": bool(ok)}
        if rank == 0:
            import numpy as np
            from oracle import oracle
            o = oracle.api()
            n = min(CHECK_ROWS, rows)
            sub = [[A.DeviceArray(t.data_ptr(), None, 0, n, dt, 0)] for t, dt in zip(ts, dts)]
            gres, grows = api.group_pipeline(q, sub, vals, gid, 6, pred)
            hi = lambda col, lo, hi_: _host_i64(o, n, col, 0, lo, hi_)
            hcols = [hi(12, 1, 51).astype(np.float64), _host_f64(o, n, 11, 0, 900.0, 105000.0), hi(13, 0, 11).astype(np.float64) / 100.0,
                     hi(14, 0, 9).astype(np.float64) / 100.0, hi(15, 0, 3).astype(np.int8), hi(16, 0, 2).astype(np.int8), hi(17, 8036, 10562).astype(np.int32)]
            hc = [[A.HostArray.from_numpy(x)] for x in hcols]
            dt_o, (ores, orows), _ = _median_time(lambda: o.group_pipeline(q, hc, vals, gid, 6, pred), runs=3)
            same = list(grows) == list(orows)
            for vi in range(len(vals)):
                for gi in range(7):
                    same = same and gres[vi][gi][1] == ores[vi][gi][1] and _close(gres[vi][gi][0], ores[vi][gi][0])
            out["parity_on_sample"] = bool(same)
            out["cpu_baseline"] = {"value": n / dt_o, "unit": "rows/s", "cores": 1, "kind": "port",
                                   "sample": f"rows [0,{n}): filter -> expressions -> 6 groups x (5 sums + counts), one materialised array per step (oracle), median of 3 warmed runs"}
        return out
    return step, 38.0 * rows, f"C5: TPC-H Q1 shape (filter -> 5 sums + counts in 6 uniform groups = 3 return flags x 2 line statuses; TPC-H's own data has 4 non-empty ones) over a {rows:.0e}-row synthetic lineitem shard per GPU", check


WORKLOADS = {"c3": workload_c3, "c4": workload_c4, "q1": workload_q1}


def run_ref_bench(args, lib, api, A):
    """The reference's ONLY benchmark shape (`bench_multiply_i32`, src/functions/scalar.rs:621-671): par_multiply::<Int32Type> over
    380 chunks of 5 values [None, 200, None, -256, None] on both sides — 1900 rows in 380 arrays, the launch-latency regime in
    which a device path is expected to LOSE to the host: which is why it is timed here, end to end (host buffers in, host buffers
    out through rdf_binary) next to the oracle's restatement of the same call on one core.  The descriptor arrays are built once, as a
    caller that holds its arrays has them; every call is checked against the known answer [None, 40000, None, 65536, None]."""
    import ctypes as C
    import numpy as np
    from oracle import oracle
    o = oracle.api()
    n = 380
    vals = np.array([0, 200, 0, -256, 0], dtype=np.int32)
    valid = np.array([False, True, False, True, False])
    a = [A.HostArray.from_numpy(vals.copy(), valid=valid, dtype=A.I32) for _ in range(n)]
    ca, cb = A._flat([a], n), A._flat([a], n)
    want = np.array([200 * 200, 65536], dtype=np.int32)

    def caller(x):
        outs = [A.HostArray.empty_out(A.I32, 5, True) for _ in range(n)]
        carr = (A.rdf_out * n)(*[q.out_struct() for q in outs])
        fn = x._fn("binary")
        code = C.c_int32(A.OP_NAMES["multiply"])
        return outs, carr, (lambda: x._check(fn(code, ca, cb, C.c_int64(n), carr)))

    def ok(outs, carr):
        for i, q in enumerate(outs):
            q.length, q.null_count = carr[i].length, carr[i].null_count
            m = q.valid_mask()
            if q.length != 5 or q.null_count != 3 or list(m) != list(valid) or not np.array_equal(q.to_numpy()[m], want):
                return False
        return True

    calls = max(1000, args.steps)
    res = {}
    for name, x in (("gpu", api), ("cpu", o)):
        outs, carr, call = caller(x)
        for _ in range(max(20, args.warmup)):
            call()
        ts = []
        t_all = time.perf_counter()
        for _ in range(calls):
            t0 = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t0)
        t_all = time.perf_counter() - t_all
        res[name] = {"median_us": statistics.median(ts) * 1e6, "mean_us": t_all / calls * 1e6, "min_us": min(ts) * 1e6,
                     "p99_us": sorted(ts)[int(0.99 * (calls - 1))] * 1e6, "parity": bool(ok(outs, carr))}
    g, c_ = res["gpu"], res["cpu"]
    line = {"metric": "calls/s par_multiply::<Int32Type>, 380 chunks x 5 values (the reference's bench_multiply_i32)", "value": 1e6 / g["median_us"], "unit": "calls/s",
            "n_gpus": 1, "steps": calls, "warmup": max(20, args.warmup), "ms_per_step": g["median_us"] * 1e-3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32", "data": "the reference's own vector",
            "config": {"workload": "rdf_binary(MULTIPLY, Int32) over 380 host-resident chunks of [None, 200, None, -256, None] on both sides, host buffers in, host buffers out "
                                   "(src/functions/scalar.rs:621-671); one call = 1900 rows", "rows_per_call": 1900, "gpu": g, "kernel": lib.last_kernel()},
            "roofline": None,
            "cpu_baseline": {"value": 1e6 / c_["median_us"], "unit": "calls/s", "cores": 1, "kind": "port",
                             "sample": f"the same call through the oracle (ora_binary: one wrapping multiply loop per chunk, validity ANDed), median of {calls} calls", **c_},
            "gpu_over_cpu_time": g["median_us"] / c_["median_us"],
            "verdict": ("the host path wins this shape" if g["median_us"] > c_["median_us"] else "the device path wins this shape")
                       + f": {g['median_us']:.1f} us per call on the device path end to end against {c_['median_us']:.1f} us on one host core"}
    emit(line)
    if not (g["parity"] and c_["parity"]):
        sys.exit("bench.py --workload ref_bench: a result differs from the known answer")


def read_probe(torch, x, runs=5):
    """Stock read-only streaming kernel over the same column (torch.sum), median of `runs`: GB/s."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(runs + 1)]
    for a, b in ev:
        a.record()
        torch.sum(x)
        b.record()
    torch.cuda.synchronize()
    ms = statistics.median(a.elapsed_time(b) for a, b in ev[1:])
    return x.numel() * x.element_size() / (ms * 1e-3) / 1e9


def _self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start one rank per GPU under torch.distributed.run
    (127.0.0.1 rendezvous on a free port), pass the command line through, let rank 0's JSON line reach stdout."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RDF_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, cwd=ROOT)


_JSON_FD = None


def _quiet_stdout():
    """RCCL prints a version banner on C stdout (buffered: it would land AFTER the result).  From here on everything
    written to fd 1 — by any library, in any rank — goes to stderr, and the one JSON line is written to the real stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def _rccl_version(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="rows per GPU (weak scaling; f64, 8 B/row)")
    ap.add_argument("--total-rows", type=int, default=0,
                    help="strong scaling: this many rows in total, cut into contiguous row ranges of whole 1024-row batches over the ranks "
                         "(BASELINE config C4 as stated: --workload c4 --total-rows 1000000000 --gpus 8)")
    ap.add_argument("--cpu-sample", type=int, default=50_000_000, help="rows for the CPU baseline leg (0 = skip)")
    ap.add_argument("--null-fraction", type=float, default=0.0, help="attach a validity bitmap with this null rate")
    ap.add_argument("--chunk-rows", type=int, default=0, help="hand the column over as RecordBatches of this many rows (0 = one chunk; 1024 = the reference readers' batch)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for a smoke test)")
    ap.add_argument("--comm", default="native", choices=["native", "torch"],
                    help="who runs the N > 1 collectives with --backend nccl: native = the library's own communicator (rdf_comm_*: RCCL loaded "
                         "behind the C ABI, ncclAllGather + grouped ncclSend / ncclRecv; torch.distributed only launches the ranks, hands the "
                         "unique id around and times the run), the default; torch = the torch.distributed harness of rust_dataframe_amd/sharding.py")
    ap.add_argument("--share-gpu", action="store_true", help="smoke test only: every rank uses device 0 (needs --backend gloo)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="N = 1 only: still create the process group (a 1-rank RCCL communicator with --backend nccl) and run every "
                         "collective of the N > 1 path — all_gather of partials, the device-resident all_to_all of the group-by")
    ap.add_argument("--workload", default="headline", choices=["headline", "ref_bench"] + sorted(WORKLOADS),
                    help="headline = BASELINE.json's metric (the default, what the driver runs); c3 / c4 / q1 = the other configs; "
                         "ref_bench = the reference's own bench_multiply_i32 shape (380 chunks x 5 Int32 values), device path end to end against the oracle")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(_self_launch(args))       # the driver's plain `python bench.py --gpus N`
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    from rust_dataframe_amd import _abi as A
    from rust_dataframe_amd import lib, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} are visible "
                 "(one rank per GPU; --share-gpu --backend gloo is the single-GPU smoke mode)")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        _quiet_stdout()
        sharding.FORCE_COLLECTIVES = bool(args.force_exchange)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        want_native = args.backend == "nccl" and args.comm == "native"
        if want_native:
            # CPU tensors (the unique id, barriers, the max-over-ranks clock) travel over gloo; torch's own RCCL communicator is
            # created only if the library's cannot be (CUDA tensors of the fallback harness)
            dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world)
        elif args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    lib.set_device(local_rank)
    api = lib.api()
    dev = torch.device("cuda", local_rank)
    native, native_note = None, None
    if dist is not None and args.backend == "nccl" and args.comm == "native":
        idt = torch.zeros(A.COMM_ID_BYTES + 1, dtype=torch.uint8)
        if rank == 0:
            try:
                idt[:A.COMM_ID_BYTES] = torch.frombuffer(bytearray(A.Comm.unique_id(api)), dtype=torch.uint8)
                idt[A.COMM_ID_BYTES] = 1
            except Exception as ex_:          # no librccl for the library to load
                native_note = f"rdf_comm_unique_id: {ex_}"
        dist.broadcast(idt, 0)
        if int(idt[A.COMM_ID_BYTES]) == 1:
            try:
                native = A.Comm.init_rank(api, world, rank, bytes(idt[:A.COMM_ID_BYTES].tolist()))
            except Exception as ex_:
                native_note = f"rdf_comm_init_rank: {ex_}"
        flag = torch.tensor([1 if native is not None else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:                # every rank or none
            if native is not None:
                native.destroy()
            native, native_note = None, (native_note or "another rank could not create its communicator")
            print(f"bench.py: the library's communicator is not available ({native_note}); using torch.distributed", file=sys.stderr)
    args.native = native
    comm_dev = None if native is not None else (dev if (dist is None or args.backend == "nccl") else None)   # gloo / the control plane exchange CPU tensors
    ninfo = native.info() if native is not None else None
    comm = {"ranks": dist.get_world_size() if dist is not None else 1,
            "backend": (args.backend if dist is not None else None),
            "collectives": (None if dist is None else "librdf_mi355x (rdf_comm_*: RCCL behind the C ABI)" if native is not None
                            else "torch.distributed (sharding.py)" + (f"; native unavailable: {native_note}" if native_note else "")),
            "rccl_version": (ninfo["rccl_version"] if ninfo else _rccl_version(torch) if (dist is not None and args.backend == "nccl") else None),
            "launcher": "self (torch.distributed.run)" if os.environ.get("RDF_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if world > 1 else None)}

    # weak scaling: every rank owns --rows rows; strong scaling (--total-rows): contiguous ranges of whole 1024-row batches
    if args.total_rows > 0:
        first_row, end_row = sharding.shard_rows(args.total_rows, world, rank)
        rows, total_rows, scaling = end_row - first_row, args.total_rows, "strong"
    else:
        rows, first_row, total_rows, scaling = args.rows, rank * args.rows, args.rows * world, "weak"
    args.rows, args.first_row, args.total, args.scaling, args.comm = rows, first_row, total_rows, scaling, comm
    if args.workload == "ref_bench":
        if world != 1:
            sys.exit("bench.py --workload ref_bench runs on one GPU")
        return run_ref_bench(args, lib, api, A)
    if args.workload != "headline":
        return run_other(args, torch, lib, api, A, sharding, dev, comm_dev, dist, rank, world)
    g = Gen(torch, lib, dev)
    x = g.f64(rows, 0, first_row, 0.0, 1.0)
    vptr, vkeep = None, None
    if args.null_fraction > 0:
        vkeep = torch.zeros((rows + 63) // 64 * 8 + 64, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        lib.fill_validity(vkeep.data_ptr(), rows, SEED, 0, first_row, args.null_fraction)
        vptr = vkeep.data_ptr()
    g.done()
    if args.chunk_rows > 0:
        cr = args.chunk_rows
        assert cr % 64 == 0, "--chunk-rows must keep chunk bitmaps 8-byte aligned"
        col = [A.DeviceArray(x.data_ptr() + 8 * i, (vptr + i // 8) if vptr else None, 0, min(cr, rows - i), A.F64, -1, keep=(x, vkeep))
               for i in range(0, rows, cr)]
    else:
        col = [A.DeviceArray(x.data_ptr(), vptr, 0, rows, A.F64, -1, keep=(x, vkeep))]

    # One chunk: the plain call.  A frame of RecordBatches (--chunk-rows): pinned once with rdf_frame_pin, as a DataFrame holding
    # its batches for its lifetime would be — the per-call walk over a million descriptors (14 ms) is not part of a query.
    frame = A.PinnedFrame(api, [col]) if args.chunk_rows else A.Prepared([col])
    e = A.Expr()
    c = e.col(0)
    pred = e.op("gt", c, e.scalar(THRESHOLD))
    combine_s = [0.0]

    # rdf_pipeline_dist (kernel + all-gather + fold on the device, one host wait) against rdf_pipeline + rdf_agg_combine (two waits):
    # on the 1-rank RCCL communicator the pair is the faster one (0.2845 against 0.2931 ms/step at 2e8 rows: the second all-gather
    # and the fold kernel cost more than the host round trip they replace), so the pair stays the default; RDF_BENCH_FUSED_COMBINE=1
    # times the fused call (profiles/r05_rccl_one_rank.jsonl has both)
    fused_combine = native is not None and bool(os.environ.get("RDF_BENCH_FUSED_COMBINE"))

    def step():
        if fused_combine:
            # N > 1, the library's communicator: kernel, all-gather of the partials and their fold in ONE call, all on the device
            # (rdf_pipeline_dist / rdf_pipeline_frame_dist): the host waits once per step
            tot = native.pipeline_dist(e, frame, [c], pred)[0]
            return tot.sum, tot.count
        local = api.pipeline(e, frame, [c], pred)          # fused filter -> {sum,min,max,count}, one pass over HBM
        tc = time.perf_counter()
        # N > 1: all-gather the partials (RCCL), fold in rank order — rdf_agg_combine inside the library, or the torch harness
        tot = (native.agg_combine(local) if native is not None else sharding.all_combine(local, device=comm_dev))[0]
        combine_s[0] += time.perf_counter() - tc
        return tot.sum, tot.count

    def sync():
        torch.cuda.synchronize()
        lib.synchronize()

    barrier = _barrier(torch, dist, native)
    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    sync()
    lib.kernel_timing_reset(True)
    combine_s[0] = 0.0          # (the warm-up steps carry the communicator's lazy initialisation)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = lib.kernel_timing_get()
    kernel_name = lib.last_kernel()
    lib.kernel_timing_reset(False)
    per_rank = None
    if dist is not None:
        # every rank's own share of the result (its shard alone, one more untimed call) travels with its times: the shares must add up
        # to the combined result BEFORE anything is printed, and a rank that ran slow or on the wrong rows shows in the line
        mine = api.pipeline(e, frame, [c], pred)[0]
        per_rank = _per_rank(torch, dist, comm_dev, elapsed, kern_ms, kern_n, args.steps, None, [float(mine.count), float(rows) * 8.0 + (rows / 8.0 if vptr else 0.0)])
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev if comm_dev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        counts = [int(v) for v in per_rank.pop("extra0")]
        algb = per_rank.pop("extra1")
        per_rank["result_count"] = counts
        per_rank["roofline_frac"] = [round(b / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k > 0 else None for b, k in zip(algb, per_rank["kernel_ms_per_step"])]
        if sum(counts) != res[1]:
            sys.exit(f"bench.py: the ranks' result counts {counts} add up to {sum(counts)}, the combined result says {res[1]}")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rows * args.steps / elapsed
        alg_bytes = rows * 8.0 + (rows / 8.0 if vptr else 0.0)   # per launch (one rank's launch)
        avg_kernel_s = kern_ms / max(kern_n, 1) * 1e-3
        achieved = alg_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):   # PMC-derived HBM bytes per launch, measured in a separate rocprofv3 --pmc pass
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if int(tj.get("rows", -1)) == rows and bool(tj.get("validity", False)) == bool(vptr) and args.chunk_rows == 0:
                    cur, note = _traffic_current(tj)
                    if cur:
                        traffic = tj.get("hbm_bytes_per_launch")
                        traffic_source = ("profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (" + str(tj.get("source", "committed profile"))
                                          + "), kernel sources " + str(tj.get("kernel_sources_sha")) + " = this tree's; not re-measured in this run")
                    else:
                        traffic_source = note
            except Exception:
                traffic = None
        probe_torch = read_probe(torch, x)
        # the second denominator: the best BARE 16-byte streaming read of the same column (rdf_probe_stream: nothing but an xor fold
        # behind the loads, several loop shapes and grids) — a ceiling the kernel can be held against, which torch.sum (slower than
        # the kernel by 15 %) was not
        try:
            probe, probe_shape = lib.probe_stream(0, x.data_ptr(), nbytes=rows * 8, reps=5)
        except Exception as ex_:
            probe, probe_shape = 0.0, f"rdf_probe_stream failed: {ex_}"
        out = {
            "metric": "rows/sec filter+sum over 1e9 f64 Arrow rows; %HBM bw at 1/2/4/8 GPU",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"filter(x>{THRESHOLD})->sum over a {rows:.0e}-row f64 Arrow RecordBatch per GPU, HBM-resident"
                                   + (f", {args.null_fraction:.0%} nulls (validity bitmap)" if vptr else ", no validity bitmap")
                                   + (f", {len(col)} RecordBatches of {args.chunk_rows} rows" if args.chunk_rows else ""),
                       "rows_per_gpu": rows, "total_rows": total_rows, "selectivity": res[1] / total_rows,
                       "result_sum": res[0], "result_count": res[1], "sharding": "row ranges per rank, no data-path collective",
                       "combine": ("on the device inside rdf_pipeline_dist (all-gather + fold behind the kernel, one host wait per step)" if fused_combine
                                   else "on the host after rdf_pipeline (all-gather of the partials, fold in rank order)"),
                       "combine_ms_per_step": ((elapsed / max(args.steps, 1) - kern_ms / max(kern_n, 1) * 1e-3 * (kern_n / args.steps if kern_n else 0)) * 1e3
                                               if fused_combine else combine_s[0] / max(args.steps, 1) * 1e3),
                       **({"per_rank": per_rank} if per_rank else {}), **comm},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "frac_wall": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,     # the same bytes over the wall-clock step (launch, result copy, combine included): what the driver's own clock can check
                         "peak_measured": probe, "peak_measured_kind": "best bare read-only stream over the same column (rdf_probe_stream: " + probe_shape + ")",
                         "frac_of_measured": achieved / probe if probe > 0 else None,
                         "torch_sum_GBps": probe_torch,
                         "kernel": kernel_name, "avg_kernel_ms": avg_kernel_s * 1e3, "launches": kern_n,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and args.cpu_sample > 0:
            sample = min(args.cpu_sample, rows)
            cb = cpu_baseline(sample)
            # not timed: the device result on the same sample prefix agrees with the oracle
            if not vptr:
                sub = A.DeviceArray(x.data_ptr(), None, 0, sample, A.F64, -1)
                gq = api.pipeline(e, [[sub]], [c], pred)[0]
                ok = gq.count == cb["_count"] and abs(gq.sum - cb["_sum"]) <= 1e-6 * abs(cb["_sum"])
                cb["parity_on_sample"] = bool(ok)
                # config 1 on the device: the same 977 batches through the C ABI (host buffers in, fused sin(x + 1.0) -> sum)
                e1 = A.Expr()
                y = e1.op("sin", e1.op("add", e1.col(0), e1.scalar(1.0)))
                n1 = min(1_000_000, sample)
                d1 = [A.DeviceArray(x.data_ptr() + 8 * i, None, 0, min(1024, n1 - i), A.F64, 0) for i in range(0, n1, 1024)]
                g1 = api.pipeline(e1, [d1], [y])[0]
                cb["c1"]["parity"] = bool(abs(g1.sum - cb["c1"]["result_sum"]) <= 1e-6 * abs(cb["c1"]["result_sum"]))
                # ... and END TO END as the reference meets it: the 977 batches in HOST memory in, the scalar out (staging over
                # PCIe + one fused kernel + the result coming back), through the same entry point
                hc = cb["_c1_chunks"]
                dt_py, gh, _ = _median_time(lambda: api.pipeline(e1, [hc], [y])[0], runs=9, warm=2)
                # the library's own time: the 977 descriptors built once, as a host that holds its RecordBatches has them (the
                # ctypes marshalling of 977 structs per call is Python's cost, not the boundary's)
                import ctypes as C
                cc = A._flat([hc], len(hc))
                prog = A.rdf_program(C.cast(e1.c_array(), C.POINTER(A.rdf_expr_node)), len(e1.nodes), -1, 1, (C.c_int32 * A.MAX_VALUES)(*([y] + [0] * (A.MAX_VALUES - 1))), A.SINK_AGG)
                aggs = (A.rdf_agg_result * A.MAX_VALUES)()
                fn = api._fn("pipeline")
                dt_g, _, ts_g = _median_time(lambda: api._check(fn(C.byref(prog), cc, C.c_int32(1), C.c_int64(len(hc)), None, aggs)), runs=15, warm=3)
                cb["c1"]["gpu_end_to_end"] = {"ms": dt_g * 1e3, "ms_min": min(ts_g) * 1e3, "value": len(hc) * 1024 / dt_g, "unit": "rows/s",
                                              "ms_with_python_marshalling": dt_py * 1e3,
                                              "what": f"{len(hc)} host-resident 1024-row batches in -> sum(sin(x + 1.0)) out, rdf_pipeline (RDF_MEM_HOST), descriptor array built once, median of 15 warmed calls",
                                              "parity": bool(abs(gh.sum - cb["c1"]["result_sum"]) <= 1e-6 * abs(cb["c1"]["result_sum"]) and abs(aggs[0].sum_f64 - gh.sum) <= 1e-9 * abs(gh.sum))}
            cb.pop("_sum"), cb.pop("_count"), cb.pop("_c1_chunks", None)
            out["cpu_baseline"] = cb
        emit(out)
    if native is not None:
        native.destroy()
    if dist is not None:
        dist.destroy_process_group()


def _per_rank(torch, dist, comm_dev, elapsed, kern_ms, kern_n, steps, extra=None, more=None):
    """What every rank measured, gathered on all ranks (rank order): wall and kernel time per step, and whatever the workload adds
    (exchange time) — so that a scaling curve explains itself: a slow rank, a slow kernel or a slow exchange."""
    mine = [elapsed / max(steps, 1) * 1e3, kern_ms / max(kern_n, 1) * (kern_n / steps if kern_n else 0)] + list(extra or []) + list(more or [])
    t = torch.tensor(mine, dtype=torch.float64, device=comm_dev if comm_dev is not None else "cpu")
    allr = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(allr, t)
    rows = [[round(float(x), 4) for x in r.tolist()] for r in allr]
    out = {"step_ms": [r[0] for r in rows], "kernel_ms_per_step": [r[1] for r in rows]}
    if extra is not None:
        out["exchange_ms_per_step"] = [r[2] for r in rows]
    base = 2 + len(extra or [])
    for j in range(len(more or [])):
        out[f"extra{j}"] = [float(a[base + j]) for a in allr]      # (unrounded: counts)
    return out


def _barrier(torch, dist, native):
    """All ranks meet.  With the library's communicator the process group is a control plane (gloo for CPU tensors): a tiny
    all_reduce of a CPU tensor, so that torch never creates an RCCL communicator of its own next to the library's."""
    if dist is None:
        return lambda: None
    if native is not None:
        z = torch.zeros(1, dtype=torch.int32)
        return lambda: dist.all_reduce(z)
    return dist.barrier


def run_other(args, torch, lib, api, A, sharding, dev, comm_dev, dist, rank, world):
    """Same timing contract as the headline, for the other BASELINE.json configurations."""
    rows, total = args.rows, args.total
    native = args.native
    opts = {"force_exchange": args.force_exchange, "native": native}
    barrier = _barrier(torch, dist, native)
    step, alg_bytes, desc, check = WORKLOADS[args.workload](torch, lib, api, A, sharding, dev, comm_dev, rank, rows, args.first_row, total, opts)

    def sync():
        torch.cuda.synchronize()
        lib.synchronize()
    sync()   # the synthetic columns were written on torch's stream; the library launches on its own
    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    sync()
    lib.kernel_timing_reset(True)
    if "exchange_ms_total" in opts:
        opts["exchange_ms_total"][0] = 0.0          # (the warm-up steps' exchanges)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = lib.kernel_timing_get()
    kernel_name = lib.last_kernel()
    lib.kernel_timing_reset(False)
    per_rank = None
    if dist is not None:
        ex = opts.get("exchange_ms_total")
        per_rank = _per_rank(torch, dist, comm_dev, elapsed, kern_ms, kern_n, args.steps, None if ex is None else [ex[0] / max(args.steps, 1)])
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev if comm_dev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    chk = check(res, world)     # untimed; collective when N > 1 (every rank takes part)
    if native is not None:
        native.destroy()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        per_launch_s = kern_ms / max(kern_n, 1) * 1e-3 * (kern_n / args.steps if kern_n else 0)   # all timed kernels of one step
        achieved = alg_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        cb = chk.pop("cpu_baseline", None)
        traffic, traffic_source = None, None
        try:   # PMC-derived HBM bytes of one step's kernels, measured in separate rocprofv3 --pmc passes (tools/profile_round.sh)
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                tj_all = json.load(f)
            tw = tj_all.get("workloads", {}).get(args.workload)
            if tw and int(tw.get("rows", -1)) == rows and world == 1:
                cur, note = _traffic_current(tj_all)
                if cur:
                    traffic = tw.get("hbm_bytes_per_step")
                    traffic_source = "profiles/hbm_traffic.json (" + str(tw.get("source")) + "), kernel sources " + str(tj_all.get("kernel_sources_sha")) + " = this tree's; not re-measured in this run"
                else:
                    traffic_source = note
        except Exception:
            traffic = None
        line = {
            "metric": f"rows/sec {args.workload}", "value": total * args.steps / elapsed, "unit": "rows/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "rows_per_gpu": rows, "total_rows": total, "result": res, **chk, **({"per_rank": per_rank} if per_rank else {}), **args.comm},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "kernel": kernel_name, "kernel_ms_per_step": per_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if cb is not None:
            line["cpu_baseline"] = cb
        emit(line)
        if not chk.get("self_check", False) or chk.get("parity_on_sample") is False:
            sys.exit(f"bench.py --workload {args.workload}: result check FAILED: {chk}")


if __name__ == "__main__":
    main()
