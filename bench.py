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
              that launch, against the 8 TB/s HBM3E peak (MI355X_MICROARCH.md)
  cpu_baseline = the oracle's structurally faithful restatement of the reference CPU path (const-array
              materialisation -> f64 casts -> compare -> bitmap -> Column::filter -> sum, one thread,
              2^20-row chunks) timed on a bounded sample, rank 0 at N = 1 only.  Test infrastructure
              used here as the reported baseline only, never as the measured product.

Multi-GPU: row ranges are sharded across ranks (rank r owns rows [r*R, (r+1)*R), weak scaling); the only
exchange is the tiny combine of per-rank {sum, count} partials (all_gather over RCCL, folded in rank
order for determinism).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
THRESHOLD = 0.5
SEED = 42


def cpu_baseline(sample_rows: int, gpu_check=None):
    """Time the oracle (kind = "port") on rows [0, sample_rows) of the same synthetic column."""
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
    t0 = time.perf_counter()
    r = o.pipeline(e, [chunks], [c], pred)[0]
    dt = time.perf_counter() - t0
    # "all host cores" variant: the same path over disjoint chunk ranges on every core, like rayon over
    # chunks (src/functions/scalar.rs:28); ctypes releases the GIL during the call.  Reported beside, never instead of, `value`.
    import concurrent.futures
    ncores = os.cpu_count() or 1
    parts = [chunks[i::ncores] for i in range(ncores) if chunks[i::ncores]]
    t1 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(len(parts)) as ex:
        rs = list(ex.map(lambda p: o.pipeline(e, [p], [c], pred)[0], parts))
    dt_all = time.perf_counter() - t1
    assert sum(x.count for x in rs) == r.count
    return {"value": sample_rows / dt, "unit": "rows/s", "cores": 1, "kind": "port",
            "sample": f"rows [0,{sample_rows}) of the same column, 2^20-row chunks, reference-shaped unfused path "
                      f"(oracle/rdf_oracle.c ora_pipeline), {dt:.2f} s",
            "all_cores": {"value": sample_rows / dt_all, "cores": len(parts), "seconds": dt_all},
            "_sum": r.sum, "_count": r.count}


# ---------------------------------------------------------------------------------------------------------
# The other BASELINE.json configurations, selectable with --workload (the driver's default run stays the headline):
# each returns (step, alg_bytes_per_launch, description, check) for this rank's HBM-resident shard.

def workload_c3(torch, lib, api, A, sharding, dev, comm_dev, rank, rows):
    """C3: fused a*b+c -> min/max/count and the i64 key's min/max/count, one pass over 4 columns (32 B/row)."""
    first = rank * rows
    cols = []
    for cid in range(3):
        t = torch.empty(rows, dtype=torch.float64, device=dev)
        lib.fill_uniform_f64(t.data_ptr(), rows, SEED, cid, first, -1.0, 1.0)
        cols.append([A.DeviceArray(t.data_ptr(), None, 0, rows, A.F64, 0, keep=t)])
    k = torch.empty(rows, dtype=torch.int64, device=dev)
    lib.fill_uniform_i64(k.data_ptr(), rows, SEED, 3, first, -2 ** 31, 2 ** 31)
    cols.append([A.DeviceArray(k.data_ptr(), None, 0, rows, A.I64, 0, keep=k)])
    e = A.Expr()
    fma = e.op("add", e.op("multiply", e.col(0), e.col(1)), e.col(2))
    roots = [fma, e.col(3)]

    def step():
        y, kk = sharding.all_combine(api.pipeline(e, cols, roots), device=comm_dev)
        return {"min_y": y.min, "max_y": y.max, "count_y": y.count, "min_k": kk.min, "max_k": kk.max}
    return step, 32.0 * rows, f"C3: fused a*b+c -> min/max/count + i64 key min/max/count over {rows:.0e} rows x 4 columns per GPU"


def workload_c4(torch, lib, api, A, sharding, dev, comm_dev, rank, rows, ngroups=1_000_000):
    """C4: SELECT key, sum(val) GROUP BY key, 1e6 keys: local hash aggregate, then (N > 1) the all-to-all of partial groups."""
    import numpy as np
    first = rank * rows
    kk = torch.empty(rows, dtype=torch.int64, device=dev)
    lib.fill_uniform_i64(kk.data_ptr(), rows, SEED, 7, first, 0, ngroups)
    v = torch.empty(rows, dtype=torch.float64, device=dev)
    lib.fill_uniform_f64(v.data_ptr(), rows, SEED, 0, first, 0.0, 1.0)
    K = A.DeviceArray(kk.data_ptr(), None, 0, rows, A.I64, 0, keep=kk)
    V = A.DeviceArray(v.data_ptr(), None, 0, rows, A.F64, 0, keep=v)
    cap = ngroups + 2
    bufs = [torch.empty(cap * 8 + 64, dtype=torch.uint8, device=dev) for _ in range(3)]
    outs = tuple(A.DeviceArray(b.data_ptr(), None, 0, cap, dt, 0, keep=b) for b, dt in zip(bufs, (A.I64, A.F64, A.I64)))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    def step():
        gk, gs, gc = api.groupby_sum([K], [V], ngroups, outs)
        ng = gk.length
        if world == 1:
            return {"groups": ng}
        # partial groups of this rank -> owners (RCCL all-to-all), merged there by a second, tiny group-by
        hk = torch.empty(ng, dtype=torch.int64); hs = torch.empty(ng, dtype=torch.float64); hc = torch.empty(ng, dtype=torch.int64)
        lib.load().rdf_copy_d2h(hk.data_ptr(), gk.values_ptr, ng * 8)
        lib.load().rdf_copy_d2h(hs.data_ptr(), gs.values_ptr, ng * 8)
        lib.load().rdf_copy_d2h(hc.data_ptr(), gc.values_ptr, ng * 8)
        rk, rs, rc = sharding.exchange_groups(hk.numpy(), hs.numpy(), hc.numpy(), comm_dev)
        Kh = [A.HostArray.from_numpy(rk)]
        mk, ms, _ = api.groupby_sum(Kh, [A.HostArray.from_numpy(rs)], ngroups)
        return {"groups_owned": mk.length}
    return step, 16.0 * rows, f"C4: hash GROUP BY key -> sum(val), {ngroups:.0e} keys over {rows:.0e} rows per GPU"


def workload_q1(torch, lib, api, A, sharding, dev, comm_dev, rank, rows):
    """C5: TPC-H Q1 shape over a synthetic lineitem shard (38 B/row): filter(shipdate <= c) -> 5 sums + counts in 6 groups."""
    g = torch.Generator(device=dev)
    g.manual_seed(SEED + rank)
    qty = torch.randint(1, 51, (rows,), device=dev, generator=g).to(torch.float64)
    price = torch.empty(rows, dtype=torch.float64, device=dev)
    lib.fill_uniform_f64(price.data_ptr(), rows, SEED, 11, rank * rows, 900.0, 105000.0)
    disc = torch.randint(0, 11, (rows,), device=dev, generator=g).to(torch.float64) / 100.0
    tax = torch.randint(0, 9, (rows,), device=dev, generator=g).to(torch.float64) / 100.0
    flag = torch.randint(0, 3, (rows,), dtype=torch.int8, device=dev, generator=g)
    status = torch.randint(0, 2, (rows,), dtype=torch.int8, device=dev, generator=g)
    ship = torch.randint(8036, 10562, (rows,), dtype=torch.int32, device=dev, generator=g)
    cols = [[A.DeviceArray(t.data_ptr(), None, 0, rows, dt, 0, keep=t)] for t, dt in
            ((qty, A.F64), (price, A.F64), (disc, A.F64), (tax, A.F64), (flag, A.I8), (status, A.I8), (ship, A.I32))]
    q = A.Expr()
    c = [q.col(i) for i in range(7)]
    pred = q.op("le", c[6], q.scalar(10471, A.I32))
    gid = q.op("add", q.op("multiply", q.cast(c[4], A.I32), q.scalar(2, A.I32)), q.cast(c[5], A.I32))
    dp = q.op("multiply", c[1], q.op("subtract", q.scalar(1.0), c[2]))
    ch = q.op("multiply", dp, q.op("add", q.scalar(1.0), c[3]))
    vals = [c[0], c[1], dp, ch, c[2]]

    def step():
        res, nrows = sharding.all_combine_groups(api.group_pipeline(q, cols, vals, gid, 6, pred), device=comm_dev)
        return {"count_star": nrows[:6], "sum_qty": [r[0] for r in res[0][:6]]}
    return step, 38.0 * rows, f"C5: TPC-H Q1 shape (filter -> 5 sums + counts in 6 groups) over a {rows:.0e}-row synthetic lineitem shard per GPU"


WORKLOADS = {"c3": workload_c3, "c4": workload_c4, "q1": workload_q1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="rows per GPU (f64, 8 B/row)")
    ap.add_argument("--cpu-sample", type=int, default=200_000_000, help="rows for the CPU baseline leg (0 = skip)")
    ap.add_argument("--null-fraction", type=float, default=0.0, help="attach a validity bitmap with this null rate")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for a smoke test)")
    ap.add_argument("--share-gpu", action="store_true", help="smoke test only: every rank uses device 0 (needs --backend gloo)")
    ap.add_argument("--workload", default="headline", choices=["headline"] + sorted(WORKLOADS),
                    help="headline = BASELINE.json's metric (the default, what the driver runs); c3 / c4 / q1 = the other configs")
    args = ap.parse_args()

    import torch
    from rust_dataframe_amd import _abi as A
    from rust_dataframe_amd import lib, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(local_rank)
    lib.set_device(local_rank)
    api = lib.api()
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if (world == 1 or args.backend == "nccl") else None   # gloo exchanges CPU tensors

    rows = args.rows
    if args.workload != "headline":
        return run_other(args, torch, lib, api, A, sharding, dev, comm_dev, dist, rank, world)
    first_row = rank * rows
    x = torch.empty(rows, dtype=torch.float64, device=dev)
    lib.fill_uniform_f64(x.data_ptr(), rows, SEED, 0, first_row, 0.0, 1.0)
    vptr, vkeep = None, None
    if args.null_fraction > 0:
        vkeep = torch.zeros((rows + 63) // 64 * 8 + 64, dtype=torch.uint8, device=dev)
        lib.fill_validity(vkeep.data_ptr(), rows, SEED, 0, first_row, args.null_fraction)
        vptr = vkeep.data_ptr()
    col = A.DeviceArray(x.data_ptr(), vptr, 0, rows, A.F64, -1, keep=(x, vkeep))

    e = A.Expr()
    c = e.col(0)
    pred = e.op("gt", c, e.scalar(THRESHOLD))
    def step():
        local = api.pipeline(e, [[col]], [c], pred)          # fused filter -> {sum,min,max,count}, one pass over HBM
        tot = sharding.all_combine(local, device=comm_dev)[0]      # N > 1: all_gather the partials (RCCL), fold in rank order
        return tot.sum, tot.count

    def sync():
        torch.cuda.synchronize()
        lib.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    lib.kernel_timing_reset(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = lib.kernel_timing_get()
    kernel_name = lib.last_kernel()
    lib.kernel_timing_reset(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev if comm_dev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        total_rows = rows * world
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rows * args.steps / elapsed
        alg_bytes = rows * 8.0 + (rows / 8.0 if vptr else 0.0)   # per launch (one rank's launch)
        avg_kernel_s = kern_ms / max(kern_n, 1) * 1e-3
        achieved = alg_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):   # PMC-derived HBM bytes per launch, measured in a separate rocprofv3 --pmc pass
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if int(tj.get("rows", -1)) == rows and bool(tj.get("validity", False)) == bool(vptr):
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "rows/sec filter+sum over 1e9 f64 Arrow rows; %HBM bw at 1/2/4/8 GPU",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"filter(x>{THRESHOLD})->sum over a {rows:.0e}-row f64 Arrow RecordBatch per GPU, HBM-resident"
                                   + (f", {args.null_fraction:.0%} nulls (validity bitmap)" if vptr else ", no validity bitmap"),
                       "rows_per_gpu": rows, "total_rows": total_rows, "selectivity": res[1] / total_rows,
                       "result_sum": res[0], "result_count": res[1], "sharding": "row ranges per rank, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_name, "avg_kernel_ms": avg_kernel_s * 1e3, "launches": kern_n,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and args.cpu_sample > 0:
            sample = min(args.cpu_sample, rows)
            cb = cpu_baseline(sample)
            # not timed: the device result on the same sample prefix agrees with the oracle
            sub = A.DeviceArray(x.data_ptr(), vptr, 0, sample, A.F64, -1)
            if not vptr:
                g = api.pipeline(e, [[sub]], [c], pred)[0]
                ok = g.count == cb["_count"] and abs(g.sum - cb["_sum"]) <= 1e-6 * abs(cb["_sum"])
                cb["parity_on_sample"] = bool(ok)
            cb.pop("_sum"), cb.pop("_count")
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_other(args, torch, lib, api, A, sharding, dev, comm_dev, dist, rank, world):
    """Same timing contract as the headline, for the other BASELINE.json configurations."""
    rows = args.rows
    step, alg_bytes, desc = WORKLOADS[args.workload](torch, lib, api, A, sharding, dev, comm_dev, rank, rows)

    def sync():
        torch.cuda.synchronize()
        lib.synchronize()
    sync()   # the synthetic columns were written on torch's stream; the library launches on its own
    for _ in range(args.warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    lib.kernel_timing_reset(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = lib.kernel_timing_get()
    kernel_name = lib.last_kernel()
    lib.kernel_timing_reset(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev if comm_dev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        dist.destroy_process_group()
    if rank == 0:
        per_launch_s = kern_ms / max(kern_n, 1) * 1e-3 * (kern_n / args.steps if kern_n else 0)   # all timed kernels of one step
        achieved = alg_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        print(json.dumps({
            "metric": f"rows/sec {args.workload}", "value": rows * world * args.steps / elapsed, "unit": "rows/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "rows_per_gpu": rows, "total_rows": rows * world, "result": res},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": kernel_name, "kernel_ms_per_step": per_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }), flush=True)


if __name__ == "__main__":
    main()
