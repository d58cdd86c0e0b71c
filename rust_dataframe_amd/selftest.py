"""First-contact driver of the multi-GPU path on real peers (used by __graft_entry__.smoke() when more than one device is
visible, and by tests): one RCCL communicator over the devices (rdf_comm_init_all), one host thread per rank, a hash GROUP BY
whose partial groups cross xGMI (rdf_groupby_agg_dist), one distributed filter -> aggregate (rdf_pipeline_dist) with an EMPTY
shard among the ranks.  It only drives the library and hands the results (and the inputs) back: the caller checks them
(tests/multi_device_check.py).  Nothing here is on the product path."""
import ctypes as C
import threading

import numpy as np

from . import _abi as A


def _dev_array(L, ptrs, np_arr, dtype, capacity=None):
    p = C.c_void_p(0)
    if L.rdf_dev_alloc(C.byref(p), max(np_arr.nbytes, 8) + 1024) != 0:
        raise RuntimeError("rdf_dev_alloc failed")
    ptrs.append(p)
    if np_arr.nbytes and L.rdf_copy_h2d(p, np_arr.ctypes.data, np_arr.nbytes) != 0:
        raise RuntimeError("rdf_copy_h2d failed")
    return A.DeviceArray(p.value, None, 0, len(np_arr), dtype, 0, capacity=capacity if capacity is not None else len(np_arr))


def multi_device_groupby(lib, api, devices, rows=1_000_000, ngroups=50_000, seed=11, kind=None, timeout_s=300):
    """-> dict(keys, sums, counts: the union of the ranks' groups; per_rank_groups; pipeline: the distributed aggregate;
    inputs: (keys, values) as numpy for the caller's check).  The last rank holds NO rows (an empty shard must still join)."""
    world = len(devices)
    rng = np.random.default_rng(seed)
    keys = (rng.integers(0, ngroups, rows).astype(np.int64) * 1_000_003) - 7
    vals = rng.uniform(0.0, 1.0, rows)
    holders = max(1, world - 1) if world > 1 else 1
    cuts = [rows * r // holders for r in range(holders + 1)] + [rows] * (world - holders)
    comms = A.Comm.init_all(api, list(devices), A.COMM_RCCL if kind is None else kind)
    res, err = [None] * world, [None] * world
    e = A.Expr()
    pred = e.op("gt", e.col(1), e.scalar(0.5))

    def body(r):
        ptrs = []
        L = lib.load()
        try:
            lib.set_device(devices[r])
            a, b = cuts[r], cuts[r + 1]
            K = _dev_array(L, ptrs, np.ascontiguousarray(keys[a:b]), A.I64)
            V = _dev_array(L, ptrs, np.ascontiguousarray(vals[a:b]), A.F64)
            cap = ngroups + 8
            outs = tuple(_dev_array(L, ptrs, np.zeros(cap + 64, dtype=A.NP_OF[dt]), dt, capacity=cap) for dt in (A.I64, A.F64, A.I64))
            ok, ov, oc = comms[r].groupby_agg([K], [V], "sum", ngroups + 4, outs, "auto")
            ng = ok.length
            hk, hv, hc = np.empty(ng, np.int64), np.empty(ng, np.float64), np.empty(ng, np.int64)
            for h, d in ((hk, ok), (hv, ov), (hc, oc)):
                if ng and L.rdf_copy_d2h(h.ctypes.data, d.values_ptr, h.nbytes) != 0:
                    raise RuntimeError("rdf_copy_d2h failed")
            agg = comms[r].pipeline_dist(e, [[K], [V]], [e.col(1)], pred)[0]
            res[r] = (hk, hv, hc, (agg.sum, agg.count, agg.min, agg.max), dict(comms[r].stats))
        except BaseException as ex:  # noqa: BLE001
            err[r] = ex
        finally:
            for p in ptrs:
                L.rdf_dev_free(p)

    th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]      # (daemon: a rank stuck in a collective must not keep the process from ending)
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=timeout_s)
    stuck = [r for r, t in enumerate(th) if t.is_alive()]
    for c in comms:
        if not stuck:
            c.destroy()
    if stuck:
        raise RuntimeError(f"multi-device self-test: ranks {stuck} did not come back within {timeout_s} s")
    for ex in err:
        if ex is not None:
            raise ex
    return {"keys": np.concatenate([r[0] for r in res]), "sums": np.concatenate([r[1] for r in res]), "counts": np.concatenate([r[2] for r in res]),
            "per_rank_groups": [len(r[0]) for r in res], "pipeline": [r[3] for r in res], "stats": [r[4] for r in res],
            "inputs": (keys, vals), "rows_per_rank": [cuts[r + 1] - cuts[r] for r in range(world)]}
