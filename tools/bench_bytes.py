#!/usr/bin/env python3
"""Int8 / UInt8 columns (they reach the path through casts, filters and aggregates: src/evaluation.rs:296-315, src/expression.rs:766-861):
the specialised kernels (16 rows per 16-byte vector) against the interpreter on the same programs, 2.5e8 rows."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402
from rust_dataframe_amd import _abi as A, lib
lib.set_device(0); api = lib.api()
n = 250_000_000
for dt, tdt in ((A.I8, torch.int8), (A.U8, torch.uint8)):
    t = torch.randint(0, 100, (n,), dtype=torch.int32, device="cuda").to(tdt)
    y = torch.rand(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    X = A.DeviceArray(t.data_ptr(), None, 0, n, dt, 0, keep=t); Y = A.DeviceArray(y.data_ptr(), None, 0, n, A.F64, 0, keep=y)
    e = A.Expr(); c0, c1 = e.col(0), e.col(1)
    progs = {"agg(x)": ([[X]], [c0], -1, 1.0), "filter(x>50)->agg(x)": ([[X]], [c0], e.op("gt", c0, e.scalar(50.0)), 1.0),
             "filter(x>50)->agg(y:f64)": ([[X], [Y]], [c1], e.op("gt", c0, e.scalar(50.0)), 9.0), "sum(cast(x->f64))": ([[X]], [e.cast(c0, A.F64)], -1, 1.0)}
    for name, (cols, vals, pred, bpr) in progs.items():
        for spec in (1, 0):
            lib.set_option("spec", spec)
            for _ in range(2): r = api.pipeline(e, cols, vals, pred)[0]
            lib.synchronize(); lib.kernel_timing_reset(True)
            for _ in range(5): r = api.pipeline(e, cols, vals, pred)[0]
            lib.synchronize(); ms, k = lib.kernel_timing_get(); lib.kernel_timing_reset(False)
            per = ms / 5
            print(json.dumps({"dtype": dt, "program": name, "spec": spec, "kernel": lib.last_kernel()[:60], "ms": round(per, 3), "frac_of_8TBps": round(bpr * n / (per * 1e-3) / 8e12, 3), "count": r.count, "sum": r.sum}))
    lib.set_option("spec", 1)
