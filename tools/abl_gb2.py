"""Ablations of the second-generation GROUP BY scatter (results invalid by construction): where a tile's time goes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rust_dataframe_amd import _abi as A, lib
lib.set_device(0); api = lib.api()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
ng = 1_000_000
k = torch.empty(n, dtype=torch.int64, device="cuda"); v = torch.empty(n, dtype=torch.float64, device="cuda")
lib.fill_uniform_i64(k.data_ptr(), n, 42, 7, 0, 0, ng); lib.fill_uniform_f64(v.data_ptr(), n, 42, 0, 0, 0.0, 1.0)
K = A.DeviceArray(k.data_ptr(), None, 0, n, A.I64, 0); V = A.DeviceArray(v.data_ptr(), None, 0, n, A.F64, 0)
bufs = [torch.empty((ng + 2) * 8 + 64, dtype=torch.uint8, device="cuda") for _ in range(3)]
outs = tuple(A.DeviceArray(b.data_ptr(), None, 0, ng + 2, dt, 0) for b, dt in zip(bufs, (A.I64, A.F64, A.I64)))
torch.cuda.synchronize()
for compact in (2, 0):
  lib.set_option("gb_compact", compact)
  for dbg, what in ((0, "full"), (21, "no global stores"), (22, "no flush phase (E)"), (23, "no staging, no flush (D, E)"), (24, "loads + hash only")):
    lib.set_option("gb_debug", dbg)
    for it in range(3):
        if it == 1:
            lib.kernel_timing_reset(True)
        try:
            api.groupby_sum([K], [V], ng, outs)
        except Exception as ex:
            pass
    ms, cnt = lib.kernel_timing_get(); lib.kernel_timing_reset(False)
    print(json.dumps({"records": "12-byte" if compact else "16-byte", "ablation": dbg, "what": what, "scatter+aggregate_ms": ms / max(cnt, 1)}), flush=True)
lib.set_option("gb_debug", 0)
lib.set_option("gb_compact", 1)
