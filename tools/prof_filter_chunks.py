"""One compaction call on 1e9 f64 rows held in chunks of argv[1] rows (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rust_dataframe_amd import _abi as A, lib
lib.set_device(0); api = lib.api()
cr = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000_000
x = torch.empty(n, dtype=torch.float64, device="cuda"); lib.fill_uniform_f64(x.data_ptr(), n, 42, 0, 0, -1.0, 1.0)
mb = torch.zeros(n // 8 + 64, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
e = A.Expr(); gt = e.op("gt", e.col(0), e.scalar(0.0))
api.predicate(e, gt, [[A.DeviceArray(x.data_ptr(), None, 0, n, A.F64, 0)]], [A.DeviceArray(mb.data_ptr(), None, 0, n, A.BOOL, 0)])
XF = A.PreparedCol([A.DeviceArray(x.data_ptr() + i * 8, None, 0, min(cr, n - i), A.F64, 0) for i in range(0, n, cr)])
MF = A.PreparedCol([A.DeviceArray(mb.data_ptr() + i // 8, None, 0, min(cr, n - i), A.BOOL, 0) for i in range(0, n, cr)])
ofb = torch.empty(n, dtype=torch.float64, device="cuda")
OF = [A.DeviceArray(ofb.data_ptr() + i * 8, None, 0, min(cr, n - i), A.F64, 0) for i in range(0, n, cr)]
torch.cuda.synchronize()
for _ in range(2):
    api.filter(XF, MF, OF)
print("done", cr)
