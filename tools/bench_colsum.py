import torch, sys
sys.path.insert(0, "/root/repo")
import htrvt_amd
from htrvt_amd._lib import lib, check
from htrvt_amd.ops import colsum, dt
for rows, cols in ((32768, 768), (32768, 3072), (32768, 2304)):
    x = torch.randn(rows, cols, device="cuda").bfloat16()
    out = torch.zeros(cols, device="cuda")
    for _ in range(3):
        colsum(x, rows, cols, cols, out, dti=dt(torch.bfloat16))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        colsum(x, rows, cols, cols, out, dti=dt(torch.bfloat16))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(rows, cols, f"{ms*1e3:.1f} us  {rows*cols*2/ms/1e9:.2f} TB/s")
