"""Development: time et_z_wgrad at Config 2 (524 288 rows) with HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from epipolar_transformers_amd import ops
rows = 128 * 64 * 64
g = torch.Generator(device="cuda").manual_seed(0)
dy = torch.randn(rows, 256, device="cuda", generator=g)
out = torch.randn(rows, 256, device="cuda", generator=g).relu_()
for _ in range(3):
    ops.z_wgrad(dy, out)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
torch.cuda.synchronize()
for a, b in ev:
    a.record(); ops.z_wgrad(dy, out); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) for a, b in ev)
gw, gb = ops.z_wgrad(dy, out)
want = dy.double().t() @ out.double()
print("%s: et_z_wgrad %.4f ms (min %.4f)  max rel err vs float64 %.2e" % (sys.argv[1] if len(sys.argv) > 1 else "default", sum(t) / len(t), t[0],
      ((gw.double() - want).abs().max() / want.abs().max()).item()))
