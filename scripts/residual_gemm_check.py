import sys, time, torch
sys.path.insert(0, '/root/repo')
from epipolar_transformers_amd import ops
torch.manual_seed(0)
dev = 'cuda'
for rows in (128 * 64 * 64, 1000, 64, 77):
    out = torch.randn(rows, 256, device=dev).relu_() * 2.5
    out[3] *= 1e-6; out[5] *= 3e4
    if rows > 70: out[70] = 0
    feat = torch.randn(rows, 256, device=dev)
    wf = torch.randn(256, 256, device=dev) * 0.05 + torch.eye(256, device=dev)
    bias = torch.randn(256, device=dev)
    packed = ops.residual_gemm_pack(wf)
    x = ops.residual_gemm(out, packed, bias, feat)
    x2 = ops.residual_gemm(out, packed, bias)
    ref64 = (out.double() @ wf.double().t() + bias.double())
    w32 = torch.addmm(bias, out, wf.t())
    e_new = (x2.double() - ref64).abs()
    e_32 = (w32.double() - ref64).abs()
    scale = ref64.abs().amax(1, keepdim=True).clamp_min(1e-30)
    print("rows %7d: split-fp16 kernel max abs err %.3e (rel to row max %.3e) | fp32 addmm %.3e (%.3e) | with feat err %.3e" % (
        rows, e_new.max(), (e_new / scale).max(), e_32.max(), (e_32 / scale).max(), (x.double() - ref64 - feat.double()).abs().max()))
rows = 128 * 64 * 64
out = torch.randn(rows, 256, device=dev).relu_(); feat = torch.randn(rows, 256, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
base = feat.clone()
wt = wf.t().contiguous()
print("residual_gemm (feat)   %.3f ms" % t(lambda: ops.residual_gemm(out, packed, bias, feat)))
print("residual_gemm (no feat) %.3f ms" % t(lambda: ops.residual_gemm(out, packed, bias)))
print("torch.addmm out=base    %.3f ms" % t(lambda: torch.addmm(base, out, wt, out=base)))
print("pack                    %.3f ms" % t(lambda: ops.residual_gemm_pack(wf)))
