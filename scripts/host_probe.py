import sys, time, torch
sys.path.insert(0, '/root/repo')
from epipolar_transformers_amd import camera, ops, synthetic as syn
torch.set_num_threads(8)
P1, P2 = syn.make_pairs(32, 4, 256, seed=1000, jitter=(0.05, 8.0))
P1p, P2p = P1.pin_memory(), P2.pin_memory()
dev = torch.device('cuda:0')
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("pair_algebra (CPU)            %.3f ms" % t(lambda: camera.pair_algebra(P1p, P2p)))
c = camera.pair_algebra(P1p, P2p)
print("pin_memory                    %.3f ms" % t(lambda: c.pin_memory()))
cp = c.pin_memory()
print("to(dev, non_blocking)         %.3f ms" % t(lambda: cp.to(dev, non_blocking=True)))
print("whole cam path                %.3f ms" % t(lambda: camera.pair_algebra(P1p, P2p).pin_memory().to(dev, non_blocking=True)))
g = torch.Generator(device=dev).manual_seed(0)
ref = torch.randn(128, 64, 64, 256, device=dev, generator=g).relu_(); src = torch.randn(128, 64, 64, 256, device=dev, generator=g).relu_()
spec = ops.LayerSpec(H=64, W=64, K=64)
cam = cp.to(dev)
wf = torch.randn(256, 256, device=dev) * 0.05 + torch.eye(256, device=dev); bias = torch.randn(256, device=dev)
packed = ops.residual_gemm_pack(wf)
def gpu_only():
    out, attn, corr = ops.forward_nhwc(spec, ref, src, cam)
    return ops.residual_gemm(out, packed, bias, ref)
print("GPU part of the step (no cam) %.3f ms" % t(gpu_only, 30))
def cpu_launch_only():
    t0 = time.perf_counter(); gpu_only(); return time.perf_counter() - t0
torch.cuda.synchronize(); xs = [cpu_launch_only() for _ in range(20)]; torch.cuda.synchronize()
print("CPU time to enqueue GPU part  %.3f ms (median)" % (sorted(xs)[10] * 1e3))
def full():
    cam = camera.pair_algebra(P1p, P2p).pin_memory().to(dev, non_blocking=True)
    out, attn, corr = ops.forward_nhwc(spec, ref, src, cam)
    return ops.residual_gemm(out, packed, bias, ref)
print("full step                     %.3f ms" % t(full, 30))
