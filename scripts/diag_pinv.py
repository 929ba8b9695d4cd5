import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from epipolar_transformers_amd import synthetic as syn, camera
print("threads", torch.get_num_threads(), torch.__config__.show().split("\n")[2:6])
P1, _ = syn.make_pairs(32, 4, 256, seed=0, jitter=(0.05, 8.0))
t = time.time(); loop = torch.stack([p.pinverse() for p in P1]); print("loop ms", (time.time() - t) * 1e3)
emu = camera.batched_pinverse(P1)
print("emu == loop", torch.equal(emu, loop), (emu - loop).abs().max().item())
U, S, Vh = torch.linalg.svd(P1, full_matrices=False)
ok = True
for i in range(8):
    u, s, vh = torch.linalg.svd(P1[i], full_matrices=False)
    ok &= torch.equal(u, U[i]) and torch.equal(s, S[i]) and torch.equal(vh, Vh[i])
print("batched svd == single svd", ok)
A = (Vh.transpose(-1, -2) * (1.0 / S).unsqueeze(-2)); B = U.transpose(-1, -2)
single = torch.stack([a @ b for a, b in zip(A, B)])
print("manual single mm == loop", torch.equal(single, loop), (single - loop).abs().max().item())
print("bmm == loop", torch.equal(torch.bmm(A, B), loop))
N = A.shape[0]
big = A.reshape(N * 4, 3) @ B.permute(1, 0, 2).reshape(3, N * 3)
blk = torch.stack([big[4 * i:4 * i + 4, 3 * i:3 * i + 3] for i in range(N)])
print("block mm == loop", torch.equal(blk, loop), (blk - loop).abs().max().item())
# candidate chains
Ad, Bd = A.double(), B.double()
def chain(order, fma_first):
    acc = None
    for k in order:
        prod = Ad[..., :, k:k + 1] * Bd[..., k:k + 1, :]
        acc = prod.float() if acc is None else (prod + acc.double()).float()
    return acc
for order in ([0, 1, 2], [2, 1, 0], [1, 0, 2], [0, 2, 1]):
    c = chain(order, True)
    print("fma chain", order, torch.equal(c, loop), (c - loop).abs().max().item())
sep = ((A[..., :, 0:1] * B[..., 0:1, :]) + (A[..., :, 1:2] * B[..., 1:2, :])) + (A[..., :, 2:3] * B[..., 2:3, :])
print("separate mul/add", torch.equal(sep, loop))
dbl = (Ad @ Bd).float()
print("double matmul rounded", torch.equal(dbl, loop))
# how does pinverse itself do it: torch.linalg.pinv on single
print("linalg.pinv single == pinverse", all(torch.equal(torch.linalg.pinv(p), p.pinverse()) for p in P1[:8]))
print("linalg.pinv batched == loop", torch.equal(torch.linalg.pinv(P1), loop))
from concurrent.futures import ThreadPoolExecutor
torch.set_num_threads(1)
with ThreadPoolExecutor(16) as ex:
    for _ in range(2):
        t = time.time(); r = list(ex.map(lambda p: p.pinverse(), P1)); dt = time.time() - t
print("threadpool(16) loop ms", dt * 1e3)
t = time.time(); loop = [p.pinverse() for p in P1]; print("1-thread loop ms", (time.time() - t) * 1e3)
