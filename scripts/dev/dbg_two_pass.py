import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from epipolar_transformers_amd import _lib, camera, ops, synthetic as syn
from oracle import oracle as orc
orc.build()
C=256; h,k=64,128
P1,P2=syn.rig_pairs("ring",1,4*h,seed=55,jitter=(0.05,8.0)); P1,P2=P1[:2],P2[:2]
f1,f2=syn.make_features(2,C,h,h,seed=56)
f2[0,100,20,10]=4.0e4; f2[1,7,40,50]=-6.0e4
cam=camera.pair_algebra(P1,P2)
want=orc.forward(orc.LayerSpec(h,h,k),f1,f2,None,None,cam=cam.numpy())
for variant,name in ((0,"two-pass"),(65536,"classic"),(16384,"per-pixel")):
    spec=ops.LayerSpec(H=h,W=h,K=k,variant=variant)
    ws=ops.tile_workspace(spec,2,C,"cuda") if variant!=16384 else None
    out,attn,corr=ops.forward_nhwc(spec,ops.to_nhwc(f1.cuda()),ops.to_nhwc(f2.cuda()),cam.cuda(),workspace=ws)
    o=out.permute(0,3,1,2).cpu().numpy()
    err=np.abs(o-want["out"]); idx=np.unravel_index(err.argmax(),err.shape)
    print(name,"max err",err.max(),"at",idx,"want",want["out"][idx],"got",o[idx],"attn err",np.abs(attn.cpu().numpy()-want["attn"]).max(),
          "rel-adjusted",(err-1e-6*np.abs(want["out"])).max())
    if ws is not None and ws.numel():
        base=(-ws.data_ptr())%256; print("  overflow tiles",int(ws[base:base+4].view(torch.int32).item()))
