import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgs_amd.synth import SynthCamera, splat_inputs
from mobgs_amd.rendering import rasterization
dev=torch.device('cuda')
cam=SynthCamera()
N=int(sys.argv[1]) if len(sys.argv)>1 else 300000
s={k:v.to(dev) for k,v in splat_inputs(N,cam,0,9).items()}
for k in ["means","quats","scales","opacities","colors","viewmats"]: s[k].requires_grad_(True)
bg=torch.zeros(1,9,device=dev)
g=torch.Generator().manual_seed(100)
v_img=torch.randn(1,cam.height,cam.width,10,generator=g).to(dev)
def step():
    img,a,meta=rasterization(s["means"],s["quats"],s["scales"],s["opacities"],s["colors"],s["viewmats"],s["Ks"],cam.width,cam.height,packed=False,backgrounds=bg,render_mode="RGB+ED")
    (img*v_img).sum().backward()
    return meta
for _ in range(3): meta=step()
torch.cuda.synchronize()
print("I =", meta["flatten_ids"].numel(), "visible", int((meta["radii"]>0).sum()))
t0=time.time()
K=20
for _ in range(K): step()
torch.cuda.synchronize()
dt=(time.time()-t0)/K
print(f"fwd+bwd {dt*1e3:.3f} ms  -> {1/dt:.1f} renders/s")
