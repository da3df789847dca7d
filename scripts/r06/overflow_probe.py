"""Which entry point survives an arena OVERFLOW without a host in the loop (rendering.StaticCapacity with a margin < 1: every
count-sized buffer is too small, the kernels must see empty lists and touch nothing out of bounds)?  One component per process:
python scripts/r06/overflow_probe.py {lean|train|many|flow|flow_many} [margin]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import bench as B
import mobgs_amd.rendering as R
from mobgs_amd.camera import PinholeCamera
from mobgs_amd.gaussian_renderer import get_flow, get_flow_many, render, render_many
torch.autograd.set_multithreading_enabled(False)
what = sys.argv[1]
margin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
FWD_ONLY = os.environ.get("FWD_ONLY") == "1"      # bisection: no backward pass
KEYS = os.environ.get("KEYS")                     # bisection: train mode -- which outputs enter the loss
dev = torch.device("cuda")
W, H = 512, 288
scam, cam, stat, dyn, raw = B.build_scene(dev, 20_000, 10_000, W, H)
bg = torch.zeros(9, device=dev)
params = B.leaves(stat, dyn)


def body():
    for p in params:
        p.grad = None
    if what == "lean":
        out = render(cam, stat, dyn, None, bg)
        (out["render"].sum() + out["depth"].sum()).backward()
    elif what == "train":
        out = render(cam, stat, dyn, None, bg, get_static=True, get_dynamic=True)
        keys = KEYS.split(",") if KEYS else ["render", "d_alpha", "s_render", "depth"]
        loss = sum(out[k].sum() for k in keys)
        if not FWD_ONLY:
            loss.backward()
    elif what == "many":
        cams = [PinholeCamera(W, H, scam.K, torch.eye(4), scam.time, scam.max_time, device=dev) for _ in range(8)]
        outs = render_many(cams, stat, dyn, None, bg, [torch.tensor(0.1 * k - 0.4, device=dev) for k in range(8)])
        loss = sum(o["render"].sum() + o["depth"].sum() for o in outs)
        if not FWD_ONLY:
            loss.backward()
    elif what == "flow":
        o = get_flow(cam, stat, dyn, None, bg, delta_exposure=0.5)
        sum(t.sum() for t in o).backward()
    elif what == "flow_many":
        outs = get_flow_many(cam, stat, dyn, None, bg, [0.25 * (k - 4) for k in range(9)])
        sum(t.sum() for o in outs for t in o).backward()
    torch.cuda.synchronize()


for _ in range(3):
    body()           # ordinary eager frames: counts and hints exist
print(what, "eager ok", flush=True)
st = R.StaticCapacity(margin)
with st:
    body()
    body()
print(what, f"static capacity margin {margin}: survived; arenas fitted: {st.check()}", flush=True)
