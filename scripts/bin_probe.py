"""Time the binning stage alone (projection -> lists) on the benchmark cloud; used with scripts/prof.sh and
MOBGS_LIB variants to see what bounds bin_kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgs_amd.synth import SynthCamera, splat_inputs
from mobgs_amd import rendering as R

dev = torch.device("cuda")
cam = SynthCamera()
s = {k: v.to(dev) for k, v in splat_inputs(300000, cam, 0, 9).items()}
with torch.no_grad():
    radii, m2d, depths, conics, tpg = R._Project.apply(s["means"], s["quats"], s["scales"], s["viewmats"], s["Ks"], cam.width, cam.height, 0.3, 0.01, 1e10, 0.0)
    for _ in range(20):
        tl = R.build_tile_lists(m2d, radii, depths, conics, s["opacities"], tpg, cam.width, cam.height)
torch.cuda.synchronize()
print("I", tl.n_isects)
