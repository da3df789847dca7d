"""Time projection -> tile lists alone (the single-pass path of render()) on the benchmark cloud; used with
scripts/prof.sh and MOBGS_LIB variants (scripts/ab/build_isect_variant.sh) to see what bounds each list kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgs_amd.synth import SynthCamera, splat_inputs
from mobgs_amd import rendering as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
dev = torch.device("cuda:0")
cam = SynthCamera()
s = {k: v.to(dev) for k, v in splat_inputs(n, cam, 0, 9).items()}
if os.environ.get("MOBGS_PROBE_SPATIAL"):  # experiment: splats physically re-ordered along a Morton curve of their 3-D position
    m = s["means"]
    q = ((m - m.min(0).values) / (m.max(0).values - m.min(0).values + 1e-9) * 1023).long().clamp(0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    perm = torch.argsort(code)
    for k in ("means", "quats", "scales", "opacities", "colors"):
        s[k] = s[k][perm].contiguous()
hint = int(os.environ.get("MOBGS_PROBE_HINT", "0"))  # ablated builds leave no valid counts behind: force the segment size
with torch.no_grad():
    for _ in range(31):
        if hint:
            R._len_hint[R._workload_key(dev, 1, n, cam.width, cam.height)] = hint
        sp = R.SharedProjection(s["means"], s["quats"], s["scales"], s["opacities"], s["viewmats"], s["Ks"], cam.width,
                                cam.height, pack_colors=s["colors"])
        n_isects = sp.tl.n_isects
torch.cuda.synchronize()
print("I", n_isects, "longest", sp.tl.max_tile_len, "fused calls", R.fused_calls[0], "rebuilds", R.list_rebuilds[0])


