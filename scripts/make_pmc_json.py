"""profiles/r02_raster_bwd_pmc.json from a scripts/prof_pmc.sh summary of bench.py (and optionally an SQ counter file):
    python scripts/make_pmc_json.py gpurun_out/<name>/pmc_summary.json [I]
The calibration block and the correction rule are the measured ones of DESIGN section 4b (profiles/r02/
r02a_fetch_calib_pmc_summary.json); the source hash ties the numbers to the kernel sources they were collected with
(bench.py ignores the file when the hash differs)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    summ = json.load(open(sys.argv[1]))
    I = int(sys.argv[2]) if len(sys.argv) > 2 else 1690528
    P = 1352 * 1014
    key = [k for k in summ if "raster_bwd_kernel<10, false>" in k or k.endswith("raster_bwd_kernel<10, false>")]
    key = key[0] if key else [k for k in summ if "raster_bwd_kernel" in k][0]
    fetch_kib, write_kib = summ[key]["FETCH_SIZE"], summ[key]["WRITE_SIZE"]
    old = json.load(open(os.path.join(ROOT, "profiles", "r02_raster_bwd_pmc.json")))
    stream = 52.0 * P  # the coalesced per-pixel stream: counted at 0.5 (16 B / lane streaming), everything else 1:1
    fetch, write = fetch_kib * 1024.0, write_kib * 1024.0
    upper = fetch + stream * 0.5 + write      # stream bytes under-counted by half: add the other half back
    lower = fetch + write
    alg = 132.0 * I + 52.0 * P
    old.update({
        "source_sha": bench.source_sha(),
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/prof_pmc.sh), bench.py "
                  f"--steps 10 --warmup 2; summary {os.path.relpath(sys.argv[1], ROOT)}; heaviest-list-first tile "
                  "schedule on",
        "FETCH_SIZE_KiB_per_launch_raw": fetch_kib, "WRITE_SIZE_KiB_per_launch_raw": write_kib,
        "hbm_bytes_per_launch": upper, "hbm_bytes_per_launch_lower_estimate": lower, "algorithmic_bytes": alg,
        "traffic_over_algorithmic": [round(lower / alg, 3), round(upper / alg, 3)],
    })
    json.dump(old, open(os.path.join(ROOT, "profiles", "r02_raster_bwd_pmc.json"), "w"), indent=1)
    print(json.dumps({k: old[k] for k in ("source_sha", "hbm_bytes_per_launch", "traffic_over_algorithmic")}))


if __name__ == "__main__":
    main()
