"""Build libmobgs_hip.so (gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
# MOBGS_LIB: load another build of the same library instead (A/B timing of kernel variants on one GPU box)
LIB_PATH = Path(os.environ["MOBGS_LIB"]).resolve() if os.environ.get("MOBGS_LIB") else CSRC / "libmobgs_hip.so"
SOURCES = ["project.hip", "isect.hip", "raster.hip", "raster_bwd_mfma.hip", "raster_layers.hip", "pipeline.hip", "prep.hip", "decoder.hip", "deform.hip", "deform_bwd.hip", "hexplane_bwd.hip", "blce.hip", "loss.hip", "flowloss.hip", "densify.hip", "normals.hip"]
ARCH = "gfx950"
# host fast path (csrc/fastpath.cpp): a plain C++ torch extension, no device code, no link against libmobgs_hip.so
FAST_SRC = CSRC / "fastpath.cpp"
FAST_PATH = CSRC.parent / "_mobgs_fast.so"
# Per-file extra flags.  -fno-slp-vectorize: the SLP vectoriser pairs independent fp32 FMAs into v_pk_fma_f32,
# which on gfx950 issues at half rate (no gain) and needs v_mov shuffles to build the 64-bit operand pairs:
# +6% renders/s on the compositing kernels without it (measured).
NOSLP_FILES = os.environ.get("MOBGS_NOSLP_FILES", "raster.hip,raster_bwd_mfma.hip,raster_layers.hip").split(",")
EXTRA_FLAGS = {f: ["-fno-slp-vectorize"] for f in NOSLP_FILES if f}
# project.hip: no implicit FMA contraction.  radii (= ceil(3 sqrt(lambda_max))), tile rectangles and the cull tests
# are integer / boolean functions of float expressions; with the compiler free to fuse a * b + c differently from the
# written order, one splat in ~8000 of an extreme scene (near-plane, 6x scales) landed on the other side of an integer
# boundary than the oracles, which evaluate the expressions as written (scripts/soak_parity.py).  The kernels are
# HBM-bound: no measurable cost.
EXTRA_FLAGS.setdefault("project.hip", [])
EXTRA_FLAGS["project.hip"] = EXTRA_FLAGS["project.hip"] + ["-ffp-contract=off"]
# raster.hip: top-down pre-RA machine scheduling.  The compositing loops are long straight-line blocks bound by VALU
# issue; of ten scheduler settings swept in round 4 (scripts/ab/build_variant_raster.sh + kernel_ab2.sh, three A/B
# repetitions on one box) this is the only one outside the noise: raster_bwd<10> 512 -> 506 us, raster_fwd_blocks<10>
# 219 -> 213.5 us, <12> forward -2 %, <16> backward -1 % (raster_fwd<16> +4 %).  Same instructions in another order:
# results are bit-identical.
EXTRA_FLAGS["raster.hip"] = EXTRA_FLAGS.get("raster.hip", []) + ["-mllvm", "-misched-prera-direction=topdown"]
for _f in ("raster.hip", "raster_bwd_mfma.hip", "raster_layers.hip"):  # experiment hook: extra flags for the compositing kernels
    EXTRA_FLAGS.setdefault(_f, [])
    EXTRA_FLAGS[_f] = EXTRA_FLAGS[_f] + os.environ.get("MOBGS_RASTER_EXTRA_FLAGS", "").split()


class _BuildLock:
    """One builder at a time per output file (several ranks of one node start together and all find the same stale
    .so): an exclusive flock on <target>.lock; the artefact itself is written to a temporary name and moved into place
    with os.replace, so a concurrent loader sees the old file or the new one, never a half-written one."""

    def __init__(self, target: Path):
        self.path = str(target) + ".lock"
        self.fd = None

    def __enter__(self):
        import fcntl
        self.fd = os.open(self.path, os.O_CREAT | os.O_RDWR, 0o644)
        fcntl.flock(self.fd, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.fd, fcntl.LOCK_UN)
        os.close(self.fd)
        return False


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libmobgs_hip.so")


def sources() -> list[Path]:
    return [CSRC / s for s in SOURCES if (CSRC / s).exists()]


def is_stale() -> bool:
    if os.environ.get("MOBGS_LIB"):
        return False
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = sources() + [CSRC / "common.h", CSRC / "raster_shared.h", CSRC / "decoder_shared.h", CSRC / "prep_shared.h", CSRC / "hexplane.h", CSRC.parent.parent / "include" / "mobgs_hip.h"]
    return any(d.exists() and d.stat().st_mtime > t for d in deps)


def build_extension(force: bool = False, verbose: bool = False) -> Path:
    if not force and not is_stale():
        return LIB_PATH
    with _BuildLock(LIB_PATH):
        if not force and not is_stale():  # another process built it while this one waited for the lock
            return LIB_PATH
        return _build_extension_locked(force, verbose)


def _build_extension_locked(force: bool, verbose: bool) -> Path:
    objs = []
    hipcc = _hipcc()
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
    procs = []
    for src in sources():
        obj = src.with_suffix(".o")
        objs.append(obj)
        if not force and obj.exists() and obj.stat().st_mtime > max(
                src.stat().st_mtime, (CSRC / "common.h").stat().st_mtime, (CSRC / "hexplane.h").stat().st_mtime,
                (CSRC / "raster_shared.h").stat().st_mtime, (CSRC / "decoder_shared.h").stat().st_mtime, (CSRC / "prep_shared.h").stat().st_mtime,
                (CSRC.parent.parent / "include" / "mobgs_hip.h").stat().st_mtime):
            continue
        cmd = [hipcc, *flags, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    tmp = LIB_PATH.with_name(LIB_PATH.name + f".tmp{os.getpid()}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *map(str, objs), "-o", str(tmp)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


def fastpath_is_stale() -> bool:
    if not FAST_PATH.exists():
        return True
    t = FAST_PATH.stat().st_mtime
    deps = [FAST_SRC, CSRC.parent.parent / "include" / "mobgs_hip.h"]
    return any(d.exists() and d.stat().st_mtime > t for d in deps)


def build_fastpath(force: bool = False, verbose: bool = False) -> Path:
    """g++ -shared against the installed libtorch (~30 s).  Needs no GPU and no hipcc."""
    if not force and not fastpath_is_stale():
        return FAST_PATH
    with _BuildLock(FAST_PATH):
        if not force and not fastpath_is_stale():
            return FAST_PATH
        return _build_fastpath_locked(verbose)


def _build_fastpath_locked(verbose: bool) -> Path:
    import sysconfig

    import torch
    import torch.utils.cpp_extension as ce
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no C++ compiler found: cannot build the host fast path")
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    tmp = FAST_PATH.with_name(FAST_PATH.name + f".tmp{os.getpid()}")
    cmd = [cxx, "-O2", "-std=c++17", "-shared", "-fPIC", str(FAST_SRC), "-o", str(tmp),
           "-DTORCH_EXTENSION_NAME=_mobgs_fast", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I" + sysconfig.get_paths()["include"], *["-I" + i for i in ce.include_paths()],
           "-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-Wl,-rpath," + libdir,
           "-Wno-unused-function"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"host fast path: compile failed:\n{r.stdout[-4000:]}")
    os.replace(tmp, FAST_PATH)
    return FAST_PATH


if __name__ == "__main__":
    print(build_extension(force="--force" in os.sys.argv, verbose=True))
