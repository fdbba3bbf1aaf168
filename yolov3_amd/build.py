"""Build libyolov3_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the library is a
plain C-ABI shared object (include/yolov3_hip.h); torch only supplies device memory and streams.

    python -m yolov3_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB = LIB_DIR / "libyolov3_hip.so"
OBJ_DIR = PKG / "build"

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# (source, extra flags).  detect_nms must round like the reference's CPU code: no FMA contraction.
SOURCES = [
    ("conv.hip", []),
    ("stem.hip", []),
    ("layout_pool.hip", []),
    ("detect_nms.hip", ["-ffp-contract=off"]),
    ("val_edge.hip", ["-ffp-contract=off"]),
    ("loss.hip", ["-ffp-contract=off"]),
    ("train.hip", []),
    ("optim.hip", ["-ffp-contract=off"]),
    ("api.cpp", []),
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libyolov3_hip.so)")


def _newer(a: Path, b: Path) -> bool:
    return a.exists() and b.exists() and a.stat().st_mtime >= b.stat().st_mtime


def build(force: bool = False, verbose: bool = True) -> Path:
    cc = hipcc()
    OBJ_DIR.mkdir(exist_ok=True)
    LIB_DIR.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "yolov3_hip.h"]
    hdr_time = max(h.stat().st_mtime for h in headers)
    jobs = []
    for src, extra in SOURCES:
        s = CSRC / src
        if not s.exists():
            continue
        o = OBJ_DIR / (s.stem + ".o")
        if force or not _newer(o, s) or o.stat().st_mtime < hdr_time:
            lang = ["-x", "hip"] if s.suffix == ".hip" else []
            jobs.append((o, [cc, *COMMON, *extra, *lang, "-c", str(s), "-o", str(o)]))
        else:
            jobs.append((o, None))

    def run(job):
        o, cmd = job
        if cmd is None:
            return o
        if verbose:
            print("[yolov3_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
        objs = list(ex.map(run, jobs))
    if force or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
        if verbose:
            print("[yolov3_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
