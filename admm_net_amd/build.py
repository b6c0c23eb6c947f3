"""Build the HIP extension in-tree: hipcc --offload-arch=gfx950 -> libadmmnet_hip.so.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libadmmnet_hip.so")
SOURCES = ["api.hip", "prep.hip", "tridiag.hip", "tridiag_reg.hip", "tridiag_big.hip", "tridiag_panel.hip", "wy_apply.hip", "tql.hip", "rotapply.hip", "dc.hip", "rebuild.hip", "rebuild_big.hip", "backrebuild.hip", "vgemm_big.hip", "arrow.hip",
           "zstep.hip", "head.hip", "spectrum.hip", "peaks.hip", "synth.hip", "vdvh.hip", "spectral.hip", "spectral_fused.hip"]
HEADERS = ["common.h", "eig_core.h", "dc_core.h", "arrow_core.h", "rebuild_lds.h", "lane_reduce.h", os.path.join("..", "..", "include", "admmnet.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file extras: the SLP vectoriser packs the rotation replay into v_pk_* ops that need a
# register shuffle per plane (7 VALU per rotation instead of 4)
EXTRA_FLAGS = {"rotapply.hip": ["-fno-slp-vectorize"],
               # here the SLP pass re-packs the DPP reduction adds into v_mov_dpp + v_pk_add (3 instructions for 2)
               "tridiag_reg.hip": ["-fno-slp-vectorize"],
               "tridiag_panel.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
