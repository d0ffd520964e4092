"""Builds ``photoverse_amd/lib/libphotoverse_hip.so`` for gfx950 with hipcc (in-tree, no torch linkage).

``python -m photoverse_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libphotoverse_hip.so")
SOURCES = ["pv_gemm.hip", "pv_convbig.hip", "pv_norm.hip", "pv_attn.hip", "pv_misc.hip", "pv_xfused.hip", "pv_xq.hip", "pv_rowgemm.hip", "pv_backward.hip", "pv_train.hip", "pv_attnbwd.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
#: per-source extra flags.  The attention kernels keep MFMA results in VGPRs (the softmax VALU work reads them directly;
#: AGPR-form costs ~200 v_accvgpr moves per tile); the 256-row GEMM variant needs the AGPR half for its accumulators.
EXTRA_FLAGS = {"pv_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
               "pv_xfused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans",
                                 "-mllvm", "-pragma-unroll-threshold=4000000"],   # the C = 640 body must unroll fully: register arrays

               "pv_xq.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
               "pv_rowgemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "pv_train.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
               "pv_attnbwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"]}   # attention backward: 136 accvgpr moves per tile otherwise


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "pv_common.h"), os.path.join(CSRC, "pv_gemm_dev.h"), os.path.join(os.path.dirname(HERE), "include", "photoverse_hip.h"), __file__]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=4) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), res.stderr))
                if verbose:
                    print("[photoverse_amd.build] compiled", os.path.basename(cmd[-3]))
    if jobs or force or _stale(LIB, objs):
        res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n" + res.stderr)
        if verbose:
            print("[photoverse_amd.build] linked", LIB)
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
