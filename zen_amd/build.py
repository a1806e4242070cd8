"""Builds zen_amd/libzen_hip.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract: every float
product and sum is rounded separately, as in oracle/zen_oracle.c.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libzen_hip.so")
OBJDIR = os.path.join(HERE, "build")
SOURCES = ["api.hip", "hpr.hip", "stft.hip", "median.hip", "median_net.hip", "box.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "zen_hip.h"))
    return hdrs


def _compile(src):
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    newest = max(os.path.getmtime(p) for p in [srcp] + _deps())
    if os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", srcp, "-o", obj]
    subprocess.check_call(cmd)
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [r[0] for r in res]
    if any(r[1] for r in res) or not os.path.exists(OUT):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
        if verbose:
            print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
