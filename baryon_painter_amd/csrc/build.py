"""Build libbp_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["conv_direct.hip", "conv_igemm.hip", "conv_small.hip", "conv_wgrad.hip", "conv_wgrad_tiles.hip", "conv_wgrad_small.hip", "pointwise.hip", "capi.hip"]
OUT = os.path.join(os.path.dirname(HERE), "libbp_hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(HERE, "common.hpp"),
                                                       os.path.join(HERE, "..", "..", "include", "bp_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-o", OUT] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
