"""Build libbp_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

Every .hip file is its own translation unit: they are compiled in parallel to objects under ``csrc/build/``
(only the stale ones) and linked into ``baryon_painter_amd/libbp_hip.so``."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))
HEADERS = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hpp")] + \
          [os.path.join(HERE, "..", "..", "include", "bp_hip.h")]
OBJ_DIR = os.path.join(HERE, "build")
OUT = os.path.join(os.path.dirname(HERE), "libbp_hip.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(OUT, [os.path.join(HERE, s) for s in SOURCES] + HEADERS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(HERE, s), os.path.join(OBJ_DIR, s[:-4] + ".o")
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HERE)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] +
        [os.path.join(OBJ_DIR, s[:-4] + ".o") for s in SOURCES])
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
