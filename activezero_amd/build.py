"""Build libazhip.so (the C-ABI library of include/azhip.h) for gfx950.

    python -m activezero_amd.build [--force]

Every csrc/*.hip is compiled with hipcc --offload-arch=gfx950 into
activezero_amd/lib/obj/*.o and linked into activezero_amd/lib/libazhip.so
(in-tree, git-ignored, shipped to the GPU box by gpurun).  hipcc cross-compiles
without a GPU.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libazhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("AZ_HIPCC_EXTRA", "").split()  # experiments, e.g. -DX6_VARIANT=1


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    headers = glob.glob(os.path.join(SRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    if force or _stale(obj, [src] + headers):
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, jobs=None):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(SRC, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    jobs = jobs or min(6, os.cpu_count() or 1)
    with cf.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
