"""Builds blub_amd/libblubhip.so (HIP kernels + C-ABI) for gfx950 with hipcc. In-tree, no JIT cache."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libblubhip.so")
SOURCES = [os.path.join(CSRC, "blub_fluid.hip"), os.path.join(CSRC, "scene_host.cpp"), os.path.join(CSRC, "scheduler_host.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, n) for n in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "blubhip.h")]
# -ffp-contract=off: element-wise kernels must round exactly like the (unfused) reference arithmetic / the oracle.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-gpu-rdc",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function"]


def source_hash():
    """sha256 (first 16 hex digits) over the kernel / host sources and the public header: stamps measured artefacts (the PMC captures under
    profiles/) with the code state they were taken from; bench.py reports whether the stamp matches the sources it runs."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(os.path.join(CSRC, n) for n in os.listdir(CSRC)) + [os.path.join(ROOT, "include", "blubhip.h")]:
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-x", "hip"] + SOURCES + ["-o", LIB, "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
