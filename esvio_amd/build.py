"""Build libesvio_fe.so (HIP kernels + host orchestration + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs on the CPU-only build container; the built
.so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libesvio_fe.so")
SOURCES = ["fe_kernels.hip", "fe_api.cpp", "fe_host.cpp"]
HEADERS = ["fe_kernels.h", "fe_host.h", os.path.join("..", "..", "include", "esvio_fe.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # keep mul+add separate: bit parity with the x86 reference build
    "-fno-fast-math",
    "-Wall", "-Wno-unused-function",
    "-Wl,-rpath,/opt/rocm/lib",
]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-x", "hip"] + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
