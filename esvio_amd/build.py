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
SOURCES = ["fe_kernels.hip", "fe_stages.cpp", "fe_track.cpp", "fe_image.cpp", "fe_api.cpp", "fe_host.cpp", "fe_evstage.cpp"]
HEADERS = ["fe_kernels.h", "fe_host.h", "fe_ctx.h", "fe_internal.h", "fe_mc.h", os.path.join("..", "..", "include", "esvio_fe.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # keep mul+add separate: bit parity with the x86 reference build
    "-fno-fast-math",
    "-fno-math-errno",            # sqrt() may be the instruction (same value), so lane loops vectorise
    "-Wall", "-Wno-unused-function",
    "-Wl,-rpath,/opt/rocm/lib",
    "-ldl",
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


TOOL_SRC = os.path.join(HERE, "..", "tools", "replay_node.cpp")
TOOL_BIN = os.path.join(HERE, "..", "tools", "replay_node")


def build_tools(force=False):
    """tools/replay_node: the C++ host harness (handle_stereo_event over the C ABI), plain g++"""
    src, out = os.path.abspath(TOOL_SRC), os.path.abspath(TOOL_BIN)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(
            os.path.getmtime(src), os.path.getmtime(LIB), os.path.getmtime(os.path.join(HERE, "..", "include", "esvio_fe.h"))):
        return out
    cxx = os.environ.get("CXX", "g++")
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-Wall", src, "-I" + os.path.abspath(os.path.join(HERE, "..", "include")),
                           "-L" + HERE, "-lesvio_fe", "-Wl,-rpath,$ORIGIN/../esvio_amd", "-Wl,-rpath,/opt/rocm/lib",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-ldl", "-o", out])
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
    print(build_tools(force="--force" in sys.argv))
