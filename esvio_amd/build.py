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
HEADERS = ["fe_kernels.h", "fe_host.h", "fe_ctx.h", "fe_internal.h", "fe_mc.h", os.path.join("..", "..", "include", "esvio_fe.h"),
           os.path.join("..", "..", "include", "esvio_fe_test.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # keep mul+add separate: bit parity with the x86 reference build
    "-fno-fast-math",
    "-fno-math-errno",            # sqrt() may be the instruction (same value), so lane loops vectorise
    "-Wall", "-Wno-unused-function",
    "-Wl,-rpath,/opt/rocm/lib",
    "-ldl",
]


OBJ = os.path.join(HERE, "build")  # per-source objects (git-ignored)


def _deps():
    return [os.path.join(CSRC, f) for f in HEADERS] + [os.path.abspath(__file__)]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES] + _deps()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """one object per source (compiled side by side, re-compiled only when the source or a header is
    newer), then the link: a host-side edit does not recompile fe_kernels.hip"""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    cflags = [f for f in FLAGS if f not in ("-shared", "-ldl") and not f.startswith("-Wl,")]
    newest_hdr = max(os.path.getmtime(d) for d in _deps())
    jobs, objs = [], []
    for f in SOURCES:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            cmd = [hipcc] + cflags + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            for _, q in jobs:
                q.wait()
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-ldl", "-o", LIB]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
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
