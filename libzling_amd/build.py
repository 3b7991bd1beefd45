"""Build the native pieces in-tree (no JIT cache): HIP kernels + C-ABI, text generator, C++ shim.

    python -m libzling_amd.build            # build everything that is stale
"""
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HIP_SO = os.path.join(PKG, "libzlng_hip.so")
TEXTGEN_SO = os.path.join(PKG, "host", "libzlng_textgen.so")
SHIM_SO = os.path.join(PKG, "libzling_amd.so")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


# per-file code generation options (measured on MI355X)
PER_FILE_FLAGS = {}


def build_hip(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "zlng.h")]
    if not (force or _stale(HIP_SO, deps)):
        return HIP_SO
    objs = []
    for s in srcs:
        o = s[:-4] + ".o"
        if force or _stale(o, deps):
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                   "-Wall", "-Wno-unused-function", "-Wno-unused-value"] + PER_FILE_FLAGS.get(os.path.basename(s), []) + \
                  os.environ.get("ZLNG_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_SO] + objs)
    return HIP_SO


def build_textgen(force=False):
    src = os.path.join(PKG, "host", "textgen.c")
    if force or _stale(TEXTGEN_SO, [src]):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-o", TEXTGEN_SO, src, "-lm"])
    return TEXTGEN_SO


def build_shim(force=False):
    srcs = sorted(glob.glob(os.path.join(PKG, "cxx", "*.cpp")))
    if not srcs:
        return None
    deps = srcs + glob.glob(os.path.join(ROOT, "include", "libzling", "*.h")) + [os.path.join(ROOT, "include", "zlng.h")]
    if force or _stale(SHIM_SO, deps) or _stale(SHIM_SO, [HIP_SO]):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "include", "libzling"), "-o", SHIM_SO] + srcs +
                              ["-L", PKG, "-lzlng_hip", "-Wl,-rpath,$ORIGIN", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return SHIM_SO


DEMO = os.path.join(ROOT, "tools", "zling_demo")


def build_demo(force=False):
    src = os.path.join(ROOT, "tools", "zling_demo.cpp")
    if force or _stale(DEMO, [src, SHIM_SO]):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-I", os.path.join(ROOT, "include", "libzling"),
                               "-I", os.path.join(ROOT, "include"), "-o", DEMO, src, "-L", PKG, "-lzling_amd", "-lzlng_hip",
                               "-Wl,-rpath," + PKG, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return DEMO


PROTOCOL_TEST = os.path.join(ROOT, "tests", "cxx", "protocol_test")


def build_protocol_test(force=False):
    """tests/cxx/protocol_test.cpp: a `-lzling` user with its own Inputter / Outputter / ActionHandler (tests/test_gpu_protocol.py)."""
    src = os.path.join(ROOT, "tests", "cxx", "protocol_test.cpp")
    if force or _stale(PROTOCOL_TEST, [src, SHIM_SO]):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-I", os.path.join(ROOT, "include", "libzling"),
                               "-I", os.path.join(ROOT, "include"), "-o", PROTOCOL_TEST, src, "-L", PKG, "-lzling_amd", "-lzlng_hip",
                               "-Wl,-rpath," + PKG, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    return PROTOCOL_TEST


def build_all(force=False, verbose=False):
    import fcntl
    with open(os.path.join(PKG, ".build.lock"), "w") as lock:       # one builder at a time (parallel test workers all ask for an up-to-date tree)
        fcntl.flock(lock, fcntl.LOCK_EX)
        build_textgen(force)
        build_hip(force, verbose)
        build_shim(force)
        build_demo(force)
        build_protocol_test(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("built:", HIP_SO)
