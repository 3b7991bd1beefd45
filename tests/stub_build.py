"""Builds the CPU stand-in of the C-ABI and everything above it into tests/cxx/_stub/ (TEST INFRASTRUCTURE; see tests/cxx/zlng_stub.c):

    libzlng_hip.so   = tests/cxx/zlng_stub.c (the context-level ABI on the CPU checker) + libzling_amd/csrc/zlng_group.hip (the
                       product's group driver: host code only, compiled as plain C++) + oracle/zlng_oracle.c
    libzling_amd.so  = the product's C++ shim, the same source file, linked against that
    zling_demo, protocol_test = the product's CLI and the protocol driver, linked against those

Nothing here is on the product path: the shipped libzling_amd.so links libzlng_hip.so (HIP) by rpath $ORIGIN and fails without it."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "cxx", "_stub")
INC = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "include", "libzling")]


def _stale(target, sources):
    return not os.path.exists(target) or any(os.path.getmtime(s) > os.path.getmtime(target) for s in sources)


def build(sanitize=False):
    """sanitize=True: the same stack under ASan + UBSan in tests/cxx/_stub_asan/ (scripts/sanitize.sh host-cpu); sanitize="thread":
    under ThreadSanitizer in tests/cxx/_stub_tsan/ (scripts/sanitize.sh host-tsan: the shim's helper thread, the group's copy-out threads)."""
    global OUT
    OUT = os.path.join(ROOT, "tests", "cxx", "_stub_tsan" if sanitize == "thread" else ("_stub_asan" if sanitize else "_stub"))
    san = (["-fsanitize=thread", "-fno-omit-frame-pointer", "-g", "-O1"] if sanitize == "thread" else
           ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"] if sanitize else ["-O2"])
    os.makedirs(OUT, exist_ok=True)
    import fcntl
    lock = open(os.path.join(OUT, ".lock"), "w")                      # several test workers may ask for the stack at once
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        return _build(OUT, san)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build(OUT, san):
    hdrs = glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "include", "libzling", "*.h")) + \
        glob.glob(os.path.join(ROOT, "oracle", "*.h"))
    abi = os.path.join(OUT, "libzlng_hip.so")
    stub_c = os.path.join(ROOT, "tests", "cxx", "zlng_stub.c")
    oracle_c = os.path.join(ROOT, "oracle", "zlng_oracle.c")
    group = os.path.join(ROOT, "libzling_amd", "csrc", "zlng_group.hip")
    if _stale(abi, [stub_c, oracle_c, group] + hdrs):
        objs = []
        for src, cc, extra in ((stub_c, "gcc", ["-std=c11"]), (oracle_c, "gcc", ["-std=c11"]), (group, "g++", ["-x", "c++", "-std=c++14"])):
            o = os.path.join(OUT, os.path.basename(src).rsplit(".", 1)[0] + ".o")
            subprocess.check_call([cc, "-fPIC", "-Wall"] + san + extra + ["-c", src, "-o", o])
            objs.append(o)
        subprocess.check_call(["g++", "-shared"] + san + ["-o", abi] + objs)
    shim = os.path.join(OUT, "libzling_amd.so")
    shim_src = sorted(glob.glob(os.path.join(ROOT, "libzling_amd", "cxx", "*.cpp")))
    if _stale(shim, shim_src + hdrs + [abi]):
        subprocess.check_call(["g++", "-std=c++14", "-fPIC", "-shared", "-pthread"] + san + INC + ["-o", shim] + shim_src +
                              ["-L", OUT, "-lzlng_hip", "-Wl,-rpath,$ORIGIN"])
    bins = {}
    for name, src in (("zling_demo", os.path.join(ROOT, "tools", "zling_demo.cpp")), ("protocol_test", os.path.join(ROOT, "tests", "cxx", "protocol_test.cpp"))):
        exe = os.path.join(OUT, name)
        if _stale(exe, [src, shim]):
            subprocess.check_call(["g++", "-std=c++14", "-Wall"] + san + INC + ["-o", exe, src, "-L", OUT, "-lzling_amd", "-lzlng_hip",
                                                                              "-Wl,-rpath," + OUT, "-pthread"])
        bins[name] = exe
    return bins


if __name__ == "__main__":
    import sys
    print(build(sanitize="thread" if "--tsan" in sys.argv else ("--sanitize" in sys.argv)))
