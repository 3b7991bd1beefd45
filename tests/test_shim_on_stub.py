"""The HOST logic above the C-ABI -- the C++ shim's Encode / Decode drivers (batching, the two-slot pipeline, the exact-pull and
the read-ahead decode paths, the order in which errors are met), the group driver (zlng_group.hip: host code), tools/zling_demo
and the callback protocol -- run on the CPU against a stand-in of the context-level ABI that restates its contract on the CPU
checker (tests/cxx/zlng_stub.c).  The test bodies are the GPU suite's own (tests/test_gpu_cli.py, test_gpu_protocol.py,
test_gpu_zz_cli_error_order.py): they only drive binaries, so the same expectations hold the real library on the GPU box and the
host code here.  What this does NOT cover: anything below the ABI (the kernels) -- that is what `-m gpu` is for."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stub():
    import stub_build
    return stub_build.build()


def run_gpu_tests_on_stub(stub, files, deselect=""):
    env = dict(os.environ, ZLNG_DEMO=stub["zling_demo"], ZLNG_PROTOCOL_TEST=stub["protocol_test"])
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-x"] + [os.path.join(ROOT, "tests", f) for f in files]
    if deselect:
        cmd += ["-k", deselect]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, cwd=ROOT)
    tail = r.stdout[-3000:]
    assert r.returncode == 0, tail
    return tail


def test_cli_tests_of_the_gpu_suite_hold_the_shim_on_the_stub(stub):
    # (the two tests that call the Python binding of the real library need the GPU)
    tail = run_gpu_tests_on_stub(stub, ["test_gpu_cli.py"], "not python_stream and not split_host_api")
    assert " passed" in tail and "failed" not in tail


def test_unterminated_final_block_through_the_cli_on_the_stub(stub):
    for ra in ("0", "1"):                                                # both Decode paths of the shim
        os.environ["ZLNG_DECODE_READAHEAD"] = ra
        try:
            tail = run_gpu_tests_on_stub(stub, ["test_gpu_decode.py"], "unterminated_final_block")
        finally:
            del os.environ["ZLNG_DECODE_READAHEAD"]
        assert "1 passed" in tail


def test_protocol_tests_of_the_gpu_suite_hold_the_shim_on_the_stub(stub):
    tail = run_gpu_tests_on_stub(stub, ["test_gpu_protocol.py"])
    assert " passed" in tail and "failed" not in tail


def test_decode_error_order_tests_of_the_gpu_suite_hold_the_shim_on_the_stub(stub):
    tail = run_gpu_tests_on_stub(stub, ["test_gpu_zz_cli_error_order.py"])
    assert " passed" in tail and "failed" not in tail


def test_the_shipped_shim_links_the_hip_library_not_the_stub():
    """The stand-in lives under tests/cxx/_stub/ only; the product's libzling_amd.so names libzlng_hip.so and finds the HIP build
    next to itself ($ORIGIN), and that library defines the kernels' entry points the stand-in does not have."""
    from libzling_amd import build
    build.build_all()
    out = subprocess.run(["readelf", "-d", build.SHIM_SO], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "libzlng_hip.so" in out and "$ORIGIN" in out
    syms = subprocess.run(["nm", "-D", "--defined-only", build.HIP_SO], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "zlng_encode_blocks_device" in syms and "zlng_debug_lengths" in syms and "zlng_stub_marker" not in syms
    needed = subprocess.run(["readelf", "-d", build.HIP_SO], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "libamdhip64" in needed                                       # the shipped library is the HIP build
    stub_so = os.path.join(ROOT, "tests", "cxx", "_stub", "libzlng_hip.so")
    stub_syms = subprocess.run(["nm", "-D", "--defined-only", stub_so], stdout=subprocess.PIPE, text=True).stdout
    assert "zlng_stub_marker" in stub_syms and "zlng_debug_lengths" not in stub_syms
    assert "libamdhip64" not in subprocess.run(["readelf", "-d", stub_so], stdout=subprocess.PIPE, text=True).stdout


def test_group_driver_on_the_stub_failed_finish_leaves_the_state_and_the_range_can_be_resubmitted(stub, oracle):
    """zlng_group.hip (the product's host code, compiled into the stand-in library) through ctypes: three members, a finish whose
    output buffer is too small for the second member fails with ZLNG_E_CAP AFTER the first member's copy-out thread was started --
    the driver joins it, reports the error, leaves the group's stream state as it was (zlng.h: "after a failed finish the group's
    state is unchanged and the range has to be submitted again"), and the resubmitted range produces the single-stream bytes with
    every block's end offset."""
    import ctypes as C

    import numpy as np
    from oracle_py import textgen
    L = C.CDLL(os.path.join(os.path.dirname(stub["zling_demo"]), "libzlng_hip.so"))
    u8p, szp = C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)
    L.zlng_group_create.restype = C.c_void_p
    L.zlng_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.zlng_group_destroy.argtypes = [C.c_void_p]
    L.zlng_group_encode_parse.argtypes = [C.c_void_p, u8p, C.c_size_t]
    L.zlng_group_encode_finish.argtypes = [C.c_void_p, u8p, C.c_size_t, szp, szp]
    L.zlng_group_get_state.argtypes = [C.c_void_p, u8p, C.POINTER(C.c_int)]
    L.zlng_encode_bound.restype = C.c_size_t
    L.zlng_encode_bound.argtypes = [C.c_size_t]
    BLOCK = 1 << 24
    x = np.ascontiguousarray(textgen(3 * BLOCK - 1234, 17))
    want = oracle.encode(x, 4)
    devs = (C.c_int * 3)(0, 0, 0)
    err = C.c_int(0)
    g = L.zlng_group_create(devs, 3, 4, 1, C.byref(err))
    assert g and err.value == 0
    p = lambda a: a.ctypes.data_as(u8p)
    st0, st1 = np.empty(65536, np.uint8), np.empty(65536, np.uint8)
    lv = C.c_int(-1)
    assert L.zlng_group_get_state(g, p(st0), C.byref(lv)) == 0 and lv.value == 4
    out = np.empty(L.zlng_encode_bound(x.size), np.uint8)
    n = C.c_size_t(0)
    ends = (C.c_size_t * 3)()
    assert L.zlng_group_encode_parse(g, p(x), x.size) == 0
    small = want.size * 1 // 2                                           # room for member 0's block, not for member 1's
    assert L.zlng_group_encode_finish(g, p(out), small, C.byref(n), ends) == -3      # ZLNG_E_CAP
    assert n.value == 0
    assert L.zlng_group_get_state(g, p(st1), C.byref(lv)) == 0 and lv.value == 4 and np.array_equal(st0, st1)
    assert L.zlng_group_encode_finish(g, p(out), out.size, C.byref(n), ends) == -1    # nothing pending any more: ZLNG_E_ARG
    assert L.zlng_group_encode_parse(g, p(x), x.size) == 0                             # submit the range again
    assert L.zlng_group_encode_finish(g, p(out), out.size, C.byref(n), ends) == 0
    assert n.value == want.size and np.array_equal(out[: n.value], want)
    assert ends[2] == want.size and all(want[ends[b] - 1] == 0 for b in range(3))
    assert L.zlng_group_get_state(g, p(st1), C.byref(lv)) == 0 and not np.array_equal(st0, st1)
    L.zlng_group_destroy(g)
