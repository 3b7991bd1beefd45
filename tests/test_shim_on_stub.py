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
    assert "zlng_encode_blocks_device" in syms and "zlng_decode_blocks_device" in syms
    stub_syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "tests", "cxx", "_stub", "libzlng_hip.so")],
                               stdout=subprocess.PIPE, text=True).stdout
    assert "zlng_encode_blocks_device" not in stub_syms
