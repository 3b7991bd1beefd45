"""The C-ABI library loads on a machine without a GPU and exports every symbol include/zlng.h declares;
argument checks that need no device behave; nothing falls back to a CPU implementation."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from libzling_amd import build
    build.build_all()
    import libzling_amd as zl
    return zl.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zlng.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zlng_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), s


def test_shim_exports_the_reference_api():
    so = os.path.join(ROOT, "libzling_amd", "libzling_amd.so")
    assert os.path.exists(so)
    out = os.popen("nm -D --defined-only %s | c++filt" % so).read()
    for want in ("baidu::zling::Encode(baidu::zling::Inputter*, baidu::zling::Outputter*, baidu::zling::ActionHandler*, int)",
                 "baidu::zling::Decode(baidu::zling::Inputter*, baidu::zling::Outputter*, baidu::zling::ActionHandler*)",
                 "baidu::zling::Inputter::GetUInt32()", "baidu::zling::Outputter::PutUInt32(unsigned int)",
                 "baidu::zling::FileInputter::IsEnd()", "baidu::zling::FileOutputter::GetOutputSize()"):
        assert want in out, want


def test_bound_and_strerror_need_no_device(lib):
    import libzling_amd as zl
    assert zl.encode_bound(0) >= 64
    assert zl.encode_bound(10 ** 9) > 15 * 10 ** 8           # worst case: 393,216 payload bytes per 262,143 input bytes
    assert zl.strerror(0) == "ok"
    assert zl.strerror(-11) == "baidu::zling::Decode(): invalid block size."      # src/libzling.cpp:327


def test_bound_covers_worst_case_of_the_oracle(oracle):
    import libzling_amd as zl
    rng = np.random.Generator(np.random.PCG64(1))
    for n in (1, 1000, 300_000, 1 << 20):
        z = oracle.encode(rng.integers(0, 256, n, dtype=np.uint8), 0)
        assert z.size <= zl.encode_bound(n)


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import libzling_amd as zl
    if lib.zlng_device_count() > 0:
        pytest.skip("a gfx950 device is present")
    with pytest.raises(zl.ZlngError) as e:
        zl.Stream(0, 0, True, 1)
    assert e.value.code == -4                                   # ZLNG_E_DEVICE
    with pytest.raises(zl.ZlngError):
        zl.encode(b"hello")
