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


def test_every_environment_variable_read_by_the_sources_is_documented():
    """include/zlng.h (or INTEGRATION.md for the C++ shim's) names every ZLNG_* variable the library, the shim, the tools,
    the build and bench.py read."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    used = set()
    for pat in ("libzling_amd/csrc/*.hip", "libzling_amd/csrc/*.h", "libzling_amd/cxx/*.cpp", "tools/*.cpp"):
        for f in glob.glob(os.path.join(root, pat)):
            used |= set(re.findall(r'getenv\("(ZLNG_[A-Z0-9_]+)"\)', open(f).read()))
    for f in ("bench.py", "__graft_entry__.py", "libzling_amd/__init__.py"):
        used |= set(re.findall(r'environ(?:\.get)?[\(\[]\s*"(ZLNG_[A-Z0-9_]+)"', open(os.path.join(root, f)).read()))
    docs = open(os.path.join(root, "include", "zlng.h")).read() + open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(v for v in used if v not in docs)
    assert not missing, missing


def test_generated_replay_loop_header_is_current():
    """csrc/replay_loop.h is generated (scripts/gen_replay_asm.py); the committed file must be what the generator writes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_replay_asm", os.path.join(ROOT, "scripts", "gen_replay_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    old = os.environ.pop("REPLAY_DROP", None)
    try:
        spec.loader.exec_module(mod)
        text = mod.render()
    finally:
        if old is not None:
            os.environ["REPLAY_DROP"] = old
    with open(os.path.join(ROOT, "libzling_amd", "csrc", "replay_loop.h")) as f:
        assert f.read() == text
