"""The Python view of the C-ABI (libzling_amd/__init__.py: Stream, Group, encode -- the plumbing bench.py and the GPU tests stand on)
and sharding.RangeEncoder with REAL Stream objects, driven on the CPU against the stand-in of the ABI (tests/cxx/zlng_stub.c; in it
a "device pointer" is a host address).  What this covers is the binding code itself -- argument types, buffer lifetimes, the state
calls, block ends, error mapping -- and the hand-off protocol with the objects bench.py really uses instead of the fake of
tests/test_sharding_gloo.py.  Every check runs in a process of its own with ZLNG_HIP_SO pointing at the stand-in: the library handle
is cached per process, and the rest of the CPU suite loads the real HIP build (for the ABI export checks)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stub_so():
    import stub_build
    bins = stub_build.build()
    return os.path.join(os.path.dirname(bins["zling_demo"]), "libzlng_hip.so")


def run_py(stub_so, body, timeout=900):
    pre = ("import os, sys\nROOT = %r\n" % ROOT +
           "for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')): sys.path.insert(0, p)\n"
           "import numpy as np\nimport libzling_amd as zl\nfrom oracle_py import Oracle, textgen\no = Oracle()\n"
           "assert zl.lib().zlng_stub_marker() == 1\n")
    r = subprocess.run([sys.executable, "-c", pre + body], env=dict(os.environ, ZLNG_HIP_SO=stub_so), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])


def test_stream_and_encode(stub_so):
    run_py(stub_so, r'''
x = np.concatenate([textgen(2 * zl.BLOCK + 4321, 3), np.random.default_rng(1).integers(0, 256, 70000, dtype=np.uint8)])
for lv in (0, 3):
    want = o.encode(x, lv)
    assert np.array_equal(zl.encode(x, lv), want)
    with zl.Stream(0, lv, True, 1) as s:                                   # block by block: tables and level carried by the context
        parts = [s.encode(x[off: off + zl.BLOCK]) for off in range(0, x.size, zl.BLOCK)]
        assert s.block_ends and np.array_equal(np.concatenate(parts), want)
    with zl.Stream(0, lv, True, 3) as s:                                   # caller-owned buffers, block ends
        out = np.empty(zl.encode_bound(x.size), np.uint8)
        n = s.encode_into(x, out)
        assert n == want.size and np.array_equal(out[:n], want) and s.block_ends[-1] == n and len(s.block_ends) == 3
        mtf, level = s.get_state()
        assert mtf.size == 65536 and level in (0, lv)
    with zl.Stream(0, 0, False, 3) as d:                                   # decode: whole prefix (one call takes at most max_blocks), then the error mapping
        assert np.array_equal(d.decode(want, x.size), x)
        bad = want.copy(); bad[0] = 9
        try:
            d.decode(bad, x.size); raise SystemExit("no error")
        except zl.ZlngError as e:
            assert e.code == -10 and "invalid encflag" in str(e)
try:
    zl.Stream(0, 7, True, 1); raise SystemExit("level 7 accepted")
except zl.ZlngError as e:
    assert e.code == -1
print("ok")
''')


def test_device_pointer_calls_and_split_form(stub_so):
    run_py(stub_so, r'''
x = textgen(3 * zl.BLOCK - 99, 5)
want = o.encode(x, 4)
cap = zl.encode_bound(x.size)
out = np.zeros(cap + 64, np.uint8)
with zl.Stream(0, 4, True, 3) as s:
    n = s.encode_device(x.ctypes.data, x.size, out.ctypes.data, cap)
    assert n == want.size and np.array_equal(out[:n], want) and len(s.block_ends) == 3
# the split form over two contexts with the state handed on through a "device" buffer (what RangeEncoder does per context)
state = np.zeros(65536 + 64, np.uint8)
with zl.Stream(0, 4, True, 2) as a, zl.Stream(0, 4, True, 2) as b:
    lv = a.get_state_device(state.ctypes.data)
    a.parse_device(x.ctypes.data, 2 * zl.BLOCK)
    b.parse_after(a)
    b.parse_device(x.ctypes.data + 2 * zl.BLOCK, x.size - 2 * zl.BLOCK)
    a.set_state_device(state.ctypes.data, lv)
    n1 = a.finish_device(out.ctypes.data, cap)
    lv = a.get_state_device(state.ctypes.data)
    pos = (n1 + 3) & ~3
    b.set_state_device(state.ctypes.data, lv)
    n2 = b.finish_device(out.ctypes.data + pos, cap - pos)
    assert np.array_equal(np.concatenate([out[:n1], out[pos: pos + n2]]), want)
    try:
        b.finish_device(out.ctypes.data + 1, cap); raise SystemExit("misaligned output accepted")
    except zl.ZlngError as e:
        assert e.code == -1
with zl.Stream(0, 0, False, 4) as d:
    raw = np.zeros(4 * zl.BLOCK, np.uint8)
    used, n = d.decode_device(want.ctypes.data, want.size, raw.ctypes.data, raw.size)
    assert used == want.size and n == x.size and np.array_equal(raw[:n], x) and d.block_ends[2] == x.size
print("ok")
''')


def test_group(stub_so):
    run_py(stub_so, r'''
rng = np.random.default_rng(5)
x = np.concatenate([textgen(2 * zl.BLOCK - 400000, 7), rng.integers(0, 256, 900000, dtype=np.uint8), textgen(2 * zl.BLOCK + 5555, 8)])
for lv in (0, 4):
    want = o.encode(x, lv)
    for split in (False, True):
        with zl.Group([0, 0, 0], lv, 2) as g:
            g.set_host_rank_contexts(4)
            z = g.encode(x, split=split)
            assert np.array_equal(z, want) and g.block_ends[-1] == want.size
            mtf, level = g.get_state()
            g.set_state(mtf, level)
    with zl.Group([0, 0], lv, 1) as g:                                     # a stream in two calls: the group carries the state between them
        z1 = g.encode(x[: 2 * zl.BLOCK]); z2 = g.encode(x[2 * zl.BLOCK: 4 * zl.BLOCK]); z3 = g.encode(x[4 * zl.BLOCK:])
        assert np.array_equal(np.concatenate([z1, z2, z3]), want)
try:
    with zl.Group([0, 0], 0, 1) as g:
        g.encode(x)                                                        # five blocks into a group of capacity two
    raise SystemExit("over-capacity range accepted")
except zl.ZlngError as e:
    assert e.code == -1
print("ok")
''')


WORKER = r'''
import torch, torch.distributed as dist
from libzling_amd import sharding
rank, world, port, level, ctx_blocks, tmp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
dist.init_process_group("gloo", rank=rank, world_size=world)
total = 5 * zl.BLOCK + 77777
whole = textgen(total, 33)
if level:                                                                  # an incompressible stretch across the middle hand-off
    whole[3 * zl.BLOCK - 600000: 3 * zl.BLOCK + 100000] = np.random.default_rng(2).integers(0, 256, 700000, dtype=np.uint8)
off, n = sharding.plan(total, world)[rank]
x = np.ascontiguousarray(whole[off: off + n])
nb = (n + zl.BLOCK - 1) // zl.BLOCK
enc = sharding.RangeEncoder(lambda blocks: zl.Stream(0, level, True, blocks), nb, ctx_blocks, parses_in_flight=2)      # REAL Stream objects
out = np.zeros(zl.encode_bound(n) + 64, np.uint8)
state = torch.zeros(sharding.STATE_BUF, dtype=torch.uint8)
buf = torch.zeros(sharding.STATE_BUF, dtype=torch.uint8)
with zl.Stream(0, level, True, 1) as s0:
    init_level = s0.get_state_device(state.data_ptr())
def load_state(b):
    state.copy_(b); return int(b[sharding.MTF_STATE].item())
def store_state(b, lv):
    b.copy_(state); b[sharding.MTF_STATE] = lv
segs, lv_out = sharding.run_handoff(enc, rank, world, dist, buf, lambda: enc.parse(x.ctypes.data, n),
                                    lambda lv: enc.finish(out.ctypes.data, out.size, state.data_ptr(), lv), load_state, store_state, init_level)
z = np.concatenate([out[o_: o_ + k] for o_, k in segs]) if segs else np.empty(0, np.uint8)
z.tofile(os.path.join(tmp, "part%d.zlng" % rank))
if rank == 0:
    o.encode(whole, level).tofile(os.path.join(tmp, "whole.zlng"))
dist.barrier(); dist.destroy_process_group(); enc.close()
print("ok")
'''


@pytest.mark.parametrize("world,level,ctx_blocks", [(2, 0, 1), (3, 4, 2)])
def test_range_encoder_with_real_streams_over_gloo(stub_so, tmp_path, world, level, ctx_blocks):
    import numpy as np
    pre = ("import os, sys\nROOT = %r\n" % ROOT +
           "for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')): sys.path.insert(0, p)\n"
           "import numpy as np\nimport libzling_amd as zl\nfrom oracle_py import Oracle, textgen\no = Oracle()\nassert zl.lib().zlng_stub_marker() == 1\n")
    port = 29700 + os.getpid() % 1500 + 10 * world + level
    procs = [subprocess.Popen([sys.executable, "-c", pre + WORKER, str(r), str(world), str(port), str(level), str(ctx_blocks), str(tmp_path)],
                              env=dict(os.environ, ZLNG_HIP_SO=stub_so), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
             for r in range(world)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and so.strip().endswith("ok"), se[-3000:]
    parts = np.concatenate([np.fromfile(str(tmp_path / ("part%d.zlng" % r)), dtype=np.uint8) for r in range(world)])
    assert np.array_equal(parts, np.fromfile(str(tmp_path / "whole.zlng"), dtype=np.uint8))
