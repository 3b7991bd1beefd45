"""world_size-2 (and 3) CPU test of the block-range sharding protocol over gloo.

The GPU stream object is replaced by a fake with the same interface whose codec is the oracle, so
this checks the host logic that bench.py / multi-GPU callers run: range planning, parse-before-state,
the 64 KiB MTF hand-off, and that the concatenated rank outputs equal the single-stream encoding.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from libzling_amd import sharding  # noqa: E402


class FakeStream:
    """Stream-shaped wrapper over the oracle's zo_stream (parse is a no-op: the oracle parses in finish)."""

    def __init__(self, level):
        import ctypes as C
        from oracle_py import Oracle
        self.C = C
        self.o = Oracle()
        self.h = self.o.lib.zo_stream_new(level)
        self.level = level

    def set_state(self, mtf, level):
        m = np.ascontiguousarray(mtf, np.uint8)
        self.o.lib.zo_stream_set_mtf(self.h, m.ctypes.data_as(self.C.POINTER(self.C.c_uint8)))

    def get_state(self):
        m = np.empty(65536, np.uint8)
        self.o.lib.zo_stream_get_mtf(self.h, m.ctypes.data_as(self.C.POINTER(self.C.c_uint8)))
        return m, self.level

    def encode(self, x):
        C = self.C
        cap = self.o.lib.zo_encode_bound(x.size)
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        rc = self.o.lib.zo_encode_blocks(self.h, x.ctypes.data_as(C.POINTER(C.c_uint8)), x.size,
                                         out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n))
        assert rc == 0
        return out[: n.value].copy()


def _worker(rank, world, port, total, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_py import textgen
    off, n = sharding.plan(total, world)[rank]
    x = textgen(total, 90)[off:off + n]           # every rank generates the same stream, keeps its range
    s = FakeStream(0)
    init, lv0 = FakeStream(0).get_state()
    buf = torch.empty(65536, dtype=torch.uint8)
    order = []
    out = {}

    def parse():
        order.append("parse")

    def finish():
        order.append("finish")
        out["z"] = s.encode(x)
        return out["z"].size

    def to_buf(b):
        m, lv = s.get_state()
        b.copy_(torch.from_numpy(m))
        return lv

    def from_buf(b, lv):
        order.append("state")
        s.set_state(b.numpy(), lv)

    sharding.run_handoff(s, rank, world, dist, buf, init, lv0, 0, parse, finish, to_buf, from_buf)
    assert order[0] == "parse" and order[-1] == "finish"
    out["z"].tofile(os.path.join(tmp, "part%d.zlng" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 2 * sharding.BLOCK + 123_457), (3, 3 * sharding.BLOCK - 5)])
def test_block_range_sharding_equals_single_stream(tmp_path, world, total):
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    from oracle_py import Oracle, textgen
    whole = Oracle().encode(textgen(total, 90), 0)
    parts = np.concatenate([np.fromfile(os.path.join(str(tmp_path), "part%d.zlng" % r), dtype=np.uint8) for r in range(world)])
    assert np.array_equal(parts, whole)


def test_plan_is_block_aligned_and_covers():
    for total in (1, sharding.BLOCK, 10 ** 9, 8 * 10 ** 9 + 7):
        for world in (1, 2, 4, 8):
            p = sharding.plan(total, world)
            assert sum(n for _, n in p) == total
            assert all(off % sharding.BLOCK == 0 or off == total for off, _ in p)
            assert all(p[i][0] + p[i][1] == p[i + 1][0] for i in range(world - 1))
