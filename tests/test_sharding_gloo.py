"""world_size-2, 3, 4 and 8 CPU tests of the block-range sharding protocol over gloo.

The GPU stream object is replaced by a fake with the same interface whose codec is the oracle, so
this checks the host logic that bench.py / multi-GPU callers run: range planning, per-rank batching over
several contexts (RangeEncoder), parse-before-state, the hand-off of the 64 KiB MTF tables AND of
current_level (src/libzling.cpp:185, 261-266), and that the concatenated rank outputs equal the
single-stream encoding -- at e0 and at e4 with a range that ENDS on an incompressible sub-block.
"""
import ctypes as C
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

BLOCK = sharding.BLOCK
_u8p = C.POINTER(C.c_uint8)


class FakeStream:
    """Stream-shaped wrapper over the oracle's zo_stream, with "device pointers" that are host addresses.
    parse_device only records the range (the oracle parses inside finish); the state calls round-trip both
    the tables and current_level, like zlng_{get,set}_state_device."""

    def __init__(self, level, blocks):
        from oracle_py import Oracle
        self.o = Oracle()
        self.h = self.o.lib.zo_stream_new(level)
        self.level = level
        self.max_blocks = blocks
        self.pending = None

    def close(self):
        pass

    def set_host_rank_contexts(self, k):
        # the fake's codec is the oracle either way; what the test pins is that the switch reaches EVERY context of every rank
        self.host_rank_contexts = k

    def parse_device(self, ptr, n):
        assert (n + BLOCK - 1) // BLOCK <= self.max_blocks
        self.pending = (ptr, n)

    def set_state_device(self, ptr, level):
        assert level in (0, self.level)
        self.o.lib.zo_stream_set_mtf(self.h, C.cast(ptr, _u8p))
        self.o.lib.zo_stream_set_level(self.h, level)

    def get_state_device(self, ptr):
        self.o.lib.zo_stream_get_mtf(self.h, C.cast(ptr, _u8p))
        return self.o.lib.zo_stream_get_level(self.h)

    def finish_device(self, out_ptr, cap):
        ptr, n = self.pending
        self.pending = None
        got = C.c_size_t(0)
        rc = self.o.lib.zo_encode_blocks(self.h, C.cast(ptr, _u8p), n, C.cast(out_ptr, _u8p), cap, C.byref(got))
        assert rc == 0
        return got.value

    def timings(self):
        return []


def make_input(kind, total):
    from oracle_py import textgen
    if kind == "text":
        return textgen(total, 90)
    # "mixed": text, with an incompressible stretch that covers the END of the first block range and the start of the
    # next one, so at e1-e4 the first range ends at current_level 0 and the next one must start there
    x = textgen(total, 91)
    rng = np.random.Generator(np.random.PCG64(7))
    edge = sharding.plan(total, 2)[1][0]                     # where rank 1's range starts in the world-2 test
    lo, hi = edge - 700_000, min(total, edge + 200_000)
    x[lo:hi] = rng.integers(0, 256, hi - lo, dtype=np.uint8)
    return x


def _worker(rank, world, port, total, tmp, kind, level, ctx_blocks, hybrid=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, n = sharding.plan(total, world)[rank]
    x = np.ascontiguousarray(make_input(kind, total)[off:off + n])      # every rank generates the same stream, keeps its range
    nb = (n + BLOCK - 1) // BLOCK
    enc = sharding.RangeEncoder(lambda blocks: FakeStream(level, blocks), nb, ctx_blocks)
    assert len(enc.streams) == (nb + ctx_blocks - 1) // ctx_blocks
    if hybrid:
        enc.set_host_rank_contexts(4)
        assert all(getattr(s, "host_rank_contexts", 0) == 4 for s in enc.streams)
    out = np.empty(FakeStream(level, 1).o.lib.zo_encode_bound(n) + 64, np.uint8)
    state = torch.zeros(sharding.STATE_BUF, dtype=torch.uint8)           # this rank's state buffer ("device" = host here)
    buf = torch.zeros(sharding.STATE_BUF, dtype=torch.uint8)             # the exchanged buffer
    init_level = FakeStream(level, 1).get_state_device(state.data_ptr())
    order, entry = [], {}

    def parse():
        order.append("parse")
        enc.parse(x.ctypes.data, n)

    def load_state(b):
        order.append("state")
        state.copy_(b)
        entry["level"] = int(b[sharding.MTF_STATE].item())
        return entry["level"]

    def finish(lv):
        order.append("finish")
        return enc.finish(out.ctypes.data, out.size, state.data_ptr(), lv)

    def store_state(b, lv):
        b.copy_(state)
        b[sharding.MTF_STATE] = lv

    segs, lv_out = sharding.run_handoff(enc, rank, world, dist, buf, parse, finish, load_state, store_state, init_level)
    assert order[0] == "parse" and order[-1] == "finish"
    z = np.concatenate([out[o:o + k] for o, k in segs]) if segs else np.empty(0, np.uint8)      # (a rank behind the end of a short stream has no range)
    z.tofile(os.path.join(tmp, "part%d.zlng" % rank))
    with open(os.path.join(tmp, "level%d.txt" % rank), "w") as f:
        f.write("%d %d" % (entry.get("level", -1), lv_out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,kind,level,ctx_blocks", [
    (2, 2 * BLOCK + 123_457, "text", 0, 240),
    (3, 3 * BLOCK - 5, "text", 0, 240),
    (2, 2 * BLOCK + 300_000, "mixed", 4, 240),          # rank 0's range ends incompressible: rank 1 must enter at level 0
    (2, 4 * BLOCK + 50_000, "text", 0, 1),              # per-rank batching: 2-3 contexts of one block per rank
    (2, 4 * BLOCK + 50_001, "text", 0, 1),              # the same with the host-rank-chain switch set on every context (odd size = hybrid)
    (4, 4 * BLOCK - 9, "mixed", 4, 240),                # BASELINE's 4- and 8-GPU shapes: the state crosses 3 / 7 hand-offs, at e4 the level with it
    (4, 4 * BLOCK + 222_222, "text", 0, 240),           # five blocks over four ranks = 2 + 2 + 1 + 0: the last rank's range is EMPTY and only passes the state on
    (8, 8 * BLOCK - 7, "text", 0, 240),                 # (rank 2 of 4 enters at level 0: the incompressible stretch ends rank 1's range)
])
def test_block_range_sharding_equals_single_stream(tmp_path, world, total, kind, level, ctx_blocks):
    port = 29500 + (os.getpid() % 2000) + world + 7 * level + ctx_blocks % 5 + total % 3
    mp.spawn(_worker, args=(world, port, total, str(tmp_path), kind, level, ctx_blocks, total % 2 == 1 and ctx_blocks == 1), nprocs=world, join=True)
    from oracle_py import Oracle
    whole = Oracle().encode(make_input(kind, total), level)
    parts = np.concatenate([np.fromfile(os.path.join(str(tmp_path), "part%d.zlng" % r), dtype=np.uint8) for r in range(world)])
    assert np.array_equal(parts, whole)
    if kind == "mixed":
        edge = sharding.plan(total, 2)[1][0]                 # where make_input put the incompressible stretch: the range that starts there
        r_in = next(r for r, (off, _n) in enumerate(sharding.plan(total, world)) if off == edge)
        entered, _ = (int(v) for v in open(os.path.join(str(tmp_path), "level%d.txt" % r_in)).read().split())
        assert entered == 0, "rank %d must receive current_level 0 from a range that ended incompressible" % r_in
        if r_in > 1:                                         # ... and the ranks in front of it entered at the stream's level
            assert int(open(os.path.join(str(tmp_path), "level1.txt")).read().split()[0]) == level


def test_handoff_refuses_a_buffer_without_room_for_the_level():
    with pytest.raises(AssertionError):
        sharding.run_handoff(None, 0, 1, None, torch.zeros(65536, dtype=torch.uint8), lambda: None, lambda lv: (None, lv),
                             lambda b: 0, lambda b, lv: None, 0)


def test_plan_is_block_aligned_and_covers():
    for total in (1, BLOCK, 10 ** 9, 8 * 10 ** 9 + 7):
        for world in (1, 2, 4, 8):
            p = sharding.plan(total, world)
            assert sum(n for _, n in p) == total
            assert all(off % BLOCK == 0 or off == total for off, _ in p)
            assert all(p[i][0] + p[i][1] == p[i + 1][0] for i in range(world - 1))


def test_split_blocks():
    assert sharding.split_blocks(512, 128) == [128] * 4
    assert sharding.split_blocks(512, 240) == [171, 171, 170]
    assert sharding.split_blocks(60, 128) == [60]
    assert sharding.split_blocks(0, 128) == []
    for n in range(1, 700, 37):
        for c in (1, 7, 128, 240):
            p = sharding.split_blocks(n, c)
            assert sum(p) == n and max(p) <= c and max(p) - min(p) <= 1


def test_range_encoder_orders_the_parses_of_its_contexts():
    """RangeEncoder(parses_in_flight=P): the parse of context k is queued behind the parse of context k - P (zlng_encode_parse_after
    through Stream.parse_after) and in front of nothing else; P = 0 queues every parse at once.  Recorded with streams that only log."""
    log = []

    class Rec:
        def __init__(self, ix):
            self.ix = ix

        def parse_after(self, first):
            log.append(("after", self.ix, first.ix))

        def parse_device(self, ptr, n):
            log.append(("parse", self.ix, n))

    for pif, want in ((2, [(2, 0), (3, 1), (4, 2)]), (3, [(3, 0), (4, 1)]), (0, [])):
        del log[:]
        made = []
        enc = sharding.RangeEncoder(lambda blocks: made.append(Rec(len(made))) or made[-1], 5 * 10, 10, pif)
        assert enc.parts == [10] * 5
        enc.parse(1 << 40, 50 * BLOCK - 123)
        assert [(a, b) for kind, a, b in log if kind == "after"] == want
        assert [a for kind, a, b in log if kind == "parse"] == [0, 1, 2, 3, 4]
        for k in range(5):                                    # a context's "after" comes immediately in front of its own parse
            if (k, k - pif) in want:
                assert log.index(("after", k, k - pif)) + 1 == [i for i, e in enumerate(log) if e[0] == "parse" and e[1] == k][0]
        assert sum(b for kind, a, b in log if kind == "parse") == 50 * BLOCK - 123


class _LogStream:
    """A stream that only logs (and can be told to fail its parse), with the finish-side methods RangeEncoder calls."""

    def __init__(self, ix, log, fail=False):
        self.ix, self.log, self.fail = ix, log, fail

    def parse_device(self, ptr, n):
        if self.fail:
            raise RuntimeError("ZLNG_E_NOMEM (made up)")
        self.log.append(("parse", self.ix))

    def set_state_device(self, d_state, level):
        pass

    def finish_device(self, d_out, cap):
        self.log.append(("finish", self.ix))
        return 8

    def get_state_device(self, d_state):
        return 0

    def close(self):
        pass


def test_staggered_parse_that_fails_on_the_helper_thread_surfaces_in_finish_instead_of_hanging():
    """the helper thread of a staggered schedule died silently when a late parse raised, and finish() then waited for
    its event forever (on several GPUs: every later rank hung in the hand-off).  Now the error is kept, every event is set and
    finish() re-raises it at the context that lacks its parse."""
    import threading
    log, made = [], []
    enc = sharding.RangeEncoder(lambda blocks: made.append(_LogStream(len(made), log, fail=(len(made) == 2))) or made[-1], 40, 10,
                                stagger=(1, 0.01))
    enc.parse(1 << 40, 40 * BLOCK)
    res = {}

    def run():
        try:
            enc.finish(1 << 41, 1 << 30, 1 << 42, 0)
        except Exception as e:
            res["err"] = e
    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(10.0)
    assert not t.is_alive(), "finish() hangs behind a parse that was never queued"
    assert isinstance(res.get("err"), RuntimeError) and "ZLNG_E_NOMEM" in str(res["err"].__cause__)
    assert ("finish", 0) in log and ("finish", 1) in log and ("finish", 2) not in log      # what was parsed still finishes in order
    # contexts 0 and 1 have moved the tables on, context 3 holds a parse nobody will finish -- the encoder says so on
    # every further use instead of queueing new parses onto streams in mixed states
    for again in (lambda: enc.parse(1 << 40, 40 * BLOCK), lambda: enc.finish(1 << 41, 1 << 30, 1 << 42, 0)):
        with pytest.raises(RuntimeError, match="close it"):
            again()
    enc.close()


def test_staggered_launch_times_need_not_be_a_zero_prefix_and_must_match_the_contexts():
    """('at', times): a context is launched at once iff ITS time is <= 0, wherever it stands in the range; a list of the wrong
    length is an error at parse(), not an IndexError on the helper thread."""
    log, made = [], []
    enc = sharding.RangeEncoder(lambda blocks: made.append(_LogStream(len(made), log)) or made[-1], 30, 10, stagger=("at", [0.05, 0.0, 0.02]))
    enc.parse(1 << 40, 30 * BLOCK)
    assert log == [("parse", 1)]                                    # only context 1 is due at t = 0
    segs, _lv = enc.finish(1 << 41, 1 << 30, 1 << 42, 0)
    assert [e for e in log if e[0] == "parse"] == [("parse", 1), ("parse", 2), ("parse", 0)] and len(segs) == 3
    enc.close()
    bad = sharding.RangeEncoder(lambda blocks: _LogStream(0, log), 30, 10, stagger=("at", [0.0, 0.1]))
    with pytest.raises(ValueError):
        bad.parse(1 << 40, 30 * BLOCK)


def _digest_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    buf = np.frombuffer(bytes([rank + 1]) * (1000 + rank), dtype=np.uint8)
    q.put((rank, bench.gather_digests(buf, world, "cpu")))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_gathers_every_ranks_size_and_sha256_on_every_rank():
    """bench.py's PASS/FAIL column at N > 1 starts from every rank's own size + SHA-256, gathered with one all_gather: three ranks
    over gloo, each with a different buffer, and every rank must end up with the same, correct list."""
    import hashlib
    world, port = 3, 29655
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_digest_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = [(1000 + r, hashlib.sha256(bytes([r + 1]) * (1000 + r)).hexdigest()) for r in range(world)]
    assert res[0] == want and res[1] == want and res[2] == want
