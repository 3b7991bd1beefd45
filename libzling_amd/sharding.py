"""Block-range sharding of ONE .zlng stream across ranks, one process per GPU.

The parse of a 16 MiB block depends only on that block (dictionary reset per block,
src/libzling.cpp:197); the literal ranks depend on the 256 MTF tables, which the reference keeps
for the whole stream (src/libzling_lz.cpp:197-209 does not reset them), and levels 1-4 carry
`current_level` from sub-block to sub-block and block to block (src/libzling.cpp:185, 261-266).
So a stream shards as

    rank r:  parse(range r)            -- all ranks at once, no communication
             recv state from r-1       -- 65,536 B of MTF tables + current_level   (the one exchange)
             rank + Huffman + frame    -- per rank
             send state to r+1

and the rank outputs concatenate to exactly the single-device stream.  A rank's range may be larger
than one context holds (240 blocks): `RangeEncoder` feeds it through several contexts, all parsing at
once, with the same state hand-off between them as between ranks.  `run_handoff` drives any object with
RangeEncoder's interface over any torch.distributed backend: RCCL ("nccl") on GPUs, gloo in the CPU tests.
"""
BLOCK = 16777216
MTF_STATE = 65536
STATE_BUF = MTF_STATE + 64          # the exchanged buffer: tables, then current_level in byte 65536 (always sent)


def plan(total_bytes, world, per_rank_bytes=None):
    """Contiguous block ranges: [(offset, length)] per rank; inner boundaries are block-aligned."""
    nblk = (total_bytes + BLOCK - 1) // BLOCK
    if per_rank_bytes is None:
        per = (nblk + world - 1) // world * BLOCK
    else:
        per = (per_rank_bytes + BLOCK - 1) // BLOCK * BLOCK
    out = []
    for r in range(world):
        off = min(r * per, total_bytes)
        end = total_bytes if r == world - 1 else min((r + 1) * per, total_bytes)
        out.append((off, end - off))
    return out


def split_blocks(nblocks, ctx_blocks):
    """Block counts of the contexts one rank uses: as few as fit, as even as possible."""
    if nblocks <= 0:
        return []
    k = (nblocks + ctx_blocks - 1) // ctx_blocks
    base, extra = divmod(nblocks, k)
    return [base + (1 if i < extra else 0) for i in range(k)]


class RangeEncoder:
    """One rank's block range, fed through as many contexts (`make_stream(nblocks)`) as it needs.

    parse(d_in, nbytes)                  queue the parse of every part (each context has its own HIP stream)
    finish(d_out, cap, d_state, level)   rank + Huffman part by part; the state buffer `d_state` (device pointer,
                                         65,536 B) holds the tables on entry and on exit; returns (segments, level)
                                         where segments = [(offset in d_out, bytes)] in stream order (every part
                                         starts on a 4-byte boundary of d_out, so the bytes of a rank are the
                                         concatenation of its segments, not one run)
    """

    def __init__(self, make_stream, nblocks, ctx_blocks=128):
        self.parts = split_blocks(nblocks, ctx_blocks)
        self.streams = [make_stream(p) for p in self.parts]
        self._lens = []

    def set_host_rank_contexts(self, k):
        """The measured hybrid of SURVEY 8(e) Option C on every context of the range: the k longest rank chains of each finish are
        walked by host threads while the device walks the others and the later ranges are still being parsed; 0 = all-device."""
        for s in self.streams:
            s.set_host_rank_contexts(k)

    def parse(self, d_in, nbytes):
        off = 0
        self._lens = []
        for s, p in zip(self.streams, self.parts):
            n = min(p * BLOCK, nbytes - off)
            s.parse_device(d_in + off, n)
            self._lens.append(n)
            off += n
        assert off == nbytes

    def finish(self, d_out, cap, d_state, level):
        segs, pos = [], 0
        for s in self.streams:
            s.set_state_device(d_state, level)
            n = s.finish_device(d_out + pos, cap - pos)
            level = s.get_state_device(d_state)
            segs.append((pos, n))
            pos = (pos + n + 3) & ~3
        return segs, level

    def timings(self):
        """Per-stage device milliseconds summed over the parts; the parse stages overlap in time, so their maximum is
        reported as well (`rolz_parse_max`)."""
        tot = {}
        for s in self.streams:
            for k, v in s.timings():
                tot[k] = tot.get(k, 0.0) + v
                if k == "rolz_parse":
                    tot["rolz_parse_max"] = max(tot.get("rolz_parse_max", 0.0), v)
        return tot

    def close(self):
        for s in self.streams:
            s.close()


def run_handoff(enc, rank, world, dist, state_buf, parse, finish, load_state, store_state, initial_level):
    """One sharded step on this rank.

    parse()                      start this rank's parse(s)
    load_state(buf) -> level     make the received buffer the encoder's entry state, return the entry level it carries
    finish(level) -> (result, level_out)
    store_state(buf, level)      put the encoder's exit state and level into the buffer
    `state_buf` is a uint8 tensor of STATE_BUF elements on the backend's device; byte 65536 carries current_level.
    Rank 0 enters with the stream's initial tables (already in the encoder's state buffer) and `initial_level`.
    """
    assert state_buf.numel() >= MTF_STATE + 1, "the exchanged buffer must carry current_level behind the 65,536 table bytes"
    parse()
    if rank == 0:
        level = initial_level
    else:
        dist.recv(state_buf, src=rank - 1)
        level = load_state(state_buf)
    result, level_out = finish(level)
    if rank < world - 1:
        store_state(state_buf, level_out)
        dist.send(state_buf, dst=rank + 1)
    return result, level_out
