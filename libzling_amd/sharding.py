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

    def __init__(self, make_stream, nblocks, ctx_blocks=128, parses_in_flight=2, stagger=None, parts=None):
        self.parts = list(parts) if parts else split_blocks(nblocks, ctx_blocks)     # parts: explicit block counts of the contexts
        assert sum(self.parts) == nblocks and all(0 < p <= 240 for p in self.parts)
        self.streams = [make_stream(p) for p in self.parts]
        self._lens = []
        # how many contexts parse at a time (0 = all at once, the schedule of rounds 1-3).  With two, the parse of context k + 2
        # starts when context k's has finished: parses END in stream order, and the rank stage of context k -- which has to wait
        # for the tables context k - 1 leaves -- runs beside the parse of context k + 1 (schedule_model below is the arithmetic)
        self.parses_in_flight = parses_in_flight
        # stagger = (first, gap_s), ("at", [t_s per context]) or "auto" -- SECONDS here (time.sleep); schedule_model* below takes the
        # same shapes in MILLISECONDS (it adds them to stage times, which the library reports in ms).  Launches spread in TIME, which no dependency between parses can express -- the first
        # `first` contexts are queued at once, context k (k - first + 1) * gap_s seconds later by a helper thread.  With one context
        # first and one more per rank-stage duration ("auto": gap = the rank stage's measured time per block, 10.4 ms before anything
        # was measured, times the context's blocks) few blocks are in flight at the start, so the first parse ends early and the
        # chain starts early; later contexts arrive at the pace the chain consumes them (profiles/r04_b: 8.08 -> 7.6 s on config 4's
        # share).  Takes precedence over parses_in_flight.
        self.stagger = stagger
        self.rank_ms_per_block = 10.4
        self._queued = []

    def set_host_rank_contexts(self, k):
        """The measured hybrid of SURVEY 8(e) Option C on every context of the range: the k longest rank chains of each finish are
        walked by host threads while the device walks the others and the later ranges are still being parsed; 0 = all-device."""
        for s in self.streams:
            s.set_host_rank_contexts(k)

    def parse(self, d_in, nbytes):
        if getattr(self, "_broken", None) is not None:
            raise RuntimeError("this RangeEncoder failed in an earlier step (a staggered parse could not be queued): close it") from self._broken
        off = 0
        self._lens = []
        jobs = []
        for k, (s, p) in enumerate(zip(self.streams, self.parts)):
            n = min(p * BLOCK, nbytes - off)
            jobs.append((k, s, d_in + off, n))
            self._lens.append(n)
            off += n
        assert off == nbytes
        if self.stagger and len(jobs) > 1:
            import threading
            import time
            plan = self.stagger_plan()
            if plan[0] == "at":                          # explicit launch times (seconds into the step), one per context
                at = list(plan[1])
                if len(at) != len(jobs):
                    raise ValueError("stagger ('at', times): %d launch times for %d contexts" % (len(at), len(jobs)))
            else:
                first, gap = plan
                at = [0.0 if k < first else (k - first + 1) * gap for k in range(len(jobs))]
            now = [j for j in jobs if at[j[0]] <= 0.0]                       # per context, wherever it stands in the range
            later = sorted((j for j in jobs if at[j[0]] > 0.0), key=lambda j: at[j[0]])
            self._queued = [threading.Event() for _ in jobs]
            self._late_error = None
            t0 = time.perf_counter()

            def late():
                try:
                    for k, s, ptr, n in later:
                        d = t0 + at[k] - time.perf_counter()
                        if d > 0:
                            time.sleep(d)
                        s.parse_device(ptr, n)
                        self._queued[k].set()
                except BaseException as e:               # finish() must not wait for a parse that will never be queued
                    self._late_error = e
                finally:
                    for ev in self._queued:
                        ev.set()
            for k, s, ptr, n in now:
                s.parse_device(ptr, n)
                self._queued[k].set()
            self._late = threading.Thread(target=late, daemon=True)
            self._late.start()
            return
        self._queued = []
        for k, s, ptr, n in jobs:
            if self.parses_in_flight > 0 and k >= self.parses_in_flight and hasattr(s, "parse_after"):
                s.parse_after(self.streams[k - self.parses_in_flight])
            s.parse_device(ptr, n)

    def stagger_plan(self):
        """(first, gap_s) the next parse() will use."""
        if self.stagger == "auto":
            return 1, self.rank_ms_per_block * max(self.parts) * 1e-3
        if self.stagger[0] == "at":
            return "at", [float(t) for t in self.stagger[1]]
        return int(self.stagger[0]), float(self.stagger[1])

    def finish(self, d_out, cap, d_state, level):
        if getattr(self, "_broken", None) is not None:
            raise RuntimeError("this RangeEncoder failed in an earlier step (a staggered parse could not be queued): close it") from self._broken
        segs, pos = [], 0
        for k, s in enumerate(self.streams):
            if self._queued:
                self._queued[k].wait()                     # (the helper thread of a staggered schedule has queued this context's parse)
                if getattr(self, "_late_error", None) is not None:
                    # Some contexts of this range hold a queued parse that will never be finished (the ones in front of k were
                    # finished above and have moved the tables on): the range cannot be completed and the encoder's contexts are in
                    # mixed states.  Wait for the helper, then refuse every further use -- close() is what is left.
                    self._late.join()
                    self._broken = self._late_error
                    raise RuntimeError("a staggered parse could not be queued (context %d or later of this range); "
                                       "this RangeEncoder is unusable now: close it" % k) from self._broken
            s.set_state_device(d_state, level)
            n = s.finish_device(d_out + pos, cap - pos)
            level = s.get_state_device(d_state)
            segs.append((pos, n))
            pos = (pos + n + 3) & ~3
        if self.stagger == "auto" and hasattr(self.streams[0], "timings"):      # pace the next step by what this one measured
            st = self.stage_times()
            tot = sum(r for _, r, _ in st)
            if tot > 0.0:
                self.rank_ms_per_block = tot / sum(self.parts)
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

    def stage_times(self):
        """[(parse_ms, rank_ms, huffman_ms)] per context of the range, in stream order (what schedule_model takes).  The rank and
        Huffman stages are timed from the moment the context's finish was queued (`idle_before_finish` is what lay between the end
        of its parse and that moment: the contexts before it being ranked; it is not part of any stage)."""
        out = []
        for s in self.streams:
            t = dict(s.timings())
            parse = t.get("dict_reset", 0.0) + t.get("rolz_parse", 0.0)
            huff = sum(t.get(k, 0.0) for k in ("histogram", "huff_lengths", "layout_scan", "huff_pack"))
            rank = sum(t.get(k, 0.0) for k in ("lit_partition", "mtf_chain", "rank_replay", "mtf_rank"))
            out.append((parse, rank, huff))
        return out

    def close(self):
        t = getattr(self, "_late", None)
        if t is not None:
            t.join()                                       # never close a context the helper thread may still queue a parse on
        for s in self.streams:
            s.close()


def schedule_model(stages, parses_in_flight=2, stagger=None):
    """Wall time (ms) of one range from the measured stage times of its contexts, [(parse, rank, huffman)] in stream order, under
    the schedule RangeEncoder runs: the parse of context k starts when context k - parses_in_flight's has finished (0: all start at
    once); rank + Huffman of context k start when its parse AND the finish of context k - 1 are done (the literal tables travel in
    stream order).  This is the Amdahl arithmetic bench.py prints as `amdahl.model_ms`: it must reproduce `ms_per_step` when the
    stage times are right (tests/test_bench_host.py checks the arithmetic on hand-made cases)."""
    return schedule_model_ranks([stages], parses_in_flight, stagger)


def schedule_model_ranks(per_rank_stages, parses_in_flight=2, stagger=None):
    """The same for a stream sharded over ranks: every rank parses its own contexts on its own GPU (all ranks start together),
    and the finishes follow one another in STREAM order across the ranks -- rank r's first context waits for rank r - 1's last
    (the 64 KiB state hand-off).  per_rank_stages[r] = [(parse, rank, huffman)] of rank r's contexts, in MILLISECONDS.  stagger =
    (first, gap_ms): context k of a rank is queued (k - first + 1) * gap_ms after the step began (k >= first) instead of behind
    another parse; ("at", [ms per context]): explicit launch times.  (RangeEncoder takes the same shapes in SECONDS.)"""
    prev = 0.0
    for stages in per_rank_stages:
        parse_end = []
        for k, (parse, rank, huff) in enumerate(stages):
            if stagger and stagger[0] == "at":
                start = stagger[1][k]
            elif stagger:
                start = (k - stagger[0] + 1) * stagger[1] if k >= stagger[0] else 0.0
            else:
                start = parse_end[k - parses_in_flight] if (parses_in_flight > 0 and k >= parses_in_flight) else 0.0
            parse_end.append(start + parse)
            prev = max(parse_end[k], prev) + rank + huff
    return prev


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
