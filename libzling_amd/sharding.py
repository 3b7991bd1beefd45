"""Block-range sharding of ONE .zlng stream across ranks.

The parse of a 16 MiB block depends only on that block (dictionary reset per block,
src/libzling.cpp:197); the literal ranks depend on the 256 MTF tables, which the reference keeps
for the whole stream (src/libzling_lz.cpp:197-209 does not reset them).  So a stream shards as

    rank r:  parse(range r)            -- all ranks at once, no communication
             recv state from r-1       -- 65,536 B of MTF tables + current_level   (the one exchange)
             rank + Huffman + frame    -- per rank
             send state to r+1

and the rank outputs concatenate to exactly the single-device stream.  `run_handoff` drives any
object with the Stream interface of libzling_amd (parse / set_state / finish / get_state), over any
torch.distributed backend: RCCL ("nccl") on GPUs, gloo in the CPU tests.
"""
import numpy as np

BLOCK = 16777216


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


def run_handoff(stream, rank, world, dist, state_buf, initial_state, initial_level, level,
                parse, finish, state_to_buf, buf_to_state):
    """One sharded step.  `parse()` starts this rank's parse; `finish()` runs rank+Huffman and returns
    the byte count; `buf_to_state(buf, level)` / `state_to_buf(buf)` move the MTF tables between the
    stream and the communication buffer `state_buf` (a tensor on the backend's device)."""
    parse()
    if rank == 0:
        stream.set_state(initial_state, initial_level)
    else:
        dist.recv(state_buf, src=rank - 1)
        buf_to_state(state_buf, int(state_buf[-1].item()) if state_buf.numel() > 65536 else level)
    n = finish()
    if rank < world - 1:
        lv = state_to_buf(state_buf)
        if state_buf.numel() > 65536:
            state_buf[-1] = lv
        dist.send(state_buf, dst=rank + 1)
    return n
