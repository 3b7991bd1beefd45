#!/usr/bin/env python3
"""would splitting K9 (k_rolz_replay, one wavefront per stream) into a RESOLVER wavefront (literals, MTF, ring
inserts, the context chain) and a COPIER wavefront (match copies from a queue) reach 1.4x?  A CPU model, like wg_parser_model.c
for the parser -- nothing here runs on the GPU; the costs are the measured ones of rounds 3-4 (LABNOTES section 6, "Decode").

What the resolver cannot give away: the context of token t + 1 is the LAST byte token t produced (src/libzling_lz.cpp:318-376), and
the word-MRU update behind a match reads its last three (:351-357).  So for every match the resolver still needs, on its one
dependent chain: the ring-head step of the context (LDS), the ring slot (memory), the source's tail bytes (LDS window or memory) --
i.e. every LATENCY of today's match path.  What it can give away is the copy's own instructions, and what it gains on top is a wait
whenever the source's tail was produced by a copy the copier has not performed yet.

The script measures, on the benchmark text and on the image's real text:
  * the token mix and the share of matches whose source's last byte was written 1, 2-3, 4-7 ... tokens earlier (the checker's
    decoder records the writing token of every byte: oracle/zlng_oracle.c zo_dstats.lag_hist) -> stalls for a queue of depth q;
  * from csrc/replay_loop.h (the generated loop): the instructions on the main match path and how many of them are the copy;
and prints the bound: speed-up <= today's time / (today's time - copy instructions x issue interval + stalls x resolve cost).
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle_py import Oracle, textgen  # noqa: E402

# measured, one lone wavefront on gfx950 (LABNOTES section 6 "Decode", scripts/ubench/issue.hip / latency.hip)
NS_LITERAL, NS_MATCH = 105.0, 335.0          # per token of the shipped loop on the benchmark text
NS_ISSUE = 2.6                               # per instruction of the match path (335 ns = 3 latencies of 83 / 54 / 54 ns + 55 instructions)
NS_RESOLVE_STALL = 120.0                     # a source tail still in the copier's queue: read the queue entry (LDS round trip) and
                                             # take the tail from ITS source instead (another window read)


def loop_counts():
    """(instructions on the near-match main path of set A, those that only serve the copy)."""
    text = open(os.path.join(ROOT, "libzling_amd", "csrc", "replay_loop.h")).read()
    ins = re.findall(r'"([^"\\]+)\\n"', text)
    a = ins.index("NOTLIT_A_%=:")
    b = next(i for i in range(a, len(ins)) if ins[i] == "s_branch LIT_B_go_%=")
    path = [s for s in ins[a:b + 1] if not s.endswith(":")]
    # the copy itself: opening EXEC to the match's lanes, the destination addresses, the window and output stores, closing EXEC
    copy = [s for s in path if s.startswith(("ds_write_b8 v47", "global_store_byte v40", "s_bfm_b64 exec", "v_add_u32 v40, s49", "v_and_or_b32 v47"))]
    # ... the two `s_mov_b64 exec, 1` that close those windows again
    copy_n = len(copy) + 2
    return len(path), copy_n, path


def corpus_real(n):
    files = []
    for base in ("/usr/include", "/usr/lib/python3/dist-packages", "/usr/share/doc"):
        for d, _, fs in os.walk(base):
            for f in sorted(fs):
                if f.endswith((".h", ".py", ".txt", ".md", ".rst")):
                    files.append(os.path.join(d, f))
    files.sort()
    out, tot = [], 0
    for f in files:
        try:
            b = np.fromfile(f, dtype=np.uint8)
        except Exception:
            continue
        out.append(b); tot += b.size
        if tot >= n:
            break
    return np.concatenate(out)[:n] if out else None


def main():
    o = Oracle()
    n_path, n_copy, _ = loop_counts()
    print("csrc/replay_loop.h, near match of at most 64 bytes (the common form): %d instructions on the main path, %d of them are the copy" % (n_path, n_copy))
    print("(literal path: 13 instructions + one LDS round trip on the chain; nothing of it can move)")
    for name, x in (("benchmark text (textgen), one 16 MiB block", textgen(1 << 24, 0)), ("real text of the image, 16 MiB", corpus_real(1 << 24))):
        if x is None or x.size < (1 << 20):
            continue
        z = o.encode(x, 0)
        rc, d = o.decode_stats(z, x.size, lag=True)
        assert rc == 0
        T, L, W, Mt = d["tokens"], d["literals"], d["words"], d["matches"]
        h = d["lag_hist"]
        print("\n== %s: %d tokens = %.1f %% literals, %.1f %% words, %.1f %% matches (mean %.1f B); %.2f B per token"
              % (name, T, 100.0 * L / T, 100.0 * W / T, 100.0 * Mt / T, d["match_bytes"] / max(Mt, 1), x.size / T))
        print("   matches whose source's LAST byte was written k tokens earlier: " +
              ", ".join("%s: %.2f %%" % (lab, 100.0 * sum(h[a:b]) / Mt) for lab, a, b in (("1", 0, 1), ("2-3", 1, 2), ("4-7", 2, 3), ("8-15", 3, 4), ("16-63", 4, 6))) +
              ", by this very copy (period shorter than the match): %.2f %%" % (100.0 * h[23] / Mt))
        today = L * NS_LITERAL + W * NS_LITERAL + Mt * NS_MATCH
        for q in (1, 2, 4, 8, 16):
            stalls = sum(h[: max(1, q.bit_length())]) if q > 1 else h[0]          # tails produced within the last q tokens
            split = today - Mt * n_copy * NS_ISSUE + stalls * NS_RESOLVE_STALL + Mt * 3 * NS_ISSUE      # + 3 instructions per match to queue {src, dst, len}
            print("   copier %2d matches behind: resolver %.1f ns per byte (today %.1f) -> %.3fx" % (q, split / x.size, today / x.size, today / split))
        ideal = today - Mt * n_copy * NS_ISSUE
        print("   ceiling (copy instructions free, no queueing cost, never a stall): %.3fx -- the three latencies of a match (ring-head step, ring slot, "
              "source tail) and the literal's LDS round trip all stay on the resolver's chain" % (today / ideal))
    print("\nverdict: far below the 1.4x bar for building it; the replay is bound by the dependent round trips of ONE chain per stream, "
          "which a second wavefront does not shorten.  What scales is streams: K contexts decode K streams side by side (one wavefront and 130 KiB of LDS each).")


if __name__ == "__main__":
    main()
