"""Generates libzling_amd/csrc/replay_loop.h: the token loop of K9 (k_rolz_replay, decode.hip) as gfx950 assembly.

    python scripts/gen_replay_asm.py        # rewrites the header; commit both

Why a generator: the loop is software-pipelined over two register sets (A / B).  While token t (set X) finishes, the
LDS reads of token t+1 (set N: ring-head step, and for a literal its two table entries) are already in flight, so the
one dependent chain of the replay -- last byte -> table row -> next byte -- carries only the LDS latency and ~13
instructions per literal; everything else (window / output / ring stores, the MRU pair, fetching token t+2) is issued
behind those reads.  The two sets differ only in register names, which is what this script substitutes.

Invariant on entry to a handler of set X: the scalars of token t (set X) and t+1 (set N) are decoded; the context
registers of X describe the bytes before token t; the ring-head step of X is issued, and for a literal its reads.
Token types are data-independent (they come from K8), so every branch is a scalar branch on them.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "libzling_amd", "csrc", "replay_loop.h")

SETS = {
    "A": dict(tok="s50", sym="s51", aux="s52", h="v24", b2a="v25", row="v26", rrow="v27", b1="v28", slot="v29",
              a1="v31", a2="v32", cc="v33", dd="v34", m="v35"),
    "B": dict(tok="s70", sym="s71", aux="s72", h="v54", b2a="v55", row="v56", rrow="v57", b1="v58", slot="v59",
              a1="v61", a2="v62", cc="v63", dd="v64", m="v65"),
}
OTHER = {"A": "B", "B": "A"}

DROP = set(filter(None, os.environ.get("REPLAY_DROP", "").split(",")))   # timing experiments only (wrong output)
lines = []      # main path
cold = []       # out-of-line parts
_uid = [0]


def emit(text, to=None):
    to = lines if to is None else to
    for ln in text.strip("\n").split("\n"):
        ln = ln.strip()
        if "stores" in DROP and ln.startswith("global_store") and "@lit" in ln:
            continue
        if "win" in DROP and "@win" in ln:
            continue
        if "mru" in DROP and "@mru" in ln:
            continue
        if "ring" in DROP and ln.startswith("global_load_dword v49"):
            to.append("v_max_u32 v49, 16, v37")
            ln = "v_subrev_u32 v49, 16, v49"
        if "dinv" in DROP and ln.startswith("s_dcache_inv"):
            continue
        if "glc" in DROP and ln.startswith("s_load_dword"):
            ln = ln.replace(" glc", "")
        if "vmwait" in DROP and "@vmwait" in ln:
            continue
        if "rstore" in DROP and "@rstore" in ln:
            continue
        if "inc" in DROP and ln.startswith("ds_inc_rtn_u32"):
            ln = "v_mov_b32 %s, 1" % ln.split()[1].rstrip(",")
        ln = ln.split(";")[0].strip()
        if ln:
            to.append(ln)


def fetch(S):
    """token (s47) -> scalars of set S; an index past the end yields the END symbol 0xffff"""
    _uid[0] += 1
    u = _uid[0]
    r = SETS[S]
    emit(f"""
        s_cmp_lt_u32 s47, s48
        s_cbranch_scc0 F_slow_{u}_%=
    F_rd_{u}_%=:
        v_readlane_b32 {r['tok']}, v22, s47
        s_add_u32 s47, s47, 1
        s_and_b32 {r['sym']}, {r['tok']}, 0xffff
        s_lshr_b32 {r['aux']}, {r['tok']}, 16
    F_done_{u}_%=:
    """)
    emit(f"""
    F_slow_{u}_%=:
        s_cmp_ge_u32 s47, s46
        s_cbranch_scc1 F_end_{u}_%=
        s_mov_b64 exec, s[60:61]
        s_waitcnt vmcnt(0)
        v_mov_b32 v22, v23
        s_add_u32 s48, s48, 64
        s_min_u32 s48, s48, s46
        s_add_u32 s56, s47, 64
        v_add_u32 v41, s56, v20
        v_mov_b32 v23, 0
        v_cmpx_gt_u32 vcc, s46, v41
        global_load_dword v23, v21, s[44:45]
        s_add_u32 s44, s44, 0x100
        s_addc_u32 s45, s45, 0
        s_mov_b64 exec, 1
        s_branch F_rd_{u}_%=
    F_end_{u}_%=:
        s_mov_b32 {r['sym']}, 0xffff
        s_mov_b32 {r['aux']}, 0
        s_branch F_done_{u}_%=
    """, cold)


def pre(S, nolit_label):
    """ring-head step of set S and, for a literal, its two table reads; falls through when S is a literal"""
    r = SETS[S]
    emit(f"""
        ds_inc_rtn_u32 {r['slot']}, {r['h']}, v38 offset:1024
        s_cmpk_lt_u32 {r['sym']}, 0x100
        s_cbranch_scc0 {nolit_label}
        v_add_u32 {r['a1']}, {r['sym']}, {r['row']}
        v_add_u32 {r['a2']}, {r['aux']}, {r['row']}
        ds_read_u8 {r['cc']}, {r['a1']}
        ds_read_u8 {r['dd']}, {r['a2']}
    """)


def book_lit(X):
    r, n = SETS[X], SETS[OTHER[X]]
    emit(f"""
        ds_read_b32 {r['m']}, {r['b2a']} ; @mru
        v_and_or_b32 v44, v37, v45, s68 ; @win
        ds_write_b8 v44, {r['cc']} ; @win
        global_store_byte v37, {r['cc']}, s[40:41] ; @lit
        v_lshl_add_u32 v30, {r['slot']}, 2, {r['rrow']}
        global_store_dword v30, v37, s[42:43] ; @lit
        v_lshl_or_b32 v36, {r['b1']}, 8, {r['cc']}
        v_mov_b32 {n['b2a']}, {r['h']}
        v_mov_b32 {n['b1']}, {r['cc']}
        v_lshlrev_b32 {n['rrow']}, 14, {r['cc']}
        s_add_u32 s49, s49, 1
        v_add_u32 v37, 1, v37
    """)
    fetch(X)
    emit(f"""
        s_waitcnt lgkmcnt(1)
        v_lshl_or_b32 {r['m']}, {r['m']}, 16, v36 ; @mru
        ds_write_b32 {r['b2a']}, {r['m']} ; @mru
    """)


def literal(X):
    N = OTHER[X]
    r, n = SETS[X], SETS[N]
    emit(f"""
    LIT_{X}_any_%=:
        s_waitcnt lgkmcnt(0)
    LIT_{X}_go_%=:
        ds_write_b8 {r['a1']}, {r['dd']}
        ds_write_b8 {r['a2']}, {r['cc']}
        v_lshl_add_u32 {n['h']}, {r['cc']}, 2, s62
        v_lshlrev_b32 {n['row']}, 8, {r['cc']}
    """)
    pre(N, f"LIT_{X}_nn_%=")
    book_lit(X)
    emit(f"s_branch LIT_{N}_go_%=")
    emit(f"LIT_{X}_nn_%=:")
    book_lit(X)
    emit(f"s_branch NOTLIT_{N}_%=")


def book_match(X):
    n = SETS[OTHER[X]]
    emit(f"""
        s_add_u32 s49, s49, s53
        v_mov_b32 v37, s49
        v_lshlrev_b32 {n['rrow']}, 14, {n['b1']}
        v_lshl_add_u32 {n['b2a']}, s59, 2, v50
        v_lshl_add_u32 v39, s64, 2, v50
        ds_read_b32 v40, v39
        s_lshl_b32 s56, s59, 8
        s_or_b32 s56, s56, s58
        s_waitcnt lgkmcnt(0)
        v_and_b32 v36, 0xffff, v40
        v_cmp_ne_u32 vcc, s56, v36
        v_lshl_or_b32 v43, v40, 16, s56
        s_nop 0
        v_cndmask_b32 v40, v40, v43, vcc
        ds_write_b32 v39, v40
    """)


def next_ctx_from_scalars(X):
    """context registers of the next set from the last bytes of a match (s58 = last byte)"""
    n = SETS[OTHER[X]]
    emit(f"""
        v_mov_b32 {n['b1']}, s58
        v_lshl_add_u32 {n['h']}, s58, 2, v50
        v_lshlrev_b32 {n['row']}, 8, {n['b1']}
    """)


def notlit(X):
    N = OTHER[X]
    r, n = SETS[X], SETS[N]
    # ---- match.  The ring slot is read with a SCALAR load: a vector load would sit in the same in-order queue as the output
    # and ring stores in flight (s_waitcnt vmcnt counts both), and waiting for those costs more than the load itself.  The
    # scalar cache is not coherent with vector stores, hence s_dcache_inv; the slot read was stored `index` tokens ago and
    # every token since issued at least one vector store, so "at most index - 1 vector operations outstanding" proves that
    # store has landed (s_waitcnt vmcnt is in order): three classes of the index, the common one (>= 17) hardly ever waits.
    # Index 0 names the token itself (GetMatchAndUpdate inserts before it looks up, src/libzling_lz.cpp:388-399; the reference
    # would copy the match onto itself, this decoder has always rejected it: "src >= pos" in k_rolz_decode); the END symbol
    # of the look-ahead carries index 0 as well, so both are told apart off the main path.
    emit(f"""
    NOTLIT_{X}_%=:
        v_readfirstlane_b32 s57, {r['b1']}
        s_waitcnt lgkmcnt(0)
        s_cmpk_lt_u32 {r['sym']}, 0x102
        s_cbranch_scc1 WORD_{X}_%=
        v_readfirstlane_b32 s56, {r['slot']}
        s_cmpk_lt_u32 {r['aux']}, 17
        s_cbranch_scc1 NEAR_{X}_%=
        s_waitcnt vmcnt(16)
    RL_{X}_%=:
        s_sub_u32 s56, s56, {r['aux']}
        s_and_b32 s56, s56, 0xfff
        s_lshl_b32 s57, s57, 12
        s_or_b32 s56, s56, s57
        s_lshl_b32 s56, s56, 2
        s_dcache_inv
        s_load_dword s54, s[42:43], s56
        v_lshl_add_u32 v30, {r['slot']}, 2, {r['rrow']}
        global_store_dword v30, v37, s[42:43]
        s_sub_u32 s53, {r['sym']}, 254
        s_sub_u32 s57, s53, 1
        s_sub_u32 s66, s53, 2
        s_sub_u32 s67, s53, 3
        s_mov_b64 exec, s[60:61]
        v_add_u32 v40, s49, v20
        v_and_or_b32 v47, v40, v45, s68
        s_mov_b64 exec, 1
    """)
    emit(f"""
    NEAR_{X}_%=:
        s_cmpk_lt_u32 {r['aux']}, 5
        s_cbranch_scc1 NEAR0_{X}_%=
        s_waitcnt vmcnt(4)
        s_branch RL_{X}_%=
    NEAR0_{X}_%=:
        s_cmpk_eq_u32 {r['sym']}, 0xffff
        s_cbranch_scc1 DONE_{X}_%=
        s_cmp_eq_u32 {r['aux']}, 0
        s_cbranch_scc1 ERR_%=
        s_waitcnt vmcnt(0)
        s_branch RL_{X}_%=
    """, cold)
    fetch(X)
    emit(f"""
        s_cmpk_gt_u32 s53, 63
        s_cbranch_scc1 BIG_{X}_%=
        s_waitcnt lgkmcnt(0)
        s_sub_u32 s55, s49, s54
        s_cmp_lt_u32 s55, s53
        s_cbranch_scc1 OVL_{X}_%=
        s_cmpk_gt_u32 s55, 0xfe00
        s_cbranch_scc1 FAR0_{X}_%=
        s_bfm_b64 exec, s53, 0
        v_add_u32 v39, s54, v20
        v_and_or_b32 v39, v39, v45, s68
        ds_read_u8 v42, v39
        s_waitcnt lgkmcnt(0)
        v_readlane_b32 s58, v42, s57
        v_readlane_b32 s59, v42, s66
        v_readlane_b32 s64, v42, s67
        s_mov_b64 exec, 1
    """)
    next_ctx_from_scalars(X)
    for lit in (True, False):
        if lit:
            pre(N, f"FT_{X}_nn_%=")
        else:
            emit(f"FT_{X}_nn_%=:")
        emit(f"""
        s_bfm_b64 exec, s53, 0
        ds_write_b8 v47, v42
        global_store_byte v40, v42, s[40:41]
        s_mov_b64 exec, 1
        """)
        book_match(X)
        emit(f"s_branch LIT_{N}_go_%=" if lit else f"s_branch NOTLIT_{N}_%=")

    # ---- the other copy forms (rare on text): overlapping source (a period shorter than the match), source beyond the
    # window, matches longer than one wavefront
    emit(f"""
    BIG_{X}_%=:
        s_waitcnt lgkmcnt(0)
        s_sub_u32 s55, s49, s54
        s_cmp_lt_u32 s55, s53
        s_cbranch_scc0 GEN_{X}_%=
    OVL_{X}_%=:
        s_cmp_ge_u32 s55, 64
        s_cbranch_scc1 GEN_{X}_%=
        s_mov_b64 exec, s[60:61]
        v_cmpx_gt_u32 vcc, s55, v20
        v_add_u32 v39, s54, v20
        v_and_or_b32 v39, v39, v45, s68
        ds_read_u8 v42, v39
        s_mov_b32 s56, 0
        s_waitcnt lgkmcnt(0)
    PER_{X}_%=:
        s_sub_u32 s57, s53, s56
        s_min_u32 s57, s57, s55
        s_add_u32 s58, s49, s56
        s_mov_b64 exec, s[60:61]
        v_cmpx_gt_u32 vcc, s57, v20
        v_add_u32 v40, s58, v20
        v_and_or_b32 v47, v40, v45, s68
        ds_write_b8 v47, v42
        global_store_byte v40, v42, s[40:41]
        s_add_u32 s56, s56, s55
        s_cmp_lt_u32 s56, s53
        s_cbranch_scc1 PER_{X}_%=
        s_sub_u32 s56, s56, s55
        s_sub_u32 s57, s53, 1
        s_sub_u32 s57, s57, s56
        s_sub_u32 s65, s55, 1
        s_sub_u32 s66, s57, 1
        s_cmp_eq_u32 s57, 0
        s_cselect_b32 s66, s65, s66
        s_sub_u32 s67, s66, 1
        s_cmp_eq_u32 s66, 0
        s_cselect_b32 s67, s65, s67
        s_branch LAST3_{X}_%=
    FAR0_{X}_%=:
        s_mov_b32 s56, 0
        s_mov_b64 exec, s[60:61]
        s_branch FAR_{X}_%=
    GEN_{X}_%=:
        s_mov_b32 s56, 0
        s_mov_b64 exec, s[60:61]
        s_cmpk_gt_u32 s55, 0xfe00
        s_cbranch_scc1 FAR_{X}_%=
    CP_{X}_%=:
        v_add_u32 v41, s56, v20
        v_cmpx_gt_u32 vcc, s53, v41
        v_add_u32 v39, s54, v41
        v_add_u32 v40, s49, v41
        v_and_or_b32 v39, v39, v45, s68
        v_and_or_b32 v47, v40, v45, s68
        ds_read_u8 v42, v39
        s_waitcnt lgkmcnt(0)
        ds_write_b8 v47, v42
        global_store_byte v40, v42, s[40:41]
        s_mov_b64 exec, s[60:61]
        s_add_u32 s56, s56, 64
        s_cmp_lt_u32 s56, s53
        s_cbranch_scc1 CP_{X}_%=
        s_branch LAST3_{X}_%=
    FAR_{X}_%=:
        v_add_u32 v41, s56, v20
        v_cmpx_gt_u32 vcc, s53, v41
        v_add_u32 v39, s54, v41
        v_add_u32 v40, s49, v41
        global_load_ubyte v42, v39, s[40:41]
        v_and_or_b32 v47, v40, v45, s68
        s_waitcnt vmcnt(0)
        ds_write_b8 v47, v42
        global_store_byte v40, v42, s[40:41]
        s_mov_b64 exec, s[60:61]
        s_add_u32 s56, s56, 64
        s_cmp_lt_u32 s56, s53
        s_cbranch_scc1 FAR_{X}_%=
    LAST3_{X}_%=:
        v_readlane_b32 s58, v42, s57
        v_readlane_b32 s59, v42, s66
        v_readlane_b32 s64, v42, s67
        s_mov_b64 exec, 1
    """)
    next_ctx_from_scalars(X)
    pre(N, f"M_{X}_nn_%=")
    book_match(X)
    emit(f"s_branch LIT_{N}_go_%=")
    emit(f"M_{X}_nn_%=:")
    book_match(X)
    emit(f"s_branch NOTLIT_{N}_%=")

    # word MRU slot 0 / 1 (rare): nothing overlapped
    emit(f"""
    WORD_{X}_%=:
        ds_read_b32 {r['m']}, {r['h']}
        v_and_or_b32 v44, v37, v45, s68
        v_add_u32 v46, 1, v37
        v_and_or_b32 v46, v46, v45, s68
        v_lshl_add_u32 v30, {r['slot']}, 2, {r['rrow']}
        global_store_dword v30, v37, s[42:43]
        s_waitcnt lgkmcnt(0)
        s_cmpk_eq_u32 {r['sym']}, 0x100
        s_cbranch_scc1 W0_{X}_%=
        v_alignbit_b32 {r['m']}, {r['m']}, {r['m']}, 16
        ds_write_b32 {r['h']}, {r['m']}
    W0_{X}_%=:
    """)
    fetch(X)
    emit(f"""
        v_bfe_u32 {r['cc']}, {r['m']}, 8, 8
        v_and_b32 {r['dd']}, 0xff, {r['m']}
        ds_write_b8 v44, {r['cc']}
        ds_write_b8 v46, {r['dd']}
        global_store_byte v37, {r['cc']}, s[40:41]
        global_store_byte v37, {r['dd']}, s[40:41] offset:1
        v_lshl_add_u32 {n['b2a']}, {r['cc']}, 2, s62
        v_mov_b32 {n['b1']}, {r['dd']}
        v_lshl_add_u32 {n['h']}, {r['dd']}, 2, s62
        v_lshlrev_b32 {n['row']}, 8, {r['dd']}
        v_lshlrev_b32 {n['rrow']}, 14, {r['dd']}
        s_add_u32 s49, s49, 2
        v_add_u32 v37, 2, v37
    """)
    pre(N, f"NOTLIT_{N}_%=")
    emit(f"s_branch LIT_{N}_any_%=")

    # end of the sub-block: the look-ahead stepped the ring head of a token that does not exist -- step it back
    emit(f"""
    DONE_{X}_%=:
        ds_dec_u32 {r['h']}, v38 offset:1024
        v_mov_b32 v28, {r['b1']}
        v_mov_b32 v25, {r['b2a']}
        s_branch EXIT_%=
    """)


def generate():
    a = SETS["A"]
    emit(f"""
        s_mov_b64 s[60:61], exec
        s_mov_b64 s[40:41], %[out]
        s_mov_b64 s[42:43], %[ring]
        s_mov_b64 s[44:45], %[tok]
        s_mov_b32 s46, %[nt]
        s_mov_b32 s47, %[ti]
        s_mov_b32 s49, %[opos]
        s_mov_b32 s62, 0x20000
        s_mov_b32 s68, 0x10000
        s_mov_b32 s63, 0
        v_mov_b32 v20, %[lane]
        v_mov_b32 {a['b1']}, %[b1]
        v_mov_b32 {a['b2a']}, %[b2]
        v_lshlrev_b32 v21, 2, v20
        v_mov_b32 v38, 0xfff
        v_mov_b32 v45, 0xffff
        v_mov_b32 v50, 0x20000
        v_mov_b32 v22, 0
        v_mov_b32 v23, 0
        v_add_u32 v41, 64, v20
        v_cmpx_gt_u32 vcc, s46, v20
        global_load_dword v22, v21, s[44:45]
        s_mov_b64 exec, s[60:61]
        v_cmpx_gt_u32 vcc, s46, v41
        global_load_dword v23, v21, s[44:45] offset:256
        s_mov_b64 exec, s[60:61]
        s_add_u32 s44, s44, 0x200
        s_addc_u32 s45, s45, 0
        s_min_u32 s48, s46, 64
        v_lshl_add_u32 {a['h']}, {a['b1']}, 2, s62
        v_lshlrev_b32 {a['row']}, 8, {a['b1']}
        v_lshlrev_b32 {a['rrow']}, 14, {a['b1']}
        v_lshl_add_u32 {a['b2a']}, {a['b2a']}, 2, s62
        v_mov_b32 v37, s49
        s_waitcnt vmcnt(0)
        s_mov_b64 exec, 1
    """)
    fetch("A")
    fetch("B")
    pre("A", "NOTLIT_A_%=")
    emit("s_branch LIT_A_any_%=")
    for X in "AB":
        literal(X)
        notlit(X)
    emit("""
    ERR_%=:
        s_mov_b32 s63, 1
    EXIT_%=:
        s_mov_b64 exec, s[60:61]
        s_waitcnt vmcnt(0) lgkmcnt(0)
        v_subrev_u32 v25, s62, v25
        v_lshrrev_b32 v25, 2, v25
        s_nop 1
        v_readfirstlane_b32 %[o_b1], v28
        v_readfirstlane_b32 %[o_b2], v25
        s_mov_b32 %[o_opos], s49
        s_mov_b32 %[o_err], s63
        s_branch END_%=
    """)
    lines.extend(cold)
    emit("END_%=:")


def render():
    generate()
    body = " \\\n".join('    "%s\\n"' % ln for ln in lines)
    return ("// replay_loop.h -- GENERATED by scripts/gen_replay_asm.py (do not edit; edit the generator and run it).\n"
            "// The software-pipelined token loop of k_rolz_replay (decode.hip), which documents the registers and the LDS layout.\n"
            "#pragma once\n#define ZLNG_REPLAY_LOOP_ASM \\\n" + body + "\n")


def main():
    text = render()
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, len(lines), "lines")


if __name__ == "__main__":
    main()
