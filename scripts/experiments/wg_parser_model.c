/*
 * wg_parser_model.c -- CPU model of the workgroup-wide ROLZ parser (libzling_amd/csrc/rolz_wg.hip).
 *
 * TEST / DESIGN INFRASTRUCTURE (not product code): it includes the oracle's source to reuse its dictionary and
 * exact match functions, runs the window algorithm of the kernel lane by lane in plain loops, and compares every
 * sub-block's tokens with the oracle's parse (zo_parse_subblock = EncodeImpl, src/libzling_lz.cpp:139-195).
 * It exists to validate the exactness rules of the window algorithm (what a lane may decide in parallel, what makes
 * it "hard") and to predict rounds / iterations / hard lanes per window size before a GPU is involved.
 *
 *   gcc -O2 -o /tmp/wgm scripts/experiments/wg_parser_model.c && /tmp/wgm FILE [NL=256] [level=0] [fix=1] [max_bytes]
 *
 * The window algorithm (one round, NL lanes, lane g <-> position P + g, P = the next token start):
 *   phase 1   every lane evaluates its position AS IF it were a token start against the dictionary as of the start
 *             of the round (read only): hash head, <= depth chain nodes, longest match, lazy probes.
 *   iterate   S = the token starts reached from lane 0 under the current per-lane token lengths;
 *             E(g, S): every lane re-evaluates its token given the accepted starts before it:
 *               - word-MRU outcome: the two MRU slots of its context after every boundary event of S up to g
 *                 (events and their effectiveness follow from the token types of S, all lane-parallel);
 *               - its match stands unless an accepted earlier start wrote what it read (same (ctx, hash13), a ring
 *                 slot it visited, or a lazy probe's read set): with fix = 1 same-key and lazy-key conflicts at
 *                 level 0 are evaluated exactly from the window's text, everything else makes the lane HARD;
 *             until no lane of S changed.  The first hard lane of S (or the sub-block's end) cuts the round.
 *   commit    S below the cut: dictionary inserts, token words, MRU slots.  A hard lane is then replayed by the exact
 *             serial code (MatchAndUpdate as written).
 * At the fixed point every lane of S was evaluated under exactly the accepted starts the reference has made by then,
 * so the result is the reference's; each iteration fixes at least the first wrong lane.
 */
#include "../../oracle/zlng_oracle.c"
#include <stdio.h>

enum { TY_NONE = 0, TY_LIT = 1, TY_W0 = 2, TY_W1 = 3, TY_MATCH = 4 };
#define MAXNL 2048

typedef struct {
    int live, canm, pos;
    uint32_t ctx, hc, chk, key, cw, ek, ew, b0;
    /* speculation */
    int sp_match, sp_len, sp_node, node0, head0, dmin, sp_veto;
    uint32_t ov0;
    uint32_t lkey[2]; int lctx[2], lrisk[2], lwant[2], ld[2];
    int ldfull[2];                     /* lazy_fix: min ring distance over the first lazy-depth nodes of the probe's chain, whatever vetoes */
    int d0, d1, has1;
    int pl[3], pn[3];      /* generic levels: best (len, node) over the first depth-1 / depth-2 chain nodes ([1], [2]; [0] = all) */
    int vpos[2], lhit2[2]; /* generic levels: index of the first snapshot node that vetoes probe j (its depth if none) */
    int len0;              /* level 0: candidate length of chain node 0 alone (0 if its check byte differs) */
    int has0;
    int lsrc1_valid; uint32_t lsrc1;   /* level 0: the lazy probe's chain head (snapshot): source offset */
    /* iteration state */
    int ty, tlen, mlen, mnode_slot;   /* mnode_slot: ring slot of the match source; -1 - lane if it is an in-window lane */
    int link_lane;                    /* in-window predecessor in the hash slot (fix), -1 = snapshot node0 */
    int ty2, tlen2, mlen2, mnode2, link2, hard, hardcls;
    int has_ev, cond, eff; uint32_t s0b;
    int slot;                         /* ring slot the committed token got (ghosts) */
    /* generic levels, ring_fix: ring distance of chain node i and the best (len, node) over the nodes in front of it */
    int nn; int nd[17], bl[17], bn[17];
} lane_t;
#define LNONE (-(1 << 20))             /* link_lane: no in-round predecessor (lane indices run from -ng: ghosts are negative) */
#define LANE_REF(a) (-1 - ((a) + MAXG)) /* match source / chain link that is a lane of the round (or a ghost), as a negative "slot" */
#define REF_LANE(m) (-1 - (m) - MAXG)

static struct {
    long rounds, tokens, iters, hard[8], serial, committed, cut_rounds, fixes, lfixes, maxit;
    long hist_it[16];
    long prefix_rounds;
    long ringfixes, lazyfixes2, lz_why[4];
    long open_lanes, open_rounds, open_tok, long_fix_evals, maxtail_sum;   /* long-match statistics (level 0): lanes / tokens with a match of 16 bytes or more */
    long windows, sync_rounds, first_rounds, ghost_sum, ghost_max, ghost_over, ghost_a1, spec_lanes, lanes_sum;
} st;

static int precise_risk = 1, only_s = 0, max_tok = 1 << 30, guess_mru = 0;
/* Round 5: the PIPELINED form, modelled before anything is built.  grid = 1: windows lie on a fixed grid of NL
 * positions (a round takes the grid window that holds the next token start, from that start on), so that the NEXT window's
 * positions are known while this one is being resolved.  stale = 1: phase 1 of window w + 1 is evaluated against the dictionary as
 * it is BEFORE the first commit of window w (what a second set of wavefronts would do beside the iterate phase of window w), and
 * every token committed since -- the rest of window w, serial replays included -- is a GHOST: a member of S in front of lane 0 that
 * is final, is never re-evaluated and takes part in nothing but the dictionary relations (same hash slot, ring slots handed out in
 * its bucket, lazy-probe keys).  More than ghost_cap ghosts (a two-word rank mask holds 128), or a window that was not the one
 * speculated (a long match jumped over it), falls back to a synchronous phase 1 ("sync round").  Exactness is checked like before:
 * every token against the oracle's, and the commit's head / link formulas against the dictionary. */
static int grid = 0, stale = 0, ghost_cap = 128;
/* ring_fix = 1 (levels 1-4, candidate for round 6): a chain node whose slot a token of this round has rewritten ENDS the walk in front of
 * it (the reference then reads a later position there and its chain-end test stops, src/libzling_lz.cpp:265) -- exact from the running
 * best phase 1 recorded per node, instead of a hard token.  Node 0 rewritten stays hard. */
static int ring_fix = 0;
/* lazy_fix = 1 (levels 1-4, candidate, model only): a token whose match length is no longer the speculation's (a token of the round is its
 * chain's head, or the ring rule cut its walk) has its lazy probes walked AGAIN under the new length -- the newest <= 2 tokens of the round with
 * the probe's key, then the first lazy-depth - h nodes of the snapshot's probe chain (src/libzling_lz.cpp:291-316) -- instead of going hard;
 * valid while no ring slot among the first lazy-depth nodes of that chain has been handed out again (ldfull). */
static int lazy_fix = 0;
/* grid = 2: FLOATING windows (a round starts at the next token start, as the kernel's do) with stale = 1: while a round is resolved the
 * positions [P + ahead_c, P + ahead_c + ahead_r) are evaluated ahead; the next round uses them where it lies inside that range and is
 * cut short where the range ends; it is a sync round when it starts in front of the range. */
static int ahead_c = 176, ahead_r = 336;
#define MAXG 512
static int prefix_pct = 0;   /* > 0: after the FIRST iteration, if the first changed token has at least prefix_pct % of the round's tokens in front of it,
                                commit that prefix and start the next round at the changed token instead of iterating (round 4 experiment) */
static int ring_dist(int node, int head0) { return (node - head0 - 1) & (ZO_RING - 1); }

/* Read-only evaluation of `pos` as a token start against the dictionary as it is now (phase 1). */
static void speculate(const zo_stream* s, const uint8_t* buf, int pos, int depth, int lazy1, int lazy2, int risk_dist, lane_t* o) {
    uint32_t h = hash4(buf + pos), chk = (h / ZO_HASH) % 256, hc = h % ZO_HASH;
    const zo_bucket* b = &s->bucket[buf[pos - 1]];
    int node = b->hash[hc], head0 = b->head, dmin = ZO_RING - 1;
    o->node0 = node; o->head0 = head0; o->ov0 = node != 65535 ? b->offset[node] : 0; o->has0 = node != 65535;
    o->len0 = 0; o->d0 = o->d1 = ZO_RING - 1; o->has1 = 0; o->nn = 0;
    int maxlen = ZO_MATCH_MIN - 1, maxnode = 0;
    int set1 = 0, set2 = 0;
    o->pl[1] = o->pl[2] = ZO_MATCH_MIN - 1; o->pn[1] = o->pn[2] = 0;
    if (depth - 1 <= 0) set1 = 1;
    if (depth - 2 <= 0) set2 = 1;
    if (node != 65535) {
        for (int i = 0; i < depth; i++) {
            if (i == depth - 1 && !set1) { o->pl[1] = maxlen; o->pn[1] = maxnode; set1 = 1; }    /* before node i: the first i nodes are in */
            if (i == depth - 2 && !set2) { o->pl[2] = maxlen; o->pn[2] = maxnode; set2 = 1; }
            int d = ring_dist(node, head0); if (d < dmin) dmin = d;
            if (i == 0) o->d0 = d;
            o->nd[i] = d; o->bl[i] = maxlen; o->bn[i] = maxnode; o->nn = i + 1;
            uint32_t off = b->offset[node] & 0xFFFFFF;
            if ((b->offset[node] >> 24) == chk && buf[pos + maxlen] == buf[off + maxlen]) {
                int len = common_len(buf + pos, buf + off);
                if (len > maxlen) { maxnode = node; maxlen = len; }
            }
            if (i == 0) o->len0 = ((b->offset[node] >> 24) == chk) ? common_len(buf + pos, buf + off) : 0;
            if (maxlen == ZO_MATCH_MAX) break;
            int nx = b->suffix[node];
            if (nx == 65535) break;
            d = ring_dist(nx, head0); if (d < dmin) dmin = d;        /* its offset is read for the chain-end test */
            if (i == 0) { o->d1 = d; o->has1 = 1; }
            if (off <= (b->offset[nx] & 0xFFFFFF)) break;
            node = nx;
        }
    }
    if (!set1) { o->pl[1] = maxlen; o->pn[1] = maxnode; }       /* the walk ended before that many nodes */
    if (!set2) { o->pl[2] = maxlen; o->pn[2] = maxnode; }
    o->pl[0] = maxlen; o->pn[0] = maxnode;
    o->bl[o->nn] = maxlen; o->bn[o->nn] = maxnode;
    o->dmin = dmin; o->sp_len = maxlen; o->sp_node = maxnode;
    int veto = 0;
    const int lz = maxlen >= ZO_MATCH_MIN && maxlen < ZO_LAZY_LIMIT;
    const int ldepth[2] = {lazy1, lazy2};
    for (int j = 0; j < 2; j++) {
        o->lwant[j] = ldepth[j] > 0; o->lrisk[j] = 0; o->lkey[j] = 0; o->lctx[j] = 0;
        if (!ldepth[j]) continue;
        const int pp = pos + 1 + j;
        const zo_bucket* lb = &s->bucket[buf[pp - 1]];
        uint32_t hh = hash4(buf + pp) % ZO_HASH;
        o->lctx[j] = buf[pp - 1]; o->lkey[j] = (uint32_t)buf[pp - 1] << 13 | hh;
        int n = lb->hash[hh], ld = ZO_RING - 1;
        if (j == 0) { o->lsrc1_valid = n != 65535; o->lsrc1 = n != 65535 ? (lb->offset[n] & 0xFFFFFF) : 0; }
        o->vpos[j] = ldepth[j];
        if (n != 65535 && lz) {
            int m = maxlen - 3;
            for (int i = 0; i < ldepth[j]; i++) {
                int d = ring_dist(n, lb->head); if (d < ld) ld = d;
                uint32_t off = lb->offset[n] & 0xFFFFFF;
                if (le32(buf + pp + m) == le32(buf + off + m)) { if (!(j == 1 && veto)) {} o->vpos[j] = i; if (j == 0 || !veto) veto = 1; break; }
                int nx = lb->suffix[n];
                if (nx == 65535) break;
                d = ring_dist(nx, lb->head); if (d < ld) ld = d;
                if (off <= (lb->offset[nx] & 0xFFFFFF)) break;
                n = nx;
            }
        } else if (n != 65535) { int d = ring_dist(n, lb->head); if (d < ld) ld = d; }   /* (level-0 fix may need the probe later) */
        o->lrisk[j] = ld < risk_dist; o->ld[j] = ld;
        {   /* lazy_fix: the same walk without the veto's early exit */
            int n2 = lb->hash[hh], lf = ZO_RING - 1;
            for (int i = 0; n2 != 65535 && i < ldepth[j]; i++) {
                int d = ring_dist(n2, lb->head); if (d < lf) lf = d;
                uint32_t off = lb->offset[n2] & 0xFFFFFF;
                int nx = lb->suffix[n2];
                if (nx == 65535) break;
                d = ring_dist(nx, lb->head); if (d < lf) lf = d;
                if (off <= (lb->offset[nx] & 0xFFFFFF)) break;
                n2 = nx;
            }
            o->ldfull[j] = lf;
        }
    }
    o->sp_veto = veto;
    o->sp_match = maxlen >= ZO_MATCH_MIN && !(lz && veto);
}

/* the exact serial token of EncodeImpl at *ipos (used for hard lanes) */
static uint32_t serial_token(zo_stream* s, const uint8_t* ibuf, int ilen, int* ipos_, int* opos_, uint16_t mru[256][2],
                             int depth, int lazy1, int lazy2, int* ty_out) {
    int ipos = *ipos_, midx, mlen;
    uint32_t word;
    if (ipos + ZO_SENTINEL < ilen && match_and_update(s, ibuf, ipos, depth, lazy1, lazy2, &midx, &mlen)) {
        word = (uint32_t)(258 + mlen - ZO_MATCH_MIN) | (uint32_t)midx << 16;
        *opos_ += 2; ipos += mlen;
        uint16_t w = (uint16_t)(ibuf[ipos - 2] << 8 | ibuf[ipos - 1]);
        uint16_t* m = mru[ibuf[ipos - 3]];
        if (m[0] != w) { m[1] = m[0]; m[0] = w; }
        *ty_out = TY_MATCH;
    } else {
        int done = 0;
        word = 0;
        if (ipos + 1 < ilen) {
            uint16_t w = (uint16_t)(ibuf[ipos] << 8 | ibuf[ipos + 1]);
            uint16_t* m = mru[ibuf[ipos - 1]];
            if (m[0] == w) { word = 256; (*opos_)++; ipos += 2; done = 1; *ty_out = TY_W0; }
            else if (m[1] == w) { word = 257; (*opos_)++; ipos += 2; m[1] = m[0]; m[0] = w; done = 1; *ty_out = TY_W1; }
        }
        if (!done) {
            word = (uint32_t)ibuf[ipos] | (uint32_t)ibuf[ipos - 1] << 16;
            (*opos_)++; ipos++;
            uint16_t* m = mru[ibuf[ipos - 3]];
            m[1] = m[0]; m[0] = (uint16_t)(ibuf[ipos - 2] << 8 | ibuf[ipos - 1]);
            *ty_out = TY_LIT;
        }
    }
    *ipos_ = ipos;
    return word;
}

/* lane <-> position: everything that depends on the text only, then phase 1 against the dictionary as it is now */
static void setup_lane(const zo_stream* s, const uint8_t* ibuf, int ilen, int pos, int depth, int lazy1, int lazy2, int NL, lane_t* l) {
    memset(l, 0, sizeof *l);
    l->pos = pos; l->live = 1; l->canm = pos + ZO_SENTINEL < ilen;
    l->ctx = ibuf[pos - 1];
    uint32_t w4 = 0; { uint8_t t[4] = {0, 0, 0, 0}; for (int k = 0; k < 4 && pos + k < ilen + 0; k++) t[k] = ibuf[pos + k]; memcpy(&w4, t, 4); }
    uint32_t h = w4 + ((w4 >> 16) & 0xFF) * 137u + (w4 >> 24) * 13337u;
    l->hc = h % ZO_HASH; l->chk = (h / ZO_HASH) % 256; l->key = l->ctx << 13 | l->hc;
    l->b0 = w4 & 0xFF;
    l->cw = (w4 & 0xFF) << 8 | ((w4 >> 8) & 0xFF);
    l->ek = pos >= 3 ? ibuf[pos - 3] : 0;
    l->ew = (uint32_t)(pos >= 2 ? ibuf[pos - 2] : 0) << 8 | ibuf[pos - 1];
    if (l->canm) speculate(s, ibuf, pos, depth, lazy1, lazy2, NL, l);
    else { l->sp_match = 0; l->sp_len = 3; l->dmin = ZO_RING - 1; }
}

/* NOTE on MRU bookkeeping: the model keeps `mru` in the reference's convention (pushes applied right after a token).
 * The window algorithm attaches the push to the NEXT token start ("boundary event"); the state the window sees at its
 * start, mru0, is the reference state with the push that follows the last token still pending, described by prevty. */

static int parse_block_model(zo_stream* s, const uint8_t* ibuf, int ilen, int level, int NL, int fix, uint32_t* tok_out, int* cuts, int* ncut) {
    const int depth = k_level_cfg[level][0], lazy1 = k_level_cfg[level][1], lazy2 = k_level_cfg[level][2];
    static lane_t Lbuf[MAXG + MAXNL];
    static uint8_t Sbuf[MAXG + MAXNL + 300];
    lane_t* const L = Lbuf + MAXG;                 /* L[-1] is the newest ghost, L[-ng] the oldest */
    uint8_t* const S = Sbuf + MAXG;
    static lane_t SP[MAXNL], SPcur[MAXNL];         /* stale mode: phase 1 of the next grid window; of the current one (its follow-up rounds) */
    long cur_log = 0; int cur_valid = 0, sp_pos = 0;
    static lane_t* clog = NULL;                    /* every committed token of the block, in order (ghost source) */
    if (!clog) clog = (lane_t*)malloc(sizeof(lane_t) * (ZO_BLOCK_IN + 64));
    long nlog = 0, sp_log = 0;                     /* sp_log: length of the log when SP was taken */
    int sp_win = -1, sp_n = 0, cur_win = -1, ng = 0;
    static int slot_buf[MAXG + MAXNL];
    int* const slot_of = slot_buf + MAXG;
    int q = 0, nt = 0, nsub = 0;
    zo_reset_buckets(s);
    while (q < ilen) {
        uint16_t mru[256][2];            /* window convention: state BEFORE the pending boundary event of prevty */
        memset(mru, 0, sizeof mru);
        int opos = 0, prevty = TY_NONE;
        const int tok_begin = nt;
        if (q == 0) { tok_out[nt++] = ibuf[0] | ZO_TOK_RAWCTX << 16; q = 1; opos = 1;
                      if (ilen > 1) { tok_out[nt++] = ibuf[1] | ZO_TOK_RAWCTX << 16; q = 2; opos = 2; } }
        int force_serial = 0;
        while (q < ilen && opos + 1 < ZO_SUBBLOCK_SYMS) {
            if (force_serial) {
                /* exact serial token; the pending boundary event is applied first (window convention -> reference convention) */
                force_serial = 0;
                if (prevty != TY_NONE && prevty != TY_W0) {
                    const uint32_t ek = ibuf[q - 3]; const uint16_t ew = (uint16_t)(ibuf[q - 2] << 8 | ibuf[q - 1]);
                    if (prevty == TY_MATCH) { if (mru[ek][0] != ew) { mru[ek][1] = mru[ek][0]; mru[ek][0] = ew; } }
                    else { mru[ek][1] = mru[ek][0]; mru[ek][0] = ew; }
                }
                /* serial_token applies the reference's push after the token; undo that by keeping a copy and re-deriving:
                 * simpler: let it push, and mark prevty = NONE-equivalent "already applied" */
                int ty;
                uint16_t before[256][2]; memcpy(before, mru, sizeof mru);
                const int spos = q;
                tok_out[nt++] = serial_token(s, ibuf, ilen, &q, &opos, mru, depth, lazy1, lazy2, &ty);
                if (stale) {                                   /* a replayed token wrote the dictionary as well: a ghost for whatever was speculated before */
                    lane_t* l = &clog[nlog];
                    setup_lane(s, ibuf, ilen, spos, 0, 0, 0, NL, l);      /* depth 0: text-derived fields only (the walk visits nothing) */
                    l->has_ev = 0; l->slot = s->bucket[ibuf[spos - 1]].head;
                    nlog++;
                }
                /* back to the window convention: state before the push of this token; W1's own swap (m[1]=m[0]; m[0]=w) IS
                 * that push with key ibuf[q-3] -- in both conventions the event is "after W1: unconditional push" */
                memcpy(mru, before, sizeof mru);
                prevty = ty;
                st.serial++;
                continue;
            }
            const int P = q;
            /* grid = 1: the round takes the grid window that holds q, from q on (the block's text starts at position 2) */
            const int win = grid == 1 ? (P - 2) / NL : -1;
            int wend = grid == 1 ? (2 + (win + 1) * NL < ilen ? 2 + (win + 1) * NL : ilen) : (ilen - P < NL ? ilen : P + NL);
            const int first_round = grid == 1 ? win != cur_win : 1;
            if (grid == 1 && first_round) { st.windows++; st.first_rounds++; cur_valid = 0; }
            /* ---- lane setup + phase 1: now, or (stale) what was evaluated while the window before this one was being resolved */
            ng = 0;
            int use_sp = 0;
            const lane_t* src = NULL; int src_base = 0; long src_log = 0;
            if (stale && grid == 1) {
                if (first_round && sp_win == win) { memcpy(SPcur, SP, sizeof(lane_t) * (size_t)sp_n); cur_log = sp_log; cur_valid = 1; }
                if (cur_valid) { src = SPcur; src_base = 2 + win * NL; src_log = cur_log; use_sp = 1; }
            } else if (stale && grid == 2 && sp_n > 0 && P >= sp_pos && P < sp_pos + sp_n) {
                src = SP; src_base = sp_pos; src_log = sp_log; use_sp = 1;
                if (wend > sp_pos + sp_n) wend = sp_pos + sp_n;            /* the range evaluated ahead ends here */
            }
            if (use_sp && (nlog - src_log > ghost_cap || nlog - src_log > MAXG)) { use_sp = 0; st.ghost_over++; if (grid == 2) wend = ilen - P < NL ? ilen : P + NL; }
            const int nlive = wend - P;
            cur_win = win;
            if (use_sp) {
                ng = (int)(nlog - src_log);
                for (int i = 0; i < ng; i++) { L[-1 - i] = clog[nlog - 1 - i]; S[-1 - i] = 1; slot_of[-1 - i] = L[-1 - i].slot; }
                for (int g = 0; g < nlive; g++) L[g] = src[P + g - src_base];
                st.ghost_sum += ng; if (ng > st.ghost_max) st.ghost_max = ng;
            } else {
                for (int g = 0; g < nlive; g++) setup_lane(s, ibuf, ilen, P + g, depth, lazy1, lazy2, NL, &L[g]);
                if (grid) st.sync_rounds++;
            }
            st.lanes_sum += nlive;
            { int no = 0, mx = 0; for (int g = 0; g < nlive; g++) if (L[g].sp_len >= 16) { no++; if (L[g].sp_len > mx) mx = L[g].sp_len; }
              st.open_lanes += no; if (no) { st.open_rounds++; st.maxtail_sum += (mx - 16 + 31) / 32; } }
            for (int g = 0; g < nlive; g++) {
                lane_t* l = &L[g];
                l->ty = l->sp_match ? TY_MATCH : TY_LIT;
                l->mlen = l->sp_len; l->tlen = l->sp_match ? l->sp_len : 1;
                if (guess_mru && !l->sp_match && l->pos + 1 < ilen) {       /* first guess from the MRU slots at the start of the round */
                    if (mru[l->ctx][0] == l->cw) { l->ty = TY_W0; l->tlen = 2; } else if (mru[l->ctx][1] == l->cw) { l->ty = TY_W1; l->tlen = 2; }
                }
                l->mnode_slot = l->sp_node; l->link_lane = LNONE;
            }
            /* stale: the next grid window is evaluated NOW -- beside this window's first iterate phase, i.e. before anything of this
             * window is committed -- and everything committed from here on is a ghost for it */
            if (stale && grid == 1 && first_round && wend < ilen) {
                sp_n = wend + NL < ilen ? NL : ilen - wend;
                for (int g = 0; g < sp_n; g++) setup_lane(s, ibuf, ilen, wend + g, depth, lazy1, lazy2, NL, &SP[g]);
                sp_win = win + 1; sp_log = nlog; st.spec_lanes += sp_n;
            }
            if (stale && grid == 2) {                              /* floating: a fixed range ahead of this round's start */
                sp_pos = P + ahead_c; sp_n = 0;
                if (sp_pos < ilen) {
                    sp_n = sp_pos + ahead_r < ilen ? ahead_r : ilen - sp_pos;
                    for (int g = 0; g < sp_n; g++) setup_lane(s, ibuf, ilen, sp_pos + g, depth, lazy1, lazy2, NL, &SP[g]);
                    sp_log = nlog; st.spec_lanes += sp_n;
                }
            }
            /* ---- iterate to the fixed point */
            int limit = nlive, limit_is_hard = 0, it = 0, tok_limit = nlive;
            for (;; it++) {
                memset(S, 0, (size_t)nlive + 1);
                { int cntS = 0, g = 0; for (; g < nlive && cntS < max_tok; g += L[g].tlen) { S[g] = 1; cntS++; } tok_limit = g < nlive ? g : nlive; }
                /* E step A: previous token type, events */
                int prev = -1;
                for (int g = 0; g < nlive; g++) {
                    lane_t* l = &L[g];
                    const int pty = prev < 0 ? prevty : L[prev].ty;
                    l->has_ev = pty == TY_MATCH || pty == TY_LIT || pty == TY_W1;
                    l->cond = pty == TY_MATCH;
                    if (S[g]) prev = g;
                }
                /* E step B: slot 0 before each event, effectiveness (lanes of S only are events) */
                for (int g = 0; g < nlive; g++) {
                    lane_t* l = &L[g];
                    int e = -1;
                    for (int k = g - 1; k >= 0; k--) if (S[k] && L[k].has_ev && L[k].ek == l->ek) { e = k; break; }
                    l->s0b = e >= 0 ? L[e].ew : mru[l->ek][0];
                    l->eff = l->has_ev && (!l->cond || l->ew != l->s0b);
                }
                /* E step C: match validity / exact in-window evaluation, MRU check, new type */
                for (int g = 0; g < nlive; g++) {
                    lane_t* l = &L[g];
                    int k = 0, a1 = LNONE, a2 = LNONE;
                    for (int j = g - 1; j >= -ng; j--) if (S[j] && L[j].canm) {
                        if (L[j].ctx == l->ctx) k++;
                        if (L[j].key == l->key) { if (a1 == LNONE) a1 = j; else if (a2 == LNONE) a2 = j; }
                    }
                    l->hard = 0; l->hardcls = 0;
                    if (only_s && !S[g]) { l->ty2 = l->ty; l->tlen2 = l->tlen; l->mlen2 = l->mlen; l->mnode2 = l->mnode_slot; l->link2 = l->link_lane; continue; }
                    int is_match = 0, mlen = 3, mnode = 0, link = LNONE;
                    if (l->canm) {
                        const int ring = l->dmin <= k;
                        /* lazy read sets: accepted starts <= g (own insert included, src/libzling_lz.cpp:271) */
                        int lconf[2] = {0, 0}, lconff[2] = {0, 0}, lhit[2] = {LNONE, LNONE}, lhits[2] = {0, 0};
                        for (int j2 = 0; j2 < 2; j2++) if (l->lwant[j2]) {
                            int cnt = 0;
                            for (int j = g; j >= -ng; j--) if (S[j] || j == g) if (L[j].canm) {
                                if (L[j].key == l->lkey[j2]) { if (lhit[j2] == LNONE) lhit[j2] = j; lhits[j2]++; }
                                if ((int)L[j].ctx == l->lctx[j2]) cnt++;
                            }
                            lconf[j2] = precise_risk ? cnt > l->ld[j2] : (l->lrisk[j2] && cnt > 0);   /* a visited ring slot at distance d is rewritten by the (d+1)-th insert */
                            lconff[j2] = cnt > l->ldfull[j2];
                        }
                        const int ring0 = l->has0 && l->d0 <= k, ring1 = l->has1 && l->d1 <= k;
                        const int lvl0fix = fix && level == 0;
                        int cutlen = -1, cutnode = 0;
                        if (ring_fix && !lvl0fix && ring && a1 == LNONE && l->nn > 0 && l->nd[0] > k) {
                            /* nodes are visited from the newest to the oldest: the first one whose slot has been handed out again ends the walk */
                            int ic = l->nn;
                            for (int i = 1; i < l->nn; i++) if (l->nd[i] <= k) { ic = i; break; }
                            cutlen = l->bl[ic]; cutnode = l->bn[ic];
                        }
                        if (cutlen >= 0) { is_match = cutlen >= ZO_MATCH_MIN; mlen = cutlen; mnode = cutnode; st.ringfixes++; }
                        else if (lvl0fix ? (a1 > LNONE ? (a2 == LNONE && ring0) : ring0) : ring) { l->hard = 1; l->hardcls = 2; }
                        else if (lvl0fix && a1 == LNONE && ring1) {
                            /* node 1's slot was rewritten by a start of this round: it now holds a later position than node 0's,
                             * so the reference's chain-end test (src/libzling_lz.cpp:265) stops the walk after node 0 */
                            mlen = l->len0 > 3 ? l->len0 : 3; mnode = l->node0; is_match = mlen >= ZO_MATCH_MIN; st.fixes++;
                        }
                        else if (a1 > LNONE) {
                            if (a1 < 0) st.ghost_a1++;
                            if (fix && level == 0) {
                                /* chain = [a1, a2 | node0] (depth 2): exact from the window's text */
                                const uint8_t* p = ibuf + l->pos;
                                int l1 = L[a1].chk == l->chk ? common_len(p, ibuf + L[a1].pos) : 0;
                                if (S[g] && l1 >= 16) st.long_fix_evals++;
                                int l2, n2;
                                if (a2 > LNONE) { l2 = L[a2].chk == l->chk ? common_len(p, ibuf + L[a2].pos) : 0; n2 = LANE_REF(a2); }
                                else { l2 = l->has0 ? l->len0 : 0; n2 = l->node0; }
                                mlen = 3; mnode = 0;
                                if (l1 > mlen) { mlen = l1; mnode = LANE_REF(a1); }
                                if (mlen != ZO_MATCH_MAX && (a2 > LNONE || l->has0) && l2 > mlen) { mlen = l2; mnode = n2; }
                                link = a1;
                                is_match = mlen >= ZO_MATCH_MIN;
                                st.fixes++;
                            } else if (fix && level > 0) {
                                /* generic depth: chain = [a1, a2] ++ the first depth - j nodes of the snapshot's chain (j <= 2) */
                                int a3 = LNONE;
                                for (int j = a2 - 1; a2 > LNONE && j >= -ng; j--) if (S[j] && L[j].canm && L[j].key == l->key) { a3 = j; break; }
                                if (a3 > LNONE || depth < 3) { l->hard = 1; l->hardcls = 1; }
                                else {
                                    const uint8_t* p = ibuf + l->pos;
                                    const int jn = a2 > LNONE ? 2 : 1;
                                    mlen = 3; mnode = 0;
                                    int l1 = L[a1].chk == l->chk ? common_len(p, ibuf + L[a1].pos) : 0;
                                    if (l1 > mlen) { mlen = l1; mnode = LANE_REF(a1); }
                                    if (a2 > LNONE && mlen != ZO_MATCH_MAX) {
                                        int l2 = L[a2].chk == l->chk ? common_len(p, ibuf + L[a2].pos) : 0;
                                        if (l2 > mlen) { mlen = l2; mnode = LANE_REF(a2); }
                                    }
                                    if (mlen != ZO_MATCH_MAX && l->pl[jn] > mlen) { mlen = l->pl[jn]; mnode = l->pn[jn]; }
                                    link = a1;
                                    is_match = mlen >= ZO_MATCH_MIN;
                                    st.fixes++;
                                }
                            } else { l->hard = 1; l->hardcls = 1; }
                        } else { is_match = l->sp_len >= ZO_MATCH_MIN; mlen = l->sp_len; mnode = l->sp_node; }
                        if (!l->hard && is_match && mlen < ZO_LAZY_LIMIT) {
                            /* lazy probes under mlen */
                            int veto = 0, need_hard = 0;
                            for (int j2 = 0; j2 < 2 && !veto; j2++) if (l->lwant[j2]) {
                                const int conflict = lhit[j2] > LNONE || lconf[j2];
                                if (!conflict && mlen == l->sp_len) { /* speculative probe stands */
                                    /* per-probe veto is not kept apart in the model: recompute on the snapshot (read-only, exact) */
                                    veto = lazy_probe(s, ibuf, l->pos + 1 + j2, mlen, j2 == 0 ? lazy1 : lazy2);
                                } else if (fix && level == 0 && !lconf[j2]) {
                                    /* depth-1 probe: the chain head is the newest accepted start with the probe's key, else the snapshot's */
                                    const int m = mlen - 3, pp = l->pos + 1;
                                    if (lhit[j2] > LNONE) { veto = le32(ibuf + pp + m) == le32(ibuf + L[lhit[j2]].pos + m); st.lfixes++; }
                                    else veto = l->lsrc1_valid && le32(ibuf + pp + m) == le32(ibuf + l->lsrc1 + m);
                                } else if (fix && level > 0 && !lconf[j2] && mlen == l->sp_len && lhits[j2] <= 2) {
                                    /* generic probe: [the newest one or two accepted starts with the probe's key] ++ the first depth - h nodes
                                     * of the snapshot's probe chain; the length is the speculation's, so the snapshot part is known (vpos) */
                                    const int Lp = j2 == 0 ? lazy1 : lazy2, m = mlen - 3, pp = l->pos + 1 + j2;
                                    int h = 0;
                                    for (int j = g; j >= -ng && h < Lp && !veto; j--) if ((S[j] || j == g) && L[j].canm && L[j].key == l->lkey[j2]) {
                                        if (le32(ibuf + pp + m) == le32(ibuf + L[j].pos + m)) veto = 1;
                                        h++;
                                    }
                                    if (!veto && h < Lp) {
                                        /* the speculative walk of probe 2 only ran if probe 1 did not veto: its vpos is valid then (same here) */
                                        veto = l->vpos[j2] < Lp - h;
                                    }
                                    st.lfixes++;
                                } else if (lazy_fix && fix && level > 0 && mlen != l->sp_len && !lconff[j2] && lhits[j2] <= 2) {
                                    /* changed length: the probe walked again under it */
                                    const int Lp = j2 == 0 ? lazy1 : lazy2, m = mlen - 3, pp = l->pos + 1 + j2;
                                    int h = 0;
                                    for (int j = g; j >= -ng && h < Lp && !veto; j--) if ((S[j] || j == g) && L[j].canm && L[j].key == l->lkey[j2]) {
                                        if (le32(ibuf + pp + m) == le32(ibuf + L[j].pos + m)) veto = 1;
                                        h++;
                                    }
                                    if (!veto && h < Lp) veto = lazy_probe(s, ibuf, pp, mlen, Lp - h);
                                    st.lazyfixes2++;
                                } else {
                                    need_hard = 1;
                                    if (S[g]) { if (lconf[j2]) st.lz_why[0]++; else if (mlen != l->sp_len) st.lz_why[1]++; else if (lhits[j2] > 2) st.lz_why[2]++; else st.lz_why[3]++; }
                                    break;
                                }
                            }
                            if (need_hard) { l->hard = 1; l->hardcls = 3; }
                            else if (veto) is_match = 0;
                        }
                    }
                    if (l->hard) { l->ty2 = l->ty; l->tlen2 = l->tlen; l->mlen2 = l->mlen; l->mnode2 = l->mnode_slot; l->link2 = l->link_lane; continue; }
                    if (is_match) { l->ty2 = TY_MATCH; l->tlen2 = mlen; l->mlen2 = mlen; l->mnode2 = mnode; l->link2 = link; continue; }
                    l->link2 = link; l->mlen2 = 3; l->mnode2 = 0;
                    /* word MRU: slots of my context after every event of S at boundaries <= g */
                    int ty = TY_LIT;
                    if (l->pos + 1 < ilen) {
                        int e1 = -1, e2 = -1;
                        for (int j = g; j >= 0; j--) if ((S[j] || j == g) && L[j].has_ev && L[j].ek == l->ctx) {
                            if (j == g && !S[g]) { /* g evaluated as a hypothetical start: its own event counts */ }
                            if (e1 < 0) e1 = j;
                            if (e2 < 0 && L[j].eff) e2 = j;
                            if (e1 >= 0 && e2 >= 0) break;
                        }
                        const uint32_t s0 = e1 >= 0 ? L[e1].ew : mru[l->ctx][0];
                        const uint32_t s1 = e2 >= 0 ? L[e2].s0b : mru[l->ctx][1];
                        if (s0 == l->cw) ty = TY_W0; else if (s1 == l->cw) ty = TY_W1;
                    }
                    l->ty2 = ty; l->tlen2 = ty == TY_LIT ? 1 : 2;
                }
                /* cut of the round: first hard lane of S, or the sub-block's end */
                limit = tok_limit; limit_is_hard = 0;
                { int sym = 0;
                  for (int g = 0; g < nlive; g++) if (S[g]) {
                      if (!(opos + sym + 1 < ZO_SUBBLOCK_SYMS)) { limit = g; break; }
                      if (L[g].hard) { limit = g; limit_is_hard = 1; break; }
                      sym += L[g].ty2 == TY_MATCH ? 2 : 1;
                  } }
                int changed = 0;
                static char chg_lane[MAXNL];
                memset(chg_lane, 0, sizeof chg_lane);
                for (int g = 0; g < limit; g++) if (S[g]) {
                    lane_t* l = &L[g];
                    if (l->ty2 != l->ty || l->tlen2 != l->tlen) { changed = 1; chg_lane[g] = 1; }
                }
                /* (a lane whose match source / link changed but not its length: S is unchanged, later lanes saw the same S) */
                for (int g = 0; g < nlive; g++) { lane_t* l = &L[g]; l->ty = l->ty2; l->tlen = l->tlen2; l->mlen = l->mlen2; l->mnode_slot = l->mnode2; l->link_lane = l->link2; }
                if (!changed) break;
                if (prefix_pct > 0 && it == 0) {
                    int gf = -1, cf = 0, ct = 0;
                    for (int g = 0; g < limit; g++) if (S[g]) { if (gf < 0 && chg_lane[g]) gf = g; if (gf < 0) cf++; ct++; }
                    if (gf > 0 && cf * 100 >= prefix_pct * ct) { limit = gf; limit_is_hard = 0; st.prefix_rounds++; break; }
                }
                if (it > 4 * NL) { fprintf(stderr, "no convergence at P=%d\n", P); return -1; }
            }
            st.rounds++; st.iters += it + 1; st.hist_it[it < 15 ? it : 15]++; if (it + 1 > st.maxit) st.maxit = it + 1;
            /* ---- commit S below the limit (sequentially here; the formulas are the kernel's) */
            uint16_t mru_seq[256][2]; memcpy(mru_seq, mru, sizeof mru);     /* sequential emulation of the events, for the assert */
            int ncom = 0, last = -1;
            for (int g = 0; g < limit; g++) if (S[g]) {
                lane_t* l = &L[g];
                /* events, sequential emulation */
                if (l->has_ev) {
                    if (l->cond) { if (mru_seq[l->ek][0] != l->ew) { mru_seq[l->ek][1] = mru_seq[l->ek][0]; mru_seq[l->ek][0] = (uint16_t)l->ew; } }
                    else { mru_seq[l->ek][1] = mru_seq[l->ek][0]; mru_seq[l->ek][0] = (uint16_t)l->ew; }
                }
                /* check the parallel MRU outcome against the sequential state */
                if (l->ty != TY_MATCH) {
                    int want = TY_LIT;
                    if (l->pos + 1 < ilen) { if (mru_seq[l->ctx][0] == l->cw) want = TY_W0; else if (mru_seq[l->ctx][1] == l->cw) want = TY_W1; }
                    if (want != l->ty) { fprintf(stderr, "MRU formula mismatch at pos %d: %d vs %d\n", l->pos, l->ty, want); return -1; }
                }
                uint32_t word;
                if (l->canm) {
                    zo_bucket* b = &s->bucket[l->ctx];
                    int k = 0; for (int j = -ng; j < g; j++) if (S[j] && L[j].canm && L[j].ctx == l->ctx) k++;
                    const int head = (l->head0 + k + 1) & (ZO_RING - 1);
                    b->head = (uint16_t)((b->head + 1) & (ZO_RING - 1));
                    if (b->head != head) { fprintf(stderr, "head formula mismatch\n"); return -1; }
                    slot_of[g] = head; l->slot = head;
                    const int link = l->link_lane > LNONE ? slot_of[l->link_lane] : l->node0;
                    if (b->hash[l->hc] != link) { fprintf(stderr, "link mismatch at pos %d: hash head %d, link %d\n", l->pos, b->hash[l->hc], link); return -1; }
                    b->suffix[head] = (uint16_t)link; b->offset[head] = (uint32_t)l->pos | l->chk << 24; b->hash[l->hc] = (uint16_t)head;
                    if (l->ty == TY_MATCH) {
                        const int mn = l->mnode_slot < 0 ? slot_of[REF_LANE(l->mnode_slot)] : l->mnode_slot;
                        word = (uint32_t)(258 + l->mlen - ZO_MATCH_MIN) | (uint32_t)((head - mn) & (ZO_RING - 1)) << 16;
                    } else word = 0;
                } else word = 0;
                if (l->ty == TY_W0) word = 256; else if (l->ty == TY_W1) word = 257; else if (l->ty == TY_LIT) word = l->b0 | l->ctx << 16;
                tok_out[nt++] = word;
                opos += l->ty == TY_MATCH ? 2 : 1;
                ncom++; last = g;
                if (l->ty == TY_MATCH && l->mlen >= 16) st.open_tok++;
                if (stale) { clog[nlog] = *l; clog[nlog].has_ev = 0; nlog++; }
            }
            /* final MRU state by the kernel's formula: per key, the last event lane writes (s0, s1) */
            for (int g = 0; g < limit; g++) if (S[g] && L[g].has_ev) {
                lane_t* l = &L[g];
                int later = 0; for (int j = g + 1; j < limit; j++) if (S[j] && L[j].has_ev && L[j].ek == l->ek) later = 1;
                if (later) continue;
                int e2 = -1; for (int j = g; j >= 0; j--) if (S[j] && L[j].eff && L[j].ek == l->ek) { e2 = j; break; }
                mru[l->ek][0] = (uint16_t)l->ew; mru[l->ek][1] = e2 >= 0 ? (uint16_t)L[e2].s0b : mru[l->ek][1];
            }
            if (memcmp(mru, mru_seq, sizeof mru)) { fprintf(stderr, "final MRU formula mismatch in round at %d\n", P); return -1; }
            st.committed += ncom;
            if (last >= 0) { q = L[last].pos + L[last].tlen; prevty = L[last].ty; }
            if (limit_is_hard) { force_serial = 1; st.hard[L[limit].hardcls]++; }
            else if (limit < nlive && ncom == 0 && !(opos + 1 < ZO_SUBBLOCK_SYMS)) { /* sub-block full */ }
            if (ncom == 0 && !limit_is_hard && opos + 1 < ZO_SUBBLOCK_SYMS) { fprintf(stderr, "no progress at %d\n", P); return -1; }
            (void)tok_limit;
        }
        cuts[4 * nsub] = tok_begin; cuts[4 * nsub + 1] = nt; cuts[4 * nsub + 2] = q; cuts[4 * nsub + 3] = opos;
        nsub++;
    }
    st.tokens += nt;
    *ncut = nsub;
    return nt;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s FILE [NL] [level] [fix] [max_bytes]\n", argv[0]); return 2; }
    const int NL = argc > 2 ? atoi(argv[2]) : 256, level = argc > 3 ? atoi(argv[3]) : 0, fix = argc > 4 ? atoi(argv[4]) : 1;
    const long maxb = argc > 5 ? atol(argv[5]) : (1L << 62);
    if (argc > 6) precise_risk = atoi(argv[6]);
    if (argc > 7) only_s = atoi(argv[7]);
    if (argc > 8) max_tok = atoi(argv[8]);
    if (argc > 9) guess_mru = atoi(argv[9]);
    if (argc > 10) prefix_pct = atoi(argv[10]);
    if (argc > 11) grid = atoi(argv[11]);
    if (argc > 12) stale = atoi(argv[12]);
    if (argc > 13) ghost_cap = atoi(argv[13]);
    if (getenv("RING_FIX")) ring_fix = atoi(getenv("RING_FIX"));
    if (getenv("LAZY_FIX")) lazy_fix = atoi(getenv("LAZY_FIX"));
    if (argc > 14) ahead_c = atoi(argv[14]);
    if (argc > 15) ahead_r = atoi(argv[15]);
    if (stale && !grid) { fprintf(stderr, "stale = 1 needs grid = 1 or 2\n"); return 2; }
    if (ahead_r > MAXNL) return 2;
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    if (n > maxb) n = maxb;
    uint8_t* x = (uint8_t*)malloc((size_t)n + 1024); memset(x + n, 0, 1024);
    if (fread(x, 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);
    tables_init();
    zo_stream* sm = zo_stream_new(level); zo_stream* so = zo_stream_new(level);
    uint32_t* tm = (uint32_t*)malloc(sizeof(uint32_t) * (ZO_BLOCK_IN + 64));
    uint32_t* to = (uint32_t*)malloc(sizeof(uint32_t) * ZO_SUBBLOCK_SYMS);
    static int cuts[4 * 200];
    int bad = 0;
    for (long off = 0; off < n && !bad; off += ZO_BLOCK_IN) {
        const int ilen = (int)(n - off < ZO_BLOCK_IN ? n - off : ZO_BLOCK_IN);
        int ncut = 0;
        const int nt = parse_block_model(sm, x + off, ilen, level, NL, fix, tm, cuts, &ncut);
        if (nt < 0) { bad = 1; break; }
        zo_reset_buckets(so);
        int enc = 0, sub = 0, at = 0;
        while (enc < ilen) {
            int rlen = 0;
            const int k = zo_parse_subblock(so, level, x + off, ilen, &enc, to, &rlen, 0);
            if (sub >= ncut || cuts[4 * sub + 1] - cuts[4 * sub] != k || cuts[4 * sub + 2] != enc || cuts[4 * sub + 3] != rlen) {
                fprintf(stderr, "block at %ld sub %d: cut mismatch (model %d tokens enc %d rlen %d, oracle %d %d %d)\n", off, sub,
                        sub < ncut ? cuts[4 * sub + 1] - cuts[4 * sub] : -1, sub < ncut ? cuts[4 * sub + 2] : -1, sub < ncut ? cuts[4 * sub + 3] : -1, k, enc, rlen);
                bad = 1;
            }
            for (int i = 0; i < k && !bad; i++) if (tm[at + i] != to[i]) {
                fprintf(stderr, "block at %ld sub %d token %d: model %08x oracle %08x\n", off, sub, i, tm[at + i], to[i]); bad = 1;
            }
            if (bad) break;
            at += k; sub++;
        }
        if (!bad && (sub != ncut || at != nt)) { fprintf(stderr, "count mismatch\n"); bad = 1; }
    }
    printf("%s NL=%d level=%d fix=%d bytes=%ld: %s | rounds %ld tokens %ld committed/round %.1f positions/round %.1f iterations/round %.2f (max %ld) serial tokens/round %.3f "
           "(key %ld ring %ld lazy %ld) fixes %ld lazyfixes %ld\n", argv[1], NL, level, fix, n, bad ? "MISMATCH" : "exact",
           st.rounds, st.tokens, (double)st.committed / (st.rounds ? st.rounds : 1), (double)n / (st.rounds ? st.rounds : 1), (double)st.iters / (st.rounds ? st.rounds : 1), st.maxit,
           (double)st.serial / (st.rounds ? st.rounds : 1), st.hard[1], st.hard[2], st.hard[3], st.fixes, st.lfixes);
    printf("   lazy-hard causes among tokens of S (per evaluation, not per round cut): probe slot rewritten %ld, length changed %ld, > 2 tokens with the probe's key %ld, other %ld\n",
           st.lz_why[0], st.lz_why[1], st.lz_why[2], st.lz_why[3]);
    printf("   long matches: lanes with a speculative match >= 16 bytes %.1f per round (in %.1f %% of the rounds; longest tail %.2f 32-byte trips per such round), "
           "committed tokens >= 16 bytes %.2f per round, same-slot fixes against a >= 16-byte candidate %.2f per round\n",
           (double)st.open_lanes / (st.rounds ? st.rounds : 1), 100.0 * st.open_rounds / (st.rounds ? st.rounds : 1), (double)st.maxtail_sum / (st.open_rounds ? st.open_rounds : 1),
           (double)st.open_tok / (st.rounds ? st.rounds : 1), (double)st.long_fix_evals / (st.rounds ? st.rounds : 1));
    if (lazy_fix) printf("   lazy_fix: %ld probes walked again under a changed length instead of going hard (%.3f per round)\n", st.lazyfixes2, (double)st.lazyfixes2 / (st.rounds ? st.rounds : 1));
    if (ring_fix) printf("   ring_fix: %ld walks ended at a rewritten node instead of going hard (%.3f per round)\n", st.ringfixes, (double)st.ringfixes / (st.rounds ? st.rounds : 1));
    printf("   iterations histogram:"); for (int i = 0; i < 16; i++) printf(" %ld", st.hist_it[i]); printf("\n");
    if (grid) printf("   grid %d stale %d ghost_cap %d: windows %ld rounds/window %.3f lanes/round %.1f sync rounds %ld (%.1f %% of rounds) ghosts/round %.1f (max %ld, over the cap %ld) "
                     "chain head is a ghost %ld (%.2f per round) lanes speculated ahead %ld\n", grid, stale, ghost_cap, st.windows, (double)st.rounds / (st.windows ? st.windows : 1),
                     (double)st.lanes_sum / (st.rounds ? st.rounds : 1), st.sync_rounds, 100.0 * st.sync_rounds / (st.rounds ? st.rounds : 1),
                     (double)st.ghost_sum / (st.rounds ? st.rounds : 1), st.ghost_max, st.ghost_over, st.ghost_a1, (double)st.ghost_a1 / (st.rounds ? st.rounds : 1), st.spec_lanes);
    return bad;
}
