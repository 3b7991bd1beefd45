// zlng_group.hip -- one .zlng stream over several contexts / devices (host code only; C-ABI in include/zlng.h).
//
// north_star: "independent 16 MB blocks shard trivially across the 8 GPUs of one node (per-GPU block ranges
// concatenated on the host)".  What shards is the PARSE (Reset per block, src/libzling.cpp:197); the literal ranks
// follow the 256 MTF tables that the reference keeps for the whole stream (src/libzling_lz.cpp:197-209 does not reset
// them) and current_level (src/libzling.cpp:185, 261-266).  A group therefore runs
//
//     every member:   H2D copy + parse of its contiguous block range        (all members at once, own HIP streams)
//     member 0..n-1:  import (64 KiB tables, level) -> rank + Huffman + frame -> export      (in stream order)
//
// and the members' bytes concatenate to exactly the single-context stream.  A member is a zlng_ctx; the same device
// may be listed more than once (several contexts per GPU keep more blocks in flight than one context's 240-block
// ceiling allows, and let a one-GPU box exercise the multi-member path).
#include <algorithm>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/zlng.h"

struct zlng_group {
    std::vector<zlng_ctx*> member;
    int level = 0;
    int max_blocks = 0;                       // per member
    std::vector<uint8_t> mtf;                 // stream state between calls (host copy)
    int current_level = 0;
    // pending parse
    std::vector<size_t> part_len;             // bytes given to each member (0 = idle this call)
    size_t pending_len = 0;
};

namespace {
const size_t kBlock = ZLNG_BLOCK_SIZE;
}

extern "C" {

zlng_group* zlng_group_create(const int* devices, int ndev, int level, int max_blocks_per_member, int* err) {
    int dummy;
    if (!err) err = &dummy;
    *err = ZLNG_OK;
    if (!devices || ndev <= 0 || ndev > 64) { *err = ZLNG_E_ARG; return nullptr; }
    zlng_group* g = new (std::nothrow) zlng_group();
    if (!g) { *err = ZLNG_E_NOMEM; return nullptr; }
    g->level = level;
    g->current_level = level;
    g->max_blocks = max_blocks_per_member;
    g->mtf.resize(ZLNG_MTF_STATE);
    g->part_len.assign((size_t)ndev, 0);
    for (int i = 0; i < ndev; i++) {
        zlng_ctx* c = zlng_create(devices[i], level, 1, max_blocks_per_member, err);
        if (!c) { zlng_group_destroy(g); return nullptr; }
        g->member.push_back(c);
    }
    if ((*err = zlng_get_state(g->member[0], g->mtf.data(), &g->current_level)) != ZLNG_OK) { zlng_group_destroy(g); return nullptr; }
    return g;
}

void zlng_group_destroy(zlng_group* g) {
    if (!g) return;
    for (zlng_ctx* c : g->member) zlng_destroy(c);
    delete g;
}

int zlng_group_set_host_rank_contexts(zlng_group* g, int k) {
    if (!g) return ZLNG_E_ARG;
    for (zlng_ctx* c : g->member) { const int rc = zlng_set_host_rank_contexts(c, k); if (rc != ZLNG_OK) return rc; }
    return ZLNG_OK;
}

int zlng_group_members(const zlng_group* g) { return g ? (int)g->member.size() : 0; }
size_t zlng_group_capacity(const zlng_group* g) { return g ? g->member.size() * (size_t)g->max_blocks * kBlock : 0; }

int zlng_group_get_state(zlng_group* g, uint8_t mtf[ZLNG_MTF_STATE], int* current_level) {
    if (!g || !mtf) return ZLNG_E_ARG;
    memcpy(mtf, g->mtf.data(), ZLNG_MTF_STATE);
    if (current_level) *current_level = g->current_level;
    return ZLNG_OK;
}

int zlng_group_set_state(zlng_group* g, const uint8_t mtf[ZLNG_MTF_STATE], int current_level) {
    if (!g || !mtf) return ZLNG_E_ARG;
    const int rc = zlng_set_state(g->member[0], mtf, current_level);      // validates tables and level
    if (rc != ZLNG_OK) return rc;
    memcpy(g->mtf.data(), mtf, ZLNG_MTF_STATE);
    g->current_level = current_level;
    return ZLNG_OK;
}

// Contiguous block ranges, as even as possible; members beyond the number of blocks stay idle.
int zlng_group_encode_parse(zlng_group* g, const uint8_t* in, size_t in_len) {
    if (!g || !in || in_len == 0) return ZLNG_E_ARG;
    const size_t nb = (in_len + kBlock - 1) / kBlock, nm = g->member.size();
    if (nb > nm * (size_t)g->max_blocks) return ZLNG_E_ARG;
    const size_t base = nb / nm, extra = nb % nm;
    size_t off = 0;
    for (size_t m = 0; m < nm; m++) {
        const size_t blocks = base + (m < extra ? 1 : 0);
        const size_t len = std::min(blocks * kBlock, in_len - off);
        g->part_len[m] = len;
        if (len) {
            const int rc = zlng_encode_parse(g->member[m], in + off, len);   // returns once the copy is staged; the parse runs on
            if (rc != ZLNG_OK) { g->pending_len = 0; return rc; }            // the member's own stream
        }
        off += len;
    }
    g->pending_len = in_len;
    return ZLNG_OK;
}

// Members finish in stream order (the tables and the level travel member to member), but a member's BYTES are not part of that
// chain: its finish leaves them in its own staging buffer (zlng_encode_finish_staged) and a helper thread brings them to the host
// (zlng_encode_copy_out) while the next member's rank stage already runs.  The last active member's copy has nothing left to hide
// behind and runs on the caller's thread.
int zlng_group_encode_finish(zlng_group* g, uint8_t* out, size_t out_cap, size_t* out_len, size_t* per_block_out_end) {
    if (!g || !g->pending_len || !out || !out_len) return ZLNG_E_ARG;
    *out_len = 0;
    size_t produced = 0, blk0 = 0;
    std::vector<uint8_t> mtf = g->mtf;                 // committed to the group only when every member succeeded
    int level = g->current_level;
    const size_t nm = g->member.size();
    size_t last = 0;
    for (size_t m = 0; m < nm; m++) if (g->part_len[m]) last = m;
    std::vector<std::thread> copies;
    std::vector<int> copy_rc(nm, ZLNG_OK);
    auto join_all = [&] { for (std::thread& t : copies) t.join(); copies.clear(); };
    for (size_t m = 0; m < nm; m++) {
        const size_t len = g->part_len[m];
        if (!len) continue;
        zlng_ctx* c = g->member[m];
        int rc = zlng_set_state(c, mtf.data(), level);
        size_t n = 0;
        size_t* ends = per_block_out_end ? per_block_out_end + blk0 : nullptr;
        if (rc == ZLNG_OK) rc = zlng_encode_finish_staged(c, out_cap - produced, &n, ends);
        if (rc == ZLNG_OK) rc = zlng_get_state(c, mtf.data(), &level);
        if (rc != ZLNG_OK) { join_all(); g->pending_len = 0; return rc; }   // the group's stream state is unchanged: submit the range again
        uint8_t* dst = out + produced;
        if (m == last) copy_rc[m] = zlng_encode_copy_out(c, dst, n);
        else copies.emplace_back([c, dst, n, m, &copy_rc] { copy_rc[m] = zlng_encode_copy_out(c, dst, n); });
        const size_t nblk = (len + kBlock - 1) / kBlock;
        if (ends) for (size_t b = 0; b < nblk; b++) ends[b] += produced;
        produced += n;
        blk0 += nblk;
    }
    join_all();
    g->pending_len = 0;
    for (size_t m = 0; m < nm; m++) if (copy_rc[m] != ZLNG_OK) return copy_rc[m];
    g->mtf.swap(mtf);
    g->current_level = level;
    *out_len = produced;
    return ZLNG_OK;
}

int zlng_group_encode_blocks(zlng_group* g, const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, size_t* out_len,
                             size_t* per_block_out_end) {
    if (!g || !out_len) return ZLNG_E_ARG;
    *out_len = 0;
    if (in_len == 0) return ZLNG_OK;
    const int rc = zlng_group_encode_parse(g, in, in_len);
    if (rc != ZLNG_OK) return rc;
    return zlng_group_encode_finish(g, out, out_cap, out_len, per_block_out_end);
}

}  // extern "C"
