// match.hip.h — the final row test on the device:
//   compileRowMatcher / matchRowBytes / match / matchLeafTokens / evalMatcherNode, row_matcher.go:257-626
// for Field / Token / FieldToken conditions.  The same resumable walker as k_ingest_rows enumerates the row; every
// emission's 256-bit base hash is compared with the conditions' (field and token strings hashed separately:
// FieldToken compares the (path, token) PAIR at one leaf, never the joined "path::token" key — row_matcher.go:587),
// satisfaction flags are monotone, and the expression is evaluated once the walk is over (the reference's early exit
// changes cost, not verdicts).  Equality is decided on all four hash words of an entry AND its keyed 64-bit fingerprint
// (ingest.hip.h, FpKey): an emission whose hashes equal a condition's but whose fingerprint differs is a murmur3 state
// collision — the row is handed to the host matcher, which compares bytes as matchRowBytes does.
// Rows outside the device walker's envelope are reported back and decided by the host matcher too.
#pragma once
#include "ingest.hip.h"

namespace bsg {

constexpr uint32_t kMatchMaxConds = 64;    // satisfaction flags live in one u64 per row
constexpr uint32_t kMatchMaxOps = 512;
constexpr uint32_t kMatchCondWords = 11;   // hf[4], ht[4], fingerprint of the field string, of the token string, kind

struct MatchArgs {
    const uint8_t *rows;
    const uint64_t *row_off;
    const uint64_t *cond_h;      // [2 * n_conds][4]: base hashes of condition c's field string (2c) and token string (2c + 1)
    const uint64_t *cond_fp;     // [2 * n_conds]: their keyed fingerprints
    const uint32_t *cond_kind;   // [n_conds]
    const uint32_t *prog;        // lowered postfix program: TERM i / AND2 / OR2 / TRUE / FALSE (opcodes 0,1,2,3,4)
    const uint32_t *lower;
    uint64_t *out_bits;          // bit r & 63 of word r >> 6: row r matches
    uint32_t *fallback_rows;
    uint32_t *n_fallback;
    uint32_t n_rows, n_conds, n_ops;
    uint32_t row_base;           // added to the row indices reported in fallback_rows (a launch covers rows [row_base, row_base + n_rows) of the call;
                                 // row_off / out_bits already point at its first row / word)
    FpKey key;
};

constexpr uint32_t kMatchLdsBytes = kMatchMaxConds * kMatchCondWords * 8 + kMatchMaxOps * 4 + kIngestThreads * kLaneLds;

__global__ __launch_bounds__(kIngestThreads) void k_match_rows(const MatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    lds_u64i *conds = (lds_u64i *)lds_raw;
    typedef __attribute__((address_space(3))) uint32_t lds_u32i;
    lds_u32i *prog = (lds_u32i *)(lds_raw + kMatchMaxConds * kMatchCondWords * 8);
    for (uint32_t c = threadIdx.x; c < a.n_conds; c += kIngestThreads) {
        lds_u64i *e = conds + c * kMatchCondWords;
        for (uint32_t j = 0; j < 4; ++j) { e[j] = a.cond_h[(uint64_t)(2 * c) * 4 + j]; e[4 + j] = a.cond_h[(uint64_t)(2 * c + 1) * 4 + j]; }
        e[8] = a.cond_fp[2 * c]; e[9] = a.cond_fp[2 * c + 1];
        e[10] = a.cond_kind[c];
    }
    for (uint32_t i = threadIdx.x; i < a.n_ops; i += kIngestThreads) prog[i] = a.prog[i];
    __syncthreads();
    const uint32_t r = blockIdx.x * kIngestThreads + threadIdx.x;
    const bool live = r < a.n_rows;
    Walker w;
    ChunkCursor cc;
    cc.chunks = reinterpret_cast<const uint64_t *>(a.rows);
    w.path = (lds_u8 *)lds_raw + kMatchMaxConds * kMatchCondWords * 8 + kMatchMaxOps * 4 + threadIdx.x * kLaneLds;
    w.lower = a.lower;
    w.key = a.key;
    w.ft_on = false;
    hs_init(w.ps, w.key); hs_init(w.tok, w.key); hs_init(w.ft, w.key);
    walker_reset(w, cc, live ? a.row_off[r] : 0, live ? a.row_off[r + 1] : 0, live);
    uint32_t res = live ? R_CONTINUE : R_DONE;
    uint64_t sat = 0, leaf_mask = 0;
    bool collided = false;       // equal hashes, different fingerprint: only the host's byte compare can decide this row
    while (__ballot(res == R_CONTINUE || w.req != Q_NONE) != 0ull) {
        while (__ballot(res == R_CONTINUE && w.req == Q_NONE) != 0ull)
            if (res == R_CONTINUE && w.req == Q_NONE) res = advance<true>(w, cc);
        const uint32_t q = w.req;
        w.req = Q_NONE;
        HashStream s;
        hs_init(s, w.key);
        const uint32_t plen = (q == Q_FIELD || q == Q_LEAF) ? w.req_len : 0u;
        for (uint32_t i = 0; __ballot(i < plen) != 0ull; ++i)
            if (i < plen) hs_absorb(s, w.path[i], w.key);
        if (q == Q_WORD) s = w.tok;
        uint64_t h[4] = {0, 0, 0, 0}, fp = 0;
        if (q != Q_NONE) fp = hs_finish(s, h, w.key);
        if (q == Q_LEAF) leaf_mask = 0;
        const bool is_path = q == Q_FIELD || q == Q_LEAF, is_word = q == Q_WORD;
        for (uint32_t c = 0; c < a.n_conds; ++c) {                       // uniform loop: the conditions come from LDS broadcasts
            const lds_u64i *e = conds + c * kMatchCondWords;
            const uint32_t kind = (uint32_t)e[10];                       // 0 Field, 1 Token, 2 FieldToken
            const uint64_t bit = 1ULL << c;
            if (kind != 1u) {                                            // conditions with a field: compare paths
                const bool heq = is_path && e[0] == h[0] && e[1] == h[1] && e[2] == h[2] && e[3] == h[3];
                const bool eq = heq && e[8] == fp;
                collided |= heq && !eq;
                if (eq && kind == 0u) sat |= bit;                        // Field: any emission of that path (row_matcher.go:511-516)
                if (eq && kind == 2u && q == Q_LEAF) leaf_mask |= bit;   // FieldToken: this leaf's words may complete the pair
            }
            if (kind != 0u) {                                            // conditions with a token: compare words
                const bool heq = is_word && e[4] == h[0] && e[5] == h[1] && e[6] == h[2] && e[7] == h[3];
                const bool eq = heq && e[9] == fp;
                collided |= heq && !eq;
                if (eq && (kind == 1u || (leaf_mask & bit))) sat |= bit; // Token anywhere; FieldToken only under its own path
            }
        }
    }
    // evalMatcherNode over the flags: one bit of stack per lane and level
    uint64_t stk = 0;
    for (uint32_t j = 0; j < a.n_ops; ++j) {
        const uint32_t op = prog[j], opc = op >> 28;
        if (opc == 0u) stk = (stk << 1) | ((sat >> (op & 63u)) & 1ULL);
        else if (opc == 3u) stk = (stk << 1) | 1ULL;
        else if (opc == 4u) stk = stk << 1;
        else {
            const uint64_t x = stk & 1ULL, y = (stk >> 1) & 1ULL;
            stk = ((stk >> 2) << 1) | (opc == 1u ? (x & y) : (x | y));
        }
    }
    const bool verdict = a.n_ops == 0 ? true : (stk & 1ULL) != 0;        // nil expression matches every row
    if (collided && res == R_DONE) res = R_FAIL;
    const uint64_t word = __ballot(live && res == R_DONE && verdict);
    if ((threadIdx.x & 63u) == 0u && (r & ~63u) < a.n_rows) a.out_bits[r >> 6] = word;
    if (res == R_FAIL) {
        const uint32_t slot = __hip_atomic_fetch_add(a.n_fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.fallback_rows[slot] = a.row_base + r;
    }
}

}  // namespace bsg
