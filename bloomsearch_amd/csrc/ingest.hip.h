// ingest.hip.h — CDNA4 (gfx950) device code for the construct side of the path:
// marshaled-JSON rows -> distinct bloom entries (fields / tokens / field::token) per set,
// exact distinct counts, then bitsets straight from the distinct sets.
//
// Reference functions this replaces on the device:
//   pathWalker.walk / walkValue / emitKeyPrefixPaths ........ row_matcher.go:51-135 (twin: tokenizer.go:51-113)
//   leafTokenInput ........................................... tokenizer.go:120-133
//   BasicWhitespaceLowerTokenizer (forEachWord/appendFoldedWord) tokenizer.go:141-143, row_matcher.go:142-202
//   bloomEntrySets.indexRow / addFieldToken / unionInto / counts  ingest.go:24-123 (key: tokenizer.go:509-511)
//   buildFilters' AddString loop ............................. ingest.go:127-145
//
// Scope of the device walker ("walker-lite"): rows made of printable ASCII (0x20..0x7E) without any
// backslash, nesting <= kMaxDepth, paths <= kPathCap bytes.  A row that leaves that envelope — or is
// malformed — is appended to the fallback list and finished by the host walker (walker.hpp), which
// handles escapes, UTF-8, Unicode white space / case folding and the lenient error semantics.
// Set semantics make the hand-over exact without a validation pre-pass: the device grammar is never
// laxer than the host's and emits in the same document order, so whatever a row inserted before it was
// flagged is a subset of what the host walker inserts for that row, and inserts are idempotent.
//
// A "set" is the distinct-entry state of one partition buffer (child) or of one file (parent, the
// union of its children): three open-addressing tables of 32-byte slots holding the four bloom/v3 base
// hashes of an entry.  Distinctness is decided on all 256 bits.  h0 == 0 marks an empty slot; an entry
// whose own h0..h3 contains a zero word cannot be represented and flags the table (status 2) so the
// caller rebuilds that set on the host path (probability 2^-62 per entry).
#pragma once
#include "kernels.hip.h"

namespace bsg {

constexpr int kIngestThreads = 256;
constexpr uint32_t kPathCap = 200;      // longest path the device walker keeps (bytes)
constexpr uint32_t kMaxDepth = 16;      // container nesting handled on the device
constexpr uint32_t kLaneLds = 228;      // 200 path + 16 stack + pad = 57 dwords (odd stride: lanes spread over LDS banks)
constexpr uint32_t kMaxProbe = 96;      // linear-probe bound before a table is declared full
constexpr uint32_t kSpinLimit = 4096;   // re-reads of a claimed slot whose h1..h3 are still in flight

constexpr uint32_t kTableOk = 0, kTableOverflow = 1, kTableExotic = 2;

struct IngestTable {
    uint64_t *slots;   // [mask + 1][4]
    uint32_t mask;     // capacity - 1 (capacity is a power of two)
    uint32_t pad;
};

// ---------------- streaming murmur3_x64_128 pair (bloom/v3 sum256) ----------------
struct HashStream {
    uint64_t h1, h2, lo, hi;
    uint32_t n;
};

__device__ __forceinline__ void hs_init(HashStream &s) { s.h1 = s.h2 = s.lo = s.hi = 0; s.n = 0; }

__device__ __forceinline__ void hs_absorb(HashStream &s, uint32_t byte)
{
    const uint32_t t = s.n & 15u;
    const uint64_t v = (uint64_t)byte << ((t & 7u) * 8u);
    if (t < 8u) s.lo |= v; else s.hi |= v;
    s.n += 1;
    if (t == 15u) { bmix(s.h1, s.h2, s.lo, s.hi); s.lo = 0; s.hi = 0; }
}

// (h0,h1) = murmur(d), (h2,h3) = murmur(d || 0x01): same tail handling as base_hashes (kernels.hip.h)
__device__ __forceinline__ void hs_finish(const HashStream &s, uint64_t h[4])
{
    const uint32_t t = s.n & 15u;
    uint64_t k1 = s.lo, k2 = s.hi;
    {
        uint64_t a1 = s.h1, a2 = s.h2;
        if (t > 8) mix_k2(a2, k2);
        if (t > 0) mix_k1(a1, k1);
        murmur_finalize(a1, a2, s.n, h[0], h[1]);
    }
    {
        if (t < 8) k1 |= 1ULL << (8 * t);
        else       k2 |= 1ULL << (8 * (t - 8));
        uint64_t b1 = s.h1, b2 = s.h2;
        if (t == 15) {
            bmix(b1, b2, k1, k2);
        } else {
            if (t + 1 > 8) mix_k2(b2, k2);
            mix_k1(b1, k1);
        }
        murmur_finalize(b1, b2, (uint64_t)s.n + 1, h[2], h[3]);
    }
}

// ---------------- distinct set insert ----------------
__device__ __forceinline__ uint64_t ld_agent(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Inserts the entry with base hashes h into table t.
// SIMT note: the loop leaves only when EVERY active lane is done (wave-uniform __ballot exit) and the winner's
// stores of h1..h3 sit inside the loop body.  With a per-lane exit the compiler places the winner's
// "store, then leave" block after the loop — executed once the whole wave has left it — so lanes of the
// same wave racing on one entry never see h1..h3 and insert duplicates (measured: 64 equal rows -> 64 "distinct").
// Every trip does a bounded amount of work (no inner wait).
__device__ __forceinline__ void set_insert(const IngestTable t, const uint64_t h[4], uint32_t *count, uint32_t *status)
{
    bool done = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kTableOk;
    if (!done && (h[0] == 0 || h[1] == 0 || h[2] == 0 || h[3] == 0)) {
        __hip_atomic_store(status, kTableExotic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        done = true;
    }
    uint32_t idx = (uint32_t)(h[1] >> 20) & t.mask;   // h0 is the claim word; index with bits of h1
    uint32_t probes = 0, spins = 0;
    do {
        if (!done) {
            uint64_t *slot = t.slots + (uint64_t)idx * 4;
            uint64_t cur = ld_agent(slot);
            if (cur == 0) {
                uint64_t expected = 0;
                if (__hip_atomic_compare_exchange_strong(slot, &expected, h[0], __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT)) {
                    st_agent(slot + 1, h[1]);
                    st_agent(slot + 2, h[2]);
                    st_agent(slot + 3, h[3]);
                    __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    done = true;
                } else {
                    cur = expected;
                }
            }
            if (!done) {
                bool advance = true;
                if (cur == h[0]) {
                    const uint64_t a = ld_agent(slot + 1), b = ld_agent(slot + 2), c = ld_agent(slot + 3);
                    if (a == h[1] && b == h[2] && c == h[3]) {
                        done = true;                      // duplicate
                        advance = false;
                    } else if ((a == 0 || b == 0 || c == 0) && spins < kSpinLimit) {
                        spins += 1;                       // the claimer's h1..h3 are still in flight: look again
                        advance = false;
                    }                                     // else: a different entry with the same h0
                }
                if (advance) {
                    idx = (idx + 1) & t.mask;
                    if (++probes > kMaxProbe) {
                        __hip_atomic_store(status, kTableOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        done = true;
                    }
                }
            }
        }
    } while (__ballot(!done) != 0ull);
}

// ---------------- row walker ----------------
struct IngestArgs {
    const uint8_t *rows;            // 8-byte aligned, >= 8 readable bytes after the last row
    const uint64_t *row_off;        // [n_rows + 1]
    const uint32_t *set_first_row;  // [n_sets + 1], ascending
    const IngestTable *tables;      // [n_sets_total * 3]
    uint32_t *counts;               // per table
    uint32_t *status;               // per table
    uint32_t *fallback_rows;        // rows the host walker must finish
    uint32_t *n_fallback;
    uint32_t n_rows;
    uint32_t n_sets;
};

struct RowReader {
    const uint8_t *base;
    uint64_t chunk;
    uint64_t chunk_pos;
    __device__ __forceinline__ uint32_t at(uint64_t pos)
    {
        const uint64_t a = pos & ~7ULL;
        if (a != chunk_pos) {
            chunk = *reinterpret_cast<const uint64_t *>(base + a);
            chunk_pos = a;
        }
        return (uint32_t)(chunk >> ((pos & 7u) * 8u)) & 0xFFu;
    }
};

typedef __attribute__((address_space(3))) uint8_t lds_u8;

struct Walker {
    RowReader rd;
    lds_u8 *path;            // this lane's path buffer [kPathCap] followed by the container stack [kMaxDepth]
    const IngestTable *tab;  // the row's set: tables[set * 3 + kind]
    uint32_t *counts;
    uint32_t *status;

    __device__ __forceinline__ void hash_path(uint32_t len, HashStream &s)
    {
        hs_init(s);
        for (uint32_t i = 0; i < len; ++i) hs_absorb(s, path[i]);
    }
    __device__ __forceinline__ void emit_field(uint32_t len)
    {
        HashStream s;
        hash_path(len, s);
        uint64_t h[4];
        hs_finish(s, h);
        set_insert(tab[0], h, counts + 0, status + 0);
    }
    // leaf with text = row bytes [s, e): field entry, then one token + one field::token entry per word
    __device__ __forceinline__ void emit_leaf(uint32_t path_len, uint64_t s, uint64_t e)
    {
        HashStream ps;
        hash_path(path_len, ps);
        uint64_t h[4];
        hs_finish(ps, h);
        set_insert(tab[0], h, counts + 0, status + 0);
        hs_absorb(ps, ':');
        hs_absorb(ps, ':');            // ps = state after path + "::" (makeFieldTokenKey, tokenizer.go:509-511)
        uint64_t p = s;
        while (p < e) {
            while (p < e && rd.at(p) == ' ') ++p;      // only 0x20 can occur: other white space sent the row to the host
            if (p >= e) break;
            HashStream tk, ft = ps;
            hs_init(tk);
            while (p < e) {
                uint32_t c = rd.at(p);
                if (c == ' ') break;
                if (c - 'A' < 26u) c += 32;            // ASCII fold (appendFoldedWord fast path)
                hs_absorb(tk, c);
                hs_absorb(ft, c);
                ++p;
            }
            hs_finish(tk, h);
            set_insert(tab[1], h, counts + 1, status + 1);
            hs_finish(ft, h);
            set_insert(tab[2], h, counts + 2, status + 2);
        }
    }
};

__device__ __forceinline__ bool is_plain(uint32_t c) { return c >= 0x20u && c < 0x7Fu && c != '\\'; }

// Returns false when the host walker must take the row.
__device__ __forceinline__ bool walk_row(Walker &w, uint64_t pos, const uint64_t end)
{
    lds_u8 *stack = w.path + kPathCap;
    uint32_t path_len = 0, depth = 0, objmask = 0;
    enum { VALUE, AFTER, KEY } st = VALUE;
    for (;;) {
        while (pos < end && w.rd.at(pos) == ' ') ++pos;
        if (st == VALUE) {
            if (pos >= end) return false;
            const uint32_t c = w.rd.at(pos);
            if (c == '{' || c == '[') {
                if (path_len > 0) w.emit_field(path_len);       // container with a non-empty path: non-leaf emission
                if (depth >= kMaxDepth) return false;
                stack[depth] = (uint8_t)path_len;
                if (c == '{') objmask |= 1u << depth; else objmask &= ~(1u << depth);
                ++depth;
                ++pos;
                while (pos < end && w.rd.at(pos) == ' ') ++pos;
                if (pos >= end) return false;
                const uint32_t c2 = w.rd.at(pos);
                if (c2 == (c == '{' ? '}' : ']')) { ++pos; --depth; st = AFTER; continue; }
                st = (c == '{') ? KEY : VALUE;
                continue;
            }
            uint64_t s = pos, e = pos;
            bool has_text = true;
            if (c == '"') {
                s = ++pos;
                for (;;) {
                    if (pos >= end) return false;
                    const uint32_t b = w.rd.at(pos);
                    if (b == '"') break;
                    if (!is_plain(b)) return false;
                    ++pos;
                }
                e = pos++;
            } else if (c == 't') {
                if (end - pos < 4 || w.rd.at(pos + 1) != 'r' || w.rd.at(pos + 2) != 'u' || w.rd.at(pos + 3) != 'e') return false;
                pos += 4; e = pos;
            } else if (c == 'f') {
                if (end - pos < 5 || w.rd.at(pos + 1) != 'a' || w.rd.at(pos + 2) != 'l' || w.rd.at(pos + 3) != 's' ||
                    w.rd.at(pos + 4) != 'e') return false;
                pos += 5; e = pos;
            } else if (c == 'n') {
                if (end - pos < 4 || w.rd.at(pos + 1) != 'u' || w.rd.at(pos + 2) != 'l' || w.rd.at(pos + 3) != 'l') return false;
                pos += 4; has_text = false;                     // null: field existence only (tokenizer.go:130-131)
            } else {
                // number: -? digits (. digits)? ([eE] [+-]? digits)?  — the text is the RAW literal (tokenizer.go:124-125)
                if (c == '-') ++pos;
                if (pos >= end || w.rd.at(pos) - '0' > 9u) return false;
                while (pos < end && w.rd.at(pos) - '0' <= 9u) ++pos;
                if (pos < end && w.rd.at(pos) == '.') {
                    ++pos;
                    if (pos >= end || w.rd.at(pos) - '0' > 9u) return false;
                    while (pos < end && w.rd.at(pos) - '0' <= 9u) ++pos;
                }
                if (pos < end && (w.rd.at(pos) | 0x20u) == 'e') {
                    ++pos;
                    if (pos < end && (w.rd.at(pos) == '+' || w.rd.at(pos) == '-')) ++pos;
                    if (pos >= end || w.rd.at(pos) - '0' > 9u) return false;
                    while (pos < end && w.rd.at(pos) - '0' <= 9u) ++pos;
                }
                e = pos;
            }
            if (path_len > 0) {
                if (has_text) w.emit_leaf(path_len, s, e);
                else w.emit_field(path_len);
            }
            st = AFTER;
            continue;
        }
        if (st == AFTER) {
            if (depth == 0) return pos == end;                  // nothing but spaces may follow the row's value
            path_len = stack[depth - 1];
            if (pos >= end) return false;
            const uint32_t c = w.rd.at(pos);
            const bool obj = (objmask >> (depth - 1)) & 1u;
            if (c == ',') { ++pos; st = obj ? KEY : VALUE; continue; }
            if (c == (obj ? '}' : ']')) { ++pos; --depth; st = AFTER; continue; }
            return false;
        }
        // KEY: "key" ':' — child path = parent + "." + key, key-prefix paths first (row_matcher.go:103-135)
        if (pos >= end || w.rd.at(pos) != '"') return false;
        ++pos;
        uint32_t pl = path_len;
        if (pl > 0) {
            if (pl >= kPathCap) return false;
            w.path[pl++] = '.';
        }
        const uint32_t key_start = pl;
        for (;;) {
            if (pos >= end) return false;
            const uint32_t b = w.rd.at(pos);
            if (b == '"') break;
            if (!is_plain(b) || pl >= kPathCap) return false;
            w.path[pl++] = (uint8_t)b;
            ++pos;
        }
        ++pos;
        while (pos < end && w.rd.at(pos) == ' ') ++pos;
        if (pos >= end || w.rd.at(pos) != ':') return false;
        ++pos;
        for (uint32_t j = key_start; j < pl; ++j)
            if (w.path[j] == '.' && j > 0) w.emit_field(j);     // every "."-split prefix of the key, empty paths skipped
        path_len = pl;
        st = VALUE;
    }
}

__global__ __launch_bounds__(kIngestThreads) void k_ingest_rows(const IngestArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t r = blockIdx.x * kIngestThreads + threadIdx.x;
    if (r >= a.n_rows) return;
    // the set this row belongs to: last s with set_first_row[s] <= r
    uint32_t lo = 0, hi = a.n_sets;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.set_first_row[mid] <= r) lo = mid; else hi = mid;
    }
    Walker w;
    w.rd.base = a.rows;
    w.rd.chunk = 0;
    w.rd.chunk_pos = ~0ULL;
    w.path = (lds_u8 *)lds_raw + threadIdx.x * kLaneLds;
    w.tab = a.tables + (uint64_t)lo * 3;
    w.counts = a.counts + (uint64_t)lo * 3;
    w.status = a.status + (uint64_t)lo * 3;
    if (!walk_row(w, a.row_off[r], a.row_off[r + 1])) {
        const uint32_t slot = __hip_atomic_fetch_add(a.n_fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.fallback_rows[slot] = r;
    }
}

// ---------------- host-walked entries (fallback rows) ----------------
__global__ __launch_bounds__(256) void k_ingest_add(const uint8_t *bytes, const uint32_t *off, const uint32_t *table_of_entry,
                                                    uint32_t n, const IngestTable *tables, uint32_t *counts, uint32_t *status)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    uint64_t h[4];
    base_hashes_words(bytes + off[e], off[e + 1] - off[e], h);
    const uint32_t t = table_of_entry[e];
    set_insert(tables[t], h, counts + t, status + t);
}

// ---------------- union / rehash: every occupied slot of src[item] is inserted into dst[item] ----------------
struct UnionItem { uint32_t src, dst; };

__global__ __launch_bounds__(256) void k_ingest_union(const IngestTable *src_tables, const IngestTable *dst_tables,
                                                      const UnionItem *items, uint32_t *dst_counts, uint32_t *dst_status)
{
    const UnionItem it = items[blockIdx.y];
    const IngestTable s = src_tables[it.src];
    const IngestTable d = dst_tables[it.dst];
    const uint64_t cap = (uint64_t)s.mask + 1;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * 256) {
        const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(s.slots + i * 4);
        const ulonglong2 x = p[0];
        if (x.x == 0) continue;
        const ulonglong2 y = p[1];
        const uint64_t h[4] = {x.x, x.y, y.x, y.y};
        set_insert(d, h, dst_counts + it.dst, dst_status + it.dst);
    }
}

// ---------------- bitsets from distinct sets ----------------
struct SetBuildItem {
    uint32_t table;       // source table == filter index (desc[table])
    uint32_t staged;      // 1: whole bitset assembled in LDS by this workgroup
    uint64_t slot_begin, slot_end;
};

struct SetBuildArgs {
    const IngestTable *tables;
    const SetBuildItem *items;
    const DevDesc *desc;
    uint64_t *out;
};

template <bool M32, typename BITS32>
__device__ __forceinline__ void build_from_slots(const IngestTable t, const SetBuildItem &it, const DevDesc &d, BITS32 bits, uint32_t tid)
{
    for (uint64_t i = it.slot_begin + tid; i < it.slot_end; i += kBuildThreads) {
        const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(t.slots + i * 4);
        const ulonglong2 x = p[0];
        if (x.x == 0) continue;
        const ulonglong2 y = p[1];
        const uint64_t h[4] = {x.x, x.y, y.x, y.y};
        set_entry_bits<M32>(bits, d, h);
    }
}

__global__ __launch_bounds__(kBuildThreads) void k_build_sets(const SetBuildArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const SetBuildItem it = a.items[blockIdx.x];
    const DevDesc d = a.desc[it.table];
    const IngestTable t = a.tables[it.table];
    const uint32_t tid = threadIdx.x;
    if (d.m == 0) return;
    const uint64_t nw = (d.m + 63) >> 6;
    const bool m32 = d.m < (1ull << 31);
    if (it.staged) {
        for (uint32_t i = tid; i < nw; i += kBuildThreads) lds64[i] = 0;
        __syncthreads();
        lds_u32 *bits = (lds_u32 *)lds64;
        if (m32) build_from_slots<true>(t, it, d, bits, tid);
        else     build_from_slots<false>(t, it, d, bits, tid);
        __syncthreads();
        uint64_t *dst = a.out + d.word_off;
        for (uint32_t i = tid; i < nw; i += kBuildThreads) dst[i] = lds64[i];
    } else {
        uint32_t *bits = reinterpret_cast<uint32_t *>(a.out + d.word_off);
        if (m32) build_from_slots<true>(t, it, d, bits, tid);
        else     build_from_slots<false>(t, it, d, bits, tid);
    }
}

}  // namespace bsg
