// bin_build.hip.h — bitsets too large for LDS (file-level filters: buildSizedBloomFilter over the file's union sets,
// flush.go:253 / merge.go:516; 11 M entries x k = 10 into 158 Mbit at BASELINE configs[2]) without one global atomic
// per bit.  Random 4-byte read-modify-writes into a 20 MB bitset run at ~24 G/s on MI355X (device-scope atomics are
// served behind the per-XCD L2s: every one moves a 64-byte sector both ways), 220 M of them took 8.5 ms.  Instead the
// locations are binned by 64 KiB window of the bitset and every window is assembled in LDS:
//     k_bin_pass<false>   per tile of table slots (or of hashed entries): locations -> LDS histogram over the windows -> one global add per window
//     k_bin_scan          exclusive prefix over the windows (one workgroup)
//     k_bin_pass<true>    the same walk; a tile reserves its run in every window with one global add, then writes the
//                         locations (low 19 bits) into the runs
//     k_bin_apply         one workgroup per window: its locations -> LDS atomic OR -> the window's words, written once
// Traffic is sequential: the slots twice, 4 bytes per location out and in, the bitset once.
#pragma once
#include "ingest.hip.h"

namespace bsg {

constexpr uint32_t kBinWindowShift = 19;                          // 2^19 bits = 64 KiB of bitset per window
constexpr uint32_t kBinWindowBits = 1u << kBinWindowShift;
constexpr uint32_t kBinMaxWindows = 4096;                         // m < 2^31
constexpr uint32_t kBinThreads = 1024;
constexpr uint32_t kBinTile = 4 * kBinThreads;                    // table slots per workgroup

struct BinArgs {
    IngestTable t;          // source set (DENSE: t.slots = the entries' base hashes, 4 words each; nothing else is read)
    DevDesc d;              // the filter (m < 2^31, magic, k, word_off)
    uint64_t n_slots;       // t.mask + 1 (DENSE: entries)
    uint32_t n_windows;
    uint32_t n_locs_cap;    // capacity of locs
    uint32_t *prefix;       // [n_windows + 1]: counts, then (k_bin_scan) exclusive prefix; [n_windows] = total
    uint32_t *cursor;       // [n_windows]: next free position of every window's run (starts at prefix)
    uint32_t *locs;         // [n_locs_cap]: location - window * 2^19, grouped by window
    uint32_t *overflow;     // set when the counts handed in were too small for what the table holds
    uint64_t *out;          // the arena's words
};

// the occupied slots of tile blockIdx.x as a dense list (fingerprint != 0 <=> occupied once the walk has ended)
__device__ __forceinline__ uint32_t bin_tile_list(const BinArgs &a, uint32_t *list, uint32_t *n_list, uint64_t base, uint32_t tid)
{
    if (tid == 0) *n_list = 0;
    __syncthreads();
    uint64_t f[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        const uint64_t i = base + u * kBinThreads + tid;
        f[u] = i < a.n_slots ? a.t.fps[i] : 0;
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        const uint64_t mask = __ballot(f[u] != 0);
        if (mask == 0) continue;
        uint32_t at = 0;
        if ((tid & 63u) == 0u) at = atomicAdd(n_list, (uint32_t)__builtin_popcountll(mask));
        at = __builtin_amdgcn_readfirstlane(at);
        if (f[u] != 0) list[at + lane_rank(mask)] = u * kBinThreads + tid;
    }
    __syncthreads();
    return *n_list;
}

// f(location) for the k locations of one entry, the recurrence of set_entry_bits
template <typename F>
__device__ __forceinline__ void bin_locations(const DevDesc &d, const uint64_t h[4], F f)
{
    for_each_location_x(d.k, h[0], h[1], h[2], h[3], [&](uint64_t x) { f((uint32_t)locate<true>(d, x)); });
}

// DENSE: the source is an array of base hashes, 4 words per entry (bsg_build / bsg_build_hashed), not a table
template <bool SCATTER, bool DENSE>
__global__ __launch_bounds__(kBinThreads) void k_bin_pass(const BinArgs a)
{
    __shared__ uint32_t list[kBinTile];
    __shared__ uint32_t n_list;
    __shared__ uint32_t hist[kBinMaxWindows];
    __shared__ uint32_t run[SCATTER ? kBinMaxWindows : 1];       // where this tile's locations of window w start in locs
    const uint32_t tid = threadIdx.x;
    for (uint32_t w = tid; w < a.n_windows; w += kBinThreads) hist[w] = 0;
    const uint64_t base = (uint64_t)blockIdx.x * kBinTile;
    uint32_t n;
    if (DENSE) {
        n = (uint32_t)(a.n_slots - base < kBinTile ? a.n_slots - base : kBinTile);
        __syncthreads();                                             // the zeroing above
    } else {
        n = bin_tile_list(a, list, &n_list, base, tid);              // (its barriers also cover the zeroing above)
    }
    // this thread's entries stay in registers across both halves (a tile holds <= 4 per thread)
    uint64_t h[4][4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t e = tid + j * kBinThreads;
        if (e < n) {
            const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(a.t.slots + (base + (DENSE ? e : list[e])) * 4);
            const ulonglong2 x = p[0], y = p[1];
            h[j][0] = x.x; h[j][1] = x.y; h[j][2] = y.x; h[j][3] = y.y;
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        if (tid + j * kBinThreads < n) bin_locations(a.d, h[j], [&](uint32_t loc) { atomicAdd(&hist[loc >> kBinWindowShift], 1u); });
    __syncthreads();
    if (!SCATTER) {
        for (uint32_t w = tid; w < a.n_windows; w += kBinThreads)
            if (hist[w]) __hip_atomic_fetch_add(a.prefix + w, hist[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    for (uint32_t w = tid; w < a.n_windows; w += kBinThreads) {
        const uint32_t c = hist[w];
        run[w] = c ? __hip_atomic_fetch_add(a.cursor + w, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        hist[w] = 0;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        if (tid + j * kBinThreads < n)
            bin_locations(a.d, h[j], [&](uint32_t loc) {
                const uint32_t w = loc >> kBinWindowShift;
                const uint32_t pos = run[w] + atomicAdd(&hist[w], 1u);      // any order inside the run will do
                if (pos < a.n_locs_cap) a.locs[pos] = loc & (kBinWindowBits - 1u);
                else *a.overflow = 1u;
            });
}

// prefix[w] = sum of the counts before w, prefix[n_windows] = total, cursor = prefix
__global__ __launch_bounds__(kBinThreads) void k_bin_scan(const BinArgs a)
{
    __shared__ uint32_t part[kBinThreads];
    const uint32_t tid = threadIdx.x;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t w = tid * 4 + u;
        v[u] = w < a.n_windows ? a.prefix[w] : 0u;
        sum += v[u];
    }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t step = 1; step < kBinThreads; step <<= 1) {
        const uint32_t add = tid >= step ? part[tid - step] : 0u;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;                                // exclusive
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
        const uint32_t w = tid * 4 + u;
        if (w < a.n_windows) { a.prefix[w] = run; a.cursor[w] = run; }
        run += v[u];
    }
    if (tid == kBinThreads - 1) a.prefix[a.n_windows] = part[tid];
}

__global__ __launch_bounds__(kBinThreads) void k_bin_apply(const BinArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t win[kBinWindowBits / 32];
    const uint32_t tid = threadIdx.x, w = blockIdx.x;
    for (uint32_t i = tid; i < kBinWindowBits / 32; i += kBinThreads) win[i] = 0;
    __syncthreads();
    const uint32_t lo = a.prefix[w], hi = a.prefix[w + 1] < a.n_locs_cap ? a.prefix[w + 1] : a.n_locs_cap;
    // four independent loads in flight per thread: one workgroup pulls its window's ~1.5 MB alone
    for (uint32_t j = lo + tid; j < hi; j += 4 * kBinThreads) {
        uint32_t l[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) l[u] = j + u * kBinThreads < hi ? a.locs[j + u * kBinThreads] : 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u)
            if (l[u] != 0xFFFFFFFFu) atomicOr(&win[l[u] >> 5], 1u << (l[u] & 31u));
    }
    __syncthreads();
    const uint64_t nw = (a.d.m + 63) >> 6, first = (uint64_t)w * (kBinWindowBits / 64);
    const uint64_t *win64 = reinterpret_cast<const uint64_t *>(win);
    uint64_t *dst = a.out + a.d.word_off + first;
    for (uint32_t i = tid; i < kBinWindowBits / 64 && first + i < nw; i += kBinThreads) dst[i] = win64[i];
}

}  // namespace bsg
