// kernels.hip.h — CDNA4 (gfx950) device code for libbloomgpu.so.
//
// Integer hash + bit manipulation only: no MFMA (the path is HBM-bound byte
// work).  Wavefront = 64 lanes everywhere (ballot masks are 64-bit).
//
// Arithmetic restated from the published algorithm of
//   github.com/bits-and-blooms/bloom/v3 v3.7.0 (murmur.go sum256, bloom.go location)
// as called by the reference at ingest.go:142 (AddString) and
// query_exec.go:141,147,154 (TestString).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsg {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kProbeThreads = 512;
constexpr int kEvalThreads = 256;
constexpr int kBuildThreads = 256;

// Device-side filter descriptor: the public (word_off, m, k) plus the Barrett
// reciprocal magic = floor(2^64 / m) (m == 1 -> 2^64 - 1) so that
// x mod m costs one 64x64 mul-high instead of a software divide.
struct DevDesc {
    uint64_t word_off;  // in u64 words into the shard's word arena (16-byte aligned: even)
    uint64_t m;         // 0 => absent filter
    uint64_t magic;
    uint32_t k;
    uint32_t pad;
};

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

__device__ __forceinline__ uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

constexpr uint64_t kC1 = 0x87c37b91114253d5ULL;
constexpr uint64_t kC2 = 0x4cf5ad432745937fULL;

__device__ __forceinline__ void mix_k1(uint64_t &h1, uint64_t k1)
{
    k1 *= kC1; k1 = rotl64(k1, 31); k1 *= kC2; h1 ^= k1;
}
__device__ __forceinline__ void mix_k2(uint64_t &h2, uint64_t k2)
{
    k2 *= kC2; k2 = rotl64(k2, 33); k2 *= kC1; h2 ^= k2;
}
__device__ __forceinline__ void bmix(uint64_t &h1, uint64_t &h2, uint64_t k1, uint64_t k2)
{
    mix_k1(h1, k1);
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729ULL;
    mix_k2(h2, k2);
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5ULL;
}
__device__ __forceinline__ void murmur_finalize(uint64_t h1, uint64_t h2, uint64_t len, uint64_t &o1, uint64_t &o2)
{
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    o1 = h1; o2 = h2;
}

// bloom/v3 sum256: (h0,h1) = murmur3_x64_128(d), (h2,h3) = murmur3_x64_128(d || 0x01), seed 0,
// computed in one pass without materialising the appended byte.
template <typename BytePtr>
__device__ __forceinline__ void base_hashes(BytePtr p, uint32_t len, uint64_t h[4])
{
    uint64_t h1 = 0, h2 = 0;
    const uint32_t nb = len >> 4;
    for (uint32_t i = 0; i < nb; ++i) {
        uint64_t k1 = 0, k2 = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            k1 |= (uint64_t)p[16 * i + j] << (8 * j);
            k2 |= (uint64_t)p[16 * i + 8 + j] << (8 * j);
        }
        bmix(h1, h2, k1, k2);
    }
    const uint32_t t = len & 15u;
    uint64_t k1 = 0, k2 = 0;
    for (uint32_t j = 0; j < t; ++j) {
        const uint64_t byte = p[16 * nb + j];
        if (j < 8) k1 |= byte << (8 * j);
        else       k2 |= byte << (8 * (j - 8));
    }
    {   // hash of d: tail of t bytes
        uint64_t a1 = h1, a2 = h2;
        if (t > 8) mix_k2(a2, k2);
        if (t > 0) mix_k1(a1, k1);
        murmur_finalize(a1, a2, len, h[0], h[1]);
    }
    {   // hash of d || 0x01: the extra byte lands at tail position t
        if (t < 8) k1 |= 1ULL << (8 * t);
        else       k2 |= 1ULL << (8 * (t - 8));
        uint64_t b1 = h1, b2 = h2;
        if (t == 15) {
            bmix(b1, b2, k1, k2);          // the padded tail is a whole 16-byte block
        } else {
            if (t + 1 > 8) mix_k2(b2, k2);
            mix_k1(b1, k1);
        }
        murmur_finalize(b1, b2, (uint64_t)len + 1, h[2], h[3]);
    }
}

// bloom/v3 location(h, i) = h[i%2] + i*h[2 + (((i + (i%2)) % 4) / 2)]  (wrapping u64).
__device__ __forceinline__ uint64_t location(uint64_t h0, uint64_t h1, uint64_t h2, uint64_t h3, uint32_t i)
{
    const uint64_t ha = (i & 1u) ? h1 : h0;
    const uint32_t r = i & 3u;
    const uint64_t hb = (r == 1u || r == 2u) ? h3 : h2;
    return ha + (uint64_t)i * hb;
}

// x mod m via Barrett: q = mulhi(x, floor(2^64/m)) is floor(x/m) or one less.
__device__ __forceinline__ uint64_t mod_m(uint64_t x, uint64_t m, uint64_t magic)
{
    const uint64_t q = __umul64hi(x, magic);
    uint64_t r = x - q * m;
    if (r >= m) r -= m;
    return r;
}

// 64x64 bit-matrix transpose across a wavefront: lane l holds row l on entry
// and column l on exit (bit r of the result = bit l of lane r's input).
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x, int lane)
{
    const uint64_t masks[6] = {0x00000000FFFFFFFFULL, 0x0000FFFF0000FFFFULL, 0x00FF00FF00FF00FFULL,
                               0x0F0F0F0F0F0F0F0FULL, 0x3333333333333333ULL, 0x5555555555555555ULL};
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        const int s = 32 >> st;
        const uint64_t m = masks[st];
        const uint32_t plo = __shfl_xor((uint32_t)x, s, kWave);
        const uint32_t phi = __shfl_xor((uint32_t)(x >> 32), s, kWave);
        const uint64_t p = ((uint64_t)phi << 32) | plo;
        if ((lane & s) == 0) x = (x & m) | ((p & m) << s);
        else                 x = (x & ~m) | ((p >> s) & m);
    }
    return x;
}

// ---------------------------------------------------------------------------
// K1  probe_terms: one workgroup per (block, referenced filter kind).
// Streams the block's bitset HBM -> LDS once with 16-byte coalesced loads, then
// every lane owns one query term: k location tests against LDS, wave-level
// early-out, __ballot folds 64 verdicts into one u64 that lane 0 stores.
// Verdict layout: V[((b >> 6) * Wt + w) * 64 + (b & 63)], w = 64-term word.
// ---------------------------------------------------------------------------
struct ProbeArgs {
    const uint64_t *words;
    const DevDesc *desc;          // [n_blocks * 3]
    const uint64_t *th;           // SoA term hashes: th[j * Tp + t], j < 4
    uint64_t *V;
    uint32_t Tp;                  // padded term count (multiple of 64)
    uint32_t Wt;                  // Tp / 64
    uint32_t n_blocks;
    uint32_t lds_cap_words;       // filters with more words take the gather path
    uint32_t kind[3];             // referenced kinds, blockIdx.y indexes this
    uint32_t term_begin[3];       // first term (multiple of 64) of that kind
    uint32_t term_count[3];       // real terms of that kind
};

// x mod m for m < 2^31: the remainder candidate x - q*m lies in [0, 2m) so only
// its low 32 bits are needed.
__device__ __forceinline__ uint32_t mod_m32(uint64_t x, uint32_t m, uint64_t magic)
{
    const uint64_t q = __umul64hi(x, magic);
    uint32_t r = (uint32_t)x - (uint32_t)q * m;
    if (r >= m) r -= m;
    return r;
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint64_t lds_u64;

template <bool M32, typename BITS32>
__device__ __forceinline__ bool test_location(BITS32 bits, const DevDesc &d, uint64_t x)
{
    if (M32) {
        const uint32_t loc = mod_m32(x, (uint32_t)d.m, d.magic);
        return (bits[loc >> 5] >> (loc & 31)) & 1u;
    } else {
        const uint64_t loc = mod_m(x, d.m, d.magic);
        return (bits[loc >> 5] >> ((uint32_t)loc & 31)) & 1u;
    }
}

// Mode A (few terms): every wave-task is one (location index i, 64-term word w) pair, so all
// k locations of all terms are tested concurrently — one probe per lane, no serial early-out
// chain.  Pass masks are AND-ed into the LDS verdict words.
template <bool M32, typename BITS32>
__device__ __forceinline__ void probe_parallel_k(const ProbeArgs &a, const DevDesc &d, BITS32 bits, uint32_t t0,
                                                 uint32_t n_real, uint32_t n_tw, lds_u64 *vw, uint32_t wave, uint32_t lane)
{
    constexpr uint32_t n_waves = kProbeThreads / kWave;
    const uint32_t n_tasks = n_tw * d.k;
    for (uint32_t task = wave; task < n_tasks; task += n_waves) {
        const uint32_t i = task / n_tw, w = task - i * n_tw;
        const uint32_t idx = w * 64 + lane;
        const uint32_t t = t0 + idx;
        const uint32_t r = i & 3u;
        const uint64_t ha = a.th[(uint64_t)(i & 1u) * a.Tp + t];
        const uint64_t hb = a.th[(uint64_t)((r == 1u || r == 2u) ? 3u : 2u) * a.Tp + t];
        bool pass = true;
        if (idx < n_real) pass = test_location<M32>(bits, d, ha + (uint64_t)i * hb);
        const uint64_t mask = __ballot(pass);
        if (lane == 0 && mask != ~0ULL) __hip_atomic_fetch_and(&vw[w], mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Mode B (many terms): persistent lanes.  A lane keeps one term in registers and tests one
// location per iteration; the moment a term is rejected (or accepted after k hits) the lane draws
// the next term index from a workgroup-wide LDS ticket (one ds_add per wave per iteration, ranks by
// mbcnt), so all 64 lanes stay busy and the expected work is ~2 probes per absent term instead of
// the wave-wide maximum.  Lanes refilling in the same iteration get consecutive tickets, so their
// hash loads stay coalesced.
template <bool M32, typename BITS32>
__device__ __forceinline__ void probe_persistent(const ProbeArgs &a, const DevDesc &d, BITS32 bits, uint32_t t0,
                                                 uint32_t n_real, lds_u32 *vbits, lds_u32 *ticket, uint32_t lane)
{
    uint64_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    uint32_t idx = 0, i = 0;
    bool busy = false, exhausted = false;
    for (;;) {
        const bool need = !busy && !exhausted;
        const uint64_t need_mask = __ballot(need);
        if (need_mask) {
            uint32_t base = 0;
            if (lane == (uint32_t)__builtin_ctzll(need_mask))
                base = __hip_atomic_fetch_add(ticket, (uint32_t)__builtin_popcountll(need_mask), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WORKGROUP);
            base = __shfl(base, __builtin_ctzll(need_mask), kWave);
            if (need) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need_mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)need_mask, 0u));
                idx = base + rank;
                if (idx < n_real) {
                    const uint32_t t = t0 + idx;
                    h0 = a.th[t]; h1 = a.th[(uint64_t)a.Tp + t];
                    h2 = a.th[2ull * a.Tp + t]; h3 = a.th[3ull * a.Tp + t];
                    i = 0; busy = true;
                } else {
                    exhausted = true;
                }
            }
        }
        if (__ballot(busy) == 0) break;
        if (busy) {
            const bool hit = test_location<M32>(bits, d, location(h0, h1, h2, h3, i));
            if (!hit) {
                busy = false;
            } else if (++i == d.k) {
                __hip_atomic_fetch_or(&vbits[idx >> 5], 1u << (idx & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                busy = false;
            }
        }
    }
}

constexpr uint32_t kParallelKMaxWords = 8;  // term words (x64 terms) up to which mode A is used

template <bool M32, typename BITS32>
__device__ __forceinline__ void probe_block(const ProbeArgs &a, const DevDesc &d, BITS32 bits, uint32_t t0,
                                            uint32_t n_real, uint32_t n_tw, lds_u64 *vw, lds_u32 *ticket,
                                            uint64_t *vout, uint32_t tid)
{
    const uint32_t lane = tid & (kWave - 1), wave = tid / kWave;
    const bool par = n_tw <= kParallelKMaxWords;
    // verdict words start as "all real terms pass" (mode A ANDs failures in) or zero (mode B ORs hits in)
    for (uint32_t w = tid; w < n_tw; w += kProbeThreads) {
        const uint32_t rem = n_real - w * 64;
        vw[w] = par ? (rem >= 64 ? ~0ULL : ((1ULL << rem) - 1)) : 0ULL;
    }
    if (tid == 0) *ticket = 0;
    __syncthreads();
    if (par) probe_parallel_k<M32>(a, d, bits, t0, n_real, n_tw, vw, wave, lane);
    else     probe_persistent<M32>(a, d, bits, t0, n_real, (lds_u32 *)vw, ticket, lane);
    __syncthreads();
    for (uint32_t w = tid; w < n_tw; w += kProbeThreads) vout[(uint64_t)w * 64] = vw[w];
}

__global__ __launch_bounds__(kProbeThreads) void k_probe_terms(const ProbeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const uint32_t b = blockIdx.x;
    const uint32_t y = blockIdx.y;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wave = tid / kWave;
    constexpr uint32_t n_waves = kProbeThreads / kWave;

    const DevDesc d = a.desc[(uint64_t)b * 3 + a.kind[y]];
    const uint32_t t0 = a.term_begin[y];
    const uint32_t n_real = a.term_count[y];
    const uint32_t n_tw = (n_real + 63) >> 6;
    uint64_t *vout = a.V + ((uint64_t)(b >> 6) * a.Wt + (t0 >> 6)) * 64 + (b & 63);

    if (d.m == 0) {  // nil filter: cannot disqualify (query_exec.go:137-151)
        for (uint32_t w = tid; w < n_tw; w += kProbeThreads) vout[(uint64_t)w * 64] = ~0ULL;
        return;
    }
    // LDS carve: [verdict words | ticket | pad to 16 B][bitset image]
    lds_u64 *vw = (lds_u64 *)lds64;
    lds_u32 *ticket = (lds_u32 *)(vw + n_tw);
    const uint32_t head_bytes = (n_tw * 8 + 4 + 15) & ~15u;
    char *image = reinterpret_cast<char *>(lds64) + head_bytes;

    const uint64_t nw = (d.m + 63) >> 6;
    const uint64_t *src = a.words + d.word_off;
    if (nw <= a.lds_cap_words) {
        // HBM -> LDS by LDS-DMA: every wave issues all of its 1 KiB pieces back to back
        // (64 lanes x 16 B, no VGPR round trip), one wait for the whole filter.
        const uint32_t nbytes = (uint32_t)(((nw + 1) >> 1) << 4);
        const char *g = reinterpret_cast<const char *>(src);
        for (uint32_t c = wave * 1024u; c < nbytes; c += n_waves * 1024u) {
            const uint32_t boff = c + lane * 16u;
            if (boff < nbytes)
                __builtin_amdgcn_global_load_lds((glb_void *)(g + boff), (lds_void *)(image + c), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const lds_u32 *lbits = (const lds_u32 *)image;
        if (d.m < (1ull << 31)) probe_block<true>(a, d, lbits, t0, n_real, n_tw, vw, ticket, vout, tid);
        else                    probe_block<false>(a, d, lbits, t0, n_real, n_tw, vw, ticket, vout, tid);
    } else {
        const uint32_t *gbits = reinterpret_cast<const uint32_t *>(src);
        if (d.m < (1ull << 31)) probe_block<true>(a, d, gbits, t0, n_real, n_tw, vw, ticket, vout, tid);
        else                    probe_block<false>(a, d, gbits, t0, n_real, n_tw, vw, ticket, vout, tid);
    }
}

// ---------------------------------------------------------------------------
// K2  eval_programs: workgroup = (group of 64 blocks, chunk of 256 queries).
// Prologue: transpose the group's verdict words so VT[term] is a 64-block mask.
// Body: every lane runs its query's binary postfix program on u64 masks, i.e.
// 64 blocks per bitwise op; result is the survivors word out[q * G + g].
// Internal ops (lowered on the host from the public n-ary form):
//   0 TERM pos | 1 AND2 | 2 OR2 | 3 TRUE | 4 FALSE | 7 NOP
// ---------------------------------------------------------------------------
struct EvalArgs {
    const uint64_t *V;
    const uint32_t *prog;       // chunk c: prog[chunk_off[c] + j * 256 + lane]
    const uint32_t *chunk_off;
    const uint32_t *chunk_len;
    uint64_t *out;              // [n_queries][G]
    uint32_t Wt;
    uint32_t n_blocks;
    uint32_t G;
    uint32_t n_queries;
};

__global__ __launch_bounds__(kEvalThreads) void k_eval_programs(const EvalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const uint32_t g = blockIdx.x;
    const uint32_t c = blockIdx.y;
    const uint32_t tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const uint32_t wave = tid / kWave;
    constexpr uint32_t n_waves = kEvalThreads / kWave;

    uint64_t *VT = lds64;
    uint64_t *stk = lds64 + (uint64_t)a.Wt * 64 + tid;  // per-lane stack, stride kEvalThreads

    const bool row_valid = (g * 64 + (uint32_t)lane) < a.n_blocks;
    for (uint32_t w = wave; w < a.Wt; w += n_waves) {
        uint64_t x = row_valid ? a.V[((uint64_t)g * a.Wt + w) * 64 + lane] : 0ULL;
        VT[w * 64 + lane] = wave_transpose64(x, lane);
    }
    __syncthreads();

    const uint32_t q = c * kEvalThreads + tid;
    const uint32_t len = a.chunk_len[c];
    const uint32_t *P = a.prog + a.chunk_off[c] + tid;
    uint64_t top = ~0ULL;  // empty program == nil query == true
    uint32_t sp = 0;       // number of values on the stack (top kept in a register)
    for (uint32_t j = 0; j < len; ++j) {
        const uint32_t op = P[(uint64_t)j * kEvalThreads];
        const uint32_t opc = op >> 28;
        if (opc == 7u) continue;
        if (opc == 1u || opc == 2u) {
            --sp;
            const uint64_t under = stk[(uint64_t)(sp - 1) * kEvalThreads];
            top = (opc == 1u) ? (under & top) : (under | top);
        } else {
            if (sp > 0) stk[(uint64_t)(sp - 1) * kEvalThreads] = top;
            ++sp;
            top = (opc == 0u) ? VT[op & 0x0FFFFFFFu] : (opc == 3u ? ~0ULL : 0ULL);
        }
    }
    const uint32_t nvalid = a.n_blocks - g * 64;
    const uint64_t valid = nvalid >= 64 ? ~0ULL : ((1ULL << nvalid) - 1);
    if (q < a.n_queries) a.out[(uint64_t)q * a.G + g] = top & valid;
}

// ---------------------------------------------------------------------------
// hash_entries: one lane per entry -> 4 x u64 base hashes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hash_entries(const uint8_t *bytes, const uint32_t *off, uint32_t n,
                                                      uint64_t *out)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    uint64_t h[4];
    base_hashes(bytes + off[e], off[e + 1] - off[e], h);
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(out + (uint64_t)e * 4);
    o[0] = make_ulonglong2(h[0], h[1]);
    o[1] = make_ulonglong2(h[2], h[3]);
}

// ---------------------------------------------------------------------------
// build: one workgroup per build item (a filter, or a slice of a large
// filter's entries).  Small filters are assembled in LDS with ds_or and
// written out once, coalesced; large ones OR straight into zeroed HBM words.
// ---------------------------------------------------------------------------
struct BuildItem {
    uint32_t filter;
    uint32_t e_begin;
    uint32_t e_end;
    uint32_t staged;  // 1: whole filter in LDS (item covers all its entries)
};

struct BuildArgs {
    const uint8_t *bytes;     // may be null when h != null
    const uint32_t *off;
    const uint64_t *h;        // optional precomputed hashes [n][4]
    const BuildItem *items;
    const DevDesc *desc;
    uint64_t *out;
};

__global__ __launch_bounds__(kBuildThreads) void k_build(const BuildArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const BuildItem it = a.items[blockIdx.x];
    const DevDesc d = a.desc[it.filter];
    const uint32_t tid = threadIdx.x;
    if (d.m == 0) return;
    const uint64_t nw = (d.m + 63) >> 6;
    uint32_t *bits;
    if (it.staged) {
        for (uint32_t i = tid; i < nw; i += kBuildThreads) lds64[i] = 0;
        __syncthreads();
        bits = reinterpret_cast<uint32_t *>(lds64);
    } else {
        bits = reinterpret_cast<uint32_t *>(a.out + d.word_off);
    }
    for (uint32_t e = it.e_begin + tid; e < it.e_end; e += kBuildThreads) {
        uint64_t h[4];
        if (a.h) {
            const ulonglong2 *hp = reinterpret_cast<const ulonglong2 *>(a.h + (uint64_t)e * 4);
            const ulonglong2 x = hp[0], y = hp[1];
            h[0] = x.x; h[1] = x.y; h[2] = y.x; h[3] = y.y;
        } else {
            base_hashes(a.bytes + a.off[e], a.off[e + 1] - a.off[e], h);
        }
        for (uint32_t i = 0; i < d.k; ++i) {
            const uint64_t loc = mod_m(location(h[0], h[1], h[2], h[3], i), d.m, d.magic);
            atomicOr(&bits[loc >> 5], 1u << (loc & 31));
        }
    }
    if (it.staged) {
        __syncthreads();
        uint64_t *dst = a.out + d.word_off;
        for (uint32_t i = tid; i < nw; i += kBuildThreads) dst[i] = lds64[i];
    }
}

// dst[i] |= OR over s < n_src of src[s * n_words + i]
__global__ __launch_bounds__(256) void k_or_words(uint64_t *dst, const uint64_t *src, uint64_t n_words,
                                                  uint32_t n_src, int overwrite)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
        uint64_t acc = overwrite ? 0ULL : dst[i];
        for (uint32_t s = 0; s < n_src; ++s) acc |= src[(uint64_t)s * n_words + i];
        dst[i] = acc;
    }
}

// OR of one kind's filter across all blocks of a shard (fixed geometry):
// out[i] = OR_b words[desc[b*3+kind].word_off + i]
__global__ __launch_bounds__(256) void k_or_reduce_blocks(const uint64_t *words, const DevDesc *desc,
                                                          uint32_t n_blocks, uint32_t kind, uint64_t n_words,
                                                          uint64_t *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
        uint64_t acc = 0;
        for (uint32_t b = 0; b < n_blocks; ++b) {
            const DevDesc d = desc[(uint64_t)b * 3 + kind];
            if (d.m != 0) acc |= words[d.word_off + i];
        }
        out[i] = acc;
    }
}

}  // namespace bsg
