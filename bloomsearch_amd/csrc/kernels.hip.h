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

#include "crc_slices.h"

namespace bsg {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// cache policy of the bitset LDS-DMA (aux field: 0 default, 2 = nt).  Every bitset is streamed once and read by one
// CU, so nt: measured on MI355X 6.29 -> 6.99 TB/s at 16 000 blocks per launch, ~3% at 1 000 (tools/probe_lab.hip).
#ifndef BSG_DMA_AUX
#define BSG_DMA_AUX 2
#endif

// Register budget of the streaming kernels (measured on MI355X, tools/fold_lab.py with -DBSG_LAB_VGPR_PAD): k_probe_terms with
// its SGPR allocation padded past 80 (incl. VCC / XNACK / FLAT_SCRATCH: > 74 numbered) runs 12-17% slower — 104.9 -> 118 us per
// 20 arenas at s79, 122.9 at s95 — although nothing else changes: 800 SGPRs per SIMD / 96 = 8 waves, exactly the four
// 8-wave workgroups the LDS admits per CU, so a new workgroup can only start once a whole old one has retired on every
// SIMD; at <= 80 there is room for 10 waves and the turnover overlaps.  VGPR padding up to 64 costs nothing measurable.
#ifndef BSG_STREAM_SGPRS
#define BSG_STREAM_SGPRS 72
#endif
constexpr int kWave = 64;
constexpr int kProbeThreads = 512;
constexpr int kEvalThreads = 256;
constexpr int kBuildThreads = 512;


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

__host__ __device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

__host__ __device__ __forceinline__ uint64_t fmix64(uint64_t k)
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

__host__ __device__ __forceinline__ void mix_k1(uint64_t &h1, uint64_t k1)
{
    k1 *= kC1; k1 = rotl64(k1, 31); k1 *= kC2; h1 ^= k1;
}
__host__ __device__ __forceinline__ void mix_k2(uint64_t &h2, uint64_t k2)
{
    k2 *= kC2; k2 = rotl64(k2, 33); k2 *= kC1; h2 ^= k2;
}
// 5 h + c as two 64-bit shift-adds (v_lshl_add_u64); spelled as a multiply the compiler makes two v_mad_u64_u32 and two moves of it
__host__ __device__ __forceinline__ uint64_t times5_plus(uint64_t h, uint64_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t h5;
    asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(h5) : "v"(h));
    return h5 + c;
#else
    return h * 5 + c;
#endif
}
__host__ __device__ __forceinline__ void bmix(uint64_t &h1, uint64_t &h2, uint64_t k1, uint64_t k2)
{
    mix_k1(h1, k1);
    h1 = rotl64(h1, 27); h1 += h2; h1 = times5_plus(h1, 0x52dce729ULL);
    mix_k2(h2, k2);
    h2 = rotl64(h2, 31); h2 += h1; h2 = times5_plus(h2, 0x38495ab5ULL);
}
__host__ __device__ __forceinline__ void murmur_finalize(uint64_t h1, uint64_t h2, uint64_t len, uint64_t &o1, uint64_t &o2)
{
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    o1 = h1; o2 = h2;
}

// bloom/v3 sum256: (h0,h1) = murmur3_x64_128(d), (h2,h3) = murmur3_x64_128(d || 0x01), seed 0,
// computed in one pass without materialising the appended byte.  __host__ too: bsg_query hashes an interactive query's
// few terms on the host with this very function (one source for both sides; tests compare them).
template <typename BytePtr>
__host__ __device__ __forceinline__ void base_hashes(BytePtr p, uint32_t len, uint64_t h[4])
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

// Word-wise variant for entries packed in a blob with >= 16 readable bytes after the last entry
// (the library's staging buffers guarantee it): two unaligned 8-byte loads per 16 bytes instead of
// byte loads; bytes past the entry's end are masked off.
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// The tail's shape by table instead of by branch (round 6): kTail[t] = {mask of the low t bytes (two words), the 0x01 of d || 0x01 at
// byte t (two words)}.  The lanes of a wave disagree on t, so every branch on it ran both ways for the whole wave (ISA of k_build);
// 512 bytes that live in the L1, two 16-byte loads issued beside the tail's own.
#define BSG_TAIL_ROW(t) { (t) >= 8 ? ~0ULL : ((1ULL << (8 * ((t) & 7))) - 1), (t) > 8 ? ((1ULL << (8 * ((t) & 7))) - 1) : 0ULL, \
                          (t) < 8 ? 1ULL << (8 * ((t) & 7)) : 0ULL, (t) >= 8 ? 1ULL << (8 * ((t) & 7)) : 0ULL }
__device__ __constant__ static const uint64_t kTail[16][4] = {
    BSG_TAIL_ROW(0), BSG_TAIL_ROW(1), BSG_TAIL_ROW(2), BSG_TAIL_ROW(3), BSG_TAIL_ROW(4), BSG_TAIL_ROW(5), BSG_TAIL_ROW(6), BSG_TAIL_ROW(7),
    BSG_TAIL_ROW(8), BSG_TAIL_ROW(9), BSG_TAIL_ROW(10), BSG_TAIL_ROW(11), BSG_TAIL_ROW(12), BSG_TAIL_ROW(13), BSG_TAIL_ROW(14), BSG_TAIL_ROW(15)};
#undef BSG_TAIL_ROW

// k * c1, rotl 31, * c2 (what mix_k1 XORs into h1) and its twin; zero for k == 0, so an absent tail word needs no branch
__device__ __forceinline__ uint64_t mixed_k1(uint64_t k1) { k1 *= kC1; k1 = rotl64(k1, 31); return k1 * kC2; }
__device__ __forceinline__ uint64_t mixed_k2(uint64_t k2) { k2 *= kC2; k2 = rotl64(k2, 33); return k2 * kC1; }

// The entry's bytes lie at base + o: with a workgroup-uniform base and a 32-bit offset the loads take the saddr + voffset form
// (no 64-bit address arithmetic per lane).  In two halves, so a loop can request the next entry's tail before it hashes this one.
struct EntryTail {
    ulonglong2 tail, mask, one;     // the 16 bytes at the tail's start; kTail[t]
    uint32_t o, len;
};
__device__ __forceinline__ EntryTail load_entry_tail(const uint8_t *base, uint32_t o, uint32_t len)
{
    EntryTail p;
    p.o = o; p.len = len;
    const ulonglong2 *row = reinterpret_cast<const ulonglong2 *>(kTail[len & 15u]);
    p.mask = row[0]; p.one = row[1];
    __builtin_memcpy(&p.tail, base + (o + (len & ~15u)), 16);
    return p;
}
__device__ __forceinline__ void base_hashes_of(const uint8_t *base, const EntryTail &p, uint64_t h[4])
{
    uint64_t h1 = 0, h2 = 0;
    const uint32_t len = p.len, nb = len >> 4, t = len & 15u;
    for (uint32_t i = 0; i < nb; ++i) {
        ulonglong2 blk;
        __builtin_memcpy(&blk, base + (p.o + 16 * i), 16);
        bmix(h1, h2, blk.x, blk.y);
    }
    const uint64_t k1 = p.tail.x & p.mask.x, k2 = p.tail.y & p.mask.y;
    // Below 8 tail bytes neither hash has a second tail word; from 8 on the appended byte leaves the first word alone.  Entries
    // arrive sorted, so a wave usually agrees on the side: the two wave-uniform branches skip 2 of the 4 tail mixes (a mixed wave
    // takes both; a lane on the other side gets the same value again, or mixes zero into zero).
    const bool any_short = __ballot(t < 8) != 0, any_long = __ballot(t >= 8) != 0;
    const uint64_t xa1 = mixed_k1(k1);
    uint64_t xb1 = xa1, xa2 = 0, xb2 = 0;
    if (any_short) xb1 = mixed_k1(k1 | p.one.x);
    if (any_long) { xa2 = mixed_k2(k2); xb2 = mixed_k2(k2 | p.one.y); }
    murmur_finalize(h1 ^ xa1, h2 ^ xa2, len, h[0], h[1]);
    // d || 0x01: the extra byte lands at tail position t; at t == 15 the padded tail is a whole block (bmix) and the tail is empty
    uint64_t b1 = h1 ^ xb1, b2 = h2;
    if (t == 15) {
        b1 = times5_plus(rotl64(b1, 27) + b2, 0x52dce729ULL);
        b2 = times5_plus(rotl64(b2 ^ xb2, 31) + b1, 0x38495ab5ULL);
    } else {
        b2 ^= xb2;
    }
    murmur_finalize(b1, b2, (uint64_t)len + 1, h[2], h[3]);
}
__device__ __forceinline__ void base_hashes_at(const uint8_t *base, uint32_t o, uint32_t len, uint64_t h[4])
{
    base_hashes_of(base, load_entry_tail(base, o, len), h);
}
__device__ __forceinline__ void base_hashes_words(const uint8_t *p, uint32_t len, uint64_t h[4]) { base_hashes_at(p, 0u, len, h); }

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
// One launch probes a GROUP of arenas (the candidate files of one query stage, bsg_probe_many): the per-arena
// pointers ride in the kernel arguments, so a group costs one dispatch ramp instead of one per arena — at 35 MB
// per arena the ~3-4 us ramp + completion of a dispatch is as long as the streaming itself.
// 128 x 24 B of per-arena records + the scalar arguments stay inside the 4 KB of kernel arguments (round 4: the records were 40 B
// and a dispatch held 64 arenas; where the arenas are small — a file's shard on one of 8 GPUs is 125 blocks — twice as many
// per dispatch halve the dispatch ramps and boundaries per step).
constexpr uint32_t kMaxGroupArenas = 128;
constexpr uint32_t kMaxFusedArenas = 8;       // k_probe_fused carries TWO tables (the group it streams, the group it evaluates)
constexpr uint32_t kMaxRowsArenas = 64;       // k_survivor_rows carries a destination table as well: it takes a group in runs of 64

// An arena of a dispatch group.  Where its verdict words and its survivors lie in the group's scratch follows from ONE number,
// the 64-block groups of the arenas in front of it: v_off = g_prefix x max(Wt, 1) x 64 words, out_off = g_prefix x n_queries words.
struct ArenaRef {
    const uint64_t *words;
    const DevDesc *desc;          // [n_blocks * 3]
    uint32_t n_blocks;
    uint32_t g_prefix;            // 64-block groups of the group's arenas before this one
    __host__ __device__ uint32_t G() const { return (n_blocks + 63u) >> 6; }
    __host__ __device__ uint64_t v_off(uint32_t Wt) const { return (uint64_t)g_prefix * (Wt ? Wt : 1u) * 64u; }
    __host__ __device__ uint64_t out_off(uint32_t n_queries) const { return (uint64_t)g_prefix * n_queries; }
};
static_assert(sizeof(ArenaRef) == 24, "ArenaRef must stay 24 bytes: 128 of them ride in the kernel arguments");

// the per-arena records of a dispatch group: a kernel argument of its own, next to the scalars
template <uint32_t N>
struct ArenaTable { ArenaRef ar[N]; };
// Groups beyond kMaxGroupArenas (many small arenas: the candidate files of a wide query, a file's shards on one of 8 GPUs) carry
// their records in DEVICE memory instead (`ext`, uploaded in front of the dispatch on the same stream): one dispatch then
// covers up to kMaxExtGroupArenas arenas and its ramp is paid once.  ext == nullptr: the records in the kernel arguments.
// The two forms are separate KERNELS: a select between the kernel-argument table and a global pointer inside one kernel turned
// the record's scalar loads into vector loads (generic pointers) and cost k_probe_terms 10-14 % at every group size.
constexpr uint32_t kMaxExtGroupArenas = 4096;

struct ProbeArgs {
    const uint64_t *th;           // SoA term hashes: th[j * Tp + t], j < 4
    uint64_t *V;
    uint32_t Tp;                  // padded term count (multiple of 64)
    uint32_t Wt;                  // Tp / 64
    uint32_t lds_cap_words;       // filters with more words take the gather path
    uint32_t gather_cost;         // a filter is gathered (not staged) when terms * k * gather_cost < its bytes
    uint32_t lds_image_bytes;     // LDS reserved for the bitset image (the verdict words and wave queues follow it)
    uint32_t compact_rounds;      // many-term mode: locations 1..R run as compaction rounds, the rest from registers
    uint32_t n_arenas;
    uint32_t max_blocks;          // most blocks of any arena of the group (grid x)
    uint32_t kind[3];             // referenced kinds, blockIdx.y indexes this
    uint32_t term_begin[3];       // first term (multiple of 64) of that kind
    uint32_t term_count[3];       // real terms of that kind
    uint64_t seq;                 // k_probe_eval: this launch's number, the tag of its verdict entries
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

// x mod m for 64 <= m <= 2^19 through the fp64 pipe (round 6; v_fma_f64 issues at full rate on gfx950, tools/ubench_valu.hip):
//   y = xh * T + xl,  T = 2^32 mod m            == x (mod m) and < 2^51: ONE v_mad_u64_u32
//   Y = as_double(0x433 << 52 | y)              = 2^52 + y exactly: one v_or_b32 on the high word
//   t = fma(Y, inv, C)                          inv = K 2^-53 with K = floor(2^53 / m) - 1 (so inv < 1/m, and 2^52 inv is a multiple of
//                                               1/2), C = 2^52 - 1/2 - 2^52 inv — both exact doubles, so the exact value of the fma is
//                                               2^52 + y inv - 1/2, in [2^52 - 1/2, 2^53): it is rounded ONCE, to an integer, and the low
//                                               word of its mantissa is q = F or F - 1 (F = floor(y / m)): y inv < y/m and frac(y/m) <=
//                                               1 - 1/m give q <= F; y inv > y/m - y 2^-52 > y/m - 1/2 gives q >= F - 1 (q = -1 included)
//   r = lo32(y) - q m  in [0, 2m)               one v_mad_u64_u32 (by 2^32 - m), one unsigned-min fix-up
// 6 VALU instructions against the 13 the compiler makes of the 64x64 Barrett quotient (ISA of k_build, round 6).  Exact: nothing
// approximate survives into r.  Both constants come out of the descriptor's magic = floor(2^64 / m) without a division.
// (Measured and dropped, round 6: the fix-up traded for an exact quotient — yd = Y - (2^52 - m), the smallest double above 1/m as
// the reciprocal, the low mantissa word is floor(y/m) + 1 always — with the result (x mod m) - m addressing the staged bitset from
// its top and a funnel shift in the copy-out: 9 instructions per location instead of 10, bit-exact in the whole suite, and on one
// box C3 builds in 355.4-356.9 us against 355.9-356.7: the v_add_f64 it needs costs what the subtract and the minimum did.
// A first version of it, positions in [1, m] with one spare bit, was wrong for x mod m < m/2 with x < m — the sum falls below 2^52,
// where doubles step by 1/2 — and tests/test_modulo_edges_gpu.py said so at once.)
struct ModF64 {
    uint32_t T;        // 2^32 mod m
    uint32_t negm;     // 2^32 - m
    double inv;        // (floor(2^53 / m) - 1) 2^-53
    double C;          // 2^52 - 1/2 - 2^52 inv
};
constexpr uint64_t kModF64MaxM = 1ull << 19;
__host__ __device__ __forceinline__ bool modf64_ok(uint64_t m) { return m >= 64 && m <= kModF64MaxM; }
__host__ __device__ __forceinline__ ModF64 make_modf64(uint64_t m, uint64_t magic)
{
    ModF64 f;
    f.negm = 0u - (uint32_t)m;
    f.T = (uint32_t)(magic >> 32) * f.negm;                      // 2^32 - floor(2^32 / m) m   (mod 2^32; < m)
    const uint64_t K = (magic >> 11) - 1;                        // floor(floor(2^64 / m) / 2^11) = floor(2^53 / m);  K < 2^47
    f.inv = (double)K * 0x1p-53;
    f.C = 0x1p52 - (double)(K + 1) * 0.5;
    return f;
}
// (__host__ too: tests/modf64_check.hip sweeps it on the CPU against % — the arithmetic is IEEE fma on both sides)
__host__ __device__ __forceinline__ uint32_t mod_f64(uint64_t x, const ModF64 &f)
{
    const uint64_t Y = ((uint64_t)(uint32_t)(x >> 32) * f.T + (uint32_t)x) | 0x4330000000000000ull;
    const uint32_t q = (uint32_t)__builtin_bit_cast(uint64_t, __builtin_fma(__builtin_bit_cast(double, Y), f.inv, f.C));
    const uint32_t r = (uint32_t)((uint64_t)q * f.negm + Y);
    const uint32_t r2 = r - (0u - f.negm);
    return r2 < r ? r2 : r;                      // unsigned minimum: r - m wraps far above r exactly when r < m
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint64_t lds_u64;
typedef __attribute__((address_space(3))) uint16_t lds_u16;

// MODE: how a location index becomes a bit position (0 / 1 are also what the older `bool M32` instantiations convert to)
constexpr int kModBarrett64 = 0, kModBarrett32 = 1, kModFp64 = 2;
__device__ __forceinline__ int mod_mode(uint64_t m) { return modf64_ok(m) ? kModFp64 : m < (1ull << 31) ? kModBarrett32 : kModBarrett64; }

// A probe workgroup's filter: the descriptor plus the fp64 route's constants (filled when mod_mode(d.m) == kModFp64)
struct ProbeDesc : DevDesc {
    ModF64 f;
};

// location index -> bit position in the filter
template <int MODE>
__device__ __forceinline__ uint64_t locate(const DevDesc &d, uint64_t x)
{
    static_assert(MODE != kModFp64, "the fp64 route needs a ProbeDesc");
    if (MODE == kModBarrett32) return mod_m32(x, (uint32_t)d.m, d.magic);
    return mod_m(x, d.m, d.magic);
}
template <int MODE>
__device__ __forceinline__ uint64_t locate(const ProbeDesc &d, uint64_t x)
{
    if (MODE == kModFp64) return mod_f64(x, d.f);
    if (MODE == kModBarrett32) return mod_m32(x, (uint32_t)d.m, d.magic);
    return mod_m(x, d.m, d.magic);
}
template <typename BITS32>
__device__ __forceinline__ bool test_bit(BITS32 bits, uint64_t loc)
{
    return (bits[loc >> 5] >> ((uint32_t)loc & 31u)) & 1u;
}
// hashes feeding location(h, i) for a wave-uniform i: (h[i%2], h[2 + (((i + i%2) % 4) / 2)])
__device__ __forceinline__ uint32_t ha_row(uint32_t i) { return i & 1u; }
__device__ __forceinline__ uint32_t hb_row(uint32_t i) { const uint32_t r = i & 3u; return (r == 1u || r == 2u) ? 3u : 2u; }

// A filter descriptor at a workgroup-uniform address, through the CONSTANT address space (descriptors are written by the arena
// load / section decode, never by a probe): scalar loads into SGPRs whatever the pointer's provenance — a descriptor pointer that
// itself came out of memory (the device-memory arena table) would otherwise be chased with vector loads.
__device__ __forceinline__ DevDesc load_desc_uniform(const DevDesc *p)
{
    typedef const __attribute__((address_space(4))) uint64_t c64;
    c64 *q = (c64 *)(uintptr_t)p;
    DevDesc d;
    d.word_off = q[0]; d.m = q[1]; d.magic = q[2];
    const uint64_t kp = q[3];
    d.k = (uint32_t)kp; d.pad = (uint32_t)(kp >> 32);
    return d;
}

constexpr uint32_t kProbeWaves = kProbeThreads / kWave;
constexpr uint32_t kParallelKMaxWords = 2;  // term words (x64 terms) up to which mode A is used

// Mode A (<= 128 terms): every wave-task is one (location index i, 64-term word w) pair, so all
// k locations of all terms are tested concurrently — one probe per lane, no serial early-out
// chain.  The hash loads and the modulo of a wave's first task are issued BEFORE the workgroup
// waits for the bitset DMA; only the LDS bit test sits behind the barrier.
struct ParTask { uint64_t loc; uint32_t w; bool real; };

template <int M32>
__device__ __forceinline__ ParTask par_task_prepare(const ProbeArgs &a, const ProbeDesc &d, uint32_t t0, uint32_t n_real,
                                                    uint32_t n_tw, uint32_t task, uint32_t lane)
{
    const uint32_t i = task / n_tw, w = task - i * n_tw;
    const uint32_t idx = w * 64 + lane;
    const uint32_t t = t0 + idx;
    const uint64_t ha = a.th[(uint64_t)ha_row(i) * a.Tp + t];
    const uint64_t hb = a.th[(uint64_t)hb_row(i) * a.Tp + t];
    ParTask p;
    p.loc = locate<M32>(d, ha + (uint64_t)i * hb);
    p.w = w;
    p.real = idx < n_real;
    return p;
}

template <typename BITS32>
__device__ __forceinline__ void par_task_finish(const ParTask &p, BITS32 bits, lds_u64 *vw, uint32_t lane)
{
    const bool pass = !p.real || test_bit(bits, p.loc);
    const uint64_t mask = __ballot(pass);
    if (lane == 0 && mask != ~0ULL) __hip_atomic_fetch_and(&vw[p.w], mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Mode C (many terms).  Each wave owns a contiguous range of term words and runs
//   round 0 : location 0 of every term (8 B/term, coalesced), survivors compacted into the wave's
//             private LDS queue (ballot + mbcnt; no atomics, no barriers);
//   rounds 1..R : one location per round for the survivors (two gathered hash rows), compacted in place;
//   tail    : what is left keeps all four hashes in registers and runs the remaining locations with a
//             wave-level early-out (no further table loads).
// Absent terms die geometrically (~2 probes each); a present term costs all k.  The mode is bound by VALU issue — at
// 4 054 terms over 1 000 blocks of 282 kbit the round-1 version spent 39 VALU instructions per 64-lane probe, of which the
// Barrett reduction is 12 (tools/ubench_mod.hip: 18 ns per wave-probe per SIMD; an FP64 quotient estimate is slower, 22 ns) —
// so this version is written for instruction count: the wave index is made scalar (readfirstlane) so that chunk bounds,
// loop trips and the queue length live in SGPRs and every "is this chunk in range" test is a scalar branch; the
// padded tail of the last term word is probed like real terms (its verdict bits are never referenced) instead of being
// masked per lane; the bitset image sits at LDS offset 0 so a bit address is two shifts; the 64x64 mul-high is spelled
// out in 32-bit pieces.
// Round 4, 4 054 terms x 64 arenas of 1 000 blocks (tools/fold_lab.py 64 needle), k_probe_terms_many per launch:
//   kGroup 4 / rounds 2 (round 3's setting) 1 109 us; kGroup 8: 1 166 (wider groups do NOT help: the mode is not waiting on
//   the hash loads); kGroup 1: 1 115; 3: 1 097; 2: 1 076; kGroup 2 with 1 / 3 compaction rounds: 1 059 / 1 156; tail batches
//   of 2 / 7 locations: 1 109 / 1 115.
#ifndef BSG_KGROUP
#define BSG_KGROUP 2
#endif
#ifndef BSG_TAIL_BATCH
#define BSG_TAIL_BATCH 4
#endif
constexpr uint32_t kGroup = BSG_KGROUP;          // chunks of 64 terms whose loads / reductions / LDS reads are in flight together (lab: -DBSG_KGROUP)
constexpr uint32_t kTailBatch = BSG_TAIL_BATCH;
#ifndef BSG_TAIL_SPLIT
#define BSG_TAIL_SPLIT 0          // lab: 1 = the many-term tail in two parts with a compaction between.  Measured round 5 (needle batch, 4 054 terms, 64 arenas
                                  // per launch): 1 191.6 vs 1 113.8 us — the second gather of the survivors' hashes and the extra ballots cost more than
                                  // the lanes the second part saves (profiles/r05_probe_needle.txt).  The structural attempts on this kernel end here.
#endif
#ifndef BSG_COMPACT_ROUNDS
#define BSG_COMPACT_ROUNDS 1
#endif

// x mod m for m < 2^31 from the halves of x and of magic = floor(2^64 / m): only the low 32 bits of the quotient matter.
__device__ __forceinline__ uint32_t mod_m32_parts(uint32_t xl, uint32_t xh, uint32_t m, uint32_t ml, uint32_t mh)
{
    const uint32_t t1 = __umulhi(xl, ml);
    const uint64_t c = (uint64_t)xh * ml + t1;
    const uint64_t dd = (uint64_t)xl * mh + (uint32_t)c;
    const uint32_t qlo = xh * mh + (uint32_t)(c >> 32) + (uint32_t)(dd >> 32);
    const uint32_t r = xl - qlo * m;
    return min(r, r - m);   // r in [0, 2m): r - m wraps far above r exactly when r < m
}
template <int M32>
__device__ __forceinline__ uint64_t locate_c(const ProbeDesc &d, uint64_t x)
{
    if (M32 == kModFp64) return mod_f64(x, d.f);
    if (M32 == kModBarrett32) return mod_m32_parts((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)d.m, (uint32_t)d.magic, (uint32_t)(d.magic >> 32));
    return mod_m(x, d.m, d.magic);
}
// One bit of the filter as 0 / 1.  The staged image sits at LDS offset 0, so its words are addressed by plain integers
// (a pointer derived from the dynamic-LDS symbol costs a v_add of the link-time base, which is 0, per access).
__device__ __forceinline__ uint32_t bit_at(const lds_u32 *, uint64_t loc)
{
    const lds_u32 *w = (const lds_u32 *)(uintptr_t)(((uint32_t)loc >> 3) & ~3u);
    return __builtin_amdgcn_ubfe(*w, (uint32_t)loc, 1u);   // v_bfe_u32 takes the offset from bits [4:0]
}
__device__ __forceinline__ uint32_t bit_at(const uint32_t *bits, uint64_t loc)
{
    return (bits[loc >> 5] >> ((uint32_t)loc & 31u)) & 1u;
}
// global load with a uniform base and a 32-bit per-lane byte offset: selects the saddr + voffset form (no 64-bit address math)
__device__ __forceinline__ uint64_t load_u64_at(const uint64_t *base, uint32_t byte_off)
{
    return *reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ uint32_t lane_rank(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <int M32, uint32_t NW, typename BITS32>
__device__ __forceinline__ void probe_rounds(const ProbeArgs &a, const ProbeDesc &d, BITS32 bits, uint32_t t0,
                                             uint32_t n_tw, lds_u16 *queues, lds_u32 *vbits,
                                             uint32_t wave, uint32_t lane, const uint64_t (&loc_first)[kGroup])
{
    // wave, n_tw, t0 are wave-uniform (scalar registers): so is everything derived from them
    const uint32_t wpw = (n_tw + NW - 1) / NW;
    const uint32_t w0 = wave * wpw;
    if (w0 >= n_tw) return;
    const uint32_t w1 = min(n_tw, w0 + wpw);
    const uint32_t nw = w1 - w0;                 // term words of this wave
    lds_u16 *q = queues + (uint32_t)w0 * 64;
    const uint32_t base = w0 * 64;
    const uint64_t *th = a.th + t0 + base;       // this wave's slice of hash row 0
    uint32_t qn = 0;                             // queue length (scalar)
    const uint32_t compact_rounds = a.compact_rounds;
    const uint32_t lane8 = lane * 8u;

    // ---- round 0: groups of kGroup chunks; the hashes of group g + 1 are requested before group g is worked on, so the
    // wave sees one L2 round trip for the whole round instead of one per group ----
    uint64_t hn[kGroup];
#pragma unroll
    for (uint32_t u = 0; u < kGroup; ++u) hn[u] = (kGroup + u < nw) ? load_u64_at(th + (kGroup + u) * 64, lane8) : 0;
    for (uint32_t c0 = 0; c0 < nw; c0 += kGroup) {
        uint64_t loc[kGroup];
        if (c0 == 0) {
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) loc[u] = loc_first[u];   // computed before the DMA wait
        } else {
            uint64_t h[kGroup];
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) h[u] = hn[u];
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) hn[u] = (c0 + kGroup + u < nw) ? load_u64_at(th + (c0 + kGroup + u) * 64, lane8) : 0;
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) loc[u] = locate_c<M32>(d, h[u]);
        }
        uint32_t hit[kGroup];
#pragma unroll
        for (uint32_t u = 0; u < kGroup; ++u) hit[u] = bit_at(bits, loc[u]);     // loc is always in range; chunks past nw are dropped below
#pragma unroll
        for (uint32_t u = 0; u < kGroup; ++u) {
            if (c0 + u < nw) {                                                    // scalar branch
                const uint64_t mask = __ballot(hit[u] != 0u);
                if (hit[u] != 0u) q[qn + lane_rank(mask)] = (uint16_t)((c0 + u) * 64 + lane);
                qn += (uint32_t)__builtin_popcountll(mask);
            }
        }
    }
    // ---- rounds 1..R: one location per round, survivors compacted in place ----
    uint32_t i = 1;
    for (; i < d.k && i <= compact_rounds && qn != 0; ++i) {
        const uint64_t *ra = th + (uint64_t)ha_row(i) * a.Tp, *rb = th + (uint64_t)hb_row(i) * a.Tp;
        uint32_t out = 0;
        for (uint32_t j = 0; j < qn; j += 64 * kGroup) {
            uint32_t li[kGroup];
            uint64_t x[kGroup];
            bool live[kGroup];
            uint32_t hit[kGroup];
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) {
                live[u] = j + u * 64 + lane < qn;
                li[u] = live[u] ? (uint32_t)q[j + u * 64 + lane] : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u)
                x[u] = (j + u * 64 < qn) ? load_u64_at(ra, li[u] * 8u) + (uint64_t)i * load_u64_at(rb, li[u] * 8u) : 0;
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) hit[u] = live[u] ? bit_at(bits, locate_c<M32>(d, x[u])) : 0u;
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) {
                if (j + u * 64 < qn) {                                            // scalar; out <= j: in place is safe
                    const uint64_t mask = __ballot(hit[u] != 0u);
                    if (hit[u] != 0u) q[out + lane_rank(mask)] = (uint16_t)li[u];
                    out += (uint32_t)__builtin_popcountll(mask);
                }
            }
        }
        qn = out;
    }
    // ---- tail: remaining locations from registers, two chunks interleaved, wave-level early-out ----
    if (i < d.k && qn != 0) {
        const uint64_t *r1 = th + (uint64_t)a.Tp, *r2 = th + 2ull * a.Tp, *r3 = th + 3ull * a.Tp;
#if BSG_TAIL_SPLIT
        // (lab) The tail in TWO parts.  What reaches it is the present terms plus the false positives of the first locations
        // (needle batch: ~150 of a wave's 512 terms, a quarter of them present); every further location halves the false positives,
        // the present terms go through all of them.  So the first kTailBatch locations run over the whole queue, the survivors —
        // now mostly present terms — are compacted in place, and only they pay for the remaining locations (their four hashes are
        // gathered again: a quarter of the entries).  One wave-level pass of ballots buys the second half of the tail ~4 x fewer lanes.
        if (i + kTailBatch < d.k && qn > 64) {
            uint32_t out = 0;
            for (uint32_t j = 0; j < qn; j += 128) {
                const bool two = j + 64 < qn;                                         // scalar
                bool aliveA = j + lane < qn, aliveB = two && (j + 64 + lane < qn);
                const uint32_t liA = aliveA ? (uint32_t)q[j + lane] : 0u, liB = aliveB ? (uint32_t)q[j + 64 + lane] : 0u;
                const uint64_t a0 = load_u64_at(th, liA * 8u), a1 = load_u64_at(r1, liA * 8u), a2 = load_u64_at(r2, liA * 8u), a3 = load_u64_at(r3, liA * 8u);
                uint64_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
                if (two) { b0 = load_u64_at(th, liB * 8u); b1 = load_u64_at(r1, liB * 8u); b2 = load_u64_at(r2, liB * 8u); b3 = load_u64_at(r3, liB * 8u); }
                uint32_t alla = 1u, allb = 1u;
#pragma unroll
                for (uint32_t v = 0; v < kTailBatch; ++v) {
                    const uint32_t iv = i + v;                                        // uniform, < d.k
                    const bool odd = iv & 1u, use3 = ((iv & 3u) == 1u) | ((iv & 3u) == 2u);
                    const uint64_t xa = (odd ? a1 : a0) + (uint64_t)iv * (use3 ? a3 : a2);
                    alla &= bit_at(bits, locate_c<M32>(d, xa));
                    if (two) {
                        const uint64_t xb = (odd ? b1 : b0) + (uint64_t)iv * (use3 ? b3 : b2);
                        allb &= bit_at(bits, locate_c<M32>(d, xb));
                    }
                }
                aliveA = aliveA & (alla != 0u);
                aliveB = aliveB & (allb != 0u);
                const uint64_t mA = __ballot(aliveA);                                 // (out <= j: in place is safe)
                if (aliveA) q[out + lane_rank(mA)] = (uint16_t)liA;
                out += (uint32_t)__builtin_popcountll(mA);
                if (two) {
                    const uint64_t mB = __ballot(aliveB);
                    if (aliveB) q[out + lane_rank(mB)] = (uint16_t)liB;
                    out += (uint32_t)__builtin_popcountll(mB);
                }
            }
            qn = out;
            i += kTailBatch;
            if (qn == 0) return;
        }
#endif
        const uint32_t i0 = i;
        for (uint32_t j = 0; j < qn; j += 128) {
            const bool two = j + 64 < qn;                                         // scalar
            bool aliveA = j + lane < qn, aliveB = two && (j + 64 + lane < qn);
            const uint32_t liA = aliveA ? (uint32_t)q[j + lane] : 0u, liB = aliveB ? (uint32_t)q[j + 64 + lane] : 0u;
            const uint64_t a0 = load_u64_at(th, liA * 8u), a1 = load_u64_at(r1, liA * 8u), a2 = load_u64_at(r2, liA * 8u), a3 = load_u64_at(r3, liA * 8u);
            uint64_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            if (two) { b0 = load_u64_at(th, liB * 8u); b1 = load_u64_at(r1, liB * 8u); b2 = load_u64_at(r2, liB * 8u); b3 = load_u64_at(r3, liB * 8u); }
            // the remaining locations in batches of kTailBatch: all reductions first, then all LDS reads in flight together,
            // then one AND — a wave pays one LDS round trip per batch instead of one per location.  (A present term — about
            // two in five lanes here — keeps its wave alive through every location anyway; the wave-level early-out only
            // fires between batches.)
            for (uint32_t ii = i0; ii < d.k; ii += kTailBatch) {
                if ((__ballot(aliveA) | __ballot(aliveB)) == 0) break;
                uint32_t ba[kTailBatch], bb[kTailBatch];
#pragma unroll
                for (uint32_t v = 0; v < kTailBatch; ++v) {
                    const uint32_t iv = ii + v;                                   // uniform
                    ba[v] = bb[v] = 1u;
                    if (iv < d.k) {
                        const bool odd = iv & 1u, use3 = ((iv & 3u) == 1u) | ((iv & 3u) == 2u);
                        const uint64_t xa = (odd ? a1 : a0) + (uint64_t)iv * (use3 ? a3 : a2);
                        ba[v] = bit_at(bits, locate_c<M32>(d, xa));
                        if (two) {
                            const uint64_t xb = (odd ? b1 : b0) + (uint64_t)iv * (use3 ? b3 : b2);
                            bb[v] = bit_at(bits, locate_c<M32>(d, xb));
                        }
                    }
                }
                uint32_t alla = 1u, allb = 1u;
#pragma unroll
                for (uint32_t v = 0; v < kTailBatch; ++v) { alla &= ba[v]; allb &= bb[v]; }
                aliveA = aliveA & (alla != 0u);
                aliveB = aliveB & (allb != 0u);
            }
            if (aliveA) __hip_atomic_fetch_or(&vbits[(base + liA) >> 5], 1u << ((base + liA) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (aliveB) __hip_atomic_fetch_or(&vbits[(base + liB) >> 5], 1u << ((base + liB) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    // every location passed during the compaction rounds (k <= R + 1): the queue holds the accepted terms
    for (uint32_t j = lane; j < qn; j += 64) {
        const uint32_t idx = base + q[j];
        __hip_atomic_fetch_or(&vbits[idx >> 5], 1u << (idx & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// ONLY_PAR: the kernel for batches of <= 128 terms per kind carries no many-term code at all (k_probe_terms); sharing one
// kernel cost the 29-term C2 probe 9% in a grouped launch and 25% in a single one (5.03 -> 5.47 us, 7.9 -> 9.9 us per
// 1 000 blocks) through nothing but register allocation and scheduling around code it never runs.
// TAGGED (k_probe_eval): a verdict word leaves as one 16-byte agent-scope (write-through, sc1) store {word, seq ^ mix(word)},
// seq = the launch's number.  The evaluators of the same dispatch — on other XCDs — poll the entry until its tag matches:
// no counter, no atomic, no fence on the probe side.  (Measured alternatives: a release fence per workgroup = buffer_wbl2,
// a walk of the whole L2: 1.17 ms for a 0.12 ms launch; one returning atomic per workgroup on a per-tile counter: 0.27-0.39 ms —
// 256 same-address device-scope atomics per tile serialise at ~50 ns each.)  The tag is keyed with the word, so a reader that
// saw the two halves of an entry from different stores (were a 16-byte store ever torn) rejects it unless the words are equal.
constexpr uint64_t kTagMix = 0x9E3779B97F4A7C15ULL;   // odd: w -> w * kTagMix is a bijection
__device__ __forceinline__ void store_tagged(uint64_t *p, uint64_t v, uint64_t seq)
{
    const uint64_t tag = seq ^ (v * kTagMix);
    const u32x4 d = {(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)tag, (uint32_t)(tag >> 32)};
#ifdef BSG_LAB_PLAIN_V    // lab only (with BSG_LAB_FOLD_SKIP): a write-back store, invisible to the other XCDs — what the write-through costs
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(d) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(d) : "memory");
#endif
}
// -> true when the entry at p belongs to launch `seq`; v = its word.
// gfx942 / gfx950 ONLY: this pair is inline assembly, not the memory model — an `sc1` (agent-scope, write-through)
// 16-byte store on one side, an `sc1` load that bypasses the non-coherent caches on the other, and the assumption that a 16-byte
// store is observed whole (the tag is keyed with the word, so a torn entry is rejected unless both halves agree).  No fence orders
// anything else around it; nothing else needs ordering (an entry carries its own validity).  k_probe_eval, the only user, is a lab
// path (bsg_set_lab key 11, off); on another architecture or compiler it must be re-derived, not recompiled.
__device__ __forceinline__ bool load_tagged(const uint64_t *p, uint64_t seq, uint64_t &v)
{
    u32x4 d;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(d) : "v"(p) : "memory");
    v = ((uint64_t)d.y << 32) | d.x;
    const uint64_t tag = ((uint64_t)d.w << 32) | d.z;
    return (tag ^ (v * kTagMix)) == seq;
}
template <bool TAGGED>
__device__ __forceinline__ void store_verdict(uint64_t *vout, uint32_t w, uint64_t v, uint64_t seq)
{
    if (TAGGED) store_tagged(vout + (uint64_t)w * 128, v, seq);      // 16-byte entries: twice the stride
    else vout[(uint64_t)w * 64] = v;
}

template <int M32, bool STAGED, uint32_t NT, bool ONLY_PAR, bool VSC1 = false>
__device__ __forceinline__ void probe_block(const ProbeArgs &a, const ProbeDesc &d, const uint64_t *src, char *image,
                                            uint32_t t0, uint32_t n_real, uint32_t n_tw, lds_u64 *vw, lds_u16 *queues,
                                            uint64_t *vout, uint32_t tid)
{
    constexpr uint32_t kProbeThreads = NT, kProbeWaves = NT / kWave;   // shadow the 512-thread defaults
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWave);   // scalar: chunk bounds and loop trips stay in SGPRs
    const bool par = ONLY_PAR || n_tw <= kParallelKMaxWords;
    const uint32_t n_tasks = n_tw * d.k;
    // ---- work that does not need the bitset: runs while the DMA is in flight ----
    for (uint32_t w = tid; w < n_tw; w += kProbeThreads) {
        const uint32_t rem = n_real - w * 64;
        vw[w] = par ? (rem >= 64 ? ~0ULL : ((1ULL << rem) - 1)) : 0ULL;   // A: AND failures in; C: OR hits in
    }
    // mode A: the first kParPre tasks of every wave (hash loads + reductions) are done while the bitset is still on its way
    // (measured per 20 arenas, C2 / the C4 batch: 1 task 103.1 / 115.7 us, 2 tasks 101.7 / 114.0, 3 tasks 103.8 / 115.6; round 6: the
    // third task's hash LOADS alone ahead of the barrier, C4 per 200 arenas on one box: 1 109-1 114 vs 1 100-1 112 us without)
#ifndef BSG_PAR_PRE
#define BSG_PAR_PRE 2
#endif
    constexpr uint32_t kParPre = BSG_PAR_PRE;
    ParTask first[kParPre] = {};
    uint64_t loc_first[kGroup] = {};
    if (par) {
#pragma unroll
        for (uint32_t u = 0; u < kParPre; ++u)
            if (wave + u * kProbeWaves < n_tasks) first[u] = par_task_prepare<M32>(a, d, t0, n_real, n_tw, wave + u * kProbeWaves, lane);
    } else if (!ONLY_PAR) {
        const uint32_t wpw = (n_tw + kProbeWaves - 1) / kProbeWaves;
        const uint32_t w0 = wave * wpw, w1 = min(n_tw, w0 + wpw);
        uint64_t h[kGroup];
#pragma unroll
        for (uint32_t u = 0; u < kGroup; ++u) h[u] = (w0 + u < w1) ? a.th[t0 + (w0 + u) * 64 + lane] : 0;
#pragma unroll
        for (uint32_t u = 0; u < kGroup; ++u) loc_first[u] = locate_c<M32>(d, h[u]);
    }
    if (STAGED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- bit tests ----
    if (STAGED) {
        const lds_u32 *bits = (const lds_u32 *)image;
        if (par) {
#pragma unroll
            for (uint32_t u = 0; u < kParPre; ++u)
                if (wave + u * kProbeWaves < n_tasks) par_task_finish(first[u], bits, vw, lane);
            for (uint32_t task = wave + kParPre * kProbeWaves; task < n_tasks; task += kProbeWaves)
                par_task_finish(par_task_prepare<M32>(a, d, t0, n_real, n_tw, task, lane), bits, vw, lane);
        } else if (!ONLY_PAR) {
            probe_rounds<M32, kProbeWaves>(a, d, bits, t0, n_tw, queues, (lds_u32 *)vw, wave, lane, loc_first);
        }
    } else {
        const uint32_t *bits = reinterpret_cast<const uint32_t *>(src);
        if (par) {
#pragma unroll
            for (uint32_t u = 0; u < kParPre; ++u)
                if (wave + u * kProbeWaves < n_tasks) par_task_finish(first[u], bits, vw, lane);
            for (uint32_t task = wave + kParPre * kProbeWaves; task < n_tasks; task += kProbeWaves)
                par_task_finish(par_task_prepare<M32>(a, d, t0, n_real, n_tw, task, lane), bits, vw, lane);
        } else if (!ONLY_PAR) {
            probe_rounds<M32, kProbeWaves>(a, d, bits, t0, n_tw, queues, (lds_u32 *)vw, wave, lane, loc_first);
        }
    }
    __syncthreads();
    for (uint32_t w = tid; w < n_tw; w += kProbeThreads) store_verdict<VSC1>(vout, w, vw[w], a.seq);
}

// LDS carve of k_probe_terms: [bitset image: a.lds_image_bytes, at offset 0 so a bit address needs no base add]
//                             [verdict words n_tw x 8 B][wave queues n_tw x 64 x 2 B]
__host__ __device__ inline uint32_t probe_lds_head_bytes(uint32_t n_tw)
{
    return (n_tw * 8u + n_tw * 128u + 15u) & ~15u;
}

template <uint32_t NT = kProbeThreads, bool ONLY_PAR = false, bool VSC1 = false>
__device__ __forceinline__ void probe_role(const ProbeArgs &a, const ArenaRef &ar, uint32_t b, uint32_t y, uint64_t *lds64)
{
    constexpr uint32_t kProbeThreads = NT, kProbeWaves = NT / kWave;   // shadow the 512-thread defaults
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wave = tid / kWave;

    if (b >= ar.n_blocks) return;   // arenas of a group may differ in size
    ProbeDesc d;
    static_cast<DevDesc &>(d) = load_desc_uniform(ar.desc + ((uint64_t)b * 3 + a.kind[y]));
    const uint32_t t0 = a.term_begin[y];
    const uint32_t n_real = a.term_count[y];
    const uint32_t n_tw = (n_real + 63) >> 6;
    // (k_probe_eval's verdict entries are 16 bytes: word + tag)
    uint64_t *vout = a.V + (ar.v_off(a.Wt) + ((uint64_t)(b >> 6) * a.Wt + (t0 >> 6)) * 64 + (b & 63)) * (VSC1 ? 2 : 1);

    if (d.m == 0) {  // nil filter: cannot disqualify (query_exec.go:137-151)
        for (uint32_t w = tid; w < n_tw; w += kProbeThreads) store_verdict<VSC1>(vout, w, ~0ULL, a.seq);
        return;
    }
    char *image = reinterpret_cast<char *>(lds64);
    lds_u64 *vw = (lds_u64 *)(lds64 + (a.lds_image_bytes >> 3));
    lds_u16 *queues = (lds_u16 *)(vw + n_tw);

    const uint64_t nw = (d.m + 63) >> 6;
    const uint64_t *src = ar.words + d.word_off;
    // The fp64 route only where the probe is bound by instruction issue — the many-term kernels (ONLY_PAR == false).  Measured on one
    // box, round 6 (tools/ab_probe_mod.sh, Barrett vs fp64 in every probe kernel): needle batch (4 054 terms, k_probe_terms_many)
    // 373-377 -> 362-364 us per 20 arenas; C2 103.6-103.8 -> 102.5-103.0; C4 (77 terms, two tasks per wave behind the barrier)
    // 1 112-1 122 -> 1 128-1 153 us: the few-term kernels stream, and there the route's set-up per workgroup costs what it saves.
#ifdef BSG_LAB_PROBE_FP64          // lab only: the fp64 route in the few-term kernels too
    const int mode = mod_mode(d.m);
#else
    const int mode = ONLY_PAR ? (d.m < (1ull << 31) ? kModBarrett32 : kModBarrett64) : mod_mode(d.m);
#endif
    d.f = ModF64{};
    if (mode == kModFp64) d.f = make_modf64(d.m, d.magic);
    // Few probes against a large bitset (a single query, Q = 1): touching <= terms * k sectors straight from L2 / HBM
    // beats streaming the whole bitset into LDS (the "gather regime" of SURVEY 8d).
    const bool gather = (uint64_t)n_real * d.k * a.gather_cost < nw * 8;
    if (nw <= a.lds_cap_words && !gather) {
        // HBM -> LDS by LDS-DMA: every wave issues all of its 1 KiB pieces back to back
        // (64 lanes x 16 B, no VGPR round trip); the single wait sits in probe_block.
        const uint32_t nbytes = (uint32_t)(((nw + 1) >> 1) << 4);
        const char *g = reinterpret_cast<const char *>(src);
        for (uint32_t c = wave * 1024u; c < nbytes; c += kProbeWaves * 1024u) {
            const uint32_t boff = c + lane * 16u;
            if (boff < nbytes)
                __builtin_amdgcn_global_load_lds((glb_void *)(g + boff), (lds_void *)(image + c), 16, 0, BSG_DMA_AUX);
        }
        if (mode == kModFp64)           probe_block<kModFp64, true, NT, ONLY_PAR, VSC1>(a, d, src, image, t0, n_real, n_tw, vw, queues, vout, tid);
        else if (mode == kModBarrett32) probe_block<kModBarrett32, true, NT, ONLY_PAR, VSC1>(a, d, src, image, t0, n_real, n_tw, vw, queues, vout, tid);
        else                            probe_block<kModBarrett64, true, NT, ONLY_PAR, VSC1>(a, d, src, image, t0, n_real, n_tw, vw, queues, vout, tid);
    } else {
        // (gathered and beyond-LDS filters: the fp64 route's range ends far below them; small gathered ones take Barrett)
        if (mode != kModBarrett64) probe_block<kModBarrett32, false, NT, ONLY_PAR, VSC1>(a, d, src, image, t0, n_real, n_tw, vw, queues, vout, tid);
        else                       probe_block<kModBarrett64, false, NT, ONLY_PAR, VSC1>(a, d, src, image, t0, n_real, n_tw, vw, queues, vout, tid);
    }
}

// grid = (max_blocks, referenced kinds, arenas of the group)
// k_probe_terms: batches whose kinds all have <= 128 distinct terms (mode A only); k_probe_terms_many: everything else.
__global__ __launch_bounds__(kProbeThreads) void k_probe_terms(const ProbeArgs a, const ArenaTable<kMaxGroupArenas> t)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
#ifdef BSG_LAB_VGPR_PAD      // lab only: the same code with a larger register allocation (does the allocation size itself cost time?)
#define BSG_STR2(x) #x
#define BSG_STR(x) BSG_STR2(x)
    asm volatile("" ::: BSG_STR(BSG_LAB_VGPR_PAD));     // -DBSG_LAB_VGPR_PAD=s63 / v31 ...
#endif
    probe_role<kProbeThreads, true>(a, t.ar[blockIdx.z], blockIdx.x, blockIdx.y, lds64);
}
// (no SGPR cap here: the many-term mode is bound by VALU issue, not by workgroup turnover — capped at 72 it spills and runs 1.4% slower)
__global__ __launch_bounds__(kProbeThreads) void k_probe_terms_many(const ProbeArgs a, const ArenaTable<kMaxGroupArenas> t)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    probe_role<kProbeThreads, false>(a, t.ar[blockIdx.z], blockIdx.x, blockIdx.y, lds64);
}
// Writes up to kMaxGroupArenas records of a device-memory table from the KERNEL ARGUMENTS: the records are captured when the
// launch is enqueued, so no host buffer has to outlive an asynchronous call (a hipMemcpyAsync would read its source later).
__global__ __launch_bounds__(kMaxGroupArenas) void k_write_arena_table(ArenaRef *dst, const ArenaTable<kMaxGroupArenas> t, uint32_t n)
{
    if (threadIdx.x < n) dst[threadIdx.x] = t.ar[threadIdx.x];
}

// One record of a device-memory table, fetched through the CONSTANT address space: the table is written before the dispatch and
// never during it, the index is workgroup-uniform — so the fields arrive by scalar loads into SGPRs, as they do from the kernel
// arguments (a plain global load put them into VGPRs: the LDS-DMA lost its scalar base and the probe 10 %).
__device__ __forceinline__ ArenaRef load_arena_ref(const ArenaRef *ext, uint32_t i)
{
    typedef const __attribute__((address_space(4))) uint64_t c64;
    c64 *p = (c64 *)(uintptr_t)(ext + i);
    ArenaRef r;
    r.words = (const uint64_t *)p[0];
    r.desc = (const DevDesc *)p[1];
    const uint64_t nb_gp = p[2];
    r.n_blocks = (uint32_t)nb_gp;
    r.g_prefix = (uint32_t)(nb_gp >> 32);
    return r;
}
// the same two kernels for groups whose arena records lie in device memory
__global__ __launch_bounds__(kProbeThreads) void k_probe_terms_ext(const ProbeArgs a, const ArenaRef *__restrict__ ext)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const ArenaRef ar = load_arena_ref(ext, blockIdx.z);
    probe_role<kProbeThreads, true>(a, ar, blockIdx.x, blockIdx.y, lds64);
}
__global__ __launch_bounds__(kProbeThreads) void k_probe_terms_many_ext(const ProbeArgs a, const ArenaRef *__restrict__ ext)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const ArenaRef ar = load_arena_ref(ext, blockIdx.z);
    probe_role<kProbeThreads, false>(a, ar, blockIdx.x, blockIdx.y, lds64);
}

// (Measured and dropped, twice now: the same kernel with 1 024 threads per block — 16 waves sharing one block's image, 32
// waves per CU instead of 24 — runs the 4 054-term batch in 28.6 us per 1 000 blocks against 19.0 us: each wave then owns
// half as many term words, and its fixed per-round costs and queue traffic stay.)

// ---------------------------------------------------------------------------
// K2  eval_programs: workgroup = (group of 64 blocks, chunk of 256 queries).
// Prologue: transpose the verdict words this chunk's programs reference (host-built
// per-chunk word list) so VT[slot * 64 + bit] is a 64-block mask of one term.
// Body: every lane runs its query's binary postfix program on u64 masks, i.e.
// 64 blocks per bitwise op; result is the survivors word out[q * G + g].
// Internal ops (lowered on the host from the public n-ary form):
//   0 TERM (slot * 64 + bit) | 1 AND2 | 2 OR2 | 3 TRUE | 4 FALSE | 7 NOP
// ---------------------------------------------------------------------------
struct EvalArgs {
    const uint64_t *V;
    const uint32_t *prog;        // chunk c, op j, lane l: prog[(c * Lmax + j) * 256 + l]  (uniform stride: no offset table to chase)
    const uint32_t *chunk_len;   // ops actually used by chunk c (<= Lmax; the rest is NOP padding)
    const uint32_t *cw;          // chunk c needs verdict words cw[c * max_cw .. + cw_cnt[c])  (padding entries are 0)
    const uint32_t *cw_cnt;
    uint64_t *out;               // arena i: [n_queries][G_i] at out + ar[i].out_off
    uint32_t Wt;
    uint32_t n_queries;
    uint32_t max_cw;             // max words of any chunk (LDS carve + cw stride)
    uint32_t Lmax;               // max program length of any chunk (prog stride)
    uint32_t identity_cw;        // bit 0: every chunk uses words 0..max_cw-1 in order (small batches): no list to load; bit 1 (k_eval_programs
                                 // only): take eval_role_all — the launch's LDS holds the transposed words of a whole tile
    uint32_t n_arenas;
    uint32_t max_G;              // most 64-block groups of any arena of the launch
};

// LDS bytes one 256-query half needs: transposed verdict words + per-lane stack
__host__ __device__ inline uint32_t eval_lds_bytes(uint32_t max_cw, uint32_t max_depth)
{
    return (max_cw * 64u + max_depth * (uint32_t)kEvalThreads) * 8u;
}

#ifndef BSG_EVAL_GROUP_TILE
#define BSG_EVAL_GROUP_TILE 4
#endif
constexpr uint32_t kEvalGroupTile = BSG_EVAL_GROUP_TILE;   // block groups one eval "half" handles back to back (=> 32-byte stores per lane; lab: -DBSG_EVAL_GROUP_TILE)
// Measured at 64 arenas per launch (C2 / the 8-term C4 batch): tile 4: 35.7 / 55.9 us, tile 8: 35.0 / 60.4, tile 16 (whole 128-byte
// rows per lane): 44.5 / 77.5.  Of the 35.7 us the survivor stores are 17 (33 MB as 32-byte pieces of 128-byte rows) and the
// programs 8; of the 55.9 us of the C4 batch the 15-op programs are 26.  Non-temporal stores for the survivors: 110 / 117 us
// (four workgroups complete every 128-byte row at different times; only a write-back L2 can merge them).

// Evaluates chunk c (256 queries) against block groups [g0, g0 + gt) with the 256 threads htid = 0..255
// of one "half" (a whole k_eval_programs workgroup, or half of a fused workgroup).  The program words
// are fetched once and reused for every group; the gt survivor words of a query are stored together.
// `active` = false halves only take part in the barriers.
__device__ __forceinline__ void eval_role(const EvalArgs &a, const ArenaRef &ar, uint32_t g0, uint32_t gt, uint32_t c, uint32_t htid,
                                          uint64_t *lds, bool active)
{
    const uint64_t *V = a.V + ar.v_off(a.Wt);
    const int lane = htid & (kWave - 1);
    const uint32_t wave = htid / kWave;
    constexpr uint32_t n_waves = kEvalThreads / kWave;
    uint64_t *VT = lds;
    uint64_t *stk = lds + (uint64_t)a.max_cw * 64 + htid;  // per-lane stack, stride kEvalThreads
    constexpr uint32_t kPre = 8;
    uint32_t pre[kPre];
    uint32_t len = 0, cw0 = 0, ncw = 0;
    const uint32_t *P = a.prog;
    if (active) {
        // Everything that depends only on the chunk index is requested at once — program words (the padded
        // layout makes their addresses computable without a table lookup), the chunk's op count and its word
        // list — so the only dependent global round trip left is the verdict words themselves.
        P = a.prog + (uint64_t)c * a.Lmax * kEvalThreads + htid;
#pragma unroll
        for (uint32_t j = 0; j < kPre; ++j) pre[j] = j < a.Lmax ? P[(uint64_t)j * kEvalThreads] : (7u << 28);
        len = a.chunk_len[c];
        cw0 = c * a.max_cw;
        ncw = a.identity_cw ? a.max_cw : a.cw_cnt[c];
    }
    const uint32_t q = c * kEvalThreads + htid;
    uint64_t res[kEvalGroupTile];
    // this wave's first verdict word of a group (s = wave) is requested one group ahead: the words of group t + 1 are in
    // flight while the programs of group t run, instead of one dependent round trip per group
    auto first_word = [&](uint32_t g) -> uint64_t {
        if (!active || wave >= ncw || (g * 64 + (uint32_t)lane) >= ar.n_blocks) return 0ULL;
        const uint32_t w = a.identity_cw ? wave : a.cw[cw0 + wave];
        return V[((uint64_t)g * a.Wt + w) * 64 + lane];
    };
    uint64_t xpre = first_word(g0);
#pragma unroll
    for (uint32_t t = 0; t < kEvalGroupTile; ++t) {
        if (t < gt) {   // workgroup-uniform
            const uint32_t g = g0 + t;
            if (t > 0) __syncthreads();   // everyone is done reading VT of the previous group
            if (active) {
                const bool row_valid = (g * 64 + (uint32_t)lane) < ar.n_blocks;
                if (wave < ncw) VT[wave * 64 + lane] = wave_transpose64(xpre, lane);
                for (uint32_t s = wave + n_waves; s < ncw; s += n_waves) {
                    const uint32_t w = a.identity_cw ? s : a.cw[cw0 + s];
                    uint64_t x = row_valid ? V[((uint64_t)g * a.Wt + w) * 64 + lane] : 0ULL;
                    VT[s * 64 + lane] = wave_transpose64(x, lane);
                }
            }
            if (t + 1 < gt) xpre = first_word(g + 1);
            __syncthreads();
            uint64_t top = ~0ULL;  // empty program == nil query == true
            if (active) {
                uint32_t sp = 0;   // number of values on the stack (top kept in a register)
                auto step = [&](uint32_t op) {
                    const uint32_t opc = op >> 28;
                    if (opc == 7u) return;
                    if (opc == 1u || opc == 2u) {
                        --sp;
                        const uint64_t under = stk[(uint64_t)(sp - 1) * kEvalThreads];
                        top = (opc == 1u) ? (under & top) : (under | top);
                    } else {
                        if (sp > 0) stk[(uint64_t)(sp - 1) * kEvalThreads] = top;
                        ++sp;
                        top = (opc == 0u) ? VT[op & 0x0FFFFFFFu] : (opc == 3u ? ~0ULL : 0ULL);
                    }
                };
#pragma unroll
                for (uint32_t j = 0; j < kPre; ++j) if (j < len) step(pre[j]);
                // longer programs: the next kPre op words requested together (one load per trip, each waited for, was a round trip
                // to L2 per op: the C4 batch's 8-term Or is 15 ops)
                for (uint32_t j0 = kPre; j0 < len; j0 += kPre) {
                    uint32_t more[kPre];
#pragma unroll
                    for (uint32_t u = 0; u < kPre; ++u) more[u] = j0 + u < len ? P[(uint64_t)(j0 + u) * kEvalThreads] : (7u << 28);
#pragma unroll
                    for (uint32_t u = 0; u < kPre; ++u) if (j0 + u < len) step(more[u]);
                }
            }
            const uint32_t nvalid = ar.n_blocks - g * 64;
            res[t] = top & (nvalid >= 64 ? ~0ULL : ((1ULL << nvalid) - 1));
        }
    }
    if (active && q < a.n_queries) {
        uint64_t *dst = a.out + ar.out_off(a.n_queries) + (uint64_t)q * ar.G() + g0;
#pragma unroll
        for (uint32_t t = 0; t < kEvalGroupTile; ++t) if (t < gt) dst[t] = res[t];
    }
}

// The same evaluation for batches whose chunks all reference the batch's few verdict words in order (identity word list: C2, C4,
// interactive batches): the words of ALL gt groups are requested at once, transposed by the waves in parallel and parked in
// LDS behind ONE barrier, then the programs of the gt groups run back to back.  eval_role pays a dependent global round trip,
// two barriers and a serial transpose per group (round 4: 15.5 -> see profiles/r04 per 20 arenas of C2).
// LDS: gt x max_cw transposed words + the per-lane stacks of tile-mask entries (eval_lds_bytes(max_cw * tile, depth * tile)).
// TILE: masks a lane carries per op (the tile's groups): TILE, or 2 for arenas of <= 2 groups (a file's shard on one of 8
// GPUs is 125 blocks: with 4 masks half of every op, stack entry and store was spent on groups that do not exist — round 5)
template <uint32_t TILE>
__device__ __forceinline__ void eval_role_all(const EvalArgs &a, const ArenaRef &ar, uint32_t g0, uint32_t gt, uint32_t c, uint32_t tid, uint64_t *lds)
{
    const uint64_t *V = a.V + ar.v_off(a.Wt);
    const uint32_t lane = tid & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    constexpr uint32_t n_waves = kEvalThreads / kWave;
    const uint32_t ncw = a.max_cw;
    uint64_t *VT = lds;                                                          // VT[(t * ncw + s) * 64 + bit]
    uint64_t *stk = lds + (uint64_t)TILE * ncw * 64 + tid;             // per-lane stack of TILE-mask entries, stride kEvalThreads
    constexpr uint32_t kPre = 8;
    uint32_t pre[kPre];
    const uint32_t *P = a.prog + (uint64_t)c * a.Lmax * kEvalThreads + tid;
#pragma unroll
    for (uint32_t j = 0; j < kPre; ++j) pre[j] = j < a.Lmax ? P[(uint64_t)j * kEvalThreads] : (7u << 28);
    const uint32_t len = a.chunk_len[c];
    // every row this wave transposes, requested before the first transpose starts
    constexpr uint32_t kRows = 4;                                                // rows per wave per trip
    const uint32_t n_rows = gt * ncw;
    for (uint32_t r0 = wave; r0 < n_rows; r0 += n_waves * kRows) {
        uint64_t x[kRows];
#pragma unroll
        for (uint32_t u = 0; u < kRows; ++u) {
            const uint32_t idx = r0 + u * n_waves;
            x[u] = 0ULL;
            if (idx < n_rows) {
                const uint32_t tt = idx / ncw, s = idx - tt * ncw, g = g0 + tt;
                if (g * 64u + lane < ar.n_blocks) x[u] = V[((uint64_t)g * a.Wt + s) * 64 + lane];
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < kRows; ++u) {
            const uint32_t idx = r0 + u * n_waves;
            if (idx < n_rows) VT[(uint64_t)idx * 64 + lane] = wave_transpose64(x[u], (int)lane);
        }
    }
    __syncthreads();
    const uint32_t q = c * kEvalThreads + tid;
    // every op is decoded ONCE and applied to the masks of all the tile's groups (the groups past gt hold zeros: harmless)
    uint64_t res[TILE];
#pragma unroll
    for (uint32_t tt = 0; tt < TILE; ++tt) res[tt] = ~0ULL;            // empty program == nil query == true
    uint32_t sp = 0;
    const uint32_t gstride = ncw * 64;
    auto step = [&](uint32_t op) {
        const uint32_t opc = op >> 28;
        if (opc == 7u) return;
        if (opc == 1u || opc == 2u) {
            --sp;
            const uint64_t *u = stk + (uint64_t)(sp - 1) * TILE * kEvalThreads;
#pragma unroll
            for (uint32_t tt = 0; tt < TILE; ++tt) {
                const uint64_t under = u[(uint64_t)tt * kEvalThreads];
                res[tt] = (opc == 1u) ? (under & res[tt]) : (under | res[tt]);
            }
        } else {
            if (sp > 0) {
                uint64_t *u = stk + (uint64_t)(sp - 1) * TILE * kEvalThreads;
#pragma unroll
                for (uint32_t tt = 0; tt < TILE; ++tt) u[(uint64_t)tt * kEvalThreads] = res[tt];
            }
            ++sp;
            if (opc == 0u) {
                const uint64_t *vt = VT + (op & 0x0FFFFFFFu);
#pragma unroll
                for (uint32_t tt = 0; tt < TILE; ++tt) res[tt] = tt < gt ? vt[(uint64_t)tt * gstride] : 0ULL;
            } else {
                const uint64_t v = opc == 3u ? ~0ULL : 0ULL;
#pragma unroll
                for (uint32_t tt = 0; tt < TILE; ++tt) res[tt] = v;
            }
        }
    };
#pragma unroll
    for (uint32_t j = 0; j < kPre; ++j) if (j < len) step(pre[j]);
    for (uint32_t j0 = kPre; j0 < len; j0 += kPre) {        // (the next kPre op words together: see eval_role)
        uint32_t more[kPre];
#pragma unroll
        for (uint32_t u = 0; u < kPre; ++u) more[u] = j0 + u < len ? P[(uint64_t)(j0 + u) * kEvalThreads] : (7u << 28);
#pragma unroll
        for (uint32_t u = 0; u < kPre; ++u) if (j0 + u < len) step(more[u]);
    }
#pragma unroll
    for (uint32_t tt = 0; tt < TILE; ++tt) {
        if (tt < gt) {
            const uint32_t nvalid = ar.n_blocks - (g0 + tt) * 64u;
            res[tt] &= nvalid >= 64 ? ~0ULL : ((1ULL << nvalid) - 1);
        }
    }
    if (q < a.n_queries) {
        uint64_t *dst = a.out + ar.out_off(a.n_queries) + (uint64_t)q * ar.G() + g0;
#pragma unroll
        for (uint32_t tt = 0; tt < TILE; ++tt) if (tt < gt) dst[tt] = res[tt];
    }
}

// 1-D grid of eval_grid_blocks(...) workgroups; tile (1 or kEvalGroupTile) is chosen by the host.
// XCD-aware numbering: a query's survivor row is written by nx = ceil(max_G / tile) workgroups, tile x 8 bytes each.  Block b
// runs on XCD b % 8 and the XCDs' L2s do not merge each other's lines, so with the tile index varying fastest the four
// 32-byte pieces of every 128-byte row left four different L2s as four partial-line writes.  Here the nx workgroups of one
// (chunk, arena) are 8 apart in the block numbering: same XCD, dispatched within 8 nx blocks of each other, so the row is
// whole in one L2 before it is written back.  (A speed affinity only: any placement gives the same bits.)
__host__ __device__ inline uint32_t eval_grid_blocks(uint32_t nx, uint32_t n_chunks, uint32_t n_arenas)
{
    return (n_chunks * n_arenas + 7u) / 8u * 8u * nx;
}
__device__ __forceinline__ bool eval_block_of(const EvalArgs &a, uint32_t nx, uint32_t n_chunks, uint32_t &combo, uint32_t &tx)
{
    const uint32_t L = blockIdx.x, span = 8u * nx;
    const uint32_t r = L % span;
    combo = L / span * 8u + (r & 7u);
    tx = r >> 3;
    return combo < n_chunks * a.n_arenas;
}
__device__ __forceinline__ void eval_block(const EvalArgs &a, const ArenaRef &ar, uint32_t tile, uint32_t tx, uint32_t c, uint64_t *lds64)
{
    const uint32_t g0 = tx * tile;
    if (g0 >= ar.G()) return;
    if (a.identity_cw & 2u) {
        if (tile == 2u) eval_role_all<2>(a, ar, g0, min(tile, ar.G() - g0), c, threadIdx.x, lds64);
        else eval_role_all<kEvalGroupTile>(a, ar, g0, min(tile, ar.G() - g0), c, threadIdx.x, lds64);
    }
    else eval_role(a, ar, g0, min(tile, ar.G() - g0), c, threadIdx.x, lds64, true);
}
__global__ __launch_bounds__(kEvalThreads) void k_eval_programs(const EvalArgs a, const ArenaTable<kMaxGroupArenas> t, const uint32_t tile, const uint32_t nx,
                                                                const uint32_t n_chunks)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    uint32_t combo, tx;
    if (!eval_block_of(a, nx, n_chunks, combo, tx)) return;
    eval_block(a, t.ar[combo / n_chunks], tile, tx, combo % n_chunks, lds64);
}
__global__ __launch_bounds__(kEvalThreads) void k_eval_programs_ext(const EvalArgs a, const ArenaRef *__restrict__ ext, const uint32_t tile, const uint32_t nx,
                                                                    const uint32_t n_chunks)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    uint32_t combo, tx;
    if (!eval_block_of(a, nx, n_chunks, combo, tx)) return;
    const ArenaRef ar = load_arena_ref(ext, combo / n_chunks);
    eval_block(a, ar, tile, tx, combo % n_chunks, lds64);
}

// ---------------------------------------------------------------------------
// K1+K2 fused launch used by bsg_probe_many: workgroups [0, n_probe) stream arena i's bitsets
// (probe role, exactly k_probe_terms), workgroups [n_probe, n_probe + n_eval) evaluate the programs
// of arena i-1 from the verdicts the previous launch left in the other scratch slot (eval role: two
// 256-query chunks per 512-thread workgroup).  The eval workgroups sit at the FRONT of the grid: their
// ~4 us chain of dependent latencies (program words, verdict words, transposes, LDS stack) runs while the
// probe workgroups keep HBM saturated, instead of costing a second kernel between two streaming kernels.
// ---------------------------------------------------------------------------
struct FusedArgs {
    ProbeArgs p;
    EvalArgs e;
    uint32_t n_probe;        // probe role workgroups = p.max_blocks * referenced kinds * p.n_arenas
    uint32_t eval_pairs;     // ceil(n_chunks / 2) of the eval role
    uint32_t eval_lds_half;  // bytes of LDS per 256-query half
    uint32_t eval_tile;      // block groups per eval half (1 or kEvalGroupTile)
    uint32_t n_kinds;
};

__global__ __launch_bounds__(kProbeThreads) void k_probe_fused(const FusedArgs f, const ArenaTable<kMaxFusedArenas> tp, const ArenaTable<kMaxFusedArenas> te)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const uint32_t id = blockIdx.x;
    const uint32_t g_tiles = (f.e.max_G + f.eval_tile - 1) / f.eval_tile;
    const uint32_t per_arena = g_tiles * f.eval_pairs;
    const uint32_t n_eval = per_arena * f.e.n_arenas;
    if (id >= n_eval) {
        const uint32_t j = id - n_eval;
        const uint32_t per_probe_arena = f.p.max_blocks * f.n_kinds;
        const uint32_t ai = j / per_probe_arena, r = j - ai * per_probe_arena;
        probe_role<kProbeThreads, true>(f.p, tp.ar[ai], r % f.p.max_blocks, r / f.p.max_blocks, lds64);   // fused launches are few-term launches
    } else {
        const uint32_t ai = id / per_arena, r = id - ai * per_arena;
        const uint32_t gtile = r / f.eval_pairs, pair = r - gtile * f.eval_pairs;
        const ArenaRef &ar = te.ar[ai];
        const uint32_t g0 = gtile * f.eval_tile;
        if (g0 >= ar.G()) return;
        const uint32_t half = threadIdx.x >> 8, htid = threadIdx.x & 255u;
        const uint32_t c = pair * 2 + half;
        const uint32_t n_chunks = (f.e.n_queries + kEvalThreads - 1) / kEvalThreads;
        eval_role(f.e, ar, g0, min(f.eval_tile, ar.G() - g0), c, htid, lds64 + (uint64_t)half * (f.eval_lds_half / 8), c < n_chunks);
    }
}

// ---------------------------------------------------------------------------
// K1 + K2 in ONE dispatch, per TILE (round 4): k_probe_eval.  Probe workgroups are exactly k_probe_terms, except that a
// verdict word leaves as a tagged 16-byte entry (store_tagged).  A tile = tile_groups x 64 consecutive blocks of one arena x
// the referenced kinds; the workgroups of the tile's last `helpers` blocks (last kind's pass) stay after their own probe:
// they poll the tile's entries until all carry this launch's tag (a bounded wait: whoever they wait for precedes them in
// the in-order dispatch), then share the tile's 256-query chunks among them: transpose the tile's verdict words once
// (identity word list: every chunk references the batch's few words), run the lowered programs on 64-block masks, store
// the survivor words.  The evaluation of tile i therefore runs under the streaming of the tiles behind it, the verdicts
// are read back while they are still in L2 / MALL, and the dispatch boundary + the serial k_eval_programs (15 us per 20
// arenas at C2) are gone.  No atomics, no fences, no counters to reset.
// Taken for few-term batches whose chunks all use the identity word list (C2, C4, interactive batches); everything
// else keeps k_probe_terms(_many) + k_eval_programs.
// ---------------------------------------------------------------------------
#ifndef BSG_FOLD_TILE
#define BSG_FOLD_TILE 4
#endif
constexpr uint32_t kFoldGroupTile = BSG_FOLD_TILE;   // 64-block groups per evaluation tile of a large launch (lab: -DBSG_FOLD_TILE=8 / 16)
struct FoldArgs {
    ProbeArgs p;             // p.V: 16-byte tagged entries (2 x the u64 of the plain layout), p.seq: the launch's tag
    EvalArgs e;
    uint32_t tile_groups;    // 64-block groups per tile (1 or kFoldGroupTile)
    uint32_t helpers;        // workgroups of a tile that share its evaluation (>= 1)
    uint32_t n_kinds;
    uint32_t lab;            // lab only (BSG_LAB_FOLD): 1 = evaluators store nothing, 2 = evaluators do not wait for the tags
};

// LDS the evaluation of one tile needs with kProbeThreads threads: transposed words of every group + per-lane stacks
__host__ __device__ inline uint32_t fold_eval_lds_bytes(uint32_t tile_groups, uint32_t max_cw, uint32_t max_depth)
{
    return (tile_groups * max_cw * 64u + max_depth * (uint32_t)kProbeThreads) * 8u;
}

// The program words of an evaluator's first chunk pair: they depend on nothing but the workgroup's position, so they are
// requested BEFORE its own probe — under a saturated memory system every dependent round trip in the evaluator's chain
// costs microseconds at the end of the launch.
constexpr uint32_t kFoldPre = 8;
struct FoldPrefetch { uint32_t pre[kFoldPre]; uint32_t len; };

__device__ __forceinline__ void fold_prefetch(const EvalArgs &a, uint32_t pair, FoldPrefetch &pf)
{
    const uint32_t half = threadIdx.x >> 8, htid = threadIdx.x & 255u;
    const uint32_t n_chunks = (a.n_queries + kEvalThreads - 1) / kEvalThreads;
    const uint32_t c = min(pair * 2 + half, n_chunks - 1);                       // (a half without a chunk loads a valid one and drops it)
    const uint32_t *P = a.prog + (uint64_t)c * a.Lmax * kEvalThreads + htid;
#pragma unroll
    for (uint32_t j = 0; j < kFoldPre; ++j) pf.pre[j] = j < a.Lmax ? P[(uint64_t)j * kEvalThreads] : (7u << 28);
    pf.len = a.chunk_len[c];
}

__device__ __forceinline__ void fold_tail(const FoldArgs &f, const ArenaRef &ar, uint32_t b, uint32_t y, uint64_t *lds64, FoldPrefetch pf)
{
    const EvalArgs &a = f.e;
    const uint32_t tid = threadIdx.x, lane = tid & (kWave - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const uint32_t T = f.tile_groups, T64 = T * 64u;
    const uint32_t tile = b / T64, first = tile * T64;
    const uint32_t nb = min(ar.n_blocks - first, T64);                          // blocks of this tile
    const uint32_t n_chunks = (a.n_queries + kEvalThreads - 1) / kEvalThreads, n_pairs = (n_chunks + 1) / 2;
    const uint32_t K = min(min(f.helpers, nb), n_pairs);
    const uint32_t role = b - first - (nb - K);

    // ---- the tile's verdict words, polled until they carry this launch's tag, transposed once:
    //      VT[(t * max_cw + s) * 64 + bit] = 64-block mask of one term ----
    uint64_t *VT = lds64;
    const uint32_t g0 = tile * T, gt = min(T, ar.G() - g0);
    const uint64_t *V = a.V + ar.v_off(a.Wt) * 2;
    const uint32_t ncw = a.max_cw;
    __syncthreads();                                                             // the probe phase is done with the LDS
    for (uint32_t idx = wave; idx < gt * ncw; idx += kProbeWaves) {
        const uint32_t tt = idx / ncw, s = idx - tt * ncw, g = g0 + tt;
        const bool row_valid = g * 64u + lane < ar.n_blocks;
        const uint64_t *e = V + (((uint64_t)g * a.Wt + s) * 64 + (row_valid ? lane : 0u)) * 2;
        uint64_t x;
        while (true) {
            const bool ok = load_tagged(e, f.p.seq, x) || !row_valid;
            if (__ballot(!ok) == 0ull || (f.lab & 2u)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!row_valid) x = 0ULL;
        VT[(uint64_t)idx * 64 + lane] = wave_transpose64(x, (int)lane);
    }
    __syncthreads();

    // ---- programs: a 512-thread workgroup is two 256-query halves; evaluator `role` takes chunk pairs role, role + K, ... ----
    uint64_t *stk = lds64 + (uint64_t)T * ncw * 64 + tid;                       // per-lane stack, stride kProbeThreads
    const uint32_t half = tid >> 8, htid = tid & 255u;
    constexpr uint32_t kPre = kFoldPre;
    for (uint32_t pair = role; pair < n_pairs; pair += K) {
        const uint32_t c = pair * 2 + half;
        uint32_t pre[kPre];
#pragma unroll
        for (uint32_t j = 0; j < kPre; ++j) pre[j] = pf.pre[j];
        const uint32_t len = pf.len;
        if (pair + K < n_pairs) fold_prefetch(a, pair + K, pf);                  // the next pair's words travel while this one runs
        if (c >= n_chunks) continue;                                             // wave-uniform (a half is 4 whole waves)
        const uint32_t *P = a.prog + (uint64_t)c * a.Lmax * kEvalThreads + htid;
        const uint32_t q = c * kEvalThreads + htid;
        uint64_t res[kFoldGroupTile];
#pragma unroll
        for (uint32_t tt = 0; tt < kFoldGroupTile; ++tt) {
            if (tt < gt) {
                const uint64_t *vt = VT + (uint64_t)tt * ncw * 64;
                uint64_t top = ~0ULL;                                            // empty program == nil query == true
                uint32_t sp = 0;
                auto step = [&](uint32_t op) {
                    const uint32_t opc = op >> 28;
                    if (opc == 7u) return;
                    if (opc == 1u || opc == 2u) {
                        --sp;
                        const uint64_t under = stk[(uint64_t)(sp - 1) * kProbeThreads];
                        top = (opc == 1u) ? (under & top) : (under | top);
                    } else {
                        if (sp > 0) stk[(uint64_t)(sp - 1) * kProbeThreads] = top;
                        ++sp;
                        top = (opc == 0u) ? vt[op & 0x0FFFFFFFu] : (opc == 3u ? ~0ULL : 0ULL);
                    }
                };
#pragma unroll
                for (uint32_t j = 0; j < kPre; ++j) if (j < len) step(pre[j]);
                for (uint32_t j = kPre; j < len; ++j) step(P[(uint64_t)j * kEvalThreads]);
                const uint32_t nvalid = ar.n_blocks - (g0 + tt) * 64u;
                res[tt] = top & (nvalid >= 64 ? ~0ULL : ((1ULL << nvalid) - 1));
            }
        }
        if (q < a.n_queries && !(f.lab & 1u)) {
            uint64_t *dst = a.out + ar.out_off(a.n_queries) + (uint64_t)q * ar.G() + g0;
            // (measured per 20 arenas, 119-122 us as written: non-temporal stores 172 us — 8-byte partial writes straight to HBM;
            //  16-byte stores 125 us; tiles of 8 / 16 groups = 64 / 128-byte pieces per lane 128 / 147 us: the longer tail costs more
            //  than the fuller lines save)
#pragma unroll
            for (uint32_t tt = 0; tt < kFoldGroupTile; ++tt) if (tt < gt) dst[tt] = res[tt];
        }
    }
}

// grid = (max_blocks, referenced kinds, arenas of the group), as k_probe_terms
__global__ __launch_bounds__(kProbeThreads) __attribute__((amdgpu_num_sgpr(BSG_STREAM_SGPRS))) void k_probe_eval(const FoldArgs f, const ArenaTable<kMaxGroupArenas> t)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const ArenaRef &ar = t.ar[blockIdx.z];
    const uint32_t b = blockIdx.x, y = blockIdx.y;
    if (b >= ar.n_blocks) return;
    // The evaluators of a tile are the workgroups of its LAST K blocks in the LAST kind's pass: every workgroup whose
    // verdicts they wait for precedes them in the dispatch order (x fastest, then y), so it is resident or finished —
    // the wait is bounded and cannot deadlock.  Everybody else runs the probe code on a path of its own, compiled as in
    // k_probe_terms (sharing one path with the evaluators cost every workgroup 8%: 104 -> 112.6 us per 20 arenas).
    const uint32_t T64 = f.tile_groups * 64u, first = b / T64 * T64, nb = min(ar.n_blocks - first, T64);
    const uint32_t n_pairs = ((f.e.n_queries + kEvalThreads - 1) / kEvalThreads + 1) / 2;
    const uint32_t K = min(min(f.helpers, nb), n_pairs);
    const bool evaluator = y + 1u == f.n_kinds && b - first + K >= nb;          // workgroup-uniform
    FoldPrefetch pf;
    if (evaluator) fold_prefetch(f.e, b - first - (nb - K), pf);
    probe_role<kProbeThreads, true, true>(f.p, ar, b, y, lds64);
    if (evaluator) fold_tail(f, ar, b, y, lds64, pf);
}

// ---------------------------------------------------------------------------
// Survivor ROWS for the host (round 4): what crosses PCIe is what the host needs, not Q x B / 8 bytes whatever they hold.
// The reference's consumer wants the surviving block IDS of a query (blockScanCandidate order, query_exec.go:321,603);
// a dense [Q][G] bitset over one PCIe link halved the delivered rate (C4: 5.2 MB per step).  One lane per query row
// reads the row's G words from the device-side survivors, tags it
//   NONE  no block survives              ALL   every block survives            (nothing but the 4-byte header travels)
//   LIST  count <= 2 G survivors: their block indices, ascending, as u32 in the row's own slot
//   DENSE the row's G words, as bsg_probe_many delivers them
// and writes header + payload STRAIGHT into page-locked host memory (zero-copy: only the bytes written cross the link; the
// row slots keep the dense layout, so addressing is fixed and nothing has to be sized before the copy).
// ---------------------------------------------------------------------------
constexpr uint32_t kRowNone = 0, kRowAll = 1, kRowList = 2, kRowDense = 3;
struct RowsDst { uint64_t row_off; uint64_t hdr_off; };   // arena i of the group: first u64 of its rows / first u32 of its headers in the caller's buffers
template <uint32_t N>
struct RowsTable { RowsDst d[N]; };
struct RowsArgs {
    const uint64_t *out;      // the group's survivors on the device (arena i: [n_queries][G_i] at out + ar[i].out_off)
    uint64_t *rows;           // host-mapped: the caller's row slots (dense layout)
    uint32_t *hdr;            // host-mapped: tag << 30 | survivor count, per (arena, query)
    uint32_t n_queries;
};

// One row: count, tag, header, payload.  `src` is the row's G words — in LDS (staged) or in device memory.
__device__ __forceinline__ void survivor_row(const uint64_t *src, uint32_t G, uint32_t n_blocks, uint32_t *hdr, uint64_t *row)
{
    uint32_t cnt = 0;
    for (uint32_t g = 0; g < G; ++g) cnt += (uint32_t)__popcll(src[g]);
    const uint32_t tag = cnt == 0u ? kRowNone : cnt == n_blocks ? kRowAll : cnt <= 2u * G ? kRowList : kRowDense;
    *hdr = (tag << 30) | cnt;
    if (tag == kRowList) {
        uint32_t *ids = reinterpret_cast<uint32_t *>(row);
        uint32_t n = 0;
        for (uint32_t g = 0; g < G; ++g) {
            uint64_t w = src[g];
            while (w) {
                ids[n++] = g * 64u + (uint32_t)__builtin_ctzll(w);
                w &= w - 1;
            }
        }
    } else if (tag == kRowDense) {
        for (uint32_t g = 0; g < G; ++g) row[g] = src[g];
    }
}

// rows of up to kRowsStageG words go through LDS: the 256 rows of a workgroup are one contiguous stretch of the survivors, read with
// coalesced loads and handed to the lanes at an odd stride (a lane walking its own row in device memory touched 64 lines per load:
// 20 arenas x 4 096 rows of 16 words took 28 us, 375 GB/s)
constexpr uint32_t kRowsStageG = 16;

// grid = (ceil(n_queries / 256), arenas of the group)
__global__ __launch_bounds__(256) void k_survivor_rows(const RowsArgs a, const ArenaTable<kMaxRowsArenas> t, const RowsTable<kMaxRowsArenas> dst)
{
    __shared__ uint64_t tile[256 * (kRowsStageG + 1)];
    const ArenaRef &ar = t.ar[blockIdx.y];
    const RowsDst d = dst.d[blockIdx.y];
    const uint32_t q0 = blockIdx.x * 256u, q = q0 + threadIdx.x;
    const uint32_t G = ar.G();
    const uint64_t *base = a.out + ar.out_off(a.n_queries);
    if (G <= kRowsStageG) {                                   // workgroup-uniform
        const uint32_t rows = min(256u, a.n_queries - q0), Gp = G | 1u;
        const uint64_t *src = base + (uint64_t)q0 * G;
        const uint32_t dr = 256u / G, dc = 256u % G;         // (one division per lane, not one per word)
        uint32_t r = threadIdx.x / G, c = threadIdx.x % G;
        // all of a lane's words are requested before the first one is used: as a loop with one load per trip (until round 6) the
        // compiler waited for every load in turn — up to 16 round trips, with 1.25 workgroups per CU nothing else to run meanwhile
        // (20 arenas x 4 096 rows: 21 us, most of it this).  rows * G <= 256 * kRowsStageG: kRowsStageG trips cover the tile.
        const uint32_t n = rows * G;
        uint64_t v[kRowsStageG];
#pragma unroll
        for (uint32_t u = 0; u < kRowsStageG; ++u) {
            const uint32_t i = threadIdx.x + u * 256u;
            v[u] = i < n ? src[i] : 0;
        }
#pragma unroll
        for (uint32_t u = 0; u < kRowsStageG; ++u) {
            if (threadIdx.x + u * 256u < n) tile[r * Gp + c] = v[u];
            r += dr; c += dc;
            if (c >= G) { c -= G; ++r; }
        }
        __syncthreads();
        if (q < a.n_queries) survivor_row(tile + threadIdx.x * Gp, G, ar.n_blocks, a.hdr + d.hdr_off + q, a.rows + d.row_off + (uint64_t)q * G);
        return;
    }
    if (q >= a.n_queries) return;
    survivor_row(base + (uint64_t)q * G, G, ar.n_blocks, a.hdr + d.hdr_off + q, a.rows + d.row_off + (uint64_t)q * G);
}

// The PACKED form of the rows (BSG_PROBE_ROWS_PACKED): the payloads of a run of 256 consecutive queries of one arena — a LIST row's
// ids as ceil(count / 2) words, a DENSE row's G words — lie back to back, in query order, from the start of the run's slot area.
// Why: a store into host memory that does not fill a line is one PCIe write of its own; the dense layout makes one per LIST row
// (17 240 per 20 C2 arenas: 18.7 of the kernel's 27 us, profiles/r06_hostwrite_lab.txt), the packed run leaves as one contiguous
// stretch written with coalesced 8-byte stores.  Rows of at most kRowsStageG words only (the host checks).  Headers: one byte per row,
// at BYTE (hdr_off + q) of the header buffer (see below).
// grid = (ceil(n_queries / 256), arenas of the group)
__global__ __launch_bounds__(256) void k_survivor_rows_packed(const RowsArgs a, const ArenaTable<kMaxRowsArenas> t, const RowsTable<kMaxRowsArenas> dst)
{
    __shared__ uint64_t tile[256 * (kRowsStageG + 1)];
    __shared__ uint32_t wave_total[4];
    const ArenaRef &ar = t.ar[blockIdx.y];
    const RowsDst d = dst.d[blockIdx.y];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t q0 = blockIdx.x * 256u, q = q0 + tid;
    const uint32_t G = ar.G();
    const uint64_t *src = a.out + ar.out_off(a.n_queries) + (uint64_t)q0 * G;
    const uint32_t rows = min(256u, a.n_queries - q0), Gp = G | 1u, n = rows * G;
    {   // the run's words into the tile, a row per lane at an odd stride (as k_survivor_rows)
        const uint32_t dr = 256u / G, dc = 256u % G;
        uint32_t r = tid / G, c = tid % G;
        uint64_t v[kRowsStageG];
#pragma unroll
        for (uint32_t u = 0; u < kRowsStageG; ++u) {
            const uint32_t i = tid + u * 256u;
            v[u] = i < n ? src[i] : 0;
        }
#pragma unroll
        for (uint32_t u = 0; u < kRowsStageG; ++u) {
            if (tid + u * 256u < n) tile[r * Gp + c] = v[u];
            r += dr; c += dc;
            if (c >= G) { c -= G; ++r; }
        }
    }
    __syncthreads();
    const bool live = q < a.n_queries;
    uint64_t w[kRowsStageG];
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t g = 0; g < kRowsStageG; ++g) {
        w[g] = (live && g < G) ? tile[tid * Gp + g] : 0ULL;
        cnt += (uint32_t)__popcll(w[g]);
    }
    const uint32_t tag = cnt == 0u ? kRowNone : cnt == ar.n_blocks ? kRowAll : cnt <= 2u * G ? kRowList : kRowDense;
    // the packed form's header is ONE BYTE per row, tag << 6 | (a LIST row's count: at most 2 kRowsStageG = 32); an ALL row's count is
    // the arena's block count, a DENSE row's the population of its words — 4 096 rows x 4 bytes per arena were 3.3 of the C4 step's
    // 3.7 us of host traffic
    if (live) reinterpret_cast<uint8_t *>(a.hdr)[d.hdr_off + q] = (uint8_t)((tag << 6) | (tag == kRowList ? cnt : 0u));
    const uint32_t size = !live ? 0u : tag == kRowList ? (cnt + 1u) >> 1 : tag == kRowDense ? G : 0u;       // payload words
    // exclusive prefix of the sizes over the run: within the wave by shuffles, across the four waves through LDS
    uint32_t incl = size;
#pragma unroll
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63u) wave_total[wave] = incl;
    __syncthreads();                                           // (every lane has its row in registers: the tile may be overwritten)
    uint32_t off = incl - size, total = 0;
#pragma unroll
    for (uint32_t v = 0; v < 4; ++v) {
        if (v < wave) off += wave_total[v];
        total += wave_total[v];
    }
    if (tag == kRowList && live) {
        uint32_t *ids = reinterpret_cast<uint32_t *>(tile + off);
        uint32_t k = 0;
#pragma unroll
        for (uint32_t g = 0; g < kRowsStageG; ++g) {
            uint64_t x = w[g];
            while (x) {
                ids[k++] = g * 64u + (uint32_t)__builtin_ctzll(x);
                x &= x - 1;
            }
        }
        if (cnt & 1u) ids[cnt] = 0u;                           // the odd count's last half word
    } else if (tag == kRowDense && live) {
#pragma unroll
        for (uint32_t g = 0; g < kRowsStageG; ++g) if (g < G) tile[off + g] = w[g];
    }
    __syncthreads();
    uint64_t *out = a.rows + d.row_off + (uint64_t)q0 * G;
    for (uint32_t i = tid; i < total; i += 256u) out[i] = tile[i];
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
    base_hashes_at(bytes, off[e], off[e + 1] - off[e], h);
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

// location(h, i), i = 0 .. k - 1, of bloom/v3 (h[i % 2] + i * h[2 + (((i + i % 2) % 4) / 2)], wrapping u64):
//     i % 4 == 0: h0 + i h2      1: h1 + i h3      2: h0 + i h3      3: h1 + i h2
// four per trip from running multiples of h2 and h3, every hash word in a register of its own.  Until round 6 the loop took one
// location per trip and picked h[i & 1] out of the array: the compiler answered the dynamic index by keeping h[] in SCRATCH memory
// (48 bytes per lane) — a scratch_load + s_waitcnt vmcnt(0) in front of every location, which is what profiles/r04_build_pmc.txt's
// "78 % of a wave's life waiting" was (found reading the ISA; k is wave-uniform, the three guards are scalar branches).
template <typename F>
__device__ __forceinline__ void for_each_location_x(uint32_t k, uint64_t h0, uint64_t h1, uint64_t h2, uint64_t h3, F f)
{
    // four running sums, one per residue of i mod 4, each advanced by 4 h2 or 4 h3 per trip: one 64-bit add per location
    uint64_t x0 = h0, x1 = h1 + h3, x2 = h0 + h3 + h3, x3 = h1 + h2 + h2 + h2;
    const uint64_t d2 = h2 << 2, d3 = h3 << 2;
    for (uint32_t i = 0; i < k; i += 4) {
        f(x0);
        if (i + 1 < k) f(x1);
        if (i + 2 < k) f(x2);
        if (i + 3 < k) f(x3);
        x0 += d2; x1 += d3; x2 += d3; x3 += d2;
    }
}

// Sets the k bits of one entry.  (Measured and dropped, round 6: an approximate-quotient modulo for m < 2^30 — lo32(xh mh) + hi32(xh ml) +
// hi32(xl mh), remainder candidate in [0, 4m), two unsigned-min fix-ups: 10 instructions on paper against 12-13 — is 7 % SLOWER,
// 239 -> 256 us per 1 000 x 19 600 entries: v_mul_hi_u32 issues slower than the v_mad_u64_u32 the exact quotient is made of.)
template <int MODE, typename BITS32>
__device__ __forceinline__ void set_entry_bits(BITS32 bits, const DevDesc &d, const ModF64 &fm, uint64_t h0, uint64_t h1, uint64_t h2, uint64_t h3)
{
    for_each_location_x(d.k, h0, h1, h2, h3, [&](uint64_t x) {
        const uint64_t loc = MODE == kModFp64 ? (uint64_t)mod_f64(x, fm) : locate<MODE == kModBarrett32>(d, x);
#ifdef BSG_LAB_NO_ATOMICS      // lab only: keep the location live without touching the bitset
        asm volatile("" ::"v"((uint32_t)loc));
#else
        uint64_t w = loc >> 5;
        if (MODE != kModBarrett64) {             // (kept opaque: otherwise the word's byte address becomes (loc >> 3) & ~3, + base: three instructions for two)
            uint32_t w32 = (uint32_t)w;
            asm volatile("" : "+v"(w32));
            w = w32;
        }
        __hip_atomic_fetch_or(&bits[w], 1u << ((uint32_t)loc & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    });
}

template <int MODE, typename BITS32>
__device__ __forceinline__ void build_entries(const BuildArgs &a, const BuildItem &it, const DevDesc &d, BITS32 bits, uint32_t tid)
{
    ModF64 fm{};
    if (MODE == kModFp64) fm = make_modf64(d.m, d.magic);
    if (a.h) {
        for (uint32_t e = it.e_begin + tid; e < it.e_end; e += kBuildThreads) {
            const ulonglong2 *hp = reinterpret_cast<const ulonglong2 *>(a.h + (uint64_t)e * 4);
            const ulonglong2 x = hp[0], y = hp[1];
            set_entry_bits<MODE>(bits, d, fm, x.x, x.y, y.x, y.y);
        }
        return;
    }
    // One entry per lane per trip.  Measured and dropped: preloading 32 bytes of 2-4 entries per lane (round 3); a two-deep software
    // pipeline — offsets two trips ahead, tail and table row one ahead (round 6: 187 vs 185 us per 1 000 x 19 600 entries at 72
    // VGPRs; eight waves per SIMD already hide the two round trips).
    // (the trip's offsets at a uniform pointer + the lane's constant 4 tid: the address costs no vector instruction)
    const uint32_t n = it.e_end - it.e_begin;
    for (uint32_t b = 0; b < n; b += kBuildThreads) {
        const uint32_t *off = a.off + it.e_begin + b;
        if (tid < n - b) {
            const uint32_t o0 = off[tid], o1 = off[tid + 1];
            uint64_t h[4];
            base_hashes_at(a.bytes, o0, o1 - o0, h);
            set_entry_bits<MODE>(bits, d, fm, h[0], h[1], h[2], h[3]);
        }
    }
}

__global__ __launch_bounds__(kBuildThreads) void k_build(const BuildArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const BuildItem it = a.items[blockIdx.x];
    const DevDesc d = a.desc[it.filter];
    const uint32_t tid = threadIdx.x;
    if (d.m == 0) return;
    const uint64_t nw = (d.m + 63) >> 6;
    const int mode = mod_mode(d.m);
    if (it.staged) {
        for (uint32_t i = tid; i < nw; i += kBuildThreads) lds64[i] = 0;
        __syncthreads();
        lds_u32 *bits = (lds_u32 *)lds64;
        if (mode == kModFp64)           build_entries<kModFp64>(a, it, d, bits, tid);
        else if (mode == kModBarrett32) build_entries<kModBarrett32>(a, it, d, bits, tid);
        else                            build_entries<kModBarrett64>(a, it, d, bits, tid);
        __syncthreads();
        uint64_t *dst = a.out + d.word_off;
        for (uint32_t i = tid; i < nw; i += kBuildThreads) dst[i] = lds64[i];
    } else {
        uint32_t *bits = reinterpret_cast<uint32_t *>(a.out + d.word_off);
        if (mode == kModBarrett64) build_entries<kModBarrett64>(a, it, d, bits, tid);
        else                       build_entries<kModBarrett32>(a, it, d, bits, tid);
    }
}

// ---------------------------------------------------------------------------
// decode_sections: filter sections exactly as stored on disk (encodeFilterSection,
// file_format.go:343-384) -> arena words, on the device.  One workgroup per section:
//   1. CRC32C (Castagnoli, reflected 0x82F63B78) of the payload: every thread runs slice-by-8 over its
//      contiguous chunk with a zero initial value, partials are combined pairwise with the GF(2)
//      "multiply by x^(8n) mod P" operator (chunks 1..255 have one common length, so each tree level
//      needs a single squaring of the operator), then the initial/final 0xFFFFFFFF are folded in;
//   2. every present filter's big-endian u64 words are byte-swapped into their 128-byte-aligned arena
//      slot (bit i <-> words[i>>6] bit i&63, native little-endian).
// A CRC mismatch marks the block (status -2 = ErrInvalidHash) and turns its filters into nil filters
// (m = 0) so a corrupt section cannot poison the batch (query_exec.go:580-590 isolates it per block).
// ---------------------------------------------------------------------------

constexpr int kDecodeThreads = 256;

struct CrcConsts {
    uint32_t table[8][256];   // slice-by-8 tables
    uint32_t x2n[64];         // x^(2^i) mod P, reflected, i < 64: NO periodicity assumed (zlib indexes its 32-entry table with k & 31 because
                              // x has order 2^32 - 1 under the CRC-32 polynomial; the Castagnoli polynomial is (x + 1) times a primitive
                              // polynomial of degree 31, so x has order 2^31 - 1 and a period-32 table is wrong from k = 32 on, i.e. for payloads
                              // of 2^29 bytes and more — found by tests/test_big_m_gpu.py on a 1 GiB filter section)
    uint32_t skip;            // x^(8 * kCrcGranule * (kDecodeThreads - 1)) mod P: a thread's hop between its granules
    uint32_t pad[3];
    uint32_t gpow[256];       // x^(8 * kCrcGranule * t) mod P: thread t's last granule ends t granules before the tail
    uint32_t bpow[64];        // x^(8 * i) mod P: the tail's length (< kCrcGranule)
    uint32_t upow[64];        // x^(8 * BSG_DECODE_UNIT * j) mod P: slice j's distance to the payload's end when the section takes the default unit
};
constexpr uint32_t kCrcGranule = 64;   // bytes one lane checksums per trip: a wave covers 4 KiB of contiguous payload
                                       // (measured per 1 000 block sections, decode: 64 -> 70.8 us, 128 -> 90.5 us, 256 -> 116.9 us)

// CRC32C (zero initial value, no final xor) of sec[0, P) by one workgroup of kDecodeThreads threads.
// CRC is linear over GF(2): the payload is cut into kCrcGranule-byte granules dealt to the threads round-robin, so a wave
// reads one contiguous stretch per trip (the first version gave every thread one contiguous chunk: a load-use chain per
// 8 bytes at a stride no other lane shared, 199 us per 1 000 sections).  A thread carries its partial across the 255
// granules it skips with one multiply by x^(8*64*255), aligns it to the end of the payload once, and the partials
// are simply XOR-ed together.  The < 64 trailing bytes are one thread's.  Result valid in thread 0.
__device__ __forceinline__ uint32_t crc32c_payload(const uint8_t *sec, uint32_t P, const uint32_t (*tab)[256], const CrcConsts *consts,
                                                  uint32_t *part, uint32_t tid)
{
    const uint32_t G = P / kCrcGranule, tail = P % kCrcGranule;
    const uint32_t skip = consts->skip;
    uint32_t crc = 0;
    // granules are dealt from the END: thread t's last one is granule G - 1 - t, so its partial sits exactly
    // tail + 64 t bytes before the end of the payload — two table multiplies instead of a square-and-multiply chain
    if (tid < kDecodeThreads && tid < G) {
        const uint32_t g_last = G - 1 - tid;
        bool any = false;
        for (uint32_t g = g_last % kDecodeThreads; g <= g_last; g += kDecodeThreads) {
            constexpr int kW = kCrcGranule / 8;
            uint64_t v[kW];
            const uint8_t *p = sec + (uint64_t)g * kCrcGranule;
#pragma unroll
            for (int u = 0; u < kW; ++u) v[u] = load_u64_unaligned(p + 8 * u);
            if (any) crc = crc_multmodp(skip, crc);
            any = true;
#pragma unroll
            for (int u = 0; u < kW; ++u) {
                const uint32_t lo = (uint32_t)v[u] ^ crc, hi = (uint32_t)(v[u] >> 32);
                crc = tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^
                      tab[3][hi & 0xFF] ^ tab[2][(hi >> 8) & 0xFF] ^ tab[1][(hi >> 16) & 0xFF] ^ tab[0][hi >> 24];
            }
        }
        if (tid) crc = crc_multmodp(consts->gpow[tid], crc);
        if (tail) crc = crc_multmodp(consts->bpow[tail], crc);
    }
    if (tid == kDecodeThreads - 1 && tail) {                            // trailing bytes: already aligned to the end
        uint32_t t = 0;
        for (uint32_t i = P - tail; i < P; ++i) t = tab[0][(t ^ sec[i]) & 0xFF] ^ (t >> 8);
        crc ^= t;
    }
    if (tid < kDecodeThreads) part[tid] = crc;      // (a launch may carry a wave beyond the checksumming threads: k_decode_sections' header wave)
    __syncthreads();
    for (uint32_t step = kDecodeThreads / 2; step > 0; step >>= 1) {
        if (tid < step) part[tid] ^= part[tid + step];
        __syncthreads();
    }
    return part[0];
}

// The same for a slice of at most one granule per thread (the usual 16 KB slice), with the thread's granule ALREADY in registers:
// k_decode_sections requests it before it copies the tables into LDS, so the payload's trip from HBM and the tables' from L2 overlap.
__device__ __forceinline__ uint32_t crc32c_payload_pre(const uint8_t *sec, uint32_t P, const uint64_t (&v)[kCrcGranule / 8], const uint32_t (*tab)[256],
                                                      const CrcConsts *consts, uint32_t *part, uint32_t tid)
{
    const uint32_t G = P / kCrcGranule, tail = P % kCrcGranule;
    uint32_t crc = 0;
    if (tid < kDecodeThreads && tid < G) {
#pragma unroll
        for (int u = 0; u < (int)(kCrcGranule / 8); ++u) {
            const uint32_t lo = (uint32_t)v[u] ^ crc, hi = (uint32_t)(v[u] >> 32);
            crc = tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^
                  tab[3][hi & 0xFF] ^ tab[2][(hi >> 8) & 0xFF] ^ tab[1][(hi >> 16) & 0xFF] ^ tab[0][hi >> 24];
        }
        if (tid) crc = crc_multmodp(consts->gpow[tid], crc);
        if (tail) crc = crc_multmodp(consts->bpow[tail], crc);
    }
    if (tid == kDecodeThreads - 1 && tail) {                            // trailing bytes: already aligned to the end
        uint32_t t = 0;
        for (uint32_t i = P - tail; i < P; ++i) t = tab[0][(t ^ sec[i]) & 0xFF] ^ (t >> 8);
        crc ^= t;
    }
    if (tid < kDecodeThreads) part[tid] = crc;
    __syncthreads();
    for (uint32_t step = kDecodeThreads / 2; step > 0; step >>= 1) {
        if (tid < step) part[tid] ^= part[tid + step];
        __syncthreads();
    }
    return part[0];
}

// One on-disk section as the region cursor knows it before any byte of it has been read: where it lies and where its
// filters will go.  Everything else — flags, lengths, (m, k), the word counts — is parsed ON THE DEVICE from the bytes
// themselves (k_decode_sections), so the host never looks inside a section on the read path.
struct SectionSlot {
    uint64_t begin;          // byte offset of the section inside the uploaded region image
    uint64_t slot_words;     // word offset of the section's slot in the arena (128-byte aligned)
    uint32_t len;            // section bytes incl. the 4-byte CRC trailer (0 => block without a section)
    uint32_t slot_cap_words; // words the slot holds (section bytes + room for the per-filter alignment)
    uint32_t block;          // local block index: where desc / status of this section live
    uint32_t init_image;     // 0xFFFFFFFF shifted over the payload, xor the final 0xFFFFFFFF (crc_init_image(len - 4)): a function of the
                             // length alone, computed by the host — on the device it was ~17 serial 32-step multiplies per workgroup
};

constexpr uint32_t kMaxHashCountDev = 1024;   // same bound as the host side (kMaxHashCount)

__device__ __forceinline__ uint64_t rd_be64_dev(const uint8_t *p) { return __builtin_bswap64(load_u64_unaligned(p)); }
__device__ __forceinline__ uint32_t rd_le32_dev(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// parseFilterSection's structural checks (file_format.go:392-448), after the CRC has passed.  Returns 0 or the status the
// host-side parser would give (-3 unknown flags, -4 truncated, -5 bad filter, -6 trailing bytes); fills woff / nw / m / k.
// One thread walks this chain while its workgroup waits, so the loads of a filter's header — length, m, k, bitset length: 28
// bytes — are issued TOGETHER wherever the payload has 28 bytes left (round 5: a round trip per filter instead of two; with the
// flags byte riding along with the first filter's, 3 dependent round trips for a section instead of 7).  Nothing is read outside
// the payload, and the checks and their order are parseFilterSection's.
__device__ inline int32_t parse_section_header(const uint8_t *sec, uint32_t plen, uint32_t &present, uint32_t woff[3], uint32_t nw[3],
                                               uint64_t m[3], uint32_t k[3])
{
    present = 0;
    // speculative first batch: the flags byte and the first header behind it
    uint32_t flen_spec = 0; uint64_t mm_spec = 0, kk_spec = 0, bl_spec = 0;
    const bool spec0 = plen >= 29;
    if (spec0) { flen_spec = rd_le32_dev(sec + 1); mm_spec = rd_be64_dev(sec + 5); kk_spec = rd_be64_dev(sec + 13); bl_spec = rd_be64_dev(sec + 21); }
    const uint32_t flags = sec[0];
    if (flags & ~7u) return -3;
    uint64_t pos = 1;
    bool first = true;
    for (uint32_t c = 0; c < 3; ++c) {
        woff[c] = nw[c] = 0; m[c] = 0; k[c] = 0;
        if (!((flags >> c) & 1u)) continue;
        if (plen - pos < 4) return -4;
        uint64_t flen, mm = 0, kk = 0, bl = 0;
        const bool batch = plen - pos >= 28;
        if (first && spec0) { flen = flen_spec; mm = mm_spec; kk = kk_spec; bl = bl_spec; }      // (pos == 1: exactly the speculative batch)
        else if (batch) { const uint8_t *q = sec + pos; flen = rd_le32_dev(q); mm = rd_be64_dev(q + 4); kk = rd_be64_dev(q + 12); bl = rd_be64_dev(q + 20); }
        else flen = rd_le32_dev(sec + pos);
        first = false;
        pos += 4;
        if (flen > plen - pos) return -4;
        if (flen < 24) return -5;
        // (flen >= 24 and flen <= plen - pos: the 24 header bytes lie inside the payload, so they were part of the batch)
        if (bl > ~0ull - 63 || mm > ~0ull - 63 || mm == 0 || kk == 0 || kk > kMaxHashCountDev) return -5;   // (mm + 63 must not wrap: an m of 2^64 - 1 would pass as 0 words)
        const uint64_t words = (bl + 63) / 64;
        if (words > (flen - 24) / 8 || (mm + 63) / 64 > words) return -5;
        m[c] = mm; k[c] = (uint32_t)kk;
        woff[c] = (uint32_t)(pos + 24);
        nw[c] = (uint32_t)((mm + 63) / 64);
        present |= 1u << c;
        pos += flen;
    }
    return pos == plen ? 0 : -6;
}

// Several workgroups per section (round 4).  One workgroup per section left a 1 000-section launch at 4 workgroups per CU, each
// walking ~70 KB behind one chain of dependent loads: 76 us = 0.23 of the HBM roofline, and a run of ~55 sections (one 4 MiB
// chunk of the region cursor) did not fill a quarter of the chip.  CRC-32C is linear over GF(2), so a section's payload is cut
// into SLICES counted from its END — slice j = bytes [P - (j + 1) U, P - j U), U = decode_unit(P) — one workgroup each:
//   * the slice's checksum (crc32c_payload, zero initial value) times x^(8 U j) is its contribution to the payload's checksum;
//     slice 0 also carries the 0xFFFFFFFF initial value shifted over the payload and the final xor;
//   * every workgroup walks the header chain itself (flags, lengths, m, k: <= 3 dependent reads, bounds-checked against the
//     payload because nothing is trusted before the checksum) and byte-swaps the words that START inside its slice into
//     the section's slot — before the checksum is known: a block whose checksum fails keeps nil descriptors (m = 0), so
//     its words are never looked at;
//   * a contribution is published and counted by ONE atomic XOR into the section's word (checksum accumulator + one arrival flag per
//     slice); the arrival that completes the flags compares with the stored checksum and writes status + descriptors.
// grid = (most slices of any section of the run, sections [first, first + gridDim.y)).
// (decode_unit / decode_splits / kDecodeMaxSplits and the GF(2) arithmetic live in crc_slices.h: tests/crc_slices_check.cpp walks the
//  same slices on the host)
struct DecodeScratch {
    uint64_t *acc;      // [slot] low half: XOR of the slices' contributions so far; high half: one flag per slice that arrived (0 before the section's launch)
};

#ifndef BSG_DECODE_IMAGE
#define BSG_DECODE_IMAGE 0        // lab: 1 = a slice of <= 16 KB is read once, coalesced, into an LDS image the checksum and the byte-swap work from (measured
                                  // round 5: 77.8 vs 66.7 us per 1 000 sections — the image's bank conflicts and 28 KB of LDS per workgroup cost more than
                                  // the 7 x fewer L1 accesses save); 0 = every lane loads its granule itself, the byte-swap re-reads the slice from L2
#endif
constexpr uint32_t kDecodeTrip = kDecodeThreads * kCrcGranule;                    // 16 KB: one granule per checksumming thread
// the image in LDS: 8-byte words, ONE word of padding behind every 8 (a granule's words then lie 9 words from the next lane's: the
// lanes of a wave reading "their" word hit different banks instead of all hitting two)
constexpr uint32_t kImageBytes = kDecodeTrip + 16 + 8 + 16;                       // alignment slack in front, the 8 bytes behind, rounding
constexpr uint32_t kImageWords = (kImageBytes / 8 + 7) / 8 * 9 + 9;
__device__ __forceinline__ uint32_t image_word(uint32_t w) { return w + (w >> 3); }
// the 8 bytes at byte offset o of the image
__device__ __forceinline__ uint64_t image_u64(const uint64_t *img, uint32_t o)
{
    const uint32_t w = o >> 3, sh = (o & 7u) * 8u;
    const uint64_t a = img[image_word(w)];
    return sh ? (a >> sh) | (img[image_word(w + 1)] << (64u - sh)) : a;
}

// crc32c_payload's checksum of the n (<= 16 KB) bytes that start `delta` bytes into the LDS image: granules dealt to the threads from
// the END, partials aligned to the end and XOR-ed.  Result valid in thread 0; ends with a barrier.
__device__ __forceinline__ uint32_t crc32c_image(const uint64_t *img, uint32_t delta, uint32_t n, const uint32_t (*tab)[256], const CrcConsts *consts,
                                                uint32_t *part, uint32_t tid)
{
    const uint32_t G = n / kCrcGranule, tail = n % kCrcGranule;
    uint32_t crc = 0;
    if (tid < kDecodeThreads && tid < G) {
        const uint32_t o = delta + (G - 1 - tid) * kCrcGranule;
        const uint32_t w = o >> 3, sh = (o & 7u) * 8u;                // (sh is the same for every lane: delta & 7)
        uint64_t x[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) x[u] = img[image_word(w + u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t vv = sh ? (x[u] >> sh) | (x[u + 1] << (64u - sh)) : x[u];
            const uint32_t lo = (uint32_t)vv ^ crc, hi = (uint32_t)(vv >> 32);
            crc = tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^
                  tab[3][hi & 0xFF] ^ tab[2][(hi >> 8) & 0xFF] ^ tab[1][(hi >> 16) & 0xFF] ^ tab[0][hi >> 24];
        }
        if (tid) crc = crc_multmodp(consts->gpow[tid], crc);
        if (tail) crc = crc_multmodp(consts->bpow[tail], crc);
    }
    if (tid == kDecodeThreads - 1 && tail) {                            // trailing bytes: already aligned to the end
        uint32_t t = 0;
        for (uint32_t i = n - tail; i < n; ++i) {
            const uint32_t o = delta + i;
            const uint32_t byte = (uint32_t)(img[image_word(o >> 3)] >> ((o & 7u) * 8u)) & 0xFFu;
            t = tab[0][(t ^ byte) & 0xFF] ^ (t >> 8);
        }
        crc ^= t;
    }
    if (tid < kDecodeThreads) part[tid] = crc;
    __syncthreads();
    for (uint32_t step = kDecodeThreads / 2; step > 0; step >>= 1) {
        if (tid < step) part[tid] ^= part[tid + step];
        __syncthreads();
    }
    return part[0];
}

#ifndef BSG_DECODE_HDR_WAVE
#define BSG_DECODE_HDR_WAVE 0     // 1: a fifth wave walks the header chain while the four others checksum; 0: thread 255 walks it in front of its own granule (lab)
#endif
constexpr int kDecodeLaunchThreads = kDecodeThreads + (BSG_DECODE_HDR_WAVE ? kWave : 0);
constexpr uint32_t kImagePieces = (kImageBytes / 16 + kDecodeLaunchThreads - 1) / kDecodeLaunchThreads;      // 16-byte pieces a thread carries from memory to the image
__global__ __launch_bounds__(kDecodeLaunchThreads) void k_decode_sections(const uint8_t *region, const SectionSlot *slots, uint32_t first,
                                                                   const CrcConsts *consts, uint64_t *arena, DevDesc *desc,
                                                                   int32_t *status, const DecodeScratch scratch)
{
    __shared__ uint32_t tab[8][256];
    __shared__ uint32_t part[kDecodeThreads];
#if BSG_DECODE_IMAGE
    __shared__ uint64_t img[kImageWords];
#endif
    __shared__ int32_t st;
    __shared__ uint32_t s_present, s_woff[3], s_nw[3], s_k[3], s_last;
    __shared__ uint64_t s_dst[3], s_m[3];
    const uint32_t tid = threadIdx.x;
    const uint32_t slot = first + blockIdx.y, j = blockIdx.x;
    const SectionSlot sl = slots[slot];
    const uint32_t b = sl.block;
    if (sl.len == 0) return;                                       // block without a section: filters stay nil, status 0
    if (sl.len < 5) { if (tid == 0 && j == 0) status[b] = -1; return; }      // parseFilterSection: too small
    const uint32_t P = sl.len - 4, U = decode_unit(P), n_split = P == 0 ? 1u : (P + U - 1) / U;
    if (j >= n_split) return;
    const uint8_t *sec = region + sl.begin;
    const uint32_t hi = P - j * U, lo = hi > U ? hi - U : 0u;       // this workgroup's slice [lo, hi) of the payload
    // (lab, BSG_DECODE_IMAGE) The slice is read from memory ONCE, in 16-byte pieces dealt to the lanes in order (a wave's load covers 1 KiB of consecutive
    // bytes: 8 cache lines), and parked in LDS; the checksum pass takes its 64-byte granules from there and the byte-swap its words.
    // Round 4 had every lane load ITS granule straight from memory — 8-byte loads at a 64-byte lane stride: 32 cache lines per load
    // instruction, 695 L1 accesses per wave for ~100 lines' worth of data (profiles/r05_decode_pmc.txt) — and re-read the slice for
    // the byte-swap.  The pieces are requested FIRST: their trip from HBM overlaps the copy of the CRC tables into LDS.
    const uint32_t n_slice = hi - lo;
    const bool one_trip = n_slice <= kDecodeTrip;
#if !BSG_DECODE_IMAGE
    // the thread's granule is requested FIRST (round 5): its trip from HBM overlaps the copy of the CRC tables into LDS
    uint64_t v[kCrcGranule / 8];
    if (one_trip && tid < (uint32_t)kDecodeThreads && tid < n_slice / kCrcGranule) {
        const uint8_t *p = sec + lo + (uint64_t)(n_slice / kCrcGranule - 1 - tid) * kCrcGranule;
        // four 16-byte loads, not eight 8-byte ones (half the L1 accesses of this phase; measured: no change, 66.7 us — the L1's access
        // rate is not what bounds the kernel either)
#pragma unroll
        for (int u = 0; u < (int)(kCrcGranule / 16); ++u) {
            struct U128 { uint64_t a, b; } w;
            __builtin_memcpy(&w, p + 16 * u, 16);
            v[2 * u] = w.a; v[2 * u + 1] = w.b;
        }
    } else {
#pragma unroll
        for (int u = 0; u < (int)(kCrcGranule / 8); ++u) v[u] = 0;
    }
#else
    const uintptr_t s0 = reinterpret_cast<uintptr_t>(sec + lo);
    const uint32_t delta = (uint32_t)(s0 & 15u);                      // the slice starts `delta` bytes into the image
    const uint4 *src16 = reinterpret_cast<const uint4 *>(s0 - delta);
    const uint32_t n16 = (delta + n_slice + 8u + 15u) / 16u;          // (+ 8: a word may start in the slice's last bytes; the region image is allocated 64 bytes longer than the file bytes it holds)
    uint4 piece[kImagePieces];
    if (one_trip) {
#pragma unroll
        for (uint32_t r = 0; r < kImagePieces; ++r) {
            const uint32_t i = tid + r * kDecodeLaunchThreads;
            piece[r] = i < n16 ? src16[i] : uint4{0, 0, 0, 0};
        }
    }
#endif
    for (uint32_t i = tid; i < 8 * 256; i += kDecodeLaunchThreads) (&tab[0][0])[i] = (&consts->table[0][0])[i];
    // the tables are in place; from here the header chain — 3 dependent round trips, one thread — runs BESIDE the checksum pass (round
    // 5: it ran in front of it, every wave waiting): the pass does not need the header, only the byte-swap behind it does, and the
    // pass ends with barriers
    __syncthreads();
    constexpr uint32_t kHdrThread = BSG_DECODE_HDR_WAVE ? kDecodeThreads : kDecodeThreads - 1;
    if (tid == kHdrThread) {
        // the header chain (untrusted until the checksum is known: every step is checked against the payload's length).  Every
        // workgroup of the section walks it itself, while its other waves are already at their granules — a launch of its own
        // for the headers (one thread per section, a parsed-header table) was measured and dropped: 68 -> 73 us in one launch,
        // 107 -> 134 us as four launches behind the copy.
        uint64_t m[3];
        uint32_t k[3];
        int32_t r = parse_section_header(sec, P, s_present, s_woff, s_nw, m, k);
        if (r == 0) {
            uint64_t cursor = sl.slot_words;
            for (uint32_t c = 0; c < 3; ++c) {
                s_m[c] = m[c]; s_k[c] = k[c];
                if (!((s_present >> c) & 1u)) continue;
                s_dst[c] = cursor;
                cursor += ((uint64_t)s_nw[c] + 15) / 16 * 16;
            }
            if (cursor - sl.slot_words > sl.slot_cap_words) r = -5;   // cannot happen for a slot sized from the section length
        }
        st = r;
    }
    if (tid == (BSG_DECODE_HDR_WAVE ? kDecodeThreads + 1 : kDecodeThreads - 2)) {
        // this slice's distance to the payload's end: a table entry for sections of the default unit (the usual case), a
        // square-and-multiply chain for the few sections large enough to take a wider one
        s_last = j == 0 ? (1u << 31) : U == (uint32_t)BSG_DECODE_UNIT ? consts->upow[j] : crc_x2nmodp((uint64_t)U * j, 3, consts->x2n);
    }
    uint32_t raw;
#if !BSG_DECODE_IMAGE
    raw = one_trip ? crc32c_payload_pre(sec + lo, n_slice, v, tab, consts, part, tid)
                   : crc32c_payload(sec + lo, n_slice, tab, consts, part, tid);      // valid in thread 0; ends with barriers: the header is visible behind them
#else
    if (one_trip) {
        // (the image is written behind the tables' barrier: nobody reads it before the barrier below)
#pragma unroll
        for (uint32_t r = 0; r < kImagePieces; ++r) {
            const uint32_t i = tid + r * kDecodeLaunchThreads;
            if (i < n16) { img[image_word(2 * i)] = (uint64_t)piece[r].x | ((uint64_t)piece[r].y << 32); img[image_word(2 * i + 1)] = (uint64_t)piece[r].z | ((uint64_t)piece[r].w << 32); }
        }
        __syncthreads();
        raw = crc32c_image(img, delta, n_slice, tab, consts, part, tid);      // valid in thread 0; ends with barriers: the header is visible behind them
    } else raw = crc32c_payload(sec + lo, hi - lo, tab, consts, part, tid);
#endif
    // the words that start inside [lo, hi), byte-swapped into the slot (speculative: see above)
    if (st == 0) {
        for (uint32_t c = 0; c < 3; ++c) {
            if (!((s_present >> c) & 1u)) continue;
            const uint32_t w0 = s_woff[c];
            const uint32_t a = lo > w0 ? (lo - w0 + 7) / 8 : 0u;
            const uint32_t e = hi > w0 ? min(s_nw[c], (hi - w0 + 7) / 8) : 0u;
            uint64_t *dst = arena + s_dst[c];
            if (BSG_DECODE_IMAGE && one_trip) {
#if BSG_DECODE_IMAGE
                for (uint32_t w = a + tid; w < e; w += kDecodeLaunchThreads) dst[w] = __builtin_bswap64(image_u64(img, delta + (w0 + 8u * w - lo)));
#endif
            } else {
                const uint8_t *src = sec + w0;
                for (uint32_t w = a + tid; w < e; w += kDecodeLaunchThreads) dst[w] = __builtin_bswap64(load_u64_unaligned(src + 8ull * w));
            }
        }
    }
    if (tid == 0) {
        const uint32_t contrib = j ? crc_multmodp(s_last, raw) : raw ^ sl.init_image;     // slice 0 carries the initial value's image and the final xor
        // ONE atomic publishes and counts (round 5; round 4: a write-through store, a wait, a counter, and the last arrival's reads):
        // the section's word holds the XOR of the contributions in its low half and one arrival flag per slice in its high half
        // (kDecodeMaxSplits = 32 slices at most).  Atomics on one address are totally ordered, so the arrival whose flags complete the
        // set has every contribution in the value it computes from what the atomic returned — nothing else needs ordering (the
        // byte-swapped words are consumed by later kernels), and nothing depends on this target's cache hierarchy.
        uint64_t now = (uint64_t)contrib | (1ull << (32 + j));
        if (n_split > 1) now ^= __hip_atomic_fetch_xor(scratch.acc + slot, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)__popcll(now >> 32) == n_split) {             // the section's last arrival
            const uint32_t total = (uint32_t)now;
            int32_t r = total != rd_le32_dev(sec + P) ? -2 : st;
            if (r == 0) {
                for (uint32_t c = 0; c < 3; ++c) {
                    if (!((s_present >> c) & 1u)) continue;
                    uint64_t magic = s_m[c] <= 1 ? ~0ULL : ~0ULL / s_m[c];
                    if (s_m[c] > 1 && (s_m[c] & (s_m[c] - 1)) == 0) magic += 1;
                    desc[(uint64_t)b * 3 + c] = DevDesc{s_dst[c], s_m[c], magic, s_k[c], 0};
                }
            }
            status[b] = r;
        }
    }
}

// ---------------------------------------------------------------------------
// encode_sections: arena words -> filter sections exactly as encodeFilterSection writes them
// (file_format.go:343-384): [u8 flags] then per present filter [u32 LE len = 24 + 8 nw][u64 BE m][u64 BE k]
// [u64 BE m (bitset length)][nw x u64 BE words], then [u32 LE CRC32C of everything before it].
// Two launches so the checksum pass reads the payload after a kernel boundary: k_encode_payload
// (one workgroup per section) and k_crc_sections (same CRC scheme as k_decode_sections).
// ---------------------------------------------------------------------------
struct EncodeInfo {
    uint64_t begin;          // byte offset of the section in the output region
    uint32_t len;            // section bytes incl. the CRC trailer
    uint32_t present;        // bit c set: filter c is written
    uint64_t src[3];         // word offset of filter c in the device word arena
    uint64_t m[3];
    uint32_t k[3];
    uint32_t nw[3];
    uint32_t init_image;     // crc_init_image(len - 4): the initial value's and the final xor's share of the checksum (host)
    uint32_t pad;
};

__device__ __forceinline__ void store_u64_unaligned(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ void store_u32_unaligned(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

__global__ __launch_bounds__(256) void k_encode_payload(const uint64_t *words, const EncodeInfo *info, uint8_t *region)
{
    const EncodeInfo e = info[blockIdx.x];
    uint8_t *sec = region + e.begin;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) sec[0] = (uint8_t)e.present;
    uint32_t o = 1;
    for (uint32_t c = 0; c < 3; ++c) {
        if (!((e.present >> c) & 1u)) continue;
        if (tid == 0) {
            store_u32_unaligned(sec + o, 24u + 8u * e.nw[c]);
            store_u64_unaligned(sec + o + 4, __builtin_bswap64(e.m[c]));
            store_u64_unaligned(sec + o + 12, __builtin_bswap64((uint64_t)e.k[c]));
            store_u64_unaligned(sec + o + 20, __builtin_bswap64(e.m[c]));
        }
        const uint64_t *src = words + e.src[c];
        uint8_t *dst = sec + o + 28;
        for (uint32_t w = tid; w < e.nw[c]; w += 256) store_u64_unaligned(dst + 8ull * w, __builtin_bswap64(src[w]));
        o += 28u + 8u * e.nw[c];
    }
}

// CRC32C of region[begin, begin + len - 4) written little-endian at its end.  Shares the chunk / combine scheme
// with k_decode_sections.
__global__ __launch_bounds__(kDecodeThreads) void k_crc_sections(uint8_t *region, const EncodeInfo *info, const CrcConsts *consts)
{
    __shared__ uint32_t tab[8][256];
    __shared__ uint32_t part[kDecodeThreads];
    const uint32_t tid = threadIdx.x;
    const EncodeInfo e = info[blockIdx.x];
    for (uint32_t i = tid; i < 8 * 256; i += kDecodeThreads) (&tab[0][0])[i] = (&consts->table[0][0])[i];
    __syncthreads();
    uint8_t *sec = region + e.begin;
    const uint32_t P = e.len - 4;
    const uint32_t raw = crc32c_payload(sec, P, tab, consts, part, tid);
    if (tid == 0) {
        store_u32_unaligned(sec + P, raw ^ e.init_image);
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
// out[i] = OR_b words[desc[b*3+kind].word_off + i].
// A pure read-once stream, so it is written like one: a workgroup is ONE wave that owns 64 x 16 bytes of the bitset and a
// group of blocks; the group's word offsets are read into LDS first (the descriptor load is off the critical path of
// every data load), then kOrInFlight independent 16-byte non-temporal loads per lane are in flight at a time (16 KiB per
// wave).  Groups merge into the zeroed output with atomicOr (distinct addresses per lane: nothing hot).
// Measured on MI355X, 1 000 filters x 360 KB (tools/or_lab.hip): round 2's kernel (256 threads, descriptor chased per
// block, 4 loads in flight) 79 us = 0.57 of peak; this one 57.7 us = 0.78 at groups of 128-256 blocks; 16 -> 8 loads in
// flight: 64 us; LDS-DMA (global_load_lds nt, OR out of LDS): 62-66 us — the read-back costs more than the VGPRs it saves.
constexpr uint32_t kOrThreads = 64;
constexpr uint32_t kOrInFlight = 16;
constexpr uint32_t kOrMaxGroup = 256;        // blocks one workgroup folds (LDS offsets); the host picks the group for ~8 waves per CU
constexpr uint32_t kOrBlocksPerGroup = 128;  // default when the grid is wide enough without splitting further

__global__ __launch_bounds__(kOrThreads) void k_or_reduce_blocks(const uint64_t *words, const DevDesc *desc,
                                                                 uint32_t n_blocks, uint32_t kind, uint64_t n_words,
                                                                 uint64_t *out, uint32_t group)
{
    __shared__ uint64_t offs[kOrMaxGroup];
    const uint32_t b0 = blockIdx.y * group;
    const uint32_t nb = min(group, n_blocks - b0);
    for (uint32_t i = threadIdx.x; i < nb; i += kOrThreads) {
        const DevDesc d = desc[(uint64_t)(b0 + i) * 3 + kind];
        offs[i] = d.m ? d.word_off : ~0ull;
    }
    __syncthreads();
    const uint64_t pair = (uint64_t)blockIdx.x * kOrThreads + threadIdx.x;   // words 2*pair, 2*pair + 1
    if (pair * 2 >= n_words) return;
    const bool two = pair * 2 + 1 < n_words;     // (filters start on 128-byte boundaries and are padded to them: the 16-byte load of a last odd word stays inside)
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t i = 0; i < nb; i += kOrInFlight) {
        u32x4 v[kOrInFlight];
#pragma unroll
        for (uint32_t u = 0; u < kOrInFlight; ++u) {
            const uint64_t o = i + u < nb ? offs[i + u] : ~0ull;
            if (o != ~0ull) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(words + o) + pair);
            else v[u] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (uint32_t u = 0; u < kOrInFlight; ++u) acc |= v[u];
    }
    const uint64_t lo = (uint64_t)acc.x | ((uint64_t)acc.y << 32), hi = (uint64_t)acc.z | ((uint64_t)acc.w << 32);
    if (gridDim.y == 1) {
        out[pair * 2] = lo;
        if (two) out[pair * 2 + 1] = hi;
    } else {
        if (lo) atomicOr((unsigned long long *)&out[pair * 2], (unsigned long long)lo);
        if (two && hi) atomicOr((unsigned long long *)&out[pair * 2 + 1], (unsigned long long)hi);
    }
}

}  // namespace bsg
