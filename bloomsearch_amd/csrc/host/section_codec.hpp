// section_codec.hpp — the on-disk bytes of a block's / file's bloom filters, exactly as the
// reference frames them (file_format.go:334-448):
//   [u8 flags: bit0 field, bit1 token, bit2 field-token]
//   per present filter, in that order: [u32 LE length][filter bytes]
//   [u32 LE CRC32C (Castagnoli) of all preceding section bytes]
// filter bytes = bloom/v3 WriteTo: [u64 BE m][u64 BE k] + bitset WriteTo [u64 BE length(=m)]
// [ceil(m/64) x u64 BE words].  In HBM / at the C-ABI the words are native little-endian
// u64 (the Go []uint64 as-is); the byte swap happens only here, at the wire.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace bsh {

inline uint32_t crc32c_sw(const uint8_t *data, size_t len)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int j = 0; j < 8; ++j) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < len; ++i) c = table[(c ^ data[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

#if defined(__x86_64__)
__attribute__((target("sse4.2"))) inline uint32_t crc32c_hw(const uint8_t *data, size_t len)
{
    uint64_t c = 0xFFFFFFFFu;
    size_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t v;
        memcpy(&v, data + i, 8);
        c = __builtin_ia32_crc32di(c, v);
    }
    uint32_t c32 = (uint32_t)c;
    for (; i < len; ++i) c32 = __builtin_ia32_crc32qi(c32, data[i]);
    return c32 ^ 0xFFFFFFFFu;
}
#endif

// crc32.Checksum(b, crc32.MakeTable(crc32.Castagnoli)) — file_format.go:378,403
inline uint32_t crc32c(const uint8_t *data, size_t len)
{
#if defined(__x86_64__)
    static const bool hw = __builtin_cpu_supports("sse4.2");
    if (hw) return crc32c_hw(data, len);
#endif
    return crc32c_sw(data, len);
}

struct FilterView {       // one filter, words native LE
    const uint64_t *words = nullptr;
    uint64_t m = 0;       // 0 => nil filter
    uint64_t k = 0;
};

struct ParsedFilter {
    bool present = false;
    uint64_t m = 0, k = 0;
    std::vector<uint64_t> words;
};

enum SectionError : int32_t {
    kSectionOk = 0, kSectionTooSmall = -1, kSectionBadHash = -2 /* ErrInvalidHash */, kSectionBadFlags = -3,
    kSectionTruncated = -4, kSectionBadFilter = -5, kSectionTrailing = -6,
};

inline void put_be64(uint8_t *p, uint64_t v) { v = __builtin_bswap64(v); memcpy(p, &v, 8); }
inline uint64_t get_be64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return __builtin_bswap64(v); }
inline void put_le32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
inline uint32_t get_le32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

// encodeFilterSection (file_format.go:343-384)
inline std::vector<uint8_t> encode_filter_section(const FilterView filters[3])
{
    size_t size = 1 + 4;
    for (int c = 0; c < 3; ++c) if (filters[c].m) size += 4 + 24 + 8 * ((filters[c].m + 63) / 64);
    std::vector<uint8_t> out(size);
    size_t pos = 0;
    uint8_t flags = 0;
    for (int c = 0; c < 3; ++c) if (filters[c].m) flags |= (uint8_t)(1u << c);
    out[pos++] = flags;
    for (int c = 0; c < 3; ++c) {
        if (!filters[c].m) continue;
        const uint64_t nw = (filters[c].m + 63) / 64;
        put_le32(&out[pos], (uint32_t)(24 + 8 * nw));
        pos += 4;
        put_be64(&out[pos], filters[c].m); put_be64(&out[pos + 8], filters[c].k); put_be64(&out[pos + 16], filters[c].m);
        pos += 24;
        for (uint64_t i = 0; i < nw; ++i, pos += 8) put_be64(&out[pos], filters[c].words[i]);
    }
    put_le32(&out[pos], crc32c(out.data(), pos));
    return out;
}

// parseFilterSection (file_format.go:392-448): CRC first, then flags, then each present filter.
constexpr uint64_t kMaxHashCount = 1024;   // bloom/v3 EstimateParameters gives k = 30 at p = 1e-9, 100 at 1e-30

inline int32_t parse_filter_section(const uint8_t *section, size_t len, ParsedFilter out[3])
{
    if (len < 5) return kSectionTooSmall;
    const size_t plen = len - 4;
    if (crc32c(section, plen) != get_le32(section + plen)) return kSectionBadHash;
    const uint8_t flags = section[0];
    if (flags & ~7u) return kSectionBadFlags;
    size_t pos = 1;
    for (int c = 0; c < 3; ++c) {
        out[c] = ParsedFilter{};
        if (!((flags >> c) & 1)) continue;
        if (plen - pos < 4) return kSectionTruncated;
        const size_t flen = get_le32(section + pos);
        pos += 4;
        if (flen > plen - pos) return kSectionTruncated;
        if (flen < 24) return kSectionBadFilter;
        const uint64_t m = get_be64(section + pos), k = get_be64(section + pos + 8), blen = get_be64(section + pos + 16);
        // same checks as the device path (bsg_arena_load_sections): the bitset must cover m, and (blen + 63) / 64 must
        // not wrap; k is bounded so that a corrupt section with a valid CRC cannot make a probe loop for minutes
        if (blen > ~0ull - 63 || m > ~0ull - 63 || m == 0 || k == 0 || k > kMaxHashCount) return kSectionBadFilter;   // (m + 63 below must not wrap either)
        const uint64_t nw = (blen + 63) / 64;
        if (nw > (flen - 24) / 8 || (m + 63) / 64 > nw) return kSectionBadFilter;
        out[c].present = true; out[c].m = m; out[c].k = k;
        out[c].words.resize(nw);
        for (uint64_t i = 0; i < nw; ++i) out[c].words[i] = get_be64(section + pos + 24 + 8 * i);
        pos += flen;
    }
    if (pos != plen) return kSectionTrailing;
    return kSectionOk;
}

}  // namespace bsh
