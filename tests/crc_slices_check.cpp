// Host check of the slicing arithmetic k_decode_sections relies on (bloomsearch_amd/csrc/crc_slices.h), built with plain g++ by
// tests/test_crc_slices.py: a payload cut into decode_splits() slices counted from its END, each checksummed with a zero initial
// value, shifted by x^(8 U j) and XOR-ed, plus crc_init_image(), must be the CRC-32C the reference stores behind a filter section
// (crc32.Checksum over the Castagnoli table, file_format.go:343-384) — computed here bit by bit, sharing nothing with the header.
// Test infrastructure only.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "crc_slices.h"

static uint32_t crc_bits(const uint8_t *p, size_t n, uint32_t crc)     // reflected CRC-32C register update, one bit at a time
{
    for (size_t i = 0; i < n; ++i) {
        crc ^= p[i];
        for (int b = 0; b < 8; ++b) crc = (crc & 1u) ? (crc >> 1) ^ 0x82F63B78u : crc >> 1;
    }
    return crc;
}

int main(int argc, char **argv)
{
    uint32_t x2n[64];
    uint32_t p = 1u << 30;
    x2n[0] = p;
    for (int n = 1; n < 64; ++n) x2n[n] = p = bsg::crc_multmodp(p, p);
    std::vector<uint64_t> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back(strtoull(argv[i], nullptr, 10));
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    int bad = 0;
    for (uint64_t P : sizes) {
        std::vector<uint8_t> buf(P);
        for (auto &b : buf) { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; b = (uint8_t)(seed >> 32); }
        const uint32_t want = crc_bits(buf.data(), P, 0xFFFFFFFFu) ^ 0xFFFFFFFFu;
        const uint32_t len = (uint32_t)P + 4, U = bsg::decode_unit((uint32_t)P), n = bsg::decode_splits(len);
        uint64_t covered = 0;
        uint32_t total = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t hi = (uint32_t)P - j * U, lo = hi > U ? hi - U : 0u;          // as the kernel cuts them
            covered += hi - lo;
            const uint32_t raw = crc_bits(buf.data() + lo, hi - lo, 0);
            total ^= j ? bsg::crc_multmodp(bsg::crc_x2nmodp((uint64_t)U * j, 3, x2n), raw) : raw ^ bsg::crc_init_image(P, x2n);
        }
        const bool ok = total == want && covered == P && n >= 1 && n <= bsg::kDecodeMaxSplits && U % 64 == 0 && (uint64_t)U * n >= P &&
                        (n == 1 || (uint64_t)U * (n - 1) < P);
        printf("%llu %u %u %08x %08x %s\n", (unsigned long long)P, U, n, total, want, ok ? "ok" : "BAD");
        bad += !ok;
    }
    // the shift table's indexing: x has order 2^31 - 1 under this polynomial (not 2^32 - 1), so entry k + 31 repeats entry k and a
    // period-32 table would be wrong; x^(8 * 2^28) computed two ways
    if (x2n[31] != x2n[0] || x2n[32] == x2n[0]) { printf("x2n period BAD\n"); ++bad; }
    if (bsg::crc_x2nmodp(1ull << 28, 3, x2n) != x2n[31]) { printf("x2nmodp(2^28 bytes) BAD\n"); ++bad; }
    // 123456789 -> 0xE3069283 through the init image alone (one slice)
    {
        const uint8_t *s = (const uint8_t *)"123456789";
        if ((crc_bits(s, 9, 0) ^ bsg::crc_init_image(9, x2n)) != 0xE3069283u) { printf("check value BAD\n"); ++bad; }
    }
    return bad ? 1 : 0;
}
