// Host-side sweep of bsg::mod_f64 / make_modf64 (bloomsearch_amd/csrc/kernels.hip.h) against the % operator: the SAME functions the
// build and the many-term probe kernels run (they are __host__ __device__; the arithmetic is an integer multiply-add and one IEEE fma
// on both sides).  No GPU: built with hipcc, only host code runs.  tests/test_modf64.py drives it.
//   every m in [64, 2^12], m around every power of two up to 2^19, random m; per m: x = q m + d for q at both ends of the range and
//   random, d in {-1, 0, 1}, the extreme 32-bit halves, random x.
#include "../bloomsearch_amd/csrc/kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

static uint64_t magic_of(uint64_t m) { return m == 1 ? ~0ull : (uint64_t)(((unsigned __int128)1 << 64) / m); }   // floor(2^64 / m), as barrett_magic (bloomgpu.hip)

int main(int argc, char **argv)
{
    const uint64_t per_m = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2000;
    std::mt19937_64 rng(20261001);
    std::vector<uint64_t> ms;
    for (uint64_t m = 64; m <= 4096; ++m) ms.push_back(m);
    for (int b = 12; b <= 19; ++b)
        for (int64_t d = -3; d <= 3; ++d) { const uint64_t m = (uint64_t)((int64_t)(1ull << b) + d); if (bsg::modf64_ok(m)) ms.push_back(m); }
    for (int i = 0; i < 3000; ++i) ms.push_back(64 + rng() % (bsg::kModF64MaxM - 63));
    ms.push_back(281629); ms.push_back(287552); ms.push_back(bsg::kModF64MaxM);
    uint64_t checked = 0;
    for (uint64_t m : ms) {
        if (!bsg::modf64_ok(m)) { printf("m = %llu is outside the route's range\n", (unsigned long long)m); return 1; }
        const bsg::ModF64 f = bsg::make_modf64(m, magic_of(m));
        if (f.T != (uint32_t)((1ull << 32) % m) || !(f.inv < 1.0 / (double)m)) { printf("m = %llu: constants\n", (unsigned long long)m); return 1; }
        const uint64_t top = ~0ull / m;
        std::vector<uint64_t> xs = {0, 1, m - 1, m, m + 1, ~0ull, ~0ull - 1, ~0ull - m, 1ull << 32, (1ull << 32) - 1, (1ull << 32) + 1,
                                    0xFFFFFFFF00000000ull, 0x00000000FFFFFFFFull, 1ull << 63, (1ull << 63) - 1, 1ull << 52, (1ull << 52) - 1, 1ull << 53};
        const uint64_t qs[] = {1, 2, 3, top, top - 1, top / 2, (1ull << 32) / m, (1ull << 32) / m + 1, (1ull << 33) / m};
        for (uint64_t q : qs) for (int d = -1; d <= 1; ++d) { const unsigned __int128 v = (unsigned __int128)q * m + d; if (v <= ~0ull) xs.push_back((uint64_t)v); }
        for (uint64_t i = 0; i < per_m; ++i) {
            const uint64_t q = rng() % (top + 1);
            xs.push_back(q * m);                                         // exact multiples: where the quotient estimate is at its edge
            if (q * m + (m - 1) >= q * m) xs.push_back(q * m + (m - 1));
            xs.push_back(rng());
            xs.push_back(((rng() & 1) ? 0xFFFFFFFF00000000ull : 0) | (uint32_t)rng());
        }
        for (uint64_t x : xs) {
            const uint32_t got = bsg::mod_f64(x, f);
            if (got != (uint32_t)(x % m)) { printf("m = %llu x = %llu: mod_f64 %u, %% %llu\n", (unsigned long long)m, (unsigned long long)x, got, (unsigned long long)(x % m)); return 1; }
        }
        checked += xs.size();
    }
    printf("%zu moduli, %llu values: ok\n", ms.size(), (unsigned long long)checked);
    return 0;
}
