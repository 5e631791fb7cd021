// build_lab: time k_build variants on synthetic entries.  ./build_lab N_FILTERS ENTRIES_PER_FILTER ENTRY_LEN [K [ENTRY_LEN_MAX]]
// (with ENTRY_LEN_MAX the lengths are uniform in [ENTRY_LEN, ENTRY_LEN_MAX]: the lanes of a wave then disagree on the tail's shape,
// as real tokens do)
#include "../bloomsearch_amd/csrc/kernels.hip.h"
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace bsg;
int main(int argc, char **argv)
{
    const uint32_t F = argc > 1 ? atoi(argv[1]) : 600, E = argc > 2 ? atoi(argv[2]) : 19600, L = argc > 3 ? atoi(argv[3]) : 13, K = argc > 4 ? atoi(argv[4]) : 10, Lmax = argc > 5 ? atoi(argv[5]) : L, lds_pad = argc > 6 ? atoi(argv[6]) : 0;   // lds_pad: extra LDS bytes per workgroup (fewer workgroups per CU)
    const uint64_t n = (uint64_t)F * E;
    std::mt19937_64 rng(1);
    std::vector<uint32_t> off(n + 1);
    off[0] = 0;
    for (uint64_t i = 0; i < n; ++i) off[i + 1] = off[i] + L + (Lmax > L ? (uint32_t)(rng() % (Lmax - L + 1)) : 0);
    std::vector<uint8_t> bytes((size_t)off[n] + 64);
    for (auto &b : bytes) b = (uint8_t)rng();
    const uint64_t m = 281629, nw = (m + 63) / 64, stride = (nw + 15) / 16 * 16;
    std::vector<DevDesc> desc(F);
    std::vector<BuildItem> items(F);
    for (uint32_t f = 0; f < F; ++f) { desc[f] = DevDesc{stride * f, m, ~0ULL / m, K, 0}; items[f] = BuildItem{f, f * E, (f + 1) * E, 1}; }
    uint8_t *db; uint32_t *doff; DevDesc *dd; BuildItem *di; uint64_t *dout;
    CHECK(hipMalloc(&db, bytes.size())); CHECK(hipMemcpy(db, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&doff, off.size() * 4)); CHECK(hipMemcpy(doff, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dd, F * sizeof(DevDesc))); CHECK(hipMemcpy(dd, desc.data(), F * sizeof(DevDesc), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&di, F * sizeof(BuildItem))); CHECK(hipMemcpy(di, items.data(), F * sizeof(BuildItem), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dout, stride * F * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    BuildArgs a{db, doff, nullptr, di, dd, dout};
    float best = 1e9, tot = 0;
    for (int it = 0; it < 6; ++it) {
        hipExtLaunchKernelGGL(k_build, dim3(F), dim3(kBuildThreads), (uint32_t)(nw * 8) + lds_pad, 0, e0, e1, 0, a);
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it) { tot += ms; best = ms < best ? ms : best; }
    }
    printf("F=%u E=%u L=%u K=%u lds+%u threads=%d: avg %.1f us  best %.1f us  (%.2f ns/entry, %.0f GB/s entry bytes+offsets)\n", F, E, L, K, lds_pad, kBuildThreads,
           tot / 5 * 1e3, best * 1e3, tot / 5 * 1e6 / n, ((double)off[n] + 4.0 * n) / (tot / 5 * 1e-3) / 1e9);
    return 0;
}
