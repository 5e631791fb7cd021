// or_lab: stand-alone timing harness for the fixed-geometry OR-reduce (BASELINE configs[4], local half).
//   ./or_lab B nw iters      (B block filters of nw words each; rotates R copies so every pass streams from HBM)
// Variants: the product kernel of round 2 (registers, 4 loads in flight), registers with the block offsets in LDS and U
// loads in flight, and LDS-DMA (global_load_lds ... nt) with the OR taken out of LDS.
#include "../bloomsearch_amd/csrc/kernels.hip.h"
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace bsg;

// ---- registers: offsets of the group's blocks in LDS, U independent 16-byte loads per lane in flight ----
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_or_regs(const uint64_t *words, const DevDesc *desc, uint32_t n_blocks, uint32_t kind, uint64_t n_words,
                                                  uint64_t *out, uint32_t group)
{
    __shared__ uint64_t offs[512];
    const uint32_t b0 = blockIdx.y * group, nb = min(group, n_blocks - b0);
    for (uint32_t i = threadIdx.x; i < nb; i += 256) {
        const DevDesc d = desc[(uint64_t)(b0 + i) * 3 + kind];
        offs[i] = d.m ? d.word_off : ~0ull;
    }
    __syncthreads();
    const uint64_t pair = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (pair * 2 >= n_words) return;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t i = 0; i < nb; i += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t o = i + u < nb ? offs[i + u] : ~0ull;
            if (o != ~0ull) {
                const u32x4 *p = reinterpret_cast<const u32x4 *>(words + o + pair * 2);
                v[u] = NT ? __builtin_nontemporal_load(p) : *p;
            } else v[u] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc |= v[u];
    }
    const uint64_t lo = (uint64_t)acc.x | ((uint64_t)acc.y << 32), hi = (uint64_t)acc.z | ((uint64_t)acc.w << 32);
    const bool two = pair * 2 + 1 < n_words;
    if (gridDim.y == 1) { out[pair * 2] = lo; if (two) out[pair * 2 + 1] = hi; }
    else {
        if (lo) atomicOr((unsigned long long *)&out[pair * 2], (unsigned long long)lo);
        if (two && hi) atomicOr((unsigned long long *)&out[pair * 2 + 1], (unsigned long long)hi);
    }
}

// ---- LDS-DMA: a workgroup owns a tile of THREADS x 16 bytes of the bitset; per round it pulls that tile of NB blocks into
// LDS (every wave issues NB 1 KiB pieces back to back, no VGPR round trip), waits once, and every lane ORs its 16 bytes of
// the NB images.  DB: two LDS halves, the DMA of round r + 1 is issued before the OR of round r. ----
template <int THREADS, int NB, bool DB>
__global__ __launch_bounds__(THREADS) void k_or_dma(const uint64_t *words, const DevDesc *desc, uint32_t n_blocks, uint32_t kind, uint64_t n_words,
                                                     uint64_t *out, uint32_t group)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    __shared__ uint64_t offs[512];
    constexpr uint32_t kTile = THREADS * 16;                       // bytes of the bitset per workgroup
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b0 = blockIdx.y * group, nb = min(group, n_blocks - b0);
    for (uint32_t i = tid; i < nb; i += THREADS) {
        const DevDesc d = desc[(uint64_t)(b0 + i) * 3 + kind];
        offs[i] = d.m ? d.word_off : ~0ull;
    }
    __syncthreads();
    const uint64_t byte0 = (uint64_t)blockIdx.x * kTile;           // tile start inside a filter
    const uint64_t total = ((n_words + 1) / 2) * 16;               // bytes per filter rounded to 16 (filters are 128-byte aligned and padded)
    const bool in = byte0 + (uint64_t)tid * 16 < total;
    char *image = reinterpret_cast<char *>(lds64);
    auto issue = [&](uint32_t r, uint32_t half) {
        const uint32_t base = r * NB;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const uint64_t o = base + u < nb ? offs[base + u] : ~0ull;
            if (o != ~0ull && in) {
                const char *g = reinterpret_cast<const char *>(words + o) + byte0 + (uint64_t)wave * 1024 + lane * 16u;
                __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)(image + (half * NB + u) * kTile + wave * 1024), 16, 0, BSG_DMA_AUX);
            }
        }
    };
    const uint32_t rounds = (nb + NB - 1) / NB;
    u32x4 acc = {0, 0, 0, 0};
    if (DB) issue(0, 0);
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t half = DB ? (r & 1u) : 0u;
        if (!DB) issue(r, 0);
        if (DB && r + 1 < rounds) {
            issue(r + 1, half ^ 1u);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");       // the older NB pieces have landed (in-order return)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // each wave reads back exactly the 1 KiB pieces it wrote itself: no workgroup barrier needed
        const uint32_t base = r * NB;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (base + u < nb && offs[base + u] != ~0ull && in)
                acc |= *reinterpret_cast<const u32x4 *>(image + (half * NB + u) * kTile + tid * 16);
        }
    }
    if (!in) return;
    const uint64_t pair = byte0 / 16 + tid;
    const uint64_t lo = (uint64_t)acc.x | ((uint64_t)acc.y << 32), hi = (uint64_t)acc.z | ((uint64_t)acc.w << 32);
    const bool two = pair * 2 + 1 < n_words;
    if (gridDim.y == 1) { out[pair * 2] = lo; if (two) out[pair * 2 + 1] = hi; }
    else {
        if (lo) atomicOr((unsigned long long *)&out[pair * 2], (unsigned long long)lo);
        if (two && hi) atomicOr((unsigned long long *)&out[pair * 2 + 1], (unsigned long long)hi);
    }
}

// ---- registers, generalized: THREADS lanes, each owning SPAN 16-byte pieces of a (THREADS x 16 x SPAN)-byte tile, U blocks per trip ----
template <int THREADS, int U, int SPAN>
__global__ __launch_bounds__(THREADS) void k_or_regs2(const uint64_t *words, const DevDesc *desc, uint32_t n_blocks, uint32_t kind, uint64_t n_words,
                                                       uint64_t *out, uint32_t group)
{
    __shared__ uint64_t offs[512];
    const uint32_t b0 = blockIdx.y * group, nb = min(group, n_blocks - b0);
    for (uint32_t i = threadIdx.x; i < nb; i += THREADS) {
        const DevDesc d = desc[(uint64_t)(b0 + i) * 3 + kind];
        offs[i] = d.m ? d.word_off : ~0ull;
    }
    __syncthreads();
    const uint64_t n16 = (n_words + 1) / 2;
    const uint64_t p0 = (uint64_t)blockIdx.x * THREADS * SPAN + threadIdx.x;      // 16-byte piece index of span 0
    u32x4 acc[SPAN];
#pragma unroll
    for (int sp = 0; sp < SPAN; ++sp) acc[sp] = u32x4{0, 0, 0, 0};
    for (uint32_t i = 0; i < nb; i += U) {
        u32x4 v[U][SPAN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t o = i + u < nb ? offs[i + u] : ~0ull;
#pragma unroll
            for (int sp = 0; sp < SPAN; ++sp) {
                const uint64_t pc = p0 + (uint64_t)sp * THREADS;
                if (o != ~0ull && pc < n16) v[u][sp] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(words + o) + pc);
                else v[u][sp] = u32x4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int sp = 0; sp < SPAN; ++sp) acc[sp] |= v[u][sp];
    }
#pragma unroll
    for (int sp = 0; sp < SPAN; ++sp) {
        const uint64_t pair = p0 + (uint64_t)sp * THREADS;
        if (pair >= n16) continue;
        const uint64_t lo = (uint64_t)acc[sp].x | ((uint64_t)acc[sp].y << 32), hi = (uint64_t)acc[sp].z | ((uint64_t)acc[sp].w << 32);
        const bool two = pair * 2 + 1 < n_words;
        if (gridDim.y == 1) { out[pair * 2] = lo; if (two) out[pair * 2 + 1] = hi; }
        else {
            if (lo) atomicOr((unsigned long long *)&out[pair * 2], (unsigned long long)lo);
            if (two && hi) atomicOr((unsigned long long *)&out[pair * 2 + 1], (unsigned long long)hi);
        }
    }
}

int main(int argc, char **argv)
{
    const uint32_t B = argc > 1 ? atoi(argv[1]) : 1000, nw = argc > 2 ? atoi(argv[2]) : 44976, iters = argc > 3 ? atoi(argv[3]) : 20;
    const uint64_t stride = (nw + 15) / 16 * 16;
    const uint64_t arena_words = stride * B + 256;
    const uint32_t R = (uint32_t)std::max<uint64_t>(2, (600ull << 20) / (arena_words * 8) + 1);
    std::mt19937_64 rng(1);
    std::vector<uint64_t> hw(arena_words);
    for (auto &x : hw) x = rng() & rng() & rng();
    std::vector<DevDesc> hd(B * 3);
    const uint64_t m = (uint64_t)nw * 64 - 13;
    for (uint32_t b = 0; b < B; ++b)
        for (int c = 0; c < 3; ++c) hd[b * 3 + c] = DevDesc{stride * b, c == 1 ? m : 0, 0, 10, 0};
    std::vector<uint64_t> want(nw, 0);
    for (uint32_t b = 0; b < B; ++b) for (uint32_t i = 0; i < nw; ++i) want[i] |= hw[stride * b + i];
    std::vector<uint64_t *> dw(R);
    for (uint32_t r = 0; r < R; ++r) { CHECK(hipMalloc(&dw[r], arena_words * 8)); CHECK(hipMemcpy(dw[r], hw.data(), arena_words * 8, hipMemcpyHostToDevice)); }
    DevDesc *dd; CHECK(hipMalloc(&dd, hd.size() * sizeof(DevDesc))); CHECK(hipMemcpy(dd, hd.data(), hd.size() * sizeof(DevDesc), hipMemcpyHostToDevice));
    uint64_t *dout; CHECK(hipMalloc(&dout, (size_t)nw * 8 + 64));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double bytes = (double)B * nw * 8 + nw * 8;
    printf("B=%u nw=%u (%.1f KB per filter, %.1f MB per pass), %u rotating copies\n", B, nw, nw * 8 / 1e3, bytes / 1e6, R);
    std::vector<uint64_t> got(nw);
    auto run = [&](const char *name, auto launch) {
        std::vector<float> ms;
        for (uint32_t it = 0; it < iters + 2; ++it) {
            CHECK(hipMemsetAsync(dout, 0, (size_t)nw * 8, st));
            launch(dw[it % R], e0, e1);
            CHECK(hipGetLastError());
            CHECK(hipStreamSynchronize(st));
            float t; CHECK(hipEventElapsedTime(&t, e0, e1));
            if (it >= 2) ms.push_back(t);
        }
        CHECK(hipMemcpy(got.data(), dout, (size_t)nw * 8, hipMemcpyDeviceToHost));
        const bool ok = memcmp(got.data(), want.data(), (size_t)nw * 8) == 0;
        std::sort(ms.begin(), ms.end());
        const float med = ms[ms.size() / 2];
        printf("%-34s median %8.1f us  min %8.1f  -> %6.0f GB/s (%.2f of 8 TB/s)  %s\n", name, med * 1e3, ms[0] * 1e3, bytes / med / 1e6, bytes / med / 1e6 / 8000.0, ok ? "ok" : "WRONG");
    };
    const uint32_t gx = (uint32_t)(((uint64_t)(nw + 1) / 2 + 255) / 256);
#define OPTIN(k) CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024))
    OPTIN((k_or_dma<256, 8, false>)); OPTIN((k_or_dma<256, 16, false>)); OPTIN((k_or_dma<256, 8, true>)); OPTIN((k_or_dma<512, 8, false>)); OPTIN((k_or_dma<512, 4, true>));
    const uint64_t n16 = ((uint64_t)nw + 1) / 2;
    for (uint32_t group : {64u, 128u, 256u, 500u}) {
        const uint32_t gy = (B + group - 1) / group;
        char name[96];
        if (group <= kOrMaxGroup) {
            snprintf(name, sizeof name, "product k_or_reduce_blocks g=%u", group);
            run(name, [&](uint64_t *w, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_or_reduce_blocks, dim3((uint32_t)((n16 + kOrThreads - 1) / kOrThreads), gy), dim3(kOrThreads), 0, st, a, b, 0, w, dd, B, 1u, (uint64_t)nw, dout, group); });
        }
        snprintf(name, sizeof name, "regs U=16 nt g=%u", group);
        run(name, [&](uint64_t *w, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_or_regs<16, true>), dim3(gx, gy), dim3(256), 0, st, a, b, 0, w, dd, B, 1u, (uint64_t)nw, dout, group); });
#define R2(T, U, SP) do { snprintf(name, sizeof name, "regs2 T=%d U=%d SPAN=%d g=%u", T, U, SP, group); \
        const uint32_t gxx = (uint32_t)((n16 + (uint64_t)T * SP - 1) / ((uint64_t)T * SP)); \
        run(name, [&](uint64_t *w, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_or_regs2<T, U, SP>), dim3(gxx, gy), dim3(T), 0, st, a, b, 0, w, dd, B, 1u, (uint64_t)nw, dout, group); }); } while (0)
        R2(256, 16, 1); R2(512, 16, 1); R2(128, 16, 1); R2(256, 8, 2); R2(256, 4, 4); R2(128, 8, 2); R2(512, 8, 2); R2(256, 24, 1); R2(64, 16, 1); R2(64, 8, 2);
        snprintf(name, sizeof name, "dma T=256 NB=8 g=%u", group);
        const uint32_t gx256 = (uint32_t)((n16 * 16 + 4095) / 4096);
        run(name, [&](uint64_t *w, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_or_dma<256, 8, false>), dim3(gx256, gy), dim3(256), 8 * 4096, st, a, b, 0, w, dd, B, 1u, (uint64_t)nw, dout, group); });
    }
    return 0;
}
