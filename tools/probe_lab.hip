// probe_lab: stand-alone timing harness for k_probe_terms variants on synthetic random filters.
//   ./probe_lab B nw T R iters    (B blocks, nw words per filter, T terms, R rotating arenas)
#include "../bloomsearch_amd/csrc/kernels.hip.h"
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace bsg;

// floor: stream each filter into LDS by LDS-DMA and touch one word
__global__ __launch_bounds__(512) void k_stream_only(const uint64_t *words, const DevDesc *desc, uint64_t *out)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const DevDesc d = desc[(uint64_t)blockIdx.x * 3 + 2];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t nw = (d.m + 63) >> 6;
    const uint32_t nbytes = (uint32_t)(((nw + 1) >> 1) << 4);
    const char *g = reinterpret_cast<const char *>(words + d.word_off);
    char *image = reinterpret_cast<char *>(lds64);
    for (uint32_t c = wave * 1024u; c < nbytes; c += 8 * 1024u) {
        const uint32_t boff = c + lane * 16u;
        if (boff < nbytes) __builtin_amdgcn_global_load_lds((glb_void *)(g + boff), (lds_void *)(image + c), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = lds64[nw - 1];
}
// floor 2: plain register streaming, grid-stride, no LDS
__global__ __launch_bounds__(256) void k_read_only(const u32x4 *p, uint64_t n16, uint32_t *out)
{
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) acc ^= p[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) out[0] = 1;
}

// variant: no descriptor load — address from blockIdx * stride, size from args (quantifies the dependent-load cost)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_stream_nodesc(const uint64_t *words, uint64_t stride, uint32_t nw, uint64_t *out)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nbytes = (uint32_t)(((nw + 1) >> 1) << 4);
    const char *g = reinterpret_cast<const char *>(words + stride * blockIdx.x);
    char *image = reinterpret_cast<char *>(lds64);
    for (uint32_t c = wave * 1024u; c < nbytes; c += (THREADS / 64) * 1024u) {
        const uint32_t boff = c + lane * 16u;
        if (boff < nbytes) __builtin_amdgcn_global_load_lds((glb_void *)(g + boff), (lds_void *)(image + c), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = lds64[nw - 1];
}
// variant: plain register loads (no LDS), one workgroup per filter
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_stream_regs(const uint64_t *words, uint64_t stride, uint32_t nw, uint64_t *out)
{
    const u32x4 *p = reinterpret_cast<const u32x4 *>(words + stride * blockIdx.x);
    const uint32_t n16 = (nw + 1) >> 1;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t i = threadIdx.x; i < n16; i += THREADS) acc ^= p[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) out[blockIdx.x] = 1;
}
__global__ void k_empty(uint64_t *out) { if (threadIdx.x == 9999) out[0] = 1; }

int main(int argc, char **argv)
{
    const uint32_t B = argc > 1 ? atoi(argv[1]) : 1000, nw = argc > 2 ? atoi(argv[2]) : 4404;
    const uint32_t T = argc > 3 ? atoi(argv[3]) : 29, R = argc > 4 ? atoi(argv[4]) : 16, iters = argc > 5 ? atoi(argv[5]) : 64;
    const uint64_t stride = (nw + 15) / 16 * 16;
    const uint64_t arena_words = stride * B + 256;
    std::mt19937_64 rng(1);
    std::vector<uint64_t> hw(arena_words);
    for (auto &x : hw) x = rng() | rng();  // ~75% ones so probes run several rounds
    std::vector<DevDesc> hd(B * 3);
    const uint64_t m = (uint64_t)nw * 64 - 13;
    for (uint32_t b = 0; b < B; ++b)
        for (int c = 0; c < 3; ++c) hd[b * 3 + c] = DevDesc{stride * b, c == 2 ? m : 0, c == 2 ? (~0ULL / m) : 0, 10, 0};
    std::vector<uint64_t *> dw(R);
    for (uint32_t r = 0; r < R; ++r) { CHECK(hipMalloc(&dw[r], arena_words * 8)); CHECK(hipMemcpy(dw[r], hw.data(), arena_words * 8, hipMemcpyHostToDevice)); }
    DevDesc *dd; CHECK(hipMalloc(&dd, hd.size() * sizeof(DevDesc))); CHECK(hipMemcpy(dd, hd.data(), hd.size() * sizeof(DevDesc), hipMemcpyHostToDevice));
    const uint32_t Tp = (T + 63) / 64 * 64;
    std::vector<uint64_t> hth(4 * Tp);
    for (auto &x : hth) x = rng();
    uint64_t *dth, *dV, *dout; CHECK(hipMalloc(&dth, hth.size() * 8)); CHECK(hipMemcpy(dth, hth.data(), hth.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&dV, (uint64_t)kMaxGroupArenas * ((B + 63) / 64) * (Tp / 64) * 64 * 8)); CHECK(hipMalloc(&dout, B * 8 + 64));
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    ProbeArgs a{}; a.th = dth; a.V = dV; a.Tp = Tp; a.Wt = Tp / 64; a.lds_cap_words = 8192;
    a.kind[0] = 2; a.term_begin[0] = 0; a.term_count[0] = T;
    a.n_arenas = 1; a.max_blocks = B; a.compact_rounds = 2;
    const uint32_t G = (B + 63) / 64;
    // arena slot i of a dispatch group reads replica r: per-arena pointers ride in the kernel arguments
    ArenaTable<kMaxGroupArenas> tab{};
    auto set_arena = [&](uint32_t slot, uint32_t r) { tab.ar[slot] = ArenaRef{dw[r], dd, (uint64_t)slot * G * (Tp / 64) * 64, 0, B, G}; };
    set_arena(0, 0);
    const size_t head = probe_lds_head_bytes(Tp / 64);
    a.lds_image_bytes = (uint32_t)((nw + 1) / 2 * 16);
    const size_t lds = head + a.lds_image_bytes;
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 8; ++i) launch(i % R, nullptr, nullptr);
        CHECK(hipStreamSynchronize(st));
        double tot = 0, tot2 = 0; float best = 1e9, best2 = 1e9;
        for (uint32_t i = 0; i < iters; ++i) {   // (a) events bracketing the launch
            CHECK(hipEventRecord(e0, st)); launch(i % R, nullptr, nullptr); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; best = ms < best ? ms : best;
        }
        for (uint32_t i = 0; i < iters; ++i) {   // (b) the dispatch's own start/stop timestamps
            launch(i % R, e0, e1); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); tot2 += ms; best2 = ms < best2 ? ms : best2;
        }
        const double bytes = (double)B * nw * 8;
        printf("%-22s bracket avg %7.2f min %7.2f us | dispatch avg %7.2f min %7.2f us  %7.1f GB/s avg %7.1f best\n", name,
               tot / iters * 1e3, best * 1e3, tot2 / iters * 1e3, best2 * 1e3, bytes / (tot2 / iters * 1e-3) / 1e9, bytes / (best2 * 1e-3) / 1e9);
    };
    printf("B=%u nw=%u (%.1f KB) T=%u R=%u  bytes/launch=%.1f MB\n", B, nw, nw * 8 / 1024.0, T, R, (double)B * nw * 8 / 1e6);
    timeit("empty kernel", [&](int, hipEvent_t a0, hipEvent_t a1) { hipExtLaunchKernelGGL(k_empty, dim3(B), dim3(512), 0, st, a0, a1, 0, dout); });
    timeit("read_only grid=2048", [&](int r, hipEvent_t a0, hipEvent_t a1) { hipExtLaunchKernelGGL(k_read_only, dim3(2048), dim3(256), 0, st, a0, a1, 0, (const u32x4 *)dw[r], (uint64_t)(stride * B / 2), (uint32_t *)dout); });
    timeit("read_only grid=8192", [&](int r, hipEvent_t a0, hipEvent_t a1) { hipExtLaunchKernelGGL(k_read_only, dim3(8192), dim3(256), 0, st, a0, a1, 0, (const u32x4 *)dw[r], (uint64_t)(stride * B / 2), (uint32_t *)dout); });
    timeit("stream_only (LDS-DMA)", [&](int r, hipEvent_t a0, hipEvent_t a1) { hipExtLaunchKernelGGL(k_stream_only, dim3(B), dim3(512), lds, st, a0, a1, 0, (const uint64_t *)dw[r], (const DevDesc *)dd, dout); });
    timeit("stream nodesc 512", [&](int r, hipEvent_t a0, hipEvent_t a1) { hipExtLaunchKernelGGL(k_stream_nodesc<512>, dim3(B), dim3(512), lds, st, a0, a1, 0, (const uint64_t *)dw[r], stride, nw, dout); });
    timeit("stream nodesc 256", [&](int r, hipEvent_t a0, hipEvent_t a1) { hipExtLaunchKernelGGL(k_stream_nodesc<256>, dim3(B), dim3(256), lds, st, a0, a1, 0, (const uint64_t *)dw[r], stride, nw, dout); });
    timeit("stream nodesc 1024", [&](int r, hipEvent_t a0, hipEvent_t a1) { hipExtLaunchKernelGGL(k_stream_nodesc<1024>, dim3(B), dim3(1024), lds, st, a0, a1, 0, (const uint64_t *)dw[r], stride, nw, dout); });
    timeit("stream regs 512", [&](int r, hipEvent_t a0, hipEvent_t a1) { hipExtLaunchKernelGGL(k_stream_regs<512>, dim3(B), dim3(512), 0, st, a0, a1, 0, (const uint64_t *)dw[r], stride, nw, dout); });
    timeit("stream regs 256", [&](int r, hipEvent_t a0, hipEvent_t a1) { hipExtLaunchKernelGGL(k_stream_regs<256>, dim3(B), dim3(256), 0, st, a0, a1, 0, (const uint64_t *)dw[r], stride, nw, dout); });
    {   // per-arena dispatch time of k_probe_terms: is the spread address dependent?
        std::vector<double> per(R, 0); std::vector<int> cnt(R, 0);
        for (uint32_t i = 0; i < iters * 2; ++i) {
            const int r = i % R; set_arena(0, r);
            hipExtLaunchKernelGGL(k_probe_terms, dim3(B, 1), dim3(kProbeThreads), lds, st, e0, e1, 0, a, tab);
            CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); per[r] += ms; cnt[r]++;
        }
        printf("per-arena avg us:"); for (uint32_t r = 0; r < R; ++r) printf(" %.2f", per[r] / cnt[r] * 1e3); printf("\n");
        // same arena every time (cache resident) for comparison
        double tot = 0; set_arena(0, 0);
        for (uint32_t i = 0; i < iters; ++i) {
            hipExtLaunchKernelGGL(k_probe_terms, dim3(B, 1), dim3(kProbeThreads), lds, st, e0, e1, 0, a, tab);
            CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
        }
        printf("same arena (MALL-resident) avg %.2f us\n", tot / iters * 1e3);
    }
    for (int flag : {0, (int)hipExtAnyOrderLaunch}) {   // do back-to-back independent launches overlap their ramps?
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipStreamSynchronize(st));
            CHECK(hipEventRecord(e0, st));
            const int N = 256;
            for (int i = 0; i < N; ++i) { set_arena(0, i % R); hipExtLaunchKernelGGL(k_probe_terms, dim3(B, 1), dim3(kProbeThreads), lds, st, nullptr, nullptr, flag, a, tab); }
            CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("back-to-back x%d flag=%d: %.2f us per launch\n", N, flag, ms / N * 1e3);
        }
    }
    timeit("k_probe_terms", [&](int r, hipEvent_t a0, hipEvent_t a1) { set_arena(0, r); hipExtLaunchKernelGGL(k_probe_terms, dim3(B, 1), dim3(kProbeThreads), lds, st, a0, a1, 0, a, tab); });
    // one dispatch over a GROUP of arenas (grid z): what a launch's ramp + completion cost per arena
    for (uint32_t n : {2u, 4u, 8u, 16u, 32u, 64u}) {
        if (n > R) break;
        a.n_arenas = n;
        double tot = 0;
        for (uint32_t i = 0; i < iters + 4; ++i) {
            for (uint32_t s = 0; s < n; ++s) set_arena(s, (i * n + s) % R);
            hipExtLaunchKernelGGL(k_probe_terms, dim3(B, 1, n), dim3(kProbeThreads), lds, st, e0, e1, 0, a, tab);
            CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (i >= 4) tot += ms;
        }
        printf("group of %2u arenas: %.2f us per dispatch, %.2f us per arena, %.1f GB/s\n", n, tot / iters * 1e3, tot / iters * 1e3 / n,
               (double)B * nw * 8 * n / (tot / iters * 1e-3) / 1e9);
    }
    return 0;
}
