// ubench_mod.hip — throughput of "x mod m" variants for 64-bit x, m < 2^21 (block filters), on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_mod tools/ubench_mod.hip && tools/ubench_mod
// Variants: 0 none (loop + xorshift only), 1 Barrett 64x64 mul-high + low-32 remainder (kernels.hip.h mod_m32),
//           2 FP64: t = xh * (2^32 mod m) + xl (exact below 2^53), q = floor(t * 1/m), r = t - q m, one fix-up,
//           3 two 32-bit steps in FP32/FP64 mix, 4 the compiler's own x % m.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct ModC { uint32_t m; uint64_t magic; double inv, md, R; float invf; };

__device__ __forceinline__ uint32_t mod_barrett(uint64_t x, const ModC &c)
{
    const uint64_t q = __umul64hi(x, c.magic);
    uint32_t r = (uint32_t)x - (uint32_t)q * c.m;
    if (r >= c.m) r -= c.m;
    return r;
}
__device__ __forceinline__ uint32_t mod_fp64(uint64_t x, const ModC &c)
{
    const double t = fma((double)(uint32_t)(x >> 32), c.R, (double)(uint32_t)x);   // < 2^21 * 2^32 + 2^32 < 2^53: exact
    const double q = floor(t * c.inv);                                              // true quotient or one less (inv rounded down)
    uint32_t r = (uint32_t)fma(-q, c.md, t);
    if (r >= c.m) r -= c.m;
    return r;
}
__device__ __forceinline__ uint32_t mod_two_step(uint64_t x, const ModC &c)
{
    // xh mod m by an FP32 quotient estimate (xh < 2^32, m < 2^21: estimate off by <= 2), then the FP64 step on a 53-bit value
    uint32_t xh = (uint32_t)(x >> 32);
    uint32_t q1 = (uint32_t)((float)xh * c.invf);
    int32_t y = (int32_t)(xh - q1 * c.m);
    if (y < 0) y += c.m;
    if ((uint32_t)y >= c.m) y -= c.m;
    const double t = fma((double)(uint32_t)y, 4294967296.0, (double)(uint32_t)x);
    const double q = floor(t * c.inv);
    uint32_t r = (uint32_t)fma(-q, c.md, t);
    if (r >= c.m) r -= c.m;
    return r;
}

template <int V>
__global__ __launch_bounds__(512) void k(const ModC c, uint32_t iters, uint64_t seed, uint32_t *out, uint32_t *bad)
{
    uint64_t x = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        uint32_t r;
        if (V == 0) r = (uint32_t)x;
        else if (V == 1) r = mod_barrett(x, c);
        else if (V == 2) r = mod_fp64(x, c);
        else if (V == 3) r = mod_two_step(x, c);
        else r = (uint32_t)(x % c.m);
        acc += r;
        if (V == 2 || V == 3 || V == 1) { if (i < 64 && r != (uint32_t)(x % c.m)) atomicAdd(bad, 1u); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int V>
double run(const ModC &c, uint32_t iters, uint32_t *d_out, uint32_t *d_bad)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<V>, dim3(256 * 4), dim3(512), 0, 0, c, 64u, 1ull, d_out, d_bad);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<V>, dim3(256 * 4), dim3(512), 0, 0, c, iters, 12345ull, d_out, d_bad);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    uint32_t *d_out, *d_bad;
    hipMalloc(&d_out, 256 * 4 * 512 * 4);
    hipMalloc(&d_bad, 4);
    hipMemset(d_bad, 0, 4);
    const uint32_t iters = 20000;
    for (uint32_t m : {281573u, 1453u, 2000003u}) {
        ModC c;
        c.m = m;
        c.magic = ~0ULL / m + (((m & (m - 1)) == 0) ? 1 : 0);
        c.md = (double)m;
        c.inv = (1.0 / (double)m) * (1.0 - 1e-15);   // rounded DOWN a hair: the quotient estimate never overshoots
        c.invf = (float)(1.0 / (double)m) * (1.0f - 1e-6f);
        c.R = (double)((1ull << 32) % m);
        const double t0 = run<0>(c, iters, d_out, d_bad);
        const double t1 = run<1>(c, iters, d_out, d_bad);
        const double t2 = run<2>(c, iters, d_out, d_bad);
        const double t3 = run<3>(c, iters, d_out, d_bad);
        const double t4 = run<4>(c, iters, d_out, d_bad);
        uint32_t bad = 0;
        hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
        // 256 CUs x 4 WGs x 8 waves = 8192 waves = 32 per CU; per SIMD 8 waves x iters wave-ops
        const double waveops = 8.0 * iters;   // per SIMD
        printf("m=%u  base %.3f ms | barrett %.3f (+%.1f ns/wave-op) | fp64 %.3f (+%.1f) | two-step %.3f (+%.1f) | x%%m %.3f (+%.1f) | mismatches %u\n", m, t0,
               t1, (t1 - t0) * 1e6 / waveops, t2, (t2 - t0) * 1e6 / waveops, t3, (t3 - t0) * 1e6 / waveops, t4, (t4 - t0) * 1e6 / waveops, bad);
    }
    return 0;
}
