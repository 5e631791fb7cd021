// Micro-benchmark: VALU issue rates on gfx950 for the integer / fp64 ops the probe's
// modulo can be built from.  Prints lane-ops per clock per CU (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 4096;
template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t a0, uint32_t b0)
{
    uint32_t a[8], b = b0 | 1;
    uint64_t w[8];
    double d[8];
    for (int j = 0; j < 8; ++j) { a[j] = a0 + threadIdx.x * 8 + j; w[j] = ((uint64_t)a[j] << 32) | (a[j] * 7u); d[j] = a[j]; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) a[j] = a[j] * b + 1;                                  // v_mul_lo_u32 (+add)
            if (OP == 1) a[j] = __umulhi(a[j], b) + a[j];                        // v_mul_hi_u32 (+add)
            if (OP == 2) w[j] = (uint64_t)(uint32_t)w[j] * b + w[j];             // v_mad_u64_u32
            if (OP == 3) a[j] = __umul24(a[j], b) + 1;                           // v_mul_u32_u24 / mad_u32_u24
            if (OP == 4) a[j] = a[j] + a[(j + 1) & 7];                            // v_add_u32 (operands the compiler cannot fold)
            if (OP == 5) w[j] = w[j] + w[(j + 1) & 7];                            // 64-bit add (v_lshl_add_u64)
            if (OP == 6) d[j] = __builtin_fma(d[j], 1.0000001, 0.5);             // v_fma_f64
            if (OP == 7) w[j] = __umul64hi(w[j], 0x9E3779B97F4A7C15ull ^ b) + 1; // 64x64 mulhi
            if (OP == 8) a[j] = (uint32_t)((double)a[j] * 0.999) + 3;            // cvt u32->f64, mul, cvt f64->u32
            if (OP == 9) a[j] = (uint32_t)((float)a[j] * 0.999f) + 3;            // cvt u32->f32, mul, cvt
            if (OP == 10) w[j] = w[j] * (0x9E3779B97F4A7C15ull ^ b) + 1;         // 64x64 mul lo
            if (OP == 11) w[j] = ((w[j] << 31) | (w[j] >> 33)) + 1;              // rotl64 (2 x v_alignbit) + add
            if (OP == 12) w[j] = (w[j] ^ (w[j] >> 33)) + 1;                      // xorshift 33 + add
            if (OP == 13) d[j] = __builtin_floor(d[j] * 1.0000001) + 0.5;        // v_mul_f64, v_floor_f64, v_add_f64
            if (OP == 14) d[j] = (double)(uint32_t)a[j] + d[j], a[j] += b;       // v_cvt_f64_u32 + v_add_f64 + v_add_u32
            if (OP == 15) w[j] = w[j] * 0xff51afd7ed558ccdULL + 1;               // 64x64 mul lo by a CONSTANT
            if (OP == 16) a[j] = a[j] >= b ? a[j] - b : a[j] + 7u;               // compare + select style fix-up
            if (OP == 17) a[j] = min(a[j] + 12345u, a[j] + 12345u - b);          // the unsigned-min fix-up
            if (OP == 18) d[j] = d[j] + d[(j + 1) & 7];                           // v_add_f64
            if (OP == 19) d[j] = d[j] * d[(j + 1) & 7];                           // v_mul_f64
        }
    }
    uint64_t acc = 0;
    for (int j = 0; j < 8; ++j) acc += a[j] + w[j] + (uint64_t)d[j];
    if (acc == 0x1234567) out[0] = acc;
}
template <int OP> int run(const char *name, uint64_t *d)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 3u, 5u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 3u, 5u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)grid * 256 * ITERS * 8;
    printf("%-28s %8.3f ms  %7.1f lane-ops/clk/CU (at 2.4 GHz, 256 CUs)\n", name, ms, ops / (ms * 1e-3) / 2.4e9 / 256);
    return 0;
}
int main()
{
    uint64_t *d; CHECK(hipMalloc(&d, 64));
    run<4>("v_add_u32", d); run<0>("v_mul_lo_u32+add", d); run<1>("v_mul_hi_u32+add", d); run<2>("v_mad_u64_u32", d);
    run<3>("mul_u24+add", d); run<5>("add_u64", d); run<6>("v_fma_f64", d); run<7>("umul64hi+add", d);
    run<10>("mul64lo+add", d); run<8>("cvt/mul/cvt f64", d); run<9>("cvt/mul/cvt f32", d);
    run<11>("rotl64+add", d); run<12>("xorshift33+add", d); run<13>("mul/floor/add f64", d); run<14>("cvt_f64_u32+add_f64+add", d);
    run<15>("mul64lo const+add", d); run<16>("cmp/select fixup", d); run<17>("min fixup", d); run<18>("v_add_f64", d); run<19>("v_mul_f64", d);
    return 0;
}
