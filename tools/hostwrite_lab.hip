// hostwrite_lab: what a kernel's stores into page-locked HOST memory cost (the survivor rows' destination), against the same stores into
// device memory and against a device-to-host copy of the same bytes.  ./hostwrite_lab [n_rows = 81920]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int W>   // W: 4-byte words a lane stores (contiguous per lane; lanes contiguous)
__global__ __launch_bounds__(256) void k_store(uint32_t *dst, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (W == 1) dst[i] = i * 2654435761u;
    if (W == 4) reinterpret_cast<uint4 *>(dst)[i] = make_uint4(i, i + 1, i + 2, i + 3);
    if (W == 32) reinterpret_cast<uint2 *>(dst)[(size_t)i * 16] = make_uint2(i, i + 1);      // 8 bytes per lane, lanes 128 bytes apart (a LIST row's ids in its slot)
}
template <int W> float run(uint32_t *dst, uint32_t n)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 8; ++it) {
        hipExtLaunchKernelGGL(k_store<W>, dim3((n + 255) / 256), dim3(256), 0, 0, e0, e1, 0, dst, n);
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
    }
    return best * 1e3f;
}
int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 81920;
    uint32_t *host, *dev;
    CHECK(hipHostMalloc(&host, (size_t)n * 16, hipHostMallocDefault));
    CHECK(hipMalloc(&dev, (size_t)n * 16));
    printf("%u lanes x 4 B  : device %.1f us, pinned host %.1f us (%.1f GB/s)\n", n, run<1>(dev, n), run<1>(host, n), n * 4.0 / (run<1>(host, n) * 1e-6) / 1e9);
    printf("%u lanes x 16 B : device %.1f us, pinned host %.1f us (%.1f GB/s)\n", n / 4, run<4>(dev, n / 4), run<4>(host, n / 4), n * 4.0 / (run<4>(host, n / 4) * 1e-6) / 1e9);
    printf("%u lanes x 16 B : device %.1f us, pinned host %.1f us (%.1f GB/s)\n", n, run<4>(dev, n), run<4>(host, n), n * 16.0 / (run<4>(host, n) * 1e-6) / 1e9);
    {
        uint32_t *host2, *dev2;
        const uint32_t n2 = 17240;                                       // the C2 batch's LIST rows per 20 arenas
        CHECK(hipHostMalloc(&host2, (size_t)n2 * 128, hipHostMallocDefault));
        CHECK(hipMalloc(&dev2, (size_t)n2 * 128));
        printf("%u lanes x 8 B, 128 B apart: device %.1f us, pinned host %.1f us (%.2f writes per ns)\n", n2, run<32>(dev2, n2), run<32>(host2, n2), n2 / (run<32>(host2, n2) * 1e3));
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (size_t bytes : {(size_t)n * 4, (size_t)n * 16}) {
        float best = 1e9;
        for (int it = 0; it < 6; ++it) {
            CHECK(hipEventRecord(e0)); CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost)); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
        }
        printf("hipMemcpyAsync D2H of %zu KB: %.1f us (%.1f GB/s)\n", bytes / 1024, best * 1e3, bytes / (best * 1e-3) / 1e9);
    }
    return 0;
}
