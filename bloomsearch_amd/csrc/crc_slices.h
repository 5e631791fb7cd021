// CRC-32C arithmetic over GF(2) shared by the device kernels (kernels.hip.h: k_decode_sections, k_crc_sections), the host
// code that prepares their constants (bloomgpu.hip: crc_consts, stream_api.inc / encode_api.inc: init_image) and a plain g++
// test (tests/crc_slices_check.cpp): no HIP type in here.  The reference checksums a filter section with
// crc32.Checksum(payload, crc32.MakeTable(crc32.Castagnoli)) (file_format.go:343-384 writes it, :392-448 checks it).
#pragma once
#include <cstdint>

#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#define BSG_CRC_SLICES_PLAIN_CXX 1
#endif

namespace bsg {

constexpr uint32_t kCrc32cPoly = 0x82F63B78u;

__host__ __device__ inline uint32_t crc_multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ kCrc32cPoly : b >> 1;
    }
    return p;
}

// x^(n * 2^k) mod P
__host__ __device__ inline uint32_t crc_x2nmodp(uint64_t n, uint32_t k, const uint32_t *x2n)
{
    uint32_t p = 1u << 31;   // x^0
    while (n) {
        if (n & 1) p = crc_multmodp(x2n[k & 63], p);   // k < 64 for every n < 2^61 at the k = 3 this file starts from
        n >>= 1;
        ++k;
    }
    return p;
}

// what the 0xFFFFFFFF initial value and the final xor contribute to the CRC-32C of a payload of P bytes
__host__ __device__ inline uint32_t crc_init_image(uint64_t P, const uint32_t *x2n)
{
    return crc_multmodp(crc_x2nmodp(P, 3, x2n), 0xFFFFFFFFu) ^ 0xFFFFFFFFu;
}

constexpr uint32_t kDecodeMaxSplits = 32;      // slices of one section: their arrivals are 32 flag bits beside the section's 32-bit checksum accumulator (k_decode_sections)
#ifndef BSG_DECODE_UNIT
#define BSG_DECODE_UNIT 16384
#endif
__host__ __device__ inline uint32_t decode_unit(uint32_t P)
{
    const uint32_t u = (uint32_t)((((uint64_t)P + kDecodeMaxSplits - 1) / kDecodeMaxSplits + 63) / 64 * 64);
    return u > (uint32_t)BSG_DECODE_UNIT ? u : (uint32_t)BSG_DECODE_UNIT;
}
__host__ __device__ inline uint32_t decode_splits(uint32_t len)      // workgroups a section of `len` bytes (CRC trailer incl.) takes
{
    if (len < 5) return 1;
    const uint32_t P = len - 4, U = decode_unit(P);
    return P == 0 ? 1u : (P + U - 1) / U;
}

}  // namespace bsg

#ifdef BSG_CRC_SLICES_PLAIN_CXX
#undef __host__
#undef __device__
#undef BSG_CRC_SLICES_PLAIN_CXX
#endif
