/* hw_crc32c.c — TEST INFRASTRUCTURE: CRC-32C computed by the CPU's own SSE4.2 `crc32` instruction (Castagnoli polynomial in
 * silicon).  An implementation of the checksum that is neither the reference's (Go's hash/crc32), nor the oracle's table
 * walk, nor the library's GF(2) combine: oracle.py::hw_crc32c checks all three against it on arbitrary input.
 *   gcc -O2 -msse4.2 -shared -fPIC -o oracle/_third_party/libhwcrc32c.so oracle/hw_crc32c.c */
#include <nmmintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

__attribute__((visibility("default"))) uint32_t hw_crc32c(const uint8_t *p, size_t n)
{
    uint64_t c = 0xFFFFFFFFu;
    while (n >= 8) { uint64_t v; memcpy(&v, p, 8); c = _mm_crc32_u64(c, v); p += 8; n -= 8; }
    while (n--) c = _mm_crc32_u8((uint32_t)c, *p++);
    return (uint32_t)c ^ 0xFFFFFFFFu;
}
