// text.hpp — UTF-8 decoding, Unicode white space / simple lower-casing and the
// whitespace+lowercase tokenizer, byte-for-byte as Go's standard library behaves
// for the reference's BasicWhitespaceLowerTokenizer = strings.Fields(strings.ToLower(v))
// (tokenizer.go:141-143) and its zero-alloc twin forEachWord + appendFoldedWord
// (row_matcher.go:142-202).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <string_view>

namespace bsh {

constexpr uint32_t kRuneError = 0xFFFD;

// utf8.DecodeRuneInString: returns the rune and its width; any invalid or
// truncated sequence (incl. surrogates, overlongs, > U+10FFFF) is (U+FFFD, 1).
inline uint32_t decode_rune(const unsigned char *p, size_t n, size_t &width)
{
    width = 1;
    const unsigned c0 = p[0];
    if (c0 < 0x80) return c0;
    if (c0 < 0xC2) return kRuneError;
    if (c0 < 0xE0) {
        if (n < 2 || (p[1] & 0xC0) != 0x80) return kRuneError;
        width = 2;
        return ((c0 & 0x1F) << 6) | (p[1] & 0x3F);
    }
    if (c0 < 0xF0) {
        if (n < 3) return kRuneError;
        const unsigned lo = c0 == 0xE0 ? 0xA0 : 0x80, hi = c0 == 0xED ? 0x9F : 0xBF;
        if (p[1] < lo || p[1] > hi || (p[2] & 0xC0) != 0x80) return kRuneError;
        width = 3;
        return ((c0 & 0x0F) << 12) | ((p[1] & 0x3F) << 6) | (p[2] & 0x3F);
    }
    if (c0 < 0xF5) {
        if (n < 4) return kRuneError;
        const unsigned lo = c0 == 0xF0 ? 0x90 : 0x80, hi = c0 == 0xF4 ? 0x8F : 0xBF;
        if (p[1] < lo || p[1] > hi || (p[2] & 0xC0) != 0x80 || (p[3] & 0xC0) != 0x80) return kRuneError;
        width = 4;
        return ((c0 & 0x07) << 18) | ((p[1] & 0x3F) << 12) | ((p[2] & 0x3F) << 6) | (p[3] & 0x3F);
    }
    return kRuneError;
}

// utf8.AppendRune (invalid runes and surrogates encode as U+FFFD)
inline void append_rune(std::string &dst, uint32_t r)
{
    if (r > 0x10FFFF || (r >= 0xD800 && r <= 0xDFFF)) r = kRuneError;
    if (r < 0x80) {
        dst.push_back((char)r);
    } else if (r < 0x800) {
        dst.push_back((char)(0xC0 | (r >> 6)));
        dst.push_back((char)(0x80 | (r & 0x3F)));
    } else if (r < 0x10000) {
        dst.push_back((char)(0xE0 | (r >> 12)));
        dst.push_back((char)(0x80 | ((r >> 6) & 0x3F)));
        dst.push_back((char)(0x80 | (r & 0x3F)));
    } else {
        dst.push_back((char)(0xF0 | (r >> 18)));
        dst.push_back((char)(0x80 | ((r >> 12) & 0x3F)));
        dst.push_back((char)(0x80 | ((r >> 6) & 0x3F)));
        dst.push_back((char)(0x80 | (r & 0x3F)));
    }
}

inline bool ascii_space(unsigned char c)  // row_matcher.go:179-181
{
    return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r';
}

// unicode.IsSpace
inline bool is_space(uint32_t r)
{
    if (r < 0x80) return ascii_space((unsigned char)r);
    return r == 0x85 || r == 0xA0 || r == 0x1680 || (r >= 0x2000 && r <= 0x200A) || r == 0x2028 || r == 0x2029 ||
           r == 0x202F || r == 0x205F || r == 0x3000;
}

struct LowerPair { uint32_t from, to; };
inline const LowerPair kLowerTable[] = {
#include "unicode_lower.inc"
};

// Code points this build's tables know nothing about (unassigned in the Unicode version they were generated from): a
// newer Unicode may have made them cased letters.  The device walker never folds them itself — it hands the row to the
// host, whose own unicode.ToLower then decides (a Go host: its toolchain's tables; this mirror: identity).
struct RuneRange { uint32_t from, to; };
inline const RuneRange kUnknownRanges[] = {
#include "unicode_unknown.inc"
};
inline bool is_unknown_rune(uint32_t r)
{
    size_t lo = 0, hi = sizeof(kUnknownRanges) / sizeof(kUnknownRanges[0]);
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (kUnknownRanges[mid].to < r) lo = mid + 1; else hi = mid;
    }
    return lo < sizeof(kUnknownRanges) / sizeof(kUnknownRanges[0]) && kUnknownRanges[lo].from <= r;
}

// unicode.ToLower: simple case mapping (one rune -> one rune)
inline uint32_t to_lower(uint32_t r)
{
    if (r < 0x80) return (r >= 'A' && r <= 'Z') ? r + 32 : r;
    size_t lo = 0, hi = sizeof(kLowerTable) / sizeof(kLowerTable[0]);
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (kLowerTable[mid].from < r) lo = mid + 1; else hi = mid;
    }
    if (lo < sizeof(kLowerTable) / sizeof(kLowerTable[0]) && kLowerTable[lo].from == r) return kLowerTable[lo].to;
    return r;
}

// forEachWord (row_matcher.go:142-177): fn(word) for each maximal run of non-space runes.
template <class F>
inline void for_each_word(std::string_view text, F &&fn)
{
    const unsigned char *p = reinterpret_cast<const unsigned char *>(text.data());
    const size_t n = text.size();
    size_t i = 0;
    while (i < n) {
        size_t w;
        const uint32_t r = decode_rune(p + i, n - i, w);
        if (is_space(r)) { i += w; continue; }
        const size_t start = i;
        while (i < n) {
            const uint32_t r2 = decode_rune(p + i, n - i, w);
            if (is_space(r2)) break;
            i += w;
        }
        if (!fn(text.substr(start, i - start))) return;
    }
}

// appendFoldedWord (row_matcher.go:187-202): dst += strings.ToLower(word)
inline void append_folded_word(std::string &dst, std::string_view word)
{
    const unsigned char *p = reinterpret_cast<const unsigned char *>(word.data());
    const size_t n = word.size();
    for (size_t i = 0; i < n;) {
        const unsigned char c = p[i];
        if (c < 0x80) {
            dst.push_back((char)((c >= 'A' && c <= 'Z') ? c + 32 : c));
            ++i;
            continue;
        }
        size_t w;
        const uint32_t r = decode_rune(p + i, n - i, w);
        i += w;
        append_rune(dst, to_lower(r));
    }
}

}  // namespace bsh
