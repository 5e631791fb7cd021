// json.hpp — a small strict JSON scanner with the value model the reference gets from
// tidwall/gjson v1.18.0 (un-vendored): object members visited in document order
// (duplicate keys kept), strings/keys JSON-unescaped, numbers kept as their raw
// literal (gjson .Raw — never a float round trip; tokenizer.go:124-125).
// Two front ends share it: the streaming row walker (walker.hpp) and a DOM for
// query / config documents.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "text.hpp"

namespace bsh {

enum class JType : uint8_t { Null, False, True, Number, String, Object, Array };

struct JScanner {
    const char *p, *end;
    bool ok = true;

    JScanner(const char *b, size_t n) : p(b), end(b + n) {}

    void skip_ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool fail() { ok = false; return false; }

    static int hexval(char c)
    {
        if (c >= '0' && c <= '9') return c - '0';
        if (c >= 'a' && c <= 'f') return c - 'a' + 10;
        if (c >= 'A' && c <= 'F') return c - 'A' + 10;
        return -1;
    }
    bool read_hex4(uint32_t &out)
    {
        if (end - p < 4) return false;
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i) {
            const int h = hexval(p[i]);
            if (h < 0) return false;
            v = v * 16 + (uint32_t)h;
        }
        p += 4;
        out = v;
        return true;
    }

    // Parses a string literal at p (which points at the opening quote); appends the decoded text.
    bool parse_string(std::string &out)
    {
        if (p >= end || *p != '"') return fail();
        ++p;
        for (;;) {
            const char *s = p;
            while (p < end && *p != '"' && *p != '\\') ++p;
            out.append(s, p - s);
            if (p >= end) return fail();
            if (*p == '"') { ++p; return true; }
            ++p;  // backslash
            if (p >= end) return fail();
            const char c = *p++;
            switch (c) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                uint32_t r;
                if (!read_hex4(r)) return fail();
                if (r >= 0xD800 && r <= 0xDBFF && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                    const char *save = p;
                    p += 2;
                    uint32_t r2;
                    if (read_hex4(r2) && r2 >= 0xDC00 && r2 <= 0xDFFF) r = 0x10000 + ((r - 0xD800) << 10) + (r2 - 0xDC00);
                    else { p = save; r = kRuneError; }
                }
                append_rune(out, r);  // lone surrogates encode as U+FFFD, as utf8.EncodeRune does
                break;
            }
            default: return fail();
            }
        }
    }

    // Skips a string literal without decoding.
    bool skip_string()
    {
        if (p >= end || *p != '"') return fail();
        ++p;
        while (p < end) {
            if (*p == '\\') { p += 2; continue; }
            if (*p == '"') { ++p; return true; }
            ++p;
        }
        return fail();
    }

    // Number literal: returns [s, p) raw text.
    bool parse_number(std::string_view &raw)
    {
        const char *s = p;
        if (p < end && *p == '-') ++p;
        if (p >= end || *p < '0' || *p > '9') return fail();
        while (p < end && *p >= '0' && *p <= '9') ++p;
        if (p < end && *p == '.') { ++p; if (p >= end || *p < '0' || *p > '9') return fail(); while (p < end && *p >= '0' && *p <= '9') ++p; }
        if (p < end && (*p == 'e' || *p == 'E')) {
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || *p < '0' || *p > '9') return fail();
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        raw = std::string_view(s, p - s);
        return true;
    }

    bool parse_literal(const char *lit)
    {
        const size_t n = strlen(lit);
        if ((size_t)(end - p) < n || memcmp(p, lit, n) != 0) return fail();
        p += n;
        return true;
    }
};

// ---------------- DOM (queries, configs) ----------------
struct JNode {
    JType type = JType::Null;
    std::string text;  // String: decoded; Number: raw literal
    std::vector<std::pair<std::string, JNode>> members;  // Object
    std::vector<JNode> items;                            // Array

    const JNode *get(std::string_view key) const
    {
        for (auto &m : members) if (m.first == key) return &m.second;
        return nullptr;
    }
    bool is_null() const { return type == JType::Null; }
};

inline bool parse_dom_value(JScanner &sc, JNode &out, int depth = 0)
{
    if (depth > 512) return sc.fail();
    sc.skip_ws();
    if (sc.p >= sc.end) return sc.fail();
    switch (*sc.p) {
    case '{': {
        out.type = JType::Object;
        ++sc.p;
        sc.skip_ws();
        if (sc.p < sc.end && *sc.p == '}') { ++sc.p; return true; }
        for (;;) {
            sc.skip_ws();
            std::string key;
            if (!sc.parse_string(key)) return false;
            sc.skip_ws();
            if (sc.p >= sc.end || *sc.p != ':') return sc.fail();
            ++sc.p;
            out.members.emplace_back(std::move(key), JNode{});
            if (!parse_dom_value(sc, out.members.back().second, depth + 1)) return false;
            sc.skip_ws();
            if (sc.p < sc.end && *sc.p == ',') { ++sc.p; continue; }
            if (sc.p < sc.end && *sc.p == '}') { ++sc.p; return true; }
            return sc.fail();
        }
    }
    case '[': {
        out.type = JType::Array;
        ++sc.p;
        sc.skip_ws();
        if (sc.p < sc.end && *sc.p == ']') { ++sc.p; return true; }
        for (;;) {
            out.items.emplace_back();
            if (!parse_dom_value(sc, out.items.back(), depth + 1)) return false;
            sc.skip_ws();
            if (sc.p < sc.end && *sc.p == ',') { ++sc.p; continue; }
            if (sc.p < sc.end && *sc.p == ']') { ++sc.p; return true; }
            return sc.fail();
        }
    }
    case '"': out.type = JType::String; return sc.parse_string(out.text);
    case 't': out.type = JType::True; return sc.parse_literal("true");
    case 'f': out.type = JType::False; return sc.parse_literal("false");
    case 'n': out.type = JType::Null; return sc.parse_literal("null");
    default: {
        std::string_view raw;
        if (!sc.parse_number(raw)) return false;
        out.type = JType::Number;
        out.text.assign(raw);
        return true;
    }
    }
}

// Validation without a DOM (the ingest path checks every row of a batch before touching a buffer,
// ingest.go:378-397): accepts exactly what parse_dom accepts.  `scratch` is reused for decoded strings.
inline bool validate_value(JScanner &sc, std::string &scratch, int depth = 0)
{
    if (depth > 512) return sc.fail();
    sc.skip_ws();
    if (sc.p >= sc.end) return sc.fail();
    switch (*sc.p) {
    case '{':
        ++sc.p;
        sc.skip_ws();
        if (sc.p < sc.end && *sc.p == '}') { ++sc.p; return true; }
        for (;;) {
            sc.skip_ws();
            scratch.clear();
            if (!sc.parse_string(scratch)) return false;
            sc.skip_ws();
            if (sc.p >= sc.end || *sc.p != ':') return sc.fail();
            ++sc.p;
            if (!validate_value(sc, scratch, depth + 1)) return false;
            sc.skip_ws();
            if (sc.p < sc.end && *sc.p == ',') { ++sc.p; continue; }
            if (sc.p < sc.end && *sc.p == '}') { ++sc.p; return true; }
            return sc.fail();
        }
    case '[':
        ++sc.p;
        sc.skip_ws();
        if (sc.p < sc.end && *sc.p == ']') { ++sc.p; return true; }
        for (;;) {
            if (!validate_value(sc, scratch, depth + 1)) return false;
            sc.skip_ws();
            if (sc.p < sc.end && *sc.p == ',') { ++sc.p; continue; }
            if (sc.p < sc.end && *sc.p == ']') { ++sc.p; return true; }
            return sc.fail();
        }
    case '"': scratch.clear(); return sc.parse_string(scratch);
    case 't': return sc.parse_literal("true");
    case 'f': return sc.parse_literal("false");
    case 'n': return sc.parse_literal("null");
    default: { std::string_view raw; return sc.parse_number(raw); }
    }
}

// A row must be one JSON object and nothing else.  When `want_key` is non-empty, the text of the FIRST top-level
// member with that key is returned in `text` if it is a string, number (raw literal), true or false (JNode::get
// semantics of the DOM path it replaces); has_text tells whether there was one.
inline bool validate_object_row(std::string_view row, std::string_view want_key, std::string &text, bool &has_text, std::string &scratch)
{
    has_text = false;
    JScanner sc(row.data(), row.size());
    sc.skip_ws();
    if (sc.p >= sc.end || *sc.p != '{') return false;
    ++sc.p;
    sc.skip_ws();
    bool seen = false;
    if (sc.p < sc.end && *sc.p == '}') { ++sc.p; }
    else {
        for (;;) {
            sc.skip_ws();
            scratch.clear();
            if (!sc.parse_string(scratch)) return false;
            const bool mine = !seen && !want_key.empty() && scratch == want_key;
            sc.skip_ws();
            if (sc.p >= sc.end || *sc.p != ':') return false;
            ++sc.p;
            if (mine) {
                seen = true;
                sc.skip_ws();
                const char *v0 = sc.p;
                const char c = sc.p < sc.end ? *sc.p : 0;
                if (!validate_value(sc, scratch, 1)) return false;
                if (c == '"') { text = scratch; has_text = true; }
                else if (c == 't') { text = "true"; has_text = true; }
                else if (c == 'f') { text = "false"; has_text = true; }
                else if (c == '-' || (c >= '0' && c <= '9')) { text.assign(v0, sc.p - v0); has_text = true; }
            } else if (!validate_value(sc, scratch, 1)) {
                return false;
            }
            sc.skip_ws();
            if (sc.p < sc.end && *sc.p == ',') { ++sc.p; continue; }
            if (sc.p < sc.end && *sc.p == '}') { ++sc.p; break; }
            return false;
        }
    }
    sc.skip_ws();
    return sc.ok && sc.p == sc.end;
}

inline bool parse_dom(std::string_view json, JNode &out)
{
    JScanner sc(json.data(), json.size());
    if (!parse_dom_value(sc, out)) return false;
    sc.skip_ws();
    return sc.p == sc.end;
}

}  // namespace bsh
