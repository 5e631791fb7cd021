// walker.hpp — the row enumeration of the reference: a marshaled-JSON row as a stream of
// (path, value, isLeaf) emissions.  One streaming pass, path kept in a reusable buffer.
//
// Semantics follow pathWalker.walk / walkValue / emitKeyPrefixPaths (row_matcher.go:51-135),
// which must stay emission-for-emission identical to forEachPathValue (tokenizer.go:51-113):
//   - object member => child path parent + "." + key; BEFORE descending, every "."-split
//     prefix of the key is emitted as a non-leaf path (skipping empty paths);
//   - object / array with a non-empty path => one non-leaf emission;
//   - array elements reuse the array's own path;
//   - primitive with a non-empty path => leaf emission; root and "" paths are never emitted;
//   - duplicates are allowed (consumers are idempotent set inserts).
// leafTokenInput (tokenizer.go:120-133): String -> decoded text, Number -> RAW literal,
// true/false -> "true"/"false", null -> no text.
#pragma once
#include <string>
#include <string_view>

#include "json.hpp"

namespace bsh {

constexpr char kDelimiter = '.';  // hard-wired "." at both reference call sites (ingest.go:57, query_exec.go:218)

struct Emission {
    std::string_view path;
    JType type;             // Object/Array for containers and key-prefix paths use JType::Object
    bool is_leaf;
    std::string_view text;  // leafTokenInput text when has_text
    bool has_text;
};

class PathWalker {
public:
    // emit(const Emission&) -> bool (false stops the walk early).  Returns false if stopped
    // early or the row is not valid JSON (malformed() tells which).
    template <class F>
    bool walk(std::string_view row, F &&emit)
    {
        path_.clear();
        malformed_ = false;
        JScanner sc(row.data(), row.size());
        const bool cont = walk_value(sc, emit, 0);
        if (!sc.ok) malformed_ = true;
        return cont && sc.ok;
    }
    bool malformed() const { return malformed_; }

private:
    std::string path_, key_, str_;
    bool malformed_ = false;

    template <class F>
    bool emit_key_prefix_paths(std::string_view key, F &emit)
    {
        if (key.find(kDelimiter) == std::string_view::npos) return true;
        const size_t parent_len = path_.size();
        size_t split_at = 0;
        for (;;) {
            const size_t idx = key.find(kDelimiter, split_at);
            if (idx == std::string_view::npos) { path_.resize(parent_len); return true; }
            split_at = idx;
            path_.resize(parent_len);
            if (parent_len != 0) path_.push_back(kDelimiter);
            path_.append(key.substr(0, split_at));
            if (!path_.empty()) {
                if (!emit(Emission{path_, JType::Object, false, {}, false})) { path_.resize(parent_len); return false; }
            }
            split_at += 1;
        }
    }

    template <class F>
    bool walk_value(JScanner &sc, F &emit, int depth)
    {
        if (depth > 512) return sc.fail();
        sc.skip_ws();
        if (sc.p >= sc.end) return sc.fail();
        const char c = *sc.p;
        if (c == '{') {
            if (!path_.empty() && !emit(Emission{path_, JType::Object, false, {}, false})) return false;
            ++sc.p;
            sc.skip_ws();
            if (sc.p < sc.end && *sc.p == '}') { ++sc.p; return true; }
            for (;;) {
                sc.skip_ws();
                std::string key;  // per level: the recursion below reuses key_/str_ scratch
                if (!sc.parse_string(key)) return false;
                sc.skip_ws();
                if (sc.p >= sc.end || *sc.p != ':') return sc.fail();
                ++sc.p;
                if (!emit_key_prefix_paths(key, emit)) return false;
                const size_t prev = path_.size();
                if (prev != 0) path_.push_back(kDelimiter);
                path_.append(key);
                const bool cont = walk_value(sc, emit, depth + 1);
                path_.resize(prev);
                if (!cont) return false;
                sc.skip_ws();
                if (sc.p < sc.end && *sc.p == ',') { ++sc.p; continue; }
                if (sc.p < sc.end && *sc.p == '}') { ++sc.p; return true; }
                return sc.fail();
            }
        }
        if (c == '[') {
            if (!path_.empty() && !emit(Emission{path_, JType::Array, false, {}, false})) return false;
            ++sc.p;
            sc.skip_ws();
            if (sc.p < sc.end && *sc.p == ']') { ++sc.p; return true; }
            for (;;) {
                if (!walk_value(sc, emit, depth + 1)) return false;  // elements contribute under the array's own path
                sc.skip_ws();
                if (sc.p < sc.end && *sc.p == ',') { ++sc.p; continue; }
                if (sc.p < sc.end && *sc.p == ']') { ++sc.p; return true; }
                return sc.fail();
            }
        }
        // primitive
        Emission e{path_, JType::Null, true, {}, false};
        if (c == '"') {
            str_.clear();
            if (!sc.parse_string(str_)) return false;
            e.type = JType::String; e.text = str_; e.has_text = true;
        } else if (c == 't') {
            if (!sc.parse_literal("true")) return false;
            e.type = JType::True; e.text = "true"; e.has_text = true;
        } else if (c == 'f') {
            if (!sc.parse_literal("false")) return false;
            e.type = JType::False; e.text = "false"; e.has_text = true;
        } else if (c == 'n') {
            if (!sc.parse_literal("null")) return false;
            e.type = JType::Null;
        } else {
            std::string_view raw;
            if (!sc.parse_number(raw)) return false;
            e.type = JType::Number; e.text = raw; e.has_text = true;
        }
        if (path_.empty()) return true;
        e.path = path_;
        return emit(e);
    }
};

}  // namespace bsh
