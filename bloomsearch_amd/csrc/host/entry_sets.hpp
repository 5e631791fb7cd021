// entry_sets.hpp — bloomEntrySets (ingest.go:24-123): the distinct bloom entries of a set of
// rows — field paths, tokens, field::token pairs — collected at ingest so that filters can be
// built right-sized from exact distinct counts on the flush path (buildFilters, ingest.go:127-145,
// which here hands the packed sets to bsg_build on the GPU).
#pragma once
#include <cstdint>
#include <string>
#include <string_view>
#include <unordered_set>
#include <vector>

#include "text.hpp"
#include "walker.hpp"

namespace bsh {

struct BloomEntryCounts { uint64_t fields = 0, tokens = 0, field_tokens = 0; };  // file_format.go:753-757

class BloomEntrySets {
public:
    std::unordered_set<std::string> fields, tokens, field_tokens;

    // indexRow (ingest.go:55-89) with the BasicWhitespaceLowerTokenizer fast path.
    // Returns false when the row is not valid JSON (nothing is rolled back: entries seen
    // before the error stay, as they would for the reference's lenient parser).
    bool index_row(std::string_view row)
    {
        return walker_.walk(row, [&](const Emission &e) {
            path_scratch_.assign(e.path);
            fields.insert(path_scratch_);
            if (!e.is_leaf || !e.has_text) return true;  // null: field existence only (tokenizer.go:130-131)
            for_each_word(e.text, [&](std::string_view word) {
                token_buf_.clear();
                append_folded_word(token_buf_, word);
                tokens.insert(token_buf_);
                add_field_token(e.path, token_buf_);
                return true;
            });
            return true;
        }) || !walker_.malformed();
    }

    // addFieldToken (ingest.go:95-102): key = path + "::" + token, no escaping (tokenizer.go:509-511)
    void add_field_token(std::string_view path, std::string_view token)
    {
        key_buf_.assign(path);
        key_buf_.append("::");
        key_buf_.append(token);
        field_tokens.insert(key_buf_);
    }

    void union_into(BloomEntrySets &dst) const  // ingest.go:105-115
    {
        dst.fields.insert(fields.begin(), fields.end());
        dst.tokens.insert(tokens.begin(), tokens.end());
        dst.field_tokens.insert(field_tokens.begin(), field_tokens.end());
    }

    BloomEntryCounts counts() const { return {fields.size(), tokens.size(), field_tokens.size()}; }  // ingest.go:117-123

    const std::unordered_set<std::string> &set_of(uint32_t kind) const
    {
        return kind == 0 ? fields : (kind == 1 ? tokens : field_tokens);
    }

    void clear() { fields.clear(); tokens.clear(); field_tokens.clear(); }

private:
    PathWalker walker_;
    std::string path_scratch_, token_buf_, key_buf_;
};

// Appends one set as packed entries (bytes + u32 end offsets) in the layout bsg_build takes.
inline void pack_entries(const std::unordered_set<std::string> &set, std::vector<uint8_t> &bytes,
                         std::vector<uint32_t> &offsets)
{
    if (offsets.empty()) offsets.push_back(0);
    for (const auto &e : set) {
        bytes.insert(bytes.end(), e.begin(), e.end());
        offsets.push_back((uint32_t)bytes.size());
    }
}

}  // namespace bsh
