// host_api.cpp — C-ABI shims of include/bloomsearch_host.h over the header-only host mirror.
#include "bloomsearch_host.h"

#include <cstdlib>
#include <cstring>
#include <string>

#include "engine.hpp"

using namespace bsh;

struct bsh_entry_sets { BloomEntrySets sets; };
struct bsh_batch { QueryBatch batch; };
struct bse_engine {
    std::unique_ptr<BloomSearchEngine> eng;
    std::string err;
};

namespace {

int32_t give(const std::string &s, char **out, uint64_t *out_len)
{
    char *p = static_cast<char *>(malloc(s.size() + 1));
    if (!p) return BSH_E_INVALID;
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    *out = p;
    if (out_len) *out_len = s.size();
    return 0;
}

void json_escape(std::string &out, std::string_view s)
{
    static const char *hex = "0123456789abcdef";
    out.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        default:
            if (c < 0x20) { out += "\\u00"; out.push_back(hex[c >> 4]); out.push_back(hex[c & 15]); }
            else out.push_back((char)c);
        }
    }
    out.push_back('"');
}

bool parse_expression_json(const char *json, uint64_t len, BloomExpression &expr, bool &is_nil)
{
    is_nil = true;
    std::string_view sv(json ? json : "", json ? len : 0);
    size_t a = 0;
    while (a < sv.size() && (sv[a] == ' ' || sv[a] == '\n' || sv[a] == '\t' || sv[a] == '\r')) ++a;
    if (a == sv.size()) return true;
    JNode dom;
    if (!parse_dom(sv, dom)) return false;
    if (dom.type == JType::Null) return true;
    is_nil = false;
    return expression_from_json(dom, expr);
}

uint64_t num_or(const JNode &cfg, const char *key, uint64_t dflt)
{
    const JNode *n = cfg.get(key);
    if (!n || n->type != JType::Number) return dflt;
    return strtoull(n->text.c_str(), nullptr, 10);
}

}  // namespace

extern "C" {

void bsh_free(void *p) { free(p); }

int32_t bsh_tokenize(const uint8_t *text, uint64_t len, char **out, uint64_t *out_len)
{
    std::string joined, tok;
    for_each_word(std::string_view(reinterpret_cast<const char *>(text), len), [&](std::string_view w) {
        tok.clear();
        append_folded_word(tok, w);
        if (!joined.empty()) joined.push_back('\n');
        joined += tok;
        return true;
    });
    return give(joined, out, out_len);
}

bsh_entry_sets *bsh_entry_sets_new(void) { return new bsh_entry_sets(); }
void bsh_entry_sets_free(bsh_entry_sets *s) { delete s; }

int32_t bsh_entry_sets_index_row(bsh_entry_sets *s, const uint8_t *row, uint64_t len)
{
    if (!s) return BSH_E_INVALID;
    return s->sets.index_row(std::string_view(reinterpret_cast<const char *>(row), len)) ? 0 : BSH_E_INVALID;
}

int32_t bsh_entry_sets_union_into(const bsh_entry_sets *src, bsh_entry_sets *dst)
{
    if (!src || !dst) return BSH_E_INVALID;
    src->sets.union_into(dst->sets);
    return 0;
}

void bsh_entry_sets_counts(const bsh_entry_sets *s, uint64_t counts[3])
{
    const BloomEntryCounts c = s->sets.counts();
    counts[0] = c.fields; counts[1] = c.tokens; counts[2] = c.field_tokens;
}

int32_t bsh_entry_sets_export_sizes(const bsh_entry_sets *s, uint32_t kind, uint64_t *n_entries, uint64_t *n_bytes)
{
    if (!s || kind > 2) return BSH_E_INVALID;
    const auto &set = s->sets.set_of(kind);
    uint64_t b = 0;
    for (auto &e : set) b += e.size();
    *n_entries = set.size();
    *n_bytes = b;
    return 0;
}

int32_t bsh_entry_sets_export(const bsh_entry_sets *s, uint32_t kind, uint8_t *bytes, uint32_t *offsets)
{
    if (!s || kind > 2) return BSH_E_INVALID;
    uint32_t pos = 0, i = 0;
    offsets[0] = 0;
    for (auto &e : s->sets.set_of(kind)) {
        memcpy(bytes + pos, e.data(), e.size());
        pos += (uint32_t)e.size();
        offsets[++i] = pos;
    }
    return 0;
}

bsh_batch *bsh_batch_new(void) { return new bsh_batch(); }
void bsh_batch_free(bsh_batch *b) { delete b; }

int32_t bsh_batch_add_query(bsh_batch *b, const char *expr_json, uint64_t len)
{
    if (!b) return BSH_E_INVALID;
    BloomExpression e;
    bool nil = true;
    if (!parse_expression_json(expr_json, len, e, nil)) return BSH_E_INVALID;
    b->batch.add_query(nil ? nullptr : &e);
    return 0;
}

void bsh_batch_sizes(const bsh_batch *b, uint32_t *n_queries, uint32_t *n_terms, uint32_t *n_ops, uint64_t *term_bytes)
{
    uint64_t tb = 0;
    for (auto &s : b->batch.term_strings) tb += s.size();
    *n_queries = b->batch.n_queries();
    *n_terms = (uint32_t)b->batch.term_strings.size();
    *n_ops = (uint32_t)b->batch.prog_ops.size();
    *term_bytes = tb;
}

int32_t bsh_batch_export(const bsh_batch *b, uint8_t *term_bytes, uint32_t *term_offsets, uint32_t *term_kinds,
                         uint32_t *prog_ops, uint32_t *prog_off)
{
    if (!b) return BSH_E_INVALID;
    uint32_t pos = 0;
    term_offsets[0] = 0;
    for (size_t i = 0; i < b->batch.term_strings.size(); ++i) {
        const std::string &s = b->batch.term_strings[i];
        memcpy(term_bytes + pos, s.data(), s.size());
        pos += (uint32_t)s.size();
        term_offsets[i + 1] = pos;
        term_kinds[i] = b->batch.term_kinds[i];
    }
    if (!b->batch.prog_ops.empty()) memcpy(prog_ops, b->batch.prog_ops.data(), b->batch.prog_ops.size() * 4);
    memcpy(prog_off, b->batch.prog_off.data(), b->batch.prog_off.size() * 4);
    return 0;
}

int32_t bsh_match_row(const char *expr_json, uint64_t expr_len, const uint8_t *row, uint64_t row_len)
{
    BloomExpression e;
    bool nil = true;
    if (!parse_expression_json(expr_json, expr_len, e, nil)) return BSH_E_INVALID;
    RowMatcher m(nil ? nullptr : &e);
    return m.match(std::string_view(reinterpret_cast<const char *>(row), row_len)) ? 1 : 0;
}

namespace {
void expression_to_json(const BloomExpression &e, std::string &out)
{
    out += "{\"ExpressionType\":";
    out += e.type == ExprType::Condition ? "\"CONDITION\"" : e.type == ExprType::And ? "\"AND\"" : e.type == ExprType::Or ? "\"OR\"" : "\"?\"";
    if (e.type == ExprType::Condition) {
        out += ",\"Condition\":";
        if (!e.has_condition) out += "null";
        else {
            const char *t = e.condition.type == CondType::Field ? "FIELD" : e.condition.type == CondType::Token ? "TOKEN"
                          : e.condition.type == CondType::FieldToken ? "FIELD_TOKEN" : "?";
            out += std::string("{\"Type\":\"") + t + "\",\"Field\":";
            json_escape(out, e.condition.field);
            out += ",\"Token\":";
            json_escape(out, e.condition.token);
            out += "}";
        }
    } else {
        out += ",\"Children\":[";
        for (size_t i = 0; i < e.children.size(); ++i) { if (i) out.push_back(','); expression_to_json(e.children[i], out); }
        out += "]";
    }
    out += "}";
}
bool parse_regex_json(const char *json, uint64_t len, RegexExpression &e, bool &nil)
{
    nil = true;
    std::string_view sv(json ? json : "", json ? len : 0);
    if (sv.empty()) return true;
    JNode dom;
    if (!parse_dom(sv, dom)) return false;
    if (dom.type == JType::Null) return true;
    if (!regex_expression_from_json(dom, e)) return false;
    nil = false;
    return true;
}
}  // namespace

// pruneBloomQuery (query_exec.go:220): AndBloomQueries(bloom, RegexFieldGuardBloomQuery(regex)) as JSON ("null" = nil query)
int32_t bsh_prune_query(const char *bloom_json, uint64_t bloom_len, const char *regex_json, uint64_t regex_len, char **out, uint64_t *out_len)
{
    BloomExpression b, guard, pruned;
    RegexExpression r;
    bool bnil = true, rnil = true;
    if (!parse_expression_json(bloom_json, bloom_len, b, bnil) || !parse_regex_json(regex_json, regex_len, r, rnil)) return BSH_E_INVALID;
    const bool has_guard = regex_field_guard(rnil ? nullptr : &r, guard);
    std::string s = "null";
    if (and_bloom_queries(bnil ? nullptr : &b, has_guard ? &guard : nullptr, pruned)) { s.clear(); expression_to_json(pruned, s); }
    return give(s, out, out_len);
}

// the regex half of matchRowBytes (row_matcher.go:548-573): 1 match, 0 no match, < 0 invalid arguments / pattern
int32_t bsh_match_row_regex(const char *regex_json, uint64_t regex_len, const uint8_t *row, uint64_t row_len)
{
    RegexExpression r;
    bool nil = true;
    if (!parse_regex_json(regex_json, regex_len, r, nil)) return BSH_E_INVALID;
    RegexRowMatcher m(nil ? nullptr : &r);
    if (!m.valid()) return BSH_E_INVALID;
    return m.match(std::string_view(reinterpret_cast<const char *>(row), row_len)) ? 1 : 0;
}

int32_t bsh_section_encode(const uint64_t *const words[3], const uint64_t m[3], const uint64_t k[3], uint8_t **out, uint64_t *out_len)
{
    FilterView fv[3];
    for (int c = 0; c < 3; ++c) fv[c] = FilterView{words[c], m[c], k[c]};
    const std::vector<uint8_t> sec = encode_filter_section(fv);
    uint8_t *p = static_cast<uint8_t *>(malloc(sec.size()));
    if (!p) return BSH_E_INVALID;
    memcpy(p, sec.data(), sec.size());
    *out = p;
    *out_len = sec.size();
    return 0;
}

int32_t bsh_section_parse(const uint8_t *section, uint64_t len, uint64_t m[3], uint64_t k[3], uint64_t *words[3])
{
    ParsedFilter pf[3];
    const int32_t rc = parse_filter_section(section, len, pf);
    if (rc) return rc;
    for (int c = 0; c < 3; ++c) {
        m[c] = pf[c].present ? pf[c].m : 0;
        k[c] = pf[c].present ? pf[c].k : 0;
        words[c] = nullptr;
        if (pf[c].present) {
            words[c] = static_cast<uint64_t *>(malloc(std::max<size_t>(pf[c].words.size(), 1) * 8));
            memcpy(words[c], pf[c].words.data(), pf[c].words.size() * 8);
        }
    }
    return 0;
}

uint32_t bsh_crc32c(const uint8_t *data, uint64_t len) { return crc32c(data, len); }

// ---- engine ----
int32_t bse_open(const char *config_json, uint64_t len, bsg_ctx *ctx, bse_engine **out)
{
    if (!out || !ctx) return BSH_E_INVALID;
    *out = nullptr;
    EngineConfig cfg;
    std::string_view sv(config_json ? config_json : "", config_json ? len : 0);
    if (!sv.empty()) {
        JNode dom;
        if (!parse_dom(sv, dom) || dom.type != JType::Object) return BSE_E_INVALID_CONFIG;
        cfg.max_row_group_rows = num_or(dom, "MaxRowGroupRows", cfg.max_row_group_rows);
        cfg.max_row_group_bytes = num_or(dom, "MaxRowGroupBytes", cfg.max_row_group_bytes);
        cfg.max_buffered_rows = num_or(dom, "MaxBufferedRows", cfg.max_buffered_rows);
        cfg.max_buffered_bytes = num_or(dom, "MaxBufferedBytes", cfg.max_buffered_bytes);
        if (const JNode *n = dom.get("BloomFalsePositiveRate")) { if (n->type == JType::Number) cfg.bloom_false_positive_rate = strtod(n->text.c_str(), nullptr); }
        if (const JNode *n = dom.get("PartitionField")) { if (n->type == JType::String) cfg.partition_field = n->text; }
        if (const JNode *n = dom.get("DeviceIngest")) cfg.device_ingest = n->type == JType::True;
        if (const JNode *n = dom.get("DeviceMatch")) cfg.device_match = n->type == JType::True;
    }
    std::string err;
    if (int32_t rc = BloomSearchEngine::validate(cfg, err)) return rc;
    auto *e = new bse_engine();
    e->eng = std::make_unique<BloomSearchEngine>(cfg, ctx);
    *out = e;
    return 0;
}

void bse_close(bse_engine *e) { delete e; }
const char *bse_last_error(bse_engine *e) { return e ? e->eng->last_error().c_str() : ""; }
int32_t bse_stop(bse_engine *e) { if (!e) return BSH_E_INVALID; e->eng->stop(); return 0; }

int32_t bse_ingest_rows(bse_engine *e, const uint8_t *ndjson, uint64_t len)
{
    if (!e) return BSH_E_INVALID;
    std::vector<std::string_view> rows;
    std::string_view all(reinterpret_cast<const char *>(ndjson), len);
    size_t pos = 0;
    while (pos < all.size()) {
        size_t nl = all.find('\n', pos);
        if (nl == std::string_view::npos) nl = all.size();
        if (nl > pos) rows.push_back(all.substr(pos, nl - pos));
        pos = nl + 1;
    }
    return e->eng->ingest_rows(rows);
}

int32_t bse_flush(bse_engine *e) { return e ? e->eng->flush() : BSH_E_INVALID; }
int32_t bse_merge(bse_engine *e) { return e ? e->eng->merge() : BSH_E_INVALID; }

int32_t bse_query(bse_engine *e, const char *query_json, uint64_t len, char **out_json, uint64_t *out_len)
{
    if (!e || !out_json) return BSH_E_INVALID;
    BloomExpression expr;
    RegexExpression regex;
    bool nil = true, has_regex = false;
    std::string_view sv(query_json ? query_json : "", query_json ? len : 0);
    if (!sv.empty()) {
        JNode dom;
        if (!parse_dom(sv, dom)) return BSE_E_INVALID_QUERY;
        if (dom.type == JType::Object) {
            const JNode *bloom = dom.get("Bloom");
            if (bloom && bloom->type == JType::Object) {
                const JNode *ex = bloom->get("Expression");
                if (ex && ex->type == JType::Object) {
                    if (!expression_from_json(*ex, expr)) return BSE_E_INVALID_QUERY;
                    nil = false;
                }
            }
            const JNode *rx = dom.get("Regex");
            if (rx && rx->type == JType::Object) {
                const JNode *ex = rx->get("Expression");
                if (ex && ex->type == JType::Object) {
                    if (!regex_expression_from_json(*ex, regex)) return BSE_E_INVALID_QUERY;
                    has_regex = true;
                }
            }
        } else if (dom.type != JType::Null) {
            return BSE_E_INVALID_QUERY;
        }
    }
    QueryResult res;
    if (int32_t rc = e->eng->query(nil ? nullptr : &expr, res, has_regex ? &regex : nullptr)) return rc;
    std::string out = "{\"rows\":[";
    for (size_t i = 0; i < res.rows.size(); ++i) { if (i) out.push_back(','); out += res.rows[i]; }
    out += "],\"stats\":{\"BlockStats\":[";
    for (size_t i = 0; i < res.block_stats.size(); ++i) {
        const BlockStats &s = res.block_stats[i];
        if (i) out.push_back(',');
        out += "{\"FileID\":" + std::to_string(s.file_id) + ",\"BlockOffset\":" + std::to_string(s.block_offset) +
               ",\"RowsProcessed\":" + std::to_string(s.rows_processed) + ",\"BytesProcessed\":" + std::to_string(s.bytes_processed) +
               ",\"TotalRows\":" + std::to_string(s.total_rows) + ",\"TotalBytes\":" + std::to_string(s.total_bytes) +
               ",\"Duration\":" + std::to_string(s.duration_ns) +
               ",\"BloomFilterSkipped\":" + (s.bloom_filter_skipped ? "true" : "false") + "}";
    }
    out += "],\"Errors\":[";
    for (size_t i = 0; i < res.errors.size(); ++i) { if (i) out.push_back(','); json_escape(out, res.errors[i]); }
    out += "],\"FilesConsidered\":" + std::to_string(res.files_considered) + ",\"FilesBloomSkipped\":" + std::to_string(res.files_bloom_skipped) + "}}";
    return give(out, out_json, out_len);
}

int32_t bse_describe(bse_engine *e, char **out_json, uint64_t *out_len)
{
    if (!e || !out_json) return BSH_E_INVALID;
    auto counts = [](const BloomEntryCounts &c) {
        return "{\"Fields\":" + std::to_string(c.fields) + ",\"Tokens\":" + std::to_string(c.tokens) + ",\"FieldTokens\":" + std::to_string(c.field_tokens) + "}";
    };
    auto filters = [](const std::vector<uint8_t> &sec) {
        ParsedFilter pf[3];
        std::string s = "[";
        if (parse_filter_section(sec.data(), sec.size(), pf) == 0)
            for (int c = 0; c < 3; ++c) {
                if (c) s.push_back(',');
                s += pf[c].present ? "{\"m\":" + std::to_string(pf[c].m) + ",\"k\":" + std::to_string(pf[c].k) + "}" : "null";
            }
        return s + "]";
    };
    std::string out = "{\"files\":[";
    bool first_f = true;
    for (const DataFile &f : e->eng->files()) {
        if (!first_f) out.push_back(',');
        first_f = false;
        out += "{\"FileID\":" + std::to_string(f.file_id) + ",\"BloomEntryCounts\":" + counts(f.counts) +
               ",\"BloomFilterSize\":" + std::to_string(f.filter_section.size()) + ",\"filters\":" + filters(f.filter_section) + ",\"blocks\":[";
        for (size_t b = 0; b < f.blocks.size(); ++b) {
            const DataBlock &blk = f.blocks[b];
            if (b) out.push_back(',');
            out += "{\"PartitionID\":";
            json_escape(out, blk.partition_id);
            out += ",\"Rows\":" + std::to_string(blk.rows.size()) + ",\"BlockOffset\":" + std::to_string(blk.block_offset) +
                   ",\"BloomEntryCounts\":" + counts(blk.counts) + ",\"BloomFalsePositiveRate\":" + std::to_string(blk.fpr) +
                   ",\"BloomFilterSize\":" + std::to_string(blk.filter_section.size()) + ",\"filters\":" + filters(blk.filter_section) + "}";
        }
        out += "]}";
    }
    out += "]}";
    return give(out, out_json, out_len);
}

int32_t bse_corrupt_section_byte(bse_engine *e, uint32_t file_index, int32_t block_index, uint64_t byte_index)
{
    if (!e) return BSH_E_INVALID;
    return e->eng->corrupt_section_byte(file_index, block_index, byte_index) ? 0 : BSH_E_INVALID;
}

int32_t bse_section_bytes(bse_engine *e, uint32_t file_index, int32_t block_index, uint8_t **out, uint64_t *out_len)
{
    if (!e || !out || !out_len) return BSH_E_INVALID;
    const auto &files = e->eng->files();
    if (file_index >= files.size()) return BSH_E_INVALID;
    const std::vector<uint8_t> *sec = &files[file_index].filter_section;
    if (block_index >= 0) {
        if ((size_t)block_index >= files[file_index].blocks.size()) return BSH_E_INVALID;
        sec = &files[file_index].blocks[block_index].filter_section;
    }
    uint8_t *p = static_cast<uint8_t *>(malloc(std::max<size_t>(sec->size(), 1)));
    memcpy(p, sec->data(), sec->size());
    *out = p;
    *out_len = sec->size();
    return 0;
}

}  // extern "C"
