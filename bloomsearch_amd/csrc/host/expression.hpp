// expression.hpp — the bloom query algebra of the reference and its two consumers:
//   * BloomExpression / BloomCondition with Field / Token / FieldToken / And / Or and the
//     same-type flattening of flattenExpressions (query.go:478-610); AndBloomQueries (:709-718);
//     JSON in the exact shape of the reference's exported structs.
//   * QueryBatch: lowering of a batch of trees to the C-ABI form (bloomgpu.h): distinct terms and
//     one postfix program per query, case by case after evaluateBloomExpression /
//     evaluateBloomCondition (query_exec.go:89-159).
//   * RowMatcher: the final exact test on a row (compiledRowMatcher semantics,
//     row_matcher.go:204-626, for bloom conditions): single walk, FieldToken compares
//     (path, token) PAIRS — not the joined key (row_matcher.go:296-301, :587).
#pragma once
#include <cstdint>
#include <map>
#include <regex>
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "bloomgpu.h"
#include "json.hpp"
#include "text.hpp"
#include "walker.hpp"

namespace bsh {

enum class CondType : uint8_t { Field, Token, FieldToken, Unknown };
enum class ExprType : uint8_t { Condition, And, Or, Unknown };

struct BloomCondition {
    CondType type = CondType::Unknown;
    std::string field, token;
};

struct BloomExpression {
    ExprType type = ExprType::Unknown;
    bool has_condition = false;   // Condition != nil
    BloomCondition condition;
    std::vector<BloomExpression> children;
};

inline BloomExpression Field(std::string field)
{
    BloomExpression e; e.type = ExprType::Condition; e.has_condition = true;
    e.condition.type = CondType::Field; e.condition.field = std::move(field);
    return e;
}
inline BloomExpression Token(std::string token)
{
    BloomExpression e; e.type = ExprType::Condition; e.has_condition = true;
    e.condition.type = CondType::Token; e.condition.token = std::move(token);
    return e;
}
inline BloomExpression FieldToken(std::string field, std::string token)
{
    BloomExpression e; e.type = ExprType::Condition; e.has_condition = true;
    e.condition.type = CondType::FieldToken; e.condition.field = std::move(field); e.condition.token = std::move(token);
    return e;
}
inline std::vector<BloomExpression> flatten_expressions(std::vector<BloomExpression> in, ExprType t)
{
    std::vector<BloomExpression> out;  // query.go:600-610
    for (auto &e : in) {
        if (e.type == t && !e.has_condition) for (auto &c : e.children) out.push_back(std::move(c));
        else out.push_back(std::move(e));
    }
    return out;
}
inline BloomExpression And(std::vector<BloomExpression> kids)
{
    BloomExpression e; e.type = ExprType::And; e.children = flatten_expressions(std::move(kids), ExprType::And);
    return e;
}
inline BloomExpression Or(std::vector<BloomExpression> kids)
{
    BloomExpression e; e.type = ExprType::Or; e.children = flatten_expressions(std::move(kids), ExprType::Or);
    return e;
}

// makeFieldTokenKey (tokenizer.go:509-511)
inline std::string make_field_token_key(std::string_view field, std::string_view token)
{
    std::string k(field);
    k.append("::");
    k.append(token);
    return k;
}

// ---- JSON (the reference's exported struct shape) ----
// {"ExpressionType":"AND|OR|CONDITION","Condition":{"Type":"FIELD|TOKEN|FIELD_TOKEN","Field":..,"Token":..},"Children":[..]}
inline bool expression_from_json(const JNode &n, BloomExpression &out)
{
    if (n.type != JType::Object) return false;
    const JNode *t = n.get("ExpressionType");
    const std::string ts = (t && t->type == JType::String) ? t->text : "";
    out.type = ts == "CONDITION" ? ExprType::Condition : ts == "AND" ? ExprType::And : ts == "OR" ? ExprType::Or : ExprType::Unknown;
    const JNode *c = n.get("Condition");
    out.has_condition = c && c->type == JType::Object;
    if (out.has_condition) {
        const JNode *ct = c->get("Type"), *f = c->get("Field"), *tk = c->get("Token");
        const std::string cts = (ct && ct->type == JType::String) ? ct->text : "";
        out.condition.type = cts == "FIELD" ? CondType::Field : cts == "TOKEN" ? CondType::Token
                           : cts == "FIELD_TOKEN" ? CondType::FieldToken : CondType::Unknown;
        out.condition.field = (f && f->type == JType::String) ? f->text : "";
        out.condition.token = (tk && tk->type == JType::String) ? tk->text : "";
    }
    const JNode *kids = n.get("Children");
    if (kids && kids->type == JType::Array) {
        out.children.resize(kids->items.size());
        for (size_t i = 0; i < kids->items.size(); ++i)
            if (kids->items[i].type == JType::Null) { out.children[i].type = ExprType::Condition; out.children[i].has_condition = false; }
            else if (!expression_from_json(kids->items[i], out.children[i])) return false;
    }
    return true;
}

// ---- regex query tree and its bloom field guard (query.go:520-538, :612-718) ----
// RegexExpression mirrors the reference's exported struct; RegexFieldGuardBloomQuery turns every regex condition into
// Field(cond.Field) — a row can only match FieldRegex(f, ..) if path f exists, so files / blocks whose field filter lacks f
// are pruned before any row is scanned (query_exec.go:220: pruneBloomQuery = AndBloomQueries(rowBloom, guard)).
enum class RegexType : uint8_t { Condition, And, Or, Unknown };
struct RegexExpression {
    RegexType type = RegexType::Unknown;
    bool has_condition = false;
    std::string field, pattern;
    std::vector<RegexExpression> children;
};
inline RegexExpression FieldRegex(std::string field, std::string pattern)
{
    RegexExpression e; e.type = RegexType::Condition; e.has_condition = true; e.field = std::move(field); e.pattern = std::move(pattern);
    return e;
}
inline bool regex_expression_from_json(const JNode &n, RegexExpression &out)
{
    if (n.type != JType::Object) return false;
    const JNode *t = n.get("ExpressionType");
    const std::string ts = (t && t->type == JType::String) ? t->text : "";
    out.type = ts == "CONDITION" ? RegexType::Condition : ts == "AND" ? RegexType::And : ts == "OR" ? RegexType::Or : RegexType::Unknown;
    const JNode *c = n.get("Condition");
    out.has_condition = c && c->type == JType::Object;
    if (out.has_condition) {
        const JNode *f = c->get("Field"), *pt = c->get("Pattern");
        out.field = (f && f->type == JType::String) ? f->text : "";
        out.pattern = (pt && pt->type == JType::String) ? pt->text : "";
    }
    const JNode *kids = n.get("Children");
    if (kids && kids->type == JType::Array) {
        out.children.resize(kids->items.size());
        for (size_t i = 0; i < kids->items.size(); ++i)
            if (!regex_expression_from_json(kids->items[i], out.children[i])) return false;
    }
    return true;
}
// regexExpressionToBloomFieldExpression (query.go:651-696): false = nil.  And / Or keep their node even when every child
// was dropped (the reference builds the node from whatever children survive; no flattening here).
inline bool regex_to_bloom_field_expression(const RegexExpression &e, BloomExpression &out)
{
    switch (e.type) {
    case RegexType::Condition:
        if (!e.has_condition) return false;
        out = Field(e.field);
        return true;
    case RegexType::And:
    case RegexType::Or: {
        out = BloomExpression{};
        out.type = e.type == RegexType::And ? ExprType::And : ExprType::Or;
        for (const RegexExpression &c : e.children) {
            BloomExpression child;
            if (regex_to_bloom_field_expression(c, child)) out.children.push_back(std::move(child));
        }
        return true;
    }
    default:
        return false;
    }
}
// True when EVERY node of the regex tree has a counterpart in the guard (no nil condition, no unknown node type).  Only then
// is the guard a necessary condition of the regex tree for a single ROW: a dropped child changes an Or's meaning
// (Or(<nil condition>, FieldRegex(f, ..)) is true for every row, its guard Or(Field(f)) is not).  The reference uses the
// guard for files and blocks only (query_exec.go:220); the engine mirror's per-row pre-selection is gated on this.
inline bool regex_guard_is_exact(const RegexExpression &e)
{
    switch (e.type) {
    case RegexType::Condition: return e.has_condition;
    case RegexType::And:
    case RegexType::Or:
        for (const RegexExpression &c : e.children) if (!regex_guard_is_exact(c)) return false;
        return true;
    default: return false;
    }
}
// RegexFieldGuardBloomQuery (query.go:698-707): nil query / nil expression / untranslatable => no guard
inline bool regex_field_guard(const RegexExpression *regex, BloomExpression &out)
{
    return regex && regex_to_bloom_field_expression(*regex, out);
}
// AndBloomQueries (query.go:709-718): a nil side yields the other; otherwise And(left, right) WITH flattening
inline bool and_bloom_queries(const BloomExpression *left, const BloomExpression *right, BloomExpression &out)
{
    if (!left && !right) return false;
    if (!left) { out = *right; return true; }
    if (!right) { out = *left; return true; }
    out = And({*left, *right});
    return true;
}

// The regex half of the final row test (row_matcher.go:548-573): a condition holds when its pattern matches the text of any
// primitive at or beneath the field path (decoded text for strings, the raw literal for numbers and booleans; null never).
// Patterns are compiled with std::regex (ECMAScript): the mirror covers the common subset it shares with Go's RE2 syntax.
// SUPPORTED SUBSET (stated, not discovered): literals, ., character classes incl. \\d \\w \\s and ranges, anchors ^ $, groups
// ( ) and (?: ), alternation, greedy and lazy quantifiers * + ? {n,m}, and ONE leading (?i).  Matching is per byte on UTF-8
// (RE2 matches runes: a multi-byte character inside a class or under '.' differs).  NOT supported — query() answers
// "regex pattern does not compile" where the reference accepts them: \\pL / \\p{..}, (?s) (?m) (?U) and inline flags other than
// a leading (?i), named groups (?P<n>..), \\Q..\\E, \\z, \\C.  A Go host keeps its own regexp; this mirror exists for the parity
// tests of the DEVICE path (the field guard and the bloom side of regex queries), which use patterns inside the subset.
class RegexRowMatcher {
public:
    explicit RegexRowMatcher(const RegexExpression *e)
    {
        if (!e) return;
        root_ = *e; has_root_ = true;
        collect(root_);
    }
    bool valid() const { return valid_; }
    bool match(std::string_view row)
    {
        if (!has_root_) return true;
        sat_.assign(conds_.size(), 0);
        walker_.walk(row, [&](const Emission &em) {
            if (!em.is_leaf || !em.has_text) return true;
            for (size_t i = 0; i < conds_.size(); ++i) {
                if (sat_[i]) continue;
                const std::string &f = conds_[i]->field;
                const bool under = em.path == f || (em.path.size() > f.size() && em.path.compare(0, f.size(), f) == 0 && em.path[f.size()] == kDelimiter);   // the walker's delimiter (walker.hpp), not a literal
                if (under && std::regex_search(em.text.begin(), em.text.end(), res_[i])) sat_[i] = 1;
            }
            return true;
        });
        size_t next = 0;
        return eval(root_, next);
    }

private:
    RegexExpression root_;
    bool has_root_ = false, valid_ = true;
    std::vector<const RegexExpression *> conds_;
    std::vector<std::regex> res_;
    std::vector<uint8_t> sat_;
    PathWalker walker_;
    void collect(const RegexExpression &e)
    {
        if (e.type == RegexType::Condition) {
            if (!e.has_condition || e.field.empty()) return;   // nil condition: true; empty field: constant false (row_matcher.go:446-451)
            conds_.push_back(&e);
            // the one RE2 idiom the mirror translates: a leading (?i) becomes the icase flag
            const bool icase = e.pattern.rfind("(?i)", 0) == 0;
            try { res_.emplace_back(icase ? e.pattern.substr(4) : e.pattern,
                                    std::regex::ECMAScript | std::regex::optimize | (icase ? std::regex::icase : std::regex::ECMAScript)); }
            catch (const std::regex_error &) { res_.emplace_back(); valid_ = false; }
            return;
        }
        for (auto &c : e.children) collect(c);
    }
    bool eval(const RegexExpression &e, size_t &next) const
    {
        switch (e.type) {
        case RegexType::Condition: return !e.has_condition ? true : (e.field.empty() ? false : (bool)sat_[next++]);
        case RegexType::And: { bool r = true; for (auto &c : e.children) r = eval(c, next) && r; return r; }
        case RegexType::Or: { bool r = false; for (auto &c : e.children) r = eval(c, next) || r; return !e.children.empty() && r; }
        default: return false;
        }
    }
};

// ---- lowering to the C-ABI ----
class QueryBatch {
public:
    std::vector<std::string> term_strings;
    std::vector<uint32_t> term_kinds;
    std::vector<uint32_t> prog_ops;
    std::vector<uint32_t> prog_off{0};

    // expression == nullptr: nil BloomQuery / nil Expression => zero ops => true (query_exec.go:81-83)
    void add_query(const BloomExpression *expression)
    {
        if (expression) emit(*expression);
        prog_off.push_back((uint32_t)prog_ops.size());
    }
    uint32_t n_queries() const { return (uint32_t)prog_off.size() - 1; }

private:
    std::map<std::pair<uint32_t, std::string>, uint32_t> index_;

    uint32_t term(uint32_t kind, std::string s)
    {
        auto key = std::make_pair(kind, std::move(s));
        auto it = index_.find(key);
        if (it != index_.end()) return it->second;
        const uint32_t i = (uint32_t)term_strings.size();
        term_strings.push_back(key.second);
        term_kinds.push_back(kind);
        index_.emplace(std::move(key), i);
        return i;
    }

    void emit(const BloomExpression &e)
    {
        switch (e.type) {
        case ExprType::Condition:
            if (!e.has_condition) { prog_ops.push_back(BSG_OP(BSG_OP_TRUE, 0)); return; }  // query_exec.go:101-104
            switch (e.condition.type) {
            case CondType::Field: prog_ops.push_back(BSG_OP(BSG_OP_TERM, term(BSG_KIND_FIELD, e.condition.field))); return;
            case CondType::Token: prog_ops.push_back(BSG_OP(BSG_OP_TERM, term(BSG_KIND_TOKEN, e.condition.token))); return;
            case CondType::FieldToken:
                prog_ops.push_back(BSG_OP(BSG_OP_TERM, term(BSG_KIND_FIELD_TOKEN, make_field_token_key(e.condition.field, e.condition.token))));
                return;
            default: prog_ops.push_back(BSG_OP(BSG_OP_FALSE, 0)); return;  // :155-156
            }
        case ExprType::And:
        case ExprType::Or:
            for (auto &c : e.children) emit(c);
            prog_ops.push_back(BSG_OP(e.type == ExprType::And ? BSG_OP_AND : BSG_OP_OR, (uint32_t)e.children.size()));
            return;
        default:
            prog_ops.push_back(BSG_OP(BSG_OP_FALSE, 0));  // :122-123
        }
    }
};

// ---- final exact row test ----
// The same tree lowered for the device row matcher (bloomgpu.h bsg_match_rows): conditions keep their field and
// token strings APART (FieldToken is a (path, token) pair at one leaf, row_matcher.go:587), the program refers to
// conditions by index.  nil condition => TRUE, unknown condition / expression => FALSE (row_matcher.go:257-290).
class MatcherProgram {
public:
    std::vector<uint32_t> kinds;
    std::vector<std::string> fields, tokens;
    std::vector<uint32_t> prog_ops;

    explicit MatcherProgram(const BloomExpression *expression) { if (expression) emit(*expression); }

private:
    void emit(const BloomExpression &e)
    {
        switch (e.type) {
        case ExprType::Condition:
            if (!e.has_condition) { prog_ops.push_back(BSG_OP(BSG_OP_TRUE, 0)); return; }
            if (e.condition.type == CondType::Unknown) { prog_ops.push_back(BSG_OP(BSG_OP_FALSE, 0)); return; }
            prog_ops.push_back(BSG_OP(BSG_OP_TERM, (uint32_t)kinds.size()));
            kinds.push_back(e.condition.type == CondType::Field ? BSG_KIND_FIELD
                                                                 : (e.condition.type == CondType::Token ? BSG_KIND_TOKEN : BSG_KIND_FIELD_TOKEN));
            fields.push_back(e.condition.field);
            tokens.push_back(e.condition.token);
            return;
        case ExprType::And:
        case ExprType::Or:
            for (auto &c : e.children) emit(c);
            prog_ops.push_back(BSG_OP(e.type == ExprType::And ? BSG_OP_AND : BSG_OP_OR, (uint32_t)e.children.size()));
            return;
        default:
            prog_ops.push_back(BSG_OP(BSG_OP_FALSE, 0));
        }
    }
};

class RowMatcher {
public:
    explicit RowMatcher(const BloomExpression *expression)
    {
        if (expression) { root_ = *expression; has_root_ = true; collect(root_); }
    }

    // matchRowBytes (row_matcher.go:486-573, bloom conditions): one walk, monotone satisfaction flags,
    // tree evaluated after the walk (the reference's early exit changes cost, not verdicts).
    bool match(std::string_view row)
    {
        if (!has_root_) return true;
        sat_.assign(conds_.size(), 0);
        walker_.walk(row, [&](const Emission &e) {
            for (size_t i = 0; i < conds_.size(); ++i)
                if (!sat_[i] && conds_[i]->type == CondType::Field && e.path == conds_[i]->field) sat_[i] = 1;
            if (!e.is_leaf || !e.has_text || !wants_tokens_) return true;
            for_each_word(e.text, [&](std::string_view word) {
                fold_.clear();
                append_folded_word(fold_, word);
                for (size_t i = 0; i < conds_.size(); ++i) {
                    if (sat_[i]) continue;
                    const BloomCondition &c = *conds_[i];
                    if (c.type == CondType::Token) { if (fold_ == c.token) sat_[i] = 1; }               // targets never normalised
                    else if (c.type == CondType::FieldToken) { if (e.path == c.field && fold_ == c.token) sat_[i] = 1; }
                }
                return true;
            });
            return true;
        });
        size_t next = 0;
        return eval(root_, next);
    }

private:
    BloomExpression root_;
    bool has_root_ = false, wants_tokens_ = false;
    std::vector<const BloomCondition *> conds_;  // pre-order, matching eval()'s traversal
    std::vector<uint8_t> sat_;
    PathWalker walker_;
    std::string fold_;

    void collect(const BloomExpression &e)
    {
        if (e.type == ExprType::Condition) {
            if (e.has_condition && e.condition.type != CondType::Unknown) {
                conds_.push_back(&e.condition);
                if (e.condition.type != CondType::Field) wants_tokens_ = true;
            }
            return;
        }
        if (e.type == ExprType::And || e.type == ExprType::Or) for (auto &c : e.children) collect(c);
    }

    // evalMatcherNode (row_matcher.go:257-290): nil condition => true, empty Or => false,
    // empty And => true, unknown expression/condition => false.
    bool eval(const BloomExpression &e, size_t &next) const
    {
        switch (e.type) {
        case ExprType::Condition:
            if (!e.has_condition) return true;
            if (e.condition.type == CondType::Unknown) return false;
            return sat_[next++] != 0;
        case ExprType::And: { bool r = true; for (auto &c : e.children) r = eval(c, next) && r; return r; }
        case ExprType::Or: { bool r = false; for (auto &c : e.children) r = eval(c, next) || r; return r; }
        default: return false;
        }
    }
};

}  // namespace bsh
