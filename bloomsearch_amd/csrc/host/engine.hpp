// engine.hpp — host-side mirror of the reference's BloomSearchEngine surface for THIS path only:
//   IngestRows  (ingest.go:170, processIngestRequest :330-531: partitioning, whole-batch validation,
//                indexRow into per-partition bloomEntrySets, flush triggers)
//   Flush       (ingest.go:197; handleFlush flush.go:138-282: per partition buffer buildFilters ->
//                encodeFilterSection; unionInto(fileEntries); file-level buildFilters; counts stamped)
//   Query       (query_exec.go:201-444: file stage over file-level filters, evaluateBlockFilters per
//                candidate block with BloomFilterSkipped stats, final matchRowBytes scan)
//   Merge       (merge.go:440-817 rebuild semantics: re-index every row, right-size, never OR)
// The bloom arithmetic AND both directions of the filter-section codec run on the GPU through the C-ABI
// (bsg_build_sections / bsg_ingest_* / bsg_arena_load_sections / bsg_probe);
// there is no CPU bloom path here.  Storage, compression, durability, goroutines and the
// DataStore / MetaStore plugins are out of scope: "files" are kept in memory with the reference's
// filter-section bytes (section_codec.hpp) so the wire layout is exercised end to end.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <map>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#include "bloomgpu.h"
#include "entry_sets.hpp"
#include "expression.hpp"
#include "section_codec.hpp"

namespace bsh {

enum EngineError : int32_t {
    kEngineOk = 0,
    kErrInvalidConfig = -101,   // ErrInvalidConfig (engine.go:16-34)
    kErrEngineStopped = -102,   // ErrEngineStopped
    kErrInvalidRow = -103,      // a row of the batch is not a JSON object: whole batch rejected (ingest.go:378-397)
    kErrInvalidQuery = -104,
    kErrGpu = -105,             // a bsg_* call failed (message carries bsg_last_error)
    kErrInvalidHash = -106,     // ErrInvalidHash: filter section CRC mismatch
};

struct EngineConfig {           // BloomSearchEngineConfig (engine.go:82-147), the knobs on this path
    uint64_t max_row_group_rows = 10000;
    uint64_t max_row_group_bytes = 10ull * 1024 * 1024;
    uint64_t max_buffered_rows = 1000;
    uint64_t max_buffered_bytes = 1ull * 1024 * 1024;
    double bloom_false_positive_rate = 0.001;
    std::string partition_field;  // PartitionFunc stand-in: top-level key whose text is the partition id ("" = none)
    // true: rows are only validated and buffered at ingest; walking, tokenizing, dedup, counting and the filter
    // build all run on the device at flush / merge time (bsg_ingest_*), the host walker only finishing the rows
    // the device walker hands back.  false: indexRow on the host at ingest time, as the reference does.
    bool device_ingest = false;
    // true: the final row test of the surviving blocks (matchRowBytes, query_exec.go:751) runs on the device too
    // (bsg_match_rows); rows it hands back, and expressions beyond its limits, go through the host matcher.
    bool device_match = false;
};

struct DataBlock {
    std::string partition_id;
    std::vector<std::string> rows;
    uint64_t row_bytes = 0;              // uncompressed row bytes incl. the 4-byte length prefixes
    uint64_t block_offset = 0;           // RowDataOffset within the (virtual) file
    BloomEntryCounts counts;             // DataBlockMetadata.BloomEntryCounts (file_format.go:753-757)
    double fpr = 0;
    std::vector<uint8_t> filter_section; // encodeFilterSection bytes
};

struct DataFile {
    uint64_t file_id = 0;
    std::vector<DataBlock> blocks;
    BloomEntryCounts counts;
    std::vector<uint8_t> filter_section; // file-level filters
    // The file's block filters as a resident probe arena live in the LIBRARY's file-arena cache (bsg_file_arena_*, keyed by
    // file_id): published by the flush / merge that built them on the device, or decoded from the stored sections by the first
    // query that misses; least recently used files leave when the byte budget is exceeded, a merged-away file is forgotten.
};

struct BlockStats {                      // query_exec.go:63-72
    uint64_t file_id = 0, block_offset = 0;
    int64_t rows_processed = 0, bytes_processed = 0, total_rows = 0, total_bytes = 0;
    // query_exec.go:578,598-600: the reference timestamps every block; a batched probe has one wall time for all of them,
    // so every candidate block gets batch time / blocks (plus its share of the scan), never zero
    // (query_handles_test.go:1062 asserts Duration > 0)
    int64_t duration_ns = 0;
    bool bloom_filter_skipped = false;
};

struct QueryResult {
    std::vector<std::string> rows;
    std::vector<BlockStats> block_stats;
    std::vector<std::string> errors;     // per-block failures joined into Results.Err by the reference
    uint64_t files_considered = 0, files_bloom_skipped = 0;
};

class BloomSearchEngine {
public:
    BloomSearchEngine(const EngineConfig &cfg, bsg_ctx *ctx) : cfg_(cfg), ctx_(ctx) {}
    ~BloomSearchEngine() { drop_arenas(); }

    static int32_t validate(const EngineConfig &c, std::string &err)
    {
        if (c.max_row_group_rows == 0 || c.max_row_group_bytes == 0) { err = "MaxRowGroupRows / MaxRowGroupBytes must be positive"; return kErrInvalidConfig; }
        if (!(c.bloom_false_positive_rate > 0.0 && c.bloom_false_positive_rate < 1.0)) { err = "BloomFalsePositiveRate must be in (0, 1)"; return kErrInvalidConfig; }
        return kEngineOk;
    }

    const std::string &last_error() const { return err_; }
    const std::vector<DataFile> &files() const { return files_; }

    void stop() { stopped_ = true; }

    // Fault injection (the reference's tests use corrupting DataStore doubles): flip one byte of a stored section.
    bool corrupt_section_byte(size_t file_index, int block_index, size_t byte_index)
    {
        if (file_index >= files_.size()) return false;
        std::vector<uint8_t> *sec = &files_[file_index].filter_section;
        if (block_index >= 0) {
            if ((size_t)block_index >= files_[file_index].blocks.size()) return false;
            sec = &files_[file_index].blocks[block_index].filter_section;
        }
        if (byte_index >= sec->size()) return false;
        (*sec)[byte_index] ^= 0x5A;
        if (block_index >= 0) forget_file_arena(files_[file_index]);   // re-read (and re-checked) from the stored bytes on next use
        else drop_files_arena();
        return true;
    }

    // rows: one marshaled JSON object per element.  Whole batch is validated before any buffer
    // is touched (ingest.go:378-397).
    int32_t ingest_rows(const std::vector<std::string_view> &rows)
    {
        if (stopped_) return fail(kErrEngineStopped, "engine stopped");
        if (rows.empty()) return kEngineOk;  // ingest.go:356-359
        std::vector<std::string> pids(rows.size());
        std::string scratch;
        for (size_t i = 0; i < rows.size(); ++i) {
            bool has_pid = false;
            if (!validate_object_row(rows[i], cfg_.partition_field, pids[i], has_pid, scratch))
                return fail(kErrInvalidRow, "row " + std::to_string(i) + " is not a JSON object");
            if (!has_pid) pids[i].clear();
        }
        bool should_flush = false;
        for (size_t i = 0; i < rows.size(); ++i) {
            PartitionBuffer &pb = buffers_[pids[i]];
            if (!cfg_.device_ingest) pb.entries.index_row(rows[i]);   // HOT: walk + tokenize + dedup (ingest.go:450)
            pb.rows.emplace_back(rows[i]);
            pb.bytes += rows[i].size() + 4;
            buffered_rows_ += 1;
            buffered_bytes_ += rows[i].size() + 4;
            if (pb.rows.size() >= cfg_.max_row_group_rows || pb.bytes >= cfg_.max_row_group_bytes) should_flush = true;
        }
        if (buffered_rows_ >= cfg_.max_buffered_rows || buffered_bytes_ >= cfg_.max_buffered_bytes) should_flush = true;
        return should_flush ? flush() : kEngineOk;
    }

    // handleFlush: one file, one data block per partition buffer, all filters built in ONE bsg_build call.
    int32_t flush()
    {
        if (buffers_.empty()) return kEngineOk;
        DataFile file;
        file.file_id = next_file_id_++;
        BloomEntrySets file_entries;
        std::vector<const BloomEntrySets *> sets;
        for (auto &kv : buffers_) {
            DataBlock blk;
            blk.partition_id = kv.first;
            blk.rows = std::move(kv.second.rows);
            blk.row_bytes = kv.second.bytes;
            blk.fpr = cfg_.bloom_false_positive_rate;
            if (!cfg_.device_ingest) {
                blk.counts = kv.second.entries.counts();
                kv.second.entries.union_into(file_entries);   // flush.go:221
            }
            file.blocks.push_back(std::move(blk));
        }
        std::vector<std::vector<uint8_t>> sections;
        if (cfg_.device_ingest) {
            if (int32_t rc = build_sections_device(file, sections)) return rc;
        } else {
            for (auto &kv : buffers_) sets.push_back(&kv.second.entries);
            sets.push_back(&file_entries);                    // file-level filters sized for the union (flush.go:253)
            if (int32_t rc = build_sections(sets, sections)) return rc;
            file.counts = file_entries.counts();
        }
        uint64_t off = 0;
        for (size_t b = 0; b < file.blocks.size(); ++b) {
            file.blocks[b].filter_section = std::move(sections[b]);
            file.blocks[b].block_offset = off;
            off += file.blocks[b].row_bytes;
        }
        file.filter_section = std::move(sections.back());
        files_.push_back(std::move(file));
        buffers_.clear();
        buffered_rows_ = buffered_bytes_ = 0;
        drop_files_arena();          // one more file-level "block"; the other files' block arenas stay where they are
        return kEngineOk;
    }

    // Merge (merge.go rebuild semantics): every source row is re-walked into fresh block + file
    // entry sets and filters are rebuilt right-sized; blocks of one partition are merged while they
    // fit MaxRowGroupRows / MaxRowGroupBytes.
    int32_t merge()
    {
        if (files_.size() < 2) return kEngineOk;
        std::map<std::string, std::vector<DataBlock *>> by_partition;
        for (auto &f : files_) for (auto &b : f.blocks) by_partition[b.partition_id].push_back(&b);
        DataFile out;
        out.file_id = next_file_id_++;
        std::vector<std::unique_ptr<BloomEntrySets>> block_sets;
        BloomEntrySets file_entries;
        for (auto &kv : by_partition) {
            DataBlock cur;
            auto start_block = [&]() { cur = DataBlock{}; cur.partition_id = kv.first; cur.fpr = cfg_.bloom_false_positive_rate; block_sets.push_back(std::make_unique<BloomEntrySets>()); };
            auto finish_block = [&]() {
                if (!cfg_.device_ingest) {
                    cur.counts = block_sets.back()->counts();
                    block_sets.back()->union_into(file_entries);
                }
                out.blocks.push_back(std::move(cur));
            };
            start_block();
            for (DataBlock *src : kv.second) {
                if (!cur.rows.empty() && (cur.rows.size() + src->rows.size() > cfg_.max_row_group_rows ||
                                          cur.row_bytes + src->row_bytes > cfg_.max_row_group_bytes)) { finish_block(); start_block(); }
                for (auto &r : src->rows) {
                    if (!cfg_.device_ingest) block_sets.back()->index_row(r);      // merge.go:746
                    cur.row_bytes += r.size() + 4;
                    cur.rows.push_back(std::move(r));
                }
            }
            finish_block();
        }
        std::vector<std::vector<uint8_t>> sections;
        if (cfg_.device_ingest) {
            if (int32_t rc = build_sections_device(out, sections)) return rc;
        } else {
            std::vector<const BloomEntrySets *> sets;
            for (auto &s : block_sets) sets.push_back(s.get());
            sets.push_back(&file_entries);
            if (int32_t rc = build_sections(sets, sections)) return rc;
            out.counts = file_entries.counts();
        }
        uint64_t off = 0;
        for (size_t b = 0; b < out.blocks.size(); ++b) {
            out.blocks[b].filter_section = std::move(sections[b]);
            out.blocks[b].block_offset = off;
            off += out.blocks[b].row_bytes;
        }
        out.filter_section = std::move(sections.back());
        for (auto &f : files_) forget_file_arena(f);    // tombstoned sources (merge.go:178-185)
        files_.clear();
        files_.push_back(std::move(out));
        drop_files_arena();
        return kEngineOk;
    }

    // Query: nil expression => no bloom conditions => no filter reads, every block scanned (query_exec.go:503-508).
    // regex: the field-scoped regex tree of the query; files and blocks are pruned by
    // pruneBloomQuery = AndBloomQueries(bloom, RegexFieldGuardBloomQuery(regex)) (query_exec.go:220), rows are matched by the
    // bloom tree AND the regex tree (row_matcher.go:353-368).
    int32_t query(const BloomExpression *row_expr, QueryResult &out, const RegexExpression *regex = nullptr)
    {
        out = QueryResult{};
        const auto t_begin = std::chrono::steady_clock::now();
        BloomExpression guard, prune_storage;
        const bool has_guard = regex_field_guard(regex, guard);
        const BloomExpression *expr = and_bloom_queries(row_expr, has_guard ? &guard : nullptr, prune_storage) ? &prune_storage : nullptr;
        std::vector<uint8_t> file_ok(files_.size(), 1);
        std::vector<std::vector<uint8_t>> block_ok(files_.size());
        for (size_t f = 0; f < files_.size(); ++f) block_ok[f].assign(files_[f].blocks.size(), 1);
        if (expr && !files_.empty()) {
            QueryBatch qb;
            qb.add_query(expr);
            std::vector<bsg_term> terms;
            if (int32_t rc = hash_terms(qb, terms)) return rc;
            uint64_t batch = 0;
            if (bsg_batch_create(ctx_, terms.data(), (uint32_t)terms.size(), qb.prog_ops.data(), qb.prog_off.data(), 1, &batch))
                return fail(kErrGpu, bsg_last_error(ctx_));
            struct FreeBatch { bsg_ctx *c; uint64_t id; ~FreeBatch() { bsg_batch_free(c, id); } } guard{ctx_, batch};
            // file stage (query_exec.go:399-406): one probe over the file-level filters, one "block" per file
            if (int32_t rc = ensure_files_arena()) return rc;
            std::vector<uint64_t> fs((files_.size() + 63) / 64);
            if (bsg_probe_batch(ctx_, files_arena_, batch, 0, fs.data())) return fail(kErrGpu, bsg_last_error(ctx_));
            // block stage: only the files that passed, each through its arena — resident in the library's cache, or decoded now
            // and published there — all in one pipelined call; the leases end when the survivors are on the host
            std::vector<uint64_t> ids;
            std::vector<size_t> which;
            std::vector<FileLease> leases;
            struct ReleaseAll { bsg_ctx *c; std::vector<FileLease> &v; ~ReleaseAll() { for (auto &l : v) bsg_file_arena_release(c, l.lease); } } release{ctx_, leases};
            size_t words = 0;
            for (size_t f = 0; f < files_.size(); ++f) {
                file_ok[f] = (fs[f >> 6] >> (f & 63)) & 1;   // a corrupt file-level section decodes to nil filters: cannot disqualify
                if (!file_ok[f] || files_[f].blocks.empty()) continue;
                leases.emplace_back();
                if (int32_t rc = lease_file_arena(files_[f], leases.back())) { leases.pop_back(); return rc; }
                ids.push_back(leases.back().arena);
                which.push_back(f);
                words += (files_[f].blocks.size() + 63) / 64;
            }
            std::vector<uint64_t> bs(std::max<size_t>(words, 1));
            if (!ids.empty() && bsg_probe_many(ctx_, ids.data(), (uint32_t)ids.size(), batch, 0, bs.data()))
                return fail(kErrGpu, bsg_last_error(ctx_));
            size_t o = 0;
            for (size_t i = 0; i < which.size(); ++i) {
                const size_t f = which[i];
                for (size_t b = 0; b < files_[f].blocks.size(); ++b) {
                    const uint32_t r = leases[i].rows[b];                      // block b's row in the arena
                    block_ok[f][b] = (bs[o + (r >> 6)] >> (r & 63)) & 1;
                    if (leases[i].status[b] != 0) block_ok[f][b] = 2;   // unreadable filters: neither pruned nor scanned (query_exec.go:580-590)
                }
                o += (files_[f].blocks.size() + 63) / 64;
            }
        }
        const int64_t probe_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
        size_t candidate_blocks = 0;
        for (size_t f = 0; f < files_.size(); ++f) if (file_ok[f]) candidate_blocks += files_[f].blocks.size();
        const int64_t probe_share = candidate_blocks ? std::max<int64_t>(1, probe_ns / (int64_t)candidate_blocks) : 0;
        RowMatcher matcher(row_expr);
        RegexRowMatcher regex_matcher(regex);
        if (!regex_matcher.valid()) return fail(kErrInvalidQuery, "regex pattern does not compile");
        // the scan list: every row of every block that survived both stages, in file / block order
        std::vector<const std::string *> scan;
        std::vector<size_t> scanned_stats;          // block_stats entries of the scanned blocks (for the scan's time share)
        const auto t_scan = std::chrono::steady_clock::now();
        for (size_t f = 0; f < files_.size(); ++f) {
            out.files_considered++;
            if (!file_ok[f]) { out.files_bloom_skipped++; continue; }   // file stage prune: no BlockStats for its blocks
            for (size_t b = 0; b < files_[f].blocks.size(); ++b) {
                const DataBlock &blk = files_[f].blocks[b];
                BlockStats st;
                st.file_id = files_[f].file_id; st.block_offset = blk.block_offset;
                st.total_rows = (int64_t)blk.rows.size();
                st.total_bytes = (int64_t)(blk.row_bytes + blk.filter_section.size());
                st.duration_ns = probe_share;
                if (block_ok[f][b] == 2) {        // recordUnreadBlocks (query_exec.go:625-639): totals only, error surfaced
                    out.errors.push_back("failed to read data block bloom filters: file " + std::to_string(files_[f].file_id) +
                                         " block offset " + std::to_string(blk.block_offset) + ": invalid hash");
                    out.block_stats.push_back(st);
                    continue;
                }
                if (!block_ok[f][b]) {                                    // query_exec.go:607-614
                    st.bloom_filter_skipped = true;
                    out.block_stats.push_back(st);
                    continue;
                }
                for (const std::string &row : blk.rows) {
                    st.rows_processed++;
                    st.bytes_processed += (int64_t)row.size() + 4;
                    scan.push_back(&row);
                }
                scanned_stats.push_back(out.block_stats.size());
                out.block_stats.push_back(st);
            }
        }
        std::vector<uint8_t> hit(scan.size(), 0);
        bool on_device = false;
        if (cfg_.device_match && row_expr && !scan.empty()) {
            if (int32_t rc = match_rows_device(row_expr, scan, matcher, hit, on_device)) return rc;
        }
        if (!on_device)
            for (size_t i = 0; i < scan.size(); ++i) hit[i] = matcher.match(*scan[i]);
        if (regex) {
            // the regex patterns only run on rows the bloom tree kept AND — when the device matcher is on — on rows whose
            // guard fields exist: the field guard the probe already used, evaluated per row by k_match_rows.  Only when every
            // regex node translated into the guard (regex_guard_is_exact): the reference prunes files and blocks with it, never rows
            std::vector<uint8_t> cand(scan.size(), 1);
            bool guard_on_device = false;
            if (cfg_.device_match && has_guard && regex_guard_is_exact(*regex) && !scan.empty()) {
                RowMatcher guard_host(&guard);
                if (int32_t rc = match_rows_device(&guard, scan, guard_host, cand, guard_on_device)) return rc;
                if (!guard_on_device) std::fill(cand.begin(), cand.end(), 1);
            }
            for (size_t i = 0; i < scan.size(); ++i)
                if (hit[i]) hit[i] = cand[i] && regex_matcher.match(*scan[i]);
        }
        for (size_t i = 0; i < scan.size(); ++i)
            if (hit[i]) out.rows.push_back(*scan[i]);
        if (!scanned_stats.empty()) {
            const int64_t scan_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_scan).count();
            for (size_t i : scanned_stats) out.block_stats[i].duration_ns += std::max<int64_t>(1, scan_ns / (int64_t)scanned_stats.size());
        }
        return kEngineOk;
    }

private:
    struct PartitionBuffer {
        BloomEntrySets entries;
        std::vector<std::string> rows;
        uint64_t bytes = 0;
    };

    EngineConfig cfg_;
    bsg_ctx *ctx_;
    std::map<std::string, PartitionBuffer> buffers_;
    uint64_t buffered_rows_ = 0, buffered_bytes_ = 0;
    std::vector<DataFile> files_;
    uint64_t next_file_id_ = 1;
    bool stopped_ = false;
    std::string err_;
    uint64_t files_arena_ = 0;           // the file-level filters of all files, one "block" per file
    bool files_arena_valid_ = false;
    std::vector<int32_t> file_status_;   // parseFilterSection outcome per file (0 ok)

    int32_t fail(int32_t code, std::string msg) { err_ = std::move(msg); return code; }

    void drop_files_arena()
    {
        if (files_arena_valid_) bsg_arena_free(ctx_, files_arena_);
        files_arena_valid_ = false;
    }
    // ---- the file's block filters through the library's resident-arena cache (cache_api.inc) ----
    struct FileLease {
        uint64_t lease = 0, arena = 0;
        std::vector<uint32_t> rows;        // block b's row in the arena
        std::vector<int32_t> status;       // parseFilterSection outcome per block (0 ok); all 0 for a resident arena (only clean decodes stay)
    };
    static void file_key(const DataFile &f, uint8_t key[8]) { for (int i = 0; i < 8; ++i) key[i] = (uint8_t)(f.file_id >> (8 * i)); }
    // block keys (RowDataOffset) and the sections' extents in the (virtual) file's filter region
    static void block_extents(const DataFile &f, std::vector<uint64_t> &keys, std::vector<uint64_t> &begin, std::vector<uint64_t> &end)
    {
        uint64_t at = 0;
        for (auto &b : f.blocks) {
            keys.push_back(b.block_offset);
            begin.push_back(at);
            at += b.filter_section.size();
            end.push_back(at);
        }
    }
    void forget_file_arena(const DataFile &f)
    {
        uint8_t key[8];
        file_key(f, key);
        bsg_file_arena_forget(ctx_, key, 8);
    }
    void drop_arenas()
    {
        drop_files_arena();
        for (auto &f : files_) forget_file_arena(f);
    }
    // a freshly built / decoded arena of ALL the file's blocks becomes the cache's (and this lease's)
    int32_t publish_file_arena(const DataFile &f, uint64_t arena, const int32_t *status, FileLease &out)
    {
        uint8_t key[8];
        file_key(f, key);
        std::vector<uint64_t> keys, begin, end;
        block_extents(f, keys, begin, end);
        if (bsg_file_arena_publish(ctx_, key, 8, arena, keys.data(), begin.data(), end.data(), status, (uint32_t)keys.size(), &out.lease, nullptr)) {
            bsg_arena_free(ctx_, arena);
            return fail(kErrGpu, bsg_last_error(ctx_));
        }
        out.arena = arena;
        out.rows.resize(keys.size());
        for (size_t b = 0; b < keys.size(); ++b) out.rows[b] = (uint32_t)b;
        if (status) out.status.assign(status, status + keys.size()); else out.status.assign(keys.size(), 0);
        return kEngineOk;
    }
    int32_t lease_file_arena(const DataFile &f, FileLease &out)
    {
        uint8_t key[8];
        file_key(f, key);
        std::vector<uint64_t> keys;
        for (auto &b : f.blocks) keys.push_back(b.block_offset);
        out.rows.assign(keys.size(), 0);
        if (bsg_file_arena_acquire(ctx_, key, 8, keys.data(), (uint32_t)keys.size(), &out.lease, &out.arena, nullptr, out.rows.data()))
            return fail(kErrGpu, bsg_last_error(ctx_));
        if (out.lease) { out.status.assign(keys.size(), 0); return kEngineOk; }
        std::vector<const std::vector<uint8_t> *> bsec;
        for (auto &b : f.blocks) bsec.push_back(&b.filter_section);
        uint64_t arena = 0;
        std::vector<int32_t> status;
        if (int32_t rc = load_arena(bsec, arena, status)) return rc;
        return publish_file_arena(f, arena, status.data(), out);
    }

    // buildFilters for many entry-set triples at once: sizes via EstimateParameters(max(n,1), fpr),
    // one bsg_build, then encodeFilterSection per triple.
    int32_t build_sections(const std::vector<const BloomEntrySets *> &sets, std::vector<std::vector<uint8_t>> &sections)
    {
        std::vector<uint8_t> bytes;
        std::vector<uint32_t> offsets{0}, fstart{0};
        std::vector<bsg_filter_desc> desc(sets.size() * 3);
        uint64_t cursor = 0;
        for (size_t s = 0; s < sets.size(); ++s)
            for (uint32_t c = 0; c < 3; ++c) {
                const auto &set = sets[s]->set_of(c);
                uint64_t m = 0, k = 0;
                if (bsg_estimate_parameters(std::max<uint64_t>(set.size(), 1), cfg_.bloom_false_positive_rate, &m, &k))
                    return fail(kErrGpu, bsg_last_error(ctx_));
                desc[s * 3 + c] = bsg_filter_desc{cursor, m, (uint32_t)k, 0};
                cursor += ((m + 63) / 64 + 1) / 2 * 2;
                pack_entries(set, bytes, offsets);
                fstart.push_back((uint32_t)offsets.size() - 1);
            }
        // build + encodeFilterSection both on the device: only the section bytes cross PCIe
        uint64_t total = 0;
        if (bsg_sections_size(desc.data(), (uint32_t)sets.size(), &total)) return fail(kErrGpu, bsg_last_error(ctx_));
        std::vector<uint8_t> region(total);
        std::vector<uint64_t> sec_off(sets.size() + 1);
        if (bsg_build_sections(ctx_, bytes.data(), offsets.data(), (uint32_t)offsets.size() - 1, fstart.data(), desc.data(),
                               (uint32_t)desc.size(), std::max<uint64_t>(cursor, 2), region.data(), region.size(), sec_off.data()))
            return fail(kErrGpu, bsg_last_error(ctx_));
        split_sections(region, sec_off, sections);
        return kEngineOk;
    }

    static void split_sections(const std::vector<uint8_t> &region, const std::vector<uint64_t> &sec_off,
                               std::vector<std::vector<uint8_t>> &sections)
    {
        sections.resize(sec_off.size() - 1);
        for (size_t s = 0; s + 1 < sec_off.size(); ++s) sections[s].assign(region.begin() + sec_off[s], region.begin() + sec_off[s + 1]);
    }

    // Device ingest of one file's blocks (flush.go:179-254 / merge.go:706-804 with a1-a5 on the GPU): rows ->
    // bsg_ingest_rows, the rows it hands back -> host walker -> bsg_ingest_add_entries, exact counts ->
    // EstimateParameters on the host -> bsg_ingest_build -> encodeFilterSection.  Fills the blocks' and the
    // file's BloomEntryCounts.  sections = one per block, then the file-level one.
    int32_t build_sections_device(DataFile &file, std::vector<std::vector<uint8_t>> &sections)
    {
        const size_t nb = file.blocks.size();
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> row_off{0};
        std::vector<uint32_t> first{0}, parent(nb, 0), set_of_row;
        for (size_t b = 0; b < nb; ++b) {
            for (const std::string &r : file.blocks[b].rows) {
                bytes.insert(bytes.end(), r.begin(), r.end());
                row_off.push_back(bytes.size());
                set_of_row.push_back((uint32_t)b);
            }
            first.push_back((uint32_t)set_of_row.size());
        }
        uint64_t ing = 0;
        if (bsg_ingest_rows(ctx_, bytes.data(), row_off.data(), (uint32_t)set_of_row.size(), first.data(), (uint32_t)nb,
                            parent.data(), 1, nullptr, BSG_INGEST_TRUSTED_JSON /* ingest_rows validated every row */, &ing))
            return fail(kErrGpu, bsg_last_error(ctx_));
        struct Free { bsg_ctx *c; uint64_t id; ~Free() { bsg_ingest_free(c, id); } } guard{ctx_, ing};
        uint32_t n_fb = 0;
        if (bsg_ingest_fallback_rows(ctx_, ing, nullptr, 0, &n_fb)) return fail(kErrGpu, bsg_last_error(ctx_));
        if (n_fb) {
            std::vector<uint32_t> fb(n_fb);
            if (bsg_ingest_fallback_rows(ctx_, ing, fb.data(), n_fb, &n_fb)) return fail(kErrGpu, bsg_last_error(ctx_));
            std::map<uint32_t, BloomEntrySets> per_set;
            for (uint32_t r : fb) {
                const uint32_t b = set_of_row[r];
                per_set[b].index_row(file.blocks[b].rows[r - first[b]]);
            }
            std::vector<uint8_t> eb;
            std::vector<uint32_t> eo{0}, es, ek;
            for (auto &kv : per_set)
                for (uint32_t c = 0; c < 3; ++c) {
                    const size_t before = eo.size() - 1;
                    pack_entries(kv.second.set_of(c), eb, eo);
                    es.insert(es.end(), eo.size() - 1 - before, kv.first);
                    ek.insert(ek.end(), eo.size() - 1 - before, c);
                }
            if (bsg_ingest_add_entries(ctx_, ing, eb.data(), eo.data(), (uint32_t)es.size(), es.data(), ek.data()))
                return fail(kErrGpu, bsg_last_error(ctx_));
        }
        std::vector<uint64_t> counts((nb + 1) * 3);
        std::vector<uint32_t> status(nb + 1);
        if (bsg_ingest_finish(ctx_, ing, counts.data(), status.data())) return fail(kErrGpu, bsg_last_error(ctx_));
        for (uint32_t st : status)
            if (st != 0) return build_sections_host_rows(file, sections);   // a set the device tables cannot represent (2^-62/entry)
        std::vector<bsg_filter_desc> desc((nb + 1) * 3);
        uint64_t cursor = 0;
        for (size_t i = 0; i < desc.size(); ++i) {
            uint64_t m = 0, k = 0;
            if (bsg_estimate_parameters(std::max<uint64_t>(counts[i], 1), cfg_.bloom_false_positive_rate, &m, &k))
                return fail(kErrGpu, bsg_last_error(ctx_));
            desc[i] = bsg_filter_desc{cursor, m, (uint32_t)k, 0};
            cursor += ((m + 63) / 64 + 1) / 2 * 2;
        }
        uint64_t total = 0;
        if (bsg_sections_size(desc.data(), (uint32_t)nb + 1, &total)) return fail(kErrGpu, bsg_last_error(ctx_));
        std::vector<uint8_t> region(total);
        std::vector<uint64_t> sec_off(nb + 2);
        // the block filters stay on the device as this file's probe arena: a query right after the flush uploads nothing
        forget_file_arena(file);
        uint64_t built_arena = 0;
        if (bsg_ingest_build_sections(ctx_, ing, desc.data(), region.data(), region.size(), sec_off.data(), &built_arena, nullptr))
            return fail(kErrGpu, bsg_last_error(ctx_));
        split_sections(region, sec_off, sections);
        uint64_t blk_off = 0;
        for (size_t s = 0; s < nb; ++s) {        // what the cache records: the blocks' keys (RowDataOffset, as the callers assign it) and section extents
            file.blocks[s].filter_section = sections[s];
            file.blocks[s].block_offset = blk_off;
            blk_off += file.blocks[s].row_bytes;
        }
        if (built_arena) {                       // 0: the context does not keep this arena; decoded from the sections on first use
            FileLease l;
            if (int32_t rc = publish_file_arena(file, built_arena, nullptr, l)) return rc;
            bsg_file_arena_release(ctx_, l.lease);
        }
        for (size_t s = 0; s <= nb; ++s) {
            BloomEntryCounts &bc = s < nb ? file.blocks[s].counts : file.counts;
            bc = BloomEntryCounts{counts[s * 3], counts[s * 3 + 1], counts[s * 3 + 2]};
        }
        return kEngineOk;
    }

    // Host walker over every row of the file (the reference's own order of work), filters still built on the GPU.
    int32_t build_sections_host_rows(DataFile &file, std::vector<std::vector<uint8_t>> &sections)
    {
        std::vector<std::unique_ptr<BloomEntrySets>> block_sets;
        BloomEntrySets file_entries;
        std::vector<const BloomEntrySets *> sets;
        for (auto &blk : file.blocks) {
            block_sets.push_back(std::make_unique<BloomEntrySets>());
            for (const std::string &r : blk.rows) block_sets.back()->index_row(r);
            blk.counts = block_sets.back()->counts();
            block_sets.back()->union_into(file_entries);
            sets.push_back(block_sets.back().get());
        }
        sets.push_back(&file_entries);
        file.counts = file_entries.counts();
        return build_sections(sets, sections);
    }

    // matchRowBytes for the whole scan list in one bsg_match_rows call.  on_device stays false (and the host matcher
    // takes over) when the expression is beyond the device matcher's limits.
    int32_t match_rows_device(const BloomExpression *expr, const std::vector<const std::string *> &scan, RowMatcher &host_matcher,
                              std::vector<uint8_t> &hit, bool &on_device)
    {
        MatcherProgram mp(expr);
        if (mp.kinds.size() > 64) return kEngineOk;
        // condition strings as the C-ABI takes them: field i, token i, ... (hashed and fingerprinted on the device)
        std::vector<uint8_t> cbytes;
        std::vector<uint32_t> coff{0};
        for (size_t c = 0; c < mp.kinds.size(); ++c) {
            cbytes.insert(cbytes.end(), mp.fields[c].begin(), mp.fields[c].end()); coff.push_back((uint32_t)cbytes.size());
            cbytes.insert(cbytes.end(), mp.tokens[c].begin(), mp.tokens[c].end()); coff.push_back((uint32_t)cbytes.size());
        }
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> row_off{0};
        for (const std::string *r : scan) { bytes.insert(bytes.end(), r->begin(), r->end()); row_off.push_back(bytes.size()); }
        std::vector<uint64_t> bits((scan.size() + 63) / 64);
        std::vector<uint32_t> fb(scan.size());
        uint32_t n_fb = 0;
        const int32_t rc = bsg_match_rows(ctx_, bytes.data(), row_off.data(), (uint32_t)scan.size(), cbytes.data(), coff.data(), mp.kinds.data(),
                                          (uint32_t)mp.kinds.size(), mp.prog_ops.data(), (uint32_t)mp.prog_ops.size(), bits.data(), fb.data(),
                                          (uint32_t)fb.size(), &n_fb);
        if (rc == BSG_E_UNSUPPORTED) return kEngineOk;      // expression too deep / long: host matcher
        if (rc) return fail(kErrGpu, bsg_last_error(ctx_));
        for (size_t i = 0; i < scan.size(); ++i) hit[i] = (bits[i >> 6] >> (i & 63)) & 1;
        for (uint32_t i = 0; i < n_fb; ++i) hit[fb[i]] = host_matcher.match(*scan[fb[i]]);   // rows outside the device walker's envelope
        on_device = true;
        return kEngineOk;
    }

    // Hand the stored section bytes to the device as they are: CRC32C + big-endian decode run in
    // k_decode_sections (bsg_arena_load_sections); the host never touches a filter word on the read path.
    int32_t load_arena(const std::vector<const std::vector<uint8_t> *> &sections, uint64_t &arena_id, std::vector<int32_t> &status)
    {
        std::vector<uint8_t> region;
        std::vector<uint64_t> off{0};
        for (auto *sec : sections) { region.insert(region.end(), sec->begin(), sec->end()); off.push_back(region.size()); }
        status.assign(std::max<size_t>(sections.size(), 1), 0);
        if (bsg_arena_load_sections(ctx_, region.data(), region.size(), off.data(), (uint32_t)sections.size(), status.data(), &arena_id))
            return fail(kErrGpu, bsg_last_error(ctx_));
        return kEngineOk;
    }

    int32_t ensure_files_arena()
    {
        if (files_arena_valid_) return kEngineOk;
        std::vector<const std::vector<uint8_t> *> fsec;
        for (auto &f : files_) fsec.push_back(&f.filter_section);
        if (int32_t rc = load_arena(fsec, files_arena_, file_status_)) return rc;
        files_arena_valid_ = true;
        return kEngineOk;
    }

    // Each distinct term is hashed once per query, on the device (bsg_hash_entries).
    int32_t hash_terms(const QueryBatch &qb, std::vector<bsg_term> &terms)
    {
        std::vector<uint8_t> bytes;
        std::vector<uint32_t> offsets{0};
        for (auto &s : qb.term_strings) { bytes.insert(bytes.end(), s.begin(), s.end()); offsets.push_back((uint32_t)bytes.size()); }
        std::vector<uint64_t> h(qb.term_strings.size() * 4);
        if (!qb.term_strings.empty() &&
            bsg_hash_entries(ctx_, bytes.data(), offsets.data(), (uint32_t)qb.term_strings.size(), h.data()))
            return fail(kErrGpu, bsg_last_error(ctx_));
        terms.resize(qb.term_strings.size());
        for (size_t i = 0; i < terms.size(); ++i) {
            for (int j = 0; j < 4; ++j) terms[i].h[j] = h[i * 4 + j];
            terms[i].kind = qb.term_kinds[i];
            terms[i].reserved = 0;
        }
        return kEngineOk;
    }
};

}  // namespace bsh
