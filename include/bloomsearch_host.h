/*
 * bloomsearch_host.h — C-ABI of the host-side mirror that sits ABOVE bloomgpu.h.
 *
 * The reference's host is Go; this image has no Go toolchain, so the host side of the path is
 * written in C++ (bloomsearch_amd/csrc/host/) mirroring the reference's own functions, and exposed
 * here so tests (and any embedding) can drive it.  A Go host does NOT need this header: it keeps its
 * own tokenizer.go / ingest.go / query.go and binds bloomgpu.h directly.
 *
 * Mirrors (reference file:line):
 *   bsh_tokenize ............. BasicWhitespaceLowerTokenizer, tokenizer.go:141-143
 *   bsh_entry_sets_* ......... bloomEntrySets.indexRow/unionInto/counts, ingest.go:24-123
 *   bsh_batch_* .............. Field/Token/FieldToken/And/Or trees (JSON of the exported structs,
 *                              query.go:478-610) lowered to bloomgpu.h terms + programs
 *   bsh_match_row ............ testJSONForBloomQuery / compiledRowMatcher, row_matcher.go:486-626
 *   bsh_section_* ............ encodeFilterSection / parseFilterSection, file_format.go:343-448
 *   bse_* .................... BloomSearchEngine IngestRows / Flush / Query / Merge
 *                              (ingest.go:170,197; query_exec.go:201; merge.go:35)
 * All functions return 0 on success or a negative code; buffers returned through `char **` are
 * malloc'd and released with bsh_free.
 */
#ifndef BLOOMSEARCH_HOST_H
#define BLOOMSEARCH_HOST_H

#include <stdint.h>
#include "bloomgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BSH_E_INVALID        -1
#define BSE_E_INVALID_CONFIG -101  /* ErrInvalidConfig */
#define BSE_E_ENGINE_STOPPED -102  /* ErrEngineStopped */
#define BSE_E_INVALID_ROW    -103
#define BSE_E_INVALID_QUERY  -104
#define BSE_E_GPU            -105
#define BSE_E_INVALID_HASH   -106  /* ErrInvalidHash */

BSG_API void bsh_free(void *p);

/* tokens of `text`, joined by '\n' (a token never contains white space) */
BSG_API int32_t bsh_tokenize(const uint8_t *text, uint64_t len, char **out, uint64_t *out_len);

typedef struct bsh_entry_sets bsh_entry_sets;
BSG_API bsh_entry_sets *bsh_entry_sets_new(void);
BSG_API void bsh_entry_sets_free(bsh_entry_sets *s);
/* indexRow on one marshaled-JSON row; BSH_E_INVALID if it is not valid JSON */
BSG_API int32_t bsh_entry_sets_index_row(bsh_entry_sets *s, const uint8_t *row, uint64_t len);
BSG_API int32_t bsh_entry_sets_union_into(const bsh_entry_sets *src, bsh_entry_sets *dst);
BSG_API void bsh_entry_sets_counts(const bsh_entry_sets *s, uint64_t counts[3]);
/* packed export of one kind (0 field, 1 token, 2 field::token): sizes, then fill */
BSG_API int32_t bsh_entry_sets_export_sizes(const bsh_entry_sets *s, uint32_t kind, uint64_t *n_entries, uint64_t *n_bytes);
BSG_API int32_t bsh_entry_sets_export(const bsh_entry_sets *s, uint32_t kind, uint8_t *bytes, uint32_t *offsets);

typedef struct bsh_batch bsh_batch;
BSG_API bsh_batch *bsh_batch_new(void);
BSG_API void bsh_batch_free(bsh_batch *b);
/* expression JSON in the reference's struct shape; "null" / empty = nil query */
BSG_API int32_t bsh_batch_add_query(bsh_batch *b, const char *expr_json, uint64_t len);
BSG_API void bsh_batch_sizes(const bsh_batch *b, uint32_t *n_queries, uint32_t *n_terms, uint32_t *n_ops, uint64_t *term_bytes);
BSG_API int32_t bsh_batch_export(const bsh_batch *b, uint8_t *term_bytes, uint32_t *term_offsets, uint32_t *term_kinds,
                                 uint32_t *prog_ops, uint32_t *prog_off);

/* final exact test of one row against one expression: 1 match, 0 no match, negative error */
BSG_API int32_t bsh_match_row(const char *expr_json, uint64_t expr_len, const uint8_t *row, uint64_t row_len);
/* pruneBloomQuery = AndBloomQueries(bloom, RegexFieldGuardBloomQuery(regex)) (query_exec.go:220, query.go:651-718) as JSON in the
 * reference's struct shape; "null" when both sides are nil.  regex_json: {"ExpressionType": "CONDITION"|"AND"|"OR",
 * "Condition": {"Field", "Pattern"}, "Children": [...]}. */
BSG_API int32_t bsh_prune_query(const char *bloom_json, uint64_t bloom_len, const char *regex_json, uint64_t regex_len, char **out, uint64_t *out_len);
/* The regex half of the final row test (row_matcher.go:548-573; std::regex ECMAScript stands in for RE2): 1 / 0 / < 0. */
BSG_API int32_t bsh_match_row_regex(const char *regex_json, uint64_t regex_len, const uint8_t *row, uint64_t row_len);

/* filter section codec; filters[c].m == 0 => absent */
BSG_API int32_t bsh_section_encode(const uint64_t *const words[3], const uint64_t m[3], const uint64_t k[3],
                                   uint8_t **out, uint64_t *out_len);
/* on success fills m/k and returns each present filter's words (malloc'd, native LE) */
BSG_API int32_t bsh_section_parse(const uint8_t *section, uint64_t len, uint64_t m[3], uint64_t k[3], uint64_t *words[3]);
BSG_API uint32_t bsh_crc32c(const uint8_t *data, uint64_t len);

/* ---- engine mirror ---- */
typedef struct bse_engine bse_engine;
/* config_json: {"MaxRowGroupRows":..,"MaxRowGroupBytes":..,"MaxBufferedRows":..,"MaxBufferedBytes":..,
 *               "BloomFalsePositiveRate":..,"PartitionField":"..","DeviceIngest":true|false,"DeviceMatch":true|false}; missing keys take the
 *               reference defaults.  DeviceIngest (default false): rows are walked / tokenized / deduplicated /
 *               counted on the GPU at flush and merge time (bloomgpu.h bsg_ingest_*) instead of by indexRow on the
 *               host at ingest time; the files it writes are byte-identical either way.  DeviceMatch (default false): the
 *               final row test of the surviving blocks runs on the GPU (bsg_match_rows) instead of in the host matcher;
 *               the delivered row set is the same. */
BSG_API int32_t bse_open(const char *config_json, uint64_t len, bsg_ctx *ctx, bse_engine **out);
BSG_API void bse_close(bse_engine *e);
BSG_API const char *bse_last_error(bse_engine *e);
BSG_API int32_t bse_stop(bse_engine *e);
/* rows: marshaled JSON objects separated by '\n' */
BSG_API int32_t bse_ingest_rows(bse_engine *e, const uint8_t *ndjson, uint64_t len);
BSG_API int32_t bse_flush(bse_engine *e);
BSG_API int32_t bse_merge(bse_engine *e);
/* query_json: {"Bloom":{"Expression":{...}}} (or {"Bloom":null}); result JSON:
 * {"rows":[...],"stats":{"BlockStats":[{"FileID","BlockOffset","RowsProcessed","BytesProcessed","TotalRows",
 *  "TotalBytes","BloomFilterSkipped"}],"Errors":[..],"FilesConsidered","FilesBloomSkipped"}} */
BSG_API int32_t bse_query(bse_engine *e, const char *query_json, uint64_t len, char **out_json, uint64_t *out_len);
/* {"files":[{"FileID","BloomEntryCounts":{..},"section_bytes","blocks":[{"PartitionID","Rows","BloomEntryCounts":{..},
 *  "BloomFalsePositiveRate","BloomFilterSize","filters":[{"m","k"}|null x3]}]}]} */
BSG_API int32_t bse_describe(bse_engine *e, char **out_json, uint64_t *out_len);
/* fault injection (the reference's tests wrap its stores to corrupt reads): XOR one byte of a stored section */
BSG_API int32_t bse_corrupt_section_byte(bse_engine *e, uint32_t file_index, int32_t block_index, uint64_t byte_index);
/* raw filter-section bytes of (file index, block index); block index -1 = the file-level section */
BSG_API int32_t bse_section_bytes(bse_engine *e, uint32_t file_index, int32_t block_index, uint8_t **out, uint64_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
