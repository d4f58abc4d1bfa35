/*
 * rdf_oracle.h — CPU restatement of the reference's scan / join / FILTER algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rdf-fusion_amd/ links, loads or calls this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and
 * only as the checker / the timed CPU baseline — never as the product path.
 *
 * Parity status: the reference is Rust and cannot be compiled here (no cargo/rustc, see
 * SURVEY.md §8c), so this restatement is pinned by the reference's own literal test
 * vectors (tests/golden/reference_kats.json, transcribed from
 * lib/storage/src/memory/storage/{mod,quad_index_data,scan,predicate_pushdown}.rs tests)
 * for the scan / prune / predicate algebra / index choice / push-down parts, and by
 * lib/model/src/xsd in-file numeric tests for promotion rules.  Hash-join, cross-join and
 * FILTER row results have no reference fixture that can be generated offline (they come
 * from DataFusion 52.0.0, absent from the tree): for those rows the oracle is
 * "parity unpinned" against the reference and cross-checked with pyarrow instead.
 *
 * The plan / expression input format is the product's public ABI structs (include/rdfgpu.h)
 * so that one description drives both sides of a parity test.
 */
#ifndef RDF_ORACLE_H
#define RDF_ORACLE_H

#include "../include/rdfgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_store orc_store;

orc_store* orc_store_new(uint32_t batch_size);
void orc_store_free(orc_store* s);
/* IndexPermutations::insert + MemIndexData::insert, bulk form. Returns #new quads. */
uint64_t orc_store_extend(orc_store* s, const uint32_t* g, const uint32_t* sub, const uint32_t* p,
                          const uint32_t* o, uint64_t n);
uint64_t orc_store_remove(orc_store* s, const uint32_t* g, const uint32_t* sub, const uint32_t* p,
                          const uint32_t* o, uint64_t n);
void orc_store_clear(orc_store* s);
uint64_t orc_store_len(const orc_store* s);
/* Bench only: adopt an already sorted+deduplicated permutation (index order columns). */
int orc_store_adopt_sorted(orc_store* s, uint32_t components, const uint32_t* c0, const uint32_t* c1,
                           const uint32_t* c2, const uint32_t* c3, uint64_t n);
int orc_store_read_index(const orc_store* s, uint32_t components, uint32_t* c0, uint32_t* c1,
                         uint32_t* c2, uint32_t* c3, uint64_t cap, uint64_t* n);
/* lexical forms of string ids: heap[offsets[i] .. offsets[i+1]) (rdfgpu_store_set_strings) */
void orc_store_set_strings(orc_store* s, const uint64_t* offsets, uint64_t n_ids, const unsigned char* heap, uint64_t heap_bytes);
/* SPARQL REGEX (regex.rs:47-141), regex_oracle.c: 1 match / 0 no match / -1 error value */
int orc_regex_is_match(const char* pattern, size_t pattern_len, const char* flags, size_t flags_len, const unsigned char* subject, size_t subject_len);
void orc_store_set_typed_values(orc_store* s, const rdfgpu_typed_value* values, uint64_t n_ids,
                                const int64_t* decimals, uint64_t n_decimals);
/* 0 = array lookup for ENC_TV (default), 1 = hash-map lookup per row like the reference's DashMap */
void orc_store_set_faithful_decode(orc_store* s, int on);

/* --- host logic --------------------------------------------------------------------- */
uint64_t orc_scan_score(const rdfgpu_scan_instruction instr[4]);
uint32_t orc_choose_index(const rdfgpu_scan_instruction gspo[4], uint32_t available);
int orc_predicate_and(const rdfgpu_predicate* lhs, const rdfgpu_predicate* rhs, rdfgpu_predicate* out,
                      uint32_t* out_ids);
int orc_pushdown_to_scan_predicate(uint32_t op, uint32_t value, rdfgpu_predicate* out);

/* --- MemColumnChunk::find_range_between (quad_index_data.rs:600-650) ------------------ */
enum { ORC_FR_BEFORE = 0, ORC_FR_NOT_CONTAINED = 1, ORC_FR_CONTAINED = 2, ORC_FR_AFTER = 3 };
int orc_find_range_between(const uint32_t* values, uint64_t n, uint32_t from, uint32_t to,
                           uint64_t* lo, uint64_t* hi);

/* --- MemIndexData::prune_relevant_row_groups (quad_index_data.rs:155-284) ------------- */
/* `instr` is in the order of index `components`. Returns the number of row-group slices
   (<= cap) and writes their (start,end) flat row offsets; `dropped_mask` bit i is set when
   predicate i was proven by pruning and removed. */
int orc_prune(const orc_store* s, uint32_t components, const rdfgpu_scan_instruction instr[4],
              const uint32_t* pool, uint64_t* starts, uint64_t* ends, uint32_t cap,
              uint32_t* dropped_mask);

/* --- MemQuadIndexScanIterator (scan.rs:104-340) ---------------------------------------- */
typedef struct orc_scan_result {
  uint32_t n_cols;          /* bound variables, G,S,P,O order of first occurrence           */
  uint32_t vars[4];
  uint64_t n_rows;
  uint32_t* cols[4];        /* malloc'ed, n_rows each                                       */
  uint32_t n_batches;
  uint32_t* batch_rows;     /* malloc'ed: rows of every emitted batch, in order             */
  uint32_t chosen_index;    /* RDFGPU_GSPO..                                                */
} orc_scan_result;
int orc_scan(const orc_store* s, const rdfgpu_scan_instruction gspo[4], const uint32_t* pool,
             int force_index /* -1 = choose */, orc_scan_result* out);
void orc_scan_result_free(orc_scan_result* r);

/* --- plans (scan -> filter -> joins), batch-at-a-time like DataFusion ------------------ */
typedef struct orc_table {
  uint32_t n_cols;
  uint64_t n_rows;
  uint32_t* cols[RDFGPU_MAX_COLUMNS];
} orc_table;
typedef struct orc_bound_table {
  const uint32_t* const* cols;
  uint32_t n_cols;
  uint64_t n_rows;
} orc_bound_table;
int orc_plan_execute(const orc_store* s, const rdfgpu_plan_desc* desc, const orc_bound_table* tables,
                     uint32_t n_tables, orc_table* out, rdfgpu_metrics* metrics);
void orc_table_free(orc_table* t);
const char* orc_last_error(void);

/* --- expression evaluation on explicit columns (FILTER semantics unit tests) ----------- */
/* out[i] = 0 false, 1 true, 2 null/error.  The program must leave a BOOL on the stack. */
int orc_eval_bool(const orc_store* s, const rdfgpu_expr_node* prog, uint32_t n, const uint32_t* const* cols,
                  uint32_t n_cols, uint64_t n_rows, uint8_t* out);
/* The program must leave a TV on the stack; decimals come back in out_hi/out.lo. */
int orc_eval_tv(const orc_store* s, const rdfgpu_expr_node* prog, uint32_t n, const uint32_t* const* cols,
                uint32_t n_cols, uint64_t n_rows, rdfgpu_typed_value* out, int64_t* out_hi);

void orc_eval_set_table(const rdfgpu_regex* regexes, uint32_t n_regexes);
/* A program that leaves a STRING on the stack, with the plan's string table (REGEX patterns / string constants) it refers to:
   out_state[i] = 0 the error value / not a string, 1 a string whose bytes are out_bytes[out_off[i] .. out_off[i + 1]); out_lang[i] =
   its language id.  out_off has n_rows + 1 entries; returns -1 (orc_last_error) when out_bytes (cap bytes) is too small. */
int orc_eval_str(const orc_store* s, const rdfgpu_expr_node* prog, uint32_t n, const rdfgpu_regex* regexes, uint32_t n_regexes,
                 const uint32_t* const* cols, uint32_t n_cols, uint64_t n_rows, uint8_t* out_state, uint32_t* out_lang,
                 uint64_t* out_off, uint8_t* out_bytes, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
