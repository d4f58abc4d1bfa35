/*
 * rdfgpu.h — C ABI of the MI355X-native BGP-scan + hash-join + FILTER path.
 *
 * This is the drop-in boundary for the one hot path of tobixdev/rdf-fusion that this
 * repository accelerates: triple-pattern scan over sorted u32 quad
 * indexes -> cross/hash join of binding tables -> SPARQL FILTER.  Everything in here is
 * plain C: pointers, sizes, PODs.  No C++/torch/Arrow-library types cross the boundary;
 * result batches leave through the Arrow C Data Interface structs declared below.
 *
 * Every entry point names the reference interface (file:line, relative to the
 * rdf-fusion tree) that it replaces.  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   - All functions return an rdfgpu_status (0 ok, >0 informational, <0 error) unless
 *     stated otherwise.  On error, rdfgpu_last_error() returns a thread-local message.
 *   - Object ids are u32.  Id 0 is the default graph AND the null / unbound marker,
 *     exactly as in the reference (lib/storage/src/memory/object_id.rs:21,80;
 *     quad_index_data.rs:438-440 stores 0 as an Arrow null).
 *   - The caller owns every input buffer; the library copies what it keeps.
 *   - A handle may be used from any thread, one in-flight call per handle.
 *   - There is NO CPU fallback: entry points that touch data fail with
 *     RDFGPU_ERR_NO_DEVICE when no gfx950 device is usable.  The host-logic entry
 *     points (section 5) never touch the device.
 */
#ifndef RDFGPU_H
#define RDFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDFGPU_ABI_VERSION 3u

/* ------------------------------------------------------------------------------------ */
/* 0. Status codes                                                                       */
/* ------------------------------------------------------------------------------------ */
typedef enum rdfgpu_status {
  RDFGPU_OK = 0,
  RDFGPU_END = 1,              /* rdfgpu_plan_next: stream exhausted                      */
  RDFGPU_ERR_INVALID = -1,     /* malformed argument / plan (DataFusionError::Plan)       */
  RDFGPU_ERR_DEVICE = -2,      /* HIP runtime error (DataFusionError::External)           */
  RDFGPU_ERR_UNSUPPORTED = -3, /* valid but not implemented on device                     */
  RDFGPU_ERR_OOM = -4,         /* device allocation failed                                */
  RDFGPU_ERR_NO_DEVICE = -5    /* no usable gfx950 device; there is no CPU fallback       */
} rdfgpu_status;

/* Thread-local text of the last error raised on the calling thread ("" if none). */
const char* rdfgpu_last_error(void);
uint32_t rdfgpu_abi_version(void);

/* ------------------------------------------------------------------------------------ */
/* 1. Arrow C Data Interface (verbatim ABI structs, https://arrow.apache.org/docs/format/CDataInterface.html) */
/* ------------------------------------------------------------------------------------ */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif

/* ------------------------------------------------------------------------------------ */
/* 2. Store: three sorted u32 quad permutations + typed-value table, resident in HBM     */
/*    replaces MemQuadStorage (lib/storage/src/memory/storage/mem_storage.rs:22-102)     */
/*    and the MemIndexData layout (quad_index_data.rs:57-64)                             */
/* ------------------------------------------------------------------------------------ */
typedef struct rdfgpu_store rdfgpu_store;

typedef struct rdfgpu_config {
  int32_t device;        /* HIP device ordinal; -1 = current device                       */
  uint32_t batch_size;   /* rows per exported batch; 0 = 8192 (store.rs:102)              */
  uint32_t flags;        /* reserved, 0                                                   */
  uint32_t reserved;
} rdfgpu_config;

/* Index permutations, in the reference's listing order (mem_storage.rs:41-45).          */
enum { RDFGPU_GSPO = 0, RDFGPU_GPOS = 1, RDFGPU_GOSP = 2, RDFGPU_N_INDEXES = 3 };

/*
 * Device-side typed value of one object id: the FILTER side table.
 * Mirrors TypedValueEncodingField type ids (lib/encoding/src/typed_value/encoding.rs:248-268)
 * and the id -> typed value decode of object_id_mapping.rs:376-399.
 *   tag 0 null/invalid (parse failure => null, typed_value.rs:349-411)
 *   tag 1 named node, 2 blank node: lo = rank of the IRI / label in str order
 *   tag 3 string: lo = rank of the lexical value in str order; aux = language id
 *                 (0 = simple literal); flags bit0 = value is the empty string
 *   tag 4 boolean: lo = 0/1
 *   tag 5 float:   lo = IEEE-754 binary32 bits (zero-extended)
 *   tag 6 double:  lo = IEEE-754 binary64 bits
 *   tag 7 decimal: lo = index into the i128 side table (value * 10^18, decimal.rs:9-21)
 *   tag 8 int (i32, sign-extended), tag 9 integer (i64)
 *   tag 10 dateTime, 11 time, 12 date: a Timestamp (lib/model/src/xsd/date_time.rs:1599-1603; the host reads it with
 *              DateTime::timestamp() / Time::timestamp() / Date::timestamp()): lo = index into the i128 side table
 *              holding Timestamp.value (seconds on the XSD time line * 10^18, already shifted to UTC when a timezone
 *              is present), aux bit0 = timezone_offset.is_some().  Compared as PartialOrd for Timestamp does
 *              (date_time.rs:1617-1654): same presence => the values; otherwise the value without a timezone may lie
 *              14 h either way and the order must hold for both, else the comparison is an error.  Same kind only.
 *   tag 13 duration, 14 other literal: opaque on device (other literals: equal iff same datatype (aux) and same
 *              lexical value (lo), typed_value.rs:253-258); an operator that would need their value yields the
 *              SPARQL error value, i.e. the row is dropped by a FILTER — durations order by calendar arithmetic
 *              (duration.rs:271-310), not restated here.
 */
typedef struct rdfgpu_typed_value {
  int64_t lo;
  uint32_t aux;
  uint8_t tag;
  uint8_t flags;
  uint16_t reserved;
} rdfgpu_typed_value; /* 16 bytes, one aligned dwordx4 gather per row */

enum {
  RDFGPU_TV_NULL = 0, RDFGPU_TV_NAMED_NODE = 1, RDFGPU_TV_BLANK_NODE = 2, RDFGPU_TV_STRING = 3,
  RDFGPU_TV_BOOLEAN = 4, RDFGPU_TV_FLOAT = 5, RDFGPU_TV_DOUBLE = 6, RDFGPU_TV_DECIMAL = 7,
  RDFGPU_TV_INT = 8, RDFGPU_TV_INTEGER = 9, RDFGPU_TV_DATE_TIME = 10, RDFGPU_TV_TIME = 11,
  RDFGPU_TV_DATE = 12, RDFGPU_TV_DURATION = 13, RDFGPU_TV_OTHER = 14
};
#define RDFGPU_TVF_EMPTY_STRING 1u
#define RDFGPU_TVF_NEEDS_HOST 0x80u   /* rdfgpu_ntriples_decoded only: the literal's value is left to the host's parser (see there) */

/* MemQuadStorage::new (mem_storage.rs:31-65): creates the store on `cfg->device`.  The first store a process creates also has the HIP
   runtime load every code object of this library (it would otherwise do so at the first launch from each translation unit: some
   tens of milliseconds that the first query would pay). */
int rdfgpu_store_create(const rdfgpu_config* cfg, rdfgpu_store** out);
void rdfgpu_store_destroy(rdfgpu_store* store);

/*
 * QuadStorage::extend (lib/extensions/src/storage/quad_storage.rs:35; mem_storage.rs:95-102;
 * IndexPermutations::insert permutations.rs:102-118): merges `n` encoded quads (host
 * buffers, ids assigned by the host dictionary) into all three permutations; duplicates
 * are ignored.  `*inserted` receives the number of quads that were new.
 */
int rdfgpu_store_extend(rdfgpu_store* store, const uint32_t* g, const uint32_t* s,
                        const uint32_t* p, const uint32_t* o, uint64_t n, uint64_t* inserted);
/* Same, with the four columns already resident in this device's HBM. */
int rdfgpu_store_extend_device(rdfgpu_store* store, const uint32_t* g, const uint32_t* s,
                               const uint32_t* p, const uint32_t* o, uint64_t n,
                               uint64_t* inserted);
/* QuadStorage::remove (quad_storage.rs:38; permutations.rs:120-128). */
int rdfgpu_store_remove(rdfgpu_store* store, const uint32_t* g, const uint32_t* s,
                        const uint32_t* p, const uint32_t* o, uint64_t n, uint64_t* removed);
/* QuadStorage::clear (quad_storage.rs:56). */
int rdfgpu_store_clear(rdfgpu_store* store);
/* QuadStorage::clear_graph (quad_storage.rs:59-62) and the quads of QuadStorage::drop_named_graph (:65-68): removes every quad
   whose graph is `graph` (0 = the default graph; the registry of named graphs is the host dictionary's). */
int rdfgpu_store_remove_graph(rdfgpu_store* store, uint32_t graph, uint64_t* removed);
/* QuadStorage::len (quad_storage.rs:71). */
int rdfgpu_store_len(const rdfgpu_store* store, uint64_t* out);
/* Cache control (no counterpart in the reference, whose HashJoinExec builds its table per query): forgets every join table
   cached on the store (direct-address / CSR / hash tables of predicate slices, decoded value tables, band-join entries) and
   bumps the store version — exactly what every extend / remove / clear does to the caches, without touching the quads.  The
   next execution of any plan locates its ranges and builds the tables it needs again, inside that execution.  Used to
   measure the "tables rebuilt in the step" figure (bench.py: config.cold_start.fused_rebuild). */
int rdfgpu_store_drop_tables(rdfgpu_store* store);

/*
 * Installs the id -> typed value table (MemObjectIdMapping::decode_array_to_typed_value,
 * object_id_mapping.rs:376-399).  values[i] describes object id i (values[0] is ignored:
 * id 0 is null).  `decimals` holds n_decimals little-endian i128 as (lo,hi) int64 pairs: the
 * values of xsd:decimal literals (tag 7) and the timestamps of tags 10..12.
 */
int rdfgpu_store_set_typed_values(rdfgpu_store* store, const rdfgpu_typed_value* values,
                                  uint64_t n_ids, const int64_t* decimals, uint64_t n_decimals);

/*
 * Installs the lexical forms of the string-valued object ids in HBM (the part of the reference's
 * MemObjectIdMapping dictionary that string builtins read: object_id_mapping.rs:48-59, 262-276).
 * The UTF-8 lexical form of object id i is heap[offsets[i] .. offsets[i + 1]); offsets has
 * n_ids + 1 entries; only ids whose typed value has tag RDFGPU_TV_STRING are ever read.
 * Needed by RDFGPU_EX_REGEX; a plan using it on a store without strings fails to compile.
 */
int rdfgpu_store_set_strings(rdfgpu_store* store, const uint64_t* offsets, uint64_t n_ids,
                             const uint8_t* heap, uint64_t heap_bytes);

/*
 * Test / debug access to a sorted permutation (MemIndexData, quad_index_data.rs:57-64):
 * copies up to `cap` rows of index `components` (RDFGPU_GSPO..) into four host columns,
 * in index order (e.g. g,p,o,s for GPOS).  `*n` receives the index length.
 */
int rdfgpu_store_read_index(const rdfgpu_store* store, uint32_t components, uint32_t* c0,
                            uint32_t* c1, uint32_t* c2, uint32_t* c3, uint64_t cap, uint64_t* n);

/* ------------------------------------------------------------------------------------ */
/* 3. Plan description: what DataFusion hands over for this path                         */
/* ------------------------------------------------------------------------------------ */

/*
 * One level of a quad pattern scan = MemIndexScanInstruction + MemIndexScanPredicate
 * (lib/storage/src/memory/storage/scan_instructions.rs:157-166, 247-252).
 */
enum { RDFGPU_TRAVERSE = 0, RDFGPU_SCAN = 1 };
enum { RDFGPU_PRED_NONE = 0, RDFGPU_PRED_FALSE = 1, RDFGPU_PRED_IN = 2, RDFGPU_PRED_BETWEEN = 3,
       RDFGPU_PRED_EQUAL_TO = 4 };

typedef struct rdfgpu_scan_instruction {
  uint8_t kind;      /* RDFGPU_TRAVERSE | RDFGPU_SCAN                                     */
  uint8_t pred;      /* RDFGPU_PRED_*                                                     */
  uint16_t reserved;
  uint32_t var;      /* SCAN: variable slot this level binds                              */
  uint32_t a;        /* IN: offset into the plan's u32 pool (sorted ascending, unique);
                        BETWEEN: from;  EQUAL_TO: variable slot it must equal             */
  uint32_t b;        /* IN: number of ids;  BETWEEN: to (inclusive)                       */
} rdfgpu_scan_instruction;

/*
 * Expression programs (postfix) = the PhysicalExpr trees DataFusion evaluates in
 * FilterExec / JoinFilter for this path, with the reference's UDF names
 * (lib/extensions/src/functions/builtin.rs:101-186).  Three value kinds live on the
 * evaluation stack: ID (u32 object id, 0 = null), TV (typed value, tag 0 = null/error),
 * BOOL (native nullable boolean).
 */
enum {
  RDFGPU_EX_COLUMN = 1,       /* -> ID      push input column `u` (join filters: left columns first, then right) */
  RDFGPU_EX_LIT_ID = 2,       /* -> ID      push object id literal `u`                                         */
  RDFGPU_EX_LIT_TV = 3,       /* -> TV      push typed literal (tag, lo, aux, flags), plan text "9:120"       */
  RDFGPU_EX_ENC_TV = 4,       /* ID -> TV   ENC_TV: with_typed_value_encoding.rs:71-79                         */
  RDFGPU_EX_GT = 5,           /* TV TV -> TV(boolean|null)  greater_than.rs:41-64                              */
  RDFGPU_EX_LT = 6,           /*            less_than.rs                                                        */
  RDFGPU_EX_GEQ = 7,
  RDFGPU_EX_LEQ = 8,
  RDFGPU_EX_EQ = 9,           /*            equal.rs (typed `=`)                                                */
  RDFGPU_EX_ADD = 10,         /* TV TV -> TV  add.rs:40-86                                                      */
  RDFGPU_EX_SUB = 11,         /*              sub.rs                                                            */
  RDFGPU_EX_EBV = 12,         /* TV -> BOOL   effective_boolean_value.rs:99-119                                 */
  RDFGPU_EX_ID_EQ = 13,       /* ID ID -> BOOL  native UInt32 `=`  (expression_simplifier.rs:162-257)          */
  RDFGPU_EX_ID_NEQ = 14,      /* ID ID -> BOOL  native `!=` (the `product != <object id>` FilterExec)          */
  RDFGPU_EX_AND = 15,         /* BOOL BOOL -> BOOL  SQL three-valued (expr_builder_context.rs:393-434)         */
  RDFGPU_EX_OR = 16,
  RDFGPU_EX_NOT = 17,         /* BOOL -> BOOL                                                                   */
  RDFGPU_EX_IS_COMPATIBLE = 18, /* ID ID -> BOOL  is_compatible.rs:97-136                                      */
  RDFGPU_EX_BOUND = 19,       /* ID -> BOOL   functional_form/bound.rs:21                                       */
  RDFGPU_EX_BOOL_AS_TV = 20,  /* BOOL -> TV   BOOLEAN_AS_TERM (expr_builder.rs:694-698)                        */
  RDFGPU_EX_LIT_BOOL = 21,    /* -> BOOL      literal true (u=1) / false (u=0) / null (u=2)                    */
  RDFGPU_EX_NEQ = 22,         /* TV TV -> TV  NOT(EQ) kept as one op for convenience                            */
  RDFGPU_EX_REGEX = 23,       /* TV -> TV(boolean|null)  REGEX(value, <constant pattern>[, <constant flags>]),
                                 scalar/strings/regex.rs:47-141; `u` indexes rdfgpu_plan_desc.regexes.  The value
                                 must come from ENC_TV of a column; simple / language strings match, anything else is
                                 the error value.  Syntax outside csrc/regex_compile.hpp's subset is RDFGPU_ERR_UNSUPPORTED at
                                 compile time; `\d \w \s \b` are compiled with their ASCII members and a row whose subject
                                 has a non-ASCII byte fails the execute with RDFGPU_ERR_UNSUPPORTED (the crate's Unicode tables
                                 are not restated).  Per-row patterns: RDFGPU_EX_REGEX_VAR.                        */
  RDFGPU_EX_CONTAINS = 24,    /* TV -> TV(boolean|null)  CONTAINS(value, <constant string>), scalar/strings/contains.rs;
                                 `u` indexes rdfgpu_plan_desc.regexes (the entry's pattern is the needle, taken literally; its
                                 flags are ignored), `lo` = language id of the constant (0 = simple literal).  Argument
                                 compatibility (string_literal.rs:80-95): the value must be a string and the constant must
                                 have no language or the value's, else the error value.                                  */
  RDFGPU_EX_STRSTARTS = 25,   /* same shape: STRSTARTS, scalar/strings/str_starts.rs                                     */
  RDFGPU_EX_STRENDS = 26,     /* same shape: STRENDS, scalar/strings/str_ends.rs                                         */
  RDFGPU_EX_LANG_IN = 27,     /* TV -> TV(boolean|null)  LANGMATCHES(LANG(value), <constant range>) — scalar/terms/lang.rs:45-58,
                                 scalar/strings/lang_matches.rs:52-69, as planned for BSBM explore Q8 (`EBV(LANGMATCHES(LANG(ENC_TV(text)),
                                 3:{value:EN,language:}))`, Q8 (Execution Plan).snap:18).  The range is resolved by the host against its
                                 (small) language dictionary: regexes[u].pattern holds ONE BYTE PER LANGUAGE ID (the `aux` numbering of
                                 string values; byte 0 = the empty tag, i.e. every literal without a language), 1 = the tag matches
                                 the range.  Named / blank nodes and null => error (lang.rs:50-52); a language id beyond the table
                                 => error.  At most 16384 language ids.                                                       */
  RDFGPU_EX_REGEX_VAR = 28,   /* TV TV -> TV(boolean|null)  REGEX(value, ?pattern[, <constant flags>]) with a PER-ROW pattern
                                 (regex.rs:59-76 compiles the pattern of every row; testsuite/oxigraph-tests/sparql/regex_variable.rq).
                                 The second operand is the pattern's typed value (ENC_TV of its column): a simple literal, else the
                                 error value.  The host announces the DISTINCT patterns that can occur: rdfgpu_plan_desc.regexes
                                 [u .. u + lo), each with its object id (rdfgpu_regex.pattern_id) and the constant flags; they are
                                 compiled at plan time and a row picks its program by the pattern's id.  A row whose pattern was not
                                 announced fails the execute (RDFGPU_ERR_UNSUPPORTED), it is never answered as "no match".            */
  /* ---- ABI 3: string-valued expressions.  A string value on the device is a VIEW: the lexical form of an object
     id in the store's string heap (rdfgpu_store_set_strings) or a constant of the plan, with a byte window (SUBSTR) and an ASCII
     case mapping (UCASE / LCASE) applied on the fly; nothing is materialised per row.  Views are consumed by REGEX / CONTAINS /
     STRSTARTS / STRENDS (whose operand may now be any string value, not only ENC_TV of a column), STRLEN, EBV and the
     comparisons (two strings of which at least one is a view compare byte-wise — `str` order — when their languages agree). */
  RDFGPU_EX_STR = 29,         /* ID -> TV(simple literal)  STR(term), scalar/terms/str.rs:42: over an object-id column the reference
                                 plans STR in the plain-term encoding (decide_input_encoding, expr_builder_context.rs:557-582:
                                 ObjectId is not supported, PlainTerm is the first that is), i.e. the lexical form AS WRITTEN of IRIs,
                                 blank nodes and every literal ("010"^^xsd:int -> "010", lib/functions/tests/snapshots/
                                 unary__STR(PLAIN_TERM).snap).  Null / an id without a lexical form in the heap => the error value. */
  RDFGPU_EX_LIT_STR = 30,     /* -> TV(string)  a string constant WITH its bytes: `u` indexes rdfgpu_plan_desc.regexes (the entry's
                                 pattern is the text), `lo` = its language id (0 = simple literal).  What a view is compared with:
                                 an RDFGPU_EX_LIT_TV string carries only its rank in the dictionary's order.                     */
  RDFGPU_EX_STRLEN = 31,      /* TV -> TV(integer)  STRLEN, scalar/strings/strlen.rs: characters (code points) of a simple or
                                 language-tagged string; anything else => error.                                                  */
  RDFGPU_EX_SUBSTR = 32,      /* TV TV [TV] -> TV(string)  SUBSTR(str, start[, length]), scalar/strings/sub_str.rs:83-121: `u` = 2 or 3
                                 operands; 1-based character positions; start < 1 or a negative length => error; a start beyond the
                                 end => ""; the language of the source is kept.  start / length must be xsd:int / xsd:integer values
                                 (a float / double / decimal argument fails the execute with RDFGPU_ERR_UNSUPPORTED).             */
  RDFGPU_EX_UCASE = 33,       /* TV -> TV(string)  UCASE, scalar/strings/ucase.rs (str::to_uppercase), language kept.  ASCII letters are
                                 mapped on the device; a string with a non-ASCII byte under a case mapping fails the execute with
                                 RDFGPU_ERR_UNSUPPORTED (Unicode case tables are not restated; never answered differently).      */
  RDFGPU_EX_LCASE = 34,       /* same: LCASE, scalar/strings/lcase.rs (str::to_lowercase)                                        */
  RDFGPU_EX_STRBEFORE = 35,   /* TV TV -> TV(string)  STRBEFORE(a, b), scalar/strings/str_before.rs: both string literals, b without a
                                 language or with a's (string_literal.rs:80-95, else error); the part of a before the first occurrence
                                 of b with a's language — a VIEW of a, nothing is copied; b does not occur => the simple literal "".    */
  RDFGPU_EX_STRAFTER = 36,    /* same: STRAFTER, scalar/strings/str_after.rs — the part of a behind the first occurrence of b             */
  RDFGPU_EX__COUNT
};

typedef struct rdfgpu_expr_node {
  uint8_t op;       /* RDFGPU_EX_*                                                        */
  uint8_t tag;      /* LIT_TV: typed value tag                                            */
  uint8_t flags;    /* LIT_TV: rdfgpu_typed_value.flags                                   */
  uint8_t reserved;
  uint32_t u;       /* COLUMN: column index; LIT_ID: id; LIT_TV: aux; LIT_BOOL: 0/1/2     */
  int64_t lo;       /* LIT_TV payload (decimal / dateTime / time / date literals: low 64 bits of the i128) */
  int64_t hi;       /* LIT_TV decimal / timestamp literal: high 64 bits; timestamps: u bit0 = has timezone */
} rdfgpu_expr_node; /* 24 bytes */

/* Physical operators of the path, named as in the reference's execution plans
   (bench/tests/plans/snapshots/..Q5 (Execution Plan).snap:10-30). */
enum {
  RDFGPU_NODE_DATA_SOURCE = 1, /* DataSourceExec(MemQuadPatternDataSource), pattern_data_source.rs:21-58  */
  RDFGPU_NODE_FILTER = 2,      /* FilterExec: keep rows whose predicate is true; optional projection       */
  RDFGPU_NODE_HASH_JOIN = 3,   /* HashJoinExec(CollectLeft), inner | left, NullEqualsNothing, filter, projection */
  RDFGPU_NODE_CROSS_JOIN = 4,  /* CrossJoinExec                                                            */
  RDFGPU_NODE_NESTED_LOOP_JOIN = 5, /* NestedLoopJoinExec: inner | left with a filter and no equi keys     */
  RDFGPU_NODE_PROJECTION = 6,  /* ProjectionExec of plain columns                                          */
  RDFGPU_NODE_TABLE = 7,       /* bindings supplied by the caller (device columns), e.g. all-gathered rows */
  RDFGPU_NODE_TOPK = 8,        /* The operators directly above the path in the reference's explore plans (SURVEY §8f-3,
                                  ..Q5 (Execution Plan).snap:5-9): AggregateExec(gby = sort keys, first_value) = DISTINCT,
                                  then SortExec TopK(fetch = k), optionally per group (a batch of queries in one tree).
                                  left = input; n_keys sort keys (<= 4), all ascending, NULLS FIRST: left_keys[i] = column,
                                  right_keys[i] = RDFGPU_SORT_BY_ID (the UInt32 id itself, `product@1 ASC`),
                                  RDFGPU_SORT_BY_TERM (ENC_SORT of the term: defined here for columns of one kind among
                                  strings / IRIs / blank nodes, whose typed value carries the rank) or
                                  RDFGPU_SORT_BY_DOUBLE; table_cols = k; table_slot = 1 + group column (0 = one group).
                                  Rows equal on (group, keys) collapse to one.  The reference's AggregateExec groups by the
                                  sort expressions AND the raw columns (`gby=[ENC_SORT(..), product, productLabel]`), so two
                                  terms that sort alike stay two rows: every output column must be the group column or a key
                                  BY_ID — the host appends the remaining gby columns as trailing BY_ID keys (which also makes
                                  the order among ties of the declared ORDER BY deterministic).                          */
  RDFGPU_NODE_UNION = 9,       /* UnionExec: the rows of `left` followed by the rows of `right` (bag union; both inputs have
                                  the same columns) — SPARQL UNION as planned in BSBM Explore - Q4 / Q11 (Execution Plan).snap;
                                  optional projection */
  RDFGPU_NODE_CLOSURE = 10     /* KleenePlusClosureExec (lib/physical/src/paths/kleene_plus/physical.rs:94-157, 246-384): `left` yields
                                  the inner paths (graph, start, end) — graph 0 = default graph; the output is the SET of all paths
                                  of one or more inner paths chained end-to-start: within one graph (join_type = 0), or continuing
                                  through the inner paths of any graph while keeping the first path's graph (join_type = 1 =
                                  allow_cross_graph_paths).  A null start / end is an execution error, as in the reference. */
};
enum { RDFGPU_SORT_BY_ID = 0, RDFGPU_SORT_BY_TERM = 1,
       RDFGPU_SORT_BY_DOUBLE = 2 /* ENC_SORT of a numeric value: the sortable encoding orders numerics by Double::from(Numeric)
                                    (lib/encoding/src/sortable_term/builder.rs:36-39, lib/model/src/xsd/double.rs:92-102) in IEEE
                                    total order (-0 < +0, NaN last); unbound and non-numeric values sort first (the `xsd:double(...)`
                                    cast of BSBM explore Q10's ORDER BY yields an error = null for them).  Exact for Q10, whose
                                    prices are xsd:double literals; a decimal's cast goes through its lexical form in the
                                    reference and through the Decimal -> Double conversion here.                            */ };
enum { RDFGPU_JOIN_INNER = 0, RDFGPU_JOIN_LEFT = 1 };
#define RDFGPU_MAX_KEYS 4u
#define RDFGPU_MAX_COLUMNS 16u
#define RDFGPU_NO_PROJECTION 0xFFFFFFFFu

typedef struct rdfgpu_plan_node {
  uint32_t kind;                          /* RDFGPU_NODE_*                                */
  int32_t left;                           /* child node index (or -1)                     */
  int32_t right;                          /* second child (joins) or -1                   */
  uint32_t join_type;                     /* RDFGPU_JOIN_*                                */
  rdfgpu_scan_instruction scan[4];        /* DATA_SOURCE: instructions in G,S,P,O order   */
  uint32_t n_keys;                        /* HASH_JOIN: number of equi-key pairs          */
  uint32_t left_keys[RDFGPU_MAX_KEYS];    /* column indices in the left child             */
  uint32_t right_keys[RDFGPU_MAX_KEYS];   /* column indices in the right child            */
  uint32_t expr_off;                      /* FILTER / join filter: offset into exprs      */
  uint32_t expr_len;                      /* 0 = no filter                                */
  uint32_t proj_off;                      /* projection: offset into the u32 pool         */
  uint32_t n_proj;                        /* RDFGPU_NO_PROJECTION = keep all columns      */
  uint32_t table_slot;                    /* TABLE: which bound table (rdfgpu_plan_bind_table) */
  uint32_t table_cols;                    /* TABLE: number of columns                     */
} rdfgpu_plan_node;

typedef struct rdfgpu_regex {             /* one constant REGEX pattern of the plan (UTF-8, not NUL-terminated) */
  const char* pattern;
  const char* flags;                      /* SPARQL flags: any of s m i x q (regex.rs:107-141); may be NULL    */
  uint32_t pattern_len;
  uint32_t flags_len;
  uint32_t pattern_id;                    /* RDFGPU_EX_REGEX_VAR entries: the object id of this pattern literal; else 0 */
  uint32_t reserved;
} rdfgpu_regex;

typedef struct rdfgpu_plan_desc {
  const rdfgpu_plan_node* nodes;
  uint32_t n_nodes;
  uint32_t root;                          /* index of the root node                       */
  const rdfgpu_expr_node* exprs;
  uint32_t n_exprs;
  const uint32_t* pool;                   /* IN-set ids and projection lists              */
  uint32_t n_pool;
  uint32_t flags;                         /* RDFGPU_PLAN_*                                */
  const rdfgpu_regex* regexes;            /* patterns referenced by RDFGPU_EX_REGEX nodes */
  uint32_t n_regexes;
  uint32_t reserved;
} rdfgpu_plan_desc;
#define RDFGPU_PLAN_ALLOW_OPAQUE 1u

/* ------------------------------------------------------------------------------------ */
/* 4. Plans: compile, execute on device, stream result batches                           */
/*    replaces plan_extension (lib/storage/src/memory/planner.rs:31-64),                 */
/*    DataSource::open (pattern_data_source.rs:42-58), the stream's poll_next             */
/*    (stream.rs:39-63) and ExecutionPlan::execute of the join/filter subtree             */
/* ------------------------------------------------------------------------------------ */
typedef struct rdfgpu_plan rdfgpu_plan;

typedef struct rdfgpu_metrics {
  uint64_t output_rows;      /* BaselineMetrics::output_rows (stream.rs:48-63)            */
  uint64_t input_rows;       /* index rows covered by the located scan ranges             */
  uint64_t intermediate_rows;/* sum of rows produced by all non-root operators            */
  uint64_t device_bytes;     /* bytes of HBM held by this plan's intermediates            */
  double elapsed_compute_ms; /* device time of the last execute (HIP events)              */
  uint32_t kernels_launched;
  uint32_t host_syncs;
  /* ABI 3: why an execution was slow, when it was */
  uint32_t exact_reruns;     /* speculative sizes did not fit: the execution ran again with exact sizes (0 or 1)            */
  uint32_t device_mallocs;   /* hipMalloc calls the store's pools had to make during this execution (0 in steady state)     */
  double device_malloc_ms;   /* .. and the host time spent inside them                                                      */
  uint32_t tables_built;     /* join tables of store slices built (not found cached) during this execution                  */
  uint32_t reserved;
} rdfgpu_metrics;

/* Validates the description, chooses an index per data source (IndexPermutations::choose_index,
   permutations.rs:81-96) and allocates a HIP stream.  Does not launch kernels. */
int rdfgpu_plan_compile(rdfgpu_store* store, const rdfgpu_plan_desc* desc, rdfgpu_plan** out);
void rdfgpu_plan_destroy(rdfgpu_plan* plan);

/* Supplies the columns of a RDFGPU_NODE_TABLE (device pointers into this device's HBM,
   n rows each; must stay valid until the next execute finishes). */
int rdfgpu_plan_bind_table(rdfgpu_plan* plan, uint32_t slot, const uint32_t* const* cols,
                           uint32_t n_cols, uint64_t n_rows);

/* Runs the whole operator tree on the device.  The result stays in HBM.  Re-executable. */
int rdfgpu_plan_execute(rdfgpu_plan* plan);

/* Result shape (waits for the stream). */
int rdfgpu_plan_result_info(rdfgpu_plan* plan, uint64_t* n_rows, uint32_t* n_cols);
/* Device pointers of the result columns (valid until the next execute / destroy of THIS plan — also across mutations of the
   store: an executed plan keeps the store generation its result may point into, and keeps showing the pre-mutation rows,
   like the reference's plan keeps its snapshot, snapshot.rs:35-37). */
int rdfgpu_plan_result_device(rdfgpu_plan* plan, const uint32_t** cols, uint32_t cap_cols);
/* Copies the whole result to caller-owned host columns (each with room for n_rows). */
int rdfgpu_plan_fetch(rdfgpu_plan* plan, uint32_t* const* host_cols, uint32_t n_cols);
/*
 * SendableRecordBatchStream::poll_next (stream.rs:39-63): exports the next batch of at
 * most batch_size rows as an Arrow struct array of UInt32 children (format "+s" / "I");
 * id 0 becomes a null.  Never yields an empty batch (scan.rs:195-198).  Returns
 * RDFGPU_END once drained.  `schema` may be NULL.
 */
int rdfgpu_plan_next(rdfgpu_plan* plan, struct ArrowArray* out, struct ArrowSchema* schema);
/* Restarts the batch stream over the current result. */
int rdfgpu_plan_rewind(rdfgpu_plan* plan);
int rdfgpu_plan_metrics(rdfgpu_plan* plan, rdfgpu_metrics* out);
/* Index chosen for a DATA_SOURCE node (RDFGPU_GSPO..), for plan display ("[GPOS] subject=…"). */
int rdfgpu_plan_selected_index(const rdfgpu_plan* plan, uint32_t node, uint32_t* components);
/* Opaque hipStream_t of the plan, so the host can order its own work after it. */
int rdfgpu_plan_stream(rdfgpu_plan* plan, void** hip_stream);
/*
 * ENC_PT of a result column (MemObjectIdMapping::decode_array, lib/storage/src/memory/object_id_mapping.rs:331-374): object
 * ids -> plain terms, decoded on the device from the store's string heap (rdfgpu_store_set_strings must hold the lexical
 * form of EVERY id that can appear) and typed-value table.  Rows [first_row, first_row + n_rows) of column `col`, as an
 * Arrow struct array
 *     struct<term_type: uint8, value: utf8, tag: uint8, aux: uint32>
 * term_type = PlainTermType (plain_term/encoding.rs:90-127: 0 named node, 1 blank node, 2 literal); value = the lexical
 * form; a null struct where the id is 0 / unknown (builder.append_null()).  data_type and language_tag of the reference's
 * struct are functions of (tag, aux) through the host's small tables — tag = the datatype of a typed literal (the ABI's
 * RDFGPU_TV_*; RDFGPU_TV_OTHER: aux = the host's datatype id), aux = the language id of a string (0 = none) — and are
 * attached there as dictionary arrays: the per-row string work, the gather of n_rows lexical forms out of the heap,
 * happens here.  The lexical forms of one call may not exceed 2^31 - 1 bytes (utf8 offsets are int32): decode fewer rows.
 */
int rdfgpu_plan_decode_terms(rdfgpu_plan* plan, uint32_t col, uint64_t first_row, uint64_t n_rows,
                             struct ArrowArray* out, struct ArrowSchema* schema);

/*
 * Filter push-down into a DataSourceExec leaf.  When a subtree is NOT fused into one rdfgpu plan, DataFusion's physical
 * filter push-down offers the leaf the filters above it (MemQuadPatternDataSource::try_pushdown_filters,
 * lib/storage/src/memory/storage/pattern_data_source.rs:107-151): id-level comparisons of a bound variable with an object
 * id, their conjunctions on one column, literal `true`, and DynamicFilterPhysicalExprs (predicate_pushdown.rs:62-157,
 * 161-249).  A filter here is the MemStoragePredicateExpr the host already extracted (MemStoragePredicateExpr::try_from).
 */
enum { RDFGPU_PUSH_UNSUPPORTED = 0,  /* try_from returned None: answered PushedDown::No, nothing changes                  */
       RDFGPU_PUSH_TRUE = 1,         /* literal true                                                                       */
       RDFGPU_PUSH_BINARY = 2,       /* column <op> object id, op = RDFGPU_OP_*                                             */
       RDFGPU_PUSH_BETWEEN = 3 };    /* from <= column <= to (what `>` AND `<=` on one column become, :189-249)             */
typedef struct rdfgpu_pushdown_filter {
  uint32_t kind;      /* RDFGPU_PUSH_*                                                        */
  uint32_t var;       /* the variable slot of the column (rdfgpu_scan_instruction.var)        */
  uint32_t op;        /* BINARY: RDFGPU_OP_EQ / GT / GTEQ / LT / LTEQ                         */
  uint32_t value;     /* BINARY: the object id                                                */
  uint32_t from, to;  /* BETWEEN, inclusive                                                   */
} rdfgpu_pushdown_filter;
/*
 * try_pushdown_filters on DATA_SOURCE node `node` of a compiled plan: pushed[i] receives 1 (PushedDown::Yes) or 0.  Every
 * supported filter is AND-ed into the instruction that binds its variable (MemIndexScanInstructions::apply_filter,
 * scan_instructions.rs:101-133, through try_and_with :170-210) and the index is chosen again (apply_pushdown_filters ->
 * try_find_better_index, pattern_data_source.rs:155-165, scan.rs:469-486).  The change is permanent for this plan.
 * Errors like the reference: a filter on a variable the pattern does not bind, or one that cannot be combined
 * (EqualTo), is RDFGPU_ERR_INVALID.
 */
int rdfgpu_plan_pushdown_filters(rdfgpu_plan* plan, uint32_t node, const rdfgpu_pushdown_filter* filters, uint32_t n, uint8_t* pushed);
/*
 * The CURRENT predicates of the leaf's dynamic filters — a HashJoinExec publishes its build side's key bounds when the
 * build finishes; the scan reads them on its first next() (combine_instructions_with_dynamic_filters, scan.rs:241-261) and
 * may switch index for them (collect_relevant_row_groups, scan.rs:217-239).  Applied on top of the static instructions by
 * every execute until replaced; n = 0 clears.  Only BINARY / BETWEEN / TRUE kinds.
 */
int rdfgpu_plan_set_dynamic_filters(rdfgpu_plan* plan, uint32_t node, const rdfgpu_pushdown_filter* filters, uint32_t n);
/*
 * One level of the leaf as it will be scanned (after push-down / with the current dynamic filters): level 0..3 in G,S,P,O
 * order; `*pred` as in rdfgpu_predicate (IN sets with one id: from = to = the id, n_ids = 1; larger sets: n_ids only).
 * For plan display: "[GPOS] subject=?s, predicate=<p>, object=?o" + "object in (2..9)" (pattern_data_source.rs:192-234).
 */
struct rdfgpu_predicate;
int rdfgpu_plan_source_predicate(const rdfgpu_plan* plan, uint32_t node, uint32_t level, struct rdfgpu_predicate* pred);

/*
 * Per-kernel device timing of the last execute (HIP events on the plan's stream, around every
 * launch) with the algorithmic bytes each kernel class had to move (formulas: DESIGN.md,
 * from SURVEY.md §8d).  The equivalent of DataFusion's per-operator BaselineMetrics
 * (elapsed_compute, stream.rs:48-63), at kernel granularity.  Off by default.
 */
typedef struct rdfgpu_kernel_stat {
  const char* kernel;        /* device function name as rocprofv3 prints it (prefix match)  */
  uint32_t launches;
  uint32_t reserved;
  double total_ms;           /* sum of HIP-event durations of those launches                */
  uint64_t algorithmic_bytes;/* sum over those launches                                     */
  uint64_t rows_in;          /* rows streamed by those launches                             */
} rdfgpu_kernel_stat;
/* on = 1: every launch of the plan's later executions is bracketed with HIP events (two events per launch: ~5 us of stream time
 * each, a tenth of a short step); on = 2: only the launches of the kernel that took longest in the last execution timed with
 * on = 1 — the one a roofline is quoted for — so that a timed region is not slowed down by its own instrumentation; 0: off. */
int rdfgpu_plan_enable_kernel_timing(rdfgpu_plan* plan, int on);
int rdfgpu_plan_kernel_stats(rdfgpu_plan* plan, rdfgpu_kernel_stat* out, uint32_t cap, uint32_t* n);

/* ------------------------------------------------------------------------------------ */
/* 4b. Engine options                                                                    */
/*     the session-level knobs of this path: the counterpart of DataFusion's SessionConfig */
/*     / ConfigOptions the reference threads through planning (lib/rdf-fusion/src/store.rs */
/*     :101-103, bench/src/environment.rs:71-87).  Every physical rewrite / table form /   */
/*     speculation mode can be switched off; results never depend on them (tested).       */
/* ------------------------------------------------------------------------------------ */
/*
 * Defaults are taken ONCE per process (first store creation) from the environment variables RDFGPU_<NAME>
 * (e.g. RDFGPU_NO_CHAIN_FUSION=1); after that the environment is never consulted again.  A store copies the
 * process defaults when it is created, a plan copies its store's options when it is compiled; the two setters
 * change one store (plans compiled later) or one plan (its later executions).  Nothing on the execute path reads
 * the environment.
 */
enum {
  RDFGPU_OPT_FORCE_GENERIC_VM = 0,      /* every FILTER / join filter through the stack VM (no specialised kernels)   */
  RDFGPU_OPT_NO_JOIN_REORDER,           /* keep (A x B) JOIN C as written                                             */
  RDFGPU_OPT_NO_SPECULATION,            /* size every operator exactly (one host sync per join)                       */
  RDFGPU_OPT_NO_FIRST_RUN_SPECULATION,  /* speculate only from a previous execution's cardinalities                   */
  RDFGPU_OPT_NO_STRING_VERDICTS,        /* REGEX / CONTAINS / .. per row instead of per distinct term                 */
  RDFGPU_OPT_NO_TABLE_CACHE,            /* join tables are built inside every execution, like HashJoinExec(CollectLeft)
                                           does per query; nothing is kept on the store between executions            */
  RDFGPU_OPT_NO_INDEX_JOIN,             /* never build on a store slice just because it is one                        */
  RDFGPU_OPT_NO_CHAIN_FUSION,           /* run follow-up look-up joins as separate operators                          */
  RDFGPU_OPT_NO_VALUE_TABLES,           /* no decoded integer tables next to direct tables                            */
  RDFGPU_OPT_NO_RANGE_INDEX,            /* no value-ordered CSR groups                                                */
  RDFGPU_OPT_NO_FILTER_FUSION,          /* materialise FilterExec children of joins                                   */
  RDFGPU_OPT_NO_LDS_JOIN,               /* chained HBM hash join (count / scan / write) for every HashJoinExec         */
  RDFGPU_OPT_NO_GLOBAL_TABLE_JOIN,      /* the fused join kernel only for builds that fit LDS                         */
  RDFGPU_OPT_NO_DIRECT_TABLE,           /* no direct-address / CSR tables for dense keys                              */
  RDFGPU_OPT_NO_BAND_JOIN,              /* no key-partitioned band join for fused look-up chains over small groups    */
  RDFGPU_OPT_NO_PARTITIONED_JOIN,       /* no radix-partitioned LDS hash join for large non-cached build sides        */
  RDFGPU_OPT_NO_VALUE_VERDICTS,         /* typed comparison per row instead of per distinct term of a sorted slice     */
  RDFGPU_OPT_NO_PRIMING,                /* no priming run over the first rows of big bound tables before a plan's first execution */
  RDFGPU_OPT_NO_ORDERED_JOIN,           /* no slice-ordered emission of a table x slice join / no sort-free band join on it */
  RDFGPU_OPT_NO_BAND_PACK16,            /* band join: both windows always tested with 32-bit arithmetic                    */
  RDFGPU_OPT_NO_RUN_COPY,               /* .. per distinct term, but rows streamed (verdict bits) instead of runs copied */
  RDFGPU_OPT_NO_RANGE_PARTITION,        /* partitioned join: hash-partition BOTH sides even when the probe side is a slice sorted by a join key */
  RDFGPU_OPT_LDS_MAX_BUILD,             /* value: largest build side (rows) joined through a per-workgroup LDS table  */
  RDFGPU_OPT_CSR_ROW_LANES_LOG2,        /* value + 1: lanes sharing one probe row of a CSR join (0 = automatic)        */
  RDFGPU_OPT_JOIN_WAVE_Q,               /* value: entries of a wave's candidate queue (0 = automatic)                  */
  RDFGPU_OPT_PARTITION_MIN_BUILD,       /* value: smallest non-cached build side (rows) that is radix-partitioned (default 2^21) */
  RDFGPU_OPT_PARTITION_TWO_PASS_ROWS,   /* value: expected output rows from which a partitioned join counts before it writes (default 50 M) */
  RDFGPU_OPT_NO_OWN_PARTITION_PASS,     /* flag: partitioned join: partition ids materialised and sorted with rocPRIM's radix sort instead of the hand-written passes */
  RDFGPU_OPT_NO_BAND_COMPACT,           /* flag: band join fed by an ordered slice join: 32-byte {record, aux} row records even where 16 bytes would do */
  RDFGPU_OPT_NO_PROBE_OUTER_JOIN,       /* flag: LEFT joins always build on their left input (no probe-preserving form over the right input's slice table) */
  RDFGPU_OPT_NO_STREAM_JOIN,            /* flag: joins against a direct-address table always take the generic queueing kernel (no register-resident streaming form) */
  RDFGPU_OPT_PARTITION_ROWS,            /* value: build rows per partition a partitioned join aims for (0 = automatic: 1024)   */
  RDFGPU_OPT_PARTITION_SLOTS,           /* value: slots of a partition's LDS table, a power of two from 1024 to 8192 (0 = automatic: 4096); a partition with more than slots / 2 build rows is joined chunk by chunk */
  RDFGPU_OPT__COUNT
};
int rdfgpu_store_set_option(rdfgpu_store* store, uint32_t option, uint64_t value);
int rdfgpu_store_get_option(const rdfgpu_store* store, uint32_t option, uint64_t* value);
int rdfgpu_plan_set_option(rdfgpu_plan* plan, uint32_t option, uint64_t value);
/* "NO_CHAIN_FUSION" for RDFGPU_OPT_NO_CHAIN_FUSION (the environment variable is RDFGPU_ + this); NULL if out of range. */
const char* rdfgpu_option_name(uint32_t option);

/* ------------------------------------------------------------------------------------ */
/* 5. Host logic of the scan planner (no device access)                                  */
/* ------------------------------------------------------------------------------------ */
/*
 * MemQuadIndex::compute_scan_score (quad_index.rs:100-130): `instr` is in the order of the
 * index being scored.
 */
uint64_t rdfgpu_scan_score(const rdfgpu_scan_instruction instr[4]);
/*
 * IndexPermutations::choose_index (permutations.rs:81-96): `gspo` is in G,S,P,O order;
 * `available` is a bit mask of RDFGPU_GSPO.. permutations the store keeps; returns the
 * chosen permutation (ties prefer the first listed).
 */
uint32_t rdfgpu_choose_index(const rdfgpu_scan_instruction gspo[4], uint32_t available);
/*
 * MemIndexScanPredicate::try_and_with (scan_instructions.rs:170-210).  Predicates are given
 * as (pred, a, b) triples where IN sets are explicit arrays.  Writes the combined predicate;
 * `out_ids` needs room for min(na, nb) ids.  Returns 1 if combinable, 0 if not (EqualTo).
 */
typedef struct rdfgpu_predicate {
  uint32_t pred;        /* RDFGPU_PRED_*                                                  */
  uint32_t from, to;    /* BETWEEN                                                        */
  const uint32_t* ids;  /* IN (sorted unique)                                             */
  uint32_t n_ids;
  uint32_t equal_to;    /* EQUAL_TO variable slot                                         */
} rdfgpu_predicate;
int rdfgpu_predicate_and(const rdfgpu_predicate* lhs, const rdfgpu_predicate* rhs,
                         rdfgpu_predicate* out, uint32_t* out_ids);
/*
 * MemStoragePredicateExpr::to_scan_predicate (predicate_pushdown.rs:120-157): rewrites
 * `column <op> value` into a scan predicate.  op: 0 Eq, 1 Gt, 2 GtEq, 3 Lt, 4 LtEq.
 * Returns 1 and fills `out` (BETWEEN / IN with one id in out->from / FALSE).
 */
enum { RDFGPU_OP_EQ = 0, RDFGPU_OP_GT = 1, RDFGPU_OP_GTEQ = 2, RDFGPU_OP_LT = 3, RDFGPU_OP_LTEQ = 4 };
int rdfgpu_pushdown_to_scan_predicate(uint32_t op, uint32_t value, rdfgpu_predicate* out);

/*
 * compile_pattern (scalar/strings/regex.rs:107-141) for the device: checks that a REGEX pattern / flags pair is inside
 * the supported subset (csrc/regex_compile.hpp) without touching a device.  Returns RDFGPU_OK and the number of
 * automaton positions in *positions (0 for a pattern that can only yield the error value: an invalid flag), or
 * RDFGPU_ERR_UNSUPPORTED with the reason in rdfgpu_last_error() — exactly what rdfgpu_plan_compile would answer.
 */
int rdfgpu_regex_check(const char* pattern, uint32_t pattern_len, const char* flags, uint32_t flags_len, uint32_t* positions);

/* ------------------------------------------------------------------------------------ */
/* 6. Multi-GPU exchange steps (one process per GPU; no reference counterpart) */
/* ------------------------------------------------------------------------------------ */
/*
 * Triples are sharded by rdfgpu_shard_of(subject) over the ranks of a communicator; object ids are global and the typed-value
 * table is replicated.  Joins on the shard key are local; the other two cases are one exchange step each, on binding tables
 * that live in HBM (e.g. rdfgpu_plan_result_device of one plan -> rdfgpu_plan_bind_table of the next):
 *   rdfgpu_exchange_allgatherv   every rank contributes its rows, every rank receives all ranks' rows (rank order)
 *   rdfgpu_exchange_repartition  row i goes to rank rdfgpu_shard_of(cols[key_col][i]): re-shards a table by the key of the
 *                                next join.  STABLE: a rank receives, source rank after source rank, that rank's rows for it
 *                                in their original order (a table sorted by key_col arrives as `world` sorted runs; the
 *                                received bytes do not depend on scheduling)
 * Row counts travel first, receive buffers are sized from them (nothing is padded or clipped).  Both calls are collective
 * (every rank of the communicator calls them in the same order) and return when the received columns are complete; the
 * columns belong to the communicator and stay valid until its next exchange.
 * Transport: RCCL over xGMI (grouped ncclSend / ncclRecv, one pair per peer; the library dlopens librccl.so.1), or — for
 * several ranks on one GPU and for tests — staging through host memory with the wire supplied by the caller.
 */
typedef struct rdfgpu_comm rdfgpu_comm;
#define RDFGPU_COMM_ID_BYTES 128
/* ncclGetUniqueId: called on one rank, the 128 bytes are handed to the others by the launcher (torchrun's store, MPI, a file). */
int rdfgpu_comm_unique_id(uint8_t id[RDFGPU_COMM_ID_BYTES]);
/* ncclCommInitRank on `device` (-1 = current). */
int rdfgpu_comm_create(const uint8_t id[RDFGPU_COMM_ID_BYTES], uint32_t rank, uint32_t world, int32_t device, rdfgpu_comm** out);
/* The caller's wire: an all-to-all of byte blocks in host memory — block p of `send` (send_bytes[p] bytes, blocks back to back)
   goes to rank p, block p of `recv` (recv_bytes[p] bytes) comes from rank p; returns 0 on success. */
typedef int (*rdfgpu_host_alltoallv_fn)(void* ctx, const void* send, const uint64_t* send_bytes, void* recv, const uint64_t* recv_bytes);
int rdfgpu_comm_create_host(uint32_t rank, uint32_t world, int32_t device, rdfgpu_host_alltoallv_fn fn, void* ctx, rdfgpu_comm** out);
void rdfgpu_comm_destroy(rdfgpu_comm* comm);
int rdfgpu_exchange_allgatherv(rdfgpu_comm* comm, const uint32_t* const* cols, uint32_t n_cols, uint64_t n_rows,
                               const uint32_t** out_cols, uint64_t* out_rows);
int rdfgpu_exchange_repartition(rdfgpu_comm* comm, const uint32_t* const* cols, uint32_t n_cols, uint64_t n_rows, uint32_t key_col,
                                const uint32_t** out_cols, uint64_t* out_rows);
/* The shard of an object id among `world` ranks (host logic, no device access): the function triples are sharded by. */
uint32_t rdfgpu_shard_of(uint32_t id, uint32_t world);

/* ------------------------------------------------------------------------------------ */
/* 7. Bulk load, first half: N-Triples text -> object ids on the device                   */
/* ------------------------------------------------------------------------------------ */
/*
 * The reference parses on the host and interns the three terms of every quad through a DashMap (Store::load_from_reader /
 * bulk_loader, lib/rdf-fusion/src/store.rs:477-493 -> MemObjectIdMapping::encode_quad, lib/storage/src/memory/
 * object_id_mapping.rs:106-116).  rdfgpu_ntriples_parse does the per-triple half on the device: line and term splitting,
 * one id per DISTINCT term (first_id .. first_id + n_terms - 1; a bijection, not the reference's insertion order — no query
 * can observe the difference), and the s / p / o id columns in HBM, ready for rdfgpu_store_extend_device (graph column =
 * zeros = the default graph).  The distinct terms come back as written in the file (`<iri>`, `_:b1`, `"lex"`, `"lex"@en`,
 * `"lex"^^<dt>`: rdfgpu_ntriples_terms) and decoded + typed (rdfgpu_ntriples_decoded): the host builds its dictionary from them —
 * per distinct term, not per triple.  Text: UTF-8, one triple per line, blank lines and `#`
 * comment lines allowed.  A malformed line is RDFGPU_ERR_INVALID with its number; two different terms with one 64-bit
 * hash (probability ~ n_terms^2 / 2^65) is RDFGPU_ERR_UNSUPPORTED — loud, never a wrong id.
 */
typedef struct rdfgpu_ntriples rdfgpu_ntriples;
int rdfgpu_ntriples_parse(int32_t device, const char* text, uint64_t text_bytes, uint32_t first_id, rdfgpu_ntriples** out);
int rdfgpu_ntriples_info(const rdfgpu_ntriples* nt, uint64_t* n_triples, uint32_t* n_terms, uint64_t* term_bytes);
/* offsets[n_terms + 1] and the terms' bytes (term t has id first_id + t); either pointer may be null */
int rdfgpu_ntriples_terms(const rdfgpu_ntriples* nt, uint64_t* offsets, uint8_t* bytes);
/*
 * The distinct terms as the reference's parser hands them to its dictionary (oxttl -> oxrdf: escapes decoded) and the typed values
 * of the literals (ABI 3).  Terms are interned by this CANONICAL form: `"a\u0041"` and `"aA"`, `"x"^^xsd:string` and `"x"`, `"v"@EN`
 * and `"v"@en` are one term with one id (rdfgpu_ntriples_terms hands out one of its spellings).  Per term t (id first_id + t):
 *   kind[t]    1 IRI, 2 blank node, 3 simple literal, 4 language-tagged literal, 5 typed literal
 *   lex        the lexical form, ECHAR / UCHAR escapes decoded to UTF-8: lex_bytes[lex_off[t] .. lex_off[t + 1])
 *   suffix     the language tag in lower case / the datatype IRI (decoded): suffix_bytes[suffix_off[t] .. suffix_off[t + 1])
 *   typed[t]   the typed-value row the device derives (encoding/typed_value.rs:27-83, lib/model/src/typed_value.rs:349-411): tag;
 *              IRIs / blank nodes / strings carry lo = 0 (their rank in `str` order is the dictionary's to give) and aux = 0 (the
 *              host numbers languages / datatypes); xsd:integer and its derived types, xsd:int, xsd:boolean, xsd:decimal (low
 *              64 bits in lo, high in dec_hi[t]: the host's decimal side table) are parsed here, an invalid lexical form gives
 *              RDFGPU_TV_NULL like the reference's Invalid; xsd:double / xsd:float are parsed when the conversion is exact in one
 *              IEEE operation (<= 15 / 7 digits, |exponent| <= 22 / 10), otherwise — and for dateTime / time / date / durations —
 *              flags carries RDFGPU_TVF_NEEDS_HOST and the tag says what to parse the lexical form as.
 * Any pointer may be null.  Sizes: rdfgpu_ntriples_decoded_info.
 */
int rdfgpu_ntriples_decoded_info(const rdfgpu_ntriples* nt, uint64_t* lex_bytes, uint64_t* suffix_bytes);
int rdfgpu_ntriples_decoded(const rdfgpu_ntriples* nt, uint8_t* kind, uint64_t* lex_off, uint8_t* lex_bytes, uint64_t* suffix_off,
                            uint8_t* suffix_bytes, rdfgpu_typed_value* typed, int64_t* dec_hi);
/* device pointers of the id columns (n_triples each, file order); valid until rdfgpu_ntriples_destroy */
int rdfgpu_ntriples_columns(const rdfgpu_ntriples* nt, const uint32_t** s, const uint32_t** p, const uint32_t** o);
void rdfgpu_ntriples_destroy(rdfgpu_ntriples* nt);

#ifdef __cplusplus
}
#endif
#endif /* RDFGPU_H */
