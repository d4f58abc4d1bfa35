// kernels.hpp — host-callable launchers of the gfx950 kernels (definitions in kernels.hip).
#pragma once
#include <functional>
#include "common.hpp"
#include "expr_device.hpp"

namespace rdfgpu {

constexpr u32 kNil = 0xFFFFFFFFu;
constexpr int kMaxCols = RDFGPU_MAX_COLUMNS;

// A binding table resident in HBM: u32 columns, 0 = null.  `cap` is an upper bound of the row
// count known on the host; when `n_dev` is non-null the exact count lives on the device (the host
// never waits for it unless it must size an allocation).
struct DevTable {
  u32 n_cols = 0;
  const u32* cols[kMaxCols] = {};
  u64 cap = 0;
  const u64* n_dev = nullptr;
  u64 stable_id = 0;   // non-zero: a pure slice of the store (same rows on every execution until the store changes)
  int sorted_col = -1; // a zero-copy slice: the column that is sorted within it (the first level below the pinned prefix)
  u32 key_min = 0, key_max = 0;   // .. and its first / last id
};

// ---- K1: range locate on a sorted permutation (prune_relevant_row_groups, quad_index_data.rs:155-284) ----
struct LocateJob {
  const u32* col[4];   // index-order columns
  u64 n;               // index length
  u32 n_levels;        // leading levels with a pruning predicate
  u32 from[4], to[4];  // inclusive id ranges per level (EqualTo => from == to)
};
// per job: lo, hi, then the level that is sorted within [lo, hi) (the first one not pinned to a single id; 4 = none) in the
// upper half and its first id in the lower half, then its last id
constexpr u32 kLocateWords = 4;
void launch_locate(const LocateJob* jobs_dev, u32 n_jobs, u64* lo_hi_dev /* kLocateWords per job */, hipStream_t s);

// ---- K2: ordered scan + residual predicates + compaction (scan.rs:264-340) ----
struct ScanLevelPred {
  u32 kind;        // RDFGPU_PRED_*
  u32 a, b;        // BETWEEN: from,to ; IN: n ids in `ids` ; EQUAL_TO: a = other level
  const u32* ids;  // IN set (device)
};
struct ScanJob {
  const u32* col[4];   // index-order columns, already offset to the located range start
  u64 n;               // rows in the located range
  ScanLevelPred pred[4];
  u32 n_out;           // bound variables
  u32 out_level[4];    // index level feeding output column k
  u32* out[4];
};
constexpr u32 kScanTile = 1024;  // rows per workgroup
void launch_scan_count(const ScanJob& job, u32* block_counts, hipStream_t s);
void launch_scan_write(const ScanJob& job, const u32* block_offsets, hipStream_t s);

// ---- K3: FilterExec — fused per-row predicate + projection + block-aggregated compaction ----
struct FilterArgs {
  const u32* in[kMaxCols];
  u32* out[kMaxCols];
  u32 n_in_cols, n_out_cols;
  u32 proj[kMaxCols];
  const u64* n_in_dev; u64 n_in_cap;
  u64* n_out_dev;       // zeroed before launch
  u32 iters, head_skip, vec_ok, vec_proj_ok;   // set by launch_filter: rounds per workgroup, alignment head, 16-B loads allowed
  TypedTable tt;
  ExprProgram prog;
  const unsigned char* verdict; u64 n_verdict;   // shape 3: per-id verdicts of a string predicate (0 false / 1 true / 2 error)
  const u32* value_bits; u32 value_min; u64 value_span;   // shape 4: one verdict BIT per id of [value_min, value_min + value_span) for a shape-2 predicate
  // streaming form (filter_streams): one verdict bit per row, tile counts (+ 1, the extra one zeroed), their scan, scan temp
  unsigned short* stream_bits; u32* stream_counts; u32* stream_offs; void* stream_temp; size_t stream_temp_bytes;
};
// shape 0 = generic VM; 1 = `col <op> object-id literal` (ID_EQ / ID_NEQ); 2 = EBV(cmp(ENC_TV(col), literal));
// 3 = EBV(REGEX | CONTAINS | STRSTARTS | STRENDS (ENC_TV(col), constant)) through a per-distinct-term verdict table
void launch_filter(const FilterArgs& a, int shape, hipStream_t s);
bool filter_streams(const FilterArgs& a, int shape);
void launch_filter_bits(const FilterArgs& a, int shape, hipStream_t s);    // pass 1: verdict bits + tile counts
void launch_filter_write(const FilterArgs& a, int shape, hipStream_t s);   // pass 2 (after the scan): ordered write
u64 filter_stream_tiles(const FilterArgs& a);           // 4096-row tiles of the streaming form   // launch_filter will take the streaming kernel (filter_stream_kernel) for these arguments
// Evaluates one string predicate for EVERY object id once (streaming through the string heap): out[id] = 0 / 1 / 2.
// FILTER as a copy of qualifying runs (a typed comparison on the sorted column of a slice with few ids, kernels.hip):
// runs per id, compaction + scan by one workgroup, copy.  a.value_min / value_span = the id range; at most 2 output columns.
constexpr u64 kRunCopyMaxIds = 65536;
struct RunCopyBuffers { u32* run_lo; u32* run_cnt; u32* c_lo; u32* c_off; u32* c_val; u32* n_runs; };
void launch_value_runs(const FilterArgs& a, const RunCopyBuffers& b, hipStream_t s);
void launch_run_scan(const FilterArgs& a, const RunCopyBuffers& b, bool verdicts_here, hipStream_t s);   // verdicts_here: b.run_lo is the slice's cached table, the comparison per id is answered in the scan kernel
void launch_run_copy(const FilterArgs& a, const RunCopyBuffers& b, hipStream_t s);
// shape 2's predicate once per id of [a.value_min, + a.value_span): bit (id - value_min) of `a.value_bits` (64-id words)
void launch_value_verdicts(const FilterArgs& a, hipStream_t s);
// ---- ENC_PT: ids -> (term type, tag, aux, length of the lexical form), then the bytes (object_id_mapping.rs:331-374) ----
struct DecodeArgs {
  const u32* ids; u64 n; TypedTable tt;
  u32* len; unsigned char* term_type; unsigned char* tag; u32* aux;   // per row (term_type 0xFF = null)
  const u32* off; unsigned char* bytes;                               // exclusive scan of len; the packed lexical forms
};
void launch_decode_lengths(const DecodeArgs& a, hipStream_t s);
void launch_decode_bytes(const DecodeArgs& a, hipStream_t s);
void launch_regex_verdicts(const RegexProg* prog_dev, const TypedTable& tt, int64_t rhs_lang, unsigned char* out, u64 n_ids, hipStream_t s);

// ---- K6: CrossJoinExec ----
struct CrossArgs {
  const u32* left[kMaxCols]; const u32* right[kMaxCols];
  u32* out[kMaxCols];
  u32 n_left_cols, n_right_cols, n_out_cols;
  u32 proj[kMaxCols];
  const u64* n_left_dev; u64 n_left_cap;
  const u64* n_right_dev; u64 n_right_cap;
  u64* n_out_dev;
};
void launch_cross(const CrossArgs& a, hipStream_t s);

// ---- UnionExec: left rows then right rows (either side's row count may live on the device) ----
struct UnionArgs {
  const u32* left[kMaxCols]; const u32* right[kMaxCols];
  u32* out[kMaxCols];
  u32 n_cols;
  const u64* n_left_dev; u64 n_left_cap;
  const u64* n_right_dev; u64 n_right_cap;
  u64* n_out_dev;
};
void launch_union(const UnionArgs& a, hipStream_t s);

// ---- KleenePlusClosureExec (closure.hip): transitive closure of (graph, start, end) paths, semi-naive on sorted u64 keys ----
struct ClosureStats { u32 iterations = 0; u64 nodes = 0, graphs = 0, initial_paths = 0; };
u64 closure_exec(const u32* g, const u32* s, const u32* e, u64 n, bool cross_graph, hipStream_t stream,
                 const std::function<u32*(u64)>& alloc, u32* out_cols[3], ClosureStats* stats);

// ---- K4/K5: HashJoinExec(CollectLeft) — chained table in HBM (v1) ----
struct JoinArgs {
  const u32* left[kMaxCols]; const u32* right[kMaxCols];
  u32* out[kMaxCols];
  u32 n_left_cols, n_right_cols, n_out_cols;
  u32 proj[kMaxCols];
  u32 n_keys; u32 left_keys[RDFGPU_MAX_KEYS]; u32 right_keys[RDFGPU_MAX_KEYS];
  const u64* n_left_dev; u64 n_left_cap;
  const u64* n_right_dev; u64 n_right_cap;
  u32* heads; u32 bucket_mask;  // heads[bucket_mask + 1], kNil-filled
  u32* next;                    // next[n_left_cap]
  u32* counts;                  // per probe row match count / inclusive offsets [n_right_cap]
  u8* visited;                  // left join: per build row
  u64* n_out_dev;               // left join: final count
  u64 matched_total;            // left-join tail: capacity of the out columns (0 = unchecked)
  u32 has_filter;
  TypedTable tt;
  ExprProgram prog;
};
void launch_join_build(const JoinArgs& a, hipStream_t s);
void launch_join_count(const JoinArgs& a, hipStream_t s);   // fills counts[j]
void launch_join_write(const JoinArgs& a, hipStream_t s);   // counts[] holds INCLUSIVE offsets
void launch_join_left_unmatched(const JoinArgs& a, hipStream_t s);
// NestedLoopJoinExec (no equi keys): same args, heads/next unused
void launch_nlj_count(const JoinArgs& a, hipStream_t s);
void launch_nlj_write(const JoinArgs& a, hipStream_t s);

// Parameters of the specialised predicate shapes.
struct TvLiteral { int64_t lo, hi; u32 aux; u8 tag, flags, arith_sub, cmp_op; };
struct WindowFilter {   // EBV(cmp0(ENC_TV(x0), y0 +/- lit0)) AND EBV(cmp1(ENC_TV(x1), y1 +/- lit1))
  u32 x0, y0, x1, y1;   // column indices in [left cols, right cols]
  TvLiteral l0, l1;
};
struct IdFilter { u32 col, lit, is_eq; };   // col <ID_EQ | ID_NEQ> object-id literal
struct IdPairFilter { u32 a, b, is_eq; };   // col a <ID_EQ | ID_NEQ> col b

// A chain of follow-up joins fused into a join's resolve phase: every stage is an inner single-key lookup of a
// key column of the BASE join in a direct-address table of a store slice (unique dense key), plus the stage's join
// filter.  Columns are addressed by (source row, pointer): nothing between the base join and the last stage is
// materialised — late materialisation by row id.
constexpr int kMaxChain = 3;
struct ColRef { const u32* ptr; u32 src; u32 pad; };   // src: 0 = base probe row, 1 = base build row, 2 + t = row found by stage t
struct ChainStage {
  ColRef key;                 // src 0 / 1 only
  const u32* direct; u32 kmin, kn;
  u32 fs;                     // 0 none / 2 `col <=|!=> col` / 3 numeric window
  ColRef f[4];                // id pair: a, b ; window: x0, y0, x1, y1
  TvLiteral l0, l1; u32 is_eq, pad;
  // window stages whose x operand is a column of the stage's own slice with all-integer values: the decoded value
  // by key, val[key - kmin] (INT64_MIN = no row) — replaces the lookup -> column gather -> typed-value gather chain
  const long long* val;
};

// ---- K4+K5 fused: LDS-staged hash join for build sides that fit one workgroup's LDS ----
constexpr u32 kLdsJoinMaxBuild = 8192;   // rows; 16384 slots x 8 B = 128 KiB of the CU's 160 KiB
constexpr u32 kOuterNull = 0xFFFFFFFEu;   // build row of a probe row's "no match" candidate (LdsJoinArgs::probe_outer)
struct LdsJoinArgs {
  const u32* cols[2 * kMaxCols];   // the operator's [left cols, right cols] schema, one pointer per column
  u32 n_left_cols;
  u32 build_is_left;               // 1: build = left input; 0: the engine swapped sides (inner joins only)
  u32* out[kMaxCols];
  u32 n_out_cols;
  u32 proj[kMaxCols];              // into [left cols, right cols]
  u32 n_keys;
  const u32* build_key[RDFGPU_MAX_KEYS];   // key columns, resolved to pointers on the host
  const u32* probe_key[RDFGPU_MAX_KEYS];
  const u64* n_build_dev; u64 n_build_cap;
  const u64* n_probe_dev; u64 n_probe_cap;
  u32 tbl_mask;             // table slots - 1 (power of two >= 2 x build rows)
  uint2* gslots;            // null: the table lives in LDS (one copy per workgroup); else ONE {key0,row} table in HBM
  const u32* direct;        // non-null: direct-address table in HBM instead (single unique dense key): row = direct[key - direct_min]
  u32 direct_min, direct_n;
  const u32* csr_off;       // non-null: CSR table instead (single dense key, duplicates allowed): direct_n + 1 offsets into csr_rows
  const u32* csr_rows;      // row ids grouped by key; null = identity (the build column is sorted by the key)
  u32 row_lanes_log2;       // CSR: lanes sharing one probe row (its matches are dealt round-robin)
  // CSR + fused chain whose first stage is an integer window on a decoded value of the stage's slice: the CSR groups
  // re-ordered by that value (range_rows / range_vals, built once per store version), so that a probe row expands only
  // the sub-range of its group that can pass the window (two binary searches) instead of the whole group.  A pruning
  // only: the stage itself still checks every candidate.
  const u32* range_rows; const u32* range_vals;   // values biased: stored = value - range_vbase + 1 (0 = the stage has no row for the key)
  long long range_vbase;
  const u32* range_link; const u32* range_link_col;   // the link column (range_link_col, a build column) in the index's order;
                                                      // non-null: candidates carry index positions, not build rows
  u32 n_chain;              // fused follow-up lookups (0 = none); then the output columns are chain_out[], not proj[]
  ChainStage chain[kMaxChain];
  ColRef chain_out[kMaxCols];
  u64* n_out_dev;           // exact number of matches (zeroed before launch)
  u64 out_cap;              // rows the out columns can hold (optimistic)
  u32 wave_q;               // entries of each wave's LDS candidate queue (>= 64; 8 queues x 8 B x wave_q of LDS)
  u32* overflow;            // set when the matches did not fit
  u8* visited;              // left join: per build row
  u32 probe_outer;          // LEFT JOIN preserved on the PROBE side (the engine built on the right input's table): a probe row without a match is
                            // emitted once with an all-null build side (candidate {kOuterNull, row}); needs no filter, no chain, one lane per row
  u32 stream_direct;        // 1: a direct-address join of a shape stream_join.hip takes runs there (registers, no queue); 0: always the generic kernel
  // stream_join.hip, window filter whose x operand is ONE build column: that column's decoded xsd:integer by KEY (index key - direct_min; INT64_MIN = the
  // table has no row for the key), built per execution next to the direct table — a probe row then costs one 8-byte gather instead of the dependent
  // chain direct[] -> value id -> typed value.  stream_need_build_row: an output column or the post filter reads the build row (else direct[] is not read).
  const long long* key_vals; u32 stream_need_build_row; u32 pad_stream;
  u32 has_filter, has_probe_filter;   // join filter: 0 none / 1 VM / 3 window ; probe filter: 0 none / 1 id-literal / 2 VM
  TypedTable tt;
  // Generic programs live in device memory (a 2.5 KB by-value kernarg block made the compiler copy the
  // whole argument struct to scratch); the specialised shapes get their few parameters inline.
  const ExprProgram* prog;        // join filter over [left cols, right cols] (VM)
  const ExprProgram* probe_prog;  // fused FilterExec of the probe child, over the probe side's columns (VM)
  u32 probe_col_base;             // first column of the probe side inside cols[]
  WindowFilter win;               // has_filter == 3
  IdPairFilter idp;               // has_filter == 2
  IdFilter pid;                   // has_probe_filter == 1; pid.col indexes cols[] directly
  // A `col <=|!=> literal` FilterExec that sat on the BUILD side's input (a store slice whose cached table must stay
  // unfiltered): applied to every candidate pair as one more conjunct of the join filter.  post.col indexes cols[].
  u32 has_post; IdFilter post;
};
// The HIP runtime loads a translation unit's code object lazily, at the first use of one of its kernels: 3 - 35 ms each, which a first
// query would pay (round 3's bench showed a 20 ms "cold start" of which 1.6 ms were kernels).  A store loads them all when it is created.
void preload_code_objects();
void preload_tu_kernels();
void preload_tu_band_join();
void preload_tu_ordered_join();
void preload_tu_ntriples();
void preload_tu_part_join();
void preload_tu_part_pass();
void preload_tu_stream_join();
void preload_tu_exchange();
void preload_tu_topk();
void preload_tu_closure();
void preload_tu_join_fs0();
void preload_tu_join_fs1();
void preload_tu_join_fs2();
void preload_tu_join_fs3();
void launch_lds_join(const LdsJoinArgs& a, hipStream_t s);
// stream_join.hip: the same operator against a direct-address table as a streaming pass (launch_lds_join routes to it when a.stream_direct)
bool direct_stream_join_ok(const LdsJoinArgs& a);
int direct_stream_join_items(u64 n_probe_cap);
void launch_direct_stream_join(const LdsJoinArgs& a, hipStream_t s);

// ---- ordered slice join (ordered_join.hip): a small table against a store slice, matches emitted in the slice's order ----
constexpr u32 kOjMaxOutCols = 8;   // output columns of an ordered slice join (its write kernel keeps the column schedule in SGPRs)
struct OrderedJoinStage { const u32* key_col; const u32* direct; u32 kmin, kn; u32* row; };   // key_col: a column of the TABLE; row[r] = the stage's row of table row r
struct OrderedJoinArgs {
  const u32* build_key; u64 n_build;                                   // the slice's join-key column, rows in slice order
  const u32* probe_key; const u64* n_probe_dev; u64 n_probe_cap;        // the table
  u32 kmin, kn;                                                        // key range of the slice (its dense table's)
  uint2* head; u32* next;                                              // multimap of the table's rows by key: head[kn] = {first row of the chain, chain length - 2} (0xFF-filled: empty; a one-row chain leaves the second word alone), next[table rows]
  u32 n_stages; u32 pad0; OrderedJoinStage stage[kMaxChain];
  u32 n_out_cols; u32 n_rec; ColRef out_ref[kMaxCols]; u32* out[kMaxCols];   // src 0 = table row, 1 = slice row, 2 + t = stage t's row
  // everything an output row takes from the table row or its stage rows, packed per table row (n_rec x 16 B, written by
  // the probe pass): ONE gather per match instead of one per column.  out_slot[c] = word of the record, 0xFF = slice column
  uint4* trec; u8 out_slot[kMaxCols];
  u64 out_cap; u64* n_out_dev; u32* overflow;
  u32* tile_count; u32* tile_off;                                      // per 1024-row tile of the slice (+ 1) and their exclusive scan
  u32* row_head; unsigned char* row_cnt;                               // per slice row, from the count pass: first table row of its chain, chain length (capped at 255: longer chains are re-walked)
};
struct BandArgs;
// The ordered slice join feeding a band join (band_join.hip) writes the band join's ROW RECORDS itself: the windows, the id
// operand and the row's output values depend on the TABLE row (the query instance) only, so they are decoded once per table
// row (oj_band_records_kernel) and a match copies 32 bytes — the intermediate table between the two joins is never written,
// the band join's decode pass (a pass over every match with two typed-value gathers each) does not run.
struct OjBandFuse {
  uint4* brec;                       // per table row: {record, aux} as band_decode_kernel would write them for a match of that row
  uint4* rec_s; uint4* aux_s;        // the band join's row records, in match order (= sorted by the band join's key)
  u32* poff; u32* bcount; u32 max_blocks;   // key boundaries of the matches; the block counts the decode pass zeroes
  u32 kmin, kn;                      // the band join's key range
  const u32* key_col;                // the slice's sorted column: the band join's key of a slice row
  u8 y0_slot[2], y1_slot[2], neq_slot, row_slot[2];   // words of the packed table record holding the window operands / id operand / output values (0xFF: none)
  u8 self_index;                     // the record's id-operand word holds the match's SLICE ROW instead (BandArgs::neq_self)
  u8 compact;                        // one 16-byte record per table row and match {lo pair, width pair, id operand, output value 0} instead of {record, aux} (BandArgs::compact)
};
void launch_oj_band_records(const OrderedJoinArgs& a, const BandArgs& b, const OjBandFuse& f, hipStream_t s);
void launch_ordered_join_write_band(const OrderedJoinArgs& a, const OjBandFuse& f, hipStream_t s);
u64 ordered_join_tiles(u64 n_build);
void launch_ordered_join_probe(const OrderedJoinArgs& a, hipStream_t s);
void launch_ordered_join_count(const OrderedJoinArgs& a, hipStream_t s);
void launch_ordered_join_write(const OrderedJoinArgs& a, hipStream_t s);

// ---- radix-partitioned LDS hash join (part_join.hip): large build sides that are not cached store slices ----
constexpr u32 kPartChunk = 2048;      // build rows per LDS table (a partition with more is joined chunk by chunk)
constexpr u32 kPartSlots = 4096;      // slots of the LDS table: load <= 0.5, ~0.25 at the target partition size
constexpr u32 kPartTargetRows = 1024; // build rows per partition aimed for
struct PartRec { u32 row, k0, k1; };   // 12 bytes: what travels through the partition sort (a padded 16-byte record cost a quarter more sort traffic)
struct PartArgs {
  const PartRec* bpart; const PartRec* ppart;   // {row, key0, key1} records of the build / probe side, grouped by partition
  const u32* bstart; const u32* pstart;     // [n_parts + 1] first record of every partition
  u32 n_parts, chunk, tbl_mask;
  u32 two_pass;   // 1: count a partition's matches first and reserve its output range once (large outputs); 0: one reservation per full queue
  // range mode (ppart == nullptr): the probe side is a store slice sorted by one of the join keys and is read in place —
  // partition p = the slice rows pstart[p] .. pstart[p + 1] (key ranges of 2^shift ids), only the build side was partitioned
  const u32* pcol0; const u32* pcol1;
  // large-output form, join filter `build column <=|!=> probe column` (FS = 2): the build side's operand lives in the LDS table beside the row id and
  // the probe row's operand in a register — the filter is decided while the chain is walked, no candidate that fails it is queued, neither pass
  // gathers for it.  inl_build / inl_probe: the two columns (null = the form is not used).
  const u32* inl_build; const u32* inl_probe;
};
// range < 0: partition = top `bits` bits of the key hash.  Else the probe side is a slice sorted by key[range]: partitions are
// key ranges of that column, equalised over the slice's rows through a coarse directory — coarse bucket c = (key - range_min)
// >> cshift owns dir[c].y consecutive partitions from dir[c].x on (in proportion to the slice rows that fall into it), and
// inside a bucket the ids are split evenly: partition = dir[c].x + (((key - bucket start) * dir[c].y) >> cshift).  Keys outside
// [range_min, range_max] join nothing.
struct PartKeyRange { int range; u32 range_min, range_max, cshift, n_coarse; const uint2* dir; };
void launch_part_keys(const u32* k0, const u32* k1, u32 n_keys, const u64* n_dev, u64 cap, u32 bits, u32 n_parts, PartKeyRange kr, u32* skey, PartRec* sval, hipStream_t s);
void launch_part_equalise(const u32* sorted_col, u64 n, PartKeyRange kr, u32 n_parts, uint2* dir, hipStream_t s);
void launch_part_range_bounds(const u32* sorted_col, u64 n, PartKeyRange kr, u32 n_parts, u32* pstart, hipStream_t s);
void launch_sorted_bounds(const u32* sorted_keys, u64 n, u32 n_keys, u32* start, hipStream_t s);   // start[k] = first position with key >= k, k = 0 .. n_keys
// the partition passes, hand-written (part_pass.hip): one or two MSD passes that recompute the partition from the keys they move —
// no key array, 64 instead of 92 bytes per row; `start` comes out of the last pass's scanned histogram
struct PartPassPlan { bool two; u32 nb_a, tiles_a, max_tiles_b; u64 hist_a, hist_b, tile_desc_bytes; };   // array sizes for n rows and `bits` partition bits
struct PartPassBuffers {
  PartRec* recs;                 // [n] the records grouped by partition (the result)
  PartRec* recs_a;               // [n] the records after pass A (two passes only)
  unsigned short* pid16;         // [n] the partition of every row (between pass A's histogram and its scatter)
  unsigned char* digit;          // [n] partition & 255 of every record after pass A (two passes only)
  u32* hist_a;                   // [hist_a] counts per (bin, tile) of pass A, scanned in place over a bin's tiles
  u32* hist_b;                   // [hist_b] the same of pass B, per (bucket, bin, tile)
  u32* total; u32* base_a;       // [max(nb_a, n_parts) + 1] rows per bin; [nb_a + 1] first record of every bucket of pass A
  void* tiles_b; u32* tb; u32* n_tiles_b;   // [tile_desc_bytes] pass B's tiles, [nb_a + 1] first tile of every bucket, [1] their number
  u32* start;                    // [n_parts + 1] first record of every partition
};
PartPassPlan part_pass_plan(u64 n, u32 bits);
void part_pass_run(const PartPassBuffers& w, const PartPassPlan& pl, const u32* k0, const u32* k1, u32 n_keys, const u64* n_dev, u64 cap, u32 bits, u32 n_parts,
                   PartKeyRange kr, void* scan_temp, size_t scan_temp_bytes, hipStream_t s);
size_t part_sort_temp_bytes(u64 n, u32 bits);
void part_sort(const u32* kin, u32* kout, const PartRec* vin, PartRec* vout, u64 n, u32 bits, void* temp, size_t temp_bytes, hipStream_t s);
void launch_part_join(const LdsJoinArgs& a, const PartArgs& pa, hipStream_t s);

// ---- key-partitioned band join (band_join.hip): the fused chain over a CSR join with small groups, group by group ----
constexpr u32 kBandMaxSideCols = 4;   // output columns taken from the group entry; from the probe row: kBandMaxRowCols
constexpr u32 kBandMaxRowCols = 2;    // (they travel inside the row's 32-byte record)
constexpr u32 kBandMaxGroup = 512;    // largest CSR group (rows of one key) the path accepts: 8 chunks of 64 entries
struct BandWin {              // one integer window stage: lo(row) <= x(entry) <= hi(row)
  const u32* key_col;         // build column holding the stage's key
  const long long* val; u32 vkmin, vkn;   // x = val[key - vkmin] (INT64_MIN = no row)
  long long vbase;            // bias of the 32-bit intervals: stored x = x - vbase + 1
  const u32* y0; const u32* y1;   // probe columns of the two halves
  TvLiteral l0, l1;
  u32 stage, pad;             // index into LdsJoinArgs::chain (full-semantics path)
};
struct BandStage { const u32* key_col; const u32* direct; u32 kmin, kn; };   // look-up of a build column in a slice's direct table
// Everything the band kernels need, and nothing else (the fused join kernel's argument block is 1.6 KB: passed by value it
// costs 118 spilled SGPRs in the pair-test loop).
struct BandArgs {
  // the CSR table of the build side and the probe side's key column
  const u32* csr_off; const u32* csr_rows; u32 kmin, kn;
  const u32* probe_key; const u64* n_probe_dev; u64 n_probe_cap;
  // what a group entry (a build row) is checked against, once per entry
  u32 n_stages, n_win; BandStage stage[kMaxChain]; BandWin win[2];
  u32 has_post, post_lit, post_is_eq, has_neq; const u32* post_col;
  const u32* neq_build; const u32* neq_probe; u32 neq_is_eq, pad0;   // base join filter `build col <ID_EQ|ID_NEQ> probe col`
  TypedTable tt;
  // when every stage hangs off the SAME build column (the usual star: ?product), the stage look-ups are done once per
  // distinct key value (pt: {x0_b, x1_b, joins, 0, stage-sourced output values 0..3}, 32 B per key of [pt_min, pt_min + pt_n))
  // and an entry fetches its key's record with one gather; pt == null: every entry does its own look-ups
  uint4* pt; u32 pt_min, pt_n; const u32* pt_key_col; u32 pt_out_slot[kBandMaxSideCols];
  // the build side's groups, decoded once per execution in CSR order (entry = position in the CSR row list)
  uint4* et;                  // per entry {x0_b, x1_b, id operand, 1 = the entry can join}: what the pair test reads
  u32* eo[kBandMaxSideCols];  // per entry: its output values (one array per entry column)
  u64 n_entries;              // = build rows
  // the probe side, decoded in row order and radix-sorted by key - kmin (kn = joins nothing)
  u32* skey_in; u32* sval_in; const u32* skey; const u32* perm;
  uint4* rec;                 // per probe row (row order), 2 x uint4: {lo0, w0, lo1, w1} {id operand, flags (1 = full semantics), output value 0, 1}
  uint4* rec_s; uint4* aux_s; // the same two halves in sorted order (what the block kernels stream)
  u32* poff; u32* boff;       // [kn + 1]: first sorted row / first 64x64 block of every key
  uint4* bdesc;               // per block {first entry, entries (<= 64), first sorted row, rows (<= 64)}
  u64* masks;                 // 64 x u64 per block: bit e of lane r = (entry e, row r) joins
  u32* bcount; u32* bofs;     // per block (+ 1): output rows / their exclusive scan
  u32 pack16;                   // every window's biased values fit 16 bits: the pair test checks both windows with packed 16-bit arithmetic
  u32 neq_self;                 // `entry id != row id` is `entry index != the row's own entry`: the band join's groups ARE the rows of the slice the ordered
                                // slice join below streamed, the compared ids are that join's keys — the one entry whose id equals the row's is the slice
                                // row the match came from; its index travels in the compact record (word z) and the pair loop drops the id compare
  u32 compact;                  // rec_s holds the WHOLE row record {lo pair, width pair, id operand, output value 0} (16 B; aux_s unused): packed form written
                                // by the ordered slice join (OjBandFuse), at most one row output column, no full-semantics pass
  u32* key_hist; u32* key_cursor;   // counting-sort form of the partition pass (small probe sides): rows per key, counted by the decode pass; the scatter's cursors
  u32 launch_blocks; u32 pad3; u64* n_blocks_out;   // waves launched by the block kernels (>= the previous execution's blocks; they stride on if there are more); the count, for next time
  u32 max_blocks, presorted;  // presorted: the probe rows arrive sorted by key — no sort, records written in place
  u32* slow_rows;             // number of probe rows that need the full typed-value semantics (usually 0)
  unsigned long long* run_stats;   // sampled (rows << 32 | runs of equal neighbouring keys): is the probe side piecewise sorted? (next execution's partition pass)
  // output
  u32 n_out_cols, n_entry_cols, n_row_cols, pad1;
  u32* out[kMaxCols]; u64 out_cap; u64* n_out_dev; u32* overflow;
  ColRef entry_col[kBandMaxSideCols]; const u32* row_col[kBandMaxSideCols];
  u8 out_from_row[kMaxCols];  // per output column: 1 = next row column, 0 = next entry column
  u8 out_sel[kMaxCols];       // .. resolved: 0 / 1 = row column 0 / 1, 2 + u = entry column u
};
void launch_band_pt(const BandArgs& b, hipStream_t s);
void launch_band_entries(const BandArgs& b, hipStream_t s);
void launch_band_rows(const BandArgs& b, hipStream_t s);      // records into sorted order
void launch_band_scatter(const BandArgs& b, hipStream_t s);   // counting-sort form: records to their key's range (poff = scan of key_hist)
void launch_band_desc(const BandArgs& b, hipStream_t s);
void launch_band_decode(const BandArgs& b, hipStream_t s);   // + sort keys
void launch_band_bounds(const u32* skey_sorted, u64 n, u32 kn, u32* poff, hipStream_t s);
size_t band_blocks_scan_temp_bytes(u32 kn);
void band_blocks_scan(const u32* csr_off, const u32* poff, u32 kn, u32* boff, void* temp, size_t temp_bytes, hipStream_t s);   // blocks per key computed inside the scan's input iterator
void launch_band_mask(const BandArgs& b, hipStream_t s);
void launch_band_slow(const LdsJoinArgs* a_dev, const BandArgs& b, hipStream_t s);   // patches masks / counts of the rare slow rows
void launch_band_emit(const BandArgs& b, hipStream_t s);
size_t sort_u32_temp_bytes(u64 n, u32 bits);
void sort_pairs_u32_u32(const u32* kin, u32* kout, const u32* vin, u32* vout, u64 n, u32 bits, void* temp, size_t temp_bytes, hipStream_t s);
void launch_val_minmax(const long long* val, u64 n, long long* out_minmax /* preset {INT64_MAX, INT64_MIN + 1} */, hipStream_t s);
void launch_csr_max_group(const u32* off, u32 kn, u32* out_dev /* zeroed */, hipStream_t s);
void launch_gjoin_build(const LdsJoinArgs& a, hipStream_t s);   // fills a.gslots (memset to 0xFF first)
int lds_join_items(u64 n_probe_cap, bool global);   // rows per lane of the instantiation that will be launched: 4 / 1
enum { kJoinTableLds = 0, kJoinTableHash = 1, kJoinTableDirect = 2, kJoinTableCsr = 3 };   // lds_join_kernel's MODE
int lds_join_mode(const LdsJoinArgs& a);
void launch_minmax_u32(const u32* col, u64 n, u32* out_dev /* {min, max}: preset to {~0, 0} */, hipStream_t s);   // nulls (0) skipped
// val[key - kmin] = xsd:integer value of valcol[row] for every row of a direct table's slice (val preset to INT64_MIN);
// *bad is raised when a value is not an xsd:integer or equals the sentinel
void launch_direct_values(const u32* keys, const u32* valcol, u64 n, u32 kmin, u32 kn, const TypedTable& tt, long long* val, u32* bad_dev, hipStream_t s);
void launch_fill_i64(long long* p, long long v, u64 n, hipStream_t s);
// range index build: key64[p] = (group << 32) | biased value of the row at CSR position p, rows_in[p] = that row
void launch_range_minmax(const u32* stage_key_col, const u32* csr_rows, u64 n, const long long* val, u32 vmin_key, u32 vn, long long* out_minmax /* {min,max} preset */, hipStream_t s);
void launch_range_keys(const u32* group_col, u32 gmin, const u32* stage_key_col, const u32* csr_rows, u64 n, const long long* val, u32 vmin_key, u32 vn,
                       long long vbase, u64* key64, u32* rows_in, hipStream_t s);
void launch_range_decode(const u64* key64_sorted, u64 n, u32* biased_vals_out, hipStream_t s);
void launch_gdirect_build(const u32* keys, u64 n, u32* direct /* 0xFF-filled */, u32 kmin, u32 kn, u32* dup_dev, hipStream_t s);

// ---- DISTINCT + TopK per group (the operators directly above the path; topk.hip) ----
struct TopkArgs {
  const u32* in[kMaxCols];
  const u64* n_in_dev; u64 n_in_cap;
  u32 has_group, group_col;
  u32 n_keys, key_col[4], key_by_term[4];   // key_by_term[i] = RDFGPU_SORT_BY_*
  u32 k, n_groups;            // rows kept per group; group ids are < n_groups
  TypedTable tt;
  u32* counts;                // [n_groups + 1], zeroed: rows per group
  u32* offsets;               // [n_groups + 1]: exclusive scan of counts
  u32* cursor;                // [n_groups]: copy of offsets, consumed by the scatter
  u32* perm;                  // [n_in_cap]: row ids grouped by group
  u32* picked;                // [n_groups * k]
  u32* out_counts;            // [n_groups + 1], zeroed
  u32* out_offsets;           // [n_groups + 1]
  u32* out[kMaxCols]; u32 n_out_cols; u32 proj[kMaxCols];
  u64* n_out_dev;
  u32* bad;                   // raised when a SORT_BY_TERM column holds a kind whose order is not defined here
};
void launch_topk_max(const u32* col, const u64* n_dev, u64 cap, u32* out_max /* zeroed */, hipStream_t s);
void launch_topk_hist(const TopkArgs& a, hipStream_t s);
void launch_topk_scatter(const TopkArgs& a, hipStream_t s);
void launch_topk_select(const TopkArgs& a, hipStream_t s);
void launch_topk_write(const TopkArgs& a, hipStream_t s);

// ---- utilities ----
void launch_fill_u32(u32* p, u32 v, u64 n, hipStream_t s);
void launch_mark_not_equal(const u32* col, u32 value, u32* keep, u64 n, hipStream_t s);   // keep[i] = col[i] != value
void launch_gather_u32(const u32* src, const u32* idx, u32* dst, u64 n, hipStream_t s);
void launch_iota_u32(u32* p, u64 n, hipStream_t s);
void launch_csr_rel_keys(const u32* keys, u64 n, u32 kmin, u32 kn, u32* rel, u32* unsorted_dev, hipStream_t s);   // rel = key - kmin (kn: joins nothing) + sortedness: the CSR build without atomics
void launch_pack_key(const u32* hi, const u32* lo, const u32* idx /*nullable*/, u64* key, u64 n, hipStream_t s);
void launch_unique_flags(const u32* c0, const u32* c1, const u32* c2, const u32* c3, u32* flags, u64 n, hipStream_t s);
void launch_scatter_if(const u32* src, const u32* flags, const u32* excl, u32* dst, u64 n, hipStream_t s);
void launch_mark_removed(const u32* const ix[4], u64 n_ix, const u32* const rm[4], u64 n_rm, u32* keep, hipStream_t s);
// device-wide scans (rocPRIM; load path + per-join offsets)
constexpr u64 kSmallScanElems = 16ull * 4096ull;   // scans of up to this many counts run in one workgroup (join_device.hpp: small_scan_kernel)
size_t scan_temp_bytes(u64 n);
void exclusive_scan_u32(const u32* in, u32* out, u64 n, void* temp, size_t temp_bytes, hipStream_t s);
void inclusive_scan_u32(const u32* in, u32* out, u64 n, void* temp, size_t temp_bytes, hipStream_t s);
size_t sort_temp_bytes(u64 n);
void sort_pairs_u64_u32(const u64* kin, u64* kout, const u32* vin, u32* vout, u64 n, void* temp, size_t temp_bytes, hipStream_t s);

}  // namespace rdfgpu
