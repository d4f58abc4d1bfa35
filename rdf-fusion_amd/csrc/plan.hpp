// plan.hpp — compiled operator tree: DataSourceExec / FilterExec / HashJoinExec / CrossJoinExec /
// NestedLoopJoinExec / ProjectionExec over HBM-resident binding tables.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "host_logic.hpp"
#include "kernels.hpp"
#include "store.hpp"

namespace rdfgpu {

struct SourceInfo {               // one DataSourceExec
  u32 node = 0;
  u32 components = RDFGPU_GSPO;   // chosen permutation (IndexPermutations::choose_index)
  ScanInstructions ix;            // instructions in the order of `components`
  PrunePlan prune;
  bool has_residual = false;      // predicates left after pruning => K2 compaction, else zero-copy slice
  u32 n_out = 0;
  u32 out_level[4] = {};          // index level feeding output column k (G,S,P,O order of first binding)
  u64 lo = 0, hi = 0;             // located range of the last execution
  u32 sorted_level = 4, key_min = 0, key_max = 0;   // the level sorted within [lo, hi) and its first / last id (4: none)
  ScanInstructions gspo;          // the pattern's instructions in G,S,P,O order, push-down filters folded in
  std::vector<std::pair<u32, ScanPredicate>> dynamic;   // current dynamic filters: (variable slot, predicate)
  bool dynamic_dirty = false;     // the effective instructions have to be derived again before the next execute
};

struct NodeInfo {
  rdfgpu_plan_node d;
  u32 width = 0;                  // output columns
  u32 n_proj = 0; u32 proj[kMaxCols] = {};
  ExprProgram prog{};             // filter / join filter
  int shape = 0;                  // filter kernel specialisation
  u32 n_enc_tv = 0;               // ENC_TV gathers per row (algorithmic bytes)
  u32 n_cols_read = 0;            // distinct input columns the kernel has to read
  int source = -1;                // index into Plan::sources
  u32 refs = 0;                   // how many operators consume this node
  u64 last_rows = 0; bool has_last = false;   // output cardinality of the previous execution (speculative sizing)
  bool last_scaled = false;
  bool transient_direct_failed = false;        // a build side that is not cached turned out not to be unique and dense: do not try the direct-address form again
  u64 band_blocks = 0;                         // blocks of the band join based on this node in its previous execution (launch sizing)
  u64 band_run_stats = 0;                      // sampled rows << 32 | runs of equal neighbouring probe keys (a piecewise sorted probe side takes the counting partition)
  int parent = -1;                             // the one operator consuming this node (-1: the root, or several)
  bool band_takes_records = false;             // this node's band join found its probe side presorted by an ordered slice join below and needed no slow pass: next time that join may write the row records itself
  bool band_ran = false; u64 band_slow_rows = 0;   // .. and how many of its probe rows needed the full typed-value semantics                    // .. extrapolated from a priming run over a prefix of the bound tables
};
struct SpecCheck { NodeInfo* node; u32 counter; bool left_join; };   // counter = n_out slot, counter+1 = overflow flag

// A run of inner single-key joins above a base join whose other inputs are store slices with cached direct-address
// tables: handed to the base join, which runs them inside its resolve phase (kernels.hpp ChainStage).
struct ChainLink { NodeInfo* node; bool slice_is_left; DevTable slice; const SliceTable* table; };
struct ChainRequest { NodeInfo* top = nullptr; NodeInfo* base = nullptr; std::vector<ChainLink> links /* bottom-up */; bool consumed = false; };

// An ordered slice join whose write pass is held back: its consumer, a band join, may have it write the band join's row
// records instead of the output table (OjBandFuse); anything else flushes it (Plan::flush_pending_oj) first.
struct PendingOj { bool active = false; OrderedJoinArgs o{}; const u32* first_col = nullptr; u64 n_build = 0; u32 n_chain = 0; };

struct BoundTable { std::vector<const u32*> cols; u64 n_rows = 0; bool bound = false; };

// Kernel classes for per-kernel timing; names are what rocprofv3 --kernel-trace prints.
enum KernelClass {
  KC_LOCATE, KC_SCAN_COUNT, KC_SCAN_WRITE, KC_FILTER_ID, KC_FILTER_TV, KC_FILTER_VM, KC_CROSS, KC_JOIN_BUILD,
  KC_JOIN_COUNT, KC_JOIN_WRITE, KC_LEFT_TAIL, KC_NLJ_COUNT, KC_NLJ_WRITE, KC_DEVICE_SCAN,
  KC_GJOIN_BUILD,
  KC_GDIRECT_BUILD, KC_MINMAX, KC_CSR_HIST, KC_CSR_SCATTER,
  KC_TOPK_MAX, KC_TOPK_HIST, KC_TOPK_SCATTER, KC_TOPK_SELECT, KC_TOPK_WRITE,
  KC_FILTER_VERDICT, KC_REGEX_VERDICTS, KC_UNION,
  KC_BAND_SLOW, KC_RADIX_SORT, KC_BAND_BOUNDS, KC_BAND_BLOCKS, KC_BAND_DECODE, KC_BAND_MASK, KC_BAND_EMIT, KC_BAND_ENTRIES, KC_BAND_DESC, KC_BAND_PT, KC_BAND_ROWS,
  KC_FILTER_BITS_ID, KC_FILTER_BITS_TV, KC_FILTER_BITS_VERDICT, KC_FILTER_BITS_VALUE, KC_VALUE_VERDICTS, KC_VALUE_RUNS, KC_RUN_SCAN, KC_RUN_COPY, KC_OJ_PROBE, KC_OJ_COUNT, KC_OJ_WRITE, KC_FILTER_WRITE,
  KC_PART_KEYS, KC_PART_JOIN, KC_OJ_BAND_RECORDS, KC_OJ_WRITE_BAND, KC_SMALL_SCAN, KC_PART_PASS, KC_STREAM_JOIN,
  KC_LDS_JOIN0,                      // 192 names: lds_join_kernel<FS in {0..3}, PFS in {0,1,2}, ITEMS in {4,1}, MODE in {0,1,2,3}, CHAIN>
  KC__N = KC_LDS_JOIN0 + 192
};
const char* kernel_class_name(int kc);
inline int lds_join_class(u32 fs, u32 pfs, int items, int mode, bool chain = false) {
  return KC_LDS_JOIN0 + (int)((((fs * 3 + pfs) * 2 + (items == 4 ? 0 : 1)) * 4 + mode) * 2) + (chain ? 1 : 0);
}

struct KernelStat { u32 launches = 0; double ms = 0; u64 bytes = 0; u64 rows = 0; };

// One timed launch whose byte count may depend on device-side row counts, resolved after the run.
struct PendingLaunch {
  int kc;
  hipEvent_t start, stop;
  u64 fixed_bytes;      // bytes known on the host
  u64 rows_cap;         // rows streamed if no device count
  const u64* rows_dev;  // device count of streamed rows (or null)
  u64 bytes_per_row;    // multiplied by the live streamed rows
  const u64* out_dev;   // device count of produced rows (or null)
  u64 out_rows;         // produced rows if known on the host
  u64 bytes_per_out;    // multiplied by the produced rows
};

struct Plan {
  Store* store = nullptr;
  EngineOptions opt;              // copied from the store at compile time (rdfgpu_plan_set_option changes this copy)
  std::vector<NodeInfo> nodes;
  std::vector<SourceInfo> sources;
  std::vector<u32> pool;          // IN-set ids of residual predicates (host copy)
  u32* pool_dev = nullptr;        // same, on device
  RegexProg* regex_dev = nullptr; // compiled REGEX patterns of the plan (device)
  unsigned char* str_consts_dev = nullptr;   // bytes of the plan's string constants (RDFGPU_EX_LIT_STR)
  std::vector<std::string> regex_strings; std::vector<rdfgpu_regex> regex_text;   // their texts (pattern, flags per entry)
  u32 root = 0;
  ExecContext* ctx = nullptr;     // stream, events, counters (pooled per store)
  hipStream_t stream = nullptr;
  std::vector<BoundTable> tables;

  // per-execution state
  std::vector<void*> allocs;      // pool blocks owned by the current result / intermediates
  u64* counters = nullptr;        // device u64 slots for operator output counts
  u32 counters_used = 0;
  u32 progs_used = 0;
  u32 arg_slots_used = 0;
  DevTable result; u64 result_rows = 0; bool executed = false;
  std::shared_ptr<IndexGeneration> held;   // the store generation the last execute ran against: zero-copy result slices point into it
  rdfgpu_metrics metrics{};
  bool timing = false;
  int timing_focus = -1;    // >= 0: only launches of this kernel class are bracketed with events (the longest class of the last fully timed execution)
  int last_top_kc = -1;
  u64 located_version = ~0ull;    // store version the cached scan ranges belong to
  bool allow_speculation = true, speculative = false;
  u64 scratch_hist[2] = {0, 0};           // intermediates of the last two completed executions (bounds what the pool keeps cached)
  bool priming = false, primed = false;   // the first execution over big bound tables is preceded by one over their first rows (Plan::prime)
  void prime();
  std::vector<SpecCheck> spec_checks;
  struct BandBlockCounter { NodeInfo* node; u32 counter; u32 slow_counter; u32 runs_counter; bool slow_skipped; };
  std::vector<BandBlockCounter> band_block_counters;   // device-side block counts of this execution's band joins -> NodeInfo::band_blocks
  NodeInfo* cur_band_node = nullptr;                   // the base join whose band join is being set up
  std::vector<PendingLaunch> pending;
  std::vector<DevTable> memo; std::vector<char> memo_valid;   // node results of the current execution
  ChainRequest* pending_chain = nullptr;                        // set while the base join of a fusable chain executes
  SliceTable* cur_build_table = nullptr;                        // the store-level table of the join being set up (if any)
  PendingOj pending_oj;
  void flush_pending_oj();
  u32 events_used = 0;
  KernelStat kstats[KC__N];
  // Arrow batch stream over a host copy of the result
  std::vector<std::vector<u32>> host_cols; bool host_valid = false; u64 cursor = 0;

  ~Plan();
  // the store's typed-value table with this execution's run-time error word attached (the last counter slot)
  TypedTable typed_table() const { TypedTable t = store->typed_table(); t.rt_error = reinterpret_cast<u32*>(counters + 255); return t; }
  void execute();
  void pushdown_filters(u32 node, const rdfgpu_pushdown_filter* filters, u32 n, u8* pushed);
  void set_dynamic_filters(u32 node, const rdfgpu_pushdown_filter* filters, u32 n);
  void derive_source(SourceInfo& src, const ScanInstructions& gspo);
  void upload_pool();
  void ensure_host_copy();

 private:
  DevTable exec_node(u32 idx);
  DevTable exec_source(NodeInfo& nd);
  DevTable exec_filter(NodeInfo& nd);
  DevTable exec_join(NodeInfo& nd);
  DevTable exec_topk(NodeInfo& nd);
  DevTable apply_filter(NodeInfo& nd, const DevTable& in);
  DevTable exec_lds_join(NodeInfo& nd, const DevTable& L, const DevTable& R, bool build_left, const NodeInfo* probe_filter, const NodeInfo* post_filter = nullptr);
  bool plan_chain(NodeInfo& top, ChainRequest& req);
  void build_dense_table(SliceTable* st, const u32* key, u64 n);
  bool build_transient_direct(LdsJoinArgs& a, u64 n);
  bool apply_chain(const ChainRequest& req, NodeInfo& base, const DevTable& L, const DevTable& R, bool build_left, LdsJoinArgs& a, u64& stage_bytes, BandArgs* band, bool* use_band);
  void prepare_partitions(const LdsJoinArgs& a, const DevTable& B, const DevTable& P, PartArgs& pa);
  void exec_band_join(LdsJoinArgs& a, BandArgs& b, const DevTable& B, const DevTable& P, u64 build_bytes_per_row, u64 probe_bytes_per_row);
  bool choose_build_left(const NodeInfo& nd, const DevTable& L, const DevTable& R, bool left_join, bool lf, bool rf, bool lpost = false, bool rpost = false) const;
  void release_intermediates();
  template <class T> T* scratch(u64 n);
  u64* new_counter();
  const ExprProgram* upload_program(const ExprProgram& p);
  u64 read_u64(const u64* dev);
  // brackets one launch with HIP events when timing is on
  template <class F>
  void timed(int kc, u64 fixed_bytes, u64 rows_cap, const u64* rows_dev, u64 bytes_per_row,
             const u64* out_dev, u64 out_rows, u64 bytes_per_out, F&& launch);
  void resolve_timing();
};

Plan* plan_compile(Store* store, const rdfgpu_plan_desc* desc);

}  // namespace rdfgpu
