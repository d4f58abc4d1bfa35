// store.hpp — the HBM-resident quad store: three sorted u32 permutations + typed-value side table.
// Replaces MemQuadStorage / IndexPermutations<MemQuadIndex> / MemIndexData
// (lib/storage/src/memory/storage/mem_storage.rs:22-102, quad_index_data.rs:57-64).
#pragma once
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "common.hpp"
#include "expr_device.hpp"

namespace rdfgpu {

// Size-bucketed caching device allocator: intermediates of successive plan executions reuse the
// same HBM blocks instead of paying hipMalloc/hipFree per operator.
class DevicePool {
 public:
  DevicePool();
  ~DevicePool();
  void* alloc(size_t bytes);
  void free(void* p);
  void trim();                      // every cached block back to the device
  void trim_to(u64 keep_bytes);     // largest cached blocks back to the device until at most keep_bytes stay cached
  u64 bytes_in_use() const { return in_use_; }
  u64 bytes_cached() const { return cached_; }
  // hipMalloc calls this pool had to make (a steady-state execution makes none) and the host time they took
  u64 mallocs() const { return n_mallocs_.load(); }
  double malloc_ms() const { return (double)malloc_ns_.load() * 1e-6; }

 private:
  std::mutex mu_;
  struct Cached { void* p; u64 epoch; };
  std::multimap<size_t, Cached> free_;   // by size; `epoch` = the trim round in which the block was handed back
  u64 epoch_ = 0;
  std::map<void*, size_t> live_;
  u64 in_use_ = 0, cached_ = 0;
  std::atomic<u64> n_mallocs_{0}, malloc_ns_{0};
};

// Per-plan execution resources (a HIP stream, a grow-on-demand event pool, the device counters and
// the locate staging buffers).  Creating these costs far more than a BSBM query takes, so finished
// plans hand them back to the store for the next plan.
struct ExecContext {
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> events;
  u64* counters = nullptr;      // 256 device u64 slots
  u64* counters_host = nullptr; // pinned mirror
  void* jobs_dev = nullptr; u64* lohi_dev = nullptr; u32 job_cap = 0;
  void* jobs_host = nullptr; u64* lohi_host = nullptr;   // pinned staging
  static constexpr u32 kProgSlots = 16;
  ExprProgram* progs_dev = nullptr; ExprProgram* progs_host = nullptr;   // VM programs of fused kernels
  static constexpr u32 kArgSlots = 4, kArgBytes = 4096;
  unsigned char* args_dev = nullptr; unsigned char* args_host = nullptr;  // kernel argument blocks handed over by pointer (pinned staging)
  hipEvent_t event(u32 i);
  ~ExecContext();
};

// Engine options (include/rdfgpu.h section 4b): process defaults are read from the environment exactly once; a store
// copies the defaults at creation, a plan copies its store's at compile time.  The execute path reads only these copies.
struct EngineOptions {
  u64 v[RDFGPU_OPT__COUNT] = {};
  bool on(int o) const { return v[o] != 0; }
};
const EngineOptions& default_engine_options();
const char* engine_option_name(u32 option);

struct Permutation {
  u32* col[4] = {nullptr, nullptr, nullptr, nullptr};  // index-order columns (flat, sorted, unique)
  u64 n = 0;
};
// The three permutations of ONE store version.  The store holds the current generation; every executed plan holds the
// generation it ran against until its next execute / destroy, so the zero-copy slices a result may consist of stay
// valid — and keep showing the pre-mutation rows — while the store moves on (the reference's snapshot: a plan keeps
// an Arc'ed read guard, snapshot.rs:35-37).  A mutation builds a complete new generation and swaps it in only when
// all three permutations succeeded; on failure the store is unchanged and the partial generation frees itself.
struct IndexGeneration {
  Permutation idx[RDFGPU_N_INDEXES];
  int device = 0;
  ~IndexGeneration();
};

// Join table of one key-column slice of the store (a param-free triple-pattern scan joined on `n_keys` of its
// columns).  The slice is the same on every execution of every plan until the store changes, so its table is built
// once per store version and shared by all plans (kept here, not on a plan node: DataFusion compiles a fresh plan per
// query).  Forms: direct-address (single unique dense key), CSR (single dense key with duplicates), else hash.
struct SliceTable {
  bool dense_tried = false, dense_failed = false;
  u32* direct = nullptr; u32 kmin = 0, kn = 0;      // row = direct[key - kmin]
  u32* csr_off = nullptr; u32* csr_rows = nullptr;  // rows of key k: csr_rows[csr_off[k - kmin] .. csr_off[k - kmin + 1]); null rows = identity
  u32 csr_max_group = 0;                            // rows of the largest group (decides whether groups are joined whole, band_join.hip)
  void* slots = nullptr; u32 mask = 0;              // {key0,row} open-addressing table (uint2[mask + 1])
  // decoded payload columns of a direct table: val[key - kmin] = the xsd:integer value of that row's `col` (INT64_MIN =
  // no row); usable == false when some value is not an xsd:integer (then the generic path is the only one)
  struct ValueColumn { const u32* col; long long* val; bool usable; long long vmin = 0, vmax = -1; };   // vmin > vmax: no value at all
  std::vector<ValueColumn> values;
  // a CSR table's groups re-ordered by a decoded value of ANOTHER slice (reached through `link_col`, a column of this
  // slice holding that slice's key): rows[p] / vals[p] for every CSR position p, ascending by value inside each group
  struct RangeIndex { const long long* val; const u32* link_col; u32* rows; u32* vals; long long vbase; u32* link; bool usable; };
  std::vector<RangeIndex> ranges;
  // a CSR table's rows decoded for one band-join chain (band_join.hip: the pair test's operands and the entries' output
  // values, in CSR order): a function of the store's slices and the chain's constants only — `key` spells them out
  struct BandEntries { std::string key; uint4* et; u32* eo[4]; };
  std::vector<BandEntries> band_entries;
  // the slice as the sorted input of a FILTER on its sort column (kernels.hip, run-copy form): lo[i] = first row whose id is >= first + i
  // (i = 0 .. span): where every distinct id's run starts — a function of the slice alone
  struct ValueStarts { u32 first; u64 span; u32* lo; };
  std::vector<ValueStarts> value_starts;
};
struct SliceKey {
  const u32* key[RDFGPU_MAX_KEYS] = {}; u32 n_keys = 0; u64 rows = 0;
  bool operator<(const SliceKey& o) const {
    if (n_keys != o.n_keys) return n_keys < o.n_keys;
    if (rows != o.rows) return rows < o.rows;
    for (u32 i = 0; i < RDFGPU_MAX_KEYS; i++) if (key[i] != o.key[i]) return key[i] < o.key[i];
    return false;
  }
};

struct Store {
  int device = 0;
  u32 batch_size = 8192;
  EngineOptions opt = default_engine_options();
  std::shared_ptr<IndexGeneration> gen;   // owns the columns `idx` points at
  Permutation idx[RDFGPU_N_INDEXES];      // the current generation's permutations (same pointers as gen->idx)
  void adopt(std::shared_ptr<IndexGeneration> fresh);   // swap in a complete generation (caller holds `mu` exclusively)
  // typed-value side table (object_id_mapping.rs:376-399), 16 B per id
  rdfgpu_typed_value* tv = nullptr; u64 n_ids = 0;
  int64_t* dec = nullptr; u64 n_dec = 0;
  // lexical forms of string ids (the slice of the dictionary string builtins read): offsets[n_str_ids + 1] + UTF-8 heap
  u64* str_off = nullptr; unsigned char* heap = nullptr; u64 n_str_ids = 0;
  DevicePool pool;
  // intermediates of the store's last two completed executions, whatever plan ran them: a FRESH plan (the reference compiles one per query) has no
  // history of its own, and trimming the pool to the fixed floor before its first execution handed ~20 cached blocks back to the device (0.2 ms of
  // hipFree each: 4 ms around a 0.55 ms execution) only to allocate them again
  std::atomic<u64> scratch_recent[2] = {{0}, {0}};
  // the slice join tables' own pool: a mutation drops every table, the next executions build them again — out of the blocks
  // the dropped ones gave back, not out of fresh hipMalloc calls (a store that is updated between queries pays kernels, not
  // the allocator, for its tables)
  DevicePool table_pool;
  template <class T> T* table_alloc(u64 n) { return static_cast<T*>(table_pool.alloc((n ? n : 1) * sizeof(T))); }
  void table_free(void* p) { table_pool.free(p); }
  hipStream_t stream = nullptr;  // load-path stream
  std::shared_mutex mu;   // readers = running plans (a snapshot), writers = extend / remove / clear
  // Shared ownership like the reference's Arc<...>: the handle holds one reference, every compiled plan one more; the
  // store goes away with the last of them, so a plan may outlive rdfgpu_store_destroy (scan.rs:418-419 keeps its
  // snapshot alive the same way).
  std::atomic<int> refs{1};
  void retain() { refs.fetch_add(1); }
  void release() { if (refs.fetch_sub(1) == 1) delete this; }
  std::atomic<u64> version{0};   // bumped by every extend / remove / clear: cached scan ranges of plans are keyed on it
  // slice join tables: looked up under slice_mu; a table is built (and its stream synchronised) with slice_build_mu
  // held, so concurrent plans never see a half-built one.  Dropped by every mutation (which holds `mu` exclusively).
  std::mutex slice_mu, slice_build_mu;
  std::map<SliceKey, SliceTable> slice_tables;
  // per-distinct-term verdicts of constant string predicates (key = function, language, flags, pattern): one byte per
  // object id, computed once per dictionary, shared by all plans; dropped when the dictionary or typed values change
  std::map<std::string, unsigned char*> string_verdicts;
  void drop_string_verdicts();
  SliceTable* slice_table(const SliceKey& k);
  const SliceTable* find_slice_table(const SliceKey& k);
  void drop_slice_tables();
  void drop_tables();   // rdfgpu_store_drop_tables: what a mutation does to the caches, without one
  std::mutex ctx_mu;
  std::vector<ExecContext*> free_ctx;
  ExecContext* acquire_context(u32 n_sources);
  void release_context(ExecContext* c);

  ~Store();
  void activate() const;
  u64 extend_device(const u32* g, const u32* s, const u32* p, const u32* o, u64 n);
  u64 extend_host(const u32* g, const u32* s, const u32* p, const u32* o, u64 n);
  u64 remove_host(const u32* g, const u32* s, const u32* p, const u32* o, u64 n);
  u64 remove_graph(u32 graph);
  void clear();
  void set_typed_values(const rdfgpu_typed_value* v, u64 n_ids, const int64_t* dec, u64 n_dec);
  void set_strings(const u64* offsets, u64 n_ids, const unsigned char* heap_host, u64 heap_bytes);
  TypedTable typed_table() const { return TypedTable{tv, n_ids, dec, n_dec, str_off, heap, n_str_ids, nullptr}; }
};

Store* store_create(const rdfgpu_config* cfg);

}  // namespace rdfgpu
