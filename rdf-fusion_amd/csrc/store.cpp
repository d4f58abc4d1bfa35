// store.cpp — see store.hpp.  Index build = device radix sort of three permutations + dedupe.
#include "store.hpp"

#include <chrono>
#include <cstdlib>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace rdfgpu {

// ------------------------------------------------------------------------------------------------
// DevicePool
// ------------------------------------------------------------------------------------------------
static size_t bucket_of(size_t bytes) {
  size_t b = 256;
  while (b < bytes) b <<= 1;
  // above 64 MiB round to 16 MiB multiples instead of powers of two (288 GB is big, not infinite)
  if (bytes > (64ull << 20)) b = (bytes + (16ull << 20) - 1) / (16ull << 20) * (16ull << 20);
  return b;
}
// every pool of the process: an allocation that fails gives the cached blocks of ALL of them back before it gives up
// (a second store must not run out of memory because the first one's pool sits on blocks it no longer uses)
static std::mutex g_pools_mu;
static std::vector<DevicePool*> g_pools;
DevicePool::DevicePool() { std::lock_guard<std::mutex> g(g_pools_mu); g_pools.push_back(this); }
DevicePool::~DevicePool() {
  { std::lock_guard<std::mutex> g(g_pools_mu); for (size_t i = 0; i < g_pools.size(); i++) if (g_pools[i] == this) { g_pools.erase(g_pools.begin() + i); break; } }
  trim();
  for (auto& kv : live_) (void)hipFree(kv.first);
}
void* DevicePool::alloc(size_t bytes) {
  const size_t b = bucket_of(bytes ? bytes : 1);
  std::lock_guard<std::mutex> g(mu_);
  // the smallest cached block that fits, if it is not wastefully large (<= 1.5 x the bucket): speculative sizes drift by a few
  // per cent from execution to execution, and a fresh hipMalloc of a 150 MB block costs about as much as the kernel that fills it
  // (an exact fit first: taking a larger block away from the request it was cached for only moves the miss)
  auto it = free_.find(b);
  if (it == free_.end()) { it = free_.upper_bound(b); if (it != free_.end() && it->first > b + b / 2) it = free_.end(); }
  void* p = nullptr;
  size_t got = b;
  if (it != free_.end()) { p = it->second.p; got = it->first; free_.erase(it); cached_ -= got; }
  else {
    const auto t0 = std::chrono::steady_clock::now();
    struct Tally { DevicePool* self; std::chrono::steady_clock::time_point t0; ~Tally() { self->n_mallocs_++; self->malloc_ns_ += (u64)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } } tally{this, t0};
    hipError_t e = hipMalloc(&p, b);
    if (e != hipSuccess) {
      // give cached blocks back (this pool's, then every other pool's) and retry once
      for (auto& kv : free_) (void)hipFree(kv.second.p);
      free_.clear(); cached_ = 0;
      (void)hipGetLastError();
      e = hipMalloc(&p, b);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> gp(g_pools_mu);
        for (DevicePool* other : g_pools) if (other != this && other->mu_.try_lock()) {
          for (auto& kv : other->free_) (void)hipFree(kv.second.p);
          other->free_.clear(); other->cached_ = 0;
          other->mu_.unlock();
        }
        e = hipMalloc(&p, b);
      }
      if (e != hipSuccess) {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        (void)hipGetLastError();
        fail(RDFGPU_ERR_OOM, "device allocation of %zu bytes failed (%s): %zu of %zu bytes free, %zu bytes held by this pool",
             b, hipGetErrorString(e), free_b, total_b, in_use_);
      }
    }
  }
  live_[p] = got;
  in_use_ += got;
  return p;
}
void DevicePool::free(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> g(mu_);
  auto it = live_.find(p);
  if (it == live_.end()) return;
  in_use_ -= it->second;
  cached_ += it->second;
  free_.emplace(it->second, Cached{p, epoch_});
  live_.erase(it);
}
// Gives cached blocks back to the device until at most keep_bytes stay cached — the ones that have gone unused longest first
// (handed back in the oldest trim round), the larger of two equally old ones first.  Largest-first regardless of age freed
// exactly the blocks the next execution wanted again: two multi-GB hipMallocs per step of a plan whose output is a few GB.
void DevicePool::trim_to(u64 keep_bytes) {
  std::lock_guard<std::mutex> g(mu_);
  epoch_++;
  while (cached_ > keep_bytes && !free_.empty()) {
    auto victim = free_.begin();
    for (auto it = free_.begin(); it != free_.end(); ++it)
      if (it->second.epoch < victim->second.epoch || (it->second.epoch == victim->second.epoch && it->first > victim->first)) victim = it;
    (void)hipFree(victim->second.p);
    cached_ -= victim->first;
    free_.erase(victim);
  }
}
void DevicePool::trim() {
  std::lock_guard<std::mutex> g(mu_);
  cached_ = 0;
  for (auto& kv : free_) (void)hipFree(kv.second.p);
  free_.clear();
}

// ------------------------------------------------------------------------------------------------
// Engine options: names, and the process defaults (environment read once, here and nowhere else)
// ------------------------------------------------------------------------------------------------
static const char* const kOptionNames[RDFGPU_OPT__COUNT] = {
    "FORCE_GENERIC_VM", "NO_JOIN_REORDER", "NO_SPECULATION", "NO_FIRST_RUN_SPECULATION", "NO_STRING_VERDICTS",
    "NO_TABLE_CACHE", "NO_INDEX_JOIN", "NO_CHAIN_FUSION", "NO_VALUE_TABLES", "NO_RANGE_INDEX", "NO_FILTER_FUSION",
    "NO_LDS_JOIN", "NO_GLOBAL_TABLE_JOIN", "NO_DIRECT_TABLE", "NO_BAND_JOIN", "NO_PARTITIONED_JOIN", "NO_VALUE_VERDICTS", "NO_PRIMING", "NO_ORDERED_JOIN", "NO_BAND_PACK16", "NO_RUN_COPY", "NO_RANGE_PARTITION",
    "LDS_MAX_BUILD", "CSR_ROW_LANES_LOG2", "JOIN_WAVE_Q", "PARTITION_MIN_BUILD", "PARTITION_TWO_PASS_ROWS", "NO_OWN_PARTITION_PASS", "NO_BAND_COMPACT", "NO_PROBE_OUTER_JOIN", "NO_STREAM_JOIN", "PARTITION_ROWS", "PARTITION_SLOTS"};
const char* engine_option_name(u32 option) { return option < RDFGPU_OPT__COUNT ? kOptionNames[option] : nullptr; }
const EngineOptions& default_engine_options() {
  static const EngineOptions defaults = [] {
    EngineOptions o;
    o.v[RDFGPU_OPT_LDS_MAX_BUILD] = 1024;
    // a hash table of up to ~2 M rows (32 MB of slots) stays in the 32 MiB of L2 / the Infinity Cache: probing it is cheaper than
    // sorting both sides into partitions (BSBM Q5 un-fused: 0.54 G probe rows against a 285 k-row build: 59 ms vs 330 ms)
    o.v[RDFGPU_OPT_PARTITION_MIN_BUILD] = 1ull << 21;
    o.v[RDFGPU_OPT_PARTITION_TWO_PASS_ROWS] = 50ull << 20;
    for (u32 i = 0; i < RDFGPU_OPT__COUNT; i++) {
      const std::string name = std::string("RDFGPU_") + kOptionNames[i];
      if (const char* e = std::getenv(name.c_str())) {
        const bool is_value = i >= RDFGPU_OPT_LDS_MAX_BUILD;
        o.v[i] = is_value ? std::strtoull(e, nullptr, 10) + (i == RDFGPU_OPT_CSR_ROW_LANES_LOG2 ? 1 : 0) : 1;
      }
    }
    return o;
  }();
  return defaults;
}

// ------------------------------------------------------------------------------------------------
// Store
// ------------------------------------------------------------------------------------------------
Store* store_create(const rdfgpu_config* cfg) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    (void)hipGetLastError();
    fail(RDFGPU_ERR_NO_DEVICE, "no usable HIP device (%s); this library has no CPU fallback",
         e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  }
  int dev = cfg ? cfg->device : -1;
  if (dev < 0) RDFGPU_HIP(hipGetDevice(&dev));
  if (dev >= count) fail(RDFGPU_ERR_NO_DEVICE, "device %d does not exist (%d devices)", dev, count);
  RDFGPU_HIP(hipSetDevice(dev));
  hipDeviceProp_t prop;
  RDFGPU_HIP(hipGetDeviceProperties(&prop, dev));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    fail(RDFGPU_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only", dev, prop.gcnArchName);
  Store* s = new Store();
  s->device = dev;
  s->batch_size = (cfg && cfg->batch_size) ? cfg->batch_size : 8192;
  s->gen = std::make_shared<IndexGeneration>();
  s->gen->device = dev;
  RDFGPU_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  preload_code_objects();   // once per process: no query pays for the runtime's lazy code-object loading
  return s;
}

void preload_code_objects() {
  static std::once_flag once;
  std::call_once(once, [] {
    preload_tu_kernels();
    preload_tu_band_join();
    preload_tu_ordered_join();
    preload_tu_ntriples();
    preload_tu_part_join();
    preload_tu_part_pass();
    preload_tu_stream_join();
    preload_tu_exchange();
    preload_tu_topk();
    preload_tu_closure();
    preload_tu_join_fs0();
    preload_tu_join_fs1();
    preload_tu_join_fs2();
    preload_tu_join_fs3();
  });
}

void Store::activate() const { RDFGPU_HIP(hipSetDevice(device)); }

// ------------------------------------------------------------------------------------------------
// ExecContext pool
// ------------------------------------------------------------------------------------------------
hipEvent_t ExecContext::event(u32 i) {
  while (events.size() <= i) {
    hipEvent_t e;
    RDFGPU_HIP(hipEventCreate(&e));
    events.push_back(e);
  }
  return events[i];
}
ExecContext::~ExecContext() {
  for (hipEvent_t e : events) (void)hipEventDestroy(e);
  if (counters) (void)hipFree(counters);
  if (counters_host) (void)hipHostFree(counters_host);
  if (progs_dev) (void)hipFree(progs_dev);
  if (progs_host) (void)hipHostFree(progs_host);
  if (args_dev) (void)hipFree(args_dev);
  if (args_host) (void)hipHostFree(args_host);
  if (jobs_dev) (void)hipFree(jobs_dev);
  if (lohi_dev) (void)hipFree(lohi_dev);
  if (jobs_host) (void)hipHostFree(jobs_host);
  if (lohi_host) (void)hipHostFree(lohi_host);
  if (stream) (void)hipStreamDestroy(stream);
}
ExecContext* Store::acquire_context(u32 n_sources) {
  activate();
  ExecContext* c = nullptr;
  {
    std::lock_guard<std::mutex> g(ctx_mu);
    if (!free_ctx.empty()) { c = free_ctx.back(); free_ctx.pop_back(); }
  }
  if (!c) {
    c = new ExecContext();
    RDFGPU_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    RDFGPU_HIP(hipMalloc((void**)&c->counters, 256 * sizeof(u64)));
    RDFGPU_HIP(hipHostMalloc((void**)&c->counters_host, 256 * sizeof(u64), hipHostMallocDefault));
    RDFGPU_HIP(hipMalloc((void**)&c->progs_dev, ExecContext::kProgSlots * sizeof(ExprProgram)));
    RDFGPU_HIP(hipHostMalloc((void**)&c->progs_host, ExecContext::kProgSlots * sizeof(ExprProgram), hipHostMallocDefault));
    RDFGPU_HIP(hipMalloc((void**)&c->args_dev, (size_t)ExecContext::kArgSlots * ExecContext::kArgBytes));
    RDFGPU_HIP(hipHostMalloc((void**)&c->args_host, (size_t)ExecContext::kArgSlots * ExecContext::kArgBytes, hipHostMallocDefault));
  }
  const u32 need = n_sources ? n_sources : 1;
  if (c->job_cap < need) {
    if (c->jobs_dev) { (void)hipFree(c->jobs_dev); (void)hipFree(c->lohi_dev); (void)hipHostFree(c->jobs_host); (void)hipHostFree(c->lohi_host); }
    const u32 cap = need < 16 ? 16 : need;
    RDFGPU_HIP(hipMalloc(&c->jobs_dev, cap * 128));          // sizeof(LocateJob) <= 128
    RDFGPU_HIP(hipMalloc((void**)&c->lohi_dev, cap * kLocateWords * sizeof(u64)));
    RDFGPU_HIP(hipHostMalloc(&c->jobs_host, cap * 128, hipHostMallocDefault));
    RDFGPU_HIP(hipHostMalloc((void**)&c->lohi_host, cap * kLocateWords * sizeof(u64), hipHostMallocDefault));
    c->job_cap = cap;
  }
  return c;
}
void Store::release_context(ExecContext* c) {
  if (!c) return;
  std::lock_guard<std::mutex> g(ctx_mu);
  free_ctx.push_back(c);
}

SliceTable* Store::slice_table(const SliceKey& k) {
  std::lock_guard<std::mutex> lock(slice_mu);
  return &slice_tables[k];   // std::map nodes never move
}
const SliceTable* Store::find_slice_table(const SliceKey& k) {
  std::lock_guard<std::mutex> lock(slice_mu);
  auto it = slice_tables.find(k);
  return it == slice_tables.end() ? nullptr : &it->second;
}
void Store::drop_slice_tables() {
  std::lock_guard<std::mutex> lock(slice_mu);
  for (auto& kv : slice_tables) {
    SliceTable& t = kv.second;
    table_free(t.direct); table_free(t.csr_off); table_free(t.csr_rows); table_free(t.slots);
    for (auto& v : t.values) table_free(v.val);
    for (auto& r : t.ranges) { table_free(r.rows); table_free(r.vals); table_free(r.link); }
    for (auto& e : t.band_entries) { table_free(e.et); for (u32* p : e.eo) table_free(p); }
    for (auto& v : t.value_starts) table_free(v.lo);
  }
  slice_tables.clear();
}
void Store::drop_tables() {
  std::unique_lock<std::shared_mutex> lock(mu);   // like a mutation: no plan is running
  activate();
  RDFGPU_HIP(hipDeviceSynchronize());
  version++;
  drop_slice_tables();
}

void Store::drop_string_verdicts() {
  std::lock_guard<std::mutex> lock(slice_mu);
  for (auto& kv : string_verdicts) if (kv.second) (void)hipFree(kv.second);
  string_verdicts.clear();
}

Store::~Store() {
  (void)hipSetDevice(device);
  drop_slice_tables();
  drop_string_verdicts();
  for (ExecContext* c : free_ctx) delete c;
  gen.reset();     // the columns go when the last plan holding this generation lets go
  if (tv) (void)hipFree(tv);
  if (dec) (void)hipFree(dec);
  if (str_off) (void)hipFree(str_off);
  if (heap) (void)hipFree(heap);
  if (stream) (void)hipStreamDestroy(stream);
}

IndexGeneration::~IndexGeneration() {
  (void)hipSetDevice(device);
  for (auto& ix : idx) for (auto& c : ix.col) if (c) (void)hipFree(c);
}
void Store::adopt(std::shared_ptr<IndexGeneration> fresh) {
  version++;
  drop_slice_tables();
  gen = std::move(fresh);
  for (u32 k = 0; k < RDFGPU_N_INDEXES; k++) idx[k] = gen->idx[k];
}

void Store::clear() {
  std::unique_lock<std::shared_mutex> lock(mu);
  activate();
  auto fresh = std::make_shared<IndexGeneration>();
  fresh->device = device;
  adopt(fresh);
}

namespace {
struct Scratch {  // pool-backed temporaries released together
  DevicePool& pool; std::vector<void*> ptrs;
  explicit Scratch(DevicePool& p) : pool(p) {}
  template <class T> T* get(u64 n) { void* p = pool.alloc((n ? n : 1) * sizeof(T)); ptrs.push_back(p); return (T*)p; }
  ~Scratch() { for (void* p : ptrs) pool.free(p); }
};
}  // namespace

// Sorted-unique compaction shared by extend / remove: keeps rows with flags[i] != 0.
static u64 compact_columns(Store& st, Scratch& sc, u32* const src[4], const u32* flags, u64 n, Permutation& dst) {
  hipStream_t s = st.stream;
  u32* excl = sc.get<u32>(n);
  const size_t tb = scan_temp_bytes(n);
  void* temp = sc.get<u8>(tb);
  exclusive_scan_u32(flags, excl, n, temp, tb, s);
  u32 last_excl = 0, last_flag = 0;
  if (n) {
    RDFGPU_HIP(hipMemcpyAsync(&last_excl, excl + n - 1, 4, hipMemcpyDeviceToHost, s));
    RDFGPU_HIP(hipMemcpyAsync(&last_flag, flags + n - 1, 4, hipMemcpyDeviceToHost, s));
    RDFGPU_HIP(hipStreamSynchronize(s));
  }
  const u64 total = (u64)last_excl + (last_flag ? 1 : 0);
  for (int k = 0; k < 4; k++) {
    dst.col[k] = nullptr;     // (owned by the generation from the moment it exists: a later failure frees it)
    if (total) {
      RDFGPU_HIP(hipMalloc((void**)&dst.col[k], total * sizeof(u32)));
      launch_scatter_if(src[k], flags, excl, dst.col[k], n, s);
    }
  }
  dst.n = total;
  RDFGPU_HIP(hipStreamSynchronize(s));
  return total;
}

// IndexPermutations::insert (permutations.rs:102-118) + MemIndexData::insert (quad_index_data.rs:287-332),
// bulk form: concatenate, LSD radix sort on (c0,c1 | c2,c3) as two stable 64-bit passes, drop duplicates.
u64 Store::extend_device(const u32* g, const u32* s_, const u32* p, const u32* o, u64 n) {
  std::unique_lock<std::shared_mutex> lock(mu);
  activate();
  if (n == 0) return 0;
  const u32* in[4] = {g, s_, p, o};
  hipStream_t s = stream;
  u64 inserted = 0;
  auto fresh_gen = std::make_shared<IndexGeneration>();   // all three permutations first, swapped in only when complete
  fresh_gen->device = device;
  for (u32 comp = 0; comp < RDFGPU_N_INDEXES; comp++) {
    const Permutation& ix = idx[comp];
    const u64 N = ix.n + n;
    if (N >= 0xFFFFFFFFull) fail(RDFGPU_ERR_UNSUPPORTED, "index would exceed 2^32-1 rows");
    Scratch sc(pool);
    u32* cat[4];
    for (int k = 0; k < 4; k++) {
      cat[k] = sc.get<u32>(N);
      if (ix.n) RDFGPU_HIP(hipMemcpyAsync(cat[k], ix.col[k], ix.n * 4, hipMemcpyDeviceToDevice, s));
      RDFGPU_HIP(hipMemcpyAsync(cat[k] + ix.n, in[PERM[comp][k]], n * 4, hipMemcpyDeviceToDevice, s));
    }
    u64* key_a = sc.get<u64>(N); u64* key_b = sc.get<u64>(N);
    u32* idx_a = sc.get<u32>(N); u32* idx_b = sc.get<u32>(N);
    const size_t tb = sort_temp_bytes(N);
    void* temp = sc.get<u8>(tb);
    launch_iota_u32(idx_a, N, s);
    launch_pack_key(cat[2], cat[3], nullptr, key_a, N, s);
    sort_pairs_u64_u32(key_a, key_b, idx_a, idx_b, N, temp, tb, s);       // by (c2, c3)
    launch_pack_key(cat[0], cat[1], idx_b, key_a, N, s);
    sort_pairs_u64_u32(key_a, key_b, idx_b, idx_a, N, temp, tb, s);       // stable by (c0, c1)
    u32* sorted[4];
    for (int k = 0; k < 4; k++) { sorted[k] = sc.get<u32>(N); launch_gather_u32(cat[k], idx_a, sorted[k], N, s); }
    u32* flags = sc.get<u32>(N);
    launch_unique_flags(sorted[0], sorted[1], sorted[2], sorted[3], flags, N, s);
    const u64 total = compact_columns(*this, sc, sorted, flags, N, fresh_gen->idx[comp]);
    if (comp && total - ix.n != inserted) fail(RDFGPU_ERR_DEVICE, "extend: the permutations disagree on the number of new quads (%llu vs %llu)", (unsigned long long)(total - ix.n), (unsigned long long)inserted);
    inserted = total - ix.n;
  }
  adopt(fresh_gen);
  pool.trim();  // the load path's big temporaries go back to the driver
  return inserted;
}

u64 Store::extend_host(const u32* g, const u32* s_, const u32* p, const u32* o, u64 n) {
  activate();
  if (n == 0) return 0;
  u32* d[4];
  const u32* h[4] = {g, s_, p, o};
  for (int k = 0; k < 4; k++) {
    d[k] = (u32*)pool.alloc(n * 4);
    RDFGPU_HIP(hipMemcpy(d[k], h[k], n * 4, hipMemcpyHostToDevice));
  }
  u64 r = 0;
  try { r = extend_device(d[0], d[1], d[2], d[3], n); } catch (...) { for (auto q : d) pool.free(q); throw; }
  for (auto q : d) pool.free(q);
  pool.trim();
  return r;
}

// IndexPermutations::remove (permutations.rs:120-128): binary-search every quad in each permutation,
// clear its keep flag, compact.
u64 Store::remove_host(const u32* g, const u32* s_, const u32* p, const u32* o, u64 n) {
  std::unique_lock<std::shared_mutex> lock(mu);
  activate();
  if (n == 0 || idx[0].n == 0) return 0;
  hipStream_t s = stream;
  const u32* h[4] = {g, s_, p, o};
  u64 removed = 0;
  auto fresh_gen = std::make_shared<IndexGeneration>();
  fresh_gen->device = device;
  for (u32 comp = 0; comp < RDFGPU_N_INDEXES; comp++) {
    const Permutation& ix = idx[comp];
    Scratch sc(pool);
    u32* rm[4];
    for (int k = 0; k < 4; k++) {
      rm[k] = sc.get<u32>(n);
      RDFGPU_HIP(hipMemcpyAsync(rm[k], h[PERM[comp][k]], n * 4, hipMemcpyHostToDevice, s));
    }
    u32* keep = sc.get<u32>(ix.n);
    launch_fill_u32(keep, 1u, ix.n, s);
    const u32* ixc[4] = {ix.col[0], ix.col[1], ix.col[2], ix.col[3]};
    const u32* rmc[4] = {rm[0], rm[1], rm[2], rm[3]};
    launch_mark_removed(ixc, ix.n, rmc, n, keep, s);
    const u64 total = compact_columns(*this, sc, ix.col, keep, ix.n, fresh_gen->idx[comp]);
    if (comp && ix.n - total != removed) fail(RDFGPU_ERR_DEVICE, "remove: the permutations disagree on the number of removed quads");
    removed = ix.n - total;
  }
  adopt(fresh_gen);
  pool.trim();
  return removed;
}

// QuadStorage::clear_graph / drop_named_graph (lib/extensions/src/storage/quad_storage.rs:59-68): every quad of one graph
// (0 = the default graph).  The graph is level 0 of all three permutations: one compare per row, one compaction each.
u64 Store::remove_graph(u32 graph) {
  std::unique_lock<std::shared_mutex> lock(mu);
  activate();
  if (idx[0].n == 0) return 0;
  hipStream_t s = stream;
  u64 removed = 0;
  auto fresh_gen = std::make_shared<IndexGeneration>();
  fresh_gen->device = device;
  for (u32 comp = 0; comp < RDFGPU_N_INDEXES; comp++) {
    const Permutation& ix = idx[comp];
    Scratch sc(pool);
    u32* keep = sc.get<u32>(ix.n);
    launch_mark_not_equal(ix.col[0], graph, keep, ix.n, s);
    const u64 total = compact_columns(*this, sc, ix.col, keep, ix.n, fresh_gen->idx[comp]);
    if (comp && ix.n - total != removed) fail(RDFGPU_ERR_DEVICE, "remove_graph: the permutations disagree on the number of removed quads");
    removed = ix.n - total;
  }
  if (removed) adopt(fresh_gen);
  pool.trim();
  return removed;
}

void Store::set_typed_values(const rdfgpu_typed_value* v, u64 n, const int64_t* d, u64 nd) {
  std::unique_lock<std::shared_mutex> lock(mu);
  activate();
  drop_string_verdicts();
  drop_slice_tables();   // (decoded value tables depend on the typed values)
  if (tv) { RDFGPU_HIP(hipFree(tv)); tv = nullptr; }
  if (dec) { RDFGPU_HIP(hipFree(dec)); dec = nullptr; }
  n_ids = n; n_dec = nd;
  if (n) {
    RDFGPU_HIP(hipMalloc((void**)&tv, n * sizeof(rdfgpu_typed_value)));
    RDFGPU_HIP(hipMemcpy(tv, v, n * sizeof(rdfgpu_typed_value), hipMemcpyHostToDevice));
  }
  if (nd) {
    RDFGPU_HIP(hipMalloc((void**)&dec, nd * 16));
    RDFGPU_HIP(hipMemcpy(dec, d, nd * 16, hipMemcpyHostToDevice));
  }
}

void Store::set_strings(const u64* offsets, u64 n, const unsigned char* heap_host, u64 heap_bytes) {
  std::unique_lock<std::shared_mutex> lock(mu);
  activate();
  drop_string_verdicts();
  if (n && !offsets) fail(RDFGPU_ERR_INVALID, "set_strings: null offsets");
  for (u64 i = 0; i < n; i++) if (offsets[i] > offsets[i + 1] || offsets[i + 1] > heap_bytes) fail(RDFGPU_ERR_INVALID, "set_strings: offsets of id %llu are not monotone / inside the heap", (unsigned long long)i);
  if (str_off) { RDFGPU_HIP(hipFree(str_off)); str_off = nullptr; }
  if (heap) { RDFGPU_HIP(hipFree(heap)); heap = nullptr; }
  n_str_ids = n;
  if (!n) return;
  RDFGPU_HIP(hipMalloc((void**)&str_off, (n + 1) * sizeof(u64)));
  RDFGPU_HIP(hipMemcpy(str_off, offsets, (n + 1) * sizeof(u64), hipMemcpyHostToDevice));
  RDFGPU_HIP(hipMalloc((void**)&heap, heap_bytes ? heap_bytes : 1));
  if (heap_bytes) RDFGPU_HIP(hipMemcpy(heap, heap_host, heap_bytes, hipMemcpyHostToDevice));
}

}  // namespace rdfgpu
