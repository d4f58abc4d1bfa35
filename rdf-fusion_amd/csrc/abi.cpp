// abi.cpp — the extern "C" boundary of include/rdfgpu.h.  No exception crosses it.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>

#include "exchange.hpp"
#include "host_logic.hpp"
#include "ntriples.hpp"
#include "plan.hpp"
#include "regex_compile.hpp"
#include "store.hpp"

namespace rdfgpu {
static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
}  // namespace rdfgpu

using namespace rdfgpu;

#define ABI_BEGIN try {
#define ABI_END                                              \
  }                                                          \
  catch (const Error& e) { set_last_error(e.what()); return e.status; } \
  catch (const std::bad_alloc&) { set_last_error("host allocation failed"); return RDFGPU_ERR_OOM; } \
  catch (const std::exception& e) { set_last_error(e.what()); return RDFGPU_ERR_INVALID; } \
  return RDFGPU_OK;

static Store* S(rdfgpu_store* s) { if (!s) fail(RDFGPU_ERR_INVALID, "null store handle"); return reinterpret_cast<Store*>(s); }
static const Store* S(const rdfgpu_store* s) { if (!s) fail(RDFGPU_ERR_INVALID, "null store handle"); return reinterpret_cast<const Store*>(s); }
static Plan* P(rdfgpu_plan* p) { if (!p) fail(RDFGPU_ERR_INVALID, "null plan handle"); return reinterpret_cast<Plan*>(p); }

extern "C" {

const char* rdfgpu_last_error(void) { return g_last_error.c_str(); }
uint32_t rdfgpu_abi_version(void) { return RDFGPU_ABI_VERSION; }

// ---- store ---------------------------------------------------------------------------------------
int rdfgpu_store_create(const rdfgpu_config* cfg, rdfgpu_store** out) {
  ABI_BEGIN
  if (!out) fail(RDFGPU_ERR_INVALID, "null out pointer");
  *out = reinterpret_cast<rdfgpu_store*>(store_create(cfg));
  ABI_END
}
void rdfgpu_store_destroy(rdfgpu_store* store) { if (store) reinterpret_cast<Store*>(store)->release(); }

int rdfgpu_store_extend(rdfgpu_store* store, const uint32_t* g, const uint32_t* s, const uint32_t* p, const uint32_t* o, uint64_t n, uint64_t* inserted) {
  ABI_BEGIN
  if (n && (!g || !s || !p || !o)) fail(RDFGPU_ERR_INVALID, "null quad column");
  const u64 r = S(store)->extend_host(g, s, p, o, n);
  if (inserted) *inserted = r;
  ABI_END
}
int rdfgpu_store_extend_device(rdfgpu_store* store, const uint32_t* g, const uint32_t* s, const uint32_t* p, const uint32_t* o, uint64_t n, uint64_t* inserted) {
  ABI_BEGIN
  if (n && (!g || !s || !p || !o)) fail(RDFGPU_ERR_INVALID, "null quad column");
  const u64 r = S(store)->extend_device(g, s, p, o, n);
  if (inserted) *inserted = r;
  ABI_END
}
int rdfgpu_store_remove(rdfgpu_store* store, const uint32_t* g, const uint32_t* s, const uint32_t* p, const uint32_t* o, uint64_t n, uint64_t* removed) {
  ABI_BEGIN
  if (n && (!g || !s || !p || !o)) fail(RDFGPU_ERR_INVALID, "null quad column");
  const u64 r = S(store)->remove_host(g, s, p, o, n);
  if (removed) *removed = r;
  ABI_END
}
int rdfgpu_store_clear(rdfgpu_store* store) { ABI_BEGIN S(store)->clear(); ABI_END }
int rdfgpu_store_drop_tables(rdfgpu_store* store) { ABI_BEGIN S(store)->drop_tables(); ABI_END }
int rdfgpu_store_remove_graph(rdfgpu_store* store, uint32_t graph, uint64_t* removed) {
  ABI_BEGIN
  const u64 r = S(store)->remove_graph(graph);
  if (removed) *removed = r;
  ABI_END
}
int rdfgpu_store_len(const rdfgpu_store* store, uint64_t* out) {
  ABI_BEGIN
  if (!out) fail(RDFGPU_ERR_INVALID, "null out pointer");
  Store* st = const_cast<Store*>(S(store));
  std::shared_lock<std::shared_mutex> lock(st->mu);
  *out = st->idx[0].n;
  ABI_END
}
int rdfgpu_store_set_strings(rdfgpu_store* store, const uint64_t* offsets, uint64_t n_ids, const uint8_t* heap, uint64_t heap_bytes) {
  ABI_BEGIN
  if (!store) fail(RDFGPU_ERR_INVALID, "null store");
  S(store)->set_strings(offsets, n_ids, heap, heap_bytes);
  ABI_END
}

int rdfgpu_store_set_typed_values(rdfgpu_store* store, const rdfgpu_typed_value* values, uint64_t n_ids, const int64_t* decimals, uint64_t n_decimals) {
  ABI_BEGIN
  if ((n_ids && !values) || (n_decimals && !decimals)) fail(RDFGPU_ERR_INVALID, "null typed-value table");
  S(store)->set_typed_values(values, n_ids, decimals, n_decimals);
  ABI_END
}
int rdfgpu_store_read_index(const rdfgpu_store* store, uint32_t components, uint32_t* c0, uint32_t* c1, uint32_t* c2, uint32_t* c3, uint64_t cap, uint64_t* n) {
  ABI_BEGIN
  const Store* st = S(store);
  if (components >= RDFGPU_N_INDEXES) fail(RDFGPU_ERR_INVALID, "bad index components %u", components);
  st->activate();
  std::shared_lock<std::shared_mutex> lock(const_cast<Store*>(st)->mu);
  const Permutation& ix = st->idx[components];
  if (n) *n = ix.n;
  uint32_t* dst[4] = {c0, c1, c2, c3};
  const u64 m = ix.n < cap ? ix.n : cap;
  for (int k = 0; k < 4; k++) if (dst[k] && m) RDFGPU_HIP(hipMemcpy(dst[k], ix.col[k], m * 4, hipMemcpyDeviceToHost));
  ABI_END
}

// ---- engine options --------------------------------------------------------------------------------
int rdfgpu_store_set_option(rdfgpu_store* store, uint32_t option, uint64_t value) {
  ABI_BEGIN
  if (option >= RDFGPU_OPT__COUNT) fail(RDFGPU_ERR_INVALID, "unknown engine option %u", option);
  Store* st = S(store);
  std::unique_lock<std::shared_mutex> lock(st->mu);   // no plan of this store is executing
  st->opt.v[option] = value;
  ABI_END
}
int rdfgpu_store_get_option(const rdfgpu_store* store, uint32_t option, uint64_t* value) {
  ABI_BEGIN
  if (option >= RDFGPU_OPT__COUNT || !value) fail(RDFGPU_ERR_INVALID, "unknown engine option %u", option);
  *value = S(store)->opt.v[option];
  ABI_END
}
int rdfgpu_plan_set_option(rdfgpu_plan* plan, uint32_t option, uint64_t value) {
  ABI_BEGIN
  if (option >= RDFGPU_OPT__COUNT) fail(RDFGPU_ERR_INVALID, "unknown engine option %u", option);
  P(plan)->opt.v[option] = value;   // one in-flight call per handle: not concurrent with an execute of this plan
  ABI_END
}
const char* rdfgpu_option_name(uint32_t option) { return engine_option_name(option); }

// ---- plans ---------------------------------------------------------------------------------------
int rdfgpu_plan_compile(rdfgpu_store* store, const rdfgpu_plan_desc* desc, rdfgpu_plan** out) {
  ABI_BEGIN
  if (!out) fail(RDFGPU_ERR_INVALID, "null out pointer");
  *out = reinterpret_cast<rdfgpu_plan*>(plan_compile(S(store), desc));
  ABI_END
}
void rdfgpu_plan_destroy(rdfgpu_plan* plan) { delete reinterpret_cast<Plan*>(plan); }

int rdfgpu_plan_bind_table(rdfgpu_plan* plan, uint32_t slot, const uint32_t* const* cols, uint32_t n_cols, uint64_t n_rows) {
  ABI_BEGIN
  Plan* p = P(plan);
  if (slot >= p->tables.size()) fail(RDFGPU_ERR_INVALID, "plan has no table slot %u", slot);
  if (n_cols && !cols) fail(RDFGPU_ERR_INVALID, "null column array");
  BoundTable& b = p->tables[slot];
  b.cols.assign(cols, cols + n_cols);
  b.n_rows = n_rows; b.bound = true;
  ABI_END
}
int rdfgpu_plan_execute(rdfgpu_plan* plan) { ABI_BEGIN P(plan)->execute(); ABI_END }

int rdfgpu_plan_result_info(rdfgpu_plan* plan, uint64_t* n_rows, uint32_t* n_cols) {
  ABI_BEGIN
  Plan* p = P(plan);
  if (!p->executed) fail(RDFGPU_ERR_INVALID, "plan has not been executed");
  if (n_rows) *n_rows = p->result_rows;
  if (n_cols) *n_cols = p->result.n_cols;
  ABI_END
}
int rdfgpu_plan_result_device(rdfgpu_plan* plan, const uint32_t** cols, uint32_t cap_cols) {
  ABI_BEGIN
  Plan* p = P(plan);
  if (!p->executed) fail(RDFGPU_ERR_INVALID, "plan has not been executed");
  if (cap_cols < p->result.n_cols) fail(RDFGPU_ERR_INVALID, "room for %u columns, result has %u", cap_cols, p->result.n_cols);
  for (u32 c = 0; c < p->result.n_cols; c++) cols[c] = p->result_rows ? p->result.cols[c] : nullptr;
  ABI_END
}
int rdfgpu_plan_fetch(rdfgpu_plan* plan, uint32_t* const* host_cols, uint32_t n_cols) {
  ABI_BEGIN
  Plan* p = P(plan);
  if (!p->executed) fail(RDFGPU_ERR_INVALID, "plan has not been executed");
  if (n_cols != p->result.n_cols) fail(RDFGPU_ERR_INVALID, "caller passed %u columns, result has %u", n_cols, p->result.n_cols);
  p->store->activate();
  for (u32 c = 0; c < n_cols; c++)
    if (p->result_rows) RDFGPU_HIP(hipMemcpyAsync(host_cols[c], p->result.cols[c], p->result_rows * 4, hipMemcpyDeviceToHost, p->stream));
  RDFGPU_HIP(hipStreamSynchronize(p->stream));
  ABI_END
}

// Arrow C Data Interface export -----------------------------------------------------------------
namespace {
struct ChildPriv { void* buffers[3] = {nullptr, nullptr, nullptr}; };
void release_child_array(struct ArrowArray* a) {
  if (!a || !a->release) return;
  ChildPriv* pr = static_cast<ChildPriv*>(a->private_data);
  std::free(pr->buffers[0]); std::free(pr->buffers[1]); std::free(pr->buffers[2]);
  delete pr;
  a->release = nullptr;
}
struct StructPriv { const void* buffers[1]; struct ArrowArray** children; u32 n; };
void release_struct_array(struct ArrowArray* a) {
  if (!a || !a->release) return;
  StructPriv* pr = static_cast<StructPriv*>(a->private_data);
  for (u32 i = 0; i < pr->n; i++) { if (pr->children[i]->release) pr->children[i]->release(pr->children[i]); delete pr->children[i]; }
  delete[] pr->children;
  delete pr;
  a->release = nullptr;
}
struct SchemaPriv { struct ArrowSchema** children; u32 n; std::string* names; };
void release_schema(struct ArrowSchema* s) {
  if (!s || !s->release) return;
  SchemaPriv* pr = static_cast<SchemaPriv*>(s->private_data);
  if (pr) {
    for (u32 i = 0; i < pr->n; i++) { if (pr->children[i]->release) pr->children[i]->release(pr->children[i]); delete pr->children[i]; }
    delete[] pr->children; delete[] pr->names; delete pr;
  }
  s->release = nullptr;
}
}  // namespace

int rdfgpu_plan_next(rdfgpu_plan* plan, struct ArrowArray* out, struct ArrowSchema* schema) {
  try {
    Plan* p = P(plan);
    if (!out) fail(RDFGPU_ERR_INVALID, "null out array");
    p->ensure_host_copy();
    if (p->cursor >= p->result_rows) return RDFGPU_END;   // never an empty batch (scan.rs:195-198)
    const u64 len = std::min<u64>(p->store->batch_size, p->result_rows - p->cursor);
    const u32 nc = p->result.n_cols;
    StructPriv* sp = new StructPriv();
    sp->buffers[0] = nullptr; sp->n = nc; sp->children = new ArrowArray*[nc ? nc : 1];
    for (u32 c = 0; c < nc; c++) {
      ArrowArray* ch = new ArrowArray();
      std::memset(ch, 0, sizeof *ch);
      ChildPriv* cp = new ChildPriv();
      const u32* src = p->host_cols[c].data() + p->cursor;
      u32* data = static_cast<u32*>(std::malloc(len * 4 ? len * 4 : 4));
      std::memcpy(data, src, len * 4);
      int64_t nulls = 0;
      for (u64 i = 0; i < len; i++) nulls += src[i] == 0;
      uint8_t* valid = nullptr;
      if (nulls) {   // object id 0 is the null marker (quad_index_data.rs:438-440)
        valid = static_cast<uint8_t*>(std::calloc((len + 7) / 8, 1));
        for (u64 i = 0; i < len; i++) if (src[i]) valid[i >> 3] |= (uint8_t)(1u << (i & 7));
      }
      cp->buffers[0] = valid; cp->buffers[1] = data;
      ch->length = (int64_t)len; ch->null_count = nulls; ch->n_buffers = 2;
      ch->buffers = const_cast<const void**>(reinterpret_cast<void**>(cp->buffers));
      ch->release = release_child_array; ch->private_data = cp;
      sp->children[c] = ch;
    }
    std::memset(out, 0, sizeof *out);
    out->length = (int64_t)len; out->n_buffers = 1; out->buffers = sp->buffers;
    out->n_children = nc; out->children = sp->children;
    out->release = release_struct_array; out->private_data = sp;
    if (schema) {
      std::memset(schema, 0, sizeof *schema);
      SchemaPriv* pr = new SchemaPriv();
      pr->n = nc; pr->children = new ArrowSchema*[nc ? nc : 1]; pr->names = new std::string[nc ? nc : 1];
      for (u32 c = 0; c < nc; c++) {
        ArrowSchema* cs = new ArrowSchema();
        std::memset(cs, 0, sizeof *cs);
        pr->names[c] = "c" + std::to_string(c);
        cs->format = "I"; cs->name = pr->names[c].c_str(); cs->flags = 2 /* ARROW_FLAG_NULLABLE */;
        cs->release = release_schema; cs->private_data = nullptr;
        pr->children[c] = cs;
      }
      schema->format = "+s"; schema->name = ""; schema->n_children = nc; schema->children = pr->children;
      schema->release = release_schema; schema->private_data = pr;
    }
    p->cursor += len;
    return RDFGPU_OK;
  } catch (const Error& e) { set_last_error(e.what()); return e.status; }
  catch (const std::exception& e) { set_last_error(e.what()); return RDFGPU_ERR_INVALID; }
}
int rdfgpu_plan_decode_terms(rdfgpu_plan* plan, uint32_t col, uint64_t first_row, uint64_t n_rows, struct ArrowArray* out, struct ArrowSchema* schema) {
  try {
    Plan* p = P(plan);
    if (!out) fail(RDFGPU_ERR_INVALID, "null out array");
    if (!p->executed) fail(RDFGPU_ERR_INVALID, "plan has not been executed");
    if (col >= p->result.n_cols) fail(RDFGPU_ERR_INVALID, "column %u of a %u-column result", col, p->result.n_cols);
    if (first_row > p->result_rows || n_rows > p->result_rows - first_row) fail(RDFGPU_ERR_INVALID, "rows %llu .. + %llu of a %llu-row result", (unsigned long long)first_row, (unsigned long long)n_rows, (unsigned long long)p->result_rows);
    Store* st = p->store;
    st->activate();
    std::shared_lock<std::shared_mutex> lock(st->mu);          // the dictionary tables must not be replaced underneath
    if (!st->str_off) fail(RDFGPU_ERR_INVALID, "ENC_PT needs the lexical forms (rdfgpu_store_set_strings)");
    const u64 n = n_rows;
    hipStream_t s = p->stream;
    DevicePool& pool = st->pool;
    std::vector<void*> dev;
    auto dalloc = [&](size_t bytes) { void* q = pool.alloc(bytes ? bytes : 4); dev.push_back(q); return q; };
    struct Guard { DevicePool& pool; std::vector<void*>& v; ~Guard() { for (void* q : v) pool.free(q); } } guard{pool, dev};
    DecodeArgs a{};
    a.ids = p->result.cols[col] + first_row; a.n = n; a.tt = st->typed_table();
    u32* len = static_cast<u32*>(dalloc((n + 1) * 4)); u32* off = static_cast<u32*>(dalloc((n + 1) * 4));
    a.len = len; a.off = off;
    a.term_type = static_cast<unsigned char*>(dalloc(n)); a.tag = static_cast<unsigned char*>(dalloc(n)); a.aux = static_cast<u32*>(dalloc(n * 4));
    RDFGPU_HIP(hipMemsetAsync(len + n, 0, 4, s));
    launch_decode_lengths(a, s);
    // the total first (64-bit on the host side: utf8 offsets are int32)
    std::vector<u32> h_len(n + 1);
    RDFGPU_HIP(hipMemcpyAsync(h_len.data(), len, (n + 1) * 4, hipMemcpyDeviceToHost, s));
    RDFGPU_HIP(hipStreamSynchronize(s));
    u64 total = 0;
    for (u64 i = 0; i < n; i++) total += h_len[i];
    if (total > 0x7FFFFFFFull) fail(RDFGPU_ERR_UNSUPPORTED, "ENC_PT: %llu bytes of lexical forms in one call (utf8 offsets are int32): decode fewer rows", (unsigned long long)total);
    const size_t tb = scan_temp_bytes(n + 1);
    void* temp = dalloc(tb);
    exclusive_scan_u32(len, off, n + 1, temp, tb, s);
    a.bytes = static_cast<unsigned char*>(dalloc(total));
    launch_decode_bytes(a, s);
    // host side of the Arrow arrays
    auto halloc = [](size_t bytes) { void* q = std::malloc(bytes ? bytes : 1); if (!q) fail(RDFGPU_ERR_OOM, "host allocation of %zu bytes failed", bytes); return q; };
    unsigned char* h_tt = static_cast<unsigned char*>(halloc(n)); unsigned char* h_tag = static_cast<unsigned char*>(halloc(n));
    u32* h_aux = static_cast<u32*>(halloc(n * 4)); int32_t* h_off = static_cast<int32_t*>(halloc((n + 1) * 4));
    char* h_bytes = static_cast<char*>(halloc(total));
    if (n) {
      RDFGPU_HIP(hipMemcpyAsync(h_tt, a.term_type, n, hipMemcpyDeviceToHost, s));
      RDFGPU_HIP(hipMemcpyAsync(h_tag, a.tag, n, hipMemcpyDeviceToHost, s));
      RDFGPU_HIP(hipMemcpyAsync(h_aux, a.aux, n * 4, hipMemcpyDeviceToHost, s));
    }
    RDFGPU_HIP(hipMemcpyAsync(h_off, off, (n + 1) * 4, hipMemcpyDeviceToHost, s));
    if (total) RDFGPU_HIP(hipMemcpyAsync(h_bytes, a.bytes, total, hipMemcpyDeviceToHost, s));
    RDFGPU_HIP(hipStreamSynchronize(s));
    int64_t nulls = 0;
    uint8_t* valid = static_cast<uint8_t*>(std::calloc((n + 7) / 8 + 1, 1));
    for (u64 i = 0; i < n; i++) { if (h_tt[i] == 0xFF) { nulls++; h_tt[i] = 0; } else valid[i >> 3] |= (uint8_t)(1u << (i & 7)); }
    StructPriv* sp = new StructPriv();
    sp->n = 4; sp->children = new ArrowArray*[4];
    auto child = [&](void* b1, void* b2, int n_buffers) {
      ArrowArray* ch = new ArrowArray();
      std::memset(ch, 0, sizeof *ch);
      ChildPriv* cp = new ChildPriv();
      cp->buffers[0] = nullptr; cp->buffers[1] = b1; cp->buffers[2] = b2;
      ch->length = (int64_t)n; ch->null_count = 0; ch->n_buffers = n_buffers;
      ch->buffers = const_cast<const void**>(reinterpret_cast<void**>(cp->buffers));
      ch->release = release_child_array; ch->private_data = cp;
      return ch;
    };
    sp->children[0] = child(h_tt, nullptr, 2);
    sp->children[1] = child(h_off, h_bytes, 3);
    sp->children[2] = child(h_tag, nullptr, 2);
    sp->children[3] = child(h_aux, nullptr, 2);
    // the struct's validity bitmap is owned through the first child's spare slot
    static_cast<ChildPriv*>(sp->children[0]->private_data)->buffers[2] = valid;
    sp->buffers[0] = nulls ? valid : nullptr;
    std::memset(out, 0, sizeof *out);
    out->length = (int64_t)n; out->null_count = nulls; out->n_buffers = 1; out->buffers = sp->buffers;
    out->n_children = 4; out->children = sp->children;
    out->release = release_struct_array; out->private_data = sp;
    if (schema) {
      std::memset(schema, 0, sizeof *schema);
      SchemaPriv* pr = new SchemaPriv();
      pr->n = 4; pr->children = new ArrowSchema*[4]; pr->names = new std::string[4];
      const char* names[4] = {"term_type", "value", "tag", "aux"};
      const char* formats[4] = {"C", "u", "C", "I"};
      for (u32 c = 0; c < 4; c++) {
        ArrowSchema* cs = new ArrowSchema();
        std::memset(cs, 0, sizeof *cs);
        pr->names[c] = names[c];
        cs->format = formats[c]; cs->name = pr->names[c].c_str(); cs->flags = 0;
        cs->release = release_schema; cs->private_data = nullptr;
        pr->children[c] = cs;
      }
      schema->format = "+s"; schema->name = ""; schema->flags = 2 /* ARROW_FLAG_NULLABLE */; schema->n_children = 4; schema->children = pr->children;
      schema->release = release_schema; schema->private_data = pr;
    }
    return RDFGPU_OK;
  } catch (const Error& e) { set_last_error(e.what()); return e.status; }
  catch (const std::exception& e) { set_last_error(e.what()); return RDFGPU_ERR_INVALID; }
}
int rdfgpu_plan_rewind(rdfgpu_plan* plan) { ABI_BEGIN P(plan)->cursor = 0; ABI_END }
int rdfgpu_plan_metrics(rdfgpu_plan* plan, rdfgpu_metrics* out) {
  ABI_BEGIN
  if (!out) fail(RDFGPU_ERR_INVALID, "null out pointer");
  *out = P(plan)->metrics;
  ABI_END
}
int rdfgpu_plan_selected_index(const rdfgpu_plan* plan, uint32_t node, uint32_t* components) {
  ABI_BEGIN
  const Plan* p = reinterpret_cast<const Plan*>(plan);
  if (!p) fail(RDFGPU_ERR_INVALID, "null plan handle");
  if (node >= p->nodes.size() || p->nodes[node].source < 0) fail(RDFGPU_ERR_INVALID, "node %u is not a data source", node);
  if (components) *components = p->sources[p->nodes[node].source].components;
  ABI_END
}
int rdfgpu_plan_pushdown_filters(rdfgpu_plan* plan, uint32_t node, const rdfgpu_pushdown_filter* filters, uint32_t n, uint8_t* pushed) {
  ABI_BEGIN
  if (n && !filters) fail(RDFGPU_ERR_INVALID, "null filter array");
  P(plan)->pushdown_filters(node, filters, n, pushed);
  ABI_END
}
int rdfgpu_plan_set_dynamic_filters(rdfgpu_plan* plan, uint32_t node, const rdfgpu_pushdown_filter* filters, uint32_t n) {
  ABI_BEGIN
  if (n && !filters) fail(RDFGPU_ERR_INVALID, "null filter array");
  P(plan)->set_dynamic_filters(node, filters, n);
  ABI_END
}
int rdfgpu_plan_source_predicate(const rdfgpu_plan* plan, uint32_t node, uint32_t level, rdfgpu_predicate* pred) {
  ABI_BEGIN
  const Plan* p = reinterpret_cast<const Plan*>(plan);
  if (!p || !pred) fail(RDFGPU_ERR_INVALID, "null pointer");
  if (node >= p->nodes.size() || p->nodes[node].source < 0 || level > 3) fail(RDFGPU_ERR_INVALID, "node %u level %u is not a data-source level", node, level);
  const SourceInfo& src = p->sources[p->nodes[node].source];
  // the instructions as they will be scanned: in the chosen index's order; report by G,S,P,O level
  ScanPredicate sp;
  for (int k = 0; k < 4; k++) if ((u32)PERM[src.components][k] == level) sp = src.ix.in[k].pred;
  std::memset(pred, 0, sizeof *pred);
  pred->pred = sp.kind; pred->from = sp.from; pred->to = sp.to; pred->equal_to = sp.equal_to;
  if (sp.kind == RDFGPU_PRED_IN) { pred->n_ids = (u32)sp.ids.size(); if (sp.ids.size() == 1) pred->from = pred->to = sp.ids[0]; }
  ABI_END
}
int rdfgpu_plan_stream(rdfgpu_plan* plan, void** hip_stream) {
  ABI_BEGIN
  if (!hip_stream) fail(RDFGPU_ERR_INVALID, "null out pointer");
  *hip_stream = P(plan)->stream;
  ABI_END
}

int rdfgpu_plan_enable_kernel_timing(rdfgpu_plan* plan, int on) {
  ABI_BEGIN
  Plan* p = P(plan);
  p->timing = on != 0;
  p->timing_focus = on >= 2 ? p->last_top_kc : -1;   // 2: only the class that took longest when every launch was timed (all of them if that never happened)
  ABI_END
}
int rdfgpu_plan_kernel_stats(rdfgpu_plan* plan, rdfgpu_kernel_stat* out, uint32_t cap, uint32_t* n) {
  ABI_BEGIN
  Plan* p = P(plan);
  if (!n || (cap && !out)) fail(RDFGPU_ERR_INVALID, "null out pointer");
  u32 w = 0;
  for (int k = 0; k < KC__N; k++) {
    const KernelStat& s = p->kstats[k];
    if (!s.launches) continue;
    if (w < cap) { out[w].kernel = kernel_class_name(k); out[w].launches = s.launches; out[w].reserved = 0; out[w].total_ms = s.ms; out[w].algorithmic_bytes = s.bytes; out[w].rows_in = s.rows; }
    w++;
  }
  *n = w < cap ? w : cap;
  ABI_END
}

// ---- host logic ----------------------------------------------------------------------------------
static void decode4(const rdfgpu_scan_instruction raw[4], ScanInstruction out[4]) {
  // host-logic entry points take explicit (pred, a, b) without a pool: an IN with b ids is scored as
  // b synthetic ids a..a+b-1 (only "one id or several" matters to the score)
  for (int i = 0; i < 4; i++) {
    rdfgpu_scan_instruction r = raw[i];
    if (r.pred == RDFGPU_PRED_IN) {
      out[i].kind = r.kind; out[i].var = r.var; out[i].pred.kind = RDFGPU_PRED_IN;
      out[i].pred.ids.clear();
      for (u32 q = 0; q < (r.b ? r.b : 1); q++) out[i].pred.ids.push_back(q);
    } else out[i] = decode_instruction(r, nullptr, 0);
  }
}
uint64_t rdfgpu_scan_score(const rdfgpu_scan_instruction instr[4]) {
  try { ScanInstruction in[4]; decode4(instr, in); return scan_score(in); }
  catch (const std::exception& e) { set_last_error(e.what()); return 0; }
}
uint32_t rdfgpu_choose_index(const rdfgpu_scan_instruction gspo[4], uint32_t available) {
  try {
    ScanInstructions g; g.components = RDFGPU_GSPO; decode4(gspo, g.in);
    return choose_index(g, available);
  } catch (const std::exception& e) { set_last_error(e.what()); return 0xFFFFFFFFu; }
}
static ScanPredicate from_abi(const rdfgpu_predicate* p) {
  ScanPredicate r; r.kind = p->pred; r.from = p->from; r.to = p->to; r.equal_to = p->equal_to;
  if (p->pred == RDFGPU_PRED_IN) { if (p->ids) r.ids.assign(p->ids, p->ids + p->n_ids); else r.ids = {p->from}; }
  return r;
}
int rdfgpu_predicate_and(const rdfgpu_predicate* lhs, const rdfgpu_predicate* rhs, rdfgpu_predicate* out, uint32_t* out_ids) {
  try {
    if (!lhs || !rhs || !out) fail(RDFGPU_ERR_INVALID, "null predicate");
    ScanPredicate r;
    if (!predicate_and(from_abi(lhs), from_abi(rhs), &r)) return 0;
    std::memset(out, 0, sizeof *out);
    out->pred = r.kind; out->from = r.from; out->to = r.to;
    if (r.kind == RDFGPU_PRED_IN) {
      if (!out_ids) fail(RDFGPU_ERR_INVALID, "null out_ids");
      for (size_t i = 0; i < r.ids.size(); i++) out_ids[i] = r.ids[i];
      out->ids = out_ids; out->n_ids = (u32)r.ids.size();
    }
    return 1;
  } catch (const Error& e) { set_last_error(e.what()); return e.status; }
}
int rdfgpu_pushdown_to_scan_predicate(uint32_t op, uint32_t value, rdfgpu_predicate* out) {
  try {
    if (!out) fail(RDFGPU_ERR_INVALID, "null out predicate");
    const ScanPredicate r = pushdown_to_scan_predicate(op, value);
    std::memset(out, 0, sizeof *out);
    out->pred = r.kind; out->from = r.from; out->to = r.to;
    if (r.kind == RDFGPU_PRED_IN) { out->from = out->to = r.ids[0]; out->n_ids = 1; }
    return 1;
  } catch (const Error& e) { set_last_error(e.what()); return e.status; }
}

// ---- multi-GPU exchange ----------------------------------------------------------------------------
static Comm* CM(rdfgpu_comm* c) { if (!c) fail(RDFGPU_ERR_INVALID, "null communicator"); return reinterpret_cast<Comm*>(c); }
int rdfgpu_comm_unique_id(uint8_t id[RDFGPU_COMM_ID_BYTES]) { ABI_BEGIN if (!id) fail(RDFGPU_ERR_INVALID, "null id"); comm_unique_id(id); ABI_END }
int rdfgpu_comm_create(const uint8_t id[RDFGPU_COMM_ID_BYTES], uint32_t rank, uint32_t world, int32_t device, rdfgpu_comm** out) {
  ABI_BEGIN
  if (!id || !out) fail(RDFGPU_ERR_INVALID, "null pointer");
  *out = reinterpret_cast<rdfgpu_comm*>(comm_create_rccl(id, rank, world, device));
  ABI_END
}
int rdfgpu_comm_create_host(uint32_t rank, uint32_t world, int32_t device, rdfgpu_host_alltoallv_fn fn, void* ctx, rdfgpu_comm** out) {
  ABI_BEGIN
  if (!out) fail(RDFGPU_ERR_INVALID, "null out pointer");
  *out = reinterpret_cast<rdfgpu_comm*>(comm_create_host(rank, world, device, fn, ctx));
  ABI_END
}
void rdfgpu_comm_destroy(rdfgpu_comm* comm) { if (comm) comm_destroy(reinterpret_cast<Comm*>(comm)); }
int rdfgpu_exchange_allgatherv(rdfgpu_comm* comm, const uint32_t* const* cols, uint32_t n_cols, uint64_t n_rows, const uint32_t** out_cols, uint64_t* out_rows) {
  ABI_BEGIN
  if (!cols || !out_cols || !out_rows) fail(RDFGPU_ERR_INVALID, "null pointer");
  *out_rows = exchange_allgatherv(CM(comm), cols, n_cols, n_rows, out_cols);
  ABI_END
}
int rdfgpu_exchange_repartition(rdfgpu_comm* comm, const uint32_t* const* cols, uint32_t n_cols, uint64_t n_rows, uint32_t key_col, const uint32_t** out_cols, uint64_t* out_rows) {
  ABI_BEGIN
  if (!cols || !out_cols || !out_rows) fail(RDFGPU_ERR_INVALID, "null pointer");
  *out_rows = exchange_repartition(CM(comm), cols, n_cols, n_rows, key_col, out_cols);
  ABI_END
}
uint32_t rdfgpu_shard_of(uint32_t id, uint32_t world) {
  if (world == 0) return 0;
  return (uint32_t)((((unsigned long long)id * 0x9E3779B97F4A7C15ull) >> 40) % world);
}

int rdfgpu_regex_check(const char* pattern, uint32_t pattern_len, const char* flags, uint32_t flags_len, uint32_t* positions) {
  try {
    RegexProg prog; std::string why;
    if (regex_compile(pattern ? pattern : "", pattern_len, flags ? flags : "", flags ? flags_len : 0, prog, why) != REGEX_OK)
      fail(RDFGPU_ERR_UNSUPPORTED, "REGEX pattern: %s", why.c_str());
    if (positions) *positions = prog.n_pos;
    return RDFGPU_OK;
  } catch (const Error& e) { set_last_error(e.what()); return e.status; }
}

int rdfgpu_ntriples_parse(int32_t device, const char* text, uint64_t text_bytes, uint32_t first_id, rdfgpu_ntriples** out) {
  try {
    if (!out) fail(RDFGPU_ERR_INVALID, "null out pointer");
    if (!text && text_bytes) fail(RDFGPU_ERR_INVALID, "null text");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); fail(RDFGPU_ERR_NO_DEVICE, "no usable HIP device; this library has no CPU fallback"); }
    *out = reinterpret_cast<rdfgpu_ntriples*>(ntriples_parse(device, text, text_bytes, first_id));
    return RDFGPU_OK;
  } catch (const Error& e) { set_last_error(e.what()); return e.status; }
  catch (const std::exception& e) { set_last_error(e.what()); return RDFGPU_ERR_INVALID; }
}
int rdfgpu_ntriples_info(const rdfgpu_ntriples* nt, uint64_t* n_triples, uint32_t* n_terms, uint64_t* term_bytes) {
  ABI_BEGIN
  if (!nt) fail(RDFGPU_ERR_INVALID, "null handle");
  const NTriples* t = reinterpret_cast<const NTriples*>(nt);
  if (n_triples) *n_triples = t->n_triples;
  if (n_terms) *n_terms = t->n_terms;
  if (term_bytes) *term_bytes = t->term_total;
  ABI_END
}
int rdfgpu_ntriples_terms(const rdfgpu_ntriples* nt, uint64_t* offsets, uint8_t* bytes) {
  ABI_BEGIN
  if (!nt) fail(RDFGPU_ERR_INVALID, "null handle");
  ntriples_terms(reinterpret_cast<const NTriples*>(nt), offsets, bytes);
  ABI_END
}
int rdfgpu_ntriples_decoded_info(const rdfgpu_ntriples* nt, uint64_t* lex_bytes, uint64_t* suffix_bytes) {
  ABI_BEGIN
  if (!nt) fail(RDFGPU_ERR_INVALID, "null handle");
  const NTriples* t = reinterpret_cast<const NTriples*>(nt);
  if (lex_bytes) *lex_bytes = t->lex_total;
  if (suffix_bytes) *suffix_bytes = t->sfx_total;
  ABI_END
}
int rdfgpu_ntriples_decoded(const rdfgpu_ntriples* nt, uint8_t* kind, uint64_t* lex_off, uint8_t* lex_bytes, uint64_t* suffix_off,
                            uint8_t* suffix_bytes, rdfgpu_typed_value* typed, int64_t* dec_hi) {
  ABI_BEGIN
  if (!nt) fail(RDFGPU_ERR_INVALID, "null handle");
  ntriples_decoded(reinterpret_cast<const NTriples*>(nt), kind, lex_off, lex_bytes, suffix_off, suffix_bytes, typed, dec_hi);
  ABI_END
}
int rdfgpu_ntriples_columns(const rdfgpu_ntriples* nt, const uint32_t** s, const uint32_t** p, const uint32_t** o) {
  ABI_BEGIN
  if (!nt) fail(RDFGPU_ERR_INVALID, "null handle");
  const NTriples* t = reinterpret_cast<const NTriples*>(nt);
  if (s) *s = t->s;
  if (p) *p = t->p;
  if (o) *o = t->o;
  ABI_END
}
void rdfgpu_ntriples_destroy(rdfgpu_ntriples* nt) { delete reinterpret_cast<NTriples*>(nt); }

}  // extern "C"
