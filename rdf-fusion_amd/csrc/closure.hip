// closure.hip — KleenePlusClosureExec on the device.
//
// Reference: lib/physical/src/paths/kleene_plus/physical.rs:246-384 — collect the inner paths (graph, start, end), then
// semi-naive iteration: every path (g, a, b) of the current delta is extended by every INITIAL path (g', b, c) — g' = g,
// or any graph when allow_cross_graph_paths — to (g, a, c); paths not seen before form the next delta; the closure is
// the SET of all paths seen.  The reference runs this on decoded terms in hash sets; here ids are renumbered densely
// (sorted dictionaries of the graphs and nodes that occur), a path becomes ONE u64 key (graph | start | end), and an
// iteration is: range look-up in the sorted initial edges -> exclusive scan -> expand -> radix sort -> unique ->
// anti-join against the sorted closure (binary search) -> merge.  All sorted-array work is rocPRIM; the kernels here
// are the encode / look-up / expand / membership / decode steps (coalesced u64 streams, binary searches in L2).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <functional>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace rdfgpu {
namespace {

constexpr int kCB = 256;
inline dim3 cgrid(u64 n) { return dim3((unsigned)((n + kCB - 1) / kCB ? (n + kCB - 1) / kCB : 1)); }

struct DBuf {   // device temporary, freed on scope exit (also when an error is thrown)
  void* p = nullptr;
  DBuf() = default;
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  ~DBuf() { if (p) (void)hipFree(p); }
  template <class T> T* alloc(u64 n) { if (p) { (void)hipFree(p); p = nullptr; } RDFGPU_HIP(hipMalloc(&p, (n ? n : 1) * sizeof(T))); return static_cast<T*>(p); }
  template <class T> T* get() const { return static_cast<T*>(p); }
  void swap(DBuf& o) { void* t = p; p = o.p; o.p = t; }
};

__device__ __forceinline__ u32 lower_bound_u32(const u32* a, u32 n, u32 v) {
  u32 b = 0, e = n;
  while (b < e) { const u32 m = b + ((e - b) >> 1); if (a[m] < v) b = m + 1; else e = m; }
  return b;
}
__device__ __forceinline__ u64 lower_bound_u64(const u64* a, u64 n, u64 v) {
  u64 b = 0, e = n;
  while (b < e) { const u64 m = b + ((e - b) >> 1); if (a[m] < v) b = m + 1; else e = m; }
  return b;
}

__global__ __launch_bounds__(kCB) void concat2_kernel(const u32* a, const u32* b, u64 n, u32* out) {
  const u64 i = (u64)blockIdx.x * kCB + threadIdx.x;
  if (i < n) { out[i] = a[i]; out[n + i] = b[i]; }
}

// (graph, start, end) -> key = gi << 2vb | si << vb | ei with gi / si / ei the ranks in the sorted dictionaries
__global__ __launch_bounds__(kCB) void encode_kernel(const u32* g, const u32* s, const u32* e, u64 n, const u32* gdict, u32 ng,
                                                     const u32* vdict, u32 nv, u32 vb, u64* keys, u32* null_seen) {
  const u64 i = (u64)blockIdx.x * kCB + threadIdx.x;
  if (i >= n) return;
  const u32 sv = s[i], ev = e[i];
  if (sv == 0 || ev == 0) { *null_seen = 1; keys[i] = 0; return; }
  const u64 gi = lower_bound_u32(gdict, ng, g[i]), si = lower_bound_u32(vdict, nv, sv), ei = lower_bound_u32(vdict, nv, ev);
  keys[i] = (gi << (2 * vb)) | (si << vb) | ei;
}

// cross-graph continuation edges: the key without its graph
__global__ __launch_bounds__(kCB) void strip_graph_kernel(const u64* keys, u64 n, u32 vb, u64* out) {
  const u64 i = (u64)blockIdx.x * kCB + threadIdx.x;
  if (i < n) out[i] = keys[i] & ((1ull << (2 * vb)) - 1ull);
}

// per delta path (g, a, b): the initial edges leaving b — in graph g (adj keyed (g, s, e)) or in any graph (adj keyed (s, e))
__global__ __launch_bounds__(kCB) void range_kernel(const u64* delta, u64 nd, const u64* adj, u64 na, u32 vb, int cross, u64* lo, u32* cnt) {
  const u64 i = (u64)blockIdx.x * kCB + threadIdx.x;
  if (i >= nd) return;
  const u64 k = delta[i];
  const u64 vmask = (1ull << vb) - 1ull;
  const u64 b = k & vmask, gi = k >> (2 * vb);
  const u64 first = cross ? (b << vb) : ((gi << (2 * vb)) | (b << vb));
  const u64 l = lower_bound_u64(adj, na, first), h = lower_bound_u64(adj, na, first + (1ull << vb));   // b + 1 in the start field (no carry: b <= vmask)
  lo[i] = l;
  cnt[i] = (u32)(h - l);
}

__global__ __launch_bounds__(kCB) void expand_kernel(const u64* delta, u64 nd, const u64* adj, u32 vb, const u64* lo, const u32* cnt, const u64* off, u64* cand) {
  const u64 i = (u64)blockIdx.x * kCB + threadIdx.x;
  if (i >= nd) return;
  const u64 vmask = (1ull << vb) - 1ull;
  const u64 head = delta[i] & ~vmask;                     // (g, a, _)
  const u64 l = lo[i], o = off[i];
  const u32 c = cnt[i];
  for (u32 j = 0; j < c; j++) cand[o + j] = head | (adj[l + j] & vmask);
}

__global__ __launch_bounds__(kCB) void widen_kernel(const u32* in, u64 n, u64* out) {
  const u64 i = (u64)blockIdx.x * kCB + threadIdx.x;
  if (i < n) out[i] = in[i];
}

__global__ __launch_bounds__(kCB) void not_member_kernel(const u64* cand, u64 n, const u64* all, u64 na, u8* flag) {
  const u64 i = (u64)blockIdx.x * kCB + threadIdx.x;
  if (i >= n) return;
  const u64 p = lower_bound_u64(all, na, cand[i]);
  flag[i] = !(p < na && all[p] == cand[i]);
}

__global__ __launch_bounds__(kCB) void decode_kernel(const u64* keys, u64 n, const u32* gdict, const u32* vdict, u32 vb, u32* g, u32* s, u32* e) {
  const u64 i = (u64)blockIdx.x * kCB + threadIdx.x;
  if (i >= n) return;
  const u64 k = keys[i], vmask = (1ull << vb) - 1ull;
  g[i] = gdict[k >> (2 * vb)];
  s[i] = vdict[(k >> vb) & vmask];
  e[i] = vdict[k & vmask];
}

u32 bits_for(u64 n) { u32 b = 1; while (b < 63 && (1ull << b) < n) b++; return b; }

template <class T> u64 sort_unique(const T* in, u64 n, DBuf& out, hipStream_t stream, u64* n_dev) {
  DBuf sorted, temp;
  T* so = sorted.alloc<T>(n);
  size_t bytes = 0;
  RDFGPU_HIP(rocprim::radix_sort_keys(nullptr, bytes, in, so, (size_t)n, 0, 8 * sizeof(T), stream));
  RDFGPU_HIP(rocprim::radix_sort_keys(temp.alloc<u8>(bytes), bytes, in, so, (size_t)n, 0, 8 * sizeof(T), stream));
  T* uo = out.alloc<T>(n);
  bytes = 0;
  RDFGPU_HIP(rocprim::unique(nullptr, bytes, so, uo, n_dev, (size_t)n, rocprim::equal_to<T>(), stream));
  DBuf temp2;
  RDFGPU_HIP(rocprim::unique(temp2.alloc<u8>(bytes), bytes, so, uo, n_dev, (size_t)n, rocprim::equal_to<T>(), stream));
  u64 m = 0;
  RDFGPU_HIP(hipMemcpyAsync(&m, n_dev, sizeof(u64), hipMemcpyDeviceToHost, stream));
  RDFGPU_HIP(hipStreamSynchronize(stream));
  return m;
}

}  // namespace

// Returns the number of paths in the closure; the three output columns come from `alloc` (rows known only at the end).
u64 closure_exec(const u32* g, const u32* s, const u32* e, u64 n, bool cross_graph, hipStream_t stream,
                 const std::function<u32*(u64)>& alloc, u32* out_cols[3], ClosureStats* stats) {
  if (stats) *stats = ClosureStats{};
  if (n == 0) return 0;
  if (n >= (1ull << 31)) fail(RDFGPU_ERR_UNSUPPORTED, "KleenePlusClosureExec over %llu inner paths", (unsigned long long)n);
  DBuf counter;
  u64* n_dev = counter.alloc<u64>(2);
  u32* null_seen = reinterpret_cast<u32*>(n_dev + 1);
  RDFGPU_HIP(hipMemsetAsync(n_dev, 0, 2 * sizeof(u64), stream));

  // dictionaries of the nodes and graphs that occur (sorted, so that rank order = id order)
  DBuf both, vdict_b, gdict_b;
  u32* cat = both.alloc<u32>(2 * n);
  hipLaunchKernelGGL(concat2_kernel, cgrid(n), dim3(kCB), 0, stream, s, e, n, cat);
  const u64 nv = sort_unique<u32>(cat, 2 * n, vdict_b, stream, n_dev);
  const u64 ng = sort_unique<u32>(g, n, gdict_b, stream, n_dev);
  const u32 vb = bits_for(nv), gb = bits_for(ng);
  if (gb + 2 * vb > 63) fail(RDFGPU_ERR_UNSUPPORTED, "KleenePlusClosureExec: %llu graphs x %llu nodes do not fit a 64-bit path key", (unsigned long long)ng, (unsigned long long)nv);
  const u32* vdict = vdict_b.get<u32>(); const u32* gdict = gdict_b.get<u32>();

  DBuf raw, all_b, adj_b;
  u64* keys = raw.alloc<u64>(n);
  hipLaunchKernelGGL(encode_kernel, cgrid(n), dim3(kCB), 0, stream, g, s, e, n, gdict, (u32)ng, vdict, (u32)nv, vb, keys, null_seen);
  u64 na = sort_unique<u64>(keys, n, all_b, stream, n_dev);       // collect_next_batch: all inner paths are part of the closure
  u32 bad = 0;
  RDFGPU_HIP(hipMemcpy(&bad, null_seen, sizeof(u32), hipMemcpyDeviceToHost));
  if (bad) fail(RDFGPU_ERR_INVALID, "KleenePlusClosureExec: could not obtain start / end value from inner paths (null)");   // physical.rs:321-326
  const u64 n_initial = na;
  // the continuation edges: the initial paths, keyed (g, s, e) — or (s, e) over all graphs when paths may cross graphs
  u64 n_adj = n_initial;
  if (cross_graph) {
    DBuf stripped;
    u64* st = stripped.alloc<u64>(n_initial);
    hipLaunchKernelGGL(strip_graph_kernel, cgrid(n_initial), dim3(kCB), 0, stream, all_b.get<u64>(), n_initial, vb, st);
    n_adj = sort_unique<u64>(st, n_initial, adj_b, stream, n_dev);
  } else {
    u64* a = adj_b.alloc<u64>(n_initial);
    RDFGPU_HIP(hipMemcpyAsync(a, all_b.get<u64>(), n_initial * sizeof(u64), hipMemcpyDeviceToDevice, stream));
  }
  const u64* adj = adj_b.get<u64>();

  DBuf delta_b;
  u64 nd = na;
  RDFGPU_HIP(hipMemcpyAsync(delta_b.alloc<u64>(nd), all_b.get<u64>(), nd * sizeof(u64), hipMemcpyDeviceToDevice, stream));
  u32 iterations = 0;
  while (nd > 0) {
    iterations++;
    DBuf lo_b, cnt_b, cnt64_b, off_b, temp;
    u64* lo = lo_b.alloc<u64>(nd); u32* cnt = cnt_b.alloc<u32>(nd);
    hipLaunchKernelGGL(range_kernel, cgrid(nd), dim3(kCB), 0, stream, delta_b.get<u64>(), nd, adj, n_adj, vb, cross_graph ? 1 : 0, lo, cnt);
    u64* cnt64 = cnt64_b.alloc<u64>(nd); u64* off = off_b.alloc<u64>(nd + 1);
    hipLaunchKernelGGL(widen_kernel, cgrid(nd), dim3(kCB), 0, stream, cnt, nd, cnt64);
    size_t bytes = 0;
    RDFGPU_HIP(rocprim::exclusive_scan(nullptr, bytes, cnt64, off, (u64)0, (size_t)nd, rocprim::plus<u64>(), stream));
    RDFGPU_HIP(rocprim::exclusive_scan(temp.alloc<u8>(bytes), bytes, cnt64, off, (u64)0, (size_t)nd, rocprim::plus<u64>(), stream));
    u64 last_off = 0; u32 last_cnt = 0;
    RDFGPU_HIP(hipMemcpyAsync(&last_off, off + nd - 1, sizeof(u64), hipMemcpyDeviceToHost, stream));
    RDFGPU_HIP(hipMemcpyAsync(&last_cnt, cnt + nd - 1, sizeof(u32), hipMemcpyDeviceToHost, stream));
    RDFGPU_HIP(hipStreamSynchronize(stream));
    const u64 total = last_off + last_cnt;
    if (total == 0) break;
    if (total >= (1ull << 32) || na + total >= (1ull << 32)) fail(RDFGPU_ERR_UNSUPPORTED, "KleenePlusClosureExec: %llu candidate paths in one iteration", (unsigned long long)total);
    DBuf cand_b, uniq_b, flag_b, fresh_b;
    u64* cand = cand_b.alloc<u64>(total);
    hipLaunchKernelGGL(expand_kernel, cgrid(nd), dim3(kCB), 0, stream, delta_b.get<u64>(), nd, adj, vb, lo, cnt, off, cand);
    const u64 m = sort_unique<u64>(cand, total, uniq_b, stream, n_dev);
    u8* flag = flag_b.alloc<u8>(m);
    hipLaunchKernelGGL(not_member_kernel, cgrid(m), dim3(kCB), 0, stream, uniq_b.get<u64>(), m, all_b.get<u64>(), na, flag);
    u64* fresh = fresh_b.alloc<u64>(m);
    DBuf temp2;
    bytes = 0;
    RDFGPU_HIP(rocprim::select(nullptr, bytes, uniq_b.get<u64>(), flag, fresh, n_dev, (size_t)m, stream));
    RDFGPU_HIP(rocprim::select(temp2.alloc<u8>(bytes), bytes, uniq_b.get<u64>(), flag, fresh, n_dev, (size_t)m, stream));
    u64 n_fresh = 0;
    RDFGPU_HIP(hipMemcpyAsync(&n_fresh, n_dev, sizeof(u64), hipMemcpyDeviceToHost, stream));
    RDFGPU_HIP(hipStreamSynchronize(stream));
    if (n_fresh == 0) break;
    DBuf merged_b, temp3;
    u64* merged = merged_b.alloc<u64>(na + n_fresh);
    bytes = 0;
    RDFGPU_HIP(rocprim::merge(nullptr, bytes, all_b.get<u64>(), fresh, merged, (size_t)na, (size_t)n_fresh, rocprim::less<u64>(), stream));
    RDFGPU_HIP(rocprim::merge(temp3.alloc<u8>(bytes), bytes, all_b.get<u64>(), fresh, merged, (size_t)na, (size_t)n_fresh, rocprim::less<u64>(), stream));
    RDFGPU_HIP(hipStreamSynchronize(stream));   // the temporaries of this iteration die at the end of the scope
    all_b.swap(merged_b);
    na += n_fresh;
    delta_b.swap(fresh_b);
    nd = n_fresh;
  }
  for (int c = 0; c < 3; c++) out_cols[c] = alloc(na);
  hipLaunchKernelGGL(decode_kernel, cgrid(na), dim3(kCB), 0, stream, all_b.get<u64>(), na, gdict, vdict, vb, out_cols[0], out_cols[1], out_cols[2]);
  RDFGPU_HIP(hipStreamSynchronize(stream));     // the dictionaries are freed on return
  if (stats) { stats->iterations = iterations; stats->nodes = nv; stats->graphs = ng; stats->initial_paths = n_initial; }
  return na;
}

// (kernels.hpp, preload_code_objects: the runtime loads a translation unit's code object at the first use of one of its kernels)
void preload_tu_closure() { hipFuncAttributes at; RDFGPU_HIP(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(concat2_kernel))); }
}  // namespace rdfgpu
