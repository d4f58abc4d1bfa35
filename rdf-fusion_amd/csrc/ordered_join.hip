// ordered_join.hip — HashJoinExec of a small table against a store slice, emitted IN THE SLICE'S ORDER.
//
// The index join (plan.cpp: build on the slice's cached CSR table, probe with the table's rows) emits its matches in
// probe-row order.  When the slice is sorted by a column that is NOT the join key — a GPOS slice (?s <p> ?o) joined on
// ?s is sorted by ?o — and the next operator partitions this join's output by that very column (the key-partitioned band
// join above it: BSBM Q5's candidates by product feature), walking the join the other way round saves that partition
// pass: the table's rows are hung into a multimap by key (head[key - kmin], next[row]), the slice is streamed once in its
// own order and every slice row emits its chain.  Two streaming passes (counts per tile -> device scan -> ordered write),
// no atomics on the output, no sort; the output is sorted like the slice.  Follow-up look-ups keyed by columns of the
// table (the fused chain's stages: `<X> numeric1 ?v` by X) are done ONCE per table row, not per match.
//
// Same row multiset as lds_join_kernel<.., MODE 3> with the same arguments (inner join, NullEqualsNothing,
// join/rewrite.rs:89,126-168); tests/test_gpu_parity.py::test_ordered_slice_join checks the two forms.
#include <hip/hip_runtime.h>

#include "join_device.hpp"

namespace rdfgpu {

constexpr int kOjBlock = 256;
constexpr int kOjRounds = 4;                       // slice rows per lane and tile: tile = 1024 rows, round-major order
constexpr u32 kOjTile = kOjBlock * kOjRounds;

// table row -> its stage rows, and into the multimap when every stage has a row (inner joins: no stage row, no match)
__global__ __launch_bounds__(256) void oj_probe_kernel(const OrderedJoinArgs a) {
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 n = live_rows(a.n_probe_dev, a.n_probe_cap);
  if (r >= n) return;
  const u32 key = a.probe_key[r];
  const u32 d = key - a.kmin;
  bool ok = key != 0 && d < a.kn;                  // null keys never join
#pragma unroll
  for (u32 t = 0; t < (u32)kMaxChain; t++) {
    if (t >= a.n_stages) continue;
    const u32 sk = a.stage[t].key_col[r];
    const u32 sd = sk - a.stage[t].kmin;
    const bool in = sk != 0 && sd < a.stage[t].kn;
    const u32 row = in ? a.stage[t].direct[sd] : kNil;
    a.stage[t].row[r] = row;
    ok = ok && row != kNil;
  }
  if (!ok) return;
  u32 w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  for (u32 c = 0; c < a.n_out_cols; c++) {
    const u32 slot = a.out_slot[c];
    if (slot == 0xFFu) continue;
    const ColRef ref = a.out_ref[c];
    const u64 src = ref.src == 0 ? r : (u64)a.stage[ref.src - 2].row[r];
    const u32 val = ref.ptr[src];
#pragma unroll
    for (u32 k = 0; k < 8; k++) w[k] = k == slot ? val : w[k];
  }
  a.trec[r * a.n_rec] = make_uint4(w[0], w[1], w[2], w[3]);
  if (a.n_rec > 1) a.trec[r * a.n_rec + 1] = make_uint4(w[4], w[5], w[6], w[7]);
  const u32 old = atomicExch(&a.head[d].x, (u32)r);
  a.next[r] = old;
  if (old != kNil) atomicAdd(&a.head[d].y, 1u);    // rows of a key beyond its first (0xFFFFFFFF + 1 = 0): the count pass reads chain head
}                                                  // and length with ONE 8-byte gather (its gathers are what it is bound by); a key with
                                                   // one row — the common case — costs one atomic here
__global__ __launch_bounds__(kOjBlock) void oj_count_kernel(const OrderedJoinArgs a) {
  __shared__ u32 wave_tot[kOjBlock / 64];
  const u64 base = (u64)blockIdx.x * kOjTile;
  u32 keys[kOjRounds];
#pragma unroll
  for (int it = 0; it < kOjRounds; it++) {
    const u64 row = base + (u64)it * kOjBlock + threadIdx.x;
    keys[it] = row < a.n_build ? a.build_key[row] : 0u;
  }
  u32 tot = 0;
#pragma unroll
  for (int it = 0; it < kOjRounds; it++) {
    const u64 row = base + (u64)it * kOjBlock + threadIdx.x;
    const u32 d = keys[it] - a.kmin;
    const uint2 hc = (keys[it] != 0 && d < a.kn) ? a.head[d] : make_uint2(kNil, 0xFFFFFFFFu);
    const u32 hd = hc.x, c = hc.x != kNil ? hc.y + 2u : 0u;   // 1 + the rows beyond the first (stored - 1)
    if (row < a.n_build) { a.row_head[row] = hd; a.row_cnt[row] = (unsigned char)(c < 255u ? c : 255u); }
    tot += c;
  }
  tot = wave_incl_scan(tot);                               // (DPP: lane 63 holds the wave's total)
  if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    a.tile_count[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    if (blockIdx.x == 0) a.tile_count[gridDim.x] = 0;   // the scan's extra element (total = its exclusive prefix): no memset launch for 4 bytes
  }
}

// Pass 2.  A tile's matches are numbered 0 .. T-1 in slice order; match j is written by lane j mod 256 — every lane has
// work whatever the chain lengths are, the stores of a wave are consecutive, and a match's gathers (table row, stage
// rows, one value per output column) are all issued before the first store.  Which slice row a match belongs to: a
// binary search over the tile's per-row start offsets in LDS (10 steps); which table row: that many hops down the
// row's chain (chains are short: rows of the table that share one key).
// NCOLS = output columns: a template parameter, so that only the live columns' pointers sit in SGPRs (with room for
// kOjMaxOutCols the kernel spilled scalars into vector lanes inside the loop: 244 v_readlane / v_writelane).  The scans are
// DPP scans (wave_incl_scan), one per round giving the round's total as well — the shuffle forms were 48 ds_bpermute.
template <int NCOLS>
__global__ __launch_bounds__(kOjBlock) void oj_write_kernel(const OrderedJoinArgs a) {
  __shared__ u32 starts[kOjTile];
  __shared__ u32 heads[kOjTile];
  __shared__ u32 rcnt[kOjRounds][kOjBlock / 64];
  const u64 base = (u64)blockIdx.x * kOjTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const u64 total = a.tile_off[gridDim.x];
    *a.n_out_dev = total;
    if (total > a.out_cap) *a.overflow = 1u;
  }
  const u32 tile_total = a.tile_count[blockIdx.x];
  if (tile_total == 0) return;                     // uniform per workgroup
  u32 cnt[kOjRounds], hd[kOjRounds], incl[kOjRounds];
#pragma unroll
  for (int it = 0; it < kOjRounds; it++) {   // what the count pass found: streamed, not gathered again
    const u64 row = base + (u64)it * kOjBlock + threadIdx.x;
    hd[it] = row < a.n_build ? a.row_head[row] : kNil;
    cnt[it] = row < a.n_build ? (u32)a.row_cnt[row] : 0u;
  }
#pragma unroll
  for (int it = 0; it < kOjRounds; it++) {
    u32 c = cnt[it];
    if (c == 255u) { c = 0; for (u32 r = hd[it]; r != kNil; r = a.next[r]) c++; }   // (a chain of 255 or more table rows: its exact length)
    cnt[it] = c;
    incl[it] = wave_incl_scan(c);
    if (lane == 63) rcnt[it][wave] = incl[it];
  }
  __syncthreads();
  u32 off = 0;
#pragma unroll
  for (int it = 0; it < kOjRounds; it++) {
    u32 woff = off;
    for (int w = 0; w < wave; w++) woff += rcnt[it][w];
    off += rcnt[it][0] + rcnt[it][1] + rcnt[it][2] + rcnt[it][3];
    starts[it * kOjBlock + threadIdx.x] = woff + (incl[it] - cnt[it]);
    heads[it * kOjBlock + threadIdx.x] = hd[it];
  }
  __syncthreads();
  const u64 tile_base = a.tile_off[blockIdx.x];
  // the column schedule out of the argument block ONCE (constant indices -> SGPRs), turned round: per WORD of the packed table
  // record the column it feeds (no per-match selection among the record's words), per slice-sourced column its source
  u32* word_out[8]; u32* ref_out[NCOLS]; const u32* ref_in[NCOLS];
#pragma unroll
  for (u32 k = 0; k < 8; k++) word_out[k] = nullptr;
#pragma unroll
  for (u32 c = 0; c < (u32)NCOLS; c++) {
    const u32 sl = a.out_slot[c];
    ref_out[c] = sl == 0xFFu ? a.out[c] : nullptr;
    ref_in[c] = a.out_ref[c].ptr;
#pragma unroll
    for (u32 k = 0; k < 8; k++) word_out[k] = sl == k ? a.out[c] : word_out[k];
  }
  const u32 n_rec = a.n_rec;
  const u64 out_cap = a.out_cap;
  for (u32 j = threadIdx.x; j < tile_total; j += kOjBlock) {
    u32 lo = 0, hi = kOjTile;                      // the last q with starts[q] <= j (rows without matches share their successor's start)
#pragma unroll
    for (int step = 0; step < 10; step++) { const u32 mid = (lo + hi) >> 1; if (starts[mid] <= j) lo = mid; else hi = mid; }
    const u32 q = lo;
    u32 r = heads[q];
    for (u32 k = j - starts[q]; k; k--) r = a.next[r];
    const u64 pos = tile_base + j;
    if (pos >= out_cap) continue;                  // the count stays exact: the plan re-runs with room for all
    const u64 brow = base + q;
    const uint4 r0 = a.trec[(u64)r * n_rec];
    uint4 r1 = make_uint4(0u, 0u, 0u, 0u);
    if (n_rec > 1) r1 = a.trec[(u64)r * n_rec + 1];
    u32 v[NCOLS];
#pragma unroll
    for (u32 c = 0; c < (u32)NCOLS; c++) v[c] = ref_out[c] ? ref_in[c][brow] : 0u;   // (wave-uniform conditions)
    const u32 w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (u32 k = 0; k < 8; k++) if (word_out[k]) word_out[k][pos] = w[k];
#pragma unroll
    for (u32 c = 0; c < (u32)NCOLS; c++) if (ref_out[c]) ref_out[c][pos] = v[c];
  }
}

// Pass 2 when the matches feed a band join (OjBandFuse, kernels.hpp): a match is the 32-byte record of its table row, copied to
// its position in match order; the key boundaries of the matches (the band join's partition of its probe side: poff[k] = first
// match whose key is >= k) fall out of the slice's sorted column — a slice row whose key differs from its predecessor's starts
// the keys in between at its own first match.  Same numbering of matches as oj_write_kernel.
__global__ __launch_bounds__(kOjBlock) void oj_write_band_kernel(const OrderedJoinArgs a, const OjBandFuse f) {
  __shared__ u32 starts[kOjTile];
  __shared__ u32 heads[kOjTile];
  __shared__ u32 rcnt[kOjRounds][kOjBlock / 64];
  const u64 base = (u64)blockIdx.x * kOjTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u64 total = a.tile_off[gridDim.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *a.n_out_dev = total;
    if (total > a.out_cap) *a.overflow = 1u;
  }
  const u32 tile_total = a.tile_count[blockIdx.x];
  const u64 tile_base = a.tile_off[blockIdx.x];
  auto key_of = [&](u64 row) { u32 kk = f.kn; if (row < a.n_build) { const u32 v = f.key_col[row]; const u32 d = v - f.kmin; if (v != 0 && d < f.kn) kk = d; } return kk; };
  auto bounds = [&](u64 row, u64 first_match) {           // the keys that start at slice row `row` (row == n_build closes the table)
    if (row > a.n_build) return;
    const u32 from = row > 0 ? key_of(row - 1) + 1u : 0u, to = key_of(row);
    for (u32 q = from; q <= to && q <= f.kn; q++) f.poff[q] = (u32)(first_match < a.out_cap ? first_match : a.out_cap);
  };
  if (tile_total == 0) {                                   // no match in this tile: every row "starts" at the tile's offset
#pragma unroll
    for (int it = 0; it < kOjRounds; it++) bounds(base + (u64)it * kOjBlock + threadIdx.x, tile_base);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && base + kOjTile == a.n_build) bounds(a.n_build, total);
    return;                                                // uniform per workgroup
  }
  u32 cnt[kOjRounds], hd[kOjRounds], incl[kOjRounds];
#pragma unroll
  for (int it = 0; it < kOjRounds; it++) {
    const u64 row = base + (u64)it * kOjBlock + threadIdx.x;
    hd[it] = row < a.n_build ? a.row_head[row] : kNil;
    cnt[it] = row < a.n_build ? (u32)a.row_cnt[row] : 0u;
  }
#pragma unroll
  for (int it = 0; it < kOjRounds; it++) {
    u32 c = cnt[it];
    if (c == 255u) { c = 0; for (u32 r = hd[it]; r != kNil; r = a.next[r]) c++; }
    cnt[it] = c;
    incl[it] = wave_incl_scan(c);
    if (lane == 63) rcnt[it][wave] = incl[it];
  }
  __syncthreads();
  u32 off = 0;
#pragma unroll
  for (int it = 0; it < kOjRounds; it++) {
    u32 woff = off;
    for (int w = 0; w < wave; w++) woff += rcnt[it][w];
    off += rcnt[it][0] + rcnt[it][1] + rcnt[it][2] + rcnt[it][3];
    const u32 st = woff + (incl[it] - cnt[it]);
    starts[it * kOjBlock + threadIdx.x] = st;
    heads[it * kOjBlock + threadIdx.x] = hd[it];
    bounds(base + (u64)it * kOjBlock + threadIdx.x, tile_base + st);
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && base + kOjTile == a.n_build) bounds(a.n_build, total);
  __syncthreads();
  const u64 out_cap = a.out_cap;
  for (u32 j = threadIdx.x; j < tile_total; j += kOjBlock) {
    u32 lo = 0, hi = kOjTile;
#pragma unroll
    for (int step = 0; step < 10; step++) { const u32 mid = (lo + hi) >> 1; if (starts[mid] <= j) lo = mid; else hi = mid; }
    u32 r = heads[lo];
    for (u32 k = j - starts[lo]; k; k--) r = a.next[r];
    const u64 pos = tile_base + j;
    if (pos >= out_cap) continue;                          // the count stays exact: the plan re-runs with room for all
    if (f.compact) {                                       // one 16-byte record per match
      uint4 rc = f.brec[r];
      if (f.self_index) rc.z = (u32)(base + lo);           // the slice row of this match = the index of the row's own entry in the band join's group
      f.rec_s[pos] = rc;
      continue;
    }
    const uint4 r0 = f.brec[2ull * r], r1 = f.brec[2ull * r + 1];
    f.rec_s[pos] = r0;
    f.aux_s[pos] = r1;
  }
}
void launch_ordered_join_write_band(const OrderedJoinArgs& a, const OjBandFuse& f, hipStream_t s) {
  hipLaunchKernelGGL(oj_write_band_kernel, dim3((unsigned)ordered_join_tiles(a.n_build)), dim3(kOjBlock), 0, s, a, f);
}

void launch_ordered_join_probe(const OrderedJoinArgs& a, hipStream_t s) {
  if (!a.n_probe_cap) return;
  hipLaunchKernelGGL(oj_probe_kernel, dim3((unsigned)((a.n_probe_cap + 255) / 256)), dim3(256), 0, s, a);
}
u64 ordered_join_tiles(u64 n_build) { return (n_build + kOjTile - 1) / kOjTile; }
void launch_ordered_join_count(const OrderedJoinArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(oj_count_kernel, dim3((unsigned)ordered_join_tiles(a.n_build)), dim3(kOjBlock), 0, s, a);
}
void launch_ordered_join_write(const OrderedJoinArgs& a, hipStream_t s) {
  const dim3 g((unsigned)ordered_join_tiles(a.n_build));
  static_assert(kOjMaxOutCols == 8, "one instantiation per column count");
  switch (a.n_out_cols) {
    case 1: hipLaunchKernelGGL(oj_write_kernel<1>, g, dim3(kOjBlock), 0, s, a); return;
    case 2: hipLaunchKernelGGL(oj_write_kernel<2>, g, dim3(kOjBlock), 0, s, a); return;
    case 3: hipLaunchKernelGGL(oj_write_kernel<3>, g, dim3(kOjBlock), 0, s, a); return;
    case 4: hipLaunchKernelGGL(oj_write_kernel<4>, g, dim3(kOjBlock), 0, s, a); return;
    case 5: hipLaunchKernelGGL(oj_write_kernel<5>, g, dim3(kOjBlock), 0, s, a); return;
    case 6: hipLaunchKernelGGL(oj_write_kernel<6>, g, dim3(kOjBlock), 0, s, a); return;
    case 7: hipLaunchKernelGGL(oj_write_kernel<7>, g, dim3(kOjBlock), 0, s, a); return;
    case 8: hipLaunchKernelGGL(oj_write_kernel<8>, g, dim3(kOjBlock), 0, s, a); return;
  }
  fail(RDFGPU_ERR_INVALID, "ordered join with %u output columns", a.n_out_cols);
}

// (kernels.hpp, preload_code_objects: the runtime loads a translation unit's code object at the first use of one of its kernels)
void preload_tu_ordered_join() { hipFuncAttributes at; RDFGPU_HIP(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(oj_probe_kernel))); }
}  // namespace rdfgpu
