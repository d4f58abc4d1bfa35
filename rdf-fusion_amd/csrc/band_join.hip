// band_join.hip — key-partitioned band join: the fused look-up chain (kernels.hpp ChainStage) over a CSR join whose
// groups are small, executed GROUP BY GROUP instead of probe row by probe row.
//
// What it replaces: the same operators as lds_join_kernel<.., CSR, CHAIN> —
//   HashJoinExec(CollectLeft) on one dense key (lib/logical/src/join/rewrite.rs:126-168, NullEqualsNothing :89)
//   + the inner single-key HashJoinExecs above it whose other inputs are store slices (BSBM Explore Q5:
//   bench/tests/plans/snapshots/..Q5 (Execution Plan).snap:10-30), their JoinFilters being integer windows
//   `EBV(cmp(ENC_TV(x), ADD|SUB(ENC_TV(y), lit))) AND EBV(cmp(ENC_TV(x), ADD|SUB(ENC_TV(y'), lit')))`
//   (greater_than.rs:41-64, add.rs:40-86, effective_boolean_value.rs:99-119) with x a column of the stage's slice
//   and y, y' probe-side columns, plus `col <ID_EQ|ID_NEQ> col` on the base join.
//
// Why: probing row by row makes every candidate pair pay dependent gathers (group offsets -> value-ordered index ->
// stage value tables); 107 M candidate pairs of a 262 144-instance BSBM Q5 batch cost 3.7 ms that way, 22x the
// compulsory bytes in L2 misses.  Both sides of the base join are partitioned by the key instead — the build side IS
// partitioned already (a CSR table = rows grouped by key), the probe side is radix-sorted by key per execution — and
// every key's group meets the key's probe rows in 64 x 64 blocks, one wave per block: the group's entries are decoded
// ONCE per execution (stage look-ups, decoded window operands), the probe rows are decoded ONCE (window bounds as
// biased 32-bit intervals), and the pair tests are register compares (two unsigned range checks and one id compare per
// pair), 64 probe rows per wave-instruction, the entry operands arriving through the scalar cache.
// Survivors are 1 bit per pair in HBM; a device scan over the per-block counts places every block's output; a second
// pass expands the bits into rows.  No atomics, no candidate queue, deterministic output order.
//
//   band_entries_kernel per build row, in CSR order: the pair test's operands (16 B) and the entry's output values
//   band_decode_kernel  per probe row, in row order: sort key (key - kmin; kn = "joins nothing"), window bounds (checked
//                       i64 -> biased u32 interval), id operand                                    -> rocPRIM radix sort
//   band_bounds_kernel  poff[k] = first sorted position with key >= k (one pass, gaps filled)
//   band_blocks_scan    blocks (64 entries x 64 rows) per key, computed inside the scan's input iterator: boff = first block of a key -> band_desc_kernel
//   band_mask_kernel    per block: the pair tests, 1 bit per pair, count per block
//   band_slow_kernel    rows whose operands are not all xsd:integer: full semantics (exits at once if there are none)
//   band_emit_kernel    per block: bits -> output rows at the block's scanned offset, coalesced
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "join_device.hpp"

namespace rdfgpu {

static inline dim3 grid256(u64 n) { const u64 g = (n + 255) / 256; return dim3((unsigned)(g ? g : 1)); }
constexpr u32 kBandInvalidLo = 0xFFFFFFFFu;   // a row / window that nothing can pass: (x - 0xFFFFFFFF) <= 0 never holds for x < 2^32 - 1

// ---- partition of the probe side by key ----------------------------------------------------------------------------
// poff[k] = number of sorted rows with key < k, for k = 0 .. kn (kn + 1 entries): every position whose key differs
// from its predecessor's fills the keys in between.
__global__ __launch_bounds__(256) void band_bounds_kernel(const u32* skey_sorted, u64 n, u32 kn, u32* poff) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  const u32 prev = i > 0 ? skey_sorted[i - 1] + 1u : 0u;          // first key not yet answered
  const u32 cur = i < n ? skey_sorted[i] : kn;                     // keys <= cur start at or before i
  for (u32 k = prev; k <= cur && k <= kn; k++) poff[k] = (u32)i;
}

// ---- per probe row: the window of every stage as a biased 32-bit interval -----------------------------------------
// The integer window  cmp0(x, y0 +/- lit0) AND cmp1(x, y1 +/- lit1)  over xsd:integer operands is  lo <= x <= hi  with
// checked i64 arithmetic (add.rs:52-80: overflow => error => the row passes nothing).  false: some operand is not an
// xsd:integer (or a comparison is = / !=): the pair goes through the full typed-value semantics instead.
__device__ __forceinline__ bool band_window_of(const TypedTable& tt, const BandWin& w, u32 iy0, u32 iy1, long long& lo, long long& hi) {
  lo = INT64_MIN + 1; hi = INT64_MAX;
  if (tt.n_ids == 0 || iy0 == 0 || iy1 == 0 || iy0 >= tt.n_ids || iy1 >= tt.n_ids) return false;
  if (w.l0.tag != RDFGPU_TV_INTEGER || w.l1.tag != RDFGPU_TV_INTEGER) return false;
  const int4* tv = reinterpret_cast<const int4*>(tt.tv);
  const int4 r0 = tv[iy0];
  const int4 r1 = iy1 == iy0 ? r0 : tv[iy1];            // both halves of a window usually read ONE value (x < y + c AND x > y - c)
  if (((u32)r0.w & 0xff) != RDFGPU_TV_INTEGER || ((u32)r1.w & 0xff) != RDFGPU_TV_INTEGER) return false;
  auto i64 = [](const int4& r) { return (long long)(((u64)(u32)r.y << 32) | (u32)r.x); };
  bool ok = true;
  auto bound = [&](const TvLiteral& l, long long y) {
    long long z;
    const bool ovf = l.arith_sub ? __builtin_sub_overflow(y, (long long)l.lo, &z) : __builtin_add_overflow(y, (long long)l.lo, &z);
    if (ovf) { ok = false; return; }
    if (l.cmp_op == RDFGPU_EX_LT) { if (z == INT64_MIN) hi = INT64_MIN; else hi = z - 1 < hi ? z - 1 : hi; }
    else if (l.cmp_op == RDFGPU_EX_LEQ) hi = z < hi ? z : hi;
    else if (l.cmp_op == RDFGPU_EX_GT) { if (z == INT64_MAX) { lo = INT64_MAX; hi = INT64_MIN; } else lo = z + 1 > lo ? z + 1 : lo; }
    else if (l.cmp_op == RDFGPU_EX_GEQ) lo = z > lo ? z : lo;
    else ok = false;
  };
  bound(w.l0, i64(r0));
  if (ok) bound(w.l1, i64(r1));
  return ok;
}
// [lo, hi] over i64 -> {lo_b, width} over the stage's biased values (stored = value - vbase + 1 in [1, 2^32 - 16]; 0 = the
// entry has no stage row).  A pair passes iff (x_b - lo_b) <= width as unsigned numbers.
__device__ __forceinline__ uint2 band_interval(long long lo, long long hi, long long vb) {
  const unsigned long long dl = (unsigned long long)lo - (unsigned long long)vb;   // exact when lo >= vb
  const unsigned long long dh = (unsigned long long)hi - (unsigned long long)vb;   // exact when hi >= vb
  const bool empty = hi < vb || lo > hi || (lo > vb && dl >= 0xFFFFFFF0ull);
  if (empty) return make_uint2(kBandInvalidLo, 0u);
  const u32 lo_b = lo <= vb ? 1u : (u32)dl + 1u;
  const u32 hi_b = dh >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (u32)dh + 1u;
  return make_uint2(lo_b, hi_b - lo_b);
}
// [lo, lo + w] over 32-bit biased values -> the same interval over 16 bits (values are <= 65533 when this form is chosen)
__device__ __forceinline__ void band_pack_interval(u32 lo, u32 w, u32& lo16, u32& w16) {
  if (lo > 65534u) { lo16 = 0xFFFFu; w16 = 0u; return; }           // empty (or the "nothing passes" marker): (x - 0xFFFF) mod 2^16 >= 1 > 0
  const u64 hi = (u64)lo + w;
  lo16 = lo; w16 = (u32)(hi < 65534ull ? hi : 65534ull) - lo;      // a dead entry (x = 0) gives 65536 - lo > w16: never passes
}
// What the pair test needs of ONE probe row: its windows as biased intervals (packed to 2 x 16 bits when the plan says so), given
// the ids of the window operands and the id operand x of the base join's filter.  flags: 1 = some operand is not an xsd:integer
// (the row takes the full typed-value semantics, its record passes nothing), 2 = x is null.
__device__ __forceinline__ void band_row_record(const BandArgs& b, const u32 iy0[2], const u32 iy1[2], u32 x, uint4& rec, u32& flags) {
  rec = make_uint4(1u, 0u, 1u, 0u);              // no window: entries carry x_b = 1 (0 when the entry is dead)
  flags = 0;
#pragma unroll
  for (u32 w = 0; w < 2; w++) {
    if (w >= b.n_win) continue;
    const BandWin& bw = b.win[w];
    long long lo, hi;
    uint2 iv = make_uint2(kBandInvalidLo, 0u);
    if (band_window_of(b.tt, bw, iy0[w], iy1[w], lo, hi)) iv = band_interval(lo, hi, bw.vbase);
    else flags |= 1u;                                  // slow row: full semantics per pair
    if (w == 0) { rec.x = iv.x; rec.y = iv.y; } else { rec.z = iv.x; rec.w = iv.y; }
  }
  if (b.has_neq && x == 0) { rec.x = kBandInvalidLo; rec.y = 0u; flags = 2u; }   // null => the comparison is not `true`
  if (flags & 1u) { rec.x = kBandInvalidLo; rec.y = 0u; }   // the register test must not pass a slow row
  if (b.pack16) {   // the pair test's packed form: both windows' {lo, width} as 2 x 16 bits each — once per row here, not once per row and block there
    u32 l0, w0, l1, w1;
    band_pack_interval(rec.x, rec.y, l0, w0);
    band_pack_interval(rec.z, rec.w, l1, w1);
    rec = make_uint4(l0 | (l1 << 16), w0 | (w1 << 16), 0u, 0u);
  }
}
// One atomic per RUN of equal keys among neighbouring lanes of a wave instead of one per lane: the re-sharded probe side of
// a sharded step arrives as N sorted runs, so neighbouring rows share their key (any input is handled: a lane whose
// neighbours differ is a run of one).  Every lane of the wave calls it; returns the lane's own position (counter value
// before the run + the lane's rank in the run) for `valid` lanes.
__device__ __forceinline__ u32 band_run_atomic_add(u32* counters, u32 k, bool valid) {
  const int lane = threadIdx.x & 63;
  const u32 prev = __shfl_up(k, 1, 64);
  const unsigned long long vm = __ballot(valid);
  const bool pvalid = lane > 0 && ((vm >> (lane - 1)) & 1ull);
  const bool head = valid && (!pvalid || prev != k);
  const unsigned long long hm = __ballot(head);
  const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
  const int leader = 63 - __builtin_clzll((hm & upto) | 1ull);                 // (| 1: defined for invalid lanes; their result is unused)
  const unsigned long long above = (hm | ~vm) & (leader == 63 ? 0ull : ~((2ull << leader) - 1ull));
  const int end = above ? __builtin_ctzll(above) : 64;
  u32 r = 0;
  if (valid && lane == leader) r = atomicAdd(counters + k, (u32)(end - leader));
  r = __shfl(r, leader, 64);
  return r + (u32)(lane - leader);
}
// One pass over the probe side in ROW order (coalesced column reads): the sort key of every row (key - kmin; kn =
// joins nothing) and, for the rows that can join, the decoded windows + the id operand of the base join's filter.
__global__ __launch_bounds__(256) void band_decode_kernel(const BandArgs b) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  for (u64 z = j; z <= b.max_blocks; z += (u64)gridDim.x * blockDim.x) b.bcount[z] = 0u;   // the blocks' counts start at 0 (no memset launches)
  const u64 n = live_rows(b.n_probe_dev, b.n_probe_cap);
  auto key_of = [&](u64 r) { u32 kk = b.kn; if (r < n) { const u32 v = b.probe_key[r]; const u32 d = v - b.kmin; if (v != 0 && d < b.kn) kk = d; } return kk; };   // null keys never join
  const u32 k = j < b.n_probe_cap ? key_of(j) : b.kn;
  if (b.key_hist) band_run_atomic_add(b.key_hist, k, k != b.kn);   // (whole waves: before any lane leaves)
  if (!b.presorted && (blockIdx.x & 63u) == 0) {      // a sample of the waves: rows that can join, and runs of equal neighbouring keys among them
    const u32 prev = __shfl_up(k, 1, 64);
    const unsigned long long vm = __ballot(k != b.kn), hm = __ballot(k != b.kn && ((threadIdx.x & 63) == 0 || prev != k));
    if ((threadIdx.x & 63) == 0 && vm) atomicAdd(b.run_stats, ((unsigned long long)__popcll(vm) << 32) | (unsigned long long)__popcll(hm));
  }
  if (j > b.n_probe_cap) return;
  if (b.presorted) {   // rows arrive sorted by key: the key boundaries (band_bounds_kernel's job) fall out here — poff[q] = first row with key >= q
    const u32 first = j > 0 ? key_of(j - 1) + 1u : 0u;
    for (u32 q = first; q <= k && q <= b.kn; q++) b.poff[q] = (u32)j;
  }
  if (j >= b.n_probe_cap) return;
  b.skey_in[j] = k; b.sval_in[j] = (u32)j;
  if (k == b.kn) return;
  u32 iy0[2] = {0u, 0u}, iy1[2] = {0u, 0u};
#pragma unroll
  for (u32 w = 0; w < 2; w++) if (w < b.n_win) { iy0[w] = b.win[w].y0[j]; iy1[w] = b.win[w].y1[j]; }
  const u32 x = b.has_neq ? b.neq_probe[j] : 0u;
  u32 rv[kBandMaxRowCols] = {0u, 0u};
#pragma unroll
  for (u32 u = 0; u < kBandMaxRowCols; u++) if (u < b.n_row_cols) rv[u] = b.row_col[u][j];
  uint4 rec; u32 flags;
  band_row_record(b, iy0, iy1, x, rec, flags);
  if (flags & 1u) atomicAdd(b.slow_rows, 1u);
  if (b.compact) {                                     // the whole row record in 16 bytes (packed windows, id operand, one output value; BandArgs::compact)
    const uint4 rc = make_uint4(rec.x, rec.y, x, rv[0]);
    if (b.presorted) b.rec_s[j] = rc; else b.rec[j] = rc;
    return;
  }
  if (b.presorted) { b.rec_s[j] = rec; b.aux_s[j] = make_uint4(x, flags, rv[0], rv[1]); return; }   // row order IS the sorted order
  b.rec[2 * j] = rec;
  b.rec[2 * j + 1] = make_uint4(x, flags, rv[0], rv[1]);
}
// The probe side comes out of an ordered slice join (ordered_join.hip) whose write pass emits these records itself: everything
// a record holds is a function of the TABLE row of that join (window operands, id operand, output values travel in its packed
// record `trec`), so it is decoded here once per table row — 262 144 instances, not 4.85 M matches — and a match copies it.
__global__ __launch_bounds__(256) void oj_band_records_kernel(const OrderedJoinArgs a, const BandArgs b, const OjBandFuse f) {
  const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  for (u64 z = r; z <= f.max_blocks; z += (u64)gridDim.x * blockDim.x) f.bcount[z] = 0u;   // (what the decode pass does on the way)
  const u64 n = live_rows(a.n_probe_dev, a.n_probe_cap);
  if (r >= n) return;
  const u32 key = a.probe_key[r];
  bool linked = key != 0 && key - a.kmin < a.kn;      // the rows oj_probe_kernel hung into the multimap (only they have a packed record)
#pragma unroll
  for (u32 t = 0; t < (u32)kMaxChain; t++) if (t < a.n_stages) linked = linked && a.stage[t].row[r] != kNil;
  if (!linked) return;
  const u32* words = reinterpret_cast<const u32*>(a.trec) + r * 4ull * a.n_rec;
  auto word = [&](u8 slot) { return slot == 0xFFu ? 0u : words[slot]; };
  u32 iy0[2], iy1[2];
#pragma unroll
  for (u32 w = 0; w < 2; w++) { iy0[w] = word(f.y0_slot[w]); iy1[w] = word(f.y1_slot[w]); }
  const u32 x = word(f.neq_slot);
  uint4 rec; u32 flags;
  band_row_record(b, iy0, iy1, x, rec, flags);
  if (flags & 1u) atomicAdd(b.slow_rows, 1u);
  if (f.compact) { f.brec[r] = make_uint4(rec.x, rec.y, x, word(f.row_slot[0])); return; }   // (packed form: rec.z / .w are unused; a flagged row's record passes nothing)
  f.brec[2 * r] = rec;
  f.brec[2 * r + 1] = make_uint4(x, flags, word(f.row_slot[0]), word(f.row_slot[1]));
}
void launch_oj_band_records(const OrderedJoinArgs& a, const BandArgs& b, const OjBandFuse& f, hipStream_t s) {
  if (a.n_probe_cap) hipLaunchKernelGGL(oj_band_records_kernel, grid256(a.n_probe_cap), dim3(256), 0, s, a, b, f);
}

// The records in sorted order: the one random access per probe row (32 contiguous bytes); everything downstream streams.
__global__ __launch_bounds__(256) void band_rows_kernel(const BandArgs b) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n_probe_cap || i >= b.poff[b.kn]) return;   // rows that join nothing sort to the end
  const u32 j = b.perm[i];
  if (b.compact) { b.rec_s[i] = b.rec[j]; return; }
  b.rec_s[i] = b.rec[2 * (u64)j];
  b.aux_s[i] = b.rec[2 * (u64)j + 1];
}

// Counting-sort form of the partition pass, for probe sides too small for rocPRIM's radix sort to leave its launch-latency
// floor (135 us for 0.6 M rows — one rank's share of a sharded step): the decode pass counts the rows of every key, a scan
// gives poff, and every row takes the next free position of its key (order inside a key is irrelevant to the join).
__global__ __launch_bounds__(256) void band_scatter_kernel(const BandArgs b) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 k = j < b.n_probe_cap ? b.skey_in[j] : b.kn;
  const u32 pos = band_run_atomic_add(b.key_cursor, k, k < b.kn);   // a run of equal keys lands in consecutive positions
  if (k >= b.kn) return;                               // joins nothing: no position
  const_cast<u32*>(b.perm)[pos] = (u32)j;
  if (b.compact) { b.rec_s[pos] = b.rec[j]; return; }
  b.rec_s[pos] = b.rec[2 * j];
  b.aux_s[pos] = b.rec[2 * j + 1];
}

// ---- per group entry: the chain's look-ups, once ------------------------------------------------------------------
struct BandEntry { u32 brow, r0, r1, r2, xb0, xb1, nq; bool ok; };
__device__ __forceinline__ BandEntry band_entry(const BandArgs& b, u32 pos, bool live) {
  BandEntry e;
  e.brow = live ? (b.csr_rows ? b.csr_rows[pos] : pos) : 0u;
  e.ok = live;
  // every look-up is issued on its own (none waits for another's verdict): one L2 round trip for all stages
  u32 r[kMaxChain] = {kNil, kNil, kNil};
#pragma unroll
  for (u32 t = 0; t < (u32)kMaxChain; t++) {
    if (t >= b.n_stages) continue;
    const BandStage& st = b.stage[t];
    const u32 key = st.key_col[e.brow];
    const u32 d = key - st.kmin;
    const bool in = live && key != 0 && d < st.kn;    // null keys never join
    const u32 row = st.direct[in ? d : 0u];
    r[t] = in ? row : kNil;
  }
  long long xv[2] = {0, 0}; bool xin[2] = {true, true};
#pragma unroll
  for (u32 w = 0; w < 2; w++) {
    if (w >= b.n_win) continue;
    const BandWin& bw = b.win[w];
    const u32 key = bw.key_col[e.brow];
    const u32 d = key - bw.vkmin;
    xin[w] = live && key != 0 && d < bw.vkn;
    xv[w] = bw.val[xin[w] ? d : 0u];
  }
#pragma unroll
  for (u32 t = 0; t < (u32)kMaxChain; t++) if (t < b.n_stages) e.ok = e.ok && r[t] != kNil;   // inner join: no stage row, no match
  e.r0 = r[0]; e.r1 = r[1]; e.r2 = r[2];
  u32 xb[2] = {1u, 1u};
#pragma unroll
  for (u32 w = 0; w < 2; w++) {
    if (w >= b.n_win) continue;
    xb[w] = (xin[w] && xv[w] != INT64_MIN) ? (u32)((unsigned long long)xv[w] - (unsigned long long)b.win[w].vbase) + 1u : 0u;
    e.ok = e.ok && xb[w] != 0u;
  }
  if (b.has_post) {   // the former build-side FilterExec `col <=|!=> literal`
    const u32 v = b.post_col[e.brow];
    e.ok = e.ok && v != 0 && b.post_lit != 0 && ((v == b.post_lit) == (b.post_is_eq != 0));
  }
  e.nq = 0;
  if (b.has_neq) { e.nq = b.neq_build[e.brow]; e.ok = e.ok && e.nq != 0; }
  e.xb0 = e.ok ? xb[0] : 0u;                          // a dead entry fails every window, the trivial one included
  e.xb1 = xb[1];
  return e;
}

// The stage look-ups once per distinct key value (all stages keyed by the same build column): 32 B per key.
__global__ __launch_bounds__(256) void band_pt_kernel(const BandArgs b) {
  const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= b.pt_n) return;
  const u32 key = b.pt_min + idx;
  bool ok = key != 0;
  u32 r[kMaxChain] = {kNil, kNil, kNil};
#pragma unroll
  for (u32 t = 0; t < (u32)kMaxChain; t++) {
    if (t >= b.n_stages) continue;
    const BandStage& st = b.stage[t];
    const u32 d = key - st.kmin;
    const bool in = key != 0 && d < st.kn;
    const u32 row = st.direct[in ? d : 0u];
    r[t] = in ? row : kNil;
    ok = ok && r[t] != kNil;
  }
  u32 xb[2] = {1u, 1u};
#pragma unroll
  for (u32 w = 0; w < 2; w++) {
    if (w >= b.n_win) continue;
    const BandWin& bw = b.win[w];
    const u32 d = key - bw.vkmin;
    const bool in = key != 0 && d < bw.vkn;
    const long long x = bw.val[in ? d : 0u];
    xb[w] = (in && x != INT64_MIN) ? (u32)((unsigned long long)x - (unsigned long long)bw.vbase) + 1u : 0u;
    ok = ok && xb[w] != 0u;
  }
  u32 ov[kBandMaxSideCols] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (u32 u = 0; u < kBandMaxSideCols; u++) {
    if (u >= b.n_entry_cols || !ok) continue;
    const ColRef c = b.entry_col[u];
    if (c.src >= 2) ov[u] = c.ptr[c.src == 2 ? r[0] : c.src == 3 ? r[1] : r[2]];
  }
  b.pt[2 * (u64)idx] = make_uint4(ok ? xb[0] : 0u, xb[1], ok ? 1u : 0u, 0u);
  b.pt[2 * (u64)idx + 1] = make_uint4(ov[0], ov[1], ov[2], ov[3]);
}
// One pass over the build side in CSR order: what the pair test needs of every entry (16 B) and the entry's output values.
__global__ __launch_bounds__(256) void band_entries_kernel(const BandArgs b) {
  const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= b.n_entries) return;
  if (b.pt == nullptr) {     // stages keyed by different columns: every entry does its own look-ups
    const BandEntry en = band_entry(b, (u32)pos, true);
    b.et[pos] = make_uint4(en.xb0, en.xb1, en.nq, en.ok ? 1u : 0u);
#pragma unroll
    for (u32 u = 0; u < kBandMaxSideCols; u++) {
      if (u >= b.n_entry_cols) continue;
      const ColRef c = b.entry_col[u];
      const u32 row = c.src == 1 ? en.brow : c.src == 2 ? en.r0 : c.src == 3 ? en.r1 : en.r2;
      b.eo[u][pos] = en.ok ? c.ptr[row] : 0u;
    }
    return;
  }
  const u32 brow = b.csr_rows ? b.csr_rows[pos] : (u32)pos;
  const u32 key = b.pt_key_col[brow];
  const u32 d = key - b.pt_min;
  const bool in = key != 0 && d < b.pt_n;
  const uint4 a0 = b.pt[2 * (u64)(in ? d : 0u)], a1 = b.pt[2 * (u64)(in ? d : 0u) + 1];
  bool ok = in && a0.z != 0;
  if (b.has_post) { const u32 v = b.post_col[brow]; ok = ok && v != 0 && b.post_lit != 0 && ((v == b.post_lit) == (b.post_is_eq != 0)); }
  u32 nq = 0;
  if (b.has_neq) { nq = b.neq_build[brow]; ok = ok && nq != 0; }
  b.et[pos] = make_uint4(ok ? a0.x : 0u, a0.y, nq, ok ? 1u : 0u);
#pragma unroll
  for (u32 u = 0; u < kBandMaxSideCols; u++) {
    if (u >= b.n_entry_cols) continue;
    const ColRef c = b.entry_col[u];
    const u32 v = c.src == 1 ? c.ptr[brow] : (u == 0 ? a1.x : u == 1 ? a1.y : u == 2 ? a1.z : a1.w);
    b.eo[u][pos] = ok ? v : 0u;
  }
}
// Block descriptors: key k owns the blocks boff[k] .. boff[k + 1), entry chunk major.
__global__ __launch_bounds__(256) void band_desc_kernel(const BandArgs b) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0 && b.n_blocks_out) *b.n_blocks_out = b.boff[b.kn];
  if (k >= b.kn) return;
  const u32 e0 = b.csr_off[k], e1 = b.csr_off[k + 1], p0 = b.poff[k], p1 = b.poff[k + 1];
  if (e0 >= e1 || p0 >= p1) return;
  u32 blk = b.boff[k];
  for (u32 eb = e0; eb < e1; eb += 64)
    for (u32 rb = p0; rb < p1; rb += 64, blk++)
      if (blk < b.max_blocks) b.bdesc[blk] = make_uint4(eb, e1 - eb < 64 ? e1 - eb : 64, rb, p1 - rb < 64 ? p1 - rb : 64);
}

// ---- the pair tests: one wave per 64 x 64 block --------------------------------------------------------------------
// lane = probe row (its decoded record in registers); the block's entries are wave-uniform: staged in LDS once per block and
// broadcast to all lanes — per pair two unsigned
// range checks and one id compare (6 VALU instructions per pair), no vector memory and no LDS inside the loop.  Measured: the
// kernel is NOT bound by those instructions (7 -> 6 per pair changed nothing; 200 k short waves with exposed scalar-load
// latency are what it waits for; persistent waves made it slower, 240 -> 325 us).
// NWIN = window stages (1..2; no window = one trivial window); NEQ = base filter: 0 none / 1 `!=` / 2 `=` / 3 `!=` by entry index (BandArgs::neq_self).
struct BandEntry8 { uint4 a[8]; };
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
template <int NWIN, int NEQ, bool PACK>
__global__ __launch_bounds__(256) void band_mask_kernel(const BandArgs b) {
  __shared__ uint4 ent[4][64];
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // one wave per block; the grid is sized from the previous execution's block count, so a wave strides on in the rare case
  // that there are more blocks than waves (sizing the grid by the upper bound launches twice as many waves as there are blocks)
  const u32 n_all = b.boff[b.kn];
  const u32 n_blocks = n_all < b.max_blocks ? n_all : b.max_blocks;
  for (u32 blk = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)); blk < n_blocks; blk += gridDim.x * 4u) {
  const uint4 d = b.bdesc[blk];
  const u32 eb = __builtin_amdgcn_readfirstlane(d.x), ne = __builtin_amdgcn_readfirstlane(d.y);
  const u32 rb = __builtin_amdgcn_readfirstlane(d.z), nr = __builtin_amdgcn_readfirstlane(d.w);
  uint4 rec = PACK ? make_uint4(0x0001FFFFu, 0u, 0u, 0u) : make_uint4(kBandInvalidLo, 0u, 1u, 0u);   // (nothing passes; PACK: the decode pass stored the packed form)
  u32 x = 0;
  if (lane < nr) { rec = b.rec_s[rb + lane]; if (NEQ) x = b.compact ? rec.z : b.aux_s[rb + lane].x; }   // (compact: the 16-byte record carries the id operand itself)
  // The block's 64 entries go through LDS: lane e fetches entry e (one coalesced 1 KB load per block), every test then reads
  // ITS entry with a broadcast ds_read_b128 (all lanes one address).  Scalar loads (s_load_dwordx8 from the entry table) looked
  // cheaper — no LDS, operands in SGPRs — but 200 k blocks x 1 KB through the scalar caches is what the kernel then waits
  // for (measured: VALU work 7 -> 6 instructions per pair changed nothing, more resident waves made it slower).
  u32 plo = 0, pw = 0;                                 // PACK: both windows' intervals as 2 x 16 bits
  {
    uint4 q = b.et[eb + lane];                          // the table is padded: reading past the group is harmless
    if (PACK) {
      q.x = (q.x & 0xFFFFu) | (q.y << 16);              // both windows' operands in one word
      plo = rec.x; pw = rec.y;
    }
    ent[wave][lane] = q;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  u32 m[2] = {0u, 0u};
#pragma unroll
  for (u32 h = 0; h < 2; h++) {
    if (h * 32 >= ne) continue;                       // wave-uniform
    u32 acc = 0;
    const u32 left = ne - h * 32;                      // entries of this half (wave-uniform): tested 8 at a time, not 32
    const u32 g_end = left >= 32 ? 4u : (left + 7) >> 3;
#pragma unroll 1
    for (u32 g = 0; g < g_end; g++) {
      uint4 q8[8];
#pragma unroll
      for (u32 e = 0; e < 8; e++) q8[e] = ent[wave][h * 32 + g * 8 + e];
#pragma unroll
      for (u32 e = 0; e < 8; e++) {
        const uint4 q = q8[e];
        // the verdict of the 64 lanes IS the compares' lane mask (SGPR pairs, AND-ed on the scalar unit): shifting it into
        // every lane's word is ONE add-with-carry (acc + acc + carry) instead of a select and an or per pair
        u64 lanes;
        if (PACK) {   // (x - lo) <= w in both 16-bit halves  <=>  min(x - lo, w) == x - lo as a 32-bit word: 3 instructions for two windows
          const us2 d = __builtin_bit_cast(us2, q.x) - __builtin_bit_cast(us2, plo);
          const us2 mn = __builtin_elementwise_min(d, __builtin_bit_cast(us2, pw));
          lanes = __builtin_amdgcn_ballot_w64(__builtin_bit_cast(u32, mn) == __builtin_bit_cast(u32, d));
        } else {
          lanes = __builtin_amdgcn_ballot_w64((q.x - rec.x) <= rec.y);
          if (NWIN > 1) lanes &= __builtin_amdgcn_ballot_w64((q.y - rec.z) <= rec.w);
        }
        if (NEQ == 1) lanes &= __builtin_amdgcn_ballot_w64(q.z != x);
        if (NEQ == 2) lanes &= __builtin_amdgcn_ballot_w64(q.z == x);
        u64 carry_out;
        asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(acc), "=s"(carry_out) : "v"(acc), "s"(lanes));
      }
    }
    m[h] = __builtin_bitreverse32(g_end == 4 ? acc : acc << (32 - 8 * g_end));   // entry 0 was shifted in first: it belongs at bit 31
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the staging area is free for the wave's next block
  // entries past the group's end belong to the next key: their bits do not count
  const u64 live = ne >= 64 ? ~0ull : ((1ull << ne) - 1ull);
  u64 mask = ((((u64)m[1]) << 32) | m[0]) & live;
  if (NEQ == 3) { const u32 self = x - eb; if (self < 64u) mask &= ~(1ull << self); }   // the row's own entry, if it is in this chunk (x = its index)
  b.masks[(u64)blk * 64 + lane] = mask;
  const u32 c = wave_incl_scan((u32)__popcll(mask));      // (DPP: lane 63 holds the block's count)
  if (lane == 63) b.bcount[blk] = c;
  }
}

// Rare: probe rows whose window operands are not all xsd:integer (or overflow i64) take the full typed-value semantics,
// pair by pair (stage_filter_slow), and patch their bits and their block's count.  Runs after the mask kernel; exits at
// once when the decode pass met no such row.
__global__ __launch_bounds__(256) void band_slow_kernel(const LdsJoinArgs* ap, const BandArgs b) {
  if (*b.slow_rows == 0) return;
  const LdsJoinArgs& a = *ap;
  const u32 lane = threadIdx.x & 63;
  // (persistent waves: the usual case — no slow row at all — must not cost the launch of one wave per block)
  const u32 n_all = b.boff[b.kn];
  const u32 n_blocks = n_all < b.max_blocks ? n_all : b.max_blocks;
  const u32 stride = gridDim.x * 4u;
  for (u32 blk = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6)); blk < n_blocks; blk += stride) {
  const uint4 d = b.bdesc[blk];
  const u32 eb = d.x, ne = d.y, rb = d.z, nr = d.w;
  u32 j = 0; uint4 aux = make_uint4(0u, 0u, 0u, 0u);
  if (lane < nr) { j = b.perm[rb + lane]; aux = b.aux_s[rb + lane]; }
  const bool slow = (aux.y & 1u) != 0;
  if (!__any(slow)) continue;
  const BandEntry en = band_entry(b, eb + lane, lane < ne);
  u32 m_lo = 0, m_hi = 0;
  for (u32 e = 0; e < ne; e++) {
    const bool eok = __builtin_amdgcn_readlane((u32)en.ok, e) != 0;
    const u32 brow = __builtin_amdgcn_readlane(en.brow, e);
    const u32 r0 = __builtin_amdgcn_readlane(en.r0, e), r1 = __builtin_amdgcn_readlane(en.r1, e), r2 = __builtin_amdgcn_readlane(en.r2, e);
    const u32 q = __builtin_amdgcn_readlane(en.nq, e);
    if (!slow || !eok) continue;
    bool pass = true;
    for (u32 w = 0; w < b.n_win; w++) {
      const u32 t = b.win[w].stage;
      pass = pass && stage_filter_slow(a, a.chain[t], brow, j, t == 0 ? r0 : t == 1 ? r1 : r2);
    }
    if (b.has_neq) pass = pass && ((q == aux.x) == (b.neq_is_eq != 0));
    if (pass) { if (e < 32) m_lo |= 1u << e; else m_hi |= 1u << (e - 32); }
  }
  if (slow) b.masks[(u64)blk * 64 + lane] |= ((u64)m_hi << 32) | m_lo;
  u32 c = (u32)__popc(m_lo) + (u32)__popc(m_hi);
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1) c += __shfl_xor(c, dd, 64);
  if (lane == 0 && c) b.bcount[blk] += c;
  }
}

// ---- bits -> rows: one wave per block --------------------------------------------------------------------------------
// Every lane (= probe row) lists its surviving entries in LDS at its prefix position, then the wave writes the block's
// output rows 64 at a time — consecutive rows, every column one coalesced 256-byte store — fetching each value from the
// owning lane's registers (ds_bpermute).
constexpr u32 kBandList = 1024;   // survivors listed at a time: a block with more is emitted 16 rows at a time (16 x 64 <= 1024)
// NCOLS = output columns (1 .. 6): a template parameter so that only the live columns' pointers and selectors sit in SGPRs (the
// six-column form kept 18 of them pinned and spilled scalars into vector lanes inside the loops)
template <int NCOLS>
__global__ __launch_bounds__(256) void band_emit_kernel(const BandArgs b) {
  __shared__ unsigned short list[4][kBandList];
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32 n_all = b.boff[b.kn];
  const u32 n_blocks = n_all < b.max_blocks ? n_all : b.max_blocks;
  // the column schedule, out of the argument block once per wave (constant indices: SGPRs)
  u32* outp[NCOLS]; u32 out_sel[NCOLS];
#pragma unroll
  for (u32 oc = 0; oc < (u32)NCOLS; oc++) { outp[oc] = b.out[oc]; out_sel[oc] = b.out_sel[oc]; }
  const u64 out_cap = b.out_cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {   // the exact total, whether or not it fitted (like the fused join kernel's count)
    const u64 total = b.bofs[b.max_blocks];
    *b.n_out_dev = total;
    if (total > b.out_cap) *b.overflow = 1u;
  }
  for (u32 blk = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + wave); blk < n_blocks; blk += gridDim.x * 4u) {   // (grid ~ blocks, see band_mask_kernel)
  // everything that is indexed by the block number in ONE round trip (count, descriptor, output offset, the rows' bits), before
  // the count decides whether the block has anything to emit: a wave lives for its memory round trips (one block each, 8 waves
  // per SIMD hide nothing of a chain), and count -> descriptor -> rows / entries were three of them
  const u32 tot_v = b.bcount[blk];
  const uint4 d = b.bdesc[blk];
  u64 run = b.bofs[blk];
  const u64 mask_raw = b.masks[(u64)blk * 64 + lane];
  asm volatile("" :: "v"(tot_v), "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w), "v"(run), "v"(mask_raw));   // (all four loads issued before the first wait: the compiler would sink them below the branch)
  const u32 tot = __builtin_amdgcn_readfirstlane(tot_v);
  if (tot == 0) continue;
  const u32 eb = __builtin_amdgcn_readfirstlane(d.x), ne = __builtin_amdgcn_readfirstlane(d.y);
  const u32 rb = __builtin_amdgcn_readfirstlane(d.z), nr = __builtin_amdgcn_readfirstlane(d.w);
  const u64 mask_all = lane < nr ? mask_raw : 0ull;
  u32 ev[kBandMaxSideCols], rv[kBandMaxRowCols];   // output values of the entry (lane = entry) / of the probe row (lane = row)
  {
    uint4 aux = make_uint4(0u, 0u, 0u, 0u);
    if (lane < nr) aux = b.compact ? b.rec_s[rb + lane] : b.aux_s[rb + lane];
    rv[0] = b.compact ? aux.w : aux.z; rv[1] = b.compact ? 0u : aux.w;
  }
#pragma unroll
  for (u32 u = 0; u < kBandMaxSideCols; u++) { ev[u] = 0; if (u < b.n_entry_cols && lane < ne) ev[u] = b.eo[u][eb + lane]; }
  // per output column, once per block: the register the column comes from (row value of lane r / entry value of lane e) — the
  // selection is the same for every survivor, and the kernel is VALU-bound (320 vector instructions per block: 67 % of the
  // SIMDs' issue slots in the SQ counters), so nothing that is uniform per block is computed per survivor
  u32 srcv[NCOLS];
#pragma unroll
  for (u32 oc = 0; oc < (u32)NCOLS; oc++) {
    const u32 sel = out_sel[oc];
    u32 src = rv[0];
    src = sel == 1 ? rv[1] : src;
#pragma unroll
    for (u32 u = 0; u < kBandMaxSideCols; u++) src = sel == 2 + u ? ev[u] : src;
    srcv[oc] = src;
  }
  // the block's output range starts at a wave-uniform offset: scalar base + 32-bit lane offset per store
  u64 run_s = ((u64)__builtin_amdgcn_readfirstlane((u32)(run >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)run);
  const u32 step = tot <= kBandList ? 64u : 16u;   // rows listed per round (wave-uniform)
  for (u32 r0 = 0; r0 < nr; r0 += step) {
    u64 mask = (lane >= r0 && lane < r0 + step) ? mask_all : 0ull;
    const u32 cnt = (u32)__popcll(mask);
    const u32 incl = wave_incl_scan(cnt);
    const u32 n_round = __builtin_amdgcn_readfirstlane(__shfl(incl, 63, 64));
    if (n_round == 0) continue;                                         // wave-uniform
    u32 at = incl - cnt;
    while (mask) {                                                      // this lane's survivors, in entry order
      const u32 e = (u32)__ffsll((long long)mask) - 1u;
      mask &= mask - 1;
      list[wave][at++] = (unsigned short)((lane << 6) | e);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const u64 room = out_cap > run_s ? out_cap - run_s : 0ull;          // rows that still fit (the count stays exact either way)
    const u32 lim = room < (u64)n_round ? (u32)room : n_round;
    for (u32 t0 = 0; t0 < n_round; t0 += 64) {
      const u32 idx = t0 + lane;
      const u32 pr = idx < n_round ? (u32)list[wave][idx] : 0u;
      const u32 r = pr >> 6, e = pr & 63u;
      u32 v[NCOLS];
#pragma unroll
      for (u32 oc = 0; oc < (u32)NCOLS; oc++) v[oc] = __shfl(srcv[oc], out_sel[oc] < 2 ? r : e, 64);   // every column's permute in flight, then one wait
      if (idx < lim) {
#pragma unroll
        for (u32 oc = 0; oc < (u32)NCOLS; oc++) (outp[oc] + run_s)[idx] = v[oc];                          // (scalar base + lane offset)
      }
    }
    run_s += n_round;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");              // the list is free again
  }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
void launch_band_decode(const BandArgs& b, hipStream_t s) {
  if (b.n_probe_cap) hipLaunchKernelGGL(band_decode_kernel, grid256(b.n_probe_cap + 1), dim3(256), 0, s, b);   // (+ 1: the row after the last closes the key boundaries)
}
void launch_band_bounds(const u32* skey_sorted, u64 n, u32 kn, u32* poff, hipStream_t s) {
  hipLaunchKernelGGL(band_bounds_kernel, grid256(n + 1), dim3(256), 0, s, skey_sorted, n, kn, poff);
}
void launch_band_pt(const BandArgs& b, hipStream_t s) {
  if (b.pt && b.pt_n) hipLaunchKernelGGL(band_pt_kernel, grid256(b.pt_n), dim3(256), 0, s, b);
}
void launch_band_scatter(const BandArgs& b, hipStream_t s) {
  if (b.n_probe_cap) hipLaunchKernelGGL(band_scatter_kernel, grid256(b.n_probe_cap), dim3(256), 0, s, b);
}
void launch_band_rows(const BandArgs& b, hipStream_t s) {
  if (b.n_probe_cap) hipLaunchKernelGGL(band_rows_kernel, grid256(b.n_probe_cap), dim3(256), 0, s, b);
}
void launch_band_entries(const BandArgs& b, hipStream_t s) {
  if (b.n_entries) hipLaunchKernelGGL(band_entries_kernel, grid256(b.n_entries), dim3(256), 0, s, b);
}
void launch_band_desc(const BandArgs& b, hipStream_t s) {
  hipLaunchKernelGGL(band_desc_kernel, grid256(b.kn), dim3(256), 0, s, b);
}
template <int NWIN, bool PACK> static void launch_band_mask_w(const BandArgs& b, dim3 g, hipStream_t s) {
  const int neq = b.has_neq ? (b.neq_is_eq ? 2 : b.neq_self ? 3 : 1) : 0;
  if (neq == 3) { if constexpr (PACK) { hipLaunchKernelGGL((band_mask_kernel<NWIN, 3, true>), g, dim3(256), 0, s, b); return; } }
  if (neq == 0) hipLaunchKernelGGL((band_mask_kernel<NWIN, 0, PACK>), g, dim3(256), 0, s, b);
  else if (neq == 1) hipLaunchKernelGGL((band_mask_kernel<NWIN, 1, PACK>), g, dim3(256), 0, s, b);
  else hipLaunchKernelGGL((band_mask_kernel<NWIN, 2, PACK>), g, dim3(256), 0, s, b);
}
void launch_band_mask(const BandArgs& b, hipStream_t s) {
  const dim3 g((b.launch_blocks + 3) / 4 ? (b.launch_blocks + 3) / 4 : 1);   // the number of blocks lives on the device: surplus waves leave at once, missing ones are made up by striding
  // (one wave per block beats persistent waves here: 2048 / 4096 / 8192 workgroups striding over the blocks took 223 / 211 / 202 us against 187)
  if (b.pack16) { if (b.n_win == 2) launch_band_mask_w<2, true>(b, g, s); else launch_band_mask_w<1, true>(b, g, s); }
  else if (b.n_win == 2) launch_band_mask_w<2, false>(b, g, s); else launch_band_mask_w<1, false>(b, g, s);   // no window = one trivial window
}
void launch_band_slow(const LdsJoinArgs* a_dev, const BandArgs& b, hipStream_t s) {
  const u32 g = (b.max_blocks + 3) / 4;
  hipLaunchKernelGGL(band_slow_kernel, dim3(g < 2048u ? (g ? g : 1u) : 2048u), dim3(256), 0, s, a_dev, b);
}
void launch_band_emit(const BandArgs& b, hipStream_t s) {
  const dim3 g((b.launch_blocks + 3) / 4 ? (b.launch_blocks + 3) / 4 : 1);
  static_assert(kBandMaxRowCols + kBandMaxSideCols == 6, "one instantiation per column count");
  switch (b.n_out_cols) {
    case 1: hipLaunchKernelGGL(band_emit_kernel<1>, g, dim3(256), 0, s, b); return;
    case 2: hipLaunchKernelGGL(band_emit_kernel<2>, g, dim3(256), 0, s, b); return;
    case 3: hipLaunchKernelGGL(band_emit_kernel<3>, g, dim3(256), 0, s, b); return;
    case 4: hipLaunchKernelGGL(band_emit_kernel<4>, g, dim3(256), 0, s, b); return;
    case 5: hipLaunchKernelGGL(band_emit_kernel<5>, g, dim3(256), 0, s, b); return;
    case 6: hipLaunchKernelGGL(band_emit_kernel<6>, g, dim3(256), 0, s, b); return;
  }
  fail(RDFGPU_ERR_INVALID, "band join with %u output columns", b.n_out_cols);
}

// blocks (64 entries x 64 rows) per key, scanned in one go: boff[k] = first block of key k, boff[kn] = all blocks.  The per-key
// counts are the scan's INPUT ITERATOR (computed on the fly from the two offset arrays): no band_blocks launch, no count array.
struct BandBlocksOfKey {
  const u32* csr_off; const u32* poff; u32 kn;
  __host__ __device__ u32 operator()(u32 k) const {
    if (k >= kn) return 0u;
    const u32 e = csr_off[k + 1] - csr_off[k], r = poff[k + 1] - poff[k];
    return (e && r) ? ((e + 63) >> 6) * ((r + 63) >> 6) : 0u;
  }
  __device__ __forceinline__ uint4 group(u32 g, u32 n) const {   // (the one-workgroup scan's input: four consecutive keys)
    const u32 k = 4u * g;
    if (k + 4u <= kn && !((reinterpret_cast<uintptr_t>(csr_off) | reinterpret_cast<uintptr_t>(poff)) & 15u)) {   // five neighbouring offsets of each array: one 16-byte load + one word
      const uint4 c = reinterpret_cast<const uint4*>(csr_off)[g], p = reinterpret_cast<const uint4*>(poff)[g];
      const u32 c4 = csr_off[k + 4u], p4 = poff[k + 4u];
      auto blocks = [](u32 e, u32 r) { return (e && r) ? ((e + 63u) >> 6) * ((r + 63u) >> 6) : 0u; };
      return make_uint4(blocks(c.y - c.x, p.y - p.x), blocks(c.z - c.y, p.z - p.y), blocks(c.w - c.z, p.w - p.z), blocks(c4 - c.w, p4 - p.w));
    }
    return make_uint4(k < n ? (*this)(k) : 0u, k + 1u < n ? (*this)(k + 1u) : 0u, k + 2u < n ? (*this)(k + 2u) : 0u, k + 3u < n ? (*this)(k + 3u) : 0u);
  }
};
size_t band_blocks_scan_temp_bytes(u32 kn) {
  size_t bytes = 0;
  auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<u32>(0u), BandBlocksOfKey{nullptr, nullptr, kn});
  (void)rocprim::exclusive_scan(nullptr, bytes, in, (u32*)nullptr, 0u, (size_t)kn + 1, rocprim::plus<u32>());
  return bytes + 256;
}
void band_blocks_scan(const u32* csr_off, const u32* poff, u32 kn, u32* boff, void* temp, size_t temp_bytes, hipStream_t s) {
  if (launch_small_scan<false>(BandBlocksOfKey{csr_off, poff, kn}, boff, (u64)kn + 1, s)) return;
  auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<u32>(0u), BandBlocksOfKey{csr_off, poff, kn});
  RDFGPU_HIP(rocprim::exclusive_scan(temp, temp_bytes, in, boff, 0u, (size_t)kn + 1, rocprim::plus<u32>(), s));
}

// rocPRIM radix sort of (u32 key, u32 value) pairs on the low `bits` bits: the partition pass of the probe side
size_t sort_u32_temp_bytes(u64 n, u32 bits) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const u32*)nullptr, (u32*)nullptr, (const u32*)nullptr, (u32*)nullptr, (size_t)(n ? n : 1), 0, bits);
  return bytes + 256;
}
void sort_pairs_u32_u32(const u32* kin, u32* kout, const u32* vin, u32* vout, u64 n, u32 bits, void* temp, size_t temp_bytes, hipStream_t s) {
  if (!n) return;
  RDFGPU_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, 0, bits, s));
}

// min / max of a decoded value table (INT64_MIN = no row): the bias of the 32-bit window intervals
__global__ __launch_bounds__(256) void val_minmax_kernel(const long long* val, u64 n, long long* out) {
  long long lo = INT64_MAX, hi = INT64_MIN + 1;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    const long long v = val[i];
    if (v == INT64_MIN) continue;
    lo = v < lo ? v : lo; hi = v > hi ? v : hi;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const long long l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64); lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
  if ((threadIdx.x & 63) == 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}
void launch_val_minmax(const long long* val, u64 n, long long* out_minmax, hipStream_t s) {
  const u64 g = (n + 256 * 16 - 1) / (256 * 16);
  hipLaunchKernelGGL(val_minmax_kernel, dim3((unsigned)(g ? (g > 1024 ? 1024 : g) : 1)), dim3(256), 0, s, val, n, out_minmax);
}
// largest group of a CSR table (rows of one key)
__global__ __launch_bounds__(256) void csr_max_group_kernel(const u32* off, u32 kn, u32* out) {
  u32 m = 0;
  for (u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x; k < kn; k += (u64)gridDim.x * blockDim.x) { const u32 g = off[k + 1] - off[k]; m = g > m ? g : m; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const u32 o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
void launch_csr_max_group(const u32* off, u32 kn, u32* out_dev, hipStream_t s) {
  const u64 g = ((u64)kn + 256 * 8 - 1) / (256 * 8);
  hipLaunchKernelGGL(csr_max_group_kernel, dim3((unsigned)(g ? (g > 1024 ? 1024 : g) : 1)), dim3(256), 0, s, off, kn, out_dev);
}

// (kernels.hpp, preload_code_objects: the runtime loads a translation unit's code object at the first use of one of its kernels)
void preload_tu_band_join() { hipFuncAttributes at; RDFGPU_HIP(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(band_bounds_kernel))); }
}  // namespace rdfgpu
