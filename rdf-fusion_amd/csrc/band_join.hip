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
// one wave owns one key: the group's entries are decoded ONCE (stage look-ups, decoded window operands), the key's
// probe rows are decoded ONCE (window bounds as biased 32-bit intervals), and the |group| x |rows| pair tests are
// register compares (two unsigned range checks and one id compare per pair), 64 probe rows per wave-instruction.
// Survivors are 1 bit per pair in HBM; a device scan over the per-key counts places every key's output; a second pass
// expands the bits into rows.  No atomics, no candidate queue, deterministic output order.
//
//   band_keys_kernel    sort key of every probe row (key - kmin; kn = "joins nothing")            -> rocPRIM radix sort
//   band_bounds_kernel  poff[k] = first sorted position with key >= k (one pass, gaps filled)
//   band_blocks_kernel  blocks (64 entries x 64 rows) per key -> exclusive scan = first block of a key
//   band_decode_kernel  per sorted probe row: window bounds (checked i64 -> biased u32 interval), id operand, row id
//   band_mask_kernel    per key: the pair tests, 1 bit per pair, count per key                       (dominant)
//   band_emit_kernel    per key: bits -> output rows at koff[k] + running count
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "join_device.hpp"

namespace rdfgpu {

constexpr u32 kBandInvalidLo = 0xFFFFFFFFu;   // a row / window that nothing can pass: (x - 0xFFFFFFFF) <= 0 never holds for x < 2^32 - 1

// ---- partition of the probe side by key ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_keys_kernel(const u32* key_col, const u64* n_dev, u64 cap, u32 kmin, u32 kn, u32* skey, u32* sval) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cap) return;
  const u64 n = live_rows(n_dev, cap);
  u32 k = kn;
  if (j < n) { const u32 v = key_col[j]; const u32 d = v - kmin; if (v != 0 && d < kn) k = d; }   // null keys never join
  skey[j] = k; sval[j] = (u32)j;
}
// poff[k] = number of sorted rows with key < k, for k = 0 .. kn (kn + 1 entries): every position whose key differs
// from its predecessor's fills the keys in between.
__global__ __launch_bounds__(256) void band_bounds_kernel(const u32* skey_sorted, u64 n, u32 kn, u32* poff) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  const u32 prev = i > 0 ? skey_sorted[i - 1] + 1u : 0u;          // first key not yet answered
  const u32 cur = i < n ? skey_sorted[i] : kn;                     // keys <= cur start at or before i
  for (u32 k = prev; k <= cur && k <= kn; k++) poff[k] = (u32)i;
}
__global__ __launch_bounds__(256) void band_blocks_kernel(const u32* csr_off, const u32* poff, u32 kn, u32* nblk) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > kn) return;
  u32 b = 0;
  if (k < kn) {
    const u32 e = csr_off[k + 1] - csr_off[k], r = poff[k + 1] - poff[k];
    b = (e && r) ? ((e + 63) >> 6) * ((r + 63) >> 6) : 0u;
  }
  nblk[k] = b;
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
  const int4 r0 = tv[iy0], r1 = tv[iy1];
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
__global__ __launch_bounds__(256) void band_decode_kernel(const LdsJoinArgs a, const BandArgs b) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b.n_sorted) return;
  if (b.skey[i] >= a.direct_n) return;                 // beyond the last joining row: never read
  const u32 j = b.perm[i];
  uint4 rec = make_uint4(1u, 0u, 1u, 0u);              // no window: entries carry x_b = 1 (0 when the entry is dead)
  u32 flags = 0;
#pragma unroll
  for (u32 w = 0; w < 2; w++) {
    if (w >= b.n_win) continue;
    const BandWin& bw = b.win[w];
    long long lo, hi;
    uint2 iv = make_uint2(kBandInvalidLo, 0u);
    if (band_window_of(a.tt, bw, bw.y0[j], bw.y1[j], lo, hi)) iv = band_interval(lo, hi, bw.vbase);
    else flags |= 1u;                                  // slow row: full semantics per pair
    if (w == 0) { rec.x = iv.x; rec.y = iv.y; } else { rec.z = iv.x; rec.w = iv.y; }
  }
  u32 x = 0;
  if (b.has_neq) { x = b.neq_probe[j]; if (x == 0) { rec.x = kBandInvalidLo; rec.y = 0u; flags = 2u; } }   // null => the comparison is not `true`
  if (flags & 1u) { rec.x = kBandInvalidLo; rec.y = 0u; }   // the fast test must not pass a slow row
  b.prec[i] = rec;
  b.paux[i] = make_uint4(x, j, flags, 0u);
}

// ---- per group entry: the chain's look-ups, once ------------------------------------------------------------------
struct BandEntry { u32 brow, r0, r1, r2, xb0, xb1, nq; bool ok; };
__device__ __forceinline__ BandEntry band_entry(const LdsJoinArgs& a, const BandArgs& b, u32 pos, bool live) {
  BandEntry e;
  e.brow = live ? (a.csr_rows ? a.csr_rows[pos] : pos) : 0u;
  e.ok = live;
  u32 r[kMaxChain] = {kNil, kNil, kNil};
#pragma unroll
  for (u32 t = 0; t < (u32)kMaxChain; t++) {
    if (t >= a.n_chain) continue;
    const ChainStage& st = a.chain[t];
    const u32 key = st.key.ptr[e.brow];               // key.src == 1: a build column (host-checked)
    const u32 d = key - st.kmin;
    const bool in = e.ok && key != 0 && d < st.kn;    // null keys never join
    r[t] = st.direct[in ? d : 0u];
    r[t] = in ? r[t] : kNil;
    e.ok = e.ok && r[t] != kNil;                      // inner join: an entry without a stage row joins nothing
  }
  e.r0 = r[0]; e.r1 = r[1]; e.r2 = r[2];
  u32 xb[2] = {1u, 1u};
#pragma unroll
  for (u32 w = 0; w < 2; w++) {
    if (w >= b.n_win) continue;
    const BandWin& bw = b.win[w];
    const u32 key = bw.key_col[e.brow];
    const u32 d = key - bw.vkmin;
    const bool in = e.ok && key != 0 && d < bw.vkn;
    const long long x = bw.val[in ? d : 0u];
    xb[w] = (in && x != INT64_MIN) ? (u32)((unsigned long long)x - (unsigned long long)bw.vbase) + 1u : 0u;
    e.ok = e.ok && xb[w] != 0u;
  }
  if (a.has_post) {   // the former build-side FilterExec `col <=|!=> literal` (host-checked: a build column)
    const u32 v = a.cols[a.post.col][e.brow];
    e.ok = e.ok && v != 0 && a.post.lit != 0 && ((v == a.post.lit) == (a.post.is_eq != 0));
  }
  e.nq = 0;
  if (b.has_neq) { e.nq = b.neq_build[e.brow]; e.ok = e.ok && e.nq != 0; }
  e.xb0 = e.ok ? xb[0] : 0u;                          // a dead entry fails every window, the trivial one included
  e.xb1 = xb[1];
  return e;
}

// ---- the pair tests: one wave per key ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_mask_kernel(const LdsJoinArgs a, const BandArgs b) {
  const u32 lane = threadIdx.x & 63;
  const u32 k = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
  if (k >= a.direct_n) return;
  const u32 e0 = a.csr_off[k], e1 = a.csr_off[k + 1], p0 = b.poff[k], p1 = b.poff[k + 1];
  if (e0 >= e1 || p0 >= p1) { if (lane == 0) b.kcount[k] = 0; return; }
  const u32 nrc = (p1 - p0 + 63) >> 6;
  u32 blk = b.boff[k];
  u32 total = 0;
  for (u32 eb = e0; eb < e1; eb += 64) {
    const u32 ne = e1 - eb < 64 ? e1 - eb : 64;       // wave-uniform
    const BandEntry en = band_entry(a, b, eb + lane, lane < ne);
    for (u32 rc = 0; rc < nrc; rc++, blk++) {
      const u32 i = p0 + rc * 64 + lane;
      const bool rlive = i < p1;
      uint4 rec = make_uint4(kBandInvalidLo, 0u, 1u, 0u);
      uint4 aux = make_uint4(0u, 0u, 0u, 0u);
      if (rlive) { rec = b.prec[i]; aux = b.paux[i]; }
      u32 m_lo = 0, m_hi = 0;
      // lane = probe row; the entries of the chunk are broadcast one by one (v_readlane): per pair two unsigned range
      // checks and one id compare, nothing is loaded inside the loop
      auto test = [&](u32 e) {
        const u32 x0 = __builtin_amdgcn_readlane(en.xb0, e), x1 = __builtin_amdgcn_readlane(en.xb1, e);
        bool pass = (x0 - rec.x) <= rec.y && (x1 - rec.z) <= rec.w;
        if (b.has_neq) { const u32 q = __builtin_amdgcn_readlane(en.nq, e); pass = pass && ((q == aux.x) == (b.neq_is_eq != 0)); }
        return pass;
      };
      const u32 n_lo = ne < 32 ? ne : 32;
      for (u32 e = 0; e < n_lo; e++) m_lo |= test(e) ? (1u << e) : 0u;
      for (u32 e = 32; e < ne; e++) m_hi |= test(e) ? (1u << (e - 32)) : 0u;
      if (__any((aux.z & 1u) != 0)) {
        // rare: a probe row whose window operands are not all xsd:integer — the full typed-value semantics, pair by pair
        for (u32 e = 0; e < ne; e++) {
          const bool eok = __builtin_amdgcn_readlane((u32)en.ok, e) != 0;
          const u32 brow = __builtin_amdgcn_readlane(en.brow, e);
          const u32 r0 = __builtin_amdgcn_readlane(en.r0, e), r1 = __builtin_amdgcn_readlane(en.r1, e), r2 = __builtin_amdgcn_readlane(en.r2, e);
          const u32 q = __builtin_amdgcn_readlane(en.nq, e);
          if (!(aux.z & 1u) || !eok) continue;
          bool pass = true;
#pragma unroll
          for (u32 w = 0; w < 2; w++) {
            if (w >= b.n_win) continue;
            const u32 t = b.win[w].stage;
            pass = pass && stage_filter_slow(a, a.chain[t], brow, aux.y, t == 0 ? r0 : t == 1 ? r1 : r2);
          }
          if (b.has_neq) pass = pass && ((q == aux.x) == (b.neq_is_eq != 0));
          if (pass) { if (e < 32) m_lo |= 1u << e; else m_hi |= 1u << (e - 32); }
        }
      }
      b.masks[(u64)blk * 64 + lane] = ((u64)m_hi << 32) | m_lo;
      u32 c = (u32)__popc(m_lo) + (u32)__popc(m_hi);
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
      total += c;
    }
  }
  if (lane == 0) b.kcount[k] = total;
}

// ---- bits -> rows ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_emit_kernel(const LdsJoinArgs a, const BandArgs b) {
  const u32 lane = threadIdx.x & 63;
  const u32 k = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
  if (k == 0 && lane == 0) {   // the exact total, whether or not it fitted (like the fused join kernel's count)
    const u64 total = b.koff[a.direct_n];
    *a.n_out_dev = total;
    if (total > a.out_cap) *a.overflow = 1u;
  }
  if (k >= a.direct_n) return;
  if (b.kcount[k] == 0) return;
  const u32 e0 = a.csr_off[k], e1 = a.csr_off[k + 1], p0 = b.poff[k], p1 = b.poff[k + 1];
  const u32 nrc = (p1 - p0 + 63) >> 6;
  u32 blk = b.boff[k];
  u64 run = b.koff[k];
  for (u32 eb = e0; eb < e1; eb += 64) {
    const u32 ne = e1 - eb < 64 ? e1 - eb : 64;
    const BandEntry en = band_entry(a, b, eb + lane, lane < ne);
    // output columns that come from the build row or from a stage's row: one value per entry (lane = entry)
    u32 ev[kBandMaxSideCols];
#pragma unroll
    for (u32 u = 0; u < kBandMaxSideCols; u++) {
      ev[u] = 0;
      if (u < b.n_entry_cols && en.ok) {
        const ColRef c = b.entry_col[u];
        const u32 row = c.src == 1 ? en.brow : c.src == 2 ? en.r0 : c.src == 3 ? en.r1 : en.r2;
        ev[u] = c.ptr[row];
      }
    }
    for (u32 rc = 0; rc < nrc; rc++, blk++) {
      const u32 i = p0 + rc * 64 + lane;
      u64 mask = i < p1 ? b.masks[(u64)blk * 64 + lane] : 0ull;
      const u32 cnt = (u32)__popcll(mask);
      const u32 incl = wave_incl_scan(cnt);
      const u32 tot = __shfl(incl, 63, 64);
      if (tot == 0) continue;                                           // wave-uniform
      u64 pos = run + (incl - cnt);
      run += tot;
      u32 rv[kBandMaxSideCols];                                         // output columns of the probe row (lane = row)
      const u32 j = mask ? b.paux[i].y : 0u;
#pragma unroll
      for (u32 u = 0; u < kBandMaxSideCols; u++) { rv[u] = 0; if (u < b.n_row_cols && mask) rv[u] = b.row_col[u][j]; }
      while (__any(mask != 0)) {
        const bool has = mask != 0;
        const u32 e = has ? (u32)__ffsll((long long)mask) - 1u : 0u;
        mask &= mask - 1;
        u32 ue = 0, ur = 0;
        for (u32 oc = 0; oc < a.n_out_cols; oc++) {                     // wave-uniform schedule of the output columns
          u32 v;
          if (b.out_from_row[oc]) {
            v = rv[0];
#pragma unroll
            for (u32 u = 1; u < kBandMaxSideCols; u++) v = ur == u ? rv[u] : v;
            ur++;
          } else {
            u32 src = ev[0];
#pragma unroll
            for (u32 u = 1; u < kBandMaxSideCols; u++) src = ue == u ? ev[u] : src;
            v = __shfl(src, e, 64);
            ue++;
          }
          if (has && pos < a.out_cap) a.out[oc][pos] = v;
        }
        pos += has ? 1u : 0u;
      }
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
static inline dim3 grid256(u64 n) { const u64 g = (n + 255) / 256; return dim3((unsigned)(g ? g : 1)); }
void launch_band_keys(const u32* key_col, const u64* n_dev, u64 cap, u32 kmin, u32 kn, u32* skey, u32* sval, hipStream_t s) {
  if (cap) hipLaunchKernelGGL(band_keys_kernel, grid256(cap), dim3(256), 0, s, key_col, n_dev, cap, kmin, kn, skey, sval);
}
void launch_band_bounds(const u32* skey_sorted, u64 n, u32 kn, u32* poff, hipStream_t s) {
  hipLaunchKernelGGL(band_bounds_kernel, grid256(n + 1), dim3(256), 0, s, skey_sorted, n, kn, poff);
}
void launch_band_blocks(const u32* csr_off, const u32* poff, u32 kn, u32* nblk, hipStream_t s) {
  hipLaunchKernelGGL(band_blocks_kernel, grid256((u64)kn + 1), dim3(256), 0, s, csr_off, poff, kn, nblk);
}
void launch_band_decode(const LdsJoinArgs& a, const BandArgs& b, hipStream_t s) {
  if (b.n_sorted) hipLaunchKernelGGL(band_decode_kernel, grid256(b.n_sorted), dim3(256), 0, s, a, b);
}
void launch_band_mask(const LdsJoinArgs& a, const BandArgs& b, hipStream_t s) {
  hipLaunchKernelGGL(band_mask_kernel, dim3((a.direct_n + 3) / 4), dim3(256), 0, s, a, b);
}
void launch_band_emit(const LdsJoinArgs& a, const BandArgs& b, hipStream_t s) {
  hipLaunchKernelGGL(band_emit_kernel, dim3((a.direct_n + 3) / 4), dim3(256), 0, s, a, b);
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

}  // namespace rdfgpu
