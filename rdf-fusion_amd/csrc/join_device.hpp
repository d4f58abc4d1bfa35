// join_device.hpp — device code of the fused hash-join kernel (lds_join_kernel) and its helpers, shared by the
// translation units that instantiate it (join_fs*.hip: one per join-filter shape, so the build parallelises) and by
// kernels.hip (table build kernels, dispatcher).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <mutex>

#include "kernels.hpp"

namespace rdfgpu {

constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kItems = 4;            // rows per lane
constexpr int kTile = kBlock * kItems;
static_assert(kTile == (int)kScanTile, "tile size");

__device__ __forceinline__ u32 lane_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}
__device__ __forceinline__ u64 live_rows(const u64* n_dev, u64 cap) {
  if (!n_dev) return cap;
  const u64 n = *n_dev;
  return n < cap ? n : cap;
}
static inline dim3 grid_for(u64 rows) { u64 g = (rows + kTile - 1) / kTile; return dim3((unsigned)(g ? g : 1)); }

// --------------------------------------------------------------------------------------------------
// K4+K5 fused, LDS-staged: the whole build side (<= 8192 rows) lives in ONE LDS open-addressing table
// per workgroup ({key0, row} slots, linear probing, load factor <= 0.5), built once per workgroup from
// L2 and then probed by that workgroup's share of the probe side, 512 x ITEMS rows per tile.
// Variable-cardinality output without a count pass over HBM: matches are compacted into wave-private
// LDS queues and leave in reserved, consecutive output ranges (see the kernel).  The total is always
// exact; if it exceeds the optimistic capacity the host re-runs with the exact size.
// This is the path every BSBM Q1/Q5 join takes after the engine's join reordering (build = the
// smaller input, cross products decomposed): J2/J3 build ~2 k rows and probe 285 k.
// --------------------------------------------------------------------------------------------------
constexpr int kLdsBlock = 512;

// Inclusive scan over the 64 lanes with DPP moves (no LDS round trips: six ds_bpermute + waits were ~600 cycles of latency and
// 18 address instructions per scan): four row_shr steps scan every 16-lane row, row_bcast15 / row_bcast31 carry the row totals
// on (GFX9 wave64).  Every lane of the wave must be active.
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1 (lanes without a source add 0)
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast15 into rows 1 and 3
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast31 into rows 2 and 3
  return v;
}
// ---- scan of up to 64 K counts by ONE workgroup ---------------------------------------------------------------------
// The scans inside a plan run over tile / key / block COUNTS (5 k - 64 k elements), not over rows: rocPRIM's device scan
// is two launches with a look-back between them (11 - 15 us each time, a tenth of a BGP scan + FILTER over 2^26 rows; three
// of them per 0.5 ms Q5 step).  One workgroup of 1024 lanes does it in one launch: lane t loads the 16-byte groups
// j * 1024 + t (coalesced, all J loads in flight), every wave scans its group sums with DPP moves, the J x 16 wave totals
// meet in LDS, wave 0 scans those, and the prefixes go back out as 16-byte stores.  Two barriers in all.
// `In` yields element i (a plain array, or a count computed on the fly from other arrays).
struct ScanArrayIn {
  const u32* p;
  __device__ __forceinline__ uint4 group(u32 g, u32 n) const {
    if (4u * g + 4u <= n) return reinterpret_cast<const uint4*>(p)[g];
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (4u * g < n) v.x = p[4u * g];
    if (4u * g + 1u < n) v.y = p[4u * g + 1u];
    if (4u * g + 2u < n) v.z = p[4u * g + 2u];
    return v;
  }
};
template <int J, bool INCLUSIVE, class In>
__global__ __launch_bounds__(1024) void small_scan_kernel(const In in, u32* out, u32 n) {
  static_assert(J == 4 || J == 8 || J == 16, "J x 16 wave totals are scanned by one wave, J / 4 per lane");
  __shared__ u32 wt[J * 16];
  const u32 t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  uint4 x[J]; u32 sum[J], incl[J];
#pragma unroll
  for (int j = 0; j < J; j++) x[j] = in.group((u32)j * 1024u + t, n);
#pragma unroll
  for (int j = 0; j < J; j++) {
    sum[j] = x[j].x + x[j].y + x[j].z + x[j].w;
    incl[j] = wave_incl_scan(sum[j]);
    if (lane == 63u) wt[j * 16 + (int)wave] = incl[j];
  }
  __syncthreads();
  if (wave == 0) {   // exclusive scan of the J x 16 totals, in (j, wave) order = element order
    constexpr int Q = J / 4;
    u32 a[Q], local = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) { a[q] = wt[lane * Q + q]; local += a[q]; }
    u32 run = wave_incl_scan(local) - local;
#pragma unroll
    for (int q = 0; q < Q; q++) { wt[lane * Q + q] = run; run += a[q]; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < J; j++) {
    const u32 g = (u32)j * 1024u + t;
    u32 b = wt[j * 16 + (int)wave] + incl[j] - sum[j];
    uint4 o;
    if (INCLUSIVE) { o.x = b + x[j].x; o.y = o.x + x[j].y; o.z = o.y + x[j].z; o.w = o.z + x[j].w; }
    else { o.x = b; o.y = b + x[j].x; o.z = o.y + x[j].y; o.w = o.z + x[j].z; }
    if (4u * g + 4u <= n) reinterpret_cast<uint4*>(out)[g] = o;
    else {
      if (4u * g < n) out[4u * g] = o.x;
      if (4u * g + 1u < n) out[4u * g + 1u] = o.y;
      if (4u * g + 2u < n) out[4u * g + 2u] = o.z;
    }
  }
}
// true: scanned by one workgroup; false: too many elements (or `out` not 16-byte aligned) — the caller takes rocPRIM's scan
template <bool INCLUSIVE, class In>
static inline bool launch_small_scan(const In& in, u32* out, u64 n, hipStream_t s) {
  if (n == 0) return true;
  if (n > kSmallScanElems || (reinterpret_cast<uintptr_t>(out) & 15u)) return false;
  if (n <= 4ull * 4096ull) hipLaunchKernelGGL((small_scan_kernel<4, INCLUSIVE, In>), dim3(1), dim3(1024), 0, s, in, out, (u32)n);
  else if (n <= 8ull * 4096ull) hipLaunchKernelGGL((small_scan_kernel<8, INCLUSIVE, In>), dim3(1), dim3(1024), 0, s, in, out, (u32)n);
  else hipLaunchKernelGGL((small_scan_kernel<16, INCLUSIVE, In>), dim3(1), dim3(1024), 0, s, in, out, (u32)n);
  return true;
}

// A build-side column at candidate `i`.  Normally i is the build row.  In range-index mode (LdsJoinArgs::range_link) a
// candidate carries its POSITION in the value-ordered CSR instead: the link column (the stage key, e.g. ?product) is
// then read from its value-ordered copy — consecutive positions, coalesced — and any other build column through the
// row ids of the index.
__device__ __forceinline__ u32 bcol(const LdsJoinArgs& a, const u32* ptr, u64 i) {
  if (a.range_link != nullptr) return ptr == a.range_link_col ? a.range_link[i] : ptr[a.range_rows[i]];   // wave-uniform branches
  return ptr[i];
}
__device__ __forceinline__ u32 ljoin_col(const LdsJoinArgs& a, u32 c, u64 i, u64 j) {  // column c of [left cols, right cols]
  const bool from_build = (c < a.n_left_cols) == (a.build_is_left != 0);   // wave-uniform
  return from_build ? bcol(a, a.cols[c], i) : a.cols[c][j];
}

// Join filter, specialised: FS 0 = none, 1 = generic VM, 3 = "window" — the BSBM Q5 shape
//   EBV(cmp1(ENC_TV(x), ADD|SUB(ENC_TV(y), lit1))) AND EBV(cmp2(ENC_TV(x'), ADD|SUB(ENC_TV(y'), lit2)))
// evaluated with all typed-value gathers issued back to back (one HBM/L2 latency instead of four).
__device__ __forceinline__ bool cmp_holds(u8 op, int o) {
  if (o == ORD_NONE) return false;   // error => null => not `true`
  return op == RDFGPU_EX_GT ? o > 0 : op == RDFGPU_EX_LT ? o < 0 : op == RDFGPU_EX_GEQ ? o >= 0
       : op == RDFGPU_EX_LEQ ? o <= 0 : op == RDFGPU_EX_EQ ? o == 0 : o != 0;
}
__device__ __forceinline__ Val lit_val(const TvLiteral& l) {
  Val y = val_tv_null(); y.tag = l.tag; y.flags = l.flags; y.aux = l.aux; y.lo = l.lo; y.hi = l.hi; return y;
}
// The filter comes in two halves.  ljoin_filter_fast decides the cheap, common cases in a handful of ops and is
// what the resolve phase runs four-wide; whatever it cannot decide (undecided = true) goes to ljoin_filter_slow,
// the full reference semantics, of which the kernel holds ONE copy run one candidate at a time — so the promotion
// machinery (i128 decimals, float/double casts) costs neither registers nor instruction cache on the fast path.
// The numeric-window predicate  cmp0(ENC_TV(x0), ENC_TV(y0) +/- lit0) AND cmp1(ENC_TV(x1), ENC_TV(y1) +/- lit1)  on four
// object ids.  window_fast decides the all-xsd:integer case (the BSBM numeric properties) with checked i64 arithmetic;
// window_slow is the full reference semantics.
// BRANCH-FREE on purpose: every lane-divergent `if` costs ~5 scalar instructions of exec-mask bookkeeping, and this
// runs once per candidate pair; dead or invalid lanes read entry 0 of the typed-value table (the null id: tag 0), which
// simply fails the all-integer test.
__device__ __forceinline__ bool window_fast(const TypedTable& tt, u32 ix0, u32 iy0, u32 ix1, u32 iy1, bool same,
                                            const TvLiteral& l0, const TvLiteral& l1, bool& undecided) {
  if (tt.n_ids == 0) { undecided = true; return false; }   // wave-uniform
  const u64 n_ids = tt.n_ids;
  ix0 = ix0 < n_ids ? ix0 : 0u; iy0 = iy0 < n_ids ? iy0 : 0u; ix1 = ix1 < n_ids ? ix1 : 0u; iy1 = iy1 < n_ids ? iy1 : 0u;
  const int4* tv = reinterpret_cast<const int4*>(tt.tv);
  const int4 rx0 = tv[ix0], ry0 = tv[iy0];
  const int4 rx1 = same ? rx0 : tv[ix1], ry1 = same ? ry0 : tv[iy1];   // `same` is wave-uniform
  const u32 tags = ((u32)rx0.w & 0xff) | (((u32)ry0.w & 0xff) << 8) | (((u32)rx1.w & 0xff) << 16) | (((u32)ry1.w & 0xff) << 24);
  undecided = tags != RDFGPU_TV_INTEGER * 0x01010101u || l0.tag != RDFGPU_TV_INTEGER || l1.tag != RDFGPU_TV_INTEGER;
  auto i64 = [](const int4& r) { return (long long)(((u64)(u32)r.y << 32) | (u32)r.x); };
  // y +/- lit as y + d with d = +/-lit (wave-uniform; lit = i64::MIN under SUB cannot be negated: left to the slow half)
  const bool neg0 = l0.arith_sub != 0, neg1 = l1.arith_sub != 0;
  undecided = undecided || (neg0 && l0.lo == INT64_MIN) || (neg1 && l1.lo == INT64_MIN);
  const long long d0 = neg0 ? -(long long)(l0.lo == INT64_MIN ? 0 : l0.lo) : (long long)l0.lo;
  const long long d1 = neg1 ? -(long long)(l1.lo == INT64_MIN ? 0 : l1.lo) : (long long)l1.lo;
  long long z0, z1;
  const bool o0 = __builtin_add_overflow(i64(ry0), d0, &z0), o1 = __builtin_add_overflow(i64(ry1), d1, &z1);
  const long long p0 = i64(rx0), p1 = i64(rx1);
  // cmp_holds as a 3-bit truth table over (less, equal, greater): wave-uniform masks, two selects per comparison
  auto mask_of = [](u8 op) -> u32 { return op == RDFGPU_EX_GT ? 4u : op == RDFGPU_EX_LT ? 1u : op == RDFGPU_EX_GEQ ? 6u : op == RDFGPU_EX_LEQ ? 3u : op == RDFGPU_EX_EQ ? 2u : 5u; };
  const u32 m0 = mask_of(l0.cmp_op), m1 = mask_of(l1.cmp_op);
  const u32 c0 = p0 < z0 ? 1u : p0 > z0 ? 4u : 2u, c1 = p1 < z1 ? 1u : p1 > z1 ? 4u : 2u;
  return !o0 && !o1 && (m0 & c0) != 0 && (m1 & c1) != 0;   // overflow => error => null => not `true`
}
// the same test with x already decoded (x0 == x1 = an xsd:integer from a slice's value table): only the y side is gathered
__device__ __forceinline__ bool window_fast_x(const TypedTable& tt, long long x, u32 iy0, u32 iy1, const TvLiteral& l0, const TvLiteral& l1, bool& undecided) {
  if (tt.n_ids == 0) { undecided = true; return false; }
  const u64 n_ids = tt.n_ids;
  iy0 = iy0 < n_ids ? iy0 : 0u; iy1 = iy1 < n_ids ? iy1 : 0u;
  const int4* tv = reinterpret_cast<const int4*>(tt.tv);
  const bool same = iy0 == iy1;
  const int4 ry0 = tv[iy0];
  const int4 ry1 = tv[iy1];
  (void)same;
  undecided = ((u32)ry0.w & 0xff) != RDFGPU_TV_INTEGER || ((u32)ry1.w & 0xff) != RDFGPU_TV_INTEGER || l0.tag != RDFGPU_TV_INTEGER || l1.tag != RDFGPU_TV_INTEGER;
  auto i64 = [](const int4& r) { return (long long)(((u64)(u32)r.y << 32) | (u32)r.x); };
  const bool neg0 = l0.arith_sub != 0, neg1 = l1.arith_sub != 0;
  undecided = undecided || (neg0 && l0.lo == INT64_MIN) || (neg1 && l1.lo == INT64_MIN);
  const long long d0 = neg0 ? -(long long)(l0.lo == INT64_MIN ? 0 : l0.lo) : (long long)l0.lo;
  const long long d1 = neg1 ? -(long long)(l1.lo == INT64_MIN ? 0 : l1.lo) : (long long)l1.lo;
  long long z0, z1;
  const bool o0 = __builtin_add_overflow(i64(ry0), d0, &z0), o1 = __builtin_add_overflow(i64(ry1), d1, &z1);
  auto mask_of = [](u8 op) -> u32 { return op == RDFGPU_EX_GT ? 4u : op == RDFGPU_EX_LT ? 1u : op == RDFGPU_EX_GEQ ? 6u : op == RDFGPU_EX_LEQ ? 3u : op == RDFGPU_EX_EQ ? 2u : 5u; };
  const u32 c0 = x < z0 ? 1u : x > z0 ? 4u : 2u, c1 = x < z1 ? 1u : x > z1 ? 4u : 2u;
  return !o0 && !o1 && (mask_of(l0.cmp_op) & c0) != 0 && (mask_of(l1.cmp_op) & c1) != 0;
}
// ... and with the two y operands known to be one column (same_y, wave-uniform): one typed-value gather per candidate
__device__ __forceinline__ bool window_fast_xy(const TypedTable& tt, long long x, u32 iy0, u32 iy1, bool same_y, const TvLiteral& l0, const TvLiteral& l1, bool& undecided) {
  if (tt.n_ids == 0) { undecided = true; return false; }
  const u64 n_ids = tt.n_ids;
  iy0 = iy0 < n_ids ? iy0 : 0u; iy1 = iy1 < n_ids ? iy1 : 0u;
  const int4* tv = reinterpret_cast<const int4*>(tt.tv);
  const int4 ry0 = tv[iy0];
  const int4 ry1 = same_y ? ry0 : tv[iy1];
  undecided = ((u32)ry0.w & 0xff) != RDFGPU_TV_INTEGER || ((u32)ry1.w & 0xff) != RDFGPU_TV_INTEGER || l0.tag != RDFGPU_TV_INTEGER || l1.tag != RDFGPU_TV_INTEGER;
  auto i64 = [](const int4& r) { return (long long)(((u64)(u32)r.y << 32) | (u32)r.x); };
  const bool neg0 = l0.arith_sub != 0, neg1 = l1.arith_sub != 0;
  undecided = undecided || (neg0 && l0.lo == INT64_MIN) || (neg1 && l1.lo == INT64_MIN);
  const long long d0 = neg0 ? -(long long)(l0.lo == INT64_MIN ? 0 : l0.lo) : (long long)l0.lo;
  const long long d1 = neg1 ? -(long long)(l1.lo == INT64_MIN ? 0 : l1.lo) : (long long)l1.lo;
  long long z0, z1;
  const bool o0 = __builtin_add_overflow(i64(ry0), d0, &z0), o1 = __builtin_add_overflow(i64(ry1), d1, &z1);
  auto mask_of = [](u8 op) -> u32 { return op == RDFGPU_EX_GT ? 4u : op == RDFGPU_EX_LT ? 1u : op == RDFGPU_EX_GEQ ? 6u : op == RDFGPU_EX_LEQ ? 3u : op == RDFGPU_EX_EQ ? 2u : 5u; };
  const u32 c0 = x < z0 ? 1u : x > z0 ? 4u : 2u, c1 = x < z1 ? 1u : x > z1 ? 4u : 2u;
  return !o0 && !o1 && (mask_of(l0.cmp_op) & c0) != 0 && (mask_of(l1.cmp_op) & c1) != 0;
}
__device__ __forceinline__ bool window_slow(const TypedTable& tt, u32 ix0, u32 iy0, u32 ix1, u32 iy1, const TvLiteral& l0, const TvLiteral& l1) {
  const Val x0 = enc_tv(tt, ix0), y0 = enc_tv(tt, iy0), x1 = enc_tv(tt, ix1), y1 = enc_tv(tt, iy1);
  const Val z0 = tv_arith(y0, lit_val(l0), l0.arith_sub != 0);
  const Val z1 = tv_arith(y1, lit_val(l1), l1.arith_sub != 0);
  return cmp_holds(l0.cmp_op, tv_partial_cmp(x0, z0)) && cmp_holds(l1.cmp_op, tv_partial_cmp(x1, z1));
}

template <int FS>
__device__ __forceinline__ bool ljoin_filter_fast(const LdsJoinArgs& a, u32 i, u32 j, bool& undecided) {
  undecided = false;
  if constexpr (FS == 0) return true;
  else if constexpr (FS == 2) {   // column <ID_EQ | ID_NEQ> column (e.g. `product != X` with the instance's X as a column)
    const u32 va = ljoin_col(a, a.idp.a, i, j), vb = ljoin_col(a, a.idp.b, i, j);
    if (va == 0 || vb == 0) return false;   // null => not `true`
    return (va == vb) == (a.idp.is_eq != 0);
  } else if constexpr (FS == 3) {
    const WindowFilter& w = a.win;
    const bool same = w.x0 == w.x1 && w.y0 == w.y1;   // wave-uniform
    const u32 ix0 = ljoin_col(a, w.x0, i, j), iy0 = ljoin_col(a, w.y0, i, j);
    const u32 ix1 = same ? ix0 : ljoin_col(a, w.x1, i, j), iy1 = same ? iy0 : ljoin_col(a, w.y1, i, j);
    return window_fast(a.tt, ix0, iy0, ix1, iy1, same, w.l0, w.l1, undecided);
  } else { undecided = true; return false; }
}
template <int FS>
__device__ __forceinline__ bool ljoin_filter_slow(const LdsJoinArgs& a, u32 i, u32 j) {
  if constexpr (FS == 3) {
    const WindowFilter& w = a.win;
    return window_slow(a.tt, ljoin_col(a, w.x0, i, j), ljoin_col(a, w.y0, i, j), ljoin_col(a, w.x1, i, j), ljoin_col(a, w.y1, i, j), w.l0, w.l1);
  } else if constexpr (FS == 1) {
    const Val r = eval_program(*a.prog, a.tt, [&](u32 col) { return ljoin_col(a, col, i, j); });
    return r.lo == 1;
  } else return false;   // FS 0 / 2 are always decided by the fast half
}
// ---- fused lookup chain (ChainStage): stage = direct-table lookup of a base key column + the stage's join filter ----
// (branch-free like window_fast: a dead lane passes live = false and reads row 0 of whatever it is pointed at)
__device__ __forceinline__ u32 chain_val(const LdsJoinArgs& a, const ColRef& c, u32 i, u32 j, u32 r) { return c.src == 1 ? bcol(a, c.ptr, i) : c.ptr[c.src == 0 ? j : r]; }
__device__ __forceinline__ u32 chain_lookup(const LdsJoinArgs& a, const ChainStage& st, u32 i, u32 j, bool live = true) {
  const u32 key = st.key.src ? bcol(a, st.key.ptr, live ? i : 0u) : st.key.ptr[live ? j : 0u];
  const u32 d = key - st.kmin;
  const bool in = live && key != 0 && d < st.kn;          // null keys never join
  const u32 row = st.direct[in ? d : 0u];
  return in ? row : kNil;                                  // kNil = no row with this key
}
__device__ __forceinline__ bool stage_filter_fast(const LdsJoinArgs& a, const ChainStage& st, u32 i, u32 j, u32 r, bool& undecided) {
  undecided = false;
  if (st.fs == 2) {   // wave-uniform
    const u32 va = chain_val(a, st.f[0], i, j, r), vb = chain_val(a, st.f[1], i, j, r);
    return va != 0 && vb != 0 && (va == vb) == (st.is_eq != 0);
  }
  const bool same = st.f[0].ptr == st.f[2].ptr && st.f[0].src == st.f[2].src && st.f[1].ptr == st.f[3].ptr && st.f[1].src == st.f[3].src;
  const u32 ix0 = chain_val(a, st.f[0], i, j, r), iy0 = chain_val(a, st.f[1], i, j, r);
  const u32 ix1 = same ? ix0 : chain_val(a, st.f[2], i, j, r), iy1 = same ? iy0 : chain_val(a, st.f[3], i, j, r);
  return window_fast(a.tt, ix0, iy0, ix1, iy1, same, st.l0, st.l1, undecided);
}
// the stage's filter with the full reference semantics (fs 2 is always decided by the fast half)
__device__ __forceinline__ bool stage_filter_slow(const LdsJoinArgs& a, const ChainStage& st, u32 i, u32 j, u32 r) {
  if (st.fs != 3) { bool und; return stage_filter_fast(a, st, i, j, r, und); }
  return window_slow(a.tt, chain_val(a, st.f[0], i, j, r), chain_val(a, st.f[1], i, j, r), chain_val(a, st.f[2], i, j, r), chain_val(a, st.f[3], i, j, r), st.l0, st.l1);
}

// Fused FilterExec of the probe child: PFS 0 = none, 1 = col <ID_EQ|ID_NEQ> literal, 2 = generic VM.
template <int PFS>
__device__ __forceinline__ bool lprobe_filter(const LdsJoinArgs& a, u64 j) {
  if constexpr (PFS == 0) return true;
  else if constexpr (PFS == 1) {
    const u32 v = a.cols[a.pid.col][j], lit = a.pid.lit;
    if (v == 0 || lit == 0) return false;
    return (v == lit) == (a.pid.is_eq != 0);
  } else {
    const Val r = eval_program(*a.probe_prog, a.tt, [&](u32 col) { return a.cols[a.probe_col_base + col][j]; });
    return r.lo == 1;
  }
}

// Static-indexed key handling (runtime n_keys <= 4 without private-memory arrays).
struct Keys { u32 k[RDFGPU_MAX_KEYS]; };
__device__ __forceinline__ u32 hash_keys4(const Keys& key, u32 n) {   // 32-bit multiply/xorshift mix: 64-bit multiplies are 4 quarter-rate ops each
  u32 h = 0x9E3779B9u;
#pragma unroll
  for (u32 i = 0; i < RDFGPU_MAX_KEYS; i++) if (i < n) { h = (h ^ key.k[i]) * 0x85EBCA6Bu; h ^= h >> 15; }
  h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ bool load_keys(const u32* const* key_cols, u32 n_keys, u64 row, Keys& key) {
  bool null_key = false;
#pragma unroll
  for (u32 q = 0; q < RDFGPU_MAX_KEYS; q++) {
    key.k[q] = 0;
    if (q < n_keys) { key.k[q] = key_cols[q][row]; null_key = null_key || key.k[q] == 0; }
  }
  return !null_key;   // NullEqualsNothing: a null key never matches
}



// Probe side: every wave alternates between two phases, with no workgroup barrier and no global atomic
// inside the probe loop:
//   fill    — lanes walk their chains comparing KEYS only; a ballot + mbcnt compacts the wave's key-equal
//             (build row, probe row) candidates into a wave-private LDS queue (a.wave_q entries);
//   resolve — when the queue is full (and once at the end) the wave takes the candidates back out, 64 x 4 at a
//             time with all lanes busy, evaluates the join filter on them (its column and typed-value gathers
//             are independent across the four, so they are in flight together instead of sitting inside a
//             divergent chain walk) and compacts the survivors in place; ONE atomicAdd reserves their output
//             range and the lanes write consecutive output rows.
// The last resolve is shared by the workgroup (one reservation for all eight queues).  Same-address atomics
// retire at only ~88 per microsecond on this chip, which is what sizes the queue: sparse joins (BSBM: a handful
// of matches per thousand probe rows) pay one atomic per workgroup, dense ones one per a.wave_q matches.
// CHAIN = true adds, between the fill and the reservation, one pass over the queue per fused follow-up join
// (ChainStage: direct-table lookup of a base key column + the stage's filter, survivors compacted in place, selective
// stages first, the base join's own filter last), and reads the output columns by (source row, pointer) — nothing
// between the base join and the last stage is materialised.  In range-index mode the fill expands, per probe row,
// only the value interval of its key's group that the first stage's window allows, and candidates carry index
// positions (see bcol).

// NK = 1: the join has ONE key column (every join of the BSBM plans, most others): key loads, the hash and the key compares are then
// straight-line code instead of loops over a run-time count with a wave-uniform branch per key — the kernel issues 2.5 scalar
// instructions per vector instruction (SQ counters, profiles/tools/join_micro_prof.sh), a good part of them exactly those branches;
// 11 - 25 % of its time on 32 M probe rows (profiles/tools/join_micro.py).  NK = 0: the key count is a.n_keys.
template <int FS, int PFS, int ITEMS, int MODE, bool CHAIN, int NK>
__device__ __forceinline__ void lds_join_body(const LdsJoinArgs& a) {
  const u32 n_keys = NK ? (u32)NK : a.n_keys;
  // candidates taken out of the queue per lane and resolve round: four keep more gathers in flight; the one-row-per-lane
  // variant takes two and fits 5 waves/SIMD without spills (95 VGPRs)
  constexpr int kResolveUnroll = ITEMS == 1 ? 2 : 4;
  constexpr bool GLOBAL = MODE != kJoinTableLds;      // the table lives in HBM / L2
  constexpr bool DIRECT = MODE == kJoinTableDirect;   // direct-address table: row = direct[key - direct_min]
  constexpr bool CSR = MODE == kJoinTableCsr;         // rows of key k: csr_rows[csr_off[k - min] .. csr_off[k - min + 1])
  extern __shared__ __align__(16) unsigned char lds_raw[];
  // dynamic LDS: [hash table (LDS variant only)] [8 wave queues]
  const uint2* slots = GLOBAL ? a.gslots : reinterpret_cast<const uint2*>(lds_raw);
  uint2* queues = reinterpret_cast<uint2*>(lds_raw) + (GLOBAL ? 0u : a.tbl_mask + 1u);
  __shared__ u32 wave_tot[kLdsBlock / 64];
  __shared__ u64 wg_base;
  // CSR mode may give every probe row 2^rl lanes (they take the row's matches round-robin), so that a small probe
  // side with a large fan-out still fills the chip; all other modes have rl = 0.
  const u32 rl = CSR ? a.row_lanes_log2 : 0u;
  const u32 kTileRows = (u32)(kLdsBlock * ITEMS) >> rl;
  const u32 tid = threadIdx.x;
  if constexpr (!GLOBAL) {
    uint2* lslots = reinterpret_cast<uint2*>(lds_raw);
    for (u32 s = tid; s <= a.tbl_mask; s += kLdsBlock) lslots[s] = make_uint2(0u, kNil);
    __syncthreads();
    const u64 nb = live_rows(a.n_build_dev, a.n_build_cap);
    for (u64 i = tid; i < nb; i += kLdsBlock) {
      Keys key;
      if (!load_keys(a.build_key, n_keys, i, key)) continue;
      u32 h = hash_keys4(key, n_keys) & a.tbl_mask;
      for (;;) {
        if (atomicCAS(&lslots[h].y, kNil, (u32)i) == kNil) { lslots[h].x = key.k[0]; break; }
        h = (h + 1) & a.tbl_mask;
      }
    }
    __syncthreads();
  }

  const u64 np = live_rows(a.n_probe_dev, a.n_probe_cap);
  const u64 n_tiles = (np + kTileRows - 1) / kTileRows;
  const u32 lane = tid & 63, wave = tid >> 6;
  const u32 qcap = a.wave_q;
  uint2* wq = queues + (size_t)wave * qcap;
  u32 qn = 0;   // candidates in this wave's queue (wave-uniform)

  auto write_out = [&](u64 base) {   // queue entries -> consecutive output rows base .. base + qn
    // four columns at a time: their pointers are fetched (scalar loads) once per call, and a lane has four
    // independent gathers in flight per entry instead of one load -> store chain per column
    for (u32 oc0 = 0; oc0 < a.n_out_cols; oc0 += 4) {
      const u32* src[4]; u32* dst[4]; u32 from[4]; bool on[4];   // from: 0 probe row, 1 build row, 2 + t chain stage t
#pragma unroll
      for (u32 u = 0; u < 4; u++) {
        on[u] = oc0 + u < a.n_out_cols;
        const u32 oc = on[u] ? oc0 + u : oc0;
        if constexpr (CHAIN) { src[u] = a.chain_out[oc].ptr; from[u] = a.chain_out[oc].src; }
        else {
          const u32 c = a.proj[oc];
          from[u] = (c < a.n_left_cols) == (a.build_is_left != 0) ? 1u : 0u;
          src[u] = a.cols[c];
        }
        dst[u] = a.out[oc];
      }
      for (u32 e = lane; e < qn; e += 64) {
        const uint2 m = wq[e];
        const u64 pos = base + e;
        if (pos >= a.out_cap) continue;
        u32 rr[kMaxChain] = {0, 0, 0};
        if constexpr (CHAIN) {   // survivors only: the stage rows are looked up again instead of being carried through the queue
#pragma unroll
          for (u32 t = 0; t < (u32)kMaxChain; t++) if (t < a.n_chain) rr[t] = chain_lookup(a, a.chain[t], m.x, m.y);
        }
        u32 v[4];
#pragma unroll
        for (u32 u = 0; u < 4; u++) if (on[u]) {
          const u32 row = from[u] == 0 ? m.y : from[u] == 1 ? m.x : from[u] == 2 ? rr[0] : from[u] == 3 ? rr[1] : rr[2];
          if (from[u] == 1 && m.x == kOuterNull) v[u] = 0u;     // a preserved probe row without a match: the build side is null
          else v[u] = from[u] == 1 ? bcol(a, src[u], row) : src[u][row];
        }
#pragma unroll
        for (u32 u = 0; u < 4; u++) if (on[u]) dst[u][pos] = v[u];
      }
    }
    if (a.visited) for (u32 e = lane; e < qn; e += 64) a.visited[wq[e].x] = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };

  bool row_live[ITEMS], had[ITEMS];   // probe_outer: the lane has a probe row / the row has produced a candidate (a match or its null row)
  Keys key[ITEMS]; u32 h[ITEMS]; uint2 s[ITEMS]; bool walking[ITEMS]; u32 pend[ITEMS];
  bool exact[ITEMS];     // range-index mode: the row's candidate range is exactly the first stage's window (integer operands)
  bool q_exact = true;   // wave-uniform: every candidate in the queue comes from such a row => the first stage pass can be skipped
  u64 tile = blockIdx.x;
  bool tile_loaded = false, exhausted = false;   // wave-uniform
  u32 nk0[ITEMS]; bool have_next = false;        // NK == 1: the next tile's keys, requested ahead
#pragma unroll
  for (int k = 0; k < ITEMS; k++) nk0[k] = 0;
  for (;;) {
    // ---- fill: walk tiles until the queue cannot take the next ballot's candidates or the tiles run out ----
    bool full = false;
    while (!full) {
      if (!tile_loaded) {
        if (tile >= n_tiles) { exhausted = true; break; }
        const u64 base = tile * kTileRows;
        // the tile's probe keys first (independent coalesced loads in flight together), then the first table slot of every row
        if constexpr (NK == 1) {
          // one key column: the NEXT tile's keys are requested as soon as this tile's are in hand and stay in flight while it is walked
          // (unconditional loads at a clamped row: a prefetch must not sit under a branch) — the waves of this kernel spent 81 % of
          // their cycles waiting, at four per SIMD, with every tile's loads heading its work
          const u32* pk0 = a.probe_key[0];
#pragma unroll
          for (int k = 0; k < ITEMS; k++) {
            const u64 j = base + (((u32)k * kLdsBlock + tid) >> rl);
            key[k].k[0] = have_next ? nk0[k] : pk0[j < np ? j : np - 1];
            key[k].k[1] = 0; key[k].k[2] = 0; key[k].k[3] = 0;
          }
          const u64 nt = tile + gridDim.x;
          have_next = nt < n_tiles;                        // wave-uniform
          if (have_next) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
              const u64 jn = nt * kTileRows + (((u32)k * kLdsBlock + tid) >> rl);
              nk0[k] = pk0[jn < np ? jn : np - 1];
            }
          }
#pragma unroll
          for (int k = 0; k < ITEMS; k++) {
            const u64 j = base + (((u32)k * kLdsBlock + tid) >> rl);
            walking[k] = j < np && key[k].k[0] != 0;       // NullEqualsNothing
          }
        } else {
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const u64 j = base + (((u32)k * kLdsBlock + tid) >> rl);
          walking[k] = j < np && load_keys(a.probe_key, n_keys, j, key[k]);
        }
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const u64 j = base + (((u32)k * kLdsBlock + tid) >> rl);
          row_live[k] = j < np && lprobe_filter<PFS>(a, j);
          had[k] = false;
          walking[k] = walking[k] && row_live[k];
          s[k] = make_uint2(0u, kNil);
          if constexpr (DIRECT) {   // unique dense keys: the one candidate is a single 4-byte load, no chain
            h[k] = key[k].k[0] - a.direct_min;
            if (walking[k] && h[k] < a.direct_n) s[k].y = a.direct[h[k]];
          } else if constexpr (CSR) {   // dense keys with duplicates: s = [cursor, end) into the key's row list
            h[k] = key[k].k[0] - a.direct_min;
            s[k] = make_uint2(0u, 0u);
            if (walking[k] && h[k] < a.direct_n) { s[k].x = a.csr_off[h[k]]; s[k].y = a.csr_off[h[k] + 1]; }
            if constexpr (CHAIN) {
              exact[k] = false;
              if (a.range_vals != nullptr && walking[k] && s[k].x < s[k].y) {
                // the group is ordered by the first stage's value: keep only [lower_bound(lo), upper_bound(hi)) of it,
                // lo / hi = the integer interval the stage's window allows for this probe row (a superset is enough:
                // the stage pass re-checks every candidate)
                const ChainStage& st0 = a.chain[0];
                const u32 iy0 = st0.f[1].ptr[j], iy1 = st0.f[3].ptr[j];
                long long lo = INT64_MIN + 1, hi = INT64_MAX;
                bool restrict_ok = a.tt.n_ids != 0 && iy0 != 0 && iy1 != 0 && iy0 < a.tt.n_ids && iy1 < a.tt.n_ids &&
                                   st0.l0.tag == RDFGPU_TV_INTEGER && st0.l1.tag == RDFGPU_TV_INTEGER;
                if (restrict_ok) {
                  const int4* tv = reinterpret_cast<const int4*>(a.tt.tv);
                  const int4 r0 = tv[iy0], r1 = tv[iy1];
                  restrict_ok = ((u32)r0.w & 0xff) == RDFGPU_TV_INTEGER && ((u32)r1.w & 0xff) == RDFGPU_TV_INTEGER;
                  auto i64 = [](const int4& r) { return (long long)(((u64)(u32)r.y << 32) | (u32)r.x); };
                  auto bound = [&](const TvLiteral& l, long long y) {
                    long long z;
                    const bool ovf = l.arith_sub ? __builtin_sub_overflow(y, (long long)l.lo, &z) : __builtin_add_overflow(y, (long long)l.lo, &z);
                    if (ovf) { restrict_ok = false; return; }              // error value: leave it to the stage pass
                    if (l.cmp_op == RDFGPU_EX_LT) { if (z == INT64_MIN) hi = INT64_MIN; else hi = z - 1 < hi ? z - 1 : hi; }
                    else if (l.cmp_op == RDFGPU_EX_LEQ) hi = z < hi ? z : hi;
                    else if (l.cmp_op == RDFGPU_EX_GT) { if (z == INT64_MAX) lo = INT64_MAX, hi = INT64_MIN; else lo = z + 1 > lo ? z + 1 : lo; }
                    else if (l.cmp_op == RDFGPU_EX_GEQ) lo = z > lo ? z : lo;
                    else restrict_ok = false;
                  };
                  if (restrict_ok) { bound(st0.l0, i64(r0)); bound(st0.l1, i64(r1)); }
                }
                if (restrict_ok) {
                  // the index stores value - vbase + 1 in 32 bits (0 = no stage row: below every lower bound)
                  const long long vb = a.range_vbase;
                  const unsigned long long dl = (unsigned long long)lo - (unsigned long long)vb;   // exact when lo >= vb (mod 2^64 otherwise unused)
                  const unsigned long long dh = (unsigned long long)hi - (unsigned long long)vb;   // exact when hi >= vb
                  const bool empty = hi < vb || lo > hi || (lo > vb && dl >= 0xFFFFFFFEull);
                  const u32 lo_b = lo <= vb ? 1u : (u32)dl + 1u;
                  const u32 hi_b = dh >= 0xFFFFFFFEull ? 0xFFFFFFFFu : (u32)dh + 1u;
                  u32 b = s[k].x, e = empty ? s[k].x : s[k].y;
                  while (b < e) { const u32 mid = b + ((e - b) >> 1); if (a.range_vals[mid] < lo_b) b = mid + 1; else e = mid; }   // lower_bound(lo)
                  const u32 first = b;
                  e = empty ? s[k].x : s[k].y;
                  while (b < e) { const u32 mid = b + ((e - b) >> 1); if (a.range_vals[mid] <= hi_b) b = mid + 1; else e = mid; }  // upper_bound(hi)
                  s[k].x = first; s[k].y = b;
                  exact[k] = true;
                }
              }
            }
            s[k].x += tid & ((1u << rl) - 1u);
          } else {
            h[k] = hash_keys4(key[k], n_keys) & a.tbl_mask;
            if (walking[k]) s[k] = slots[h[k]];
          }
          pend[k] = kNil;
        }
        tile_loaded = true;
      }
#pragma unroll
      for (int k = 0; k < ITEMS; k++) {
        const u32 j = (u32)(tile * kTileRows + (((u32)k * kLdsBlock + tid) >> rl));
        while (!full) {
          u32 hit = pend[k];   // a candidate that did not fit before the last resolve, else the lane's next key-equal build row
          pend[k] = kNil;
          if constexpr (DIRECT) {
            if (walking[k]) { hit = s[k].y; walking[k] = false; }
          } else if constexpr (CSR) {
            if (hit == kNil && walking[k]) {
              if (s[k].x < s[k].y) {
                if (CHAIN && a.range_link != nullptr) hit = s[k].x;   // the position itself (see bcol)
                else hit = a.csr_rows ? a.csr_rows[s[k].x] : s[k].x;
                s[k].x += 1u << rl;
              }
              else walking[k] = false;
            }
          } else if (hit == kNil) {
            // The walk keeps no state but the slot in hand (s[k] = slots[h[k]]): a lane stops at an empty slot (no further match) or at
            // a key-equal one; the loop body is one read, the compares and one exit.  (The `continue / break` form of this loop cost ~35
            // scalar instructions and 7 branches per hop: what the partitioned join was bound by, part_join.hip.)
            if (walking[k]) {
              for (;;) {
                const uint2 c = s[k];
                if (c.y == kNil) { walking[k] = false; break; }
                bool eq = c.x == key[k].k[0];
                if (eq) {
#pragma unroll
                  for (u32 q = 1; q < RDFGPU_MAX_KEYS; q++) if (q < n_keys) eq = eq && a.build_key[q][c.y] == key[k].k[q];
                }
                h[k] = (h[k] + 1) & a.tbl_mask;
                s[k] = slots[h[k]];   // the next slot: what the walk resumes from after a match, and its next hop otherwise
                if (eq) { hit = c.y; break; }
              }
            }
          }
          if (a.probe_outer) {                            // wave-uniform
            if (hit == kNil && !walking[k] && row_live[k] && !had[k]) hit = kOuterNull;   // the row's walk is over and nothing matched: its one null-extended row
            had[k] = had[k] || hit != kNil;
          }
          const unsigned long long found = __ballot(hit != kNil);
          if (found == 0) break;
          const u32 n_found = (u32)__popcll(found);
          if (qn + n_found > qcap) { pend[k] = hit; full = true; break; }
          if constexpr (CSR && CHAIN) q_exact = q_exact && !__any(hit != kNil && !exact[k]);
          if (hit != kNil) wq[qn + lane_prefix(found)] = make_uint2(hit, j);
          qn += n_found;
        }
      }
      if (!full) { tile += gridDim.x; tile_loaded = false; }
    }
    // ---- resolve: join filter over the queued candidates, survivors compacted in place ----
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    auto base_pass = [&]() {   // the base join's own filter (+ a former build-side FilterExec), survivors compacted in place
      u32 kept = 0;
      for (u32 g0 = 0; g0 < qn; g0 += 64 * kResolveUnroll) {
        uint2 m[kResolveUnroll]; bool ok[kResolveUnroll];
#pragma unroll
        for (int u = 0; u < kResolveUnroll; u++) {
          const u32 e = g0 + (u32)u * 64 + lane;
          ok[u] = e < qn; m[u] = make_uint2(0u, 0u);
          if (ok[u]) m[u] = wq[e];
        }
        bool slow[kResolveUnroll];
        if (a.has_post) {   // wave-uniform: the former build-side FilterExec (`col <=|!=> literal`)
          const bool post_from_build = (a.post.col < a.n_left_cols) == (a.build_is_left != 0);
          const u32* pc = a.cols[a.post.col];
#pragma unroll
          for (int u = 0; u < kResolveUnroll; u++) {
            const u32 v = post_from_build ? bcol(a, pc, ok[u] ? m[u].x : 0u) : pc[ok[u] ? m[u].y : 0u];
            ok[u] = ok[u] && v != 0 && a.post.lit != 0 && ((v == a.post.lit) == (a.post.is_eq != 0));
          }
        }
#pragma unroll
        for (int u = 0; u < kResolveUnroll; u++) {
          slow[u] = false;
          if (ok[u]) ok[u] = ljoin_filter_fast<FS>(a, m[u].x, m[u].y, slow[u]);
        }
        if constexpr (FS == 1 || FS == 3) {
          for (;;) {   // the undecided candidates, one per lane and round, through the single copy of the full semantics
            int pick = -1;
#pragma unroll
            for (int u = kResolveUnroll - 1; u >= 0; u--) pick = slow[u] ? u : pick;
            if (!__any(pick >= 0)) break;
            if (pick >= 0) {
              uint2 mm = m[0];
#pragma unroll
              for (int u = 1; u < kResolveUnroll; u++) mm = pick == u ? m[u] : mm;   // value selects keep m[] in registers
              const bool r = ljoin_filter_slow<FS>(a, mm.x, mm.y);
#pragma unroll
              for (int u = 0; u < kResolveUnroll; u++) { ok[u] = pick == u ? r : ok[u]; slow[u] = pick == u ? false : slow[u]; }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kResolveUnroll; u++) {   // every read of this round is done: writing below g0 + 256 is safe
          const unsigned long long mask = __ballot(ok[u]);
          if (ok[u]) wq[kept + lane_prefix(mask)] = m[u];
          kept += (u32)__popcll(mask);
        }
      }
      qn = kept;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    // Conjuncts commute: with a fused chain the (selective) stage filters run first and the base filter — in practice
    // a barely selective `!=` — only sees what they left; without a chain it is the only pass.
    if constexpr (!CHAIN) { if (FS != 0 || a.has_post) base_pass(); }
    if constexpr (CHAIN) {
      // One pass over the (shrinking) queue per fused stage, survivors compacted in place after each: a selective
      // stage (a numeric window keeps ~10 %) leaves the later stages a tenth of the candidates, packed into full waves.
      for (u32 t = 0; t < a.n_chain; t++) {
        const ChainStage& st = a.chain[t];
        // range-index mode: the candidates were cut out of their groups by exactly this stage's window — nothing to re-check
        if (t == 0 && a.range_link != nullptr && st.val != nullptr && q_exact) continue;
        u32 kept = 0;
        for (u32 g0 = 0; g0 < qn; g0 += 64 * kResolveUnroll) {
          uint2 m[kResolveUnroll]; bool ok[kResolveUnroll], slow[kResolveUnroll]; u32 r[kResolveUnroll];
#pragma unroll
          for (int u = 0; u < kResolveUnroll; u++) {
            const u32 e = g0 + (u32)u * 64 + lane;
            ok[u] = e < qn; m[u] = make_uint2(0u, 0u); slow[u] = false;
            if (ok[u]) m[u] = wq[e];
          }
          if (st.val != nullptr) {   // wave-uniform: integer window over the slice's decoded value table — one load by key
                                     // instead of lookup -> column gather -> typed-value gather
#pragma unroll
            for (int u = 0; u < kResolveUnroll; u++) {
              long long x; bool in;
              if (t == 0 && a.range_link != nullptr) {   // range-index mode: the candidate's position carries the first stage's value
                const u32 biased = a.range_vals[ok[u] ? m[u].x : 0u];
                in = ok[u] && biased != 0;
                x = in ? (long long)(biased - 1u) + a.range_vbase : INT64_MIN;
              } else {
                const u32 key = st.key.src ? bcol(a, st.key.ptr, ok[u] ? m[u].x : 0u) : st.key.ptr[ok[u] ? m[u].y : 0u];
                const u32 d = key - st.kmin;
                in = ok[u] && key != 0 && d < st.kn;
                x = st.val[in ? d : 0u];
              }
              const bool have = in && x != INT64_MIN;
              const u32 iy0 = chain_val(a, st.f[1], have ? m[u].x : 0u, have ? m[u].y : 0u, 0u);   // y operands are base columns here
              const u32 iy1 = chain_val(a, st.f[3], have ? m[u].x : 0u, have ? m[u].y : 0u, 0u);
              bool und;
              const bool pass = window_fast_x(a.tt, x, iy0, iy1, st.l0, st.l1, und);
              slow[u] = have && und;
              ok[u] = have && !und && pass;
              r[u] = kNil;
            }
            bool any_slow = false;
#pragma unroll
            for (int u = 0; u < kResolveUnroll; u++) any_slow = any_slow || slow[u];
            if (__any(any_slow)) {   // rare: a y operand that is not an xsd:integer — look the row up after all
#pragma unroll
              for (int u = 0; u < kResolveUnroll; u++) if (slow[u]) r[u] = chain_lookup(a, st, m[u].x, m[u].y, true);
            }
          } else {
#pragma unroll
          for (int u = 0; u < kResolveUnroll; u++) r[u] = chain_lookup(a, st, m[u].x, m[u].y, ok[u]);   // branch-free: dead lanes read row 0
#pragma unroll
          for (int u = 0; u < kResolveUnroll; u++) ok[u] = ok[u] && r[u] != kNil;
          }
          if (st.fs != 0 && st.val == nullptr) {   // wave-uniform
#pragma unroll
            for (int u = 0; u < kResolveUnroll; u++) {
              bool und;
              const u32 ci = ok[u] ? m[u].x : 0u, cj = ok[u] ? m[u].y : 0u, cr = ok[u] ? r[u] : 0u;
              const bool pass = stage_filter_fast(a, st, ci, cj, cr, und);
              slow[u] = ok[u] && und;
              ok[u] = ok[u] && !und && pass;
            }
          }
          if (st.fs != 0) {
            for (;;) {   // undecided candidates: this stage's filter with the full semantics, one per lane and round
              int pick = -1;
#pragma unroll
              for (int u = kResolveUnroll - 1; u >= 0; u--) pick = slow[u] ? u : pick;
              if (!__any(pick >= 0)) break;
              if (pick >= 0) {
                uint2 mm = m[0]; u32 rr = r[0];
#pragma unroll
                for (int u = 1; u < kResolveUnroll; u++) { mm = pick == u ? m[u] : mm; rr = pick == u ? r[u] : rr; }
                const bool res = stage_filter_slow(a, st, mm.x, mm.y, rr);
#pragma unroll
                for (int u = 0; u < kResolveUnroll; u++) { ok[u] = pick == u ? res : ok[u]; slow[u] = pick == u ? false : slow[u]; }
              }
            }
          }
#pragma unroll
          for (int u = 0; u < kResolveUnroll; u++) {
            const unsigned long long mask = __ballot(ok[u]);
            if (ok[u]) wq[kept + lane_prefix(mask)] = m[u];
            kept += (u32)__popcll(mask);
          }
        }
        qn = kept;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
    if constexpr (CHAIN) { if (FS != 0 || a.has_post) base_pass(); }
    if (exhausted) break;
    unsigned long long b = 0;
    if (lane == 0 && qn) {
      b = atomicAdd((unsigned long long*)a.n_out_dev, (unsigned long long)qn);
      if (b + qn > a.out_cap) *a.overflow = 1u;
    }
    b = __shfl(b, 0, 64);
    write_out(b);
    qn = 0;
    q_exact = true;
  }

  // what is still queued leaves with one reservation for the whole workgroup
  if (lane == 0) wave_tot[wave] = qn;
  __syncthreads();
  if (tid == 0) {
    u32 t = 0;
    for (int w = 0; w < kLdsBlock / 64; w++) t += wave_tot[w];
    u64 b = 0;
    if (t) {
      b = atomicAdd((unsigned long long*)a.n_out_dev, (unsigned long long)t);
      if (b + t > a.out_cap) *a.overflow = 1u;
    }
    wg_base = b;
  }
  __syncthreads();
  u64 out_base = wg_base;
  for (u32 w = 0; w < wave; w++) out_base += wave_tot[w];
  write_out(out_base);
}

// (two kernels, not one kernel with a branch: a kernel's registers and occupancy are those of its larger body)
template <int FS, int PFS, int ITEMS, int MODE, bool CHAIN, int NK>
__global__ __launch_bounds__(kLdsBlock) __attribute__((amdgpu_waves_per_eu(ITEMS == 1 ? 5 : 4))) void lds_join_kernel(const LdsJoinArgs a) {
  lds_join_body<FS, PFS, ITEMS, MODE, CHAIN, NK>(a);
}

template <int FS, int PFS, int ITEMS, int MODE, bool CHAIN, int NK>
static void launch_lds_join_tk(const LdsJoinArgs& a, dim3 g, size_t lds, hipStream_t s) {
  static std::once_flag attr_once;   // dynamic LDS above 64 KiB has to be opted into, once per kernel instance (plans run on many host threads)
  std::call_once(attr_once, [] {
    RDFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_join_kernel<FS, PFS, ITEMS, MODE, CHAIN, NK>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  });
  hipLaunchKernelGGL((lds_join_kernel<FS, PFS, ITEMS, MODE, CHAIN, NK>), g, dim3(kLdsBlock), lds, s, a);
}
template <int FS, int PFS, int ITEMS, int MODE, bool CHAIN>
static void launch_lds_join_tc(const LdsJoinArgs& a, dim3 g, size_t lds, hipStream_t s) {
  if (a.n_keys == 1) launch_lds_join_tk<FS, PFS, ITEMS, MODE, CHAIN, 1>(a, g, lds, s);
  else launch_lds_join_tk<FS, PFS, ITEMS, MODE, CHAIN, 0>(a, g, lds, s);
}
template <int FS, int PFS, int ITEMS, int MODE>
static void launch_lds_join_t(const LdsJoinArgs& a, dim3 g, size_t lds, hipStream_t s) {
  // the fused lookup chain exists for HBM-table joins without a VM filter or a fused probe-side FilterExec
  if constexpr (FS != 1 && PFS == 0 && MODE != kJoinTableLds) { if (a.n_chain) return launch_lds_join_tc<FS, PFS, ITEMS, MODE, true>(a, g, lds, s); }
  if (a.n_chain) fail(RDFGPU_ERR_INVALID, "lds join: lookup chain on an unsupported join shape");
  launch_lds_join_tc<FS, PFS, ITEMS, MODE, false>(a, g, lds, s);
}
// One definition per join-filter shape FS (join_fs0.hip .. join_fs3.hip): dispatch over PFS / ITEMS / MODE / chain.
template <int FS> void launch_lds_join_fs(const LdsJoinArgs& a, int pfs, int items, int mode, dim3 g, size_t lds, hipStream_t s);

#define RDFGPU_DEFINE_JOIN_FS(F)                                                                                                   \
  template <> void launch_lds_join_fs<F>(const LdsJoinArgs& a, int pfs, int items, int mode, dim3 g, size_t lds, hipStream_t s) {  \
    RDFGPU_JOIN_P(F, 0) RDFGPU_JOIN_P(F, 1) RDFGPU_JOIN_P(F, 2)                                                                     \
    fail(RDFGPU_ERR_INVALID, "lds join: bad probe filter shape %d", pfs);                                                           \
  }
#define RDFGPU_JOIN_I(F, P, M) { if (items == 4) return launch_lds_join_t<F, P, 4, M>(a, g, lds, s); return launch_lds_join_t<F, P, 1, M>(a, g, lds, s); }
#define RDFGPU_JOIN_P(F, P) if (pfs == P) { if (mode == kJoinTableCsr) RDFGPU_JOIN_I(F, P, kJoinTableCsr) else if (mode == kJoinTableDirect) RDFGPU_JOIN_I(F, P, kJoinTableDirect) else if (mode == kJoinTableHash) RDFGPU_JOIN_I(F, P, kJoinTableHash) else RDFGPU_JOIN_I(F, P, kJoinTableLds) }

}  // namespace rdfgpu
