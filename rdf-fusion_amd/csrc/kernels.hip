// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the scan / filter / join path.
//
// All of this is HBM-bound integer work (no MFMA): the rules that matter are coalesced streaming
// access, wave64 ballot + mbcnt prefix sums for the variable-cardinality outputs, one atomic per
// workgroup (never per lane / per wave) when reserving output space, and grids of >> 256
// workgroups so all 8 XCDs stay busy.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "kernels.hpp"
#include "join_device.hpp"

namespace rdfgpu {



// --------------------------------------------------------------------------------------------------
// K1 range locate: successive binary searches narrow [lo, hi) on the leading levels.
// Behavioural twin of MemIndexData::prune_relevant_row_groups (quad_index_data.rs:155-284) on a flat
// sorted column (the reference walks 8192-row groups linearly; here it is O(log n) per level).
// --------------------------------------------------------------------------------------------------
// One wave per job; each search step tests 64 pivots at once (a 65-ary search: log65(1e8) ~ 4.4 dependent
// HBM round trips per bound instead of 27 for a scalar binary search).
// wave_partition_point returns the first index in [lo, hi) whose value is > key (UPPER) or >= key (!UPPER).
template <bool UPPER>
__device__ __forceinline__ u64 wave_partition_point(const u32* c, u64 lo, u64 hi, u32 key) {
  const u32 lane = threadIdx.x & 63;
  while (hi - lo > 64) {
    const u64 step = (hi - lo + 64) / 65;          // 64 pivots split [lo, hi) into 65 pieces
    const u64 idx = lo + (u64)(lane + 1) * step - 1;
    bool below = false;                             // "pivot is left of the partition point"
    if (idx < hi) { const u32 v = c[idx]; below = UPPER ? v <= key : v < key; }
    const unsigned long long m = __ballot(below);
    const u32 cnt = (u32)__popcll(m);               // pivots are sorted: the below-lanes are a prefix
    const u64 nlo = cnt ? lo + (u64)cnt * step : lo;
    const u64 nhi = cnt < 64 ? lo + (u64)(cnt + 1) * step - 1 : hi;   // pivot cnt (0-based) is >= point
    lo = nlo; hi = nhi < hi ? nhi : hi;
  }
  bool below = false;
  if (lo + lane < hi) { const u32 v = c[lo + lane]; below = UPPER ? v <= key : v < key; }
  return lo + (u64)__popcll(__ballot(below));
}

__global__ __launch_bounds__(64) void locate_kernel(const LocateJob* jobs, u32 n_jobs, u64* lo_hi) {
  const u32 j = blockIdx.x;
  if (j >= n_jobs) return;
  const LocateJob job = jobs[j];
  u64 lo = 0, hi = job.n;
  u32 sorted_level = 0;                // the first level not pinned to one id: sorted within [lo, hi)
  for (u32 k = 0; k < job.n_levels && lo < hi; k++) {
    const u32* c = job.col[k];
    const u32 from = job.from[k], to = job.to[k];
    const u64 nlo = wave_partition_point<false>(c, lo, hi, from);   // lower_bound(from)
    const u64 nhi = wave_partition_point<true>(c, nlo, hi, to);     // upper_bound(to)
    lo = nlo; hi = nhi;
    if (from != to) break;            // below a proper range the inner levels are not contiguous (:240)
    sorted_level = k + 1;
  }
  if (lo > hi) lo = hi;
  if (threadIdx.x == 0) {
    lo_hi[kLocateWords * j] = lo; lo_hi[kLocateWords * j + 1] = hi;
    const bool any = lo < hi && sorted_level < 4;
    lo_hi[kLocateWords * j + 2] = ((u64)(any ? sorted_level : 4u) << 32) | (any ? job.col[sorted_level][lo] : 0u);
    lo_hi[kLocateWords * j + 3] = any ? job.col[sorted_level][hi - 1] : 0u;
  }
}
void launch_locate(const LocateJob* jobs_dev, u32 n_jobs, u64* lo_hi_dev, hipStream_t s) {
  if (!n_jobs) return;
  hipLaunchKernelGGL(locate_kernel, dim3(n_jobs), dim3(64), 0, s, jobs_dev, n_jobs, lo_hi_dev);
}

// --------------------------------------------------------------------------------------------------
// K2 ordered scan with residual predicates (compute_selection_vector / apply_predicate scan.rs:264-340):
// pass 1 counts matches per 1024-row tile, a device scan turns counts into offsets, pass 2 re-evaluates
// and writes in index order (so the output stays sorted like the reference's).
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool scan_match(const ScanJob& j, u64 row) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const ScanLevelPred& p = j.pred[k];
    if (p.kind == RDFGPU_PRED_NONE) continue;
    if (p.kind == RDFGPU_PRED_FALSE) { ok = false; continue; }
    const u32 v = j.col[k][row];
    if (p.kind == RDFGPU_PRED_BETWEEN) ok = ok && (v >= p.a && v <= p.b);
    else if (p.kind == RDFGPU_PRED_IN) { bool any = false; for (u32 q = 0; q < p.b; q++) any = any || (p.ids[q] == v); ok = ok && any; }
    else ok = ok && (j.col[p.a][row] == v);  // EQUAL_TO another level
  }
  return ok;
}

__global__ __launch_bounds__(kBlock) void scan_count_kernel(const ScanJob job, u32* block_counts) {
  __shared__ u32 wave_tot[kBlock / 64];
  const u64 base = (u64)blockIdx.x * kTile;
  u32 tot = 0;
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    const u64 row = base + (u64)k * kBlock + threadIdx.x;
    const bool m = row < job.n && scan_match(job, row);
    tot += (u32)__popcll(__ballot(m));
  }
  if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

__global__ __launch_bounds__(kBlock) void scan_write_kernel(const ScanJob job, const u32* block_offsets) {
  __shared__ u32 cnt[kItems][kBlock / 64];
  const u64 base = (u64)blockIdx.x * kTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool keep[kItems]; u32 pre[kItems];
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    const u64 row = base + (u64)k * kBlock + threadIdx.x;
    keep[k] = row < job.n && scan_match(job, row);
    const unsigned long long mask = __ballot(keep[k]);
    pre[k] = lane_prefix(mask);
    if (lane == 0) cnt[k][wave] = (u32)__popcll(mask);
  }
  __syncthreads();
  u32 off = block_offsets[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    u32 before = 0;
    for (int w = 0; w < wave; w++) before += cnt[k][w];
    if (keep[k]) {
      const u64 row = base + (u64)k * kBlock + threadIdx.x;
      for (u32 c = 0; c < job.n_out; c++) job.out[c][off + before + pre[k]] = job.col[job.out_level[c]][row];
    }
    off += cnt[k][0] + cnt[k][1] + cnt[k][2] + cnt[k][3];
  }
}
void launch_scan_count(const ScanJob& job, u32* block_counts, hipStream_t s) {
  hipLaunchKernelGGL(scan_count_kernel, grid_for(job.n), dim3(kBlock), 0, s, job, block_counts);
}
void launch_scan_write(const ScanJob& job, const u32* block_offsets, hipStream_t s) {
  hipLaunchKernelGGL(scan_write_kernel, grid_for(job.n), dim3(kBlock), 0, s, job, block_offsets);
}

// --------------------------------------------------------------------------------------------------
// K3 FilterExec: predicate + projection + compaction in ONE pass over the input.
// Each lane tests 4 rows (coalesced 256-B wave loads), wave ballots give the lane prefix, the four
// waves meet once in LDS and ONE atomicAdd per 1024-row workgroup reserves the output range
// (FILTER keeps a row iff its EBV is true: logical_plan_builder.rs:114-129).
// --------------------------------------------------------------------------------------------------
// Predicate on a value already in a register (the specialised shapes read exactly one column).
template <int SHAPE>
__device__ __forceinline__ bool filter_pred_value(const FilterArgs& a, u32 v) {
  if constexpr (SHAPE == 3) {  // string predicate, answered per distinct term beforehand: one byte gather (L2-resident table)
    if (v == 0 || v >= a.n_verdict) return false;
    const unsigned char verdict = a.verdict[v];
    if (verdict == 3 && a.tt.rt_error) atomicOr(a.tt.rt_error, 1u);   // this row's string needs the Unicode tables: fail loudly
    return verdict == 1;
  } else if constexpr (SHAPE == 1) {  // col <ID_EQ|ID_NEQ> object-id literal; null on either side => dropped
    const u32 lit = a.prog.nodes[1].u;
    if (v == 0 || lit == 0) return false;
    return (v == lit) == (a.prog.nodes[2].op == RDFGPU_EX_ID_EQ);
  } else {                     // EBV(cmp(ENC_TV(col), typed literal)) — the BSBM Q1 numeric FILTER
    const Val x = enc_tv(a.tt, v);
    const rdfgpu_expr_node& l = a.prog.nodes[2];
    if (x.tag == RDFGPU_TV_INTEGER && l.tag == RDFGPU_TV_INTEGER) {   // xsd:integer vs xsd:integer: plain i64 compare
      const u8 op = a.prog.nodes[3].op;
      const int64_t p = x.lo, q = l.lo;
      return op == RDFGPU_EX_GT ? p > q : op == RDFGPU_EX_LT ? p < q : op == RDFGPU_EX_GEQ ? p >= q
           : op == RDFGPU_EX_LEQ ? p <= q : op == RDFGPU_EX_EQ ? p == q : p != q;
    }
    Val y = val_tv_null(); y.tag = l.tag; y.flags = l.flags; y.aux = l.u; y.lo = l.lo; y.hi = l.hi;
    const int o = tv_partial_cmp(x, y);
    if (o == ORD_NONE) return false;
    const u8 op = a.prog.nodes[3].op;
    return op == RDFGPU_EX_GT ? o > 0 : op == RDFGPU_EX_LT ? o < 0 : op == RDFGPU_EX_GEQ ? o >= 0
         : op == RDFGPU_EX_LEQ ? o <= 0 : op == RDFGPU_EX_EQ ? o == 0 : o != 0;
  }
}

// Workgroup = 256 lanes x `iters` rounds x 4 consecutive rows per lane (one 16-byte load per column and
// round when the columns are 16-B aligned modulo a common head skip `mis`).  Pass 1 keeps each lane's
// verdicts in a 64-bit register mask, so nothing is read or evaluated twice; ONE atomicAdd per workgroup
// (<= 16 K rows) reserves the output range — same-address atomics retire at only ~88 per microsecond on
// this chip, so per-1024-row reservations would cap a 5.5 M-row filter at ~60 us; pass 2 turns the masks
// into wave ballots and writes each round's survivors contiguously.
constexpr int kFilterMaxIters = 16;
template <int SHAPE>
__global__ __launch_bounds__(kBlock) void filter_kernel(const FilterArgs a) {
  __shared__ u32 wave_tot[kBlock / 64];
  __shared__ u64 block_base;
  const u64 n = live_rows(a.n_in_dev, a.n_in_cap);
  const u64 mis = a.head_skip;                      // virtual row v maps to real row v - mis
  const u64 nv = n + mis;
  const u64 chunk = (u64)kTile * a.iters;
  const u64 base = (u64)blockIdx.x * chunk;
  if (base >= nv) return;                           // uniform per workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32* pcol = (SHAPE != 0) ? a.in[a.prog.nodes[0].u] : nullptr;
  unsigned long long bits = 0;
#pragma unroll 2
  for (u32 it = 0; it < a.iters; it++) {
    const u64 v0 = base + (u64)it * kTile + (u64)threadIdx.x * 4;
    if (v0 >= nv) break;
    u32 val[4] = {0, 0, 0, 0};
    if constexpr (SHAPE != 0) {
      if (a.vec_ok && v0 >= mis && v0 + 4 <= nv) {  // aligned 16-byte load, entirely inside the column
        const uint4 q = *reinterpret_cast<const uint4*>(pcol + (v0 - mis));
        val[0] = q.x; val[1] = q.y; val[2] = q.z; val[3] = q.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) if (v0 + r >= mis && v0 + r < nv) val[r] = pcol[v0 + r - mis];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const u64 v = v0 + r;
      bool keep = v >= mis && v < nv;
      if (keep) {
        if constexpr (SHAPE != 0) keep = filter_pred_value<SHAPE>(a, val[r]);
        else if (a.prog.n) { const u64 row = v - mis; const Val res = eval_program(a.prog, a.tt, [&](u32 c) { return a.in[c][row]; }); keep = res.lo == 1; }
      }
      bits |= (unsigned long long)keep << (it * 4 + r);
    }
  }
  const u32 mine = (u32)__popcll(bits);
  u32 wsum = mine;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wsum += __shfl_xor(wsum, d, 64);
  if (lane == 0) wave_tot[wave] = wsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 t = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    block_base = t ? atomicAdd((unsigned long long*)a.n_out_dev, (unsigned long long)t) : 0ull;
  }
  __syncthreads();
  u64 off = block_base;
  for (int w = 0; w < wave; w++) off += wave_tot[w];
  if (wsum == 0) return;                            // wave-uniform
  for (u32 it = 0; it < a.iters; it++) {
    const u64 v0 = base + (u64)it * kTile + (u64)threadIdx.x * 4;
    const u32 nib = (u32)(bits >> (it * 4)) & 15u;
    if (__ballot(nib != 0) == 0) continue;          // wave-uniform: nothing survived this round
    u64 pos[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const bool keep = (nib >> r) & 1u;
      const unsigned long long m = __ballot(keep);
      pos[r] = off + lane_prefix(m);
      off += (u32)__popcll(m);
    }
    if (nib == 0) continue;
    for (u32 c = 0; c < a.n_out_cols; c++) {
      const u32* src = a.in[a.proj[c]];
      u32 val[4];
      if (a.vec_proj_ok && v0 >= mis && v0 + 4 <= nv) {   // one aligned 16-byte load per column and round
        const uint4 q = *reinterpret_cast<const uint4*>(src + (v0 - mis));
        val[0] = q.x; val[1] = q.y; val[2] = q.z; val[3] = q.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) val[r] = ((nib >> r) & 1u) ? src[v0 + r - mis] : 0u;
      }
      u32* dst = a.out[c];
#pragma unroll
      for (int r = 0; r < 4; r++) if ((nib >> r) & 1u) dst[pos[r]] = val[r];
    }
  }
}
// REGEX / CONTAINS / STRSTARTS / STRENDS per DISTINCT TERM: one lane per object id walks its string once (consecutive
// ids = consecutive heap ranges, so the dictionary streams through), instead of once per row of every query — a 50 M-row
// filter over 4 M distinct strings then gathers one byte per row from a 4 MB table.
__global__ __launch_bounds__(256) void regex_verdict_kernel(const RegexProg* prog, const TypedTable tt, int64_t rhs_lang, unsigned char* out, u64 n_ids) {
  const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_ids) return;
  const Val v = enc_tv(tt, (u32)id);
  const Val r = tv_regex(*prog, tt, v, rhs_lang);
  out[id] = r.tag == RDFGPU_TV_BOOLEAN ? (unsigned char)(r.lo != 0) : r.aux == kRegexNeedsUnicode ? (unsigned char)3 : (unsigned char)2;   // 3: needs the Unicode tables
}
// ENC_PT (object_id_mapping.rs:331-374).  Pass 1: per row the term type, the typed value's tag / aux and the length of the
// lexical form; pass 2 (after a scan of the lengths): a wave copies the forms of 16 consecutive rows, 64 bytes per step.
__global__ __launch_bounds__(256) void decode_lengths_kernel(const DecodeArgs a) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const u32 id = a.ids[i];
  const bool known = id != 0 && id < a.tt.n_ids;
  unsigned char tt = 0xFF, tag = 0; u32 aux = 0, len = 0;
  if (known) {
    const rdfgpu_typed_value v = a.tt.tv[id];
    tag = v.tag; aux = v.aux;
    tt = v.tag == RDFGPU_TV_NAMED_NODE ? 0 : v.tag == RDFGPU_TV_BLANK_NODE ? 1 : 2;   // PlainTermType
    if (v.tag == RDFGPU_TV_NULL) tt = 0xFF;
    if (tt != 0xFF && a.tt.str_off && id < a.tt.n_str_ids) len = (u32)(a.tt.str_off[id + 1] - a.tt.str_off[id]);
  }
  a.term_type[i] = tt; a.tag[i] = tag; a.aux[i] = aux; a.len[i] = len;
}
__global__ __launch_bounds__(256) void decode_bytes_kernel(const DecodeArgs a) {
  const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  for (u32 r = 0; r < 16; r++) {
    const u64 i = wave * 16 + r;
    if (i >= a.n) return;                                  // wave-uniform
    const u32 n = a.len[i];
    if (n == 0) continue;
    const unsigned char* src = a.tt.heap + a.tt.str_off[a.ids[i]];
    unsigned char* dst = a.bytes + a.off[i];
    for (u32 b = lane; b < n; b += 64) dst[b] = src[b];
  }
}
void launch_decode_lengths(const DecodeArgs& a, hipStream_t s) {
  if (a.n) hipLaunchKernelGGL(decode_lengths_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
}
void launch_decode_bytes(const DecodeArgs& a, hipStream_t s) {
  const u64 waves = (a.n + 15) / 16;
  if (a.n) hipLaunchKernelGGL(decode_bytes_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a);
}
void launch_regex_verdicts(const RegexProg* prog_dev, const TypedTable& tt, int64_t rhs_lang, unsigned char* out, u64 n_ids, hipStream_t s) {
  if (!n_ids) return;
  hipLaunchKernelGGL(regex_verdict_kernel, dim3((unsigned)((n_ids + 255) / 256)), dim3(256), 0, s, prog_dev, tt, rhs_lang, out, n_ids);
}

// Streaming form of the specialised FilterExec shapes (1: `col <=|!=> id`, 2: `EBV(cmp(ENC_TV(col), literal))`, 3: per-term
// verdict byte) for at most two output columns whose pointers share the predicate column's 16-byte phase — a slice of
// one permutation, the BGP-scan case.  Two passes, NO atomics: a same-address returning atomic per 4096-row tile retires
// at ~88 per microsecond chip-wide, so the 16 384 tiles of a 2^26-row scan would spend 186 us on output reservations alone
// (measured: a single-pass kernel with one reservation per tile took 230 us whatever its body did).
//   pass 1  one verdict bit per row (2 bytes per 16 rows) + the tile's survivor count; reads the predicate column only
//   scan    exclusive scan of the tile counts (rocPRIM, 16 K elements)
//   pass 2  reads the bits and the output columns, writes the survivors at the tile's offset IN INDEX ORDER (the output
//           keeps the order of the scanned permutation, like the reference's filter over sorted row groups)
// One tile = 256 lanes x 4 rounds x 4 consecutive rows (one 16-byte load per column and round), all loads of a tile
// requested before anything is waited for.  The argument block is ~230 bytes (the generic kernel's carries a 40-node
// expression program by value: 73 spilled SGPRs).
struct FilterStreamArgs {
  const u32* pcol; const u32* proj[2]; u32* out[2];
  u32 n_out_cols, head_skip;
  const u64* n_in_dev; u64 n_in_cap; u64* n_out_dev;
  TypedTable tt;
  rdfgpu_expr_node lit; u32 op, id_lit, is_eq, pad;
  const unsigned char* verdict; u64 n_verdict;
  unsigned short* bits; u32* tile_count; u32* tile_off;   // pass 1 -> scan -> pass 2
};
constexpr int kStreamRounds = 4;
template <int SHAPE>
__device__ __forceinline__ bool stream_pred(const FilterStreamArgs& a, u32 v) {
  if constexpr (SHAPE == 3) {
    if (v == 0 || v >= a.n_verdict) return false;
    const unsigned char verdict = a.verdict[v];
    if (verdict == 3 && a.tt.rt_error) atomicOr(a.tt.rt_error, 1u);
    return verdict == 1;
  }
  else if constexpr (SHAPE == 4) {   // the comparison was answered once per id of the sorted slice's id range: one bit per id
    const u32 d = v - a.id_lit;
    if ((u64)d >= a.n_verdict) return false;
    return (reinterpret_cast<const u32*>(a.verdict)[d >> 5] >> (d & 31u)) & 1u;
  }
  else if constexpr (SHAPE == 1) { if (v == 0 || a.id_lit == 0) return false; return (v == a.id_lit) == (a.is_eq != 0); }
  else {
    const Val x = enc_tv(a.tt, v);
    const u8 op = (u8)a.op;
    if (x.tag == RDFGPU_TV_INTEGER && a.lit.tag == RDFGPU_TV_INTEGER) {   // xsd:integer vs xsd:integer: plain i64 compare
      const int64_t p = x.lo, q = a.lit.lo;
      return op == RDFGPU_EX_GT ? p > q : op == RDFGPU_EX_LT ? p < q : op == RDFGPU_EX_GEQ ? p >= q
           : op == RDFGPU_EX_LEQ ? p <= q : op == RDFGPU_EX_EQ ? p == q : p != q;
    }
    Val y = val_tv_null(); y.tag = a.lit.tag; y.flags = a.lit.flags; y.aux = a.lit.u; y.lo = a.lit.lo; y.hi = a.lit.hi;
    const int o = tv_partial_cmp(x, y);
    if (o == ORD_NONE) return false;
    return op == RDFGPU_EX_GT ? o > 0 : op == RDFGPU_EX_LT ? o < 0 : op == RDFGPU_EX_GEQ ? o >= 0
         : op == RDFGPU_EX_LEQ ? o <= 0 : op == RDFGPU_EX_EQ ? o == 0 : o != 0;
  }
}
// the 16-byte group of rows v0 .. v0 + 3 (virtual numbering: row v lives at col[v - mis]), zero outside [mis, nv)
__device__ __forceinline__ uint4 stream_load4(const u32* col, u64 v0, u64 mis, u64 nv) {
  if (v0 >= mis && v0 + 4 <= nv) return *reinterpret_cast<const uint4*>(col + (v0 - mis));
  u32 t[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int r = 0; r < 4; r++) if (v0 + r >= mis && v0 + r < nv) t[r] = col[v0 + r - mis];
  return make_uint4(t[0], t[1], t[2], t[3]);
}
template <int SHAPE>
__global__ __launch_bounds__(kBlock) void filter_bits_kernel(const FilterStreamArgs a) {
  __shared__ u32 wave_tot[kBlock / 64];
  const u64 n = live_rows(a.n_in_dev, a.n_in_cap);
  const u64 mis = a.head_skip, nv = n + mis;
  const u64 base = (u64)blockIdx.x * (kTile * kStreamRounds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 pv[kStreamRounds];
#pragma unroll
  for (int it = 0; it < kStreamRounds; it++) pv[it] = stream_load4(a.pcol, base + (u64)it * kTile + (u64)threadIdx.x * 4, mis, nv);
  u32 bits = 0;
  if constexpr (SHAPE == 2) {
    // xsd:integer value against an xsd:integer literal: one 16-byte gather, a tag test and an i64 compare per row, all 16
    // gathers of the lane in flight together.  Any other kind is left "undecided" and goes, one row at a time, through
    // the single copy of the full typed-value comparison below (promotion, decimals, strings, errors).
    const bool lit_int = a.lit.tag == RDFGPU_TV_INTEGER;
    const long long q = a.lit.lo;
    const u8 op = (u8)a.op;
    const u32 truth = op == RDFGPU_EX_GT ? 4u : op == RDFGPU_EX_LT ? 1u : op == RDFGPU_EX_GEQ ? 6u : op == RDFGPU_EX_LEQ ? 3u : op == RDFGPU_EX_EQ ? 2u : 5u;   // (less, equal, greater)
    const int4* tv = reinterpret_cast<const int4*>(a.tt.tv);
    const u64 n_ids = a.tt.n_ids;
    int4 raw[kStreamRounds * 4];
    u32 und = 0;
#pragma unroll
    for (int it = 0; it < kStreamRounds; it++) {
      const u32 val[4] = {pv[it].x, pv[it].y, pv[it].z, pv[it].w};
#pragma unroll
      for (int r = 0; r < 4; r++) raw[it * 4 + r] = tv[val[r] < n_ids ? val[r] : 0u];   // id 0 = null: tag 0
    }
#pragma unroll
    for (int it = 0; it < kStreamRounds; it++) {
      const u64 v0 = base + (u64)it * kTile + (u64)threadIdx.x * 4;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int4 x = raw[it * 4 + r];
        const bool live = v0 + r >= mis && v0 + r < nv;
        const u32 tag = (u32)x.w & 0xffu;
        const long long p = (long long)(((u64)(u32)x.y << 32) | (u32)x.x);
        const u32 c = p < q ? 1u : p > q ? 4u : 2u;
        const bool fast = lit_int && tag == RDFGPU_TV_INTEGER;
        bits |= (u32)(live && fast && (truth & c) != 0) << (it * 4 + r);
        und |= (u32)(live && !fast && tag != RDFGPU_TV_NULL) << (it * 4 + r);          // null / unknown id: the error value, dropped
      }
    }
    if (__any(und != 0)) {
      for (int k = 0; k < kStreamRounds * 4; k++) {
        if (!((und >> k) & 1u)) continue;
        const uint4 g = pv[0];
        (void)g;
        u32 v = 0;
#pragma unroll
        for (int it = 0; it < kStreamRounds; it++) {
          const u32 val[4] = {pv[it].x, pv[it].y, pv[it].z, pv[it].w};
#pragma unroll
          for (int r = 0; r < 4; r++) v = (it * 4 + r) == k ? val[r] : v;
        }
        bits |= (u32)stream_pred<2>(a, v) << k;
      }
    }
  } else if constexpr (SHAPE == 4) {
    // One verdict bit per id of the slice's id range.  The column is the slice's SORTED column: the four ids of a 16-byte group
    // are neighbours in the id range and almost always share their 32-bit verdict word — two word loads per group (the first id's
    // and the last id's; they are the same cache line, usually the same word) instead of four, any other word only when an id in
    // between falls on neither (never, for sorted ids whose ends are at most a word apart; the general case is kept for ids out of
    // range / unsorted tails).
    const u32* words = reinterpret_cast<const u32*>(a.verdict);
    const u64 nvd = a.n_verdict;
    u32 wx[kStreamRounds], ww[kStreamRounds];
#pragma unroll
    for (int it = 0; it < kStreamRounds; it++) {
      const u32 dx = pv[it].x - a.id_lit, dw = pv[it].w - a.id_lit;
      wx[it] = (u64)dx < nvd ? words[dx >> 5] : 0u;
      ww[it] = (u64)dw < nvd ? words[dw >> 5] : 0u;
    }
#pragma unroll
    for (int it = 0; it < kStreamRounds; it++) {
      const u64 v0 = base + (u64)it * kTile + (u64)threadIdx.x * 4;
      const u32 val[4] = {pv[it].x, pv[it].y, pv[it].z, pv[it].w};
      const u32 ix = (val[0] - a.id_lit) >> 5, iw = (val[3] - a.id_lit) >> 5;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const u32 d = val[r] - a.id_lit;
        bool keep = false;
        if ((u64)d < nvd) {
          const u32 i = d >> 5;
          const u32 word = (r == 0 || i == ix) ? wx[it] : (r == 3 || i == iw) ? ww[it] : words[i];
          keep = (word >> (d & 31u)) & 1u;
        }
        const u64 v = v0 + r;
        bits |= (u32)(keep && v >= mis && v < nv) << (it * 4 + r);
      }
    }
  } else {
#pragma unroll
    for (int it = 0; it < kStreamRounds; it++) {
      const u64 v0 = base + (u64)it * kTile + (u64)threadIdx.x * 4;
      const u32 val[4] = {pv[it].x, pv[it].y, pv[it].z, pv[it].w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const u64 v = v0 + r;
        const bool keep = v >= mis && v < nv && stream_pred<SHAPE>(a, val[r]);
        bits |= (u32)keep << (it * 4 + r);
      }
    }
  }
  a.bits[(u64)blockIdx.x * kBlock + threadIdx.x] = (unsigned short)bits;
  u32 wsum = (u32)__popc(bits);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wsum += __shfl_xor(wsum, d, 64);
  if (lane == 0) wave_tot[wave] = wsum;
  __syncthreads();
  if (threadIdx.x == 0) a.tile_count[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}
// A BGP scan hands FILTER a slice that is sorted by the predicate's column (GPOS: the objects of one predicate): the ids
// it can hold are [first, last] of that column, usually far fewer than its rows.  The typed comparison (any kinds: the
// full promotion / decimal / string path of stream_pred<2>) runs ONCE per id; the streaming pass then tests one bit per
// row — a 4-byte load that neighbouring lanes share instead of a 16-byte typed-value gather and 64 live VGPRs per lane.
__global__ __launch_bounds__(256) void value_verdict_kernel(const FilterStreamArgs a, u32 first, u64 span, u32* words) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const bool p = i < span && stream_pred<2>(a, first + (u32)i);
  const u64 m = __ballot(p);
  if ((threadIdx.x & 63) == 0) { words[(i >> 6) * 2] = (u32)m; words[(i >> 6) * 2 + 1] = (u32)(m >> 32); }
}
// When the ids of the sorted column are FEW next to the rows (a numeric property: thousands of distinct literals under
// millions of triples), the rows of one id are one long run and FILTER is a copy of the qualifying runs — the predicate
// column is never streamed: one wave per id answers the comparison and finds its run (two 65-ary searches), one
// workgroup compacts the qualifying runs and scans their lengths, and the copy moves 4 bytes in + 4 bytes out per
// SURVIVING row and column (8 sigma N instead of 8 N + 4 sigma N).  Output order = index order, as in the streaming form.
__global__ __launch_bounds__(256) void value_runs_kernel(const FilterStreamArgs a, u32 first, u32 span, u64 n, u32* run_lo, u32* run_cnt) {
  // ONE search per id (where its run starts; it ends where the next id's starts; entry `span` = where the last one ends): the
  // searches are chains of dependent HBM loads, and two per qualifying id were most of this kernel's time.  (More pivots per step
  // — 2 or 8 per lane: 129-ary / 513-ary, one or two round trips fewer — were tried: 13.8 and 18.9 us against 10.3; the
  // fully divergent gathers cost more than the round trips they save.)
  const u32 i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i > span) return;                                    // wave-uniform
  const u64 l = i < span ? wave_partition_point<false>(a.pcol, 0, n, first + i) : wave_partition_point<true>(a.pcol, 0, n, first + span - 1u);
  const bool ok = i < span && stream_pred<2>(a, first + i);   // the same answer in every lane
  if ((threadIdx.x & 63) == 0) { run_lo[i] = (u32)l; if (i < span) run_cnt[i] = ok ? 1u : 0u; }
}
// one workgroup: qualifying runs compacted in id order, exclusive scan of their lengths (span <= kRunCopyMaxIds).
// VERDICTS = true: the runs' starts come from the slice's cached table (SliceTable::ValueStarts) and the comparison of every id is
// answered HERE (run_cnt unused) — the whole plan of the copy in one launch instead of value_runs_kernel's searches + this scan.
template <bool VERDICTS>
__global__ __launch_bounds__(1024) void run_scan_kernel(const FilterStreamArgs a, const u32* run_lo, const u32* run_cnt, u32 span, u32 first,
                                                         u32* c_lo, u32* c_off, u32* c_val, u32* n_runs, u64* n_out) {
  __shared__ u32 w_rows[16], w_runs[16];
  const u32 per = (span + 1023) / 1024;                   // <= 64 ids per lane (span <= kRunCopyMaxIds): their verdicts are one 64-bit word
  const u32 i0 = threadIdx.x * per, i1 = i0 + per < span ? i0 + per : span;
  unsigned long long verdicts = 0;
  if constexpr (VERDICTS) {
    for (u32 i = i0; i < i1; i++) verdicts |= (unsigned long long)(stream_pred<2>(a, first + i) ? 1u : 0u) << (i - i0);
  }
  auto qualifies = [&](u32 i) -> u32 { if constexpr (VERDICTS) return (u32)((verdicts >> (i - i0)) & 1ull); else return run_cnt[i]; };
  u32 rows = 0, runs = 0;
  // (every load unconditional: `run_cnt[i] ? run_lo[i + 1] - run_lo[i] : 0` made the second load wait for the first one's verdict — two
  //  dependent round trips per id in a kernel that is nothing but latency)
  for (u32 i = i0; i < i1; i++) { const u32 q = qualifies(i), lo = run_lo[i], hi = run_lo[i + 1]; const u32 c = q ? hi - lo : 0u; rows += c; runs += c != 0; }   // (q: the id qualifies)
  u32 inc_rows = rows, inc_runs = runs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 t = __shfl_up(inc_rows, d, 64), u = __shfl_up(inc_runs, d, 64);
    if (lane >= d) { inc_rows += t; inc_runs += u; }
  }
  if (lane == 63) { w_rows[wave] = inc_rows; w_runs[wave] = inc_runs; }
  __syncthreads();
  u32 base_rows = 0, base_runs = 0;
  for (int w = 0; w < wave; w++) { base_rows += w_rows[w]; base_runs += w_runs[w]; }
  u32 off = base_rows + inc_rows - rows, k = base_runs + inc_runs - runs;
  for (u32 i = i0; i < i1; i++) {
    const u32 q = qualifies(i), lo = run_lo[i], hi = run_lo[i + 1];
    const u32 c = q ? hi - lo : 0u;
    if (c) { c_lo[k] = lo; c_off[k] = off; c_val[k] = first + i; k++; off += c; }
  }
  if (threadIdx.x == 1023) { c_off[k] = off; *n_runs = k; *n_out = off; }
}
constexpr u32 kRunChunk = 4096;
__global__ __launch_bounds__(256) void run_copy_kernel(const u32* c_lo, const u32* c_off, const u32* c_val, const u32* n_runs,
                                                        const u32* pcol, const u32* proj0, const u32* proj1, u32* out0, u32* out1, u32 n_out_cols) {
  const u32 R = *n_runs;
  const u32 total = c_off[R];
  const u64 P0 = (u64)blockIdx.x * kRunChunk;
  if (P0 >= total) return;
  const u32 P1 = (u32)(P0 + kRunChunk < total ? P0 + kRunChunk : total);
  u32 k = (u32)wave_partition_point<true>(c_off, 0, R, (u32)P0) - 1;     // the run that holds output row P0 (runs are non-empty)
  while (true) {
    const u32 off = c_off[k], nxt = c_off[k + 1];
    const u32 seg0 = off > (u32)P0 ? off : (u32)P0, seg1 = nxt < P1 ? nxt : P1;
    const u32 src = c_lo[k] + (seg0 - off);
    for (u32 c = 0; c < n_out_cols; c++) {
      const u32* in = c == 0 ? proj0 : proj1;
      u32* out = c == 0 ? out0 : out1;
      if (in == pcol) {                                     // the sorted column itself: one id per run
        const u32 v = c_val[k];
        for (u32 p = seg0 + threadIdx.x; p < seg1; p += 256) out[p] = v;
      } else {
        // 16 bytes per lane and access: the stores aligned (the output position is what the lanes are laid out by), the loads at
        // whatever 4-byte alignment the run starts with
        typedef u32 u32x4a __attribute__((ext_vector_type(4), aligned(16)));
        typedef u32 u32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const u32 h = (seg0 + 3u) & ~3u;                    // first output row that is a multiple of 4
        const u32 body0 = h < seg1 ? h : seg1, body1 = body0 + ((seg1 - body0) & ~3u);
        if (threadIdx.x < body0 - seg0) out[seg0 + threadIdx.x] = in[src + threadIdx.x];
        u32 p = body0 + threadIdx.x * 4u;
        for (; p + 1024 < body1; p += 2048) {               // two 16-byte loads in flight per lane
          const u32x4u v0 = *reinterpret_cast<const u32x4u*>(in + src + (p - seg0)), v1 = *reinterpret_cast<const u32x4u*>(in + src + (p - seg0) + 1024);
          *reinterpret_cast<u32x4a*>(out + p) = v0; *reinterpret_cast<u32x4a*>(out + p + 1024) = v1;
        }
        for (; p < body1; p += 1024) *reinterpret_cast<u32x4a*>(out + p) = *reinterpret_cast<const u32x4u*>(in + src + (p - seg0));
        if (threadIdx.x < seg1 - body1) out[body1 + threadIdx.x] = in[src + (body1 - seg0) + threadIdx.x];
      }
    }
    if (nxt >= P1) break;
    k++;
  }
}
// (requesting the output columns before the verdict bits have arrived — no dependent round trip bits -> columns, for filters
//  that keep a good part of their rows — was tried: 85 us against 83, the 8 resident workgroups per CU hide that latency already)
template <int NOUT>
__global__ __launch_bounds__(kBlock) void filter_write_kernel(const FilterStreamArgs a) {
  __shared__ u32 rcnt[kStreamRounds][kBlock / 64];
  const u64 n = live_rows(a.n_in_dev, a.n_in_cap);
  const u64 mis = a.head_skip, nv = n + mis;
  const u64 base = (u64)blockIdx.x * (kTile * kStreamRounds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.n_out_dev = a.tile_off[gridDim.x];   // the total: the scan runs over tiles + 1 counts
  // (offsets from per-64-tile sums kept by no-return atomics in pass 1, instead of the 12 us scan launch, were tried: the
  //  atomics cost pass 1 18 us)
  const u64 tile_base = a.tile_off[blockIdx.x];
  if (NOUT == 0 || a.tile_count[blockIdx.x] == 0) return;                            // uniform per workgroup
  const u32 bits = a.bits[(u64)blockIdx.x * kBlock + threadIdx.x];
  uint4 ov[NOUT > 0 ? NOUT : 1][kStreamRounds];
#pragma unroll
  for (int c = 0; c < NOUT; c++) {
#pragma unroll
    for (int it = 0; it < kStreamRounds; it++) {
      ov[c][it] = make_uint4(0u, 0u, 0u, 0u);
      if (((bits >> (it * 4)) & 15u) != 0) ov[c][it] = stream_load4(a.proj[c], base + (u64)it * kTile + (u64)threadIdx.x * 4, mis, nv);
    }
  }
  // index order inside the tile: row = base + it * 1024 + tid * 4 + r  =>  round-major, then wave, then lane, then r
#pragma unroll
  for (int it = 0; it < kStreamRounds; it++) {
    u32 c = (u32)__popc((bits >> (it * 4)) & 15u);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if (lane == 0) rcnt[it][wave] = c;
  }
  __syncthreads();
  u64 off = tile_base;
#pragma unroll
  for (int it = 0; it < kStreamRounds; it++) {
    u64 woff = off;
    for (int w = 0; w < wave; w++) woff += rcnt[it][w];
    off += rcnt[it][0] + rcnt[it][1] + rcnt[it][2] + rcnt[it][3];
    const u32 nib = (bits >> (it * 4)) & 15u;
    const u32 mine = (u32)__popc(nib);
    u32 incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    u64 pos = woff + (incl - mine);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (!((nib >> r) & 1u)) continue;
#pragma unroll
      for (int c = 0; c < NOUT; c++) {
        const u32 val[4] = {ov[c][it].x, ov[c][it].y, ov[c][it].z, ov[c][it].w};
        a.out[c][pos] = val[r];
      }
      pos++;
    }
  }
}
static void filter_alignment(FilterArgs& a, int shape) {
  // 16-byte loads need every column the predicate reads to share one misalignment (slices of one
  // permutation do; freshly allocated tables are aligned)
  a.head_skip = 0; a.vec_ok = 0;
  if (shape != 0) {
    const uintptr_t p = reinterpret_cast<uintptr_t>(a.in[a.prog.nodes[0].u]);
    if ((p & 3) == 0) { a.head_skip = (u32)((p >> 2) & 3); a.vec_ok = 1; }
  }
  a.vec_proj_ok = 1;
  for (u32 c = 0; c < a.n_out_cols; c++) {
    const uintptr_t p = reinterpret_cast<uintptr_t>(a.in[a.proj[c]]);
    if ((p & 3) != 0 || ((p >> 2) & 3) != a.head_skip) a.vec_proj_ok = 0;
  }
}
u64 filter_stream_tiles(const FilterArgs& a0) {
  FilterArgs a = a0;
  filter_alignment(a, 1);
  const u64 tile = (u64)kTile * kStreamRounds;
  return (a.n_in_cap + a.head_skip + tile - 1) / tile;
}
// the streaming form: a specialised shape, at most two output columns, every column in the predicate column's 16-byte phase
bool filter_streams(const FilterArgs& a0, int shape) {
  FilterArgs a = a0;
  filter_alignment(a, shape);
  // (below ~1 M rows one kernel with one reservation per 16 K rows is the shorter path: fewer launches)
  return shape != 0 && a.vec_ok && a.vec_proj_ok && a.n_out_cols <= 2 && a.n_in_cap >= (1ull << 20);
}
static FilterStreamArgs stream_args(const FilterArgs& a0, int shape) {
  FilterArgs a = a0;
  filter_alignment(a, shape);
  if (!a.stream_bits || !a.stream_counts || !a.stream_offs) fail(RDFGPU_ERR_INVALID, "streaming filter without its pass buffers");
  FilterStreamArgs f{};
  f.pcol = a.in[a.prog.nodes[0].u];
  for (u32 c = 0; c < a.n_out_cols; c++) { f.proj[c] = a.in[a.proj[c]]; f.out[c] = a.out[c]; }
  f.n_out_cols = a.n_out_cols; f.head_skip = a.head_skip;
  f.n_in_dev = a.n_in_dev; f.n_in_cap = a.n_in_cap; f.n_out_dev = a.n_out_dev;
  f.tt = a.tt; f.verdict = a.verdict; f.n_verdict = a.n_verdict;
  f.bits = a.stream_bits; f.tile_count = a.stream_counts; f.tile_off = a.stream_offs;
  if (shape == 1) { f.id_lit = a.prog.nodes[1].u; f.is_eq = a.prog.nodes[2].op == RDFGPU_EX_ID_EQ; }
  if (shape == 2 || shape == 4) { f.lit = a.prog.nodes[2]; f.op = a.prog.nodes[3].op; }
  if (shape == 4) { f.verdict = reinterpret_cast<const unsigned char*>(a.value_bits); f.n_verdict = a.value_span; f.id_lit = a.value_min; }
  return f;
}
void launch_value_runs(const FilterArgs& a, const RunCopyBuffers& b, hipStream_t s) {
  const FilterStreamArgs f = stream_args(a, 2);
  const u32 span = (u32)a.value_span;
  hipLaunchKernelGGL(value_runs_kernel, dim3((span + 1 + 3) / 4), dim3(256), 0, s, f, a.value_min, span, a.n_in_cap, b.run_lo, b.run_cnt);
}
void launch_run_scan(const FilterArgs& a, const RunCopyBuffers& b, bool verdicts_here, hipStream_t s) {
  const FilterStreamArgs f = stream_args(a, 2);
  if (verdicts_here) hipLaunchKernelGGL(run_scan_kernel<true>, dim3(1), dim3(1024), 0, s, f, b.run_lo, b.run_cnt, (u32)a.value_span, a.value_min, b.c_lo, b.c_off, b.c_val, b.n_runs, a.n_out_dev);
  else hipLaunchKernelGGL(run_scan_kernel<false>, dim3(1), dim3(1024), 0, s, f, b.run_lo, b.run_cnt, (u32)a.value_span, a.value_min, b.c_lo, b.c_off, b.c_val, b.n_runs, a.n_out_dev);
}
void launch_run_copy(const FilterArgs& a, const RunCopyBuffers& b, hipStream_t s) {
  if (!a.n_out_cols) return;
  const u32* pcol = a.in[a.prog.nodes[0].u];
  hipLaunchKernelGGL(run_copy_kernel, dim3((unsigned)((a.n_in_cap + kRunChunk - 1) / kRunChunk)), dim3(256), 0, s, b.c_lo, b.c_off, b.c_val, b.n_runs,
                     pcol, a.in[a.proj[0]], a.n_out_cols > 1 ? a.in[a.proj[1]] : nullptr, a.out[0], a.n_out_cols > 1 ? a.out[1] : nullptr, a.n_out_cols);
}
void launch_value_verdicts(const FilterArgs& a, hipStream_t s) {
  if (!a.value_span) return;
  const FilterStreamArgs f = stream_args(a, 2);
  hipLaunchKernelGGL(value_verdict_kernel, dim3((unsigned)((a.value_span + 255) / 256)), dim3(256), 0, s, f, a.value_min, a.value_span, const_cast<u32*>(a.value_bits));
}
void launch_filter_bits(const FilterArgs& a, int shape, hipStream_t s) {     // pass 1 of the streaming form
  const FilterStreamArgs f = stream_args(a, shape);
  const dim3 sg((unsigned)filter_stream_tiles(a));
  if (shape == 1) hipLaunchKernelGGL((filter_bits_kernel<1>), sg, dim3(kBlock), 0, s, f);
  else if (shape == 2) hipLaunchKernelGGL((filter_bits_kernel<2>), sg, dim3(kBlock), 0, s, f);
  else if (shape == 4) hipLaunchKernelGGL((filter_bits_kernel<4>), sg, dim3(kBlock), 0, s, f);
  else hipLaunchKernelGGL((filter_bits_kernel<3>), sg, dim3(kBlock), 0, s, f);
}
void launch_filter_write(const FilterArgs& a, int shape, hipStream_t s) {    // pass 2, after the scan of stream_counts into stream_offs
  const FilterStreamArgs f = stream_args(a, shape);
  const dim3 sg((unsigned)filter_stream_tiles(a));
  if (a.n_out_cols == 0) hipLaunchKernelGGL((filter_write_kernel<0>), sg, dim3(kBlock), 0, s, f);
  else if (a.n_out_cols == 1) hipLaunchKernelGGL((filter_write_kernel<1>), sg, dim3(kBlock), 0, s, f);
  else hipLaunchKernelGGL((filter_write_kernel<2>), sg, dim3(kBlock), 0, s, f);
}
void launch_filter(const FilterArgs& a0, int shape, hipStream_t s) {
  FilterArgs a = a0;
  filter_alignment(a, shape);
  if (filter_streams(a0, shape)) fail(RDFGPU_ERR_INVALID, "launch_filter: these arguments take the streaming form (launch_filter_bits / _write)");
  // rows per workgroup: enough workgroups to fill the chip (>= ~1024), few enough reservations
  u64 iters = (a.n_in_cap + (u64)kTile * 1024 - 1) / ((u64)kTile * 1024);
  a.iters = (u32)(iters < 1 ? 1 : iters > kFilterMaxIters ? kFilterMaxIters : iters);
  const u64 chunk = (u64)kTile * a.iters;
  const u64 g = (a.n_in_cap + a.head_skip + chunk - 1) / chunk;
  const dim3 grid((unsigned)(g ? g : 1));
  if (shape == 3) hipLaunchKernelGGL(filter_kernel<3>, grid, dim3(kBlock), 0, s, a);
  else if (shape == 1) hipLaunchKernelGGL(filter_kernel<1>, grid, dim3(kBlock), 0, s, a);
  else if (shape == 2) hipLaunchKernelGGL(filter_kernel<2>, grid, dim3(kBlock), 0, s, a);
  else hipLaunchKernelGGL(filter_kernel<0>, grid, dim3(kBlock), 0, s, a);
}

// --------------------------------------------------------------------------------------------------
// K6 CrossJoinExec: output row r = (left r / m, right r % m); writes are fully coalesced, the small
// side is re-read from L2.  (join/rewrite.rs:74-96: no shared variable => cross product)
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void cross_kernel(const CrossArgs a) {
  const u64 nl = live_rows(a.n_left_dev, a.n_left_cap), nr = live_rows(a.n_right_dev, a.n_right_cap);
  const u64 n = nl * nr;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.n_out_dev) *a.n_out_dev = n;
  const u64 base = (u64)blockIdx.x * kTile;
  if (base >= n) return;
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    const u64 r = base + (u64)k * kBlock + threadIdx.x;
    if (r >= n) break;
    const u64 i = r / nr, j = r - i * nr;
    for (u32 c = 0; c < a.n_out_cols; c++) {
      const u32 p = a.proj[c];
      a.out[c][r] = p < a.n_left_cols ? a.left[p][i] : a.right[p - a.n_left_cols][j];
    }
  }
}
void launch_cross(const CrossArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(cross_kernel, grid_for(a.n_left_cap * a.n_right_cap), dim3(kBlock), 0, s, a);
}

// --------------------------------------------------------------------------------------------------
// UnionExec: out = left rows ++ right rows, column by column (coalesced copies; the split point is the left
// input's live row count, which may only exist on the device).
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void union_kernel(const UnionArgs a) {
  const u64 nl = live_rows(a.n_left_dev, a.n_left_cap), nr = live_rows(a.n_right_dev, a.n_right_cap);
  const u64 n = nl + nr;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.n_out_dev) *a.n_out_dev = n;
  const u64 base = (u64)blockIdx.x * kTile;
  if (base >= n) return;
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    const u64 r = base + (u64)k * kBlock + threadIdx.x;
    if (r >= n) break;
    for (u32 c = 0; c < a.n_cols; c++) a.out[c][r] = r < nl ? a.left[c][r] : a.right[c][r - nl];
  }
}
void launch_union(const UnionArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(union_kernel, grid_for(a.n_left_cap + a.n_right_cap), dim3(kBlock), 0, s, a);
}

// --------------------------------------------------------------------------------------------------
// K4/K5 HashJoinExec(CollectLeft), v1: chained hash table in HBM.
//   build : next[i] = atomicExch(&heads[h(keys_i)], i)            (rows with a null key never enter:
//           NullEqualsNothing, join/rewrite.rs:89,217)
//   count : per probe row, walk the chain, compare keys, run the residual filter program -> counts[j]
//   scan  : device inclusive scan of counts (rocPRIM) -> offsets ; the total sizes the output
//   write : walk again, write matches at offsets (deterministic, probe order)
// Multiplicity-exact (bag semantics before DISTINCT).
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 hash_keys(const u32* k, u32 n) {
  u64 h = 0x9E3779B97F4A7C15ull;
  for (u32 i = 0; i < n; i++) { h ^= k[i]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
  h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 29;
  return (u32)h;
}

__global__ __launch_bounds__(kBlock) void join_build_kernel(const JoinArgs a) {
  const u64 n = live_rows(a.n_left_dev, a.n_left_cap);
  const u64 base = (u64)blockIdx.x * kTile;
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    const u64 i = base + (u64)k * kBlock + threadIdx.x;
    if (i >= a.n_left_cap) break;
    u32 key[RDFGPU_MAX_KEYS]; bool null_key = i >= n;
    if (!null_key) for (u32 q = 0; q < a.n_keys; q++) { key[q] = a.left[a.left_keys[q]][i]; null_key = null_key || key[q] == 0; }
    u32 nx = kNil;
    if (!null_key) nx = atomicExch(&a.heads[hash_keys(key, a.n_keys) & a.bucket_mask], (u32)i);
    a.next[i] = nx;
    if (a.visited) a.visited[i] = 0;
  }
}

template <bool WRITE>
__global__ __launch_bounds__(kBlock) void join_probe_kernel(const JoinArgs a) {
  const u64 n = live_rows(a.n_right_dev, a.n_right_cap);
  const u64 base = (u64)blockIdx.x * kTile;
#pragma unroll 1
  for (int k = 0; k < kItems; k++) {
    const u64 j = base + (u64)k * kBlock + threadIdx.x;
    if (j >= a.n_right_cap) break;
    u32 c = 0;
    u64 pos = 0;
    if (WRITE) pos = j ? a.counts[j - 1] : 0;
    if (j < n) {
      u32 key[RDFGPU_MAX_KEYS]; bool null_key = false;
      for (u32 q = 0; q < a.n_keys; q++) { key[q] = a.right[a.right_keys[q]][j]; null_key = null_key || key[q] == 0; }
      if (!null_key) {
        for (u32 i = a.heads[hash_keys(key, a.n_keys) & a.bucket_mask]; i != kNil; i = a.next[i]) {
          bool eq = true;
          for (u32 q = 0; q < a.n_keys; q++) eq = eq && a.left[a.left_keys[q]][i] == key[q];
          if (!eq) continue;
          if (a.has_filter) {
            const Val r = eval_program(a.prog, a.tt, [&](u32 col) { return col < a.n_left_cols ? a.left[col][i] : a.right[col - a.n_left_cols][j]; });
            if (r.lo != 1) continue;
          }
          if (WRITE) {
            for (u32 oc = 0; oc < a.n_out_cols; oc++) {
              const u32 p = a.proj[oc];
              a.out[oc][pos] = p < a.n_left_cols ? a.left[p][i] : a.right[p - a.n_left_cols][j];
            }
            pos++;
            if (a.visited) a.visited[i] = 1;
          }
          c++;
        }
      }
    }
    if (!WRITE) a.counts[j] = c;
  }
}

// NestedLoopJoinExec: no equi keys; every probe row meets every build row (small inputs only).
template <bool WRITE>
__global__ __launch_bounds__(kBlock) void nlj_kernel(const JoinArgs a) {
  const u64 n = live_rows(a.n_right_dev, a.n_right_cap);
  const u64 nl = live_rows(a.n_left_dev, a.n_left_cap);
  const u64 base = (u64)blockIdx.x * kTile;
#pragma unroll 1
  for (int k = 0; k < kItems; k++) {
    const u64 j = base + (u64)k * kBlock + threadIdx.x;
    if (j >= a.n_right_cap) break;
    u32 c = 0;
    u64 pos = 0;
    if (WRITE) pos = j ? a.counts[j - 1] : 0;
    if (j < n) {
      for (u64 i = 0; i < nl; i++) {
        if (a.has_filter) {
          const Val r = eval_program(a.prog, a.tt, [&](u32 col) { return col < a.n_left_cols ? a.left[col][i] : a.right[col - a.n_left_cols][j]; });
          if (r.lo != 1) continue;
        }
        if (WRITE) {
          for (u32 oc = 0; oc < a.n_out_cols; oc++) {
            const u32 p = a.proj[oc];
            a.out[oc][pos] = p < a.n_left_cols ? a.left[p][i] : a.right[p - a.n_left_cols][j];
          }
          pos++;
          if (a.visited) a.visited[i] = 1;
        }
        c++;
      }
    }
    if (!WRITE) a.counts[j] = c;
  }
}

// Left join tail: build rows that met no probe row are emitted once with nulls on the right
// (join/logical.rs:262-277).  Block-aggregated append behind the matched rows.
__global__ __launch_bounds__(kBlock) void join_left_unmatched_kernel(const JoinArgs a) {
  __shared__ u32 wave_tot[kBlock / 64];
  __shared__ u64 block_base;
  const u64 n = live_rows(a.n_left_dev, a.n_left_cap);
  const u64 base = (u64)blockIdx.x * kTile;
  if (base >= n) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool keep[kItems]; u32 pre[kItems]; u32 wtot = 0;
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    const u64 i = base + (u64)k * kBlock + threadIdx.x;
    keep[k] = i < n && a.visited[i] == 0;
    const unsigned long long mask = __ballot(keep[k]);
    pre[k] = wtot + lane_prefix(mask);
    wtot += (u32)__popcll(mask);
  }
  if (lane == 0) wave_tot[wave] = wtot;
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 t = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    block_base = t ? atomicAdd((unsigned long long*)a.n_out_dev, (unsigned long long)t) : 0ull;
  }
  __syncthreads();
  u64 off = block_base;
  for (int w = 0; w < wave; w++) off += wave_tot[w];
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    if (keep[k]) {
      const u64 i = base + (u64)k * kBlock + threadIdx.x;
      if (a.matched_total && off + pre[k] >= a.matched_total) continue;   // speculative sizing: never write past the block
      for (u32 oc = 0; oc < a.n_out_cols; oc++) {
        const u32 p = a.proj[oc];
        a.out[oc][off + pre[k]] = p < a.n_left_cols ? a.left[p][i] : 0u;
      }
    }
  }
}

// Build sides above the LDS limit: the same {key0,row} open-addressing table, but ONE copy in HBM (8 B per
// slot, load <= 0.5; a 285 k-row build = 8 MiB, i.e. L2 / Infinity-Cache resident), filled by this kernel.
// The probe is lds_join_kernel<.., GLOBAL = true>: identical code, the slot reads go to L2 instead of LDS.
__global__ __launch_bounds__(kBlock) void gjoin_build_kernel(const LdsJoinArgs a) {
  const u64 nb = live_rows(a.n_build_dev, a.n_build_cap);
  const u64 base = (u64)blockIdx.x * kTile;
#pragma unroll
  for (int k = 0; k < kItems; k++) {
    const u64 i = base + (u64)k * kBlock + threadIdx.x;
    if (i >= nb) break;
    Keys key;
    if (!load_keys(a.build_key, a.n_keys, i, key)) continue;
    u32 h = hash_keys4(key, a.n_keys) & a.tbl_mask;
    for (;;) {
      if (atomicCAS(&a.gslots[h].y, kNil, (u32)i) == kNil) { a.gslots[h].x = key.k[0]; break; }
      h = (h + 1) & a.tbl_mask;
    }
  }
}
void launch_gjoin_build(const LdsJoinArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(gjoin_build_kernel, grid_for(a.n_build_cap), dim3(kBlock), 0, s, a);
}

// Rows per lane and tile: 4 for multi-million-row probes and for LDS tables over ~1 M-row probes (amortises
// the per-workgroup LDS build), else 1 (many short workgroups; an HBM table has no per-workgroup build to
// amortise and its probes are latency chains that want parallelism).
int lds_join_items(u64 n_probe_cap, bool global) {
  constexpr u64 min_global = 4ull << 20, min_lds = 1ull << 20;   // measured cross-over points
  return n_probe_cap >= (global ? min_global : min_lds) ? 4 : 1;
}
int lds_join_mode(const LdsJoinArgs& a) { return a.csr_off ? kJoinTableCsr : a.direct ? kJoinTableDirect : a.gslots ? kJoinTableHash : kJoinTableLds; }
void launch_lds_join(const LdsJoinArgs& a, hipStream_t s) {
  if (a.stream_direct && direct_stream_join_ok(a)) return launch_direct_stream_join(a, s);
  const int mode = lds_join_mode(a);
  const bool global = mode != kJoinTableLds;
  const size_t tbl_lds = global ? 0 : (size_t)(a.tbl_mask + 1) * sizeof(uint2);
  const size_t lds = tbl_lds + (size_t)(kLdsBlock / 64) * a.wave_q * sizeof(uint2);
  if (a.wave_q < 64 || lds > 152 * 1024) fail(RDFGPU_ERR_INVALID, "lds join: %zu bytes of LDS", lds);
  if ((mode == kJoinTableDirect || mode == kJoinTableCsr) && a.n_keys != 1) fail(RDFGPU_ERR_INVALID, "dense join table needs exactly one key");
  const u32 rl = mode == kJoinTableCsr ? a.row_lanes_log2 : 0u;
  if (rl > 6) fail(RDFGPU_ERR_INVALID, "lds join: %u lanes per row", 1u << rl);
  // enough workgroups to cover all 256 CUs; each builds its LDS copy once and strides over the tiles
  // HBM table: no per-workgroup build, so one tile per workgroup and let the hardware overlap them.  LDS
  // table: the build is repeated per workgroup, so cap the grid by what that costs (tiny tables: no cap).
  u64 max_wg = global ? (1ull << 22) : tbl_lds > 64 * 1024 ? 256 : tbl_lds > 32 * 1024 ? 512 : 1024;
  const int items = lds_join_items(a.n_probe_cap << rl, global);
  const u64 rows = ((u64)kLdsBlock * items) >> rl;
  const u64 n_tiles = (a.n_probe_cap + rows - 1) / rows;
  const dim3 g((unsigned)(n_tiles < max_wg ? (n_tiles ? n_tiles : 1) : max_wg));
  const int fs = a.has_filter, pfs = a.has_probe_filter;   // 0 none / 1 VM / 2 col-col / 3 window ; 0 none / 1 id-literal / 2 VM
  if (fs == 0) return launch_lds_join_fs<0>(a, pfs, items, mode, g, lds, s);
  if (fs == 1) return launch_lds_join_fs<1>(a, pfs, items, mode, g, lds, s);
  if (fs == 2) return launch_lds_join_fs<2>(a, pfs, items, mode, g, lds, s);
  if (fs == 3) return launch_lds_join_fs<3>(a, pfs, items, mode, g, lds, s);
  fail(RDFGPU_ERR_INVALID, "lds join: bad filter shape %d/%d", fs, pfs);
}

// Direct-address table of a single-key build side whose keys are unique and dense (dictionary-encoded stores hand
// out contiguous ids per entity class, so `?product <p> ?v` slices usually are): direct[key - min] = row.  4 bytes
// per key instead of a 16-byte-per-row hash table at load 0.5, so a 285 k-row build is 1.1 MiB — L2-resident on
// every XCD — and a probe is ONE load with no chain.  A duplicate key raises *dup (the caller falls back to hashing).
__global__ __launch_bounds__(256) void minmax_u32_kernel(const u32* col, u64 n, u32* out /* {min, max} */) {
  u32 lo = 0xFFFFFFFFu, hi = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    const u32 v = col[i];
    if (v == 0) continue;   // null keys never join
    lo = v < lo ? v : lo; hi = v > hi ? v : hi;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const u32 l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64); lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
  // one pair of atomics per WORKGROUP, and few workgroups: same-address atomics retire at ~88 per microsecond on this part — one pair per wave
  // of 2048 workgroups was 16 K of them, 62 us for a 5.6 M-row column that streams in 5
  __shared__ u32 wlo[4], whi[4];
  if ((threadIdx.x & 63) == 0) { wlo[threadIdx.x >> 6] = lo; whi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) { lo = wlo[w] < lo ? wlo[w] : lo; hi = whi[w] > hi ? whi[w] : hi; }
    if (lo <= hi) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
  }
}
void launch_minmax_u32(const u32* col, u64 n, u32* out_dev, hipStream_t s) {
  const u64 g = (n + 256 * 16 - 1) / (256 * 16);
  hipLaunchKernelGGL(minmax_u32_kernel, dim3((unsigned)(g ? (g > 512 ? 512 : g) : 1)), dim3(256), 0, s, col, n, out_dev);
}
__global__ __launch_bounds__(kBlock) void gdirect_build_kernel(const u32* keys, u64 n, u32* direct, u32 kmin, u32 kn, u32* dup) {
  const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const u32 k = keys[i];
  if (k == 0) return;
  const u32 d = k - kmin;
  if (d >= kn) { *dup = 1u; return; }   // cannot happen when kmin / kn come from minmax of the same column
  if (atomicCAS(&direct[d], kNil, (u32)i) != kNil) *dup = 1u;
}
__global__ __launch_bounds__(256) void fill_i64_kernel(long long* p, long long v, u64 n) { const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
void launch_fill_i64(long long* p, long long v, u64 n, hipStream_t s) { if (n) hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, v, n); }
__global__ __launch_bounds__(256) void direct_values_kernel(const u32* keys, const u32* valcol, u64 n, u32 kmin, u32 kn, const TypedTable tt, long long* val, u32* bad) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 k = keys[i];
  if (k == 0) return;
  const u32 d = k - kmin;
  if (d >= kn) return;
  const Val v = enc_tv(tt, valcol[i]);
  if (v.tag != RDFGPU_TV_INTEGER || v.lo == INT64_MIN) { *bad = 1u; return; }
  val[d] = v.lo;
}
void launch_direct_values(const u32* keys, const u32* valcol, u64 n, u32 kmin, u32 kn, const TypedTable& tt, long long* val, u32* bad_dev, hipStream_t s) {
  if (n) hipLaunchKernelGGL(direct_values_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, valcol, n, kmin, kn, tt, val, bad_dev);
}
// ---- range index: CSR groups re-ordered by a decoded value of another slice (see LdsJoinArgs::range_rows) ----
__device__ __forceinline__ long long range_value_of(const u32* stage_key_col, const u32* csr_rows, u64 p, const long long* val, u32 vmin_key, u32 vn, u32& row) {
  row = csr_rows ? csr_rows[p] : (u32)p;
  const u32 k = stage_key_col[row];
  const u32 d = k - vmin_key;
  return (k != 0 && d < vn) ? val[d] : INT64_MIN;   // INT64_MIN: the stage has no row for this key — such a candidate can never pass
}
__global__ __launch_bounds__(256) void range_minmax_kernel(const u32* stage_key_col, const u32* csr_rows, u64 n, const long long* val, u32 vmin_key, u32 vn, long long* out) {
  long long lo = INT64_MAX, hi = INT64_MIN + 1;
  for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (u64)gridDim.x * blockDim.x) {
    u32 row; const long long v = range_value_of(stage_key_col, csr_rows, p, val, vmin_key, vn, row);
    if (v == INT64_MIN) continue;
    lo = v < lo ? v : lo; hi = v > hi ? v : hi;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const long long l2 = __shfl_xor(lo, d, 64), h2 = __shfl_xor(hi, d, 64); lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
  if ((threadIdx.x & 63) == 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}
__global__ __launch_bounds__(256) void range_keys_kernel(const u32* group_col, u32 gmin, const u32* stage_key_col, const u32* csr_rows, u64 n, const long long* val,
                                                          u32 vmin_key, u32 vn, long long vbase, u64* key64, u32* rows_in) {
  const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  u32 row; const long long v = range_value_of(stage_key_col, csr_rows, p, val, vmin_key, vn, row);
  const u64 g = group_col[row] - gmin;
  key64[p] = (g << 32) | (v == INT64_MIN ? 0ull : (u64)(v - vbase) + 1ull);   // biased value; 0 = no stage row
  rows_in[p] = row;
}
__global__ __launch_bounds__(256) void range_decode_kernel(const u64* key64, u64 n, u32* vals) {
  const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) vals[p] = (u32)(key64[p] & 0xFFFFFFFFull);   // the biased value (0 = no stage row)
}
void launch_range_minmax(const u32* stage_key_col, const u32* csr_rows, u64 n, const long long* val, u32 vmin_key, u32 vn, long long* out, hipStream_t s) {
  const u64 g = (n + 256 * 16 - 1) / (256 * 16);
  hipLaunchKernelGGL(range_minmax_kernel, dim3((unsigned)(g ? (g > 2048 ? 2048 : g) : 1)), dim3(256), 0, s, stage_key_col, csr_rows, n, val, vmin_key, vn, out);
}
void launch_range_keys(const u32* group_col, u32 gmin, const u32* stage_key_col, const u32* csr_rows, u64 n, const long long* val, u32 vmin_key, u32 vn,
                       long long vbase, u64* key64, u32* rows_in, hipStream_t s) {
  if (n) hipLaunchKernelGGL(range_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, group_col, gmin, stage_key_col, csr_rows, n, val, vmin_key, vn, vbase, key64, rows_in);
}
void launch_range_decode(const u64* key64_sorted, u64 n, u32* biased_vals_out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(range_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, key64_sorted, n, biased_vals_out);
}

void launch_gdirect_build(const u32* keys, u64 n, u32* direct, u32 kmin, u32 kn, u32* dup_dev, hipStream_t s) {
  const u64 g = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(gdirect_build_kernel, dim3((unsigned)(g ? g : 1)), dim3(kBlock), 0, s, keys, n, direct, kmin, kn, dup_dev);
}

// CSR form of the same idea for dense keys WITH duplicates (`?product bsbm:productFeature ?f` keyed by either end):
// off[k - min] .. off[k - min + 1] delimit the rows of key k inside rows[] (row ids grouped by key).  When the column is already sorted by
// the key (a GPOS slice keyed by its object), rows[] is the identity and is not materialised: the store's own permutation IS the join index.
// The CSR build without per-row atomics: rel[i] = key - kmin (kn for a row that joins nothing: null / out of range), and whether the column
// is sorted by the key already.  Sorted (a GPOS slice keyed by its object): the offsets are kn + 1 boundary searches over rel, nothing else.
// Otherwise one radix sort of (rel, row) pairs gives rows[] grouped by key in row order, and the same searches over the sorted rel.
// (The counting sort it replaces — csr_hist_kernel + scan + csr_scatter_kernel — pays a global atomic per row in each pass: 0.10 + 0.52 ms for
// the 5.6 M rows of BSBM-100M's productFeature slice keyed by product; this form 0.15 ms.)
__global__ __launch_bounds__(256) void csr_rel_keys_kernel(const u32* keys, u64 n, u32 kmin, u32 kn, u32* rel, u32* unsorted) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 k = keys[i];
  if (k == 0 || (i > 0 && keys[i - 1] > k)) *unsorted = 1u;   // identity rows[] needs sorted, null-free keys
  const u32 d = k - kmin;
  rel[i] = (k != 0 && d < kn) ? d : kn;
}
void launch_csr_rel_keys(const u32* keys, u64 n, u32 kmin, u32 kn, u32* rel, u32* unsorted_dev, hipStream_t s) {
  if (n) hipLaunchKernelGGL(csr_rel_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, keys, n, kmin, kn, rel, unsorted_dev);
}

void launch_join_build(const JoinArgs& a, hipStream_t s) { hipLaunchKernelGGL(join_build_kernel, grid_for(a.n_left_cap), dim3(kBlock), 0, s, a); }
void launch_join_count(const JoinArgs& a, hipStream_t s) { hipLaunchKernelGGL(join_probe_kernel<false>, grid_for(a.n_right_cap), dim3(kBlock), 0, s, a); }
void launch_join_write(const JoinArgs& a, hipStream_t s) { hipLaunchKernelGGL(join_probe_kernel<true>, grid_for(a.n_right_cap), dim3(kBlock), 0, s, a); }
void launch_join_left_unmatched(const JoinArgs& a, hipStream_t s) { hipLaunchKernelGGL(join_left_unmatched_kernel, grid_for(a.n_left_cap), dim3(kBlock), 0, s, a); }
void launch_nlj_count(const JoinArgs& a, hipStream_t s) { hipLaunchKernelGGL(nlj_kernel<false>, grid_for(a.n_right_cap), dim3(kBlock), 0, s, a); }
void launch_nlj_write(const JoinArgs& a, hipStream_t s) { hipLaunchKernelGGL(nlj_kernel<true>, grid_for(a.n_right_cap), dim3(kBlock), 0, s, a); }

// --------------------------------------------------------------------------------------------------
// Load-path utilities (index build = radix sort of three permutations + dedupe)
// --------------------------------------------------------------------------------------------------
__global__ void fill_u32_kernel(u32* p, u32 v, u64 n) { u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void iota_u32_kernel(u32* p, u64 n) { u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (u32)i; }
__global__ void gather_u32_kernel(const u32* src, const u32* idx, u32* dst, u64 n) { u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[idx[i]]; }
__global__ void pack_key_kernel(const u32* hi, const u32* lo, const u32* idx, u64* key, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const u64 r = idx ? idx[i] : i; key[i] = ((u64)hi[r] << 32) | lo[r]; }
}
__global__ void unique_flags_kernel(const u32* c0, const u32* c1, const u32* c2, const u32* c3, u32* flags, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (i == 0 || c0[i] != c0[i - 1] || c1[i] != c1[i - 1] || c2[i] != c2[i - 1] || c3[i] != c3[i - 1]) ? 1u : 0u;
}
__global__ void scatter_if_kernel(const u32* src, const u32* flags, const u32* excl, u32* dst, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) dst[excl[i]] = src[i];
}
struct Ptr4 { const u32* p[4]; };
__global__ void mark_removed_kernel(Ptr4 ix, u64 n_ix, Ptr4 rm, u64 n_rm, u32* keep) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rm) return;
  u32 q[4] = {rm.p[0][r], rm.p[1][r], rm.p[2][r], rm.p[3][r]};
  u64 a = 0, b = n_ix;
  while (a < b) {
    const u64 m = (a + b) >> 1;
    int c = 0;
    for (int k = 0; k < 4 && c == 0; k++) { const u32 v = ix.p[k][m]; c = v < q[k] ? -1 : v > q[k] ? 1 : 0; }
    if (c < 0) a = m + 1; else b = m;
  }
  if (a < n_ix && ix.p[0][a] == q[0] && ix.p[1][a] == q[1] && ix.p[2][a] == q[2] && ix.p[3][a] == q[3]) keep[a] = 0;
}
__global__ void mark_not_equal_kernel(const u32* col, u32 value, u32* keep, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keep[i] = col[i] != value ? 1u : 0u;
}
static inline dim3 flat_grid(u64 n) { u64 g = (n + 255) / 256; return dim3((unsigned)(g ? g : 1)); }
void launch_mark_not_equal(const u32* col, u32 value, u32* keep, u64 n, hipStream_t s) { if (n) hipLaunchKernelGGL(mark_not_equal_kernel, flat_grid(n), dim3(256), 0, s, col, value, keep, n); }
void launch_fill_u32(u32* p, u32 v, u64 n, hipStream_t s) { if (n) hipLaunchKernelGGL(fill_u32_kernel, flat_grid(n), dim3(256), 0, s, p, v, n); }
void launch_iota_u32(u32* p, u64 n, hipStream_t s) { if (n) hipLaunchKernelGGL(iota_u32_kernel, flat_grid(n), dim3(256), 0, s, p, n); }
void launch_gather_u32(const u32* src, const u32* idx, u32* dst, u64 n, hipStream_t s) { if (n) hipLaunchKernelGGL(gather_u32_kernel, flat_grid(n), dim3(256), 0, s, src, idx, dst, n); }
void launch_pack_key(const u32* hi, const u32* lo, const u32* idx, u64* key, u64 n, hipStream_t s) { if (n) hipLaunchKernelGGL(pack_key_kernel, flat_grid(n), dim3(256), 0, s, hi, lo, idx, key, n); }
void launch_unique_flags(const u32* c0, const u32* c1, const u32* c2, const u32* c3, u32* flags, u64 n, hipStream_t s) { if (n) hipLaunchKernelGGL(unique_flags_kernel, flat_grid(n), dim3(256), 0, s, c0, c1, c2, c3, flags, n); }
void launch_scatter_if(const u32* src, const u32* flags, const u32* excl, u32* dst, u64 n, hipStream_t s) { if (n) hipLaunchKernelGGL(scatter_if_kernel, flat_grid(n), dim3(256), 0, s, src, flags, excl, dst, n); }
void launch_mark_removed(const u32* const ix[4], u64 n_ix, const u32* const rm[4], u64 n_rm, u32* keep, hipStream_t s) {
  if (!n_rm || !n_ix) return;
  Ptr4 a{{ix[0], ix[1], ix[2], ix[3]}}, b{{rm[0], rm[1], rm[2], rm[3]}};
  hipLaunchKernelGGL(mark_removed_kernel, flat_grid(n_rm), dim3(256), 0, s, a, n_ix, b, n_rm, keep);
}

// rocPRIM device-wide primitives: prefix sums of per-row / per-tile counts and the load-path radix sort.
size_t scan_temp_bytes(u64 n) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const u32*)nullptr, (u32*)nullptr, 0u, (size_t)(n ? n : 1), rocprim::plus<u32>());
  size_t b2 = 0;
  (void)rocprim::inclusive_scan(nullptr, b2, (const u32*)nullptr, (u32*)nullptr, (size_t)(n ? n : 1), rocprim::plus<u32>());
  return (bytes > b2 ? bytes : b2) + 256;
}
void exclusive_scan_u32(const u32* in, u32* out, u64 n, void* temp, size_t temp_bytes, hipStream_t s) {
  if (!n) return;
  if (!(reinterpret_cast<uintptr_t>(in) & 15u) && launch_small_scan<false>(ScanArrayIn{in}, out, n, s)) return;   // counts, not rows: one workgroup
  RDFGPU_HIP(rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, (size_t)n, rocprim::plus<u32>(), s));
}
void inclusive_scan_u32(const u32* in, u32* out, u64 n, void* temp, size_t temp_bytes, hipStream_t s) {
  if (!n) return;
  if (!(reinterpret_cast<uintptr_t>(in) & 15u) && launch_small_scan<true>(ScanArrayIn{in}, out, n, s)) return;
  RDFGPU_HIP(rocprim::inclusive_scan(temp, temp_bytes, in, out, (size_t)n, rocprim::plus<u32>(), s));
}
size_t sort_temp_bytes(u64 n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const u64*)nullptr, (u64*)nullptr, (const u32*)nullptr, (u32*)nullptr, (size_t)(n ? n : 1), 0, 64);
  return bytes + 256;
}
void sort_pairs_u64_u32(const u64* kin, u64* kout, const u32* vin, u32* vout, u64 n, void* temp, size_t temp_bytes, hipStream_t s) {
  if (!n) return;
  RDFGPU_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, 0, 64, s));
}

// (kernels.hpp, preload_code_objects: the runtime loads a translation unit's code object at the first use of one of its kernels)
void preload_tu_kernels() { hipFuncAttributes at; RDFGPU_HIP(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(locate_kernel))); }
}  // namespace rdfgpu
