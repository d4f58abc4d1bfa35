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
  for (u32 k = 0; k < job.n_levels && lo < hi; k++) {
    const u32* c = job.col[k];
    const u32 from = job.from[k], to = job.to[k];
    const u64 nlo = wave_partition_point<false>(c, lo, hi, from);   // lower_bound(from)
    const u64 nhi = wave_partition_point<true>(c, nlo, hi, to);     // upper_bound(to)
    lo = nlo; hi = nhi;
    if (from != to) break;            // below a proper range the inner levels are not contiguous (:240)
  }
  if (lo > hi) lo = hi;
  if (threadIdx.x == 0) { lo_hi[2 * j] = lo; lo_hi[2 * j + 1] = hi; }
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
  if constexpr (SHAPE == 1) {  // col <ID_EQ|ID_NEQ> object-id literal; null on either side => dropped
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
void launch_filter(const FilterArgs& a0, int shape, hipStream_t s) {
  FilterArgs a = a0;
  // rows per workgroup: enough workgroups to fill the chip (>= ~1024), few enough reservations
  u64 iters = (a.n_in_cap + (u64)kTile * 1024 - 1) / ((u64)kTile * 1024);
  a.iters = (u32)(iters < 1 ? 1 : iters > kFilterMaxIters ? kFilterMaxIters : iters);
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
  const u64 chunk = (u64)kTile * a.iters;
  const u64 g = (a.n_in_cap + a.head_skip + chunk - 1) / chunk;
  const dim3 grid((unsigned)(g ? g : 1));
  if (shape == 1) hipLaunchKernelGGL(filter_kernel<1>, grid, dim3(kBlock), 0, s, a);
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

__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(v, d, 64); if ((int)(threadIdx.x & 63) >= d) v += t; }
  return v;
}
__device__ __forceinline__ u32 ljoin_col(const LdsJoinArgs& a, u32 c, u64 i, u64 j) {  // column c of [left cols, right cols]
  const bool from_build = (c < a.n_left_cols) == (a.build_is_left != 0);   // wave-uniform
  return a.cols[c][from_build ? i : j];
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
__device__ __forceinline__ u32 chain_val(const ColRef& c, u32 i, u32 j, u32 r) { return c.ptr[c.src == 0 ? j : c.src == 1 ? i : r]; }
__device__ __forceinline__ u32 chain_lookup(const ChainStage& st, u32 i, u32 j, bool live = true) {
  const u32 key = st.key.ptr[live ? (st.key.src ? i : j) : 0u];
  const u32 d = key - st.kmin;
  const bool in = live && key != 0 && d < st.kn;          // null keys never join
  const u32 row = st.direct[in ? d : 0u];
  return in ? row : kNil;                                  // kNil = no row with this key
}
__device__ __forceinline__ bool stage_filter_fast(const LdsJoinArgs& a, const ChainStage& st, u32 i, u32 j, u32 r, bool& undecided) {
  undecided = false;
  if (st.fs == 2) {   // wave-uniform
    const u32 va = chain_val(st.f[0], i, j, r), vb = chain_val(st.f[1], i, j, r);
    return va != 0 && vb != 0 && (va == vb) == (st.is_eq != 0);
  }
  const bool same = st.f[0].ptr == st.f[2].ptr && st.f[0].src == st.f[2].src && st.f[1].ptr == st.f[3].ptr && st.f[1].src == st.f[3].src;
  const u32 ix0 = chain_val(st.f[0], i, j, r), iy0 = chain_val(st.f[1], i, j, r);
  const u32 ix1 = same ? ix0 : chain_val(st.f[2], i, j, r), iy1 = same ? iy0 : chain_val(st.f[3], i, j, r);
  return window_fast(a.tt, ix0, iy0, ix1, iy1, same, st.l0, st.l1, undecided);
}
// the stage's filter with the full reference semantics (fs 2 is always decided by the fast half)
__device__ __forceinline__ bool stage_filter_slow(const LdsJoinArgs& a, const ChainStage& st, u32 i, u32 j, u32 r) {
  if (st.fs != 3) { bool und; return stage_filter_fast(a, st, i, j, r, und); }
  return window_slow(a.tt, chain_val(st.f[0], i, j, r), chain_val(st.f[1], i, j, r), chain_val(st.f[2], i, j, r), chain_val(st.f[3], i, j, r), st.l0, st.l1);
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
constexpr int kResolveUnroll = 4;

template <int FS, int PFS, int ITEMS, int MODE, bool CHAIN>
__global__ __launch_bounds__(kLdsBlock) void lds_join_kernel(const LdsJoinArgs a) {
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
      if (!load_keys(a.build_key, a.n_keys, i, key)) continue;
      u32 h = hash_keys4(key, a.n_keys) & a.tbl_mask;
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
          for (u32 t = 0; t < (u32)kMaxChain; t++) if (t < a.n_chain) rr[t] = chain_lookup(a.chain[t], m.x, m.y);
        }
        u32 v[4];
#pragma unroll
        for (u32 u = 0; u < 4; u++) if (on[u]) {
          const u32 row = from[u] == 0 ? m.y : from[u] == 1 ? m.x : from[u] == 2 ? rr[0] : from[u] == 3 ? rr[1] : rr[2];
          v[u] = src[u][row];
        }
#pragma unroll
        for (u32 u = 0; u < 4; u++) if (on[u]) dst[u][pos] = v[u];
      }
    }
    if (a.visited) for (u32 e = lane; e < qn; e += 64) a.visited[wq[e].x] = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };

  Keys key[ITEMS]; u32 h[ITEMS]; uint2 s[ITEMS]; bool walking[ITEMS]; u32 pend[ITEMS];
  u64 tile = blockIdx.x;
  bool tile_loaded = false, exhausted = false;   // wave-uniform
  for (;;) {
    // ---- fill: walk tiles until the queue cannot take the next ballot's candidates or the tiles run out ----
    bool full = false;
    while (!full) {
      if (!tile_loaded) {
        if (tile >= n_tiles) { exhausted = true; break; }
        const u64 base = tile * kTileRows;
        // the tile's probe keys first (independent coalesced loads in flight together), then the first table slot of every row
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const u64 j = base + (((u32)k * kLdsBlock + tid) >> rl);
          walking[k] = j < np && load_keys(a.probe_key, a.n_keys, j, key[k]);
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
          const u64 j = base + (((u32)k * kLdsBlock + tid) >> rl);
          walking[k] = walking[k] && lprobe_filter<PFS>(a, j);
          s[k] = make_uint2(0u, kNil);
          if constexpr (DIRECT) {   // unique dense keys: the one candidate is a single 4-byte load, no chain
            h[k] = key[k].k[0] - a.direct_min;
            if (walking[k] && h[k] < a.direct_n) s[k].y = a.direct[h[k]];
          } else if constexpr (CSR) {   // dense keys with duplicates: s = [cursor, end) into the key's row list
            h[k] = key[k].k[0] - a.direct_min;
            s[k] = make_uint2(0u, 0u);
            if (walking[k] && h[k] < a.direct_n) { s[k].x = a.csr_off[h[k]] + (tid & ((1u << rl) - 1u)); s[k].y = a.csr_off[h[k] + 1]; }
          } else {
            h[k] = hash_keys4(key[k], a.n_keys) & a.tbl_mask;
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
              if (s[k].x < s[k].y) { hit = a.csr_rows ? a.csr_rows[s[k].x] : s[k].x; s[k].x += 1u << rl; }
              else walking[k] = false;
            }
          } else if (hit == kNil) {
            while (walking[k]) {
              if (s[k].y == kNil) { walking[k] = false; break; }
              const uint2 c = s[k];
              h[k] = (h[k] + 1) & a.tbl_mask;
              s[k] = slots[h[k]];   // issued before the candidate is examined
              if (c.x != key[k].k[0]) continue;
              bool eq = true;
#pragma unroll
              for (u32 q = 1; q < RDFGPU_MAX_KEYS; q++) if (q < a.n_keys) eq = eq && a.build_key[q][c.y] == key[k].k[q];
              if (!eq) continue;
              hit = c.y;
              break;
            }
          }
          const unsigned long long found = __ballot(hit != kNil);
          if (found == 0) break;
          const u32 n_found = (u32)__popcll(found);
          if (qn + n_found > qcap) { pend[k] = hit; full = true; break; }
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
            const u32 v = pc[ok[u] ? (post_from_build ? m[u].x : m[u].y) : 0u];
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
        u32 kept = 0;
        for (u32 g0 = 0; g0 < qn; g0 += 64 * kResolveUnroll) {
          uint2 m[kResolveUnroll]; bool ok[kResolveUnroll], slow[kResolveUnroll]; u32 r[kResolveUnroll];
#pragma unroll
          for (int u = 0; u < kResolveUnroll; u++) {
            const u32 e = g0 + (u32)u * 64 + lane;
            ok[u] = e < qn; m[u] = make_uint2(0u, 0u); slow[u] = false;
            if (ok[u]) m[u] = wq[e];
          }
#pragma unroll
          for (int u = 0; u < kResolveUnroll; u++) r[u] = chain_lookup(st, m[u].x, m[u].y, ok[u]);   // branch-free: dead lanes read row 0
#pragma unroll
          for (int u = 0; u < kResolveUnroll; u++) ok[u] = ok[u] && r[u] != kNil;
          if (st.fs != 0) {   // wave-uniform
#pragma unroll
            for (int u = 0; u < kResolveUnroll; u++) {
              bool und;
              const u32 ci = ok[u] ? m[u].x : 0u, cj = ok[u] ? m[u].y : 0u, cr = ok[u] ? r[u] : 0u;
              const bool pass = stage_filter_fast(a, st, ci, cj, cr, und);
              slow[u] = ok[u] && und;
              ok[u] = ok[u] && !und && pass;
            }
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

template <int FS, int PFS, int ITEMS, int MODE, bool CHAIN>
static void launch_lds_join_tc(const LdsJoinArgs& a, dim3 g, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {   // dynamic LDS above 64 KiB has to be opted into, per kernel instance
    RDFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_join_kernel<FS, PFS, ITEMS, MODE, CHAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((lds_join_kernel<FS, PFS, ITEMS, MODE, CHAIN>), g, dim3(kLdsBlock), lds, s, a);
}
template <int FS, int PFS, int ITEMS, int MODE>
static void launch_lds_join_t(const LdsJoinArgs& a, dim3 g, size_t lds, hipStream_t s) {
  // the fused lookup chain exists for HBM-table joins without a VM filter or a fused probe-side FilterExec
  if constexpr (FS != 1 && PFS == 0 && MODE != kJoinTableLds) { if (a.n_chain) return launch_lds_join_tc<FS, PFS, ITEMS, MODE, true>(a, g, lds, s); }
  if (a.n_chain) fail(RDFGPU_ERR_INVALID, "lds join: lookup chain on an unsupported join shape");
  launch_lds_join_tc<FS, PFS, ITEMS, MODE, false>(a, g, lds, s);
}
// Rows per lane and tile: 4 for multi-million-row probes and for LDS tables over ~1 M-row probes (amortises
// the per-workgroup LDS build), else 1 (many short workgroups; an HBM table has no per-workgroup build to
// amortise and its probes are latency chains that want parallelism).
int lds_join_items(u64 n_probe_cap, bool global) {
  static const u64 min_global = [] { const char* e = std::getenv("RDFGPU_JOIN_ITEMS4_MIN_GLOBAL"); return e ? std::strtoull(e, nullptr, 10) : (4ull << 20); }();
  static const u64 min_lds = [] { const char* e = std::getenv("RDFGPU_JOIN_ITEMS4_MIN_LDS"); return e ? std::strtoull(e, nullptr, 10) : (1ull << 20); }();
  return n_probe_cap >= (global ? min_global : min_lds) ? 4 : 1;
}
int lds_join_mode(const LdsJoinArgs& a) { return a.csr_off ? kJoinTableCsr : a.direct ? kJoinTableDirect : a.gslots ? kJoinTableHash : kJoinTableLds; }
void launch_lds_join(const LdsJoinArgs& a, hipStream_t s) {
  const int mode = lds_join_mode(a);
  const bool global = mode != kJoinTableLds;
  const size_t tbl_lds = global ? 0 : (size_t)(a.tbl_mask + 1) * sizeof(uint2);
  const size_t lds = tbl_lds + (size_t)(kLdsBlock / 64) * a.wave_q * sizeof(uint2);
  if (a.wave_q < 64 || lds > 152 * 1024) fail(RDFGPU_ERR_INVALID, "lds join: %zu bytes of LDS", lds);
  if ((mode == kJoinTableDirect || mode == kJoinTableCsr) && a.n_keys != 1) fail(RDFGPU_ERR_INVALID, "dense join table needs exactly one key");
  const u32 rl = mode == kJoinTableCsr ? a.row_lanes_log2 : 0u;
  if (rl > 6) fail(RDFGPU_ERR_INVALID, "lds join: %u lanes per row", 1u << rl);
  // enough workgroups to cover all 256 CUs; each builds its LDS copy once and strides over the tiles
  static const u64 wg_cap = [] { const char* e = std::getenv("RDFGPU_JOIN_MAX_WG"); return e ? std::strtoull(e, nullptr, 10) : 0ull; }();
  // HBM table: no per-workgroup build, so one tile per workgroup and let the hardware overlap them.  LDS
  // table: the build is repeated per workgroup, so cap the grid by what that costs (tiny tables: no cap).
  u64 max_wg = global ? (1ull << 22) : tbl_lds > 64 * 1024 ? 256 : tbl_lds > 32 * 1024 ? 512 : 1024;
  if (wg_cap) max_wg = wg_cap;
  const int items = lds_join_items(a.n_probe_cap << rl, global);
  const u64 rows = ((u64)kLdsBlock * items) >> rl;
  const u64 n_tiles = (a.n_probe_cap + rows - 1) / rows;
  const dim3 g((unsigned)(n_tiles < max_wg ? (n_tiles ? n_tiles : 1) : max_wg));
  const int fs = a.has_filter, pfs = a.has_probe_filter;   // 0 none / 1 VM / 2 col-col / 3 window ; 0 none / 1 id-literal / 2 VM
#define RDFGPU_LJI(F, P, M) { if (items == 4) return launch_lds_join_t<F, P, 4, M>(a, g, lds, s); return launch_lds_join_t<F, P, 1, M>(a, g, lds, s); }
#define RDFGPU_LJ(F, P) if (fs == F && pfs == P) { if (mode == kJoinTableCsr) RDFGPU_LJI(F, P, kJoinTableCsr) else if (mode == kJoinTableDirect) RDFGPU_LJI(F, P, kJoinTableDirect) else if (mode == kJoinTableHash) RDFGPU_LJI(F, P, kJoinTableHash) else RDFGPU_LJI(F, P, kJoinTableLds) }
  RDFGPU_LJ(0, 0) RDFGPU_LJ(0, 1) RDFGPU_LJ(0, 2)
  RDFGPU_LJ(1, 0) RDFGPU_LJ(1, 1) RDFGPU_LJ(1, 2)
  RDFGPU_LJ(2, 0) RDFGPU_LJ(2, 1) RDFGPU_LJ(2, 2)
  RDFGPU_LJ(3, 0) RDFGPU_LJ(3, 1) RDFGPU_LJ(3, 2)
#undef RDFGPU_LJ
#undef RDFGPU_LJI
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
  if ((threadIdx.x & 63) == 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}
void launch_minmax_u32(const u32* col, u64 n, u32* out_dev, hipStream_t s) {
  const u64 g = (n + 256 * 16 - 1) / (256 * 16);
  hipLaunchKernelGGL(minmax_u32_kernel, dim3((unsigned)(g ? (g > 2048 ? 2048 : g) : 1)), dim3(256), 0, s, col, n, out_dev);
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
void launch_gdirect_build(const u32* keys, u64 n, u32* direct, u32 kmin, u32 kn, u32* dup_dev, hipStream_t s) {
  const u64 g = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(gdirect_build_kernel, dim3((unsigned)(g ? g : 1)), dim3(kBlock), 0, s, keys, n, direct, kmin, kn, dup_dev);
}

// CSR form of the same idea for dense keys WITH duplicates (`?product bsbm:productFeature ?f` keyed by either end):
// off[k - min] .. off[k - min + 1] delimit the rows of key k inside rows[] (row ids grouped by key, by a counting
// sort: histogram -> exclusive scan -> scatter).  When the column is already sorted by the key (a GPOS slice keyed
// by its object), rows[] is the identity and is not materialised: the store's own permutation IS the join index.
__global__ __launch_bounds__(kBlock) void csr_hist_kernel(const u32* keys, u64 n, u32 kmin, u32 kn, u32* counts, u32* unsorted) {
  const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const u32 k = keys[i];
  if (k == 0 || (i > 0 && keys[i - 1] > k)) *unsorted = 1u;   // identity rows[] needs sorted, null-free keys
  if (k == 0) return;
  const u32 d = k - kmin;
  if (d < kn) atomicAdd(&counts[d], 1u);
}
__global__ __launch_bounds__(kBlock) void csr_scatter_kernel(const u32* keys, u64 n, u32 kmin, u32 kn, u32* cursor, u32* rows) {
  const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const u32 k = keys[i];
  if (k == 0) return;
  const u32 d = k - kmin;
  if (d < kn) rows[atomicAdd(&cursor[d], 1u)] = (u32)i;
}
void launch_csr_hist(const u32* keys, u64 n, u32 kmin, u32 kn, u32* counts, u32* unsorted_dev, hipStream_t s) {
  const u64 g = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(csr_hist_kernel, dim3((unsigned)(g ? g : 1)), dim3(kBlock), 0, s, keys, n, kmin, kn, counts, unsorted_dev);
}
void launch_csr_scatter(const u32* keys, u64 n, u32 kmin, u32 kn, u32* cursor, u32* rows, hipStream_t s) {
  const u64 g = (n + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(csr_scatter_kernel, dim3((unsigned)(g ? g : 1)), dim3(kBlock), 0, s, keys, n, kmin, kn, cursor, rows);
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
static inline dim3 flat_grid(u64 n) { u64 g = (n + 255) / 256; return dim3((unsigned)(g ? g : 1)); }
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
  RDFGPU_HIP(rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, (size_t)n, rocprim::plus<u32>(), s));
}
void inclusive_scan_u32(const u32* in, u32* out, u64 n, void* temp, size_t temp_bytes, hipStream_t s) {
  if (!n) return;
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

}  // namespace rdfgpu
