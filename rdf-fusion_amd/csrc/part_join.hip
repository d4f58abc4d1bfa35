// part_join.hip — radix-partitioned hash join with LDS-staged hash buckets: HashJoinExec(CollectLeft) for build sides
// that are NOT a cached store slice (bound tables, join outputs, sparse or multi-column keys, every build when the
// table cache is off) and are too large for one workgroup's LDS.
//
// Reference behaviour: SparqlJoinLoweringRule -> DataFusion Join -> HashJoinExec(CollectLeft), inner | left,
// NullEqualsNothing (lib/logical/src/join/rewrite.rs:126-168, :89), residual JoinFilter, projection; the build happens
// per query, inside the operator.
//
// Both sides are partitioned by the TOP bits of the key hash (rocPRIM radix sort of (partition, {row, key0, key1})
// 12-byte records: the records travel with the sort, so neither side is gathered afterwards) into partitions of ~1 K build rows;
// one workgroup owns one partition at a time: it builds the partition's {key0, index} open-addressing table in LDS (the
// LOW hash bits pick the slot; second key and row id in LDS next to it), streams the partition's probe records
// (12-byte records, coalesced), and compacts the key-equal (build row, probe row) candidates with ballot + mbcnt into
// wave-private LDS queues; a full queue runs the join filter on its candidates, reserves its output range with ONE
// atomicAdd and writes consecutive rows (the resolve phase of the fused join kernel, join_device.hpp).
// A partition with more build rows than a table holds (duplicate-heavy keys) is joined chunk by chunk: every chunk's
// table meets all probe records of the partition.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "join_device.hpp"

namespace rdfgpu {

// sort key of a row = its partition (top `bits` bits of the key hash).  A row that joins nothing (null key / beyond the live
// rows) rides in the last partition with row = kNil — the sort then needs exactly `bits` bits (16 bits = two radix passes)
__global__ __launch_bounds__(256) void part_keys_kernel(const u32* k0, const u32* k1, u32 n_keys, const u64* n_dev, u64 cap, u32 bits, u32 n_parts,
                                                         PartKeyRange kr, u32* skey, PartRec* sval) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  const u64 n = live_rows(n_dev, cap);
  Keys key; key.k[0] = 0; key.k[1] = 0; key.k[2] = 0; key.k[3] = 0;
  u32 pid = n_parts - 1, row = kNil;
  if (i < n) {
    key.k[0] = k0[i]; key.k[1] = n_keys > 1 ? k1[i] : 0u;
    const bool null_key = key.k[0] == 0 || (n_keys > 1 && key.k[1] == 0);   // NullEqualsNothing
    if (!null_key && kr.range >= 0) {                // range partitions of the probe slice's sort key; outside its id range: joins nothing
      const u32 k = key.k[kr.range];
      if (k >= kr.range_min && k <= kr.range_max) {
        const u32 rel = k - kr.range_min, c = rel >> kr.cshift;
        const uint2 d = kr.dir[c];
        pid = d.x + (u32)(((unsigned long long)(rel - (c << kr.cshift)) * d.y) >> kr.cshift);
        row = (u32)i;
      }
    } else if (!null_key) { pid = bits ? hash_keys4(key, n_keys) >> (32 - bits) : 0u; row = (u32)i; }
  }
  skey[i] = pid;
  sval[i] = PartRec{row, key.k[0], key.k[1]};
}

__device__ __forceinline__ u64 lower_bound_u32(const u32* col, u64 n, u64 target) {   // first row with col[row] >= target
  u64 lo = 0, hi = n;
  while (lo < hi) { const u64 mid = (lo + hi) >> 1; if ((u64)col[mid] < target) lo = mid + 1; else hi = mid; }
  return lo;
}
// start[k] = first position of a sorted key array with key >= k (the partition boundaries after the sort: 65 537 searches
// instead of a pass over every row)
__global__ __launch_bounds__(256) void sorted_bounds_kernel(const u32* keys, u64 n, u32 n_keys, u32* start) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k <= n_keys) start[k] = (u32)lower_bound_u32(keys, n, k);
}
// range mode, step 1 (one workgroup): slice rows per coarse id bucket -> partitions per bucket (one each + the rest of the
// n_parts in proportion to the rows) -> dir[c] = {first partition, partitions}; dir[n_coarse].x = partitions in use
constexpr u32 kPartMaxCoarse = 4096;
__global__ __launch_bounds__(1024) void part_equalise_kernel(const u32* col, u64 n, PartKeyRange kr, u32 n_parts, uint2* dir) {
  __shared__ u32 cs[kPartMaxCoarse + 1];
  __shared__ u32 sums[1024];
  const u32 tid = threadIdx.x;
  for (u32 c = tid; c <= kr.n_coarse; c += 1024) cs[c] = c == kr.n_coarse ? (u32)n : (u32)lower_bound_u32(col, n, (u64)kr.range_min + ((u64)c << kr.cshift));
  __syncthreads();
  const u32 per = (kr.n_coarse + 1023) / 1024;       // <= 4
  const u64 budget = n_parts - kr.n_coarse;
  u32 local[4] = {0u, 0u, 0u, 0u}, tot = 0;
  for (u32 u = 0; u < per; u++) {
    const u32 c = tid * per + u;
    if (c < kr.n_coarse) { local[u] = 1u + (u32)(((u64)(cs[c + 1] - cs[c]) * budget) / n); tot += local[u]; }
  }
  sums[tid] = tot;
  __syncthreads();
  for (u32 d = 1; d < 1024; d <<= 1) {
    const u32 v = tid >= d ? sums[tid - d] : 0u;
    __syncthreads();
    sums[tid] += v;
    __syncthreads();
  }
  u32 run = sums[tid] - tot;
  for (u32 u = 0; u < per; u++) {
    const u32 c = tid * per + u;
    if (c < kr.n_coarse) { dir[c] = make_uint2(run, local[u]); run += local[u]; }
  }
  if (tid == 1023) dir[kr.n_coarse] = make_uint2(sums[1023], 0u);
}
// range mode, step 2: where the slice rows of partition p start = the first row whose key is >= the partition's first id
__global__ __launch_bounds__(256) void part_range_bounds_kernel(const u32* col, u64 n, PartKeyRange kr, u32 n_parts, u32* pstart) {
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_parts) return;
  if (p >= kr.dir[kr.n_coarse].x) { pstart[p] = (u32)n; return; }
  u32 lo = 0, hi = kr.n_coarse;                      // the bucket of partition p: the last c with dir[c].x <= p (every bucket owns >= 1)
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (kr.dir[mid].x <= p) lo = mid; else hi = mid; }
  const uint2 d = kr.dir[lo];
  const u64 sub = p - d.x;
  const u64 first = (u64)kr.range_min + ((u64)lo << kr.cshift) + (((sub << kr.cshift) + d.y - 1) / d.y);   // smallest id of the bucket that maps to `sub`
  pstart[p] = (u32)lower_bound_u32(col, n, first);
}

#ifndef RDFGPU_PART_PREFETCH_ROUNDS
#define RDFGPU_PART_PREFETCH_ROUNDS 4
#endif
constexpr u32 kPartPrefetchRounds = RDFGPU_PART_PREFETCH_ROUNDS;
template <int FS>
__global__ __launch_bounds__(kLdsBlock) __attribute__((amdgpu_waves_per_eu(4))) void part_join_kernel(const LdsJoinArgs a, const PartArgs pa) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  // dynamic LDS: [slots: tbl_mask + 1 x {key0, local index}] [k1: chunk] [rows: chunk] [8 wave queues]
  uint2* slots = reinterpret_cast<uint2*>(lds_raw);
  u32* k1s = reinterpret_cast<u32*>(slots + (pa.tbl_mask + 1u));
  u32* rows = k1s + pa.chunk;
  uint2* queues = reinterpret_cast<uint2*>(rows + pa.chunk);
  __shared__ u32 wg_count, wg_cursor;
  __shared__ u64 wg_base;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 qcap = a.wave_q;
  uint2* wq = queues + (size_t)wave * qcap;
  u32 qn = 0;   // candidates in this wave's queue (wave-uniform)

  auto write_out = [&](u64 base) {   // queue entries -> consecutive output rows base .. base + qn
    for (u32 oc0 = 0; oc0 < a.n_out_cols; oc0 += 4) {
      const u32* src[4]; u32* dst[4]; bool from_build[4], on[4];
#pragma unroll
      for (u32 u = 0; u < 4; u++) {
        on[u] = oc0 + u < a.n_out_cols;
        const u32 c = a.proj[on[u] ? oc0 + u : oc0];
        from_build[u] = (c < a.n_left_cols) == (a.build_is_left != 0);
        src[u] = a.cols[c];
        dst[u] = a.out[on[u] ? oc0 + u : oc0];
      }
      for (u32 e = lane; e < qn; e += 64) {
        const uint2 m = wq[e];
        const u64 pos = base + e;
        if (pos >= a.out_cap) continue;
        u32 v[4];
#pragma unroll
        for (u32 u = 0; u < 4; u++) if (on[u]) v[u] = src[u][from_build[u] ? m.x : m.y];
#pragma unroll
        for (u32 u = 0; u < 4; u++) if (on[u]) dst[u][pos] = v[u];
      }
    }
    if (a.visited) for (u32 e = lane; e < qn; e += 64) a.visited[wq[e].x] = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };
  auto resolve = [&]() {   // join filter over the queued candidates, survivors compacted in place
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (FS == 0) return;
    u32 kept = 0;
    for (u32 g0 = 0; g0 < qn; g0 += 64) {
      const u32 e = g0 + lane;
      uint2 m = make_uint2(0u, 0u);
      bool ok = e < qn, slow = false;
      if (ok) { m = wq[e]; ok = ljoin_filter_fast<FS>(a, m.x, m.y, slow); }
      if constexpr (FS == 1 || FS == 3) { if (slow) ok = ljoin_filter_slow<FS>(a, m.x, m.y); }
      const unsigned long long mask = __ballot(ok);
      if (ok) wq[kept + lane_prefix(mask)] = m;   // kept + prefix <= e: never ahead of an unread entry of a later round
      kept += (u32)__popcll(mask);
    }
    qn = kept;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };
  // A partition is joined TWICE: a counting pass (matches that survive the filter), ONE reservation of the partition's whole
  // output range in the global counter, then the writing pass, whose waves take their places from a cursor in LDS.  One
  // reservation per queue-full was 2.1 M same-address atomics for a 0.54 G-row output — 24 ms at the chip's 88 per
  // microsecond, and an order of magnitude more when that rate collapsed (observed: the same join at 27 ms or at 250-350 ms).
  // A join with a small output (LUBM Q9's closing join: 2 M rows out of 98 M x 229 M) keeps the single pass: its few
  // reservations cost nothing, a second walk over the partition would.
  auto flush = [&](bool counting) {
    resolve();
    if (counting) { if (lane == 0 && qn) atomicAdd(&wg_count, qn); }
    else if (!pa.two_pass) {
      unsigned long long b = 0;
      if (lane == 0 && qn) {
        b = atomicAdd((unsigned long long*)a.n_out_dev, (unsigned long long)qn);
        if (b + qn > a.out_cap) *a.overflow = 1u;
      }
      b = __shfl(b, 0, 64);
      write_out(b);
    } else {
      u32 off = 0;
      if (lane == 0 && qn) off = atomicAdd(&wg_cursor, qn);
      off = __shfl(off, 0, 64);
      write_out(wg_base + off);
    }
    qn = 0;
  };

  // one probe row of the partition against the chunk's table: walk its chain, queue the key-equal candidates
  auto load_probe = [&](u32 ps, u32 t, bool live) -> uint4 {
    uint4 r = make_uint4(kNil, 0u, 0u, 0u);
    if (live) {
      if (pa.ppart) { const PartRec q = pa.ppart[ps + t]; r = make_uint4(q.row, q.k0, q.k1, 0u); }
      else {                                         // range mode: the slice itself, in place
        r.x = ps + t; r.y = pa.pcol0[ps + t]; r.z = a.n_keys > 1 ? pa.pcol1[ps + t] : 0u;
        if (r.y == 0 || (a.n_keys > 1 && r.z == 0)) r.x = kNil;   // NullEqualsNothing
      }
    }
    return r;
  };
  auto probe_one = [&](const uint4 r, bool live, bool counting) {
    Keys key; key.k[0] = r.y; key.k[1] = r.z; key.k[2] = 0; key.k[3] = 0;
    u32 h = hash_keys4(key, a.n_keys) & pa.tbl_mask;
    bool walking = live && r.x != kNil;
    for (;;) {
      u32 hit = kNil;
      while (walking) {
        const uint2 c = slots[h];
        if (c.y == kNil) { walking = false; break; }
        h = (h + 1) & pa.tbl_mask;
        if (c.x != r.y) continue;
        if (a.n_keys > 1 && k1s[c.y] != r.z) continue;
        hit = rows[c.y];
        break;
      }
      const unsigned long long found = __ballot(hit != kNil);
      if (found == 0) break;
      const u32 n_found = (u32)__popcll(found);
      if (qn + n_found > qcap) flush(counting);      // wave-uniform: the queue is empty afterwards and 64 <= qcap
      if (hit != kNil) wq[qn + lane_prefix(found)] = make_uint2(hit, r.x);
      qn += n_found;
    }
  };
  // Every memory round trip of a partition used to be exposed: `load a round of records -> use it`, ~3 build rounds and ~7 probe rounds
  // per partition, with two workgroups per CU (the LDS table) to hide them behind — most of the 26 us a partition took.  Now the
  // records of the first kPfBuild build rounds and the first kPfProbe probe rounds are requested TOGETHER, before the table is even
  // cleared, and sit in registers (36 VGPRs) while the LDS phases run: one round trip per partition instead of ten.  Rounds beyond
  // the prefetch (big partitions, later chunks) load in the loop as before.
  // Only for the join without a VM / window filter (FS 0, 2: 40 VGPRs without the prefetch): the filtered forms sit at ~100 VGPRs already
  // and would spill the prefetched records into scratch inside the probe loop (measured at compile time: 107 spilled VGPRs for FS 1).
  // Depth: kPartPrefetchRounds probe rounds — deep enough for the usual partition (~3.5 k probe rows = 7 rounds of 512), shallow enough
  // to stay under 84 VGPRs = 6 waves per SIMD = the three workgroups per CU a 52 KB-LDS join fits (108 VGPRs at depth 8: two).
  constexpr u32 kPfProbe = (FS == 0 || FS == 2) ? kPartPrefetchRounds : 0u, kPfBuild = (FS == 0 || FS == 2) ? 4u : 0u;

  for (u32 p = blockIdx.x; p < pa.n_parts; p += gridDim.x) {
    const u32 bs = pa.bstart[p], be = pa.bstart[p + 1], ps = pa.pstart[p], pe = pa.pstart[p + 1];
    if (bs >= be || ps >= pe) continue;              // uniform per workgroup
    const u32 n_probe = pe - ps;
    PartRec pf_b[kPfBuild ? kPfBuild : 1]; uint4 pf_p[kPfProbe ? kPfProbe : 1];
    {
      const u32 nb0 = be - bs < pa.chunk ? be - bs : pa.chunk;
#pragma unroll
      for (u32 r = 0; r < kPfBuild; r++) { const u32 i = r * kLdsBlock + tid; pf_b[r] = i < nb0 ? pa.bpart[bs + i] : PartRec{kNil, 0u, 0u}; }
#pragma unroll
      for (u32 r = 0; r < kPfProbe; r++) { const u32 t = r * kLdsBlock + tid; pf_p[r] = load_probe(ps, t, t < n_probe); }
    }
    __syncthreads();
    if (tid == 0) { wg_count = 0; wg_cursor = 0; }
    for (int pass = pa.two_pass ? 0 : 1; pass < 2; pass++) {
      const bool counting = pass == 0;
      for (u32 cb = bs; cb < be; cb += pa.chunk) {
        const u32 nb = be - cb < pa.chunk ? be - cb : pa.chunk;
        __syncthreads();                             // every wave is done with the previous table
        for (u32 s = tid; s <= pa.tbl_mask; s += kLdsBlock) slots[s] = make_uint2(0u, kNil);
        __syncthreads();
        auto insert = [&](const PartRec q, u32 i) {
          k1s[i] = q.k1; rows[i] = q.row;
          if (q.row == kNil) return;                 // a row that joins nothing
          Keys key; key.k[0] = q.k0; key.k[1] = q.k1; key.k[2] = 0; key.k[3] = 0;
          u32 h = hash_keys4(key, a.n_keys) & pa.tbl_mask;
          for (;;) {
            if (atomicCAS(&slots[h].y, kNil, i) == kNil) { slots[h].x = q.k0; break; }
            h = (h + 1) & pa.tbl_mask;
          }
        };
        if (cb == bs) {                              // the first chunk: its first rounds are in registers already
#pragma unroll
          for (u32 r = 0; r < kPfBuild; r++) { const u32 i = r * kLdsBlock + tid; if (i < nb) insert(pf_b[r], i); }
          for (u32 i = kPfBuild * kLdsBlock + tid; i < nb; i += kLdsBlock) insert(pa.bpart[cb + i], i);
        } else for (u32 i = tid; i < nb; i += kLdsBlock) insert(pa.bpart[cb + i], i);
        __syncthreads();
#pragma unroll
        for (u32 r = 0; r < kPfProbe; r++) {         // uniform trip count per workgroup
          if (r * kLdsBlock >= n_probe) break;
          probe_one(pf_p[r], r * kLdsBlock + tid < n_probe, counting);
        }
        for (u32 t0 = kPfProbe * kLdsBlock; t0 < n_probe; t0 += kLdsBlock) {
          const u32 t = t0 + tid;
          const bool live = t < n_probe;
          probe_one(load_probe(ps, t, live), live, counting);
        }
      }
      if (pa.two_pass) flush(counting);              // what is still queued belongs to this partition's range (single pass: the queue carries over)
      __syncthreads();
      if (counting) {
        if (tid == 0) {
          const u32 t = wg_count;
          u64 b = 0;
          if (t) {
            b = atomicAdd((unsigned long long*)a.n_out_dev, (unsigned long long)t);
            if (b + t > a.out_cap) *a.overflow = 1u;
          }
          wg_base = b;
        }
        __syncthreads();
        if (wg_count == 0) break;                    // uniform: nothing to write for this partition
      }
    }
  }
  if (!pa.two_pass) flush(false);                    // single pass: what is still queued leaves with one last reservation per wave
}

// ---- host side -------------------------------------------------------------------------------------------------------
void launch_part_keys(const u32* k0, const u32* k1, u32 n_keys, const u64* n_dev, u64 cap, u32 bits, u32 n_parts, PartKeyRange kr, u32* skey, PartRec* sval, hipStream_t s) {
  if (cap) hipLaunchKernelGGL(part_keys_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, s, k0, k1, n_keys, n_dev, cap, bits, n_parts, kr, skey, sval);
}
void launch_part_equalise(const u32* sorted_col, u64 n, PartKeyRange kr, u32 n_parts, uint2* dir, hipStream_t s) {
  if (kr.n_coarse == 0 || kr.n_coarse > kPartMaxCoarse || kr.n_coarse > n_parts || n == 0) fail(RDFGPU_ERR_INVALID, "range partitions: %u coarse buckets for %u partitions", kr.n_coarse, n_parts);
  hipLaunchKernelGGL(part_equalise_kernel, dim3(1), dim3(1024), 0, s, sorted_col, n, kr, n_parts, dir);
}
void launch_part_range_bounds(const u32* sorted_col, u64 n, PartKeyRange kr, u32 n_parts, u32* pstart, hipStream_t s) {
  hipLaunchKernelGGL(part_range_bounds_kernel, dim3((n_parts + 1 + 255) / 256), dim3(256), 0, s, sorted_col, n, kr, n_parts, pstart);
}
void launch_sorted_bounds(const u32* sorted_keys, u64 n, u32 n_keys, u32* start, hipStream_t s) {
  hipLaunchKernelGGL(sorted_bounds_kernel, dim3((n_keys + 1 + 255) / 256), dim3(256), 0, s, sorted_keys, n, n_keys, start);
}
size_t part_sort_temp_bytes(u64 n, u32 bits) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const u32*)nullptr, (u32*)nullptr, (const PartRec*)nullptr, (PartRec*)nullptr, (size_t)(n ? n : 1), 0, bits);
  return bytes + 256;
}
void part_sort(const u32* kin, u32* kout, const PartRec* vin, PartRec* vout, u64 n, u32 bits, void* temp, size_t temp_bytes, hipStream_t s) {
  if (!n) return;
  RDFGPU_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, 0, bits, s));
}
size_t part_join_lds_bytes(const LdsJoinArgs& a, const PartArgs& pa) {
  return (size_t)(pa.tbl_mask + 1) * sizeof(uint2) + 2ull * pa.chunk * sizeof(u32) + (size_t)(kLdsBlock / 64) * a.wave_q * sizeof(uint2);
}
template <int FS> static void launch_part_join_fs(const LdsJoinArgs& a, const PartArgs& pa, dim3 g, size_t lds, hipStream_t s) {
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    RDFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(part_join_kernel<FS>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  });
  hipLaunchKernelGGL((part_join_kernel<FS>), g, dim3(kLdsBlock), lds, s, a, pa);
}
void launch_part_join(const LdsJoinArgs& a, const PartArgs& pa, hipStream_t s) {
  const size_t lds = part_join_lds_bytes(a, pa);
  if (lds > 152 * 1024 || a.wave_q < 64) fail(RDFGPU_ERR_INVALID, "partitioned join: %zu bytes of LDS", lds);
  if (a.n_keys < 1 || a.n_keys > 2) fail(RDFGPU_ERR_INVALID, "partitioned join takes one or two key columns");
  const u32 wgs = pa.n_parts < 8192 ? pa.n_parts : 8192;   // every workgroup strides over the partitions
  const dim3 g(wgs ? wgs : 1);
  switch (a.has_filter) {
    case 0: return launch_part_join_fs<0>(a, pa, g, lds, s);
    case 1: return launch_part_join_fs<1>(a, pa, g, lds, s);
    case 2: return launch_part_join_fs<2>(a, pa, g, lds, s);
    case 3: return launch_part_join_fs<3>(a, pa, g, lds, s);
  }
  fail(RDFGPU_ERR_INVALID, "partitioned join: bad filter shape %u", a.has_filter);
}

}  // namespace rdfgpu
