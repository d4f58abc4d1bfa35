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
// one workgroup owns one partition at a time: it builds the partition's {key0, key1} open-addressing table in LDS (the
// LOW hash bits pick the slot; the build row id in a parallel array indexed by slot), streams the partition's probe records
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
  {   // cs[c] = first row of coarse bucket c: a lane's (up to five) binary searches advance TOGETHER — five loads in flight per step instead
      // of five searches of 28 dependent steps one after the other (90 us for this one-workgroup kernel; the searches were all of it)
    constexpr u32 kS = (kPartMaxCoarse + 1024) / 1024;   // 5
    u64 lo[kS], hi[kS], target[kS];
#pragma unroll
    for (u32 u = 0; u < kS; u++) {
      const u32 c = tid + u * 1024;
      target[u] = (u64)kr.range_min + ((u64)c << kr.cshift);
      lo[u] = 0; hi[u] = c < kr.n_coarse ? n : 0;         // (bucket n_coarse and beyond: nothing to search)
    }
    for (int step = 0; step < 33; step++) {              // n < 2^32 rows
      u32 v[kS]; u64 mid[kS];
#pragma unroll
      for (u32 u = 0; u < kS; u++) { mid[u] = (lo[u] + hi[u]) >> 1; v[u] = lo[u] < hi[u] ? col[mid[u]] : 0u; }
#pragma unroll
      for (u32 u = 0; u < kS; u++) if (lo[u] < hi[u]) { if ((u64)v[u] < target[u]) lo[u] = mid[u] + 1; else hi[u] = mid[u]; }
    }
#pragma unroll
    for (u32 u = 0; u < kS; u++) { const u32 c = tid + u * 1024; if (c <= kr.n_coarse) cs[c] = c == kr.n_coarse ? (u32)n : (u32)lo[u]; }
  }
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

// BIG: the two-pass form of a join with a large output (pa.two_pass, known to the host).  There the kernel is bound by the latency of
// the gathers behind every full queue — the filter's operands in both passes, the payload columns in the second — and not by the walk:
// its resolve and write-out phases keep four 64-entry groups of the queue in flight at once (8 / 16 gathers per lane outstanding instead
// of 2 / 4).  The single-pass form (small outputs: LUBM Q9's closing join) keeps the lean loops, whose registers its walk wants.
// INL (BIG, FS = 2, one operand of the `col <=|!=> col` filter on each side): the filter is decided during the walk from the LDS table
// (rowof holds {row, the build row's operand}) and the probe row's operand in a register.
template <int FS, bool RANGE, bool BIG, bool INL = false>
__global__ __launch_bounds__(kLdsBlock) __attribute__((amdgpu_waves_per_eu(4))) void part_join_kernel(const LdsJoinArgs a, const PartArgs pa) {
  static_assert(!INL || (BIG && FS == 2), "the in-walk filter belongs to the large-output form of a col-col filter");
  constexpr int FSQ = INL ? 0 : FS;                  // the filter shape the queue phases still have to evaluate
  extern __shared__ __align__(16) unsigned char lds_raw[];
  // dynamic LDS: [slots: tbl_mask + 1 x {key0, key1}; key0 == 0 = empty: a null key joins nothing and is never inserted]
  //              [rowof: tbl_mask + 1 build row ids, by slot] [8 wave queues]
  // One hop of a probe is ONE ds_read_b64 and two compares.  The first form kept {key0, local index} in the slot and the second
  // key and the row id in arrays by index: two to four dependent LDS round trips per hop inside a divergent loop the compiler
  // lowered to ~35 scalar instructions and 7 branches per hop — 1.0 G scalar against 0.53 G vector instructions for LUBM Q9's
  // closing join, waves executing 26 % of their cycles (profiles/r03_lubm_part_join_sq_counters_before.json).
  uint2* slots = reinterpret_cast<uint2*>(lds_raw);
  u32* rowof = reinterpret_cast<u32*>(slots + (pa.tbl_mask + 1u));
  uint2* rowfv = reinterpret_cast<uint2*>(rowof);   // INL: {row, filter operand} by slot instead
  uint2* queues = reinterpret_cast<uint2*>(rowof + (size_t)(pa.tbl_mask + 1u) * (INL ? 2u : 1u));
  u64& wg_base = *reinterpret_cast<u64*>(queues + (size_t)(kLdsBlock / 64) * a.wave_q);   // (behind the queues: the table starts at LDS address 0)
  u32& wg_count = reinterpret_cast<u32*>(&wg_base)[2];
  u32& wg_cursor = reinterpret_cast<u32*>(&wg_base)[3];
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
      if constexpr (BIG) {
        for (u32 e0 = 0; e0 < qn; e0 += 256) {       // (qn > 0 inside: the clamped entry exists)
          uint2 m[4]; bool live[4]; u32 v[4][4];
#pragma unroll
          for (u32 g = 0; g < 4; g++) {
            const u32 e = e0 + g * 64 + lane;
            live[g] = e < qn && base + e < a.out_cap;
            m[g] = wq[e < qn ? e : qn - 1];
          }
#pragma unroll
          for (u32 g = 0; g < 4; g++) {
#pragma unroll
            for (u32 u = 0; u < 4; u++) v[g][u] = src[u][from_build[u] ? m[g].x : m[g].y];   // (an unused column slot repeats column oc0)
          }
#pragma unroll
          for (u32 g = 0; g < 4; g++) {
            const u64 pos = base + e0 + g * 64 + lane;
#pragma unroll
            for (u32 u = 0; u < 4; u++) if (live[g] && on[u]) dst[u][pos] = v[g][u];
          }
        }
      } else {
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
    }
    if (a.visited) for (u32 e = lane; e < qn; e += 64) a.visited[wq[e].x] = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // vmcnt(0): without it the compiler carries "a gather above may still be in flight" around the probe loop and waits for ALL
    // memory operations — the prefetched next round included — in front of every slot read of every hop (this path is the rare one)
    __builtin_amdgcn_s_waitcnt(0x0F70);
  };
  auto resolve = [&]() {   // join filter over the queued candidates, survivors compacted in place
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (FSQ == 0) return;
    u32 kept = 0;
    if constexpr (BIG) {
      for (u32 g0 = 0; g0 < qn; g0 += 256) {
        uint2 m[4]; bool ok[4], slow[4];
#pragma unroll
        for (u32 g = 0; g < 4; g++) {
          const u32 e = g0 + g * 64 + lane;
          ok[g] = e < qn; slow[g] = false;
          m[g] = wq[e < qn ? e : qn - 1];
        }
#pragma unroll
        for (u32 g = 0; g < 4; g++) { bool sl; const bool f = ljoin_filter_fast<FS>(a, m[g].x, m[g].y, sl); slow[g] = ok[g] && sl; ok[g] = ok[g] && f; }   // (unconditional: the loads of all four groups leave together)
        if constexpr (FS == 1 || FS == 3) {
#pragma unroll
          for (u32 g = 0; g < 4; g++) if (slow[g]) ok[g] = ljoin_filter_slow<FS>(a, m[g].x, m[g].y);
        }
#pragma unroll
        for (u32 g = 0; g < 4; g++) {                // every read of this round is done: writing below g0 + 256 is safe
          const unsigned long long mask = __ballot(ok[g]);
          if (ok[g]) wq[kept + lane_prefix(mask)] = m[g];
          kept += (u32)__popcll(mask);
        }
      }
      qn = kept;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      return;
    }
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
  // BIG, writing pass: filter and write-out in ONE memory round trip.  The filter's operands and the candidates' output columns (the
  // first four) are requested together, for four 64-entry groups at once; the survivors' places come from the ballots and the
  // partition's cursor in LDS, their values go from registers to the output — no compaction through the queue, no second gather.
  // (The counting pass has told that the partition has survivors at all; a candidate the filter rejects costs its gathers.)
  auto write_fused = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const u32* src[4]; u32* dst[4]; bool from_build[4], on[4];
#pragma unroll
    for (u32 u = 0; u < 4; u++) {
      on[u] = u < a.n_out_cols;
      const u32 c = a.proj[on[u] ? u : 0u];
      from_build[u] = (c < a.n_left_cols) == (a.build_is_left != 0);
      src[u] = a.cols[c];
      dst[u] = a.out[on[u] ? u : 0u];
    }
    for (u32 g0 = 0; g0 < qn; g0 += 256) {         // (qn > 0 inside: the clamped entry exists)
      uint2 m[4]; bool ok[4], slow[4]; u32 v[4][4];
#pragma unroll
      for (u32 g = 0; g < 4; g++) {
        const u32 e = g0 + g * 64 + lane;
        ok[g] = e < qn; slow[g] = false;
        m[g] = wq[e < qn ? e : qn - 1];
      }
#pragma unroll
      for (u32 g = 0; g < 4; g++) {
#pragma unroll
        for (u32 u = 0; u < 4; u++) v[g][u] = src[u][from_build[u] ? m[g].x : m[g].y];   // (an unused column slot repeats column 0)
      }
      if constexpr (FSQ != 0) {
#pragma unroll
        for (u32 g = 0; g < 4; g++) { bool sl; const bool f = ljoin_filter_fast<FS>(a, m[g].x, m[g].y, sl); slow[g] = ok[g] && sl; ok[g] = ok[g] && f; }
        if constexpr (FS == 1 || FS == 3) {
#pragma unroll
          for (u32 g = 0; g < 4; g++) if (slow[g]) ok[g] = ljoin_filter_slow<FS>(a, m[g].x, m[g].y);
        }
      }
      unsigned long long mk[4]; u32 goff[4]; u32 n = 0;
#pragma unroll
      for (u32 g = 0; g < 4; g++) { mk[g] = __ballot(ok[g]); goff[g] = n; n += (u32)__popcll(mk[g]); }
      u32 off = 0;
      if (lane == 0 && n) off = atomicAdd(&wg_cursor, n);
      off = __shfl(off, 0, 64);
      u64 pos[4];
#pragma unroll
      for (u32 g = 0; g < 4; g++) { pos[g] = wg_base + off + goff[g] + lane_prefix(mk[g]); ok[g] = ok[g] && pos[g] < a.out_cap; }
#pragma unroll
      for (u32 g = 0; g < 4; g++) {
#pragma unroll
        for (u32 u = 0; u < 4; u++) if (ok[g] && on[u]) dst[u][pos[g]] = v[g][u];
      }
      for (u32 oc0 = 4; oc0 < a.n_out_cols; oc0 += 4) {   // further output columns: gathered for the survivors
#pragma unroll
        for (u32 u = 0; u < 4; u++) {
          if (oc0 + u >= a.n_out_cols) break;
          const u32 c = a.proj[oc0 + u];
          const bool fb = (c < a.n_left_cols) == (a.build_is_left != 0);
          const u32* sc = a.cols[c]; u32* dc = a.out[oc0 + u];
#pragma unroll
          for (u32 g = 0; g < 4; g++) if (ok[g]) dc[pos[g]] = sc[fb ? m[g].x : m[g].y];
        }
      }
      if (a.visited) {
#pragma unroll
        for (u32 g = 0; g < 4; g++) if (ok[g]) a.visited[m[g].x] = 1;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_s_waitcnt(0x0F70);
  };
  auto flush = [&](bool counting) {
    if constexpr (BIG) { if (!counting) { write_fused(); qn = 0; return; } }
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

  // The record loads are PREFETCHES (a round is requested while the round before it is worked on), so they are unconditional and
  // branch-free — the index is clamped into the partition, whether the lane has a row at all is decided when the record is used:
  // a load under a branch made the compiler wait for it where it was issued.
  const u32* pcol1x = a.n_keys > 1 ? pa.pcol1 : pa.pcol0;
  auto load_probe = [&](u32 ps, u32 pe, u32 t) -> uint4 {   // {probe row, key0, key1, -} of row min(ps + t, pe - 1)
    const u32 i = ps + t < pe ? ps + t : pe - 1;
    if constexpr (RANGE) return make_uint4(i, pa.pcol0[i], pcol1x[i], 0u);   // range mode: the slice itself, in place
    else { const PartRec q = pa.ppart[i]; return make_uint4(q.row, q.k0, q.k1, 0u); }
  };
  // One round of probe rows (one per lane) against the chunk's table.  The wave walks in LOCKSTEP and keeps no per-lane state but
  // its slot: a lane that has reached an empty slot (no further match) or a key-equal slot (a match) STAYS on it — reading it again
  // gives the same answer — until every lane has stopped; the only branch of a hop is the wave-uniform "is any lane still walking".
  // The lanes on a key-equal slot are queued and step past it, the next trip of the outer loop resumes there; a lane on an empty
  // slot stops at once in every later trip.  Lanes without a row (beyond the partition, null key) never walk.
  // The slot of a key: the top bits of two multiplicative hashes (4 vector instructions).  The partition took the top bits of the
  // general 4-round mix (part_keys_kernel) or an id range; this one only has to spread a partition's keys over its table.
  const u32 slot_shift = 32u - (u32)__builtin_popcount(pa.tbl_mask);
  auto slot_of = [&](u32 k0, u32 k1) -> u32 { return ((k0 * 0x9E3779B1u) ^ (k1 * 0x85EBCA77u)) >> slot_shift; };
  auto probe_round = [&](uint4 r, bool live, bool counting) {
    if (a.n_keys < 2) r.z = 0u;
    const u32 wrap = pa.tbl_mask << 3;
    u32 hb = slot_of(r.y, r.z) << 3;                 // the walk keeps the slot's byte offset
    const bool dead = !live | (r.x == kNil) | (r.y == 0u) | ((a.n_keys > 1) & (r.z == 0u));   // no row in this lane, or NullEqualsNothing
    const unsigned char* sb = reinterpret_cast<const unsigned char*>(slots);
    u32 pv = 0;
    if constexpr (INL) pv = pa.inl_probe[dead ? 0u : r.x];   // the probe row's filter operand, once per row
    for (;;) {
      bool eq;
      for (;;) {                                     // (divergent, but the body is one read, three compares and one exit)
        const uint2 c = *reinterpret_cast<const uint2*>(sb + hb);
        eq = (c.x == r.y) & (c.y == r.z);
        if ((c.x == 0u) | eq | dead) break;
        hb = (hb + 8u) & wrap;
      }
      if constexpr (INL) {
        const bool on_eq = eq & !dead;
        if (__builtin_amdgcn_ballot_w64(on_eq) == 0) break;
        const uint2 rf = rowfv[hb >> 3];             // (a lane that is not on a key-equal slot reads its slot's entry and ignores it)
        const bool pass = on_eq && rf.y != 0u && pv != 0u && ((rf.y == pv) == (a.idp.is_eq != 0));   // null => not `true` (ljoin_filter_fast<2>)
        const unsigned long long found = __builtin_amdgcn_ballot_w64(pass);
        const u32 n_found = (u32)__popcll(found);
        if (n_found) {
          if (qn + n_found > qcap) flush(counting);
          if (pass) wq[qn + lane_prefix(found)] = make_uint2(rf.x, r.x);
          qn += n_found;
        }
        if (on_eq) hb = (hb + 8u) & wrap;
        continue;
      }
      const bool hit = eq & !dead;                   // a live key is not 0: key-equal implies occupied
      const unsigned long long found = __builtin_amdgcn_ballot_w64(hit);
      if (found == 0) break;
      const u32 n_found = (u32)__popcll(found);
      if (qn + n_found > qcap) flush(counting);      // wave-uniform: the queue is empty afterwards and 64 <= qcap
      if (hit) { wq[qn + lane_prefix(found)] = make_uint2(rowof[hb >> 3], r.x); hb = (hb + 8u) & wrap; }
      qn += n_found;
    }
  };

  for (u32 p = blockIdx.x; p < pa.n_parts; p += gridDim.x) {
    const u32 bs = pa.bstart[p], be = pa.bstart[p + 1], ps = pa.pstart[p], pe = pa.pstart[p + 1];
    if (bs >= be || ps >= pe) continue;              // uniform per workgroup
    const u32 n_probe = pe - ps;
    __syncthreads();
    if (tid == 0) { wg_count = 0; wg_cursor = 0; }
    for (int pass = pa.two_pass ? 0 : 1; pass < 2; pass++) {
      const bool counting = pass == 0;
      for (u32 cb = bs; cb < be; cb += pa.chunk) {
        const u32 nb = be - cb < pa.chunk ? be - cb : pa.chunk;
        // the first round of both sides is requested before the table is cleared, every later round while the one before it is
        // worked on: the loads of a partition overlap its LDS phases instead of heading each of them
        PartRec bnext = pa.bpart[cb + (tid < nb ? tid : nb - 1)];
        uint4 pnext = load_probe(ps, pe, tid);
        // a partition of ONE chunk keeps its table from the counting pass to the writing pass (the probes do not change it): with
        // ~100 rows per key the build is the most expensive phase of the partition, see below
        const bool rebuild = counting || !pa.two_pass || be - bs > pa.chunk;   // uniform per workgroup
        if (rebuild) {
        __syncthreads();                             // every wave is done with the previous table
        for (u32 s = tid; s <= pa.tbl_mask; s += kLdsBlock) slots[s] = make_uint2(0u, 0u);
        __syncthreads();
        for (u32 i0 = 0; i0 < nb; i0 += kLdsBlock) {  // uniform trip count per workgroup
          const PartRec q = bnext;
          const u32 in = i0 + kLdsBlock + tid;
          bnext = pa.bpart[cb + (in < nb ? in : nb - 1)];
          const bool mine = i0 + tid < nb && q.row != kNil;   // else: no row in this lane, or a row that joins nothing
          const u32 qk1 = a.n_keys > 1 ? q.k1 : 0u;
          u32 h = slot_of(q.k0, qk1);
          if constexpr (BIG) {
            // Rows with EQUAL keys all start at the key's home slot and the loser of every CAS steps on by one: k rows of one key cost
            // k^2 / 2 serialised same-address atomics — 108 rows per key made the build 2.8 of the candidate join's 6.9 ms (knock-out
            // runs).  The equal-key lanes of a wave start on CONSECUTIVE slots instead, home + rank: every slot between home
            // and a row's final place is still tried by someone (the lane of that rank, which takes it or finds it taken), so the
            // chain a probe walks from home has no hole.
            u32 rank = 0;
            unsigned long long todo = __ballot(mine);
            while (todo) {                           // one trip per distinct key among the wave's rows
              const int leader = __builtin_ctzll(todo);
              const u32 lk0 = (u32)__builtin_amdgcn_readlane((int)q.k0, leader), lk1 = (u32)__builtin_amdgcn_readlane((int)qk1, leader);
              const unsigned long long same = __ballot(mine && q.k0 == lk0 && qk1 == lk1);
              if ((same >> lane) & 1ull) rank = (u32)__popcll(same & ((1ull << lane) - 1ull));
              todo &= ~same;
            }
            h = (h + rank) & pa.tbl_mask;
          }
          u32 fvb = 0;
          if constexpr (INL) fvb = pa.inl_build[mine ? q.row : 0u];
          if (!mine) continue;
          for (;;) {
            if (atomicCAS(&slots[h].x, 0u, q.k0) == 0u) {
              slots[h].y = q.k1;
              if constexpr (INL) rowfv[h] = make_uint2(q.row, fvb); else rowof[h] = q.row;
              break;
            }
            h = (h + 1) & pa.tbl_mask;
          }
        }
        }
        __syncthreads();
        for (u32 t0 = 0; t0 < n_probe; t0 += kLdsBlock) {   // uniform trip count per workgroup
          const uint4 r = pnext;
          pnext = load_probe(ps, pe, t0 + kLdsBlock + tid);
          probe_round(r, t0 + tid < n_probe, counting);
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
  return (size_t)(pa.tbl_mask + 1) * (sizeof(uint2) + (pa.inl_build ? sizeof(uint2) : sizeof(u32))) + (size_t)(kLdsBlock / 64) * a.wave_q * sizeof(uint2) + 16;
}
template <int FS, bool RANGE, bool BIG> static void launch_part_join_frb(const LdsJoinArgs& a, const PartArgs& pa, dim3 g, size_t lds, hipStream_t s) {
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    RDFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(part_join_kernel<FS, RANGE, BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
  });
  hipLaunchKernelGGL((part_join_kernel<FS, RANGE, BIG>), g, dim3(kLdsBlock), lds, s, a, pa);
}
template <int FS, bool RANGE> static void launch_part_join_fr(const LdsJoinArgs& a, const PartArgs& pa, dim3 g, size_t lds, hipStream_t s) {
  if constexpr (FS == 2) {
    if (pa.inl_build) {
      if (!pa.two_pass || !pa.inl_probe) fail(RDFGPU_ERR_INVALID, "partitioned join: the in-walk filter belongs to the two-pass form");
      static std::once_flag attr_once;
      std::call_once(attr_once, [] {
        RDFGPU_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(part_join_kernel<2, RANGE, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
      });
      hipLaunchKernelGGL((part_join_kernel<2, RANGE, true, true>), g, dim3(kLdsBlock), lds, s, a, pa);
      return;
    }
  }
  if (pa.two_pass) launch_part_join_frb<FS, RANGE, true>(a, pa, g, lds, s);
  else launch_part_join_frb<FS, RANGE, false>(a, pa, g, lds, s);
}
template <int FS> static void launch_part_join_fs(const LdsJoinArgs& a, const PartArgs& pa, dim3 g, size_t lds, hipStream_t s) {
  if (pa.ppart) launch_part_join_fr<FS, false>(a, pa, g, lds, s);
  else launch_part_join_fr<FS, true>(a, pa, g, lds, s);   // range mode: the probe side is a sorted slice read in place
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

// (kernels.hpp, preload_code_objects: the runtime loads a translation unit's code object at the first use of one of its kernels)
void preload_tu_part_join() { hipFuncAttributes at; RDFGPU_HIP(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(part_keys_kernel))); }
}  // namespace rdfgpu
