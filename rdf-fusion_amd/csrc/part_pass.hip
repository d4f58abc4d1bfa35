// part_pass.hip — the partition passes of the radix-partitioned hash join (part_join.hip), hand-written.
//
// What they replace: `part_keys_kernel` (per row: partition id + {row, key0, key1} record, 8 B read + 16 B written) followed by
// rocPRIM's radix sort of (partition id, record) pairs (histogram 4 B, two onesweep passes of 16 B read + 16 B written each):
// 92 bytes per row, 2.37 ms for the 98 M build rows of LUBM-8000 Q9's closing join.  Here the sort key never becomes a 4-byte column
// that travels with the records: the partition id (top bits of the key hash, or the id range of the probe slice's sort key) is
// computed ONCE, by pass A's histogram, and kept as 2 bytes per row until pass A's scatter has used it; that scatter leaves the
// low digit (1 byte) next to every record for pass B, whose histogram then reads ONE byte per row:
//
//   pass A  (hi = partition >> 8)   histogram: key columns 8 B -> id 2 B            scatter: key columns 8 B + id 2 B -> records 12 B + digit 1 B
//   pass B  (lo = partition & 255)  histogram: digit 1 B                            scatter: records 12 B + digit 1 B -> records 12 B
//
// 59 bytes per row, and the partition boundaries fall out of pass B's bin totals (no search).  A partition of <= 256 parts is pass A
// alone.  MSD order: pass B works inside the 256 buckets of pass A (its tiles never straddle a bucket), so a tile's rows go to <= 256
// neighbouring partitions and every (tile, bin) run lands next to the previous tile's.
// (First form: both passes recomputed the id from the keys, 64 B per row — in range mode that is a divergent 8-byte gather from the
// 32 KB directory and a 64-bit multiply, four times per row: the passes took 2.76 ms against rocPRIM's 2.37.  The directory is staged
// in LDS now, where a divergent read costs a few cycles instead of one lane per cycle.)
//
// One pass = histogram per 4096-row tile -> the counts of every bin scanned over its tiles (one workgroup per bin in pass A, one wave
// per (bucket, bin) in pass B: a flat device scan over the 6 - 14 M counts of the LUBM join took rocPRIM 0.3 - 0.4 ms, twice per
// side) + a scan of the <= 64 K bin totals, which in pass B ARE the partition starts -> scatter: the tile is
// ordered by bin in LDS first (rank inside the tile by an LDS atomic: the order of equal-bin rows is irrelevant to a hash join),
// then written out by consecutive lanes — runs of a bin are contiguous 12-byte records, not one 12-byte store per lane and line.
// No global atomics, deterministic placement of the runs.
//
// Rows that join nothing (null key, outside the probe slice's id range, beyond the live row count of a filtered input) are DROPPED by pass
// A: they have no record and no count.  (rocPRIM's sort had to place every element, so they rode in the last partition with row = kNil —
// and a FILTER's output that kept 313 K of 2.13 M rows left 1.8 M dead records there, which ONE workgroup then "joined" chunk by chunk
// against all of that partition's probe rows: LUBM Q9's OPTIONAL join took 11.5 ms instead of 3.)
//
// Reference behaviour is that of part_join.hip (HashJoinExec(CollectLeft), join/rewrite.rs:126-168): this file only decides where
// a row waits for its partition's workgroup.
#include <hip/hip_runtime.h>
#include <algorithm>

#include "join_device.hpp"

namespace rdfgpu {

constexpr u32 kPtBlock = 1024, kPtItems = 4, kPtTile = kPtBlock * kPtItems;   // rows per tile (4096: 59 KB of LDS staging in the scatter = two workgroups per CU, 32 waves)

// partition of a key pair; joins = false: the row joins nothing (null key, outside the probe slice's id range) and rides in the
// last partition with row = kNil
struct PartPid {
  u32 n_keys, bits, n_parts; PartKeyRange kr;
  // false: the row joins nothing (null key, outside the probe slice's id range) and rides in the last partition with row = kNil
  __device__ __forceinline__ bool joins(u32 k0, u32 k1) const {
    if (k0 == 0 || (n_keys > 1 && k1 == 0)) return false;              // NullEqualsNothing
    if (kr.range < 0) return true;
    const u32 k = kr.range == 0 ? k0 : k1;
    return k >= kr.range_min && k <= kr.range_max;
  }
  // the partition of a joining row; `dir` = the range directory (kr.dir or its copy in LDS)
  __device__ __forceinline__ u32 of(u32 k0, u32 k1, const uint2* dir) const {
    if (kr.range >= 0) {                                                 // range partitions of the probe slice's sort key
      const u32 k = kr.range == 0 ? k0 : k1;
      const u32 rel = k - kr.range_min, c = rel >> kr.cshift;
      const uint2 d = dir[c];
      return d.x + (u32)(((unsigned long long)(rel - (c << kr.cshift)) * d.y) >> kr.cshift);
    }
    Keys key; key.k[0] = k0; key.k[1] = k1; key.k[2] = 0; key.k[3] = 0;
    return bits ? hash_keys4(key, n_keys) >> (32 - bits) : 0u;
  }
};

struct PartTile { u32 first, n, hbase, stride, bucket; };   // records [first, first + n); counts at hist[hbase + bin * stride]; bucket of pass A (0 in pass A)
struct PartPassArgs {
  PartPid pid;
  // pass A: the key columns (row i of `cap`, live below *n_dev); pass B: the records of pass A
  const u32* k0; const u32* k1; const u64* n_dev; u64 cap;
  const PartRec* in;
  unsigned short* pid16;     // [cap] the partition of every row: written by pass A's histogram, read by its scatter
  unsigned char* digit;      // [cap] partition & 255 of every record after pass A, in its order: written by pass A's scatter, read by pass B
  // tiles: pass A has cap / 4096 of them, implicit; pass B reads its tiles and their number from the device (the launch covers the upper bound)
  const PartTile* tiles; const u32* n_tiles_dev; u32 n_tiles;
  u32 shift, nbins;          // pass A: bin of a row = partition >> shift; bins in use
  u32* hist;                 // counts per (bin, tile); then, scanned in place over the tiles of a bin: the rows of the bin in the tiles before
  const u32* base;           // first output record of a bin: base[bucket * 256 + bin]
  PartRec* out;
};

template <bool PASS_B>
__device__ __forceinline__ bool part_tile_of(const PartPassArgs& a, u32 g, PartTile& t) {
  if constexpr (PASS_B) {
    if (g >= *a.n_tiles_dev) return false;
    t = a.tiles[g];
  } else {
    const u64 first = (u64)g * kPtTile;
    const u64 left = a.cap - first;
    t = PartTile{(u32)first, (u32)(left < kPtTile ? left : kPtTile), g, a.n_tiles, 0u};
  }
  return true;
}

// One count per `valid` lane into cnt[bin]; returns the lane's rank = the count before + its place among the wave's lanes of the same
// bin.  Neighbouring rows often share their partition (a range-partitioned build side arrives clustered by key: LUBM's join output),
// and 64 lanes adding to ONE LDS word serialise — pass B's histogram took 276 us on such rows against 95 us on hashed ones.  So the
// lanes that share the first pending lane's bin are counted by that lane alone when they are a crowd (>= 16: one atomic for the group,
// ranks from the ballot; at most two such rounds), everything else goes one atomic per lane — on hashed rows the test is a ballot and
// a branch, no LDS round trip is added.  Every lane of the wave calls it.
__device__ __forceinline__ u32 wave_bin_add(u32* cnt, u32 bin, bool valid) {
  const u32 lane = threadIdx.x & 63u;
  unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
  u32 rank = 0;
  for (int round = 0; round < 2 && todo; round++) {                       // wave-uniform
    const int leader = __builtin_ctzll(todo);
    const u32 b = (u32)__builtin_amdgcn_readlane((int)bin, leader);
    const unsigned long long m = __builtin_amdgcn_ballot_w64(valid && bin == b) & todo;
    if (__popcll(m) < 16) break;                                          // no crowd (decided without touching LDS): one atomic per lane below
    u32 base = 0;
    if ((int)lane == leader) base = atomicAdd(&cnt[b], (u32)__popcll(m));
    base = (u32)__builtin_amdgcn_readlane((int)base, leader);
    if ((m >> lane) & 1ull) rank = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
    todo &= ~m;
  }
  if ((todo >> lane) & 1ull) rank = atomicAdd(&cnt[bin], 1u);
  return rank;
}

constexpr u32 kPartDirMax = 4096;   // entries of the range directory (plan.cpp: n_coarse <= 4096)
// Histogram workgroups are small (256 lanes x 16 rows of a tile) and PERSISTENT: a tile is 4 - 48 KB of input, a workgroup per tile
// lived for its launch and its two memory round trips (pass B, 1 byte per row: 276 us for 98 MB), and pass A's workgroups staged the
// 32 KB range directory once per tile — as many bytes as the tile's keys (444 us against 209 us in hash mode).
constexpr u32 kPhBlock = 256, kPhItems = kPtTile / kPhBlock;
template <bool PASS_B>
__global__ __launch_bounds__(kPhBlock) void part_hist_kernel(const PartPassArgs a) {
  __shared__ u32 cnt[256];
  __shared__ uint2 dir_l[PASS_B ? 1 : kPartDirMax];
  const u32 tid = threadIdx.x;
  const uint2* dir = a.pid.kr.dir;
  if constexpr (!PASS_B) {
    if (a.pid.kr.range >= 0 && a.pid.kr.n_coarse <= kPartDirMax) {       // the directory into LDS: every row reads one random entry of it
      for (u32 c = tid; c < a.pid.kr.n_coarse; c += kPhBlock) dir_l[c] = a.pid.kr.dir[c];
      dir = dir_l;
    }
  }
  const u32 n_tiles = PASS_B ? *a.n_tiles_dev : a.n_tiles;
  const u64 live = PASS_B ? 0 : live_rows(a.n_dev, a.cap);
  for (u32 g = blockIdx.x; g < n_tiles; g += gridDim.x) {                 // uniform per workgroup
    PartTile t;
    (void)part_tile_of<PASS_B>(a, g, t);
    cnt[tid] = 0u;
    __syncthreads();
    u32 bin[kPhItems]; bool live_item[kPhItems];
    if constexpr (PASS_B) {
#pragma unroll
      for (u32 it = 0; it < kPhItems; it++) {                             // (all loads of the tile in flight together, then the counting)
        const u32 j = it * kPhBlock + tid;
        bin[it] = j < t.n ? (u32)a.digit[t.first + j] : 0u;
        live_item[it] = true;                                             // (pass A kept only rows that join)
      }
    } else {
      u32 k0[kPhItems], k1[kPhItems];
#pragma unroll
      for (u32 it = 0; it < kPhItems; it++) {
        const u32 i = t.first + it * kPhBlock + tid;
        const bool in = it * kPhBlock + tid < t.n && i < live;
        k0[it] = in ? a.k0[i] : 0u; k1[it] = (in && a.pid.n_keys > 1) ? a.k1[i] : 0u;
      }
#pragma unroll
      for (u32 it = 0; it < kPhItems; it++) {
        const u32 j = it * kPhBlock + tid;
        live_item[it] = a.pid.joins(k0[it], k1[it]);       // (a row beyond the live count has key 0: joins nothing)
        const u32 p = live_item[it] ? a.pid.of(k0[it], k1[it], dir) : 0u;
        if (j < t.n) a.pid16[t.first + j] = (unsigned short)p;
        bin[it] = p >> a.shift;
      }
    }
#pragma unroll
    for (u32 it = 0; it < kPhItems; it++) (void)wave_bin_add(cnt, bin[it], it * kPhBlock + tid < t.n && live_item[it]);
    __syncthreads();
    if (tid < a.nbins) a.hist[(u64)t.hbase + (u64)tid * t.stride] = cnt[tid];   // (lane `tid` zeroes cnt[tid] itself at the top of the next tile)
  }
}

// The counts of one bin over its tiles -> exclusive prefix in place + the bin's total.
// Pass A: one workgroup of 1024 per bin (a row of cap / 4096 counts): every WAVE takes a contiguous sixteenth of the row and walks it
// 64 counts at a time (coalesced) — once for its sum, and, after the 16 sums have met in LDS, again (L2) writing the running prefix.
__global__ __launch_bounds__(1024) void part_row_scan_a_kernel(u32* hist, u32 tiles, u32 nbins, u32* total) {
  __shared__ u32 wt[16];
  const u32 bin = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32* row = hist + (u64)bin * tiles;
  const u32 seg = ((tiles + 15u) / 16u + 63u) & ~63u;                     // counts per wave, whole groups of 64
  const u32 s0 = wave * seg < tiles ? wave * seg : tiles, s1 = s0 + seg < tiles ? s0 + seg : tiles;
  u32 sum = 0;
  for (u32 i = s0 + lane; i < s1; i += 64) sum += row[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
  if (lane == 0) wt[wave] = sum;
  __syncthreads();
  u32 carry = 0, all = 0;
  for (u32 w = 0; w < 16; w++) { const u32 v = wt[w]; if (w < wave) carry += v; all += v; }
  for (u32 i0 = s0; i0 < s1; i0 += 64) {                                  // wave-uniform trip count
    const u32 i = i0 + lane;
    const u32 v = i < s1 ? row[i] : 0u;
    const u32 incl = wave_incl_scan(v);
    if (i < s1) row[i] = carry + incl - v;
    carry += (u32)__builtin_amdgcn_readlane((int)incl, 63);
  }
  if (threadIdx.x == 0) { total[bin] = all; if (bin == 0) total[nbins] = 0u; }   // (the extra element of the totals' scan: its exclusive prefix is the number of records)
}
// Pass B: one wave per (bucket, bin) — a row of the bucket's tiles (~ records of the bucket / 4096 counts, a hundred for the LUBM
// join); the totals, in (bucket, bin) order, are the partition sizes.
__global__ __launch_bounds__(256) void part_row_scan_b_kernel(u32* hist, const u32* tb, u32 nb, u32* total, u32* start_end, const u32* n_records) {
  const u32 lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) *start_end = *n_records;       // start[n_parts] = all records (pass A's total: the rows that join something); the totals' scan stops before it
  const u32 row_id = blockIdx.x * 4u + (threadIdx.x >> 6);                // bucket * 256 + bin
  if (row_id >= nb * 256u) return;                                        // (whole waves)
  const u32 b = row_id >> 8, bin = row_id & 255u;
  const u32 t0 = tb[b], ntb = tb[b + 1] - t0;
  u32* row = hist + 256ull * t0 + (u64)bin * ntb;
  u32 carry = 0;
  for (u32 i0 = 0; i0 < ntb; i0 += 64) {                                  // wave-uniform trip count
    const u32 i = i0 + lane;
    const u32 v = i < ntb ? row[i] : 0u;
    const u32 incl = wave_incl_scan(v);
    if (i < ntb) row[i] = carry + incl - v;
    carry += (u32)__builtin_amdgcn_readlane((int)incl, 63);
  }
  if (lane == 0) total[row_id] = carry;
}

template <bool PASS_B>
__global__ __launch_bounds__(kPtBlock) void part_scatter_kernel(const PartPassArgs a) {
  __shared__ u32 cnt[256], lstart[256], gofs[256], wt[4];
  __shared__ u32 s_row[kPtTile], s_k0[kPtTile], s_k1[kPtTile];
  __shared__ unsigned char s_bin[kPtTile];
  __shared__ unsigned char s_lo[PASS_B ? 1 : kPtTile];
  PartTile t;
  if (!part_tile_of<PASS_B>(a, blockIdx.x, t)) return;                   // uniform per workgroup
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 256) cnt[tid] = 0u;
  __syncthreads();
  PartRec rec[kPtItems]; u32 bin[kPtItems], rank[kPtItems], lo[kPtItems]; bool valid[kPtItems];
  if constexpr (PASS_B) {
#pragma unroll
    for (u32 it = 0; it < kPtItems; it++) {                               // (all loads of the tile in flight together)
      const u32 j = it * kPtBlock + tid;
      bin[it] = 0; lo[it] = 0; valid[it] = j < t.n;
      if (j < t.n) { rec[it] = a.in[t.first + j]; bin[it] = (u32)a.digit[t.first + j]; }
    }
  } else {
    const u64 live = live_rows(a.n_dev, a.cap);
#pragma unroll
    for (u32 it = 0; it < kPtItems; it++) {
      const u32 j = it * kPtBlock + tid, i = t.first + j;
      rec[it] = PartRec{kNil, 0u, 0u}; bin[it] = 0; lo[it] = 0; valid[it] = false;
      if (j < t.n) {
        const u32 p = a.pid16[i];
        if (i < live) { rec[it].k0 = a.k0[i]; rec[it].k1 = a.pid.n_keys > 1 ? a.k1[i] : 0u; }
        valid[it] = a.pid.joins(rec[it].k0, rec[it].k1);   // a row that joins nothing has no record (the histogram did not count it)
        rec[it].row = i;
        bin[it] = p >> a.shift; lo[it] = p & 255u;
      }
    }
  }
#pragma unroll
  for (u32 it = 0; it < kPtItems; it++) rank[it] = wave_bin_add(cnt, bin[it], valid[it]);   // the row's place among the tile's rows of its bin
  __syncthreads();
  u32 c = 0, incl = 0;
  if (tid < 256) {                                                        // waves 0 - 3, whole: exclusive scan of the 256 counts
    c = cnt[tid];
    incl = wave_incl_scan(c);
    if (lane == 63) wt[wave] = incl;
  }
  __syncthreads();
  if (tid < 256) {
    u32 base = 0;
    for (u32 w = 0; w < wave; w++) base += wt[w];
    lstart[tid] = base + incl - c;
    // where this tile's run of the bin starts in the output: the bin's first record + the bin's rows in the tiles before this one
    gofs[tid] = tid < a.nbins ? a.base[t.bucket * 256u + tid] + a.hist[(u64)t.hbase + (u64)tid * t.stride] : 0u;
  }
  __syncthreads();
#pragma unroll
  for (u32 it = 0; it < kPtItems; it++) {
    if (valid[it]) {
      const u32 p = lstart[bin[it]] + rank[it];
      s_row[p] = rec[it].row; s_k0[p] = rec[it].k0; s_k1[p] = rec[it].k1; s_bin[p] = (unsigned char)bin[it];
      if constexpr (!PASS_B) s_lo[p] = (unsigned char)lo[it];
    }
  }
  __syncthreads();
  const u32 n_valid = lstart[255] + cnt[255];                            // the tile's rows that have a record
  for (u32 j = tid; j < n_valid; j += kPtBlock) {                         // consecutive lanes -> consecutive records of a run
    const u32 b = s_bin[j];
    const u64 at = (u64)gofs[b] + (j - lstart[b]);
    a.out[at] = PartRec{s_row[j], s_k0[j], s_k1[j]};
    if constexpr (!PASS_B) { if (a.digit) a.digit[at] = s_lo[j]; }      // (two passes: the digit pass B will bin the record by)
  }
}

// Between the passes: the buckets of pass A (base_a[b] = first record of bucket b, base_a[nb] = all records) cut into pass B's tiles.
// tb[b] = first tile of bucket b (tb[nb] = all tiles): one workgroup.
__global__ __launch_bounds__(256) void part_bucket_tiles_kernel(const u32* base_a, u32 nb, u32* tb, u32* n_tiles_dev) {
  __shared__ u32 wt[4];
  const u32 b = threadIdx.x, lane = b & 63, wave = b >> 6;
  const u32 nt = b < nb ? (base_a[b + 1] - base_a[b] + kPtTile - 1) / kPtTile : 0u;
  const u32 incl = wave_incl_scan(nt);
  if (lane == 63) wt[wave] = incl;
  __syncthreads();
  u32 base = 0;
  for (u32 w = 0; w < wave; w++) base += wt[w];
  if (b < nb) tb[b] = base + incl - nt;
  if (b == 255) { tb[nb] = base + incl; *n_tiles_dev = base + incl; }
}
__global__ __launch_bounds__(256) void part_tile_desc_kernel(const u32* base_a, u32 nb, const u32* tb, PartTile* tiles) {
  const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= tb[nb]) return;
  u32 lo = 0, hi = nb;                                                     // the last bucket whose first tile is <= g (it has tiles: g < tb[nb])
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (tb[mid] <= g) lo = mid; else hi = mid; }
  const u32 b = lo, tin = g - tb[b], ntb = tb[b + 1] - tb[b];
  const u32 start = base_a[b] + tin * kPtTile, end = base_a[b + 1];
  tiles[g] = PartTile{start, end - start < kPtTile ? end - start : kPtTile, 256u * tb[b] + tin, ntb, b};
}

// ---- host side -------------------------------------------------------------------------------------------------------
// Sizes of the scratch arrays one side needs (elements): the caller allocates, so that the plan's allocator sees them.
PartPassPlan part_pass_plan(u64 n, u32 bits) {
  PartPassPlan p{};
  p.two = bits > 8;
  p.nb_a = p.two ? 1u << (bits - 8) : 1u << bits;
  p.tiles_a = (u32)((n + kPtTile - 1) / kPtTile);
  if (p.tiles_a == 0) p.tiles_a = 1;
  p.hist_a = (u64)p.nb_a * p.tiles_a;
  p.max_tiles_b = p.two ? p.tiles_a + p.nb_a + 1 : 0;
  p.hist_b = 256ull * p.max_tiles_b;
  p.tile_desc_bytes = (u64)p.max_tiles_b * sizeof(PartTile);
  return p;
}
void part_pass_run(const PartPassBuffers& w, const PartPassPlan& pl, const u32* k0, const u32* k1, u32 n_keys, const u64* n_dev, u64 cap, u32 bits, u32 n_parts,
                   PartKeyRange kr, void* scan_temp, size_t scan_temp_bytes, hipStream_t s) {
  if (!cap) return;
  if (cap >> 32) fail(RDFGPU_ERR_INVALID, "partitioned join: %llu rows on one side", (unsigned long long)cap);
  // pass A: bins = the top bits of the partition (all of them when there are <= 256 partitions: then base_a IS the partitions' starts)
  u32* base_a = pl.two ? w.base_a : w.start;
  PartPassArgs a{};
  a.pid = PartPid{n_keys, bits, n_parts, kr};
  a.k0 = k0; a.k1 = k1; a.n_dev = n_dev; a.cap = cap;
  a.n_tiles = pl.tiles_a; a.shift = pl.two ? 8u : 0u; a.nbins = pl.nb_a;
  a.hist = w.hist_a; a.base = base_a; a.out = pl.two ? w.recs_a : w.recs;
  a.pid16 = w.pid16; a.digit = pl.two ? w.digit : nullptr;
  hipLaunchKernelGGL(part_hist_kernel<false>, dim3(std::min<u32>(pl.tiles_a, 1024u)), dim3(kPhBlock), 0, s, a);   // (33 KB of LDS: four workgroups per CU)
  hipLaunchKernelGGL(part_row_scan_a_kernel, dim3(pl.nb_a), dim3(1024), 0, s, w.hist_a, pl.tiles_a, pl.nb_a, w.total);
  exclusive_scan_u32(w.total, base_a, (u64)pl.nb_a + 1, scan_temp, scan_temp_bytes, s);
  hipLaunchKernelGGL(part_scatter_kernel<false>, dim3(pl.tiles_a), dim3(kPtBlock), 0, s, a);
  if (!pl.two) return;
  hipLaunchKernelGGL(part_bucket_tiles_kernel, dim3(1), dim3(256), 0, s, base_a, pl.nb_a, w.tb, w.n_tiles_b);
  hipLaunchKernelGGL(part_tile_desc_kernel, dim3((pl.max_tiles_b + 255) / 256), dim3(256), 0, s, base_a, pl.nb_a, w.tb, reinterpret_cast<PartTile*>(w.tiles_b));
  PartPassArgs b = a;
  b.in = w.recs_a; b.tiles = reinterpret_cast<const PartTile*>(w.tiles_b); b.n_tiles_dev = w.n_tiles_b; b.n_tiles = pl.max_tiles_b;
  b.shift = 0; b.nbins = 256;
  b.hist = w.hist_b; b.base = w.start; b.out = w.recs;
  hipLaunchKernelGGL(part_hist_kernel<true>, dim3(std::min<u32>(pl.max_tiles_b, 2048u)), dim3(kPhBlock), 0, s, b);
  hipLaunchKernelGGL(part_row_scan_b_kernel, dim3((pl.nb_a * 256u + 3) / 4), dim3(256), 0, s, w.hist_b, w.tb, pl.nb_a, w.total, w.start + n_parts, base_a + pl.nb_a);
  // the (bucket, bin) totals are the partition sizes: their exclusive scan = the partitions' first records (n_parts <= 65536: one workgroup)
  exclusive_scan_u32(w.total, w.start, (u64)n_parts, scan_temp, scan_temp_bytes, s);
  hipLaunchKernelGGL(part_scatter_kernel<true>, dim3(pl.max_tiles_b), dim3(kPtBlock), 0, s, b);
}

// (kernels.hpp, preload_code_objects: the runtime loads a translation unit's code object at the first use of one of its kernels)
void preload_tu_part_pass() { hipFuncAttributes at; RDFGPU_HIP(hipFuncGetAttributes(&at, reinterpret_cast<const void*>((part_hist_kernel<false>)))); }
}  // namespace rdfgpu
