// stream_join.hip — HashJoinExec against a DIRECT-ADDRESS table (single unique dense key: row = direct[key - min]) as a streaming pass.
//
// Every probe row has at most one partner, so nothing has to be queued: a lane keeps its rows in registers from the key load to the
// output store.  The generic kernel (join_device.hpp) sends every candidate through a wave queue in LDS, filters it there and gathers
// the surviving rows' columns a second time; on the 0.54 G-row candidate table of the un-fused BSBM Q5 plan (hash_join.rs-style
// per-query builds, RDFGPU_OPT_NO_TABLE_CACHE) its waves sat waiting for 73 % of their cycles at four per SIMD — one tile's dependent
// chain key -> direct[] -> filter columns -> typed values -> reservation -> payload gathers -> stores per wave, 2 wave-instructions per
// row in flight (profiles/tools/nc_prof.sh).  Here a lane holds IT rows at once, all loads of a step are issued for all of them before
// the first is used (branch-free: a dead lane reads row 0 / the last row), the probe side's columns are read at the lane's OWN row
// (coalesced, not gathered), and the kernel needs no LDS beyond eight wave totals, so twice the waves fit a SIMD.
//
// One reservation of output rows per tile and workgroup (kSjBlock x IT rows): same-address atomics retire at ~88 per microsecond on
// this part, 4096-row tiles keep a 0.54 G-row probe side at 132 K of them.  Output order: by tile reservation, inside a tile by
// (wave, item, lane) — a multiset like every join of the engine.
// Reference behaviour: datafusion HashJoinExec(CollectLeft) as planned by lib/logical/src/join/rewrite.rs:126-168 (inner join, equi-key,
// optional residual filter); the filter semantics are join_device.hpp's (ljoin_filter_fast / ljoin_filter_slow), shared with the generic kernel.
#include "join_device.hpp"

namespace rdfgpu {

constexpr int kSjBlock = 512;

// VT (direct table, window filter FS = 3 whose x operand is one build column): the operand comes decoded from a.key_vals, by key.
template <int FS, int PFS, int IT, int MODE, bool VT = false>
__global__ __launch_bounds__(kSjBlock) __attribute__((amdgpu_waves_per_eu(4))) void stream_join_kernel(const LdsJoinArgs a) {
  static_assert(!VT || (FS == 3 && MODE == kJoinTableDirect), "value tables: direct-address joins with a window filter");
  constexpr bool HASH = MODE == kJoinTableHash;              // the {key, row} open-addressing table in HBM (gjoin_build_kernel); else the direct-address table
  __shared__ u32 wave_tot[kSjBlock / 64];
  __shared__ u64 wg_base;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64 np = live_rows(a.n_probe_dev, a.n_probe_cap);
  constexpr u32 kTile = (u32)kSjBlock * IT;
  const u64 n_tiles = (np + kTile - 1) / kTile;
  const u32* pk = a.probe_key[0];
  u32 nkey[HASH ? 1 : IT]; bool have_next = false;             // direct table: the next tile's keys, requested while this tile's output is reserved and written
  for (u64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {   // (np >= 1 inside)
    u64 j[IT]; u32 key[IT], b[IT]; bool ok[IT];
    u32 h[IT]; bool more[IT];                                  // HASH: where a row's walk stands; it stopped ON a further key-equal slot
    long long xv[VT ? IT : 1];                                 // VT: the window's x operand of the row's key (INT64_MIN: no row)
#pragma unroll
    for (int k = 0; k < IT; k++) {
      const u64 r = tile * kTile + (u32)k * kSjBlock + tid;
      ok[k] = r < np; j[k] = ok[k] ? r : np - 1;
      if constexpr (HASH) key[k] = pk[j[k]];
      else key[k] = have_next ? nkey[k] : pk[j[k]];            // (have_next is uniform)
    }
#pragma unroll
    for (int k = 0; k < IT; k++) {
      ok[k] = ok[k] && key[k] != 0;                            // NullEqualsNothing
      if constexpr (PFS != 0) ok[k] = lprobe_filter<PFS>(a, j[k]) && ok[k];   // the probe child's fused FilterExec
      more[k] = false;
      if constexpr (!HASH) {
        const u32 d = key[k] - a.direct_min;
        ok[k] = ok[k] && d < a.direct_n;                       // a key outside the table's range has no partner
        if constexpr (VT) {
          xv[k] = a.key_vals[ok[k] ? d : 0u];
          b[k] = a.stream_need_build_row ? a.direct[ok[k] ? d : 0u] : 0u;   // (wave-uniform; else nothing reads the build row)
        } else b[k] = a.direct[ok[k] ? d : 0u];
      }
    }
    if constexpr (HASH) {
      // Every row's chain is walked to its end in LOCKSTEP rounds — the IT slot reads of a round leave together — keeping the first key-equal
      // row.  A row that meets a SECOND one stops on it (more[k]): its further matches are emitted by the extra passes below, one per pass;
      // with unique build keys (a `?s <p> ?o` slice keyed by ?s) there are none and the tile is done in one pass.
      bool walking[IT];
      const uint2* slots = a.gslots;
#pragma unroll
      for (int k = 0; k < IT; k++) {
        Keys kk; kk.k[0] = key[k]; kk.k[1] = 0; kk.k[2] = 0; kk.k[3] = 0;
        h[k] = hash_keys4(kk, 1) & a.tbl_mask;
        walking[k] = ok[k]; b[k] = kNil;
      }
      for (;;) {
        uint2 c[IT];
#pragma unroll
        for (int k = 0; k < IT; k++) c[k] = slots[h[k]];       // (a row that has stopped reads its slot again: no branch around the load)
        bool any = false;
#pragma unroll
        for (int k = 0; k < IT; k++) {
          const u32 nx = (h[k] + 1) & a.tbl_mask;
          const bool eq = c[k].x == key[k], empty = c[k].y == kNil;
          const bool second = walking[k] && !empty && eq && b[k] != kNil;
          if (walking[k] && !empty && eq && b[k] == kNil) b[k] = c[k].y;
          more[k] = more[k] || second;
          walking[k] = walking[k] && !empty && !second;
          h[k] = walking[k] ? nx : h[k];
          any = any || walking[k];
        }
        if (!__any(any)) break;
      }
    }
#pragma unroll
    for (int k = 0; k < IT; k++) {
      if constexpr (VT) { ok[k] = ok[k] && xv[k] != INT64_MIN; b[k] = ok[k] && b[k] != kNil ? b[k] : 0u; }
      else { ok[k] = ok[k] && b[k] != kNil; b[k] = ok[k] ? b[k] : 0u; }
    }
    for (;;) {   // one trip per tile, plus one per further match of its most duplicated key (HASH)
      if (a.has_post) {   // wave-uniform: the former build-side FilterExec (`col <=|!=> literal`), one more conjunct
        const bool post_from_build = (a.post.col < a.n_left_cols) == (a.build_is_left != 0);
        const u32* pc = a.cols[a.post.col];
#pragma unroll
        for (int k = 0; k < IT; k++) {
          const u32 v = post_from_build ? pc[b[k]] : pc[j[k]];
          ok[k] = ok[k] && v != 0 && a.post.lit != 0 && ((v == a.post.lit) == (a.post.is_eq != 0));
        }
      }
      if constexpr (VT) {
        const WindowFilter& w = a.win;
        const bool same_y = w.y0 == w.y1;                        // wave-uniform
        const u32* cy0 = a.cols[w.y0]; const u32* cy1 = a.cols[w.y1];
        bool slow[IT];
#pragma unroll
        for (int k = 0; k < IT; k++) {
          const u32 iy0 = cy0[j[k]], iy1 = same_y ? iy0 : cy1[j[k]];   // the probe side's operands: the lane's own row
          bool sl;
          const bool f = window_fast_xy(a.tt, xv[k], iy0, iy1, same_y, w.l0, w.l1, sl);
          slow[k] = ok[k] && sl; ok[k] = ok[k] && f;
        }
#pragma unroll
        for (int k = 0; k < IT; k++) if (slow[k]) ok[k] = ljoin_filter_slow<FS>(a, a.direct[key[k] - a.direct_min], (u32)j[k]);   // (rare: a non-integer y operand)
      } else if constexpr (FS != 0) {
        bool slow[IT];
#pragma unroll
        for (int k = 0; k < IT; k++) {                           // unconditional: the operand gathers of all IT rows leave together
          bool sl;
          const bool f = ljoin_filter_fast<FS>(a, b[k], (u32)j[k], sl);
          slow[k] = ok[k] && sl; ok[k] = ok[k] && f;
        }
        if constexpr (FS == 3) {
#pragma unroll
          for (int k = 0; k < IT; k++) if (slow[k]) ok[k] = ljoin_filter_slow<FS>(a, b[k], (u32)j[k]);   // (rare: non-integer operands)
        }
      }
      // the first four output columns of every row are requested BEFORE the reservation (their addresses do not depend on it; a row the
      // filter rejected shares its cache lines with rows it kept), and so are the next tile's keys: both are in flight across the two
      // barriers and the atomic's round trip instead of queueing up behind them
      const u32* src0[4]; u32* dst0[4]; bool on0[4]; u32 v0[IT][4];
#pragma unroll
      for (u32 u = 0; u < 4; u++) {
        on0[u] = u < a.n_out_cols;
        const u32 c = a.proj[on0[u] ? u : 0u];
        const bool fb = (c < a.n_left_cols) == (a.build_is_left != 0);
        src0[u] = a.cols[c]; dst0[u] = a.out[on0[u] ? u : 0u];
#pragma unroll
        for (int k = 0; k < IT; k++) v0[k][u] = fb ? src0[u][b[k]] : src0[u][j[k]];   // (an unused column slot repeats column 0)
      }
      if constexpr (!HASH) {
        const u64 nt = tile + gridDim.x;
        have_next = nt < n_tiles;                                // uniform
        if (have_next) {
#pragma unroll
          for (int k = 0; k < IT; k++) { const u64 r = nt * kTile + (u32)k * kSjBlock + tid; nkey[k] = pk[r < np ? r : np - 1]; }
        }
      }
      // ---- positions: one reservation for the workgroup's tile; inside it (wave, item, lane) ----
      unsigned long long mk[IT]; u32 koff[IT]; u32 wtot = 0;
#pragma unroll
      for (int k = 0; k < IT; k++) { mk[k] = __ballot(ok[k]); koff[k] = wtot; wtot += (u32)__popcll(mk[k]); }
      if (lane == 0) wave_tot[wave] = wtot;
      __syncthreads();
      u32 woff = 0, total = 0;
#pragma unroll
      for (u32 w = 0; w < (u32)kSjBlock / 64; w++) { const u32 t = wave_tot[w]; woff += w < wave ? t : 0u; total += t; }
      if (tid == 0) {
        unsigned long long base = 0;
        if (total) {
          base = atomicAdd((unsigned long long*)a.n_out_dev, (unsigned long long)total);
          if (base + total > a.out_cap) *a.overflow = 1u;
        }
        wg_base = base;
      }
      __syncthreads();
      if (total != 0) {                                          // uniform per workgroup
        const u64 base = wg_base + woff;
        u64 pos[IT];
#pragma unroll
        for (int k = 0; k < IT; k++) { pos[k] = base + koff[k] + lane_prefix(mk[k]); ok[k] = ok[k] && pos[k] < a.out_cap; }
#pragma unroll
        for (int k = 0; k < IT; k++) {
#pragma unroll
          for (u32 u = 0; u < 4; u++) if (ok[k] && on0[u]) dst0[u][pos[k]] = v0[k][u];
        }
        // ---- further output columns: four at a time, the loads of a group of items before its stores ----
        for (u32 oc0 = 4; oc0 < a.n_out_cols; oc0 += 4) {
          const u32* src[4]; u32* dst[4]; bool from_build[4], on[4];
#pragma unroll
          for (u32 u = 0; u < 4; u++) {
            on[u] = oc0 + u < a.n_out_cols;
            const u32 c = a.proj[on[u] ? oc0 + u : oc0];
            from_build[u] = (c < a.n_left_cols) == (a.build_is_left != 0);
            src[u] = a.cols[c];
            dst[u] = a.out[on[u] ? oc0 + u : oc0];
          }
          constexpr int G = IT < 4 ? IT : 4;
#pragma unroll
          for (int k0 = 0; k0 < IT; k0 += G) {
            u32 v[G][4];
#pragma unroll
            for (int g = 0; g < G; g++) {
#pragma unroll
              for (u32 u = 0; u < 4; u++) v[g][u] = from_build[u] ? src[u][b[k0 + g]] : src[u][j[k0 + g]];   // (an unused column slot repeats column oc0)
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
#pragma unroll
              for (u32 u = 0; u < 4; u++) if (ok[k0 + g] && on[u]) dst[u][pos[k0 + g]] = v[g][u];
            }
          }
        }
      }
      if constexpr (!HASH) break;
      else {
        bool any_more = false;
#pragma unroll
        for (int k = 0; k < IT; k++) any_more = any_more || more[k];
        if (!__syncthreads_or(any_more ? 1 : 0)) break;          // uniform per workgroup (and wave_tot / wg_base are free again)
        // the rows standing on a further key-equal slot take it as their candidate and walk on to the next one (or to the chain's end)
        const uint2* slots = a.gslots;
#pragma unroll
        for (int k = 0; k < IT; k++) {
          ok[k] = more[k]; b[k] = 0u;
          if (more[k]) {
            b[k] = slots[h[k]].y;
            more[k] = false;
            for (;;) {
              h[k] = (h[k] + 1) & a.tbl_mask;
              const uint2 c = slots[h[k]];
              if (c.y == kNil) break;
              if (c.x == key[k]) { more[k] = true; break; }
            }
          }
        }
      }
    }
  }
}

// What the streaming form takes: a direct-address table with at least one entry, an inner join without a fused lookup chain, without
// the VM shapes (join filter FS 1, probe filter PFS 2: those stay with the generic kernel, which holds the expression VM once).
bool direct_stream_join_ok(const LdsJoinArgs& a) {
  const bool direct = a.direct != nullptr && a.direct_n > 0, hash = a.direct == nullptr && a.gslots != nullptr;
  return (direct || hash) && a.csr_off == nullptr && a.n_build_cap > 0 && a.n_keys == 1 && a.n_chain == 0 && a.visited == nullptr &&
         a.probe_outer == 0 && a.range_link == nullptr && a.has_filter != 1 && a.has_probe_filter != 2;
}
int direct_stream_join_items(u64 n_probe_cap) { return n_probe_cap >= (1ull << 20) ? 8 : 1; }
// (the hash form carries a walk position per row: four rows per lane keep it at five waves per SIMD)
static int stream_join_items(const LdsJoinArgs& a) { const int it = direct_stream_join_items(a.n_probe_cap); return it == 8 && a.direct == nullptr ? 4 : it; }

template <int FS, int PFS, int MODE> static void launch_sj_fpm(const LdsJoinArgs& a, int items, dim3 g, hipStream_t s) {
  if (items == 8) hipLaunchKernelGGL((stream_join_kernel<FS, PFS, 8, MODE>), g, dim3(kSjBlock), 0, s, a);
  else if (items == 4) hipLaunchKernelGGL((stream_join_kernel<FS, PFS, 4, MODE>), g, dim3(kSjBlock), 0, s, a);
  else hipLaunchKernelGGL((stream_join_kernel<FS, PFS, 1, MODE>), g, dim3(kSjBlock), 0, s, a);
}
template <int FS, int PFS> static void launch_sj_fp(const LdsJoinArgs& a, int items, dim3 g, hipStream_t s) {
  if (a.direct) launch_sj_fpm<FS, PFS, kJoinTableDirect>(a, items, g, s);
  else launch_sj_fpm<FS, PFS, kJoinTableHash>(a, items, g, s);
}
template <int FS> static void launch_sj_f(const LdsJoinArgs& a, int items, dim3 g, hipStream_t s) {
  if (a.has_probe_filter == 1) launch_sj_fp<FS, 1>(a, items, g, s);
  else launch_sj_fp<FS, 0>(a, items, g, s);
}
void launch_direct_stream_join(const LdsJoinArgs& a, hipStream_t s) {
  if (!direct_stream_join_ok(a)) fail(RDFGPU_ERR_INVALID, "streaming direct-table join: unsupported join shape");
  const int items = stream_join_items(a);
  const u64 rows = (u64)kSjBlock * items;
  const u64 n_tiles = (a.n_probe_cap + rows - 1) / rows;
  const dim3 g((unsigned)(n_tiles < 8192 ? (n_tiles ? n_tiles : 1) : 8192));   // every workgroup strides over the tiles
  if (a.key_vals != nullptr) {
    if (a.has_filter != 3 || a.has_probe_filter != 0 || items != 8 || a.direct == nullptr) fail(RDFGPU_ERR_INVALID, "streaming join: a value table on an unsupported join shape");
    hipLaunchKernelGGL((stream_join_kernel<3, 0, 8, kJoinTableDirect, true>), g, dim3(kSjBlock), 0, s, a);
    return;
  }
  switch (a.has_filter) {
    case 0: return launch_sj_f<0>(a, items, g, s);
    case 2: return launch_sj_f<2>(a, items, g, s);
    case 3: return launch_sj_f<3>(a, items, g, s);
  }
  fail(RDFGPU_ERR_INVALID, "streaming direct-table join: bad filter shape %u", a.has_filter);
}

// (kernels.hpp, preload_code_objects: the runtime loads a translation unit's code object at the first use of one of its kernels)
void preload_tu_stream_join() { hipFuncAttributes at; RDFGPU_HIP(hipFuncGetAttributes(&at, reinterpret_cast<const void*>((stream_join_kernel<0, 0, 1, kJoinTableDirect>)))); }
}  // namespace rdfgpu
