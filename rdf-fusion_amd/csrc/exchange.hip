// exchange.hip — the multi-GPU exchange steps of the path, behind the C ABI (include/rdfgpu.h section 6).
//
// The reference has no multi-process code; BASELINE's north star adds it: triples sharded by hash(subject)
// across the GPUs of one node, one process per GPU, intermediate bindings exchanged over RCCL / xGMI:
//   all-gatherv    every rank contributes a binding table of its own length, every rank receives all of them
//                  (the constant-subject bindings of a BSBM batch: each lives on one shard, all shards need them)
//   repartition    hash all-to-all: row i goes to rank shard(cols[key][i]) — the join key of the next join is not the
//                  shard key (LUBM Q9's triangle: student -> advisor -> course), so the smaller side is re-sharded by it
// Counts travel first (one small all-gather), buffers are sized from them — nothing is padded or clipped.  Payloads are
// grouped ncclSend / ncclRecv pairs: xGMI is point to point, every peer pair has its own link, a ring would serialise
// on one.  RCCL is loaded at run time (dlopen "librccl.so.1": the copy PyTorch already mapped, if any), so the library has
// no link-time dependency on it and a single-GPU process never touches it.
// A second transport stages through host memory and hands the wire to a caller-supplied all-to-all function: the same
// device-side packing and unpacking with the ranks of a rehearsal (several ranks on ONE GPU, where RCCL refuses to run)
// or of a test (gloo).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"
#include "exchange.hpp"

namespace rdfgpu {

// ---- RCCL, by name -------------------------------------------------------------------------------------------------
namespace {
struct NcclId { char internal[128]; };
typedef void* NcclComm;
enum { kNcclUint8 = 1, kNcclUint32 = 3, kNcclUint64 = 5 };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    auto sym = [&](const char* n) { return dlsym(r.lib, n); };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.AllGather)
    fail(RDFGPU_ERR_UNSUPPORTED, "RCCL is not available in this process (dlopen librccl.so.1: %s)", dlerror() ? dlerror() : "symbols missing");
  return r;
}
void nccl_check(int rc, const char* what) {
  if (rc != 0) fail(RDFGPU_ERR_DEVICE, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
}
}  // namespace

// ---- device side: rows -> destination ranks -------------------------------------------------------------------------
// shard(id) = the same multiplicative hash the host uses to shard triples by subject (sharding.shard_of): dense id ranges
// spread evenly.
__device__ __forceinline__ u32 shard_of(u32 id, u32 world) {
  const unsigned long long h = ((unsigned long long)id * 0x9E3779B97F4A7C15ull) >> 40;
  return (u32)(h % world);
}
constexpr int kMaxWorld = 64;
// The repartition is STABLE: rows of one destination keep their order (a table sorted by the key column leaves as sorted
// blocks, so the receiver gets `world` sorted runs — what the band join's partition pass counts on, and the same bytes on
// every run).  Every wave owns a contiguous piece of the table: pass 1 counts its rows per destination, a scan over the
// (wave, destination) matrix gives every wave its start inside each destination's block, pass 2 ranks the rows of a
// 64-row round with one ballot per destination.  No atomics, nothing depends on scheduling.
constexpr int kRepartWaves = 4;                    // waves per workgroup
__device__ __forceinline__ void repart_piece(u64 n, u64& lo, u64& hi) {   // rows of this wave
  const u64 waves = (u64)gridDim.x * kRepartWaves;
  const u64 per = ((n + waves - 1) / waves + 63) / 64 * 64;
  const u64 w = (u64)blockIdx.x * kRepartWaves + (threadIdx.x >> 6);
  lo = w * per < n ? w * per : n;
  hi = lo + per < n ? lo + per : n;
}
__global__ __launch_bounds__(256) void repart_count_kernel(const u32* key, u64 n, u32 world, u32* wave_counts) {
  __shared__ u32 h[kRepartWaves][kMaxWorld];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < kMaxWorld) h[wave][lane] = 0;
  __syncthreads();
  u64 lo, hi;
  repart_piece(n, lo, hi);
  for (u64 i = lo + lane; i < hi; i += 64) atomicAdd(&h[wave][shard_of(key[i], world)], 1u);
  __syncthreads();
  if ((u32)lane < world) wave_counts[((u64)blockIdx.x * kRepartWaves + wave) * world + lane] = h[wave][lane];
}
// one workgroup per destination: exclusive scan of its column of the matrix (in place), the total into counts[d]
__global__ __launch_bounds__(256) void repart_scan_kernel(u32* wave_counts, u64 n_waves, u32 world, unsigned long long* counts) {
  __shared__ u32 part[256];
  const u32 d = blockIdx.x;
  const u64 chunk = (n_waves + 255) / 256, lo = threadIdx.x * chunk, hi = lo + chunk < n_waves ? lo + chunk : n_waves;
  u32 sum = 0;
  for (u64 w = lo; w < hi; w++) sum += wave_counts[w * world + d];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = 0;
    for (int t = 0; t < 256; t++) { const u32 v = part[t]; part[t] = run; run += v; }
    counts[d] = run;
  }
  __syncthreads();
  u32 run = part[threadIdx.x];
  for (u64 w = lo; w < hi; w++) { const u32 v = wave_counts[w * world + d]; wave_counts[w * world + d] = run; run += v; }
}
// pass 2: out column c of destination d starts at out[c] + offset[d]
struct RepartArgs { const u32* in[kMaxCols]; u32* out[kMaxCols]; u32 n_cols, world, key_col, pad; u64 n; const u32* wave_base; const unsigned long long* offset; };
__global__ __launch_bounds__(256) void repart_scatter_kernel(const RepartArgs a) {
  __shared__ unsigned long long next[kRepartWaves][kMaxWorld];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if ((u32)lane < a.world) next[wave][lane] = a.offset[lane] + a.wave_base[((u64)blockIdx.x * kRepartWaves + wave) * a.world + lane];
  __syncthreads();                                  // (only so that the compiler keeps the LDS writes before the reads: a wave reads its own row)
  u64 lo, hi;
  repart_piece(a.n, lo, hi);
  for (u64 r = lo; r < hi; r += 64) {
    const u64 i = r + lane;
    const bool live = i < hi;
    const u32 d = live ? shard_of(a.in[a.key_col][i], a.world) : 0xFFFFFFFFu;
    u64 pos = 0;
    for (u32 t = 0; t < a.world; t++) {             // uniform loop: one ballot per destination
      const unsigned long long m = __ballot(d == t);
      if (m == 0) continue;
      const unsigned long long base = next[wave][t];
      if (d == t) pos = base + (u64)__popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) next[wave][t] = base + (u64)__popcll(m);
    }
    if (live) for (u32 c = 0; c < a.n_cols; c++) a.out[c][pos] = a.in[c][i];
  }
}

// ---- communicator -----------------------------------------------------------------------------------------------------
struct Comm {
  u32 rank = 0, world = 1;
  int device = 0;
  NcclComm nccl = nullptr;                       // RCCL transport
  rdfgpu_host_alltoallv_fn host_fn = nullptr;    // host-staged transport
  void* host_ctx = nullptr;
  hipStream_t stream = nullptr;
  // grow-only buffers owned by the communicator: outputs stay valid until the next exchange on this communicator
  std::vector<void*> dev; std::vector<size_t> dev_bytes;
  unsigned long long* counts_dev = nullptr;      // [world * world + 2 * world]
  void* buf(size_t slot, size_t bytes) {
    if (dev.size() <= slot) { dev.resize(slot + 1, nullptr); dev_bytes.resize(slot + 1, 0); }
    if (dev_bytes[slot] < bytes) {
      if (dev[slot]) RDFGPU_HIP(hipFree(dev[slot]));
      dev[slot] = nullptr; dev_bytes[slot] = 0;
      const size_t want = bytes + bytes / 4 + 256;
      RDFGPU_HIP(hipMalloc(&dev[slot], want));
      dev_bytes[slot] = want;
    }
    return dev[slot];
  }
  ~Comm() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (void* p : dev) if (p) (void)hipFree(p);
    if (counts_dev) (void)hipFree(counts_dev);
    if (nccl && rccl().CommDestroy) (void)rccl().CommDestroy(nccl);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

static Comm* comm_base(u32 rank, u32 world, int device) {
  if (world == 0 || rank >= world) fail(RDFGPU_ERR_INVALID, "communicator: rank %u of %u", rank, world);
  if (world > (u32)kMaxWorld) fail(RDFGPU_ERR_UNSUPPORTED, "communicator: %u ranks (max %d)", world, kMaxWorld);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); fail(RDFGPU_ERR_NO_DEVICE, "no usable HIP device"); }
  if (device < 0) RDFGPU_HIP(hipGetDevice(&device));
  if (device >= count) fail(RDFGPU_ERR_NO_DEVICE, "device %d does not exist (%d devices)", device, count);
  RDFGPU_HIP(hipSetDevice(device));
  Comm* c = new Comm();
  c->rank = rank; c->world = world; c->device = device;
  RDFGPU_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  RDFGPU_HIP(hipMalloc((void**)&c->counts_dev, ((size_t)world * world + 2 * world) * sizeof(unsigned long long)));
  return c;
}
void comm_unique_id(unsigned char id[128]) {
  NcclId nid;
  nccl_check(rccl().GetUniqueId(&nid), "ncclGetUniqueId");
  std::memcpy(id, nid.internal, 128);
}
Comm* comm_create_rccl(const unsigned char id[128], u32 rank, u32 world, int device) {
  Comm* c = comm_base(rank, world, device);
  try {
    NcclId nid; std::memcpy(nid.internal, id, 128);
    nccl_check(rccl().CommInitRank(&c->nccl, (int)world, nid, (int)rank), "ncclCommInitRank");
  } catch (...) { delete c; throw; }
  return c;
}
Comm* comm_create_host(u32 rank, u32 world, int device, rdfgpu_host_alltoallv_fn fn, void* ctx) {
  if (!fn) fail(RDFGPU_ERR_INVALID, "host transport without an all-to-all function");
  Comm* c = comm_base(rank, world, device);
  c->host_fn = fn; c->host_ctx = ctx;
  return c;
}
void comm_destroy(Comm* c) { delete c; }
u32 comm_rank(const Comm* c) { return c->rank; }
u32 comm_world(const Comm* c) { return c->world; }

// all-to-all of `send_rows[d]` rows per destination d out of column-major device send buffers (send_cols[c] + send_off[d]):
// counts first, then the payload.  Returns the received row count per source and fills out_cols (comm-owned).
static u64 alltoallv_tables(Comm* c, const u32* const* send_cols, u32 n_cols, const u64* send_rows, const u64* send_off,
                            const u32** out_cols, size_t out_slot0) {
  const u32 W = c->world;
  std::vector<u64> recv_rows(W, 0), recv_off(W + 1, 0);
  // ---- counts
  if (c->nccl) {
    unsigned long long* mine = c->counts_dev + (size_t)W * W;       // [W] my send counts
    RDFGPU_HIP(hipMemcpyAsync(mine, send_rows, W * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    nccl_check(rccl().AllGather(mine, c->counts_dev, W, kNcclUint64, c->nccl, c->stream), "ncclAllGather(counts)");
    std::vector<u64> all((size_t)W * W);
    RDFGPU_HIP(hipMemcpyAsync(all.data(), c->counts_dev, all.size() * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    RDFGPU_HIP(hipStreamSynchronize(c->stream));
    for (u32 r = 0; r < W; r++) recv_rows[r] = all[(size_t)r * W + c->rank];
  } else {
    std::vector<u64> sb(W, sizeof(u64)), rb(W, sizeof(u64));
    if (c->host_fn(c->host_ctx, send_rows, sb.data(), recv_rows.data(), rb.data()) != 0) fail(RDFGPU_ERR_DEVICE, "host all-to-all (counts) failed");
  }
  for (u32 r = 0; r < W; r++) recv_off[r + 1] = recv_off[r] + recv_rows[r];
  const u64 total = recv_off[W];
  for (u32 col = 0; col < n_cols; col++) out_cols[col] = static_cast<const u32*>(c->buf(out_slot0 + col, (total ? total : 1) * sizeof(u32)));
  // ---- payload
  if (c->nccl) {
    nccl_check(rccl().GroupStart(), "ncclGroupStart");
    for (u32 col = 0; col < n_cols; col++)
      for (u32 p = 0; p < W; p++) {
        if (send_rows[p]) nccl_check(rccl().Send(send_cols[col] + send_off[p], send_rows[p], kNcclUint32, (int)p, c->nccl, c->stream), "ncclSend");
        if (recv_rows[p]) nccl_check(rccl().Recv(const_cast<u32*>(out_cols[col]) + recv_off[p], recv_rows[p], kNcclUint32, (int)p, c->nccl, c->stream), "ncclRecv");
      }
    nccl_check(rccl().GroupEnd(), "ncclGroupEnd");
    RDFGPU_HIP(hipStreamSynchronize(c->stream));
  } else {
    // host staging: per destination one contiguous block [col 0 rows .. col n-1 rows]
    u64 send_total = 0;
    for (u32 p = 0; p < W; p++) send_total += send_rows[p];
    std::vector<u32> hs((size_t)(send_total ? send_total : 1) * n_cols), hr((size_t)(total ? total : 1) * n_cols);
    std::vector<u64> sb(W), rb(W), soff(W + 1, 0);
    for (u32 p = 0; p < W; p++) { sb[p] = send_rows[p] * n_cols * sizeof(u32); rb[p] = recv_rows[p] * n_cols * sizeof(u32); soff[p + 1] = soff[p] + send_rows[p] * n_cols; }
    for (u32 p = 0; p < W; p++)
      for (u32 col = 0; col < n_cols; col++)
        if (send_rows[p]) RDFGPU_HIP(hipMemcpyAsync(hs.data() + soff[p] + (size_t)col * send_rows[p], send_cols[col] + send_off[p], send_rows[p] * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    RDFGPU_HIP(hipStreamSynchronize(c->stream));
    if (c->host_fn(c->host_ctx, hs.data(), sb.data(), hr.data(), rb.data()) != 0) fail(RDFGPU_ERR_DEVICE, "host all-to-all (payload) failed");
    u64 at = 0;
    for (u32 p = 0; p < W; p++) {
      for (u32 col = 0; col < n_cols; col++)
        if (recv_rows[p]) RDFGPU_HIP(hipMemcpyAsync(const_cast<u32*>(out_cols[col]) + recv_off[p], hr.data() + at + (size_t)col * recv_rows[p], recv_rows[p] * sizeof(u32), hipMemcpyHostToDevice, c->stream));
      at += recv_rows[p] * n_cols;
    }
    RDFGPU_HIP(hipStreamSynchronize(c->stream));
  }
  return total;
}

u64 exchange_allgatherv(Comm* c, const u32* const* cols, u32 n_cols, u64 n_rows, const u32** out_cols) {
  if (n_cols == 0 || n_cols > (u32)kMaxCols) fail(RDFGPU_ERR_INVALID, "exchange of %u columns", n_cols);
  RDFGPU_HIP(hipSetDevice(c->device));
  const u32 W = c->world;
  // the same rows to every rank: W "destinations" all reading the same send range
  std::vector<u64> rows(W, n_rows), off(W, 0);
  return alltoallv_tables(c, cols, n_cols, rows.data(), off.data(), out_cols, 0);
}

u64 exchange_repartition(Comm* c, const u32* const* cols, u32 n_cols, u64 n_rows, u32 key_col, const u32** out_cols) {
  if (n_cols == 0 || n_cols > (u32)kMaxCols || key_col >= n_cols) fail(RDFGPU_ERR_INVALID, "repartition of %u columns by column %u", n_cols, key_col);
  RDFGPU_HIP(hipSetDevice(c->device));
  const u32 W = c->world;
  unsigned long long* counts = c->counts_dev + (size_t)W * W + W;   // [W] rows per destination, then their offsets in the send buffer
  const u64 g = n_rows ? std::min<u64>((n_rows + 256 * 16 - 1) / (256 * 16), 4096) : 1;
  const u64 n_waves = g * kRepartWaves;
  u32* wave_counts = static_cast<u32*>(c->buf(2 * kMaxCols, n_waves * W * sizeof(u32)));
  std::vector<u64> rows(W, 0), off(W + 1, 0);
  if (n_rows) {
    hipLaunchKernelGGL(repart_count_kernel, dim3((unsigned)g), dim3(256), 0, c->stream, cols[key_col], n_rows, W, wave_counts);
    hipLaunchKernelGGL(repart_scan_kernel, dim3(W), dim3(256), 0, c->stream, wave_counts, n_waves, W, counts);
    RDFGPU_HIP(hipMemcpyAsync(rows.data(), counts, W * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    RDFGPU_HIP(hipStreamSynchronize(c->stream));
  }
  for (u32 d = 0; d < W; d++) off[d + 1] = off[d] + rows[d];
  // rows grouped by destination (send side, comm-owned slots behind the output slots)
  RepartArgs a{};
  a.n_cols = n_cols; a.world = W; a.key_col = key_col; a.n = n_rows; a.wave_base = wave_counts; a.offset = counts;
  std::vector<const u32*> send(n_cols);
  for (u32 col = 0; col < n_cols; col++) { a.in[col] = cols[col]; a.out[col] = static_cast<u32*>(c->buf(kMaxCols + col, (n_rows ? n_rows : 1) * sizeof(u32))); send[col] = a.out[col]; }
  if (n_rows) {
    RDFGPU_HIP(hipMemcpyAsync(counts, off.data(), W * sizeof(u64), hipMemcpyHostToDevice, c->stream));   // where each destination's block starts
    hipLaunchKernelGGL(repart_scatter_kernel, dim3((unsigned)g), dim3(256), 0, c->stream, a);
  }
  RDFGPU_HIP(hipStreamSynchronize(c->stream));
  return alltoallv_tables(c, send.data(), n_cols, rows.data(), off.data(), out_cols, 0);
}

// (kernels.hpp, preload_code_objects: the runtime loads a translation unit's code object at the first use of one of its kernels)
void preload_tu_exchange() { hipFuncAttributes at; RDFGPU_HIP(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(repart_count_kernel))); }
}  // namespace rdfgpu
