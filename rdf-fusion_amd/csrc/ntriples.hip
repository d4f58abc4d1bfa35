// ntriples.hip — bulk load, first half: N-Triples text -> object ids, on the device.
//
// The reference parses the file on the host and interns every term of every quad through a DashMap
// (Store::bulk_loader / load_from_reader, lib/rdf-fusion/src/store.rs:477-493 -> MemObjectIdMapping::encode_quad,
// lib/storage/src/memory/object_id_mapping.rs:106-116: three hash-map probes per triple, an insertion per new term).
// Here the per-TRIPLE work runs on the device:
//   nt_line_flags     which bytes start a triple line (not blank, not a `#` comment)           -> rocPRIM select = line starts
//   nt_terms          one lane per line: the three term spans exactly as written (`<iri>`, `_:b1`, `"lex"`, `"lex"@en`,
//                     `"lex"^^<dt>`; escapes inside a quoted string are skipped over, not rewritten), the closing `.`,
//                     and a 64-bit hash of each term's bytes
//   radix sort        (hash, term occurrence) pairs                                              -> equal terms adjacent
//   nt_unique         first occurrence of every distinct term; neighbours with one hash must be the same bytes (two
//                     different terms with one 64-bit hash fail the call loudly — it is not papered over)
//   scan + nt_assign  id = first_id + rank of the term among the distinct hashes; ids scattered into the s / p / o columns
//   nt_term_lengths / copy   the distinct terms, in id order, for the host dictionary
// What stays on the host is per DISTINCT term, not per triple: the dictionary (id -> term string, which this hands
// over) and the typed values of the new literals (rdfgpu_store_set_typed_values).  Ids are a bijection onto the distinct
// terms; which term gets which id differs from the reference's insertion order, which no query can observe.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <memory>
#include <vector>

#include "kernels.hpp"
#include "ntriples.hpp"

namespace rdfgpu {

namespace {

__device__ __forceinline__ bool nt_space(unsigned char c) { return c == ' ' || c == '\t' || c == '\r'; }

__global__ __launch_bounds__(256) void nt_line_flags_kernel(const unsigned char* text, u64 n, unsigned char* flag) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned char f = 0;
  if (i == 0 || text[i - 1] == '\n') {
    u64 j = i;
    while (j < n && nt_space(text[j])) j++;
    f = (j < n && text[j] != '\n' && text[j] != '#') ? 1 : 0;
  }
  flag[i] = f;
}

// error codes in err[0] (first error wins through atomicMin on the line number in err[1])
__device__ __forceinline__ void nt_fail(u64* err, u64 line, u32 code) {
  const unsigned long long packed = ((unsigned long long)line << 8) | code;
  atomicMin(reinterpret_cast<unsigned long long*>(err), packed);
}

__global__ __launch_bounds__(256) void nt_terms_kernel(const unsigned char* text, u64 n, const u64* line_start, u64 n_lines,
                                                        u64* span_off, u32* span_len, u64* hash, u32* occ, u64* err) {
  const u64 l = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_lines) return;
  u64 p = line_start[l];
  for (int k = 0; k < 3; k++) {
    while (p < n && nt_space(text[p])) p++;
    const u64 b = p;
    if (p >= n) { nt_fail(err, l, 1); return; }
    const unsigned char c = text[p];
    if (c == '<') {
      while (p < n && text[p] != '>' && text[p] != '\n') p++;
      if (p >= n || text[p] != '>') { nt_fail(err, l, 2); return; }
      p++;
    } else if (c == '_' && p + 1 < n && text[p + 1] == ':') {
      while (p < n && !nt_space(text[p]) && text[p] != '\n') p++;
      if (k == 2 && p > b + 2 && text[p - 1] == '.') p--;                     // `_:b1.` without a space before the dot
    } else if (c == '"' && k == 2) {
      p++;
      while (p < n && text[p] != '"' && text[p] != '\n') p += text[p] == '\\' ? 2 : 1;
      if (p >= n || text[p] != '"') { nt_fail(err, l, 3); return; }
      p++;
      if (p < n && text[p] == '@') { p++; while (p < n && (((text[p] | 32) >= 'a' && (text[p] | 32) <= 'z') || (text[p] >= '0' && text[p] <= '9') || text[p] == '-')) p++; }
      else if (p + 2 < n && text[p] == '^' && text[p + 1] == '^' && text[p + 2] == '<') {
        while (p < n && text[p] != '>' && text[p] != '\n') p++;
        if (p >= n || text[p] != '>') { nt_fail(err, l, 2); return; }
        p++;
      }
    } else { nt_fail(err, l, k == 2 ? 4 : 5); return; }                       // subject / predicate must be an IRI or a blank node
    unsigned long long h = 0xcbf29ce484222325ull;                             // FNV-1a over the term's bytes, then a finaliser
    for (u64 q = b; q < p; q++) { h ^= text[q]; h *= 0x100000001b3ull; }
    h ^= h >> 32; h *= 0xd6e8feb86659fd93ull; h ^= h >> 32;
    const u64 t = 3 * l + k;
    span_off[t] = b; span_len[t] = (u32)(p - b); hash[t] = h; occ[t] = (u32)t;
  }
  while (p < n && nt_space(text[p])) p++;
  if (p >= n || text[p] != '.') { nt_fail(err, l, 6); return; }
  p++;
  while (p < n && nt_space(text[p])) p++;
  if (p < n && text[p] != '\n' && text[p] != '#') nt_fail(err, l, 7);
}

// sorted by hash: first[j] = 1 when position j opens a new distinct term; same hash as the predecessor => same bytes, or fail
__global__ __launch_bounds__(256) void nt_unique_kernel(const unsigned char* text, const u64* shash, const u32* socc, u64 m,
                                                         const u64* span_off, const u32* span_len, u32* first, u64* err) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  u32 f = 1;
  if (j > 0 && shash[j] == shash[j - 1]) {
    f = 0;
    const u32 a = socc[j], b = socc[j - 1];
    bool same = span_len[a] == span_len[b];
    const unsigned char* x = text + span_off[a]; const unsigned char* y = text + span_off[b];
    for (u32 q = 0; same && q < span_len[a]; q++) same = x[q] == y[q];
    if (!same) nt_fail(err, (u64)a / 3, 8);
  }
  first[j] = f;
}
// rank[j] = (number of firsts in [0, j]) - 1: the term's index; ids into the s / p / o columns, the distinct terms' spans
__global__ __launch_bounds__(256) void nt_assign_kernel(const u32* socc, const u32* first, const u32* rank_incl, u64 m, u32 first_id,
                                                         u32* s, u32* p, u32* o, const u32* span_len, u32* term_occ, u32* term_len) {
  const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const u32 t = socc[j], rank = rank_incl[j] - 1;
  const u32 id = first_id + rank;
  const u64 line = t / 3; const u32 k = t % 3;
  (k == 0 ? s : k == 1 ? p : o)[line] = id;
  if (first[j]) { term_occ[rank] = t; term_len[rank] = span_len[t]; }
}
__global__ __launch_bounds__(256) void nt_term_bytes_kernel(const unsigned char* text, const u64* span_off, const u32* term_occ, const u32* term_len,
                                                             const u64* term_off, u64 n_terms, unsigned char* out) {
  const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  for (u32 r = 0; r < 16; r++) {
    const u64 t = wave * 16 + r;
    if (t >= n_terms) return;
    const unsigned char* src = text + span_off[term_occ[t]];
    unsigned char* dst = out + term_off[t];
    for (u32 b = lane; b < term_len[t]; b += 64) dst[b] = src[b];
  }
}

struct DevBuf {   // frees what a failed parse has allocated so far
  std::vector<void*> v;
  template <class T> T* get(u64 n) { void* p = nullptr; RDFGPU_HIP(hipMalloc(&p, (n ? n : 1) * sizeof(T))); v.push_back(p); return static_cast<T*>(p); }
  void release(void* p) { for (auto& q : v) if (q == p) q = nullptr; }
  ~DevBuf() { for (void* q : v) if (q) (void)hipFree(q); }
};
inline dim3 g256(u64 n) { return dim3((unsigned)((n + 255) / 256 ? (n + 255) / 256 : 1)); }
const char* const kNtErrors[] = {"", "line ends inside a term", "unterminated IRI", "unterminated string literal", "object is not an IRI, a blank node or a literal",
                                 "subject / predicate is not an IRI or a blank node", "missing `.` after the object", "text after the closing `.`",
                                 "two different terms share one 64-bit hash"};
}  // namespace

NTriples::~NTriples() {
  for (void* p : {(void*)s, (void*)p, (void*)o, (void*)term_off, (void*)term_bytes}) if (p) (void)hipFree(p);
}

NTriples* ntriples_parse(int device, const char* text, u64 n, u32 first_id) {
  if (device >= 0) RDFGPU_HIP(hipSetDevice(device));
  if (first_id == 0) fail(RDFGPU_ERR_INVALID, "object id 0 is the null marker: first_id must be at least 1");
  hipStream_t st = nullptr;   // the null stream: a load is not on the query path
  DevBuf buf;
  std::unique_ptr<NTriples> out(new NTriples());
  out->first_id = first_id;
  if (n == 0) return out.release();
  unsigned char* d_text = buf.get<unsigned char>(n + 16);
  RDFGPU_HIP(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, st));
  RDFGPU_HIP(hipMemsetAsync(d_text + n, '\n', 16, st));
  // ---- triple lines
  unsigned char* flag = buf.get<unsigned char>(n);
  hipLaunchKernelGGL(nt_line_flags_kernel, g256(n), dim3(256), 0, st, d_text, n, flag);
  u64* n_sel = buf.get<u64>(1);
  u64* line_start = nullptr;
  {
    u64* total = buf.get<u64>(1);                  // how many triple lines: sizes the list exactly
    size_t rb = 0;
    RDFGPU_HIP(rocprim::reduce(nullptr, rb, flag, total, (u64)0, (size_t)n, rocprim::plus<u64>(), st));
    void* rtemp = buf.get<unsigned char>(rb + 256);
    RDFGPU_HIP(rocprim::reduce(rtemp, rb, flag, total, (u64)0, (size_t)n, rocprim::plus<u64>(), st));
    u64 h_total = 0;
    RDFGPU_HIP(hipMemcpyAsync(&h_total, total, 8, hipMemcpyDeviceToHost, st));
    RDFGPU_HIP(hipStreamSynchronize(st));
    out->n_triples = h_total;
    line_start = buf.get<u64>(h_total);
    if (h_total) {
      size_t tb = 0;
      rocprim::counting_iterator<u64> iota(0);
      RDFGPU_HIP(rocprim::select(nullptr, tb, iota, flag, line_start, n_sel, (size_t)n, st));
      void* temp = buf.get<unsigned char>(tb + 256);
      RDFGPU_HIP(rocprim::select(temp, tb, iota, flag, line_start, n_sel, (size_t)n, st));
    }
  }
  const u64 L = out->n_triples;
  if (L == 0) return out.release();
  if (3 * L >= (1ull << 32)) fail(RDFGPU_ERR_UNSUPPORTED, "%llu triples in one call (term occurrences are 32-bit): load in pieces", (unsigned long long)L);
  const u64 m = 3 * L;
  // ---- terms
  u64* span_off = buf.get<u64>(m); u32* span_len = buf.get<u32>(m); u64* hash = buf.get<u64>(m); u32* occ = buf.get<u32>(m);
  u64* err = buf.get<u64>(1);
  RDFGPU_HIP(hipMemsetAsync(err, 0xFF, 8, st));
  hipLaunchKernelGGL(nt_terms_kernel, g256(L), dim3(256), 0, st, d_text, n, line_start, L, span_off, span_len, hash, occ, err);
  auto check = [&]() {
    u64 h_err = 0;
    RDFGPU_HIP(hipMemcpyAsync(&h_err, err, 8, hipMemcpyDeviceToHost, st));
    RDFGPU_HIP(hipStreamSynchronize(st));
    if (h_err != ~0ull) {
      const u32 code = (u32)(h_err & 0xFF);
      fail(code == 8 ? RDFGPU_ERR_UNSUPPORTED : RDFGPU_ERR_INVALID, "N-Triples, triple line %llu: %s", (unsigned long long)(h_err >> 8) + 1, kNtErrors[code < 9 ? code : 0]);
    }
  };
  check();
  // ---- distinct terms
  u64* shash = buf.get<u64>(m); u32* socc = buf.get<u32>(m);
  {
    const size_t tb = sort_temp_bytes(m);
    void* temp = buf.get<unsigned char>(tb);
    sort_pairs_u64_u32(hash, shash, occ, socc, m, temp, tb, st);
  }
  u32* first = buf.get<u32>(m); u32* rank = buf.get<u32>(m);
  hipLaunchKernelGGL(nt_unique_kernel, g256(m), dim3(256), 0, st, d_text, shash, socc, m, span_off, span_len, first, err);
  {
    const size_t tb = scan_temp_bytes(m);
    void* temp = buf.get<unsigned char>(tb);
    inclusive_scan_u32(first, rank, m, temp, tb, st);
  }
  check();
  u32 n_terms = 0;
  RDFGPU_HIP(hipMemcpyAsync(&n_terms, rank + (m - 1), 4, hipMemcpyDeviceToHost, st));
  RDFGPU_HIP(hipStreamSynchronize(st));
  if ((u64)first_id + n_terms > 0xFFFFFFFFull) fail(RDFGPU_ERR_UNSUPPORTED, "%u distinct terms from id %u on do not fit 32-bit object ids", n_terms, first_id);
  out->n_terms = n_terms;
  RDFGPU_HIP(hipMalloc((void**)&out->s, L * 4)); RDFGPU_HIP(hipMalloc((void**)&out->p, L * 4)); RDFGPU_HIP(hipMalloc((void**)&out->o, L * 4));
  u32* term_occ = buf.get<u32>(n_terms); u32* term_len = buf.get<u32>((u64)n_terms + 1);
  hipLaunchKernelGGL(nt_assign_kernel, g256(m), dim3(256), 0, st, socc, first, rank, m, first_id, out->s, out->p, out->o, span_len, term_occ, term_len);
  // ---- the distinct terms, packed in id order
  RDFGPU_HIP(hipMalloc((void**)&out->term_off, ((u64)n_terms + 1) * 8));
  {
    RDFGPU_HIP(hipMemsetAsync(term_len + n_terms, 0, 4, st));
    size_t tb = 0;
    RDFGPU_HIP(rocprim::exclusive_scan(nullptr, tb, term_len, out->term_off, (u64)0, (size_t)n_terms + 1, rocprim::plus<u64>(), st));
    void* temp = buf.get<unsigned char>(tb + 256);
    RDFGPU_HIP(rocprim::exclusive_scan(temp, tb, term_len, out->term_off, (u64)0, (size_t)n_terms + 1, rocprim::plus<u64>(), st));
  }
  u64 bytes = 0;
  RDFGPU_HIP(hipMemcpyAsync(&bytes, out->term_off + n_terms, 8, hipMemcpyDeviceToHost, st));
  RDFGPU_HIP(hipStreamSynchronize(st));
  out->term_total = bytes;
  RDFGPU_HIP(hipMalloc((void**)&out->term_bytes, bytes ? bytes : 1));
  {
    const u64 waves = ((u64)n_terms + 15) / 16;
    hipLaunchKernelGGL(nt_term_bytes_kernel, dim3((unsigned)((waves + 3) / 4 ? (waves + 3) / 4 : 1)), dim3(256), 0, st, d_text, span_off, term_occ, term_len, out->term_off, (u64)n_terms, out->term_bytes);
  }
  RDFGPU_HIP(hipStreamSynchronize(st));
  return out.release();
}

void ntriples_terms(const NTriples* t, u64* offsets, unsigned char* bytes) {
  if (offsets) {
    if (t->n_terms == 0 && !t->term_off) offsets[0] = 0;
    else RDFGPU_HIP(hipMemcpy(offsets, t->term_off, ((u64)t->n_terms + 1) * 8, hipMemcpyDeviceToHost));
  }
  if (bytes && t->term_total) RDFGPU_HIP(hipMemcpy(bytes, t->term_bytes, t->term_total, hipMemcpyDeviceToHost));
}

}  // namespace rdfgpu
