// ntriples.hip — bulk load, first half: N-Triples text -> object ids, on the device.
//
// The reference parses the file on the host and interns every term of every quad through a DashMap
// (Store::bulk_loader / load_from_reader, lib/rdf-fusion/src/store.rs:477-493 -> MemObjectIdMapping::encode_quad,
// lib/storage/src/memory/object_id_mapping.rs:106-116: three hash-map probes per triple, an insertion per new term).
// Here the per-TRIPLE work runs on the device:
//   nt_line_flags     which bytes start a triple line (not blank, not a `#` comment)           -> rocPRIM select = line starts
//   nt_terms          one lane per line: the three term spans (`<iri>`, `_:b1`, `"lex"`, `"lex"@en`, `"lex"^^<dt>`), the closing
//                     `.`, and a 64-bit hash of each term's CANONICAL bytes: kind, the lexical form with its escapes decoded
//                     (ECHAR \t \b \n \r \f \" \' \\ and UCHAR \uXXXX / \UXXXXXXXX -> UTF-8, in literals and IRIs), the language
//                     tag in lower case or the datatype IRI (`^^xsd:string` = a simple literal) — what the reference's parser
//                     hands its interner (oxttl + oxrdf), so two spellings of one term get one id
//   radix sort        (hash, term occurrence) pairs                                              -> equal terms adjacent
//   nt_unique         first occurrence of every distinct term; neighbours with one hash must be the same bytes (two
//                     different terms with one 64-bit hash fail the call loudly — it is not papered over)
//   scan + nt_assign  id = first_id + rank of the term among the distinct hashes; ids scattered into the s / p / o columns
//   nt_term_lengths / copy   the distinct terms, in id order, for the host dictionary (one spelling each, as written)
//   nt_decode_lengths / nt_decode_bytes   per distinct term: kind, the DECODED lexical form, the language tag / datatype IRI, and the
//                     typed-value row of the literals the device can type exactly (encoding/typed_value.rs:27-83 + lib/model/src/
//                     xsd/*.rs FromStr): xsd:integer and its derived types, xsd:int, xsd:boolean, xsd:decimal (checked i128), and
//                     xsd:double / xsd:float when the decimal-to-binary conversion is exact in one IEEE operation (<= 15 / 7
//                     significant digits, |exponent| <= 22 / 10 — Clinger's fast path); anything else (long mantissas, dateTime,
//                     durations) is flagged RDFGPU_TVF_NEEDS_HOST for the host's parser, never rounded differently
// What stays on the host is per DISTINCT term, not per triple: the dictionary (id -> term string), the ranks of strings / IRIs in
// `str` order, language / datatype ids, and the flagged literals.  Ids are a bijection onto the distinct terms; which term gets
// which id differs from the reference's insertion order, which no query can observe.
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

// ---- canonical form of a term ----------------------------------------------------------------------------------------
// A term span as written -> kind + the sub-spans of its lexical part and its suffix (language tag / datatype IRI, without delimiters)
enum : u32 { NT_IRI = 1, NT_BNODE = 2, NT_SIMPLE = 3, NT_LANG = 4, NT_TYPED = 5 };
struct NtSplit { u32 kind; u64 lb, le, sb, se; };
__device__ __forceinline__ bool nt_eq_str(const unsigned char* t, u64 b, u64 e, const char* lit) {
  u64 k = 0;
  for (; lit[k]; k++) if (b + k >= e || t[b + k] != (unsigned char)lit[k]) return false;
  return b + k == e;
}
__device__ __forceinline__ NtSplit nt_split(const unsigned char* t, u64 b, u64 e) {
  NtSplit s{NT_IRI, b, e, e, e};
  if (t[b] == '<') { s.kind = NT_IRI; s.lb = b + 1; s.le = e - 1; return s; }
  if (t[b] == '_') { s.kind = NT_BNODE; s.lb = b + 2; s.le = e; return s; }
  u64 q = b + 1;                                              // a literal: the closing quote is the first unescaped one
  while (q < e && t[q] != '"') q += t[q] == '\\' ? 2 : 1;
  s.lb = b + 1; s.le = q;
  s.kind = NT_SIMPLE;
  if (q + 1 < e && t[q + 1] == '@') { s.kind = NT_LANG; s.sb = q + 2; s.se = e; }
  else if (q + 3 < e && t[q + 1] == '^') {
    s.sb = q + 4; s.se = e - 1;                               // ^^< ... >
    if (!nt_eq_str(t, s.sb, s.se, "http://www.w3.org/2001/XMLSchema#string")) s.kind = NT_TYPED;
    else { s.sb = s.se = e; }                                 // "x"^^xsd:string IS the simple literal "x" (RDF 1.1; oxrdf normalises alike)
  }
  return s;
}
// Decodes [b, e) — ECHAR / UCHAR escapes -> bytes — feeding every byte to `emit`; false on a malformed escape.
template <class F>
__device__ __forceinline__ bool nt_decode(const unsigned char* t, u64 b, u64 e, bool echar_ok, F&& emit) {
  for (u64 q = b; q < e;) {
    const unsigned char c = t[q];
    if (c != '\\') { emit(c); q++; continue; }
    if (q + 1 >= e) return false;
    const unsigned char x = t[q + 1];
    if (x == 'u' || x == 'U') {
      const u32 nd = x == 'u' ? 4u : 8u;
      if (q + 2 + nd > e) return false;
      u32 cp = 0;
      for (u32 k = 0; k < nd; k++) {
        const unsigned char h = t[q + 2 + k];
        const int d = h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : -1;
        if (d < 0) return false;
        cp = cp * 16u + (u32)d;
      }
      if (cp > 0x10FFFFu || (cp >= 0xD800u && cp <= 0xDFFFu)) return false;   // not a Unicode scalar value
      if (cp < 0x80u) emit((unsigned char)cp);
      else if (cp < 0x800u) { emit((unsigned char)(0xC0u | (cp >> 6))); emit((unsigned char)(0x80u | (cp & 63u))); }
      else if (cp < 0x10000u) { emit((unsigned char)(0xE0u | (cp >> 12))); emit((unsigned char)(0x80u | ((cp >> 6) & 63u))); emit((unsigned char)(0x80u | (cp & 63u))); }
      else { emit((unsigned char)(0xF0u | (cp >> 18))); emit((unsigned char)(0x80u | ((cp >> 12) & 63u))); emit((unsigned char)(0x80u | ((cp >> 6) & 63u))); emit((unsigned char)(0x80u | (cp & 63u))); }
      q += 2 + nd;
      continue;
    }
    if (!echar_ok) return false;                              // IRIs take UCHAR escapes only
    unsigned char v;
    switch (x) {
      case 't': v = '\t'; break; case 'b': v = '\b'; break; case 'n': v = '\n'; break; case 'r': v = '\r'; break; case 'f': v = '\f'; break;
      case '"': v = '"'; break; case '\'': v = '\''; break; case '\\': v = '\\'; break;
      default: return false;
    }
    emit(v); q += 2;
  }
  return true;
}
// every byte of the canonical form, in order: kind, decoded lexical form, 0xFF, suffix (language tag lower-cased / decoded datatype IRI)
template <class F>
__device__ __forceinline__ bool nt_canonical(const unsigned char* t, u64 b, u64 e, F&& emit) {
  const NtSplit s = nt_split(t, b, e);
  emit((unsigned char)s.kind);
  if (s.kind == NT_BNODE) { for (u64 q = s.lb; q < s.le; q++) emit(t[q]); return true; }
  if (!nt_decode(t, s.lb, s.le, s.kind != NT_IRI, emit)) return false;
  if (s.kind == NT_LANG) { emit(0xFF); for (u64 q = s.sb; q < s.se; q++) { const unsigned char c = t[q]; emit((unsigned char)(c >= 'A' && c <= 'Z' ? c + 32 : c)); } }
  if (s.kind == NT_TYPED) { emit(0xFF); if (!nt_decode(t, s.sb, s.se, false, emit)) return false; }
  return true;
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
    unsigned long long h = 0xcbf29ce484222325ull;                             // FNV-1a over the term's CANONICAL bytes, then a finaliser
    if (!nt_canonical(text, b, p, [&](unsigned char ch) { h ^= ch; h *= 0x100000001b3ull; })) { nt_fail(err, l, 9); return; }
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
    if (!same) {   // two spellings: the same term iff their canonical forms agree — compared through a second, independent 64-bit mix
                   // and their lengths (the canonical bytes are produced by a callback, not by an iterator that two could walk in step)
      unsigned long long ha = 0x9E3779B97F4A7C15ull, hb = 0x9E3779B97F4A7C15ull; u64 na = 0, nb = 0;
      nt_canonical(text, span_off[a], span_off[a] + span_len[a], [&](unsigned char ch) { ha = (ha ^ ch) * 0xff51afd7ed558ccdull; ha ^= ha >> 29; na++; });
      nt_canonical(text, span_off[b], span_off[b] + span_len[b], [&](unsigned char ch) { hb = (hb ^ ch) * 0xff51afd7ed558ccdull; hb ^= hb >> 29; nb++; });
      if (ha != hb || na != nb) nt_fail(err, (u64)a / 3, 8);
    }
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

// ---- per distinct term: decoded lexical form, suffix, typed value --------------------------------------------------------
__global__ __launch_bounds__(256) void nt_decode_lengths_kernel(const unsigned char* text, const u64* span_off, const u32* span_len, const u32* term_occ, u64 n_terms,
                                                                 unsigned char* kind, u32* lex_len, u32* sfx_len) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  const u64 b = span_off[term_occ[t]], e = b + span_len[term_occ[t]];
  const NtSplit s = nt_split(text, b, e);
  u32 nl = 0, ns = 0;
  if (s.kind == NT_BNODE) nl = (u32)(s.le - s.lb);
  else nt_decode(text, s.lb, s.le, s.kind != NT_IRI, [&](unsigned char) { nl++; });
  if (s.kind == NT_LANG) ns = (u32)(s.se - s.sb);
  if (s.kind == NT_TYPED) nt_decode(text, s.sb, s.se, false, [&](unsigned char) { ns++; });
  kind[t] = (unsigned char)s.kind; lex_len[t] = nl; sfx_len[t] = ns;
}
// lib/model/src/xsd FromStr restated on bytes.  i64::from_str / i32::from_str: an optional sign, at least one digit, nothing else.
__device__ __forceinline__ bool nt_parse_i64(const unsigned char* p, u32 n, long long lo_bound, long long hi_bound, long long& out) {
  u32 i = 0; bool neg = false;
  if (n && (p[0] == '+' || p[0] == '-')) { neg = p[0] == '-'; i = 1; }
  if (i >= n) return false;
  long long v = 0;
  for (; i < n; i++) {
    if (p[i] < '0' || p[i] > '9') return false;
    const long long d = p[i] - '0';
    if (__builtin_mul_overflow(v, 10ll, &v)) return false;
    if (neg ? __builtin_sub_overflow(v, d, &v) : __builtin_add_overflow(v, d, &v)) return false;
  }
  if (v < lo_bound || v > hi_bound) return false;
  out = v; return true;
}
// Decimal::from_str, decimal.rs:496-568: (\+|-)?([0-9]+(\.[0-9]*)?|\.[0-9]+), value * 10^18 in a checked i128
__device__ __forceinline__ bool nt_parse_decimal(const unsigned char* p, u32 n, __int128& out) {
  if (n == 0) return false;
  u32 i = 0; __int128 sign = 1;
  if (p[0] == '+') i = 1; else if (p[0] == '-') { sign = -1; i = 1; }
  __int128 v = 0;
  const bool before = i < n && p[i] >= '0' && p[i] <= '9';
  for (; i < n && p[i] >= '0' && p[i] <= '9'; i++)
    if (__builtin_mul_overflow(v, (__int128)10, &v) || __builtin_add_overflow(v, sign * (__int128)(p[i] - '0'), &v)) return false;
  __int128 exp = (__int128)1000000000000000000ll;
  if (i < n) {
    if (p[i] != '.') return false;
    i++;
    if (i >= n && !before) return false;                       // only a dot
    u32 end = n;
    while (end > i && p[end - 1] == '0') end--;                 // trailing zeros carry nothing ("hack to avoid underflows")
    for (; i < end; i++) {
      if (p[i] < '0' || p[i] > '9') return false;
      exp /= 10;
      if (__builtin_mul_overflow(v, (__int128)10, &v) || __builtin_add_overflow(v, sign * (__int128)(p[i] - '0'), &v)) return false;
    }
    if (exp == 0) return false;                                 // more than 18 fractional digits: underflow
  } else if (!before) return false;
  return !__builtin_mul_overflow(v, exp, &out);
}
// f64::from_str / f32::from_str when the conversion is EXACT in one IEEE operation (Clinger's fast path): 1 = parsed, 0 = not a
// float by the grammar (the caller lets the HOST decide: Rust's grammar has corners — "infinity", "1e", "1." — that are not restated),
// likewise everything off the fast path.
__device__ __forceinline__ bool nt_parse_float_fast(const unsigned char* p, u32 n, bool single, double& out) {
  u32 i = 0; bool neg = false;
  if (n && (p[0] == '+' || p[0] == '-')) { neg = p[0] == '-'; i = 1; }
  if (i >= n) return false;
  auto lc = [&](u32 k) { const unsigned char c = p[k]; return (unsigned char)(c >= 'A' && c <= 'Z' ? c + 32 : c); };
  if (n - i == 3 && lc(i) == 'i' && lc(i + 1) == 'n' && lc(i + 2) == 'f') { out = neg ? -__builtin_inf() : __builtin_inf(); return true; }
  if (n - i == 3 && lc(i) == 'n' && lc(i + 1) == 'a' && lc(i + 2) == 'n') { out = __builtin_nan(""); return !neg && i == 0; }   // (a signed NaN: the host's call)
  unsigned long long m = 0; int digits = 0, scale = 0; bool any = false;
  for (; i < n && p[i] >= '0' && p[i] <= '9'; i++) { any = true; if (m || p[i] != '0') { if (++digits > 19) return false; m = m * 10 + (p[i] - '0'); } }
  if (i < n && p[i] == '.') {
    i++;
    for (; i < n && p[i] >= '0' && p[i] <= '9'; i++) { any = true; scale--; if (m || p[i] != '0') { if (++digits > 19) return false; m = m * 10 + (p[i] - '0'); } }
  }
  if (!any) return false;
  if (i < n) {
    if (p[i] != 'e' && p[i] != 'E') return false;
    i++;
    bool eneg = false;
    if (i < n && (p[i] == '+' || p[i] == '-')) { eneg = p[i] == '-'; i++; }
    if (i >= n) return false;
    int ex = 0;
    for (; i < n; i++) { if (p[i] < '0' || p[i] > '9') return false; ex = ex * 10 + (p[i] - '0'); if (ex > 10000) return false; }
    scale += eneg ? -ex : ex;
  }
  if (m == 0) { out = neg ? -0.0 : 0.0; return true; }
  static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  if (single) {
    if (m >= (1ull << 24) || scale < -10 || scale > 10) return false;           // mantissa and power of ten exact in f32: one correctly rounded operation
    const float mf = (float)m, pf = (float)p10[scale < 0 ? -scale : scale];
    const float r = scale < 0 ? __fdiv_rn(mf, pf) : __fmul_rn(mf, pf);
    out = (double)(neg ? -r : r); return true;
  }
  if (m >= (1ull << 53) || scale < -22 || scale > 22) return false;
  const double md = (double)m;
  const double r = scale < 0 ? __ddiv_rn(md, p10[-scale]) : __dmul_rn(md, p10[scale]);
  out = neg ? -r : r; return true;
}
constexpr unsigned char kNeedsHost = 0x80;   // = RDFGPU_TVF_NEEDS_HOST
__global__ __launch_bounds__(256) void nt_decode_bytes_kernel(const unsigned char* text, const u64* span_off, const u32* span_len, const u32* term_occ, u64 n_terms,
                                                               const u64* lex_off, unsigned char* lex, const u64* sfx_off, unsigned char* sfx,
                                                               rdfgpu_typed_value* typed, long long* dec_hi) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  const u64 b = span_off[term_occ[t]], e = b + span_len[term_occ[t]];
  const NtSplit s = nt_split(text, b, e);
  unsigned char* lp = lex + lex_off[t]; unsigned char* sp = sfx + sfx_off[t];
  u32 nl = 0, ns = 0;
  if (s.kind == NT_BNODE) for (u64 q = s.lb; q < s.le; q++) lp[nl++] = text[q];
  else nt_decode(text, s.lb, s.le, s.kind != NT_IRI, [&](unsigned char c) { lp[nl++] = c; });
  if (s.kind == NT_LANG) for (u64 q = s.sb; q < s.se; q++) { const unsigned char c = text[q]; sp[ns++] = (unsigned char)(c >= 'A' && c <= 'Z' ? c + 32 : c); }
  if (s.kind == NT_TYPED) nt_decode(text, s.sb, s.se, false, [&](unsigned char c) { sp[ns++] = c; });
  rdfgpu_typed_value v{};   // tag NULL
  long long hi = 0;
  if (s.kind == NT_IRI) v.tag = RDFGPU_TV_NAMED_NODE;
  else if (s.kind == NT_BNODE) v.tag = RDFGPU_TV_BLANK_NODE;
  else if (s.kind == NT_SIMPLE || s.kind == NT_LANG) { v.tag = RDFGPU_TV_STRING; v.flags = nl == 0 ? RDFGPU_TVF_EMPTY_STRING : 0; }
  else {
    // typed_value.rs:349-411: which datatype parses as what; a value its parser rejects is Invalid = the null typed value
    static const char kXsd[] = "http://www.w3.org/2001/XMLSchema#";
    bool xsd = ns > sizeof(kXsd) - 1;
    for (u32 k = 0; xsd && k < sizeof(kXsd) - 1; k++) xsd = sp[k] == (unsigned char)kXsd[k];
    const unsigned char* nm = sp + (sizeof(kXsd) - 1); const u32 nn = xsd ? ns - (u32)(sizeof(kXsd) - 1) : 0;
    auto is = [&](const char* name) { u32 k = 0; for (; name[k]; k++) if (k >= nn || nm[k] != (unsigned char)name[k]) return false; return k == nn; };
    v.tag = RDFGPU_TV_OTHER;                                                    // any other datatype: an opaque literal (the host numbers the datatype)
    if (xsd) {
      long long iv; __int128 dv; double fv;
      if (is("integer") || is("byte") || is("short") || is("long") || is("unsignedByte") || is("unsignedShort") || is("unsignedInt") || is("unsignedLong") ||
          is("positiveInteger") || is("negativeInteger") || is("nonPositiveInteger") || is("nonNegativeInteger")) {
        if (nt_parse_i64(lp, nl, INT64_MIN, INT64_MAX, iv)) { v.tag = RDFGPU_TV_INTEGER; v.lo = iv; } else v.tag = RDFGPU_TV_NULL;
      } else if (is("int")) {
        if (nt_parse_i64(lp, nl, INT32_MIN, INT32_MAX, iv)) { v.tag = RDFGPU_TV_INT; v.lo = iv; } else v.tag = RDFGPU_TV_NULL;
      } else if (is("boolean")) {
        const bool tr = (nl == 4 && lp[0] == 't' && lp[1] == 'r' && lp[2] == 'u' && lp[3] == 'e') || (nl == 1 && lp[0] == '1');
        const bool fa = (nl == 5 && lp[0] == 'f' && lp[1] == 'a' && lp[2] == 'l' && lp[3] == 's' && lp[4] == 'e') || (nl == 1 && lp[0] == '0');
        if (tr || fa) { v.tag = RDFGPU_TV_BOOLEAN; v.lo = tr ? 1 : 0; } else v.tag = RDFGPU_TV_NULL;
      } else if (is("decimal")) {
        if (nt_parse_decimal(lp, nl, dv)) { v.tag = RDFGPU_TV_DECIMAL; v.lo = (long long)(unsigned long long)(unsigned __int128)dv; hi = (long long)(unsigned long long)((unsigned __int128)dv >> 64); }
        else v.tag = RDFGPU_TV_NULL;
      } else if (is("double")) {
        if (nt_parse_float_fast(lp, nl, false, fv)) { v.tag = RDFGPU_TV_DOUBLE; v.lo = __double_as_longlong(fv); } else { v.tag = RDFGPU_TV_DOUBLE; v.flags = kNeedsHost; }
      } else if (is("float")) {
        if (nt_parse_float_fast(lp, nl, true, fv)) { v.tag = RDFGPU_TV_FLOAT; v.lo = (long long)(unsigned long long)__float_as_uint((float)fv); } else { v.tag = RDFGPU_TV_FLOAT; v.flags = kNeedsHost; }
      } else if (is("dateTime")) { v.tag = RDFGPU_TV_DATE_TIME; v.flags = kNeedsHost; }
      else if (is("time")) { v.tag = RDFGPU_TV_TIME; v.flags = kNeedsHost; }
      else if (is("date")) { v.tag = RDFGPU_TV_DATE; v.flags = kNeedsHost; }
      else if (is("duration") || is("yearMonthDuration") || is("dayTimeDuration")) { v.tag = RDFGPU_TV_DURATION; v.flags = kNeedsHost; }
    }
  }
  typed[t] = v; dec_hi[t] = hi;
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
                                 "two different terms share one 64-bit hash", "malformed escape sequence (ECHAR / UCHAR)"};
}  // namespace

NTriples::~NTriples() {
  for (void* p : {(void*)s, (void*)p, (void*)o, (void*)term_off, (void*)term_bytes, (void*)kind, (void*)lex_off, (void*)lex, (void*)sfx_off, (void*)sfx, (void*)typed, (void*)dec_hi}) if (p) (void)hipFree(p);
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
      fail(code == 8 ? RDFGPU_ERR_UNSUPPORTED : RDFGPU_ERR_INVALID, "N-Triples, triple line %llu: %s", (unsigned long long)(h_err >> 8) + 1, kNtErrors[code < 10 ? code : 0]);
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
  // ---- per distinct term: kind, decoded lexical form, suffix, typed value
  {
    RDFGPU_HIP(hipMalloc((void**)&out->kind, n_terms ? n_terms : 1));
    u32* lex_len = buf.get<u32>((u64)n_terms + 1); u32* sfx_len = buf.get<u32>((u64)n_terms + 1);
    hipLaunchKernelGGL(nt_decode_lengths_kernel, g256(n_terms), dim3(256), 0, st, d_text, span_off, span_len, term_occ, (u64)n_terms, out->kind, lex_len, sfx_len);
    RDFGPU_HIP(hipMemsetAsync(lex_len + n_terms, 0, 4, st)); RDFGPU_HIP(hipMemsetAsync(sfx_len + n_terms, 0, 4, st));
    RDFGPU_HIP(hipMalloc((void**)&out->lex_off, ((u64)n_terms + 1) * 8)); RDFGPU_HIP(hipMalloc((void**)&out->sfx_off, ((u64)n_terms + 1) * 8));
    size_t tb = 0;
    RDFGPU_HIP(rocprim::exclusive_scan(nullptr, tb, lex_len, out->lex_off, (u64)0, (size_t)n_terms + 1, rocprim::plus<u64>(), st));
    void* temp = buf.get<unsigned char>(tb + 256);
    RDFGPU_HIP(rocprim::exclusive_scan(temp, tb, lex_len, out->lex_off, (u64)0, (size_t)n_terms + 1, rocprim::plus<u64>(), st));
    RDFGPU_HIP(rocprim::exclusive_scan(temp, tb, sfx_len, out->sfx_off, (u64)0, (size_t)n_terms + 1, rocprim::plus<u64>(), st));
    u64 totals[2] = {0, 0};
    RDFGPU_HIP(hipMemcpyAsync(&totals[0], out->lex_off + n_terms, 8, hipMemcpyDeviceToHost, st));
    RDFGPU_HIP(hipMemcpyAsync(&totals[1], out->sfx_off + n_terms, 8, hipMemcpyDeviceToHost, st));
    RDFGPU_HIP(hipStreamSynchronize(st));
    out->lex_total = totals[0]; out->sfx_total = totals[1];
    RDFGPU_HIP(hipMalloc((void**)&out->lex, totals[0] ? totals[0] : 1)); RDFGPU_HIP(hipMalloc((void**)&out->sfx, totals[1] ? totals[1] : 1));
    RDFGPU_HIP(hipMalloc((void**)&out->typed, (n_terms ? n_terms : 1) * sizeof(rdfgpu_typed_value))); RDFGPU_HIP(hipMalloc((void**)&out->dec_hi, (n_terms ? n_terms : 1) * 8));
    hipLaunchKernelGGL(nt_decode_bytes_kernel, g256(n_terms), dim3(256), 0, st, d_text, span_off, span_len, term_occ, (u64)n_terms, out->lex_off, out->lex, out->sfx_off, out->sfx,
                       out->typed, reinterpret_cast<long long*>(out->dec_hi));
  }
  RDFGPU_HIP(hipStreamSynchronize(st));
  return out.release();
}

void ntriples_decoded(const NTriples* t, unsigned char* kind, u64* lex_off, unsigned char* lex, u64* sfx_off, unsigned char* sfx, rdfgpu_typed_value* typed, int64_t* dec_hi) {
  const u64 n = t->n_terms;
  if (kind && n) RDFGPU_HIP(hipMemcpy(kind, t->kind, n, hipMemcpyDeviceToHost));
  if (lex_off) { if (!t->lex_off) lex_off[0] = 0; else RDFGPU_HIP(hipMemcpy(lex_off, t->lex_off, (n + 1) * 8, hipMemcpyDeviceToHost)); }
  if (sfx_off) { if (!t->sfx_off) sfx_off[0] = 0; else RDFGPU_HIP(hipMemcpy(sfx_off, t->sfx_off, (n + 1) * 8, hipMemcpyDeviceToHost)); }
  if (lex && t->lex_total) RDFGPU_HIP(hipMemcpy(lex, t->lex, t->lex_total, hipMemcpyDeviceToHost));
  if (sfx && t->sfx_total) RDFGPU_HIP(hipMemcpy(sfx, t->sfx, t->sfx_total, hipMemcpyDeviceToHost));
  if (typed && n) RDFGPU_HIP(hipMemcpy(typed, t->typed, n * sizeof(rdfgpu_typed_value), hipMemcpyDeviceToHost));
  if (dec_hi && n) RDFGPU_HIP(hipMemcpy(dec_hi, t->dec_hi, n * 8, hipMemcpyDeviceToHost));
}

void ntriples_terms(const NTriples* t, u64* offsets, unsigned char* bytes) {
  if (offsets) {
    if (t->n_terms == 0 && !t->term_off) offsets[0] = 0;
    else RDFGPU_HIP(hipMemcpy(offsets, t->term_off, ((u64)t->n_terms + 1) * 8, hipMemcpyDeviceToHost));
  }
  if (bytes && t->term_total) RDFGPU_HIP(hipMemcpy(bytes, t->term_bytes, t->term_total, hipMemcpyDeviceToHost));
}

// (kernels.hpp, preload_code_objects: the runtime loads a translation unit's code object at the first use of one of its kernels)
void preload_tu_ntriples() { hipFuncAttributes at; RDFGPU_HIP(hipFuncGetAttributes(&at, reinterpret_cast<const void*>(nt_line_flags_kernel))); }
}  // namespace rdfgpu
