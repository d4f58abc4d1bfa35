// ntriples.hpp — N-Triples text -> object ids on the device (ntriples.hip): the handle behind rdfgpu_ntriples_*.
#pragma once
#include "common.hpp"

namespace rdfgpu {

struct NTriples {
  u32 first_id = 1;
  u64 n_triples = 0;
  u32 n_terms = 0;
  u64 term_total = 0;                    // bytes of the distinct terms
  u32* s = nullptr; u32* p = nullptr; u32* o = nullptr;   // device: one id per triple line, in file order
  u64* term_off = nullptr;               // device: [n_terms + 1] offsets into term_bytes, term t has id first_id + t
  unsigned char* term_bytes = nullptr;   // device: the distinct terms as written in the file (one spelling each)
  // per distinct term, escapes decoded: kind (1 IRI, 2 blank node, 3 simple / 4 language-tagged / 5 typed literal), lexical form,
  // suffix (language tag in lower case / datatype IRI), and the typed-value row the device could derive (flags & 0x80: the host parses it)
  unsigned char* kind = nullptr; u64* lex_off = nullptr; unsigned char* lex = nullptr; u64* sfx_off = nullptr; unsigned char* sfx = nullptr;
  u64 lex_total = 0, sfx_total = 0;
  rdfgpu_typed_value* typed = nullptr; int64_t* dec_hi = nullptr;
  ~NTriples();
};
void ntriples_decoded(const NTriples* t, unsigned char* kind, u64* lex_off, unsigned char* lex, u64* sfx_off, unsigned char* sfx, rdfgpu_typed_value* typed, int64_t* dec_hi);
NTriples* ntriples_parse(int device, const char* text, u64 n, u32 first_id);
void ntriples_terms(const NTriples* t, u64* offsets, unsigned char* bytes);

}  // namespace rdfgpu
