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
  unsigned char* term_bytes = nullptr;   // device: the distinct terms exactly as written in the file
  ~NTriples();
};
NTriples* ntriples_parse(int device, const char* text, u64 n, u32 first_id);
void ntriples_terms(const NTriples* t, u64* offsets, unsigned char* bytes);

}  // namespace rdfgpu
