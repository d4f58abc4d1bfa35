// regex_prog.hpp — the compiled form of a SPARQL REGEX pattern shared by the host compiler (regex_compile.hpp)
// and the device simulation (expr_device.hpp): a Glushkov position automaton over bytes, <= 64 positions.
#pragma once
#include <stdint.h>

namespace rdfgpu {

struct RegexProg {           // device-readable; 3.7 KB
  uint64_t first, last;
  uint64_t follow[64];
  uint64_t byte_mask[256];   // positions whose byte set contains the byte
  uint32_t n_pos;
  uint8_t nullable, anchor_start, anchor_end;
  uint8_t ml_start, ml_end;       // `m`: ^ also matches after, $ also before a '\n' (\A / \z never do)
  uint8_t always_error;           // invalid flag letter: every row evaluates to the error value (regex.rs:137)
  // `\b` / `\B`: transitions, starts, ends and empty matches that additionally need a word boundary (_b) or its absence (_nb)
  // at the point they cross; has_assert == 0: all of these are zero
  uint8_t has_assert;
  // the pattern uses `\d \w \s \b` (or a negation): compiled with their ASCII members, which is what the crate's Unicode
  // classes are on an all-ASCII subject; a subject with a non-ASCII byte would need the Unicode tables — it raises the
  // plan's run-time error instead of being answered differently from the crate
  uint8_t ascii_only;
  uint64_t first_b, first_nb, last_b, last_nb;
  uint64_t follow_b[64], follow_nb[64];
  uint8_t nullable_b, nullable_nb, pad2[2];
  uint32_t pattern_id;            // REGEX with a per-row pattern: the object id of the pattern literal this program was compiled from
};

}  // namespace rdfgpu
