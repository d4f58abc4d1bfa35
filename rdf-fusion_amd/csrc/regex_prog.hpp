// regex_prog.hpp — the compiled form of a SPARQL REGEX pattern shared by the host compiler (regex_compile.hpp)
// and the device simulation (expr_device.hpp): a Glushkov position automaton over bytes, <= 64 positions.
#pragma once
#include <stdint.h>

namespace rdfgpu {

struct RegexProg {           // device-readable; 2.6 KB
  uint64_t first, last;
  uint64_t follow[64];
  uint64_t byte_mask[256];   // positions whose byte set contains the byte
  uint32_t n_pos;
  uint8_t nullable, anchor_start, anchor_end;
  uint8_t ml_start, ml_end;       // `m`: ^ also matches after, $ also before a '\n' (\A / \z never do)
  uint8_t always_error, pad[2];   // invalid flag letter: every row evaluates to the error value (regex.rs:137)
};

}  // namespace rdfgpu
