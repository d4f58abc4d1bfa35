// host_logic.hpp — scan planning on the host: predicate algebra, index choice, pruning levels.
// Mirrors lib/storage/src/memory/storage/{scan_instructions,quad_index}.rs and
// lib/storage/src/index/permutations.rs.  Never touches the device.
#pragma once
#include <vector>

#include "common.hpp"

namespace rdfgpu {

// MemIndexScanPredicate (scan_instructions.rs:157-166)
struct ScanPredicate {
  u32 kind = RDFGPU_PRED_NONE;
  std::vector<u32> ids;   // IN (sorted, unique)
  u32 from = 0, to = 0;   // BETWEEN (inclusive)
  u32 equal_to = 0;       // EQUAL_TO variable slot
};
// MemIndexScanInstruction (scan_instructions.rs:247-252)
struct ScanInstruction {
  u32 kind = RDFGPU_TRAVERSE;
  u32 var = 0;
  ScanPredicate pred;
};
// MemIndexScanInstructions (scan_instructions.rs:13): components + 4 instructions in that order
struct ScanInstructions {
  u32 components = RDFGPU_GSPO;
  ScanInstruction in[4];
};

ScanInstruction decode_instruction(const rdfgpu_scan_instruction& raw, const u32* pool, u32 n_pool);
// MemIndexScanInstructions::new (scan_instructions.rs:19-46): a variable bound twice becomes EqualTo
ScanInstructions make_gspo(const rdfgpu_scan_instruction raw[4], const u32* pool, u32 n_pool);
// ScanInstructions::reorder (scan_instructions.rs:137-152)
ScanInstructions reorder(const ScanInstructions& gspo, u32 components);
// compute_scan_score (quad_index.rs:100-130)
u64 scan_score(const ScanInstruction in[4]);
// choose_index (permutations.rs:81-96)
u32 choose_index(const ScanInstructions& gspo, u32 available_mask);
// try_and_with (scan_instructions.rs:170-210); false = not combinable
bool predicate_and(const ScanPredicate& a, const ScanPredicate& b, ScanPredicate* out);
// to_scan_predicate (predicate_pushdown.rs:120-157)
ScanPredicate pushdown_to_scan_predicate(u32 op, u32 value);

// What prune_relevant_row_groups (quad_index_data.rs:155-284) decides without looking at data:
// the leading levels that narrow the range, and which predicates the narrowing makes redundant.
struct PrunePlan {
  u32 n_levels = 0;
  u32 from[4] = {}, to[4] = {};
  u32 dropped_mask = 0;
};
PrunePlan plan_pruning(const ScanInstructions& ix);

}  // namespace rdfgpu
