// host_logic.cpp — see host_logic.hpp.
#include "host_logic.hpp"

#include <algorithm>

namespace rdfgpu {

ScanInstruction decode_instruction(const rdfgpu_scan_instruction& raw, const u32* pool, u32 n_pool) {
  ScanInstruction out;
  if (raw.kind != RDFGPU_TRAVERSE && raw.kind != RDFGPU_SCAN) fail(RDFGPU_ERR_INVALID, "scan instruction: bad kind %u", raw.kind);
  out.kind = raw.kind;
  out.var = raw.var;
  out.pred.kind = raw.pred;
  switch (raw.pred) {
    case RDFGPU_PRED_NONE: case RDFGPU_PRED_FALSE: break;
    case RDFGPU_PRED_IN:
      if (raw.b == 0) fail(RDFGPU_ERR_INVALID, "scan instruction: empty In set");
      if ((u64)raw.a + raw.b > n_pool) fail(RDFGPU_ERR_INVALID, "scan instruction: In set outside the pool");
      out.pred.ids.assign(pool + raw.a, pool + raw.a + raw.b);
      std::sort(out.pred.ids.begin(), out.pred.ids.end());   // BTreeSet semantics
      out.pred.ids.erase(std::unique(out.pred.ids.begin(), out.pred.ids.end()), out.pred.ids.end());
      break;
    case RDFGPU_PRED_BETWEEN: out.pred.from = raw.a; out.pred.to = raw.b; break;
    case RDFGPU_PRED_EQUAL_TO: out.pred.equal_to = raw.a; break;
    default: fail(RDFGPU_ERR_INVALID, "scan instruction: bad predicate %u", raw.pred);
  }
  return out;
}

ScanInstructions make_gspo(const rdfgpu_scan_instruction raw[4], const u32* pool, u32 n_pool) {
  ScanInstructions out;
  out.components = RDFGPU_GSPO;
  for (int i = 0; i < 4; i++) {
    out.in[i] = decode_instruction(raw[i], pool, n_pool);
    if (out.in[i].kind == RDFGPU_SCAN) {
      for (int j = 0; j < i; j++) {
        if (out.in[j].kind == RDFGPU_SCAN && out.in[j].var == out.in[i].var) {
          // second binding of a variable => Traverse(EqualTo(var)); an own predicate is dropped (:33-37)
          ScanInstruction eq;
          eq.kind = RDFGPU_TRAVERSE;
          eq.pred.kind = RDFGPU_PRED_EQUAL_TO;
          eq.pred.equal_to = out.in[i].var;
          out.in[i] = eq;
          break;
        }
      }
    }
  }
  return out;
}

ScanInstructions reorder(const ScanInstructions& gspo, u32 components) {
  ScanInstructions out;
  out.components = components;
  for (int k = 0; k < 4; k++) out.in[k] = gspo.in[PERM[components][k]];
  return out;
}

// MemIndexPruningPredicate::from (scan_instructions.rs:399-423)
static bool pruning_predicate(const ScanPredicate& p, u32* from, u32* to) {
  if (p.kind == RDFGPU_PRED_IN) { *from = p.ids.front(); *to = p.ids.back(); return true; }
  if (p.kind == RDFGPU_PRED_BETWEEN) { *from = p.from; *to = p.to; return true; }
  return false;
}

u64 scan_score(const ScanInstruction in[4]) {
  u64 score = 0;
  for (int i = 0; i < 4; i++) {
    u32 from, to;
    if (!pruning_predicate(in[i].pred, &from, &to)) break;
    const u64 potent = (u64)(4 - i) * 2;
    const u64 reward = from == to ? 2 : 1;   // EqualTo / Between(x,x) = 2, proper Between = 1
    score += reward << potent;
    if (from != to) break;                   // cannot prune below a range
  }
  return score;
}

u32 choose_index(const ScanInstructions& gspo, u32 available_mask) {
  int best = -1;
  u64 best_score = 0;
  for (u32 comp = 0; comp < RDFGPU_N_INDEXES; comp++) {   // listing order; ties keep the first (`.rev().max_by`)
    if (!(available_mask & (1u << comp))) continue;
    const ScanInstructions r = reorder(gspo, comp);
    const u64 sc = scan_score(r.in);
    if (best < 0 || sc > best_score) { best = (int)comp; best_score = sc; }
  }
  if (best < 0) fail(RDFGPU_ERR_INVALID, "choose_index: no index available");
  return (u32)best;
}

bool predicate_and(const ScanPredicate& a, const ScanPredicate& b, ScanPredicate* out) {
  ScanPredicate r;
  if (a.kind == RDFGPU_PRED_FALSE || b.kind == RDFGPU_PRED_FALSE) { r.kind = RDFGPU_PRED_FALSE; *out = r; return true; }
  if (a.kind == RDFGPU_PRED_IN && b.kind == RDFGPU_PRED_IN) {
    std::set_intersection(a.ids.begin(), a.ids.end(), b.ids.begin(), b.ids.end(), std::back_inserter(r.ids));
    r.kind = r.ids.empty() ? RDFGPU_PRED_FALSE : RDFGPU_PRED_IN;
    *out = r; return true;
  }
  if ((a.kind == RDFGPU_PRED_IN && b.kind == RDFGPU_PRED_BETWEEN) || (a.kind == RDFGPU_PRED_BETWEEN && b.kind == RDFGPU_PRED_IN)) {
    const ScanPredicate& in = a.kind == RDFGPU_PRED_IN ? a : b;
    const ScanPredicate& bt = a.kind == RDFGPU_PRED_IN ? b : a;
    for (u32 v : in.ids) if (v >= bt.from && v <= bt.to) r.ids.push_back(v);
    r.kind = r.ids.empty() ? RDFGPU_PRED_FALSE : RDFGPU_PRED_IN;
    *out = r; return true;
  }
  if (a.kind == RDFGPU_PRED_BETWEEN && b.kind == RDFGPU_PRED_BETWEEN) {
    const u32 from = std::max(a.from, b.from), to = std::min(a.to, b.to);
    if (from > to) r.kind = RDFGPU_PRED_FALSE; else { r.kind = RDFGPU_PRED_BETWEEN; r.from = from; r.to = to; }
    *out = r; return true;
  }
  return false;
}

ScanPredicate pushdown_to_scan_predicate(u32 op, u32 value) {
  ScanPredicate r;
  switch (op) {
    case RDFGPU_OP_GT:
      if (value == 0xFFFFFFFFu) r.kind = RDFGPU_PRED_FALSE;                       // value.next() is None
      else { r.kind = RDFGPU_PRED_BETWEEN; r.from = value + 1; r.to = 0xFFFFFFFFu; }
      break;
    case RDFGPU_OP_GTEQ: r.kind = RDFGPU_PRED_BETWEEN; r.from = value; r.to = 0xFFFFFFFFu; break;
    case RDFGPU_OP_LT:
      if (value == 0) r.kind = RDFGPU_PRED_FALSE;                                 // value.previous() is None
      else { r.kind = RDFGPU_PRED_BETWEEN; r.from = 0; r.to = value - 1; }
      break;
    case RDFGPU_OP_LTEQ: r.kind = RDFGPU_PRED_BETWEEN; r.from = 0; r.to = value; break;
    case RDFGPU_OP_EQ: r.kind = RDFGPU_PRED_IN; r.ids = {value}; break;
    default: fail(RDFGPU_ERR_INVALID, "push-down: unsupported operator %u", op);
  }
  return r;
}

PrunePlan plan_pruning(const ScanInstructions& ix) {
  PrunePlan p;
  for (int k = 0; k < 4; k++) {
    u32 from, to;
    if (!pruning_predicate(ix.in[k].pred, &from, &to)) break;
    p.from[p.n_levels] = from; p.to[p.n_levels] = to; p.n_levels++;
    if (from != to) break;
  }
  // which predicates the range makes redundant (quad_index_data.rs:245-265)
  for (int k = 0; k < 4; k++) {
    const ScanPredicate& q = ix.in[k].pred;
    if (q.kind == RDFGPU_PRED_IN) { if (q.ids.size() == 1) p.dropped_mask |= 1u << k; else break; }
    else if (q.kind == RDFGPU_PRED_BETWEEN) { p.dropped_mask |= 1u << k; if (q.from != q.to) break; }
    else break;
  }
  return p;
}

}  // namespace rdfgpu
