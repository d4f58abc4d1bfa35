// plan.cpp — plan compilation (host) and execution (device) of the scan / filter / join subtree.
//
// What replaces what (reference paths relative to the rdf-fusion tree):
//   plan_compile            MemQuadStorePlanner::plan_extension  lib/storage/src/memory/planner.rs:31-64
//                           + plan_pattern_evaluation             storage/snapshot.rs:84-131
//   Plan::exec_source       DataSource::open + MemQuadIndexScanIterator::next  pattern_data_source.rs:42-58, scan.rs:104-212
//   Plan::exec_filter       FilterExec over ENC_TV/GT/ADD/EBV UDFs  (DataFusion 52 + lib/functions)
//   Plan::exec_join         HashJoinExec(CollectLeft) / CrossJoinExec / NestedLoopJoinExec (DataFusion 52),
//                           semantics from lib/logical/src/join/rewrite.rs:71-221
#include "plan.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <shared_mutex>

#include "regex_compile.hpp"

namespace rdfgpu {

// ------------------------------------------------------------------------------------------------
// compile
// ------------------------------------------------------------------------------------------------
namespace {

bool is_cmp(u8 op) {
  return op == RDFGPU_EX_GT || op == RDFGPU_EX_LT || op == RDFGPU_EX_GEQ || op == RDFGPU_EX_LEQ || op == RDFGPU_EX_EQ || op == RDFGPU_EX_NEQ;
}

// Type-checks a postfix program against `n_cols` input columns; returns the kind it leaves.
u32 check_program(const rdfgpu_expr_node* p, u32 n, u32 n_cols, u32 n_regexes = 0) {
  if (n > (u32)kMaxExpr) fail(RDFGPU_ERR_UNSUPPORTED, "expression has %u nodes (max %d)", n, kMaxExpr);
  u32 st[kMaxStack]; bool no_bytes[kMaxStack]; int sp = 0;
  bool views = false, rank_only_string = false;   // computed strings (views) / a string literal given by its rank in the dictionary only
  bool out_no_bytes = false;                      // the value being pushed is such a literal: it has no lexical form on the device
  auto pop = [&](u32 kind, const char* what) {
    if (sp < 1) fail(RDFGPU_ERR_INVALID, "expression: stack underflow at %s", what);
    if (st[--sp] != kind) fail(RDFGPU_ERR_INVALID, "expression: %s got an operand of the wrong kind", what);
  };
  // an operand whose BYTES the op reads (REGEX / CONTAINS / STRSTARTS / STRENDS / STRLEN / SUBSTR / UCASE / LCASE): a string literal
  // that came with its rank only would be the error value on every row — refused here, loudly, instead
  auto pop_bytes = [&](const char* what) {
    if (sp >= 1 && no_bytes[sp - 1]) fail(RDFGPU_ERR_UNSUPPORTED, "%s over a string literal given by its dictionary rank only: it has no lexical form on the device (pass it as RDFGPU_EX_LIT_STR)", what);
    pop(VK_TV, what);
  };
  for (u32 i = 0; i < n; i++) {
    const rdfgpu_expr_node& e = p[i];
    u32 out;
    out_no_bytes = false;
    switch (e.op) {
      case RDFGPU_EX_COLUMN: if (e.u >= n_cols) fail(RDFGPU_ERR_INVALID, "expression: column %u out of range (%u columns)", e.u, n_cols); out = VK_ID; break;
      case RDFGPU_EX_LIT_ID: out = VK_ID; break;
      case RDFGPU_EX_LIT_TV: if (e.tag > RDFGPU_TV_OTHER) fail(RDFGPU_ERR_INVALID, "expression: bad literal tag %u", e.tag); out = VK_TV;
        out_no_bytes = e.tag == RDFGPU_TV_STRING && e.hi == 0;
        rank_only_string = rank_only_string || out_no_bytes; break;
      case RDFGPU_EX_LIT_BOOL: out = VK_BOOL; break;
      case RDFGPU_EX_ENC_TV: pop(VK_ID, "ENC_TV"); out = VK_TV; break;
      case RDFGPU_EX_GT: case RDFGPU_EX_LT: case RDFGPU_EX_GEQ: case RDFGPU_EX_LEQ: case RDFGPU_EX_EQ: case RDFGPU_EX_NEQ:
      case RDFGPU_EX_ADD: case RDFGPU_EX_SUB: pop(VK_TV, "binary typed op"); pop(VK_TV, "binary typed op"); out = VK_TV; break;
      case RDFGPU_EX_EBV: pop(VK_TV, "EBV"); out = VK_BOOL; break;
      case RDFGPU_EX_REGEX: case RDFGPU_EX_CONTAINS: case RDFGPU_EX_STRSTARTS: case RDFGPU_EX_STRENDS:
        if (e.u >= n_regexes) fail(RDFGPU_ERR_INVALID, "expression: REGEX pattern %u out of range (%u patterns)", e.u, n_regexes);
        // (the operand is any string value: ENC_TV of a column, or a view — STR / SUBSTR / UCASE / LCASE / a constant with bytes)
        pop_bytes("REGEX / CONTAINS / STRSTARTS / STRENDS"); out = VK_TV; break;
      case RDFGPU_EX_STR: pop(VK_ID, "STR"); out = VK_TV; views = true; break;
      case RDFGPU_EX_LIT_STR:
        if (e.u >= n_regexes) fail(RDFGPU_ERR_INVALID, "expression: string constant %u out of range (%u entries)", e.u, n_regexes);
        out = VK_TV; views = true; break;
      case RDFGPU_EX_STRLEN: pop_bytes("STRLEN"); out = VK_TV; break;
      case RDFGPU_EX_SUBSTR:
        if (e.u != 2 && e.u != 3) fail(RDFGPU_ERR_INVALID, "expression: SUBSTR takes 2 or 3 operands, not %u", e.u);
        for (u32 k = 0; k + 1 < e.u; k++) pop(VK_TV, "SUBSTR position / length");
        pop_bytes("SUBSTR");
        out = VK_TV; views = true; break;
      case RDFGPU_EX_UCASE: case RDFGPU_EX_LCASE: pop_bytes("UCASE / LCASE"); out = VK_TV; views = true; break;
      case RDFGPU_EX_STRBEFORE: case RDFGPU_EX_STRAFTER: pop_bytes("STRBEFORE / STRAFTER"); pop_bytes("STRBEFORE / STRAFTER"); out = VK_TV; views = true; break;
      case RDFGPU_EX_REGEX_VAR:
        if (e.lo < 1 || (u64)e.u + (u64)e.lo > n_regexes) fail(RDFGPU_ERR_INVALID, "expression: REGEX pattern table %u .. +%lld out of range (%u patterns)", e.u, (long long)e.lo, n_regexes);
        pop(VK_TV, "REGEX pattern"); pop_bytes("REGEX"); out = VK_TV; break;
      case RDFGPU_EX_LANG_IN:
        if (e.u >= n_regexes) fail(RDFGPU_ERR_INVALID, "expression: language table %u out of range (%u tables)", e.u, n_regexes);
        pop(VK_TV, "LANGMATCHES(LANG())"); out = VK_TV; break;
      case RDFGPU_EX_ID_EQ: case RDFGPU_EX_ID_NEQ: case RDFGPU_EX_IS_COMPATIBLE: pop(VK_ID, "id comparison"); pop(VK_ID, "id comparison"); out = VK_BOOL; break;
      case RDFGPU_EX_AND: case RDFGPU_EX_OR: pop(VK_BOOL, "AND/OR"); pop(VK_BOOL, "AND/OR"); out = VK_BOOL; break;
      case RDFGPU_EX_NOT: pop(VK_BOOL, "NOT"); out = VK_BOOL; break;
      case RDFGPU_EX_BOUND: pop(VK_ID, "BOUND"); out = VK_BOOL; break;
      case RDFGPU_EX_BOOL_AS_TV: pop(VK_BOOL, "BOOLEAN_AS_TERM"); out = VK_TV; break;
      default: fail(RDFGPU_ERR_INVALID, "expression: unknown op %u", e.op);
    }
    if (sp >= kMaxStack) fail(RDFGPU_ERR_UNSUPPORTED, "expression: stack deeper than %d", kMaxStack);
    no_bytes[sp] = out_no_bytes;
    st[sp++] = out;
  }
  if (sp != 1) fail(RDFGPU_ERR_INVALID, "expression leaves %d values on the stack", sp);
  // a computed string compares byte-wise; a string literal that comes with its dictionary rank only has no bytes on the device
  if (views && rank_only_string) fail(RDFGPU_ERR_UNSUPPORTED, "expression mixes computed strings (STR / SUBSTR / UCASE / LCASE) with a string literal given by rank: pass the literal as RDFGPU_EX_LIT_STR");
  return st[0];
}

int detect_shape(const ExprProgram& pr, bool force_vm) {
  if (force_vm) return 0;
  const rdfgpu_expr_node* e = pr.nodes;
  if (pr.n == 3 && e[0].op == RDFGPU_EX_COLUMN && e[1].op == RDFGPU_EX_LIT_ID && (e[2].op == RDFGPU_EX_ID_EQ || e[2].op == RDFGPU_EX_ID_NEQ)) return 1;
  if (pr.n == 5 && e[0].op == RDFGPU_EX_COLUMN && e[1].op == RDFGPU_EX_ENC_TV && e[2].op == RDFGPU_EX_LIT_TV && is_cmp(e[3].op) && e[4].op == RDFGPU_EX_EBV) return 2;
  // EBV(REGEX | CONTAINS | STRSTARTS | STRENDS (ENC_TV(col), constant)): answered per distinct term (shape 3) when the
  // table is large enough to pay for a pass over the dictionary, else by the VM per row
  if (pr.n == 4 && e[0].op == RDFGPU_EX_COLUMN && e[1].op == RDFGPU_EX_ENC_TV && e[3].op == RDFGPU_EX_EBV &&
      (e[2].op == RDFGPU_EX_REGEX || e[2].op == RDFGPU_EX_CONTAINS || e[2].op == RDFGPU_EX_STRSTARTS || e[2].op == RDFGPU_EX_STRENDS)) return 3;
  return 0;
}

// Join-filter specialisation: 3 = the BSBM Q5 "window" shape
//   EBV(cmp(ENC_TV(x), ADD|SUB(ENC_TV(y), lit))) AND EBV(cmp(ENC_TV(x'), ADD|SUB(ENC_TV(y'), lit')))
// (Q5 (Execution Plan).snap:10,12), 1 = generic VM, 0 = no filter.
int detect_join_filter_shape(const ExprProgram& pr, bool force_vm) {
  if (pr.n == 0) return 0;
  if (force_vm) return 1;
  const rdfgpu_expr_node* e = pr.nodes;
  auto half = [&](u32 o) {
    return e[o].op == RDFGPU_EX_COLUMN && e[o + 1].op == RDFGPU_EX_ENC_TV && e[o + 2].op == RDFGPU_EX_COLUMN &&
           e[o + 3].op == RDFGPU_EX_ENC_TV && e[o + 4].op == RDFGPU_EX_LIT_TV &&
           (e[o + 5].op == RDFGPU_EX_ADD || e[o + 5].op == RDFGPU_EX_SUB) && is_cmp(e[o + 6].op) && e[o + 7].op == RDFGPU_EX_EBV;
  };
  if (pr.n == 17 && half(0) && half(8) && e[16].op == RDFGPU_EX_AND) return 3;
  // 2 = column <ID_EQ | ID_NEQ> column
  if (pr.n == 3 && e[0].op == RDFGPU_EX_COLUMN && e[1].op == RDFGPU_EX_COLUMN && (e[2].op == RDFGPU_EX_ID_EQ || e[2].op == RDFGPU_EX_ID_NEQ)) return 2;
  return 1;
}

void load_program(NodeInfo& nd, const rdfgpu_plan_desc* d, u32 n_cols, const char* what, const RegexProg* regex_dev, const unsigned char* str_consts = nullptr) {
  const rdfgpu_plan_node& r = nd.d;
  nd.prog.n = 0;
  if (r.expr_len == 0) return;
  if ((u64)r.expr_off + r.expr_len > d->n_exprs) fail(RDFGPU_ERR_INVALID, "%s: expression outside the expression array", what);
  if (check_program(d->exprs + r.expr_off, r.expr_len, n_cols, d->n_regexes) != VK_BOOL) fail(RDFGPU_ERR_INVALID, "%s: predicate does not yield a boolean", what);
  nd.prog.n = r.expr_len;
  std::memcpy(nd.prog.nodes, d->exprs + r.expr_off, r.expr_len * sizeof(rdfgpu_expr_node));
  nd.prog.regex = regex_dev;
  nd.prog.str_consts = str_consts;
}

void load_projection(NodeInfo& nd, const rdfgpu_plan_desc* d, u32 full, const char* what) {
  const rdfgpu_plan_node& r = nd.d;
  if (r.n_proj == RDFGPU_NO_PROJECTION) {
    if (full > (u32)kMaxCols) fail(RDFGPU_ERR_UNSUPPORTED, "%s: %u columns (max %d)", what, full, kMaxCols);
    nd.n_proj = full;
    for (u32 i = 0; i < full; i++) nd.proj[i] = i;
  } else {
    if (r.n_proj > (u32)kMaxCols) fail(RDFGPU_ERR_UNSUPPORTED, "%s: %u columns (max %d)", what, r.n_proj, kMaxCols);
    if ((u64)r.proj_off + r.n_proj > d->n_pool) fail(RDFGPU_ERR_INVALID, "%s: projection outside the pool", what);
    nd.n_proj = r.n_proj;
    for (u32 i = 0; i < r.n_proj; i++) {
      nd.proj[i] = d->pool[r.proj_off + i];
      if (nd.proj[i] >= full) fail(RDFGPU_ERR_INVALID, "%s: projection column %u out of range (%u columns)", what, nd.proj[i], full);
    }
  }
  nd.width = nd.n_proj;
}

}  // namespace

Plan* plan_compile(Store* store, const rdfgpu_plan_desc* d) {
  if (!store) fail(RDFGPU_ERR_INVALID, "plan_compile: null store");
  if (!d || !d->nodes || d->n_nodes == 0) fail(RDFGPU_ERR_INVALID, "plan_compile: empty plan");
  if (d->root >= d->n_nodes) fail(RDFGPU_ERR_INVALID, "plan_compile: root %u out of range", d->root);
  std::unique_ptr<Plan> plan(new Plan());
  plan->store = store;
  plan->opt = store->opt;
  store->retain();
  plan->root = d->root;
  plan->nodes.resize(d->n_nodes);
  std::vector<u32> scan_ids;  // sorted IN sets of all sources, uploaded once
  if (d->n_regexes) {   // REGEX patterns are plan constants: compiled here, simulated per row on the device
    if (!d->regexes) fail(RDFGPU_ERR_INVALID, "plan_compile: %u regexes but no table", d->n_regexes);
    // how each table entry is used decides how it is compiled: REGEX = a pattern with flags; CONTAINS / STRSTARTS /
    // STRENDS = a literal needle (like the `q` flag), anchored at the start / end for the latter two
    std::vector<int> use(d->n_regexes, -1);
    for (u32 i = 0; i < d->n_exprs; i++) {
      const rdfgpu_expr_node& e = d->exprs[i];
      if (e.op == RDFGPU_EX_REGEX_VAR) {   // a table of per-row patterns: entries u .. u + lo, all REGEX patterns
        for (int64_t k = 0; k < e.lo && (u64)e.u + (u64)k < d->n_regexes; k++) use[e.u + k] = RDFGPU_EX_REGEX;
        continue;
      }
      if (e.op != RDFGPU_EX_REGEX && e.op != RDFGPU_EX_CONTAINS && e.op != RDFGPU_EX_STRSTARTS && e.op != RDFGPU_EX_STRENDS && e.op != RDFGPU_EX_LANG_IN && e.op != RDFGPU_EX_LIT_STR) continue;
      if (e.u >= d->n_regexes) fail(RDFGPU_ERR_INVALID, "expression: string pattern %u out of range", e.u);
      if (use[e.u] >= 0 && use[e.u] != (int)e.op) fail(RDFGPU_ERR_INVALID, "string pattern %u is used by two different functions", e.u);
      use[e.u] = (int)e.op;
    }
    std::vector<RegexProg> progs(d->n_regexes);
    std::vector<unsigned char> consts;   // the bytes of the string constants (RDFGPU_EX_LIT_STR), back to back
    for (u32 r = 0; r < d->n_regexes; r++) {
      const rdfgpu_regex& rx = d->regexes[r];
      if (use[r] == RDFGPU_EX_LIT_STR) {   // not a pattern: raw bytes — the slot holds where they are
        std::memset(&progs[r], 0, sizeof(RegexProg));
        progs[r].first = consts.size(); progs[r].n_pos = rx.pattern_len;
        if (rx.pattern_len && !rx.pattern) fail(RDFGPU_ERR_INVALID, "string constant %u: null text", r);
        consts.insert(consts.end(), reinterpret_cast<const unsigned char*>(rx.pattern), reinterpret_cast<const unsigned char*>(rx.pattern) + rx.pattern_len);
        continue;
      }
      if (use[r] == RDFGPU_EX_LANG_IN) {   // not a pattern: one verdict byte per language id -> a bit set in the slot
        std::memset(&progs[r], 0, sizeof(RegexProg));
        if (rx.pattern_len > 256u * 64u) fail(RDFGPU_ERR_UNSUPPORTED, "language table %u: %u language ids (max 16384)", r, rx.pattern_len);
        for (u32 l = 0; l < rx.pattern_len; l++) if (rx.pattern[l]) progs[r].byte_mask[l >> 6] |= 1ull << (l & 63u);
        progs[r].n_pos = rx.pattern_len;
        continue;
      }
      if (use[r] >= 0 && !store->str_off) fail(RDFGPU_ERR_INVALID, "plan uses string functions but the store has no strings (rdfgpu_store_set_strings)");
      std::string why;
      const bool literal = use[r] == RDFGPU_EX_CONTAINS || use[r] == RDFGPU_EX_STRSTARTS || use[r] == RDFGPU_EX_STRENDS;
      const char* flags = literal ? "q" : (rx.flags ? rx.flags : "");
      const size_t n_flags = literal ? 1 : (rx.flags ? rx.flags_len : 0);
      if (regex_compile(rx.pattern ? rx.pattern : "", rx.pattern_len, flags, n_flags, progs[r], why) != REGEX_OK)
        fail(RDFGPU_ERR_UNSUPPORTED, "string pattern %u: %s", r, why.c_str());
      progs[r].pattern_id = rx.pattern_id;
      if (use[r] == RDFGPU_EX_STRSTARTS) progs[r].anchor_start = 1;
      if (use[r] == RDFGPU_EX_STRENDS) progs[r].anchor_end = 1;
    }
    for (u32 r = 0; r < d->n_regexes; r++) {   // own copies of the texts: they key the store's per-term verdict tables
      const rdfgpu_regex& rx = d->regexes[r];
      plan->regex_strings.emplace_back(rx.pattern ? std::string(rx.pattern, rx.pattern_len) : std::string());
      plan->regex_strings.emplace_back(rx.flags ? std::string(rx.flags, rx.flags_len) : std::string());
    }
    for (u32 r = 0; r < d->n_regexes; r++) {
      rdfgpu_regex rx{};
      rx.pattern = plan->regex_strings[2 * r].data(); rx.pattern_len = (u32)plan->regex_strings[2 * r].size();
      rx.flags = plan->regex_strings[2 * r + 1].data(); rx.flags_len = (u32)plan->regex_strings[2 * r + 1].size();
      plan->regex_text.push_back(rx);
    }
    store->activate();
    RDFGPU_HIP(hipMalloc((void**)&plan->regex_dev, progs.size() * sizeof(RegexProg)));
    RDFGPU_HIP(hipMemcpy(plan->regex_dev, progs.data(), progs.size() * sizeof(RegexProg), hipMemcpyHostToDevice));
    bool any_const = false;
    for (int u_ : use) any_const = any_const || u_ == RDFGPU_EX_LIT_STR;
    if (any_const) {   // (at least one byte: the empty string is a constant too, and a null base would read as "no bytes on the device")
      RDFGPU_HIP(hipMalloc((void**)&plan->str_consts_dev, consts.size() + 1));
      if (!consts.empty()) RDFGPU_HIP(hipMemcpy(plan->str_consts_dev, consts.data(), consts.size(), hipMemcpyHostToDevice));
    }
  }

  for (u32 i = 0; i < d->n_exprs; i++) {
    const u8 op = d->exprs[i].op;
    if ((op == RDFGPU_EX_STR || op == RDFGPU_EX_STRLEN || op == RDFGPU_EX_SUBSTR || op == RDFGPU_EX_UCASE || op == RDFGPU_EX_LCASE || op == RDFGPU_EX_STRBEFORE || op == RDFGPU_EX_STRAFTER) && !store->str_off)
      fail(RDFGPU_ERR_INVALID, "plan uses string functions but the store has no strings (rdfgpu_store_set_strings)");
  }
  for (u32 i = 0; i < d->n_nodes; i++) {
    NodeInfo& nd = plan->nodes[i];
    nd.d = d->nodes[i];
    const rdfgpu_plan_node& r = nd.d;
    auto child = [&](int32_t c, const char* what) -> const NodeInfo& {
      if (c < 0 || (u32)c >= i) fail(RDFGPU_ERR_INVALID, "node %u: %s child %d must precede the node", i, what, c);
      return plan->nodes[c];
    };
    switch (r.kind) {
      case RDFGPU_NODE_DATA_SOURCE: {
        SourceInfo src;
        src.node = i;
        src.gspo = make_gspo(r.scan, d->pool, d->n_pool);
        plan->derive_source(src, src.gspo);
        nd.width = src.n_out;
        nd.source = (int)plan->sources.size();
        plan->sources.push_back(src);
        break;
      }
      case RDFGPU_NODE_FILTER: {
        const NodeInfo& c = child(r.left, "input");
        load_program(nd, d, c.width, "FilterExec", plan->regex_dev, plan->str_consts_dev);
        load_projection(nd, d, c.width, "FilterExec");
        nd.shape = detect_shape(nd.prog, plan->opt.on(RDFGPU_OPT_FORCE_GENERIC_VM));
        break;
      }
      case RDFGPU_NODE_PROJECTION: {
        const NodeInfo& c = child(r.left, "input");
        load_projection(nd, d, c.width, "ProjectionExec");
        break;
      }
      case RDFGPU_NODE_HASH_JOIN: case RDFGPU_NODE_CROSS_JOIN: case RDFGPU_NODE_NESTED_LOOP_JOIN: {
        const NodeInfo& l = child(r.left, "left");
        const NodeInfo& rr = child(r.right, "right");
        if (r.join_type != RDFGPU_JOIN_INNER && r.join_type != RDFGPU_JOIN_LEFT) fail(RDFGPU_ERR_UNSUPPORTED, "node %u: join type %u", i, r.join_type);
        if (r.kind == RDFGPU_NODE_HASH_JOIN) {
          if (r.n_keys == 0 || r.n_keys > RDFGPU_MAX_KEYS) fail(RDFGPU_ERR_INVALID, "node %u: HashJoinExec needs 1..%u keys", i, RDFGPU_MAX_KEYS);
          for (u32 k = 0; k < r.n_keys; k++)
            if (r.left_keys[k] >= l.width || r.right_keys[k] >= rr.width) fail(RDFGPU_ERR_INVALID, "node %u: join key out of range", i);
        }
        if (r.kind == RDFGPU_NODE_CROSS_JOIN && (r.expr_len || r.join_type != RDFGPU_JOIN_INNER)) fail(RDFGPU_ERR_INVALID, "node %u: CrossJoinExec takes no filter / join type", i);
        if (l.width + rr.width > 2u * kMaxCols) fail(RDFGPU_ERR_UNSUPPORTED, "node %u: too many columns", i);
        load_program(nd, d, l.width + rr.width, "join filter", plan->regex_dev, plan->str_consts_dev);
        load_projection(nd, d, l.width + rr.width, "join");
        nd.shape = detect_join_filter_shape(nd.prog, plan->opt.on(RDFGPU_OPT_FORCE_GENERIC_VM));
        if (l.width > (u32)kMaxCols || rr.width > (u32)kMaxCols) fail(RDFGPU_ERR_UNSUPPORTED, "node %u: too many columns", i);
        break;
      }
      case RDFGPU_NODE_CLOSURE: {
        const NodeInfo& c = child(r.left, "inner paths");
        if (c.width != 3) fail(RDFGPU_ERR_INVALID, "node %u: KleenePlusClosureExec input has %u columns, not (graph, start, end)", i, c.width);
        if (r.join_type > 1) fail(RDFGPU_ERR_INVALID, "node %u: allow_cross_graph_paths is 0 or 1", i);
        load_projection(nd, d, 3, "KleenePlusClosureExec");
        break;
      }
      case RDFGPU_NODE_UNION: {
        const NodeInfo& l = child(r.left, "left");
        const NodeInfo& rr = child(r.right, "right");
        if (l.width != rr.width) fail(RDFGPU_ERR_INVALID, "node %u: UnionExec inputs have %u and %u columns", i, l.width, rr.width);
        load_projection(nd, d, l.width, "UnionExec");
        break;
      }
      case RDFGPU_NODE_TABLE: {
        if (r.table_cols > (u32)kMaxCols) fail(RDFGPU_ERR_UNSUPPORTED, "node %u: table with %u columns", i, r.table_cols);
        nd.width = r.table_cols;
        if (plan->tables.size() <= r.table_slot) plan->tables.resize(r.table_slot + 1);
        break;
      }
      case RDFGPU_NODE_TOPK: {   // DISTINCT + TopK(fetch) per group, ..Q5 (Execution Plan).snap:5-9
        const NodeInfo& c = child(r.left, "input");
        if (r.n_keys < 1 || r.n_keys > RDFGPU_MAX_KEYS) fail(RDFGPU_ERR_UNSUPPORTED, "node %u: TopK with %u sort keys (1 to %u)", i, r.n_keys, RDFGPU_MAX_KEYS);
        if (r.table_cols < 1 || r.table_cols > 1024) fail(RDFGPU_ERR_UNSUPPORTED, "node %u: TopK fetch = %u", i, r.table_cols);
        for (u32 k = 0; k < r.n_keys; k++) {
          if (r.left_keys[k] >= c.width) fail(RDFGPU_ERR_INVALID, "node %u: sort key column %u out of range", i, r.left_keys[k]);
          if (r.right_keys[k] > RDFGPU_SORT_BY_DOUBLE) fail(RDFGPU_ERR_INVALID, "node %u: unknown sort mode %u", i, r.right_keys[k]);
        }
        if (r.table_slot > c.width) fail(RDFGPU_ERR_INVALID, "node %u: group column out of range", i);
        load_projection(nd, d, c.width, "TopK");
        for (u32 q = 0; q < nd.n_proj; q++) {   // DISTINCT is over (group, keys): the output may not carry anything else
          bool covered = r.table_slot != 0 && nd.proj[q] == r.table_slot - 1;
          for (u32 k = 0; k < r.n_keys; k++) covered = covered || (nd.proj[q] == r.left_keys[k] && r.right_keys[k] == RDFGPU_SORT_BY_ID);
          if (!covered) fail(RDFGPU_ERR_UNSUPPORTED, "node %u: TopK output column %u is neither the group nor a sort key by id", i, nd.proj[q]);
        }
        nd.width = nd.n_proj;
        break;
      }
      default: fail(RDFGPU_ERR_INVALID, "node %u: unknown kind %u", i, r.kind);
    }
  }

  // Join reordering (physical rewrite, results unchanged): an inner HashJoinExec whose build child is a
  // CrossJoinExec(A, B) and whose equi-keys come partly from A and partly from B
  //     (A x B) JOIN C ON a = c1 AND b = c2          (Q5 (Execution Plan).snap:18-26: label x features(X))
  // is the join graph A - C - B; it runs as  A JOIN (B JOIN C ON b = c2) ON a = c1  without ever
  // materialising |A| x |B| rows.  Column order [A, B, C] is preserved, so filter and projection stay valid.
  if (!plan->opt.on(RDFGPU_OPT_NO_JOIN_REORDER)) {
    const u32 n0 = (u32)plan->nodes.size();
    std::vector<u32> refs(n0, 0);
    for (u32 i = 0; i < n0; i++) {
      if (plan->nodes[i].d.left >= 0) refs[plan->nodes[i].d.left]++;
      if (plan->nodes[i].d.right >= 0) refs[plan->nodes[i].d.right]++;
    }
    for (u32 i = 0; i < n0; i++) {
      if (plan->nodes[i].d.kind != RDFGPU_NODE_HASH_JOIN || plan->nodes[i].d.join_type != RDFGPU_JOIN_INNER) continue;
      const u32 ci = (u32)plan->nodes[i].d.left;
      if (plan->nodes[ci].d.kind != RDFGPU_NODE_CROSS_JOIN || refs[ci] != 1) continue;
      const u32 ai = (u32)plan->nodes[ci].d.left, bi = (u32)plan->nodes[ci].d.right, cri = (u32)plan->nodes[i].d.right;
      const u32 wA = plan->nodes[ai].width, wB = plan->nodes[bi].width, wC = plan->nodes[cri].width;
      bool identity = plan->nodes[ci].n_proj == wA + wB;
      for (u32 k = 0; identity && k < wA + wB; k++) identity = plan->nodes[ci].proj[k] == k;
      if (!identity || wB + wC > (u32)kMaxCols) continue;
      rdfgpu_plan_node jd = plan->nodes[i].d;
      u32 nA = 0, nB = 0;
      rdfgpu_plan_node td{};   // T = B JOIN C
      td.kind = RDFGPU_NODE_HASH_JOIN; td.join_type = RDFGPU_JOIN_INNER; td.left = (int32_t)bi; td.right = (int32_t)cri;
      td.n_proj = RDFGPU_NO_PROJECTION;
      u32 la[RDFGPU_MAX_KEYS], ra[RDFGPU_MAX_KEYS];
      for (u32 k = 0; k < jd.n_keys; k++) {
        if (jd.left_keys[k] < wA) { la[nA] = jd.left_keys[k]; ra[nA] = wB + jd.right_keys[k]; nA++; }
        else { td.left_keys[nB] = jd.left_keys[k] - wA; td.right_keys[nB] = jd.right_keys[k]; nB++; }
      }
      if (nA == 0 || nB == 0) continue;
      td.n_keys = nB;
      NodeInfo t;
      t.d = td; t.width = wB + wC; t.n_proj = wB + wC;
      for (u32 k = 0; k < wB + wC; k++) t.proj[k] = k;
      plan->nodes.push_back(t);
      NodeInfo& j = plan->nodes[i];
      j.d.left = (int32_t)ai; j.d.right = (int32_t)(plan->nodes.size() - 1);
      j.d.n_keys = nA;
      for (u32 k = 0; k < nA; k++) { j.d.left_keys[k] = la[k]; j.d.right_keys[k] = ra[k]; }
    }
  }

  // consumers per node, counted over the operators reachable from the root only (a rewritten-away
  // CrossJoinExec must not keep its former inputs "shared")
  for (NodeInfo& nd : plan->nodes) nd.refs = 0;
  {
    std::vector<u32> stack{plan->root};
    std::vector<bool> seen(plan->nodes.size(), false);
    while (!stack.empty()) {
      const u32 i = stack.back(); stack.pop_back();
      if (seen[i]) continue;
      seen[i] = true;
      const NodeInfo& nd = plan->nodes[i];
      if (nd.d.kind == RDFGPU_NODE_DATA_SOURCE || nd.d.kind == RDFGPU_NODE_TABLE) continue;
      const bool binary = nd.d.kind == RDFGPU_NODE_HASH_JOIN || nd.d.kind == RDFGPU_NODE_CROSS_JOIN || nd.d.kind == RDFGPU_NODE_NESTED_LOOP_JOIN || nd.d.kind == RDFGPU_NODE_UNION;
      if (nd.d.left >= 0) { plan->nodes[nd.d.left].refs++; stack.push_back((u32)nd.d.left); }
      if (binary && nd.d.right >= 0) { plan->nodes[nd.d.right].refs++; stack.push_back((u32)nd.d.right); }
    }
  }
  for (u32 i = 0; i < plan->nodes.size(); i++) {   // the one consumer of a node consumed once
    const NodeInfo& nd = plan->nodes[i];
    if (nd.d.kind == RDFGPU_NODE_DATA_SOURCE || nd.d.kind == RDFGPU_NODE_TABLE) continue;
    const bool binary = nd.d.kind == RDFGPU_NODE_HASH_JOIN || nd.d.kind == RDFGPU_NODE_CROSS_JOIN || nd.d.kind == RDFGPU_NODE_NESTED_LOOP_JOIN || nd.d.kind == RDFGPU_NODE_UNION;
    if (i != plan->root && nd.refs == 0) continue;   // (rewritten away: not an operator of this plan any more)
    if (nd.d.left >= 0 && plan->nodes[nd.d.left].refs == 1) plan->nodes[nd.d.left].parent = (int)i;
    if (binary && nd.d.right >= 0 && plan->nodes[nd.d.right].refs == 1) plan->nodes[nd.d.right].parent = (int)i;
  }
  // per-node byte accounting inputs: distinct columns read, typed gathers per row
  for (NodeInfo& nd : plan->nodes) {
    bool used[2 * kMaxCols] = {};
    for (u32 i = 0; i < nd.prog.n; i++) {
      if (nd.prog.nodes[i].op == RDFGPU_EX_COLUMN) used[nd.prog.nodes[i].u] = true;
      if (nd.prog.nodes[i].op == RDFGPU_EX_ENC_TV) nd.n_enc_tv++;
    }
    if (nd.d.kind == RDFGPU_NODE_FILTER) for (u32 c = 0; c < nd.n_proj; c++) used[nd.proj[c]] = true;
    for (bool b : used) nd.n_cols_read += b;
  }

  static_assert(sizeof(LocateJob) <= 128, "ExecContext staging assumes LocateJob <= 128 bytes");
  plan->ctx = store->acquire_context((u32)plan->sources.size());
  plan->stream = plan->ctx->stream;
  plan->counters = plan->ctx->counters;
  plan->upload_pool();
  return plan.release();
}

// The scan of a pattern from its G,S,P,O instructions: index choice (IndexPermutations::choose_index, permutations.rs:81-96),
// the instructions in that index's order, the pruning levels and what they make redundant, the output columns in G,S,P,O
// order of first binding (patterns/mod.rs:68-107).
void Plan::derive_source(SourceInfo& src, const ScanInstructions& gspo) {
  src.components = choose_index(gspo, 0b111);
  src.ix = reorder(gspo, src.components);
  src.prune = plan_pruning(src.ix);
  src.n_out = 0; src.has_residual = false;
  for (int k = 0; k < 4; k++) {
    if (gspo.in[k].kind != RDFGPU_SCAN) continue;
    bool first = true;
    for (int j = 0; j < k; j++) if (gspo.in[j].kind == RDFGPU_SCAN && gspo.in[j].var == gspo.in[k].var) first = false;
    if (!first) continue;
    for (int lvl = 0; lvl < 4; lvl++)
      if (src.ix.in[lvl].kind == RDFGPU_SCAN && src.ix.in[lvl].var == gspo.in[k].var) src.out_level[src.n_out] = lvl;
    src.n_out++;
  }
  for (int k = 0; k < 4; k++)
    if (src.ix.in[k].pred.kind != RDFGPU_PRED_NONE && !(src.prune.dropped_mask & (1u << k))) src.has_residual = true;
}
// IN sets of residual predicates live on the device: uploaded at compile time and again when push-down changed them
void Plan::upload_pool() {
  pool.clear();
  for (SourceInfo& s : sources)
    for (int k = 0; k < 4; k++)
      if (s.ix.in[k].pred.kind == RDFGPU_PRED_IN && !(s.prune.dropped_mask & (1u << k))) {
        s.ix.in[k].pred.from = (u32)pool.size();
        pool.insert(pool.end(), s.ix.in[k].pred.ids.begin(), s.ix.in[k].pred.ids.end());
      }
  store->activate();
  if (stream) RDFGPU_HIP(hipStreamSynchronize(stream));
  if (pool_dev) { RDFGPU_HIP(hipFree(pool_dev)); pool_dev = nullptr; }
  if (!pool.empty()) {
    RDFGPU_HIP(hipMalloc((void**)&pool_dev, pool.size() * 4));
    RDFGPU_HIP(hipMemcpy(pool_dev, pool.data(), pool.size() * 4, hipMemcpyHostToDevice));
  }
}

namespace {
// MemStoragePredicateExpr -> MemIndexScanPredicate (to_scan_predicate, predicate_pushdown.rs:120-157); false = `true` (no predicate)
bool filter_to_predicate(const rdfgpu_pushdown_filter& f, ScanPredicate* out) {
  if (f.kind == RDFGPU_PUSH_TRUE) return false;
  if (f.kind == RDFGPU_PUSH_BINARY) { *out = pushdown_to_scan_predicate(f.op, f.value); return true; }
  if (f.kind == RDFGPU_PUSH_BETWEEN) {
    ScanPredicate p;
    if (f.from > f.to) p.kind = RDFGPU_PRED_FALSE; else { p.kind = RDFGPU_PRED_BETWEEN; p.from = f.from; p.to = f.to; }
    *out = p; return true;
  }
  fail(RDFGPU_ERR_INVALID, "push-down filter of kind %u", f.kind);
}
// MemIndexScanInstructions::apply_filter (scan_instructions.rs:101-133)
void and_into_instructions(ScanInstructions& gspo, u32 var, const ScanPredicate& p) {
  int at = -1;
  for (int k = 0; k < 4 && at < 0; k++) if (gspo.in[k].kind == RDFGPU_SCAN && gspo.in[k].var == var) at = k;   // instructions_for_column: the first binding
  if (at < 0) fail(RDFGPU_ERR_INVALID, "Could not find scan instruction for column: variable %u", var);
  ScanPredicate combined = p;
  if (gspo.in[at].pred.kind != RDFGPU_PRED_NONE && !predicate_and(gspo.in[at].pred, p, &combined))
    fail(RDFGPU_ERR_INVALID, "Could not apply predicate to scan instruction.");
  gspo.in[at].pred = combined;
}
}  // namespace

void Plan::pushdown_filters(u32 node, const rdfgpu_pushdown_filter* filters, u32 n, u8* pushed) {
  if (node >= nodes.size() || nodes[node].source < 0) fail(RDFGPU_ERR_INVALID, "node %u is not a data source", node);
  SourceInfo& src = sources[nodes[node].source];
  ScanInstructions gspo = src.gspo;
  bool any = false;
  for (u32 i = 0; i < n; i++) {
    const bool yes = filters[i].kind != RDFGPU_PUSH_UNSUPPORTED;   // rewritten => PushedDown::Yes (pattern_data_source.rs:121-127)
    if (pushed) pushed[i] = yes ? 1 : 0;
    if (!yes) continue;
    any = true;
    ScanPredicate p;
    if (filter_to_predicate(filters[i], &p)) and_into_instructions(gspo, filters[i].var, p);
  }
  if (!any) return;                       // "Don't create a new node if no filters were pushed down"
  src.gspo = gspo;
  derive_source(src, src.gspo);           // apply_pushdown_filters ends in try_find_better_index
  src.dynamic_dirty = !src.dynamic.empty();
  upload_pool();
  located_version = ~0ull;                // the cached ranges belong to the old instructions
  for (NodeInfo& nd : nodes) { nd.has_last = false; nd.last_rows = 0; }   // and so do the cardinalities
}

void Plan::set_dynamic_filters(u32 node, const rdfgpu_pushdown_filter* filters, u32 n) {
  if (node >= nodes.size() || nodes[node].source < 0) fail(RDFGPU_ERR_INVALID, "node %u is not a data source", node);
  SourceInfo& src = sources[nodes[node].source];
  std::vector<std::pair<u32, ScanPredicate>> dyn;
  for (u32 i = 0; i < n; i++) {
    if (filters[i].kind == RDFGPU_PUSH_UNSUPPORTED) continue;      // `current_predicate_expr().ok()`: unsupported ones are skipped (scan.rs:246-249)
    ScanPredicate p;
    if (filter_to_predicate(filters[i], &p)) dyn.emplace_back(filters[i].var, p);
  }
  // validate now, against the static instructions, so that execute cannot fail on them
  ScanInstructions probe = src.gspo;
  for (auto& d : dyn) and_into_instructions(probe, d.first, d.second);
  src.dynamic = dyn;
  src.dynamic_dirty = true;
}

Plan::~Plan() {
  if (store) (void)hipSetDevice(store->device);
  if (stream) (void)hipStreamSynchronize(stream);
  release_intermediates();
  if (pool_dev) (void)hipFree(pool_dev);
  if (regex_dev) (void)hipFree(regex_dev);
  if (str_consts_dev) (void)hipFree(str_consts_dev);

  if (store && ctx) store->release_context(ctx);
  if (store) store->release();
}

// which scan an n-element scan of counts takes (kernels.hip: one workgroup up to kSmallScanElems elements, rocPRIM's device scan beyond)
static int scan_class(u64 n) { return n <= kSmallScanElems ? KC_SMALL_SCAN : KC_DEVICE_SCAN; }
// Names as rocprofv3 --kernel-trace prints them (prefix up to the argument list).
const char* kernel_class_name(int kc) {
  static const char* const fixed[KC_LDS_JOIN0] = {
      "rdfgpu::locate_kernel", "rdfgpu::scan_count_kernel", "rdfgpu::scan_write_kernel",
      "void rdfgpu::filter_kernel<1>", "void rdfgpu::filter_kernel<2>", "void rdfgpu::filter_kernel<0>",
      "rdfgpu::cross_kernel", "rdfgpu::join_build_kernel", "void rdfgpu::join_probe_kernel<false>",
      "void rdfgpu::join_probe_kernel<true>", "rdfgpu::join_left_unmatched_kernel", "void rdfgpu::nlj_kernel<false>",
      "void rdfgpu::nlj_kernel<true>", "rocprim device scan", "rdfgpu::gjoin_build_kernel", "rdfgpu::gdirect_build_kernel",
      "rdfgpu::minmax_u32_kernel", "rdfgpu::csr_rel_keys_kernel", "rocprim radix sort (CSR rows)",
      "rdfgpu::topk_max_kernel", "rdfgpu::topk_hist_kernel", "rdfgpu::topk_scatter_kernel", "rdfgpu::topk_select_kernel",
      "rdfgpu::topk_write_kernel", "void rdfgpu::filter_kernel<3>", "rdfgpu::regex_verdict_kernel", "rdfgpu::union_kernel",
      "rdfgpu::band_slow_kernel", "rocprim radix sort", "rdfgpu::band_bounds_kernel", "rdfgpu::band_blocks_kernel",
      "rdfgpu::band_decode_kernel", "void rdfgpu::band_mask_kernel", "void rdfgpu::band_emit_kernel", "rdfgpu::band_entries_kernel",
      "rdfgpu::band_desc_kernel", "rdfgpu::band_pt_kernel", "rdfgpu::band_rows_kernel",
      "void rdfgpu::filter_bits_kernel<1>", "void rdfgpu::filter_bits_kernel<2>", "void rdfgpu::filter_bits_kernel<3>", "void rdfgpu::filter_bits_kernel<4>", "rdfgpu::value_verdict_kernel",
      "rdfgpu::value_runs_kernel", "void rdfgpu::run_scan_kernel", "rdfgpu::run_copy_kernel",
      "rdfgpu::oj_probe_kernel", "rdfgpu::oj_count_kernel", "void rdfgpu::oj_write_kernel",
      "void rdfgpu::filter_write_kernel", "rdfgpu::part_keys_kernel", "void rdfgpu::part_join_kernel",
      "rdfgpu::oj_band_records_kernel", "rdfgpu::oj_write_band_kernel", "void rdfgpu::small_scan_kernel",
      "rdfgpu::part_pass (hist + scan + scatter)", "void rdfgpu::stream_join_kernel"};
  if (kc < KC_LDS_JOIN0) return fixed[kc];
  static std::string names[192];
  static std::once_flag once;
  std::call_once(once, [] {
    const char* items[2] = {"4", "1"};
    for (int f = 0; f < 4; f++) for (int p = 0; p < 3; p++) for (int w = 0; w < 2; w++) for (int m = 0; m < 4; m++) for (int c = 0; c < 2; c++)
      names[(((f * 3 + p) * 2 + w) * 4 + m) * 2 + c] = "void rdfgpu::lds_join_kernel<" + std::to_string(f) + ", " + std::to_string(p) + ", " +
                                                       items[w] + ", " + std::to_string(m) + ", " + (c ? "true" : "false");   // (a prefix: the key-count argument follows)
  });
  return names[kc - KC_LDS_JOIN0].c_str();
}

template <class F>
void Plan::timed(int kc, u64 fixed_bytes, u64 rows_cap, const u64* rows_dev, u64 bytes_per_row,
                 const u64* out_dev, u64 out_rows, u64 bytes_per_out, F&& launch) {
  metrics.kernels_launched++;
  if (!timing || (timing_focus >= 0 && kc != timing_focus)) { launch(); return; }
  PendingLaunch p{kc, ctx->event(events_used), ctx->event(events_used + 1), fixed_bytes, rows_cap, rows_dev, bytes_per_row, out_dev, out_rows, bytes_per_out};
  events_used += 2;
  RDFGPU_HIP(hipEventRecord(p.start, stream));
  launch();
  RDFGPU_HIP(hipEventRecord(p.stop, stream));
  pending.push_back(p);
}

// After the final sync: event durations + byte counts (device-side cardinalities come from the
// counters mirror copied back with the result count).
void Plan::resolve_timing() {
  for (KernelStat& k : kstats) k = KernelStat{};
  if (!timing) return;
  auto live = [&](const u64* dev, u64 fallback) -> u64 {
    if (!dev) return fallback;
    const u64 v = ctx->counters_host[dev - counters];
    return v < fallback || fallback == 0 ? v : fallback;
  };
  for (const PendingLaunch& p : pending) {
    float ms = 0;
    RDFGPU_HIP(hipEventElapsedTime(&ms, p.start, p.stop));
    KernelStat& k = kstats[p.kc];
    const u64 rows = live(p.rows_dev, p.rows_cap);
    const u64 out = p.out_dev ? ctx->counters_host[p.out_dev - counters] : p.out_rows;
    k.launches++; k.ms += ms; k.rows += rows;
    k.bytes += p.fixed_bytes + rows * p.bytes_per_row + out * p.bytes_per_out;
  }
  pending.clear();
  if (timing_focus < 0) {   // every launch was timed: remember the class that took longest (rdfgpu_plan_enable_kernel_timing(plan, 2))
    double top = 0;
    for (int kc = 0; kc < (int)(sizeof kstats / sizeof kstats[0]); kc++) if (kstats[kc].ms > top) { top = kstats[kc].ms; last_top_kc = kc; }
  }
}

// ------------------------------------------------------------------------------------------------
// execute
// ------------------------------------------------------------------------------------------------
void Plan::release_intermediates() {
  for (void* p : allocs) store->pool.free(p);
  allocs.clear();
}
template <class T> T* Plan::scratch(u64 n) {
  void* p = store->pool.alloc((n ? n : 1) * sizeof(T));
  allocs.push_back(p);
  metrics.device_bytes += (n ? n : 1) * sizeof(T);
  return (T*)p;
}
// Generic (VM) programs of the LDS join live in device memory; staged through the context's pinned slots.
const ExprProgram* Plan::upload_program(const ExprProgram& p) {
  if (progs_used >= ExecContext::kProgSlots) fail(RDFGPU_ERR_UNSUPPORTED, "plan needs more than %u device expression programs", ExecContext::kProgSlots);
  const u32 slot = progs_used++;
  ctx->progs_host[slot] = p;
  RDFGPU_HIP(hipMemcpyAsync(ctx->progs_dev + slot, ctx->progs_host + slot, sizeof(ExprProgram), hipMemcpyHostToDevice, stream));
  return ctx->progs_dev + slot;
}
u64* Plan::new_counter() {
  if (counters_used >= 255) fail(RDFGPU_ERR_UNSUPPORTED, "plan needs more than 255 cardinality counters");   // slot 255: run-time error flags
  return counters + counters_used++;
}
u64 Plan::read_u64(const u64* dev) {
  u64 v = 0;
  RDFGPU_HIP(hipMemcpyAsync(&v, dev, sizeof v, hipMemcpyDeviceToHost, stream));
  RDFGPU_HIP(hipStreamSynchronize(stream));
  metrics.host_syncs++;
  return v;
}

// First execution of a plan over big caller-supplied tables (a batch of query instances): nothing is known yet, so every
// join would be sized exactly and run un-fused — on the full batch that materialises every candidate pair (126 GiB of
// intermediates for 262 144 BSBM Q5 instances).  Instead the plan first runs over the first kPrimeRows rows of each
// bound table: that builds every join table of the store slices (cached per store version) and leaves cardinalities,
// which are extrapolated by the row ratio; the full batch then takes the speculative, fused path at once.  A wrong
// extrapolation is caught like any failed speculation (overflow flag -> exact re-run).
constexpr u64 kPrimeRows = 2048;
void Plan::prime() {
  primed = true;
  if (!allow_speculation || opt.on(RDFGPU_OPT_NO_SPECULATION) || opt.on(RDFGPU_OPT_NO_PRIMING) || opt.on(RDFGPU_OPT_NO_TABLE_CACHE) || opt.on(RDFGPU_OPT_NO_CHAIN_FUSION)) return;
  u64 full = 0;
  for (const BoundTable& b : tables) if (b.bound) full = std::max(full, b.n_rows);
  if (full < 16 * kPrimeRows) return;
  for (const NodeInfo& nd : nodes) if (nd.has_last) return;
  std::vector<u64> saved(tables.size());
  for (size_t i = 0; i < tables.size(); i++) { saved[i] = tables[i].n_rows; tables[i].n_rows = std::min<u64>(tables[i].n_rows, kPrimeRows); }
  priming = true;
  try { execute(); } catch (...) { priming = false; for (size_t i = 0; i < tables.size(); i++) tables[i].n_rows = saved[i]; throw; }
  priming = false;
  for (size_t i = 0; i < tables.size(); i++) tables[i].n_rows = saved[i];
  // operators above a bound table scale with it; pure store subtrees keep their exact history
  std::vector<int> dep(nodes.size(), -1);
  std::function<bool(u32)> depends = [&](u32 i) -> bool {
    if (dep[i] >= 0) return dep[i] != 0;
    const NodeInfo& nd = nodes[i];
    bool d = false;
    if (nd.d.kind == RDFGPU_NODE_TABLE) d = true;
    else if (nd.d.kind != RDFGPU_NODE_DATA_SOURCE) {
      const bool binary = nd.d.kind == RDFGPU_NODE_HASH_JOIN || nd.d.kind == RDFGPU_NODE_CROSS_JOIN || nd.d.kind == RDFGPU_NODE_NESTED_LOOP_JOIN || nd.d.kind == RDFGPU_NODE_UNION;
      if (nd.d.left >= 0) d = depends((u32)nd.d.left);
      if (binary && nd.d.right >= 0) d = depends((u32)nd.d.right) || d;
    }
    dep[i] = d ? 1 : 0;
    return d;
  };
  const u64 ratio = (full + kPrimeRows - 1) / kPrimeRows;
  for (u32 i = 0; i < nodes.size(); i++) {
    NodeInfo& nd = nodes[i];
    if (nd.has_last && depends(i)) { nd.last_rows = nd.last_rows * ratio + 1024; nd.last_scaled = true; }
  }
}

void Plan::execute() {
  if (!primed && !priming) prime();
  store->activate();
  std::shared_lock<std::shared_mutex> lock(store->mu);   // a plan holds the snapshot while it runs (snapshot.rs:35-37)
  RDFGPU_HIP(hipStreamSynchronize(stream));
  release_intermediates();
  // the pool keeps what an execution hands back for the next one — bounded by twice what the previous executions (this plan's, or any plan's of the store) used
  // (+ 1 GiB): the giant blocks of a one-off exact run go back to the device instead of staying cached for ever
  store->pool.trim_to(2 * std::max(std::max(scratch_hist[0], scratch_hist[1]), std::max(store->scratch_recent[0].load(), store->scratch_recent[1].load())) + (1ull << 30));
  held = store->gen;         // ... and the generation it read until its next execute: the result may be zero-copy slices of it
  metrics = rdfgpu_metrics{};
  const u64 mallocs0 = store->pool.mallocs() + store->table_pool.mallocs();
  const double malloc_ms0 = store->pool.malloc_ms() + store->table_pool.malloc_ms();
  counters_used = 0;
  progs_used = 0;
  arg_slots_used = 0;
  events_used = 2;   // events 0/1 bracket the whole execute
  pending.clear();
  host_valid = false; cursor = 0; executed = false;
  const hipEvent_t ev_start = ctx->event(0), ev_stop = ctx->event(1);
  RDFGPU_HIP(hipEventRecord(ev_start, stream));
  RDFGPU_HIP(hipMemsetAsync(counters, 0, 256 * sizeof(u64), stream));

  spec_checks.clear();
  pending_oj.active = false;
  band_block_counters.clear();
  memo.assign(nodes.size(), DevTable{}); memo_valid.assign(nodes.size(), 0);
  speculative = allow_speculation && !opt.on(RDFGPU_OPT_NO_SPECULATION);

  // Dynamic filters (scan.rs:217-261): the effective instructions of a leaf = its static ones AND the filters' current
  // predicates, index re-chosen for them; derived when the filters changed, the ranges located again.
  {
    bool changed = false;
    for (SourceInfo& s : sources) {
      if (!s.dynamic_dirty) continue;
      ScanInstructions eff = s.gspo;
      for (auto& dflt : s.dynamic) and_into_instructions(eff, dflt.first, dflt.second);
      derive_source(s, eff);
      s.dynamic_dirty = false;
      changed = true;
    }
    if (changed) {
      lock.unlock();
      upload_pool();
      lock.lock();
      located_version = ~0ull;
      for (NodeInfo& nd : nodes) { nd.has_last = false; nd.last_rows = 0; }
      speculative = false;
    }
  }
  // K1: locate every data source's range in one launch, one host round trip for all of them.  The ranges
  // depend only on the plan's constants and the store's content: a re-execution on an unchanged store
  // reuses them (no launch, no sync).
  if (!sources.empty() && located_version == store->version.load()) {
    for (const SourceInfo& s : sources) metrics.input_rows += s.hi - s.lo;
  } else if (!sources.empty()) {
    located_version = store->version.load();
    LocateJob* jobs = static_cast<LocateJob*>(ctx->jobs_host);
    for (size_t i = 0; i < sources.size(); i++) {
      const SourceInfo& s = sources[i];
      const Permutation& ix = store->idx[s.components];
      LocateJob& j = jobs[i];
      for (int k = 0; k < 4; k++) j.col[k] = ix.col[k];
      j.n = ix.n;
      j.n_levels = s.prune.n_levels;
      for (int k = 0; k < 4; k++) { j.from[k] = s.prune.from[k]; j.to[k] = s.prune.to[k]; }
    }
    LocateJob* jobs_dev = static_cast<LocateJob*>(ctx->jobs_dev);
    RDFGPU_HIP(hipMemcpyAsync(jobs_dev, jobs, sources.size() * sizeof(LocateJob), hipMemcpyHostToDevice, stream));
    timed(KC_LOCATE, 0, sources.size(), nullptr, 0, nullptr, 0, 0, [&] { launch_locate(jobs_dev, (u32)sources.size(), ctx->lohi_dev, stream); });
    RDFGPU_HIP(hipMemcpyAsync(ctx->lohi_host, ctx->lohi_dev, sources.size() * kLocateWords * sizeof(u64), hipMemcpyDeviceToHost, stream));
    RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
    for (size_t i = 0; i < sources.size(); i++) {
      const u64* w = ctx->lohi_host + kLocateWords * i;
      SourceInfo& s = sources[i];
      s.lo = w[0]; s.hi = w[1]; s.sorted_level = (u32)(w[2] >> 32); s.key_min = (u32)w[2]; s.key_max = (u32)w[3];
      metrics.input_rows += s.hi - s.lo;
    }
  }

  result = exec_node(root);
  flush_pending_oj();
  // one copy brings back every device-side cardinality (the result's and, for timing, the others')
  RDFGPU_HIP(hipMemcpyAsync(ctx->counters_host, counters, 256 * sizeof(u64), hipMemcpyDeviceToHost, stream));
  RDFGPU_HIP(hipEventRecord(ev_stop, stream));
  RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
  RDFGPU_HIP(hipGetLastError());
  if (const u32 rt = (u32)(ctx->counters_host[255] & 0xFFFFFFFFull)) {   // a row asked for something that is refused loudly, not answered differently
    if (rt & 1u) fail(RDFGPU_ERR_UNSUPPORTED, "REGEX with \\d \\w \\s or \\b over a string with non-ASCII characters needs the regex crate's Unicode tables (not restated)");
    if (rt & 4u) fail(RDFGPU_ERR_UNSUPPORTED, "a string expression met what the device does not restate: UCASE / LCASE of a string with non-ASCII characters (Unicode case tables), "
                                               "a float / double / decimal SUBSTR position, or a comparison with a string that has no bytes on the device");
    fail(RDFGPU_ERR_UNSUPPORTED, "REGEX with a per-row pattern: a row's pattern literal was not announced in the plan's pattern table");
  }
  bool slow_missed = false;
  for (const BandBlockCounter& c : band_block_counters) {
    if (!c.node) continue;
    c.node->band_blocks = ctx->counters_host[c.counter];
    c.node->band_slow_rows = ctx->counters_host[c.slow_counter] & 0xFFFFFFFFull;
    c.node->band_run_stats = ctx->counters_host[c.runs_counter];
    c.node->band_ran = true;
    if (c.slow_skipped && c.node->band_slow_rows) slow_missed = true;   // rows with non-integer operands, and their pass was not launched
  }
  // speculative joins: did every output fit the size taken from the previous run?
  bool spec_failed = slow_missed;
  for (const SpecCheck& c : spec_checks) {
    if ((ctx->counters_host[c.counter + 1] & 0xFFFFFFFFull) != 0) spec_failed = true;
    else { c.node->last_rows = ctx->counters_host[c.counter]; c.node->has_last = true; c.node->last_scaled = false; }
  }
  if (spec_failed) {   // rare: run again with exact sizes (one sync per join), then speculate again next time
    metrics.host_syncs++;
    lock.unlock();
    allow_speculation = false;
    try { execute(); } catch (...) { allow_speculation = true; throw; }
    allow_speculation = true;
    metrics.exact_reruns = 1;
    return;
  }
  result_rows = result.n_dev ? ctx->counters_host[result.n_dev - counters] : result.cap;
  if (result_rows > result.cap) result_rows = result.cap;
  float ms = 0;
  RDFGPU_HIP(hipEventElapsedTime(&ms, ev_start, ev_stop));
  metrics.elapsed_compute_ms = ms;
  metrics.output_rows = result_rows;
  metrics.device_mallocs = (u32)(store->pool.mallocs() + store->table_pool.mallocs() - mallocs0);
  metrics.device_malloc_ms = store->pool.malloc_ms() + store->table_pool.malloc_ms() - malloc_ms0;
  resolve_timing();
  executed = true;
  scratch_hist[1] = scratch_hist[0]; scratch_hist[0] = metrics.device_bytes;
  store->scratch_recent[1] = store->scratch_recent[0].load(); store->scratch_recent[0] = metrics.device_bytes;
}

DevTable Plan::exec_node(u32 idx) {
  if (memo_valid[idx]) return memo[idx];   // a node runs once per execution, however many operators consume it
  NodeInfo& nd = nodes[idx];
  DevTable t;
  switch (nd.d.kind) {
    case RDFGPU_NODE_DATA_SOURCE: t = exec_source(nd); break;
    case RDFGPU_NODE_FILTER: t = exec_filter(nd); break;
    case RDFGPU_NODE_PROJECTION: {
      const DevTable in = exec_node((u32)nd.d.left);
      t.n_cols = nd.n_proj; t.cap = in.cap; t.n_dev = in.n_dev;
      for (u32 c = 0; c < nd.n_proj; c++) t.cols[c] = in.cols[nd.proj[c]];
      break;
    }
    case RDFGPU_NODE_HASH_JOIN: case RDFGPU_NODE_CROSS_JOIN: case RDFGPU_NODE_NESTED_LOOP_JOIN: t = exec_join(nd); break;
    case RDFGPU_NODE_TOPK: t = exec_topk(nd); break;
    case RDFGPU_NODE_CLOSURE: {
      const DevTable in = exec_node((u32)nd.d.left);
      u64 n = in.cap;
      if (in.n_dev && in.cap) {
        RDFGPU_HIP(hipMemcpyAsync(&n, in.n_dev, sizeof(u64), hipMemcpyDeviceToHost, stream));
        RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
      }
      u32* out[3] = {nullptr, nullptr, nullptr};
      ClosureStats cs;
      const u64 rows = closure_exec(in.cols[0], in.cols[1], in.cols[2], n, nd.d.join_type == 1, stream, [&](u64 m) { return scratch<u32>(m); }, out, &cs);
      metrics.host_syncs += 4 + 3 * cs.iterations;
      t.n_cols = nd.n_proj; t.cap = rows; t.n_dev = nullptr;
      for (u32 c = 0; c < nd.n_proj; c++) t.cols[c] = out[nd.proj[c]];
      break;
    }
    case RDFGPU_NODE_UNION: {
      const DevTable L = exec_node((u32)nd.d.left), R = exec_node((u32)nd.d.right);
      t.n_cols = nd.n_proj;
      const u64 cap = L.cap + R.cap;
      if (cap >= 0xFFFFFFF0ull) fail(RDFGPU_ERR_UNSUPPORTED, "UnionExec of %llu rows", (unsigned long long)cap);
      if (cap == 0) { t.cap = 0; break; }
      UnionArgs a{};
      a.n_cols = nd.n_proj;
      for (u32 c = 0; c < nd.n_proj; c++) {
        a.left[c] = L.cap ? L.cols[nd.proj[c]] : nullptr; a.right[c] = R.cap ? R.cols[nd.proj[c]] : nullptr;
        a.out[c] = scratch<u32>(cap); t.cols[c] = a.out[c];
      }
      a.n_left_dev = L.n_dev; a.n_left_cap = L.cap; a.n_right_dev = R.n_dev; a.n_right_cap = R.cap;
      const bool dyn = L.n_dev || R.n_dev;
      a.n_out_dev = dyn ? new_counter() : nullptr;
      // bytes: every projected cell read once and written once
      timed(KC_UNION, R.n_dev ? 0 : 8ull * nd.n_proj * R.cap, L.cap, L.n_dev, 8ull * nd.n_proj, nullptr, 0, 0, [&] { launch_union(a, stream); });
      t.cap = cap; t.n_dev = a.n_out_dev;
      break;
    }
    case RDFGPU_NODE_TABLE: {
      const BoundTable& b = tables[nd.d.table_slot];
      if (!b.bound) fail(RDFGPU_ERR_INVALID, "table slot %u is not bound", nd.d.table_slot);
      if (b.cols.size() != nd.d.table_cols) fail(RDFGPU_ERR_INVALID, "table slot %u: %zu columns bound, node declares %u", nd.d.table_slot, b.cols.size(), nd.d.table_cols);
      t.n_cols = nd.d.table_cols; t.cap = b.n_rows;
      for (u32 c = 0; c < t.n_cols; c++) t.cols[c] = b.cols[c];
      break;
    }
    default: fail(RDFGPU_ERR_INVALID, "unknown node kind");
  }
  if (idx != root) metrics.intermediate_rows += t.cap;   // upper bound when the exact count stays on the device
  if (!(pending_chain && pending_chain->consumed)) { memo[idx] = t; memo_valid[idx] = 1; }   // (a fused chain's output belongs to its top node)
  return t;
}

// DataSourceExec: a prefix-bound pattern is a zero-copy slice of the permutation (like the
// reference's untouched row-group slices, scan.rs:146-170); residual predicates go through K2.
DevTable Plan::exec_source(NodeInfo& nd) {
  SourceInfo& s = sources[nd.source];
  const Permutation& ix = store->idx[s.components];
  DevTable t;
  t.n_cols = s.n_out;
  const u64 n = s.hi - s.lo;
  if (n == 0) { t.cap = 0; return t; }
  if (!s.has_residual) {
    for (u32 c = 0; c < s.n_out; c++) t.cols[c] = ix.col[s.out_level[c]] + s.lo;
    t.cap = n;
    t.stable_id = (u64)nd.source + 1;   // a pure slice of the store: identical on every execution until the store changes
    for (u32 c = 0; c < s.n_out; c++) if (s.out_level[c] == s.sorted_level && t.sorted_col < 0) { t.sorted_col = (int)c; t.key_min = s.key_min; t.key_max = s.key_max; }
    return t;
  }
  ScanJob job{};
  for (int k = 0; k < 4; k++) {
    job.col[k] = ix.col[k] + s.lo;
    const ScanPredicate& p = s.ix.in[k].pred;
    ScanLevelPred& q = job.pred[k];
    q.kind = (s.prune.dropped_mask & (1u << k)) ? (u32)RDFGPU_PRED_NONE : p.kind;
    if (q.kind == RDFGPU_PRED_BETWEEN) { q.a = p.from; q.b = p.to; }
    else if (q.kind == RDFGPU_PRED_IN) { q.b = (u32)p.ids.size(); q.ids = pool_dev + p.from; }
    else if (q.kind == RDFGPU_PRED_EQUAL_TO) {
      int other = -1;
      for (int l = 0; l < 4; l++) if (s.ix.in[l].kind == RDFGPU_SCAN && s.ix.in[l].var == p.equal_to) { other = l; break; }
      if (other < 0) q.kind = RDFGPU_PRED_NONE;   // `position(..)?` => no mask (scan.rs:310-313)
      else q.a = (u32)other;
    }
  }
  job.n = n;
  job.n_out = s.n_out;
  for (u32 c = 0; c < s.n_out; c++) job.out_level[c] = s.out_level[c];
  const u64 n_blocks = (n + kScanTile - 1) / kScanTile;
  u32* counts = scratch<u32>(n_blocks + 1);
  u32* offs = scratch<u32>(n_blocks + 1);
  const size_t tb = scan_temp_bytes(n_blocks + 1);
  void* temp = scratch<u8>(tb);
  RDFGPU_HIP(hipMemsetAsync(counts + n_blocks, 0, 4, stream));
  u32 n_pred_cols = 0;   // columns the residual predicates read
  for (int k = 0; k < 4; k++) n_pred_cols += job.pred[k].kind != RDFGPU_PRED_NONE && job.pred[k].kind != RDFGPU_PRED_FALSE;
  timed(KC_SCAN_COUNT, 0, n, nullptr, 4ull * n_pred_cols, nullptr, 0, 0, [&] { launch_scan_count(job, counts, stream); });
  timed(scan_class(n_blocks + 1), 0, n_blocks + 1, nullptr, 8, nullptr, 0, 0, [&] { exclusive_scan_u32(counts, offs, n_blocks + 1, temp, tb, stream); });
  u32 total = 0;
  RDFGPU_HIP(hipMemcpyAsync(&total, offs + n_blocks, 4, hipMemcpyDeviceToHost, stream));
  RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
  t.cap = total;
  if (total == 0) return t;
  for (u32 c = 0; c < s.n_out; c++) { job.out[c] = scratch<u32>(total); t.cols[c] = job.out[c]; }
  // scan+filter+compact: 4·c_r·N + 4·c_w·σN, c_r = predicate ∪ output columns (upper bound: both)
  if (s.n_out) timed(KC_SCAN_WRITE, 0, n, nullptr, 4ull * n_pred_cols, nullptr, total, 8ull * s.n_out, [&] { launch_scan_write(job, offs, stream); });
  return t;
}

DevTable Plan::exec_filter(NodeInfo& nd) {
  const DevTable in = exec_node((u32)nd.d.left);
  return apply_filter(nd, in);
}

DevTable Plan::apply_filter(NodeInfo& nd, const DevTable& in) {
  DevTable t;
  t.n_cols = nd.n_proj;
  if (in.cap == 0) { t.cap = 0; return t; }
  if (nd.prog.n == 0) {   // no predicate: a projection
    t.cap = in.cap; t.n_dev = in.n_dev;
    for (u32 c = 0; c < nd.n_proj; c++) t.cols[c] = in.cols[nd.proj[c]];
    return t;
  }
  FilterArgs a{};
  for (u32 c = 0; c < in.n_cols; c++) a.in[c] = in.cols[c];
  a.n_in_cols = in.n_cols; a.n_out_cols = nd.n_proj;
  for (u32 c = 0; c < nd.n_proj; c++) { a.proj[c] = nd.proj[c]; a.out[c] = scratch<u32>(in.cap); t.cols[c] = a.out[c]; }
  a.n_in_dev = in.n_dev; a.n_in_cap = in.cap;
  a.n_out_dev = new_counter();
  a.tt = typed_table();
  a.prog = nd.prog;
  int shape = nd.shape;
  if (shape == 3) {
    // per-distinct-term verdicts: worth a pass over the dictionary when the table has at least a quarter as many rows
    // as there are ids (or the verdicts exist already); the table lives on the store, keyed by the predicate
    const rdfgpu_expr_node& e = nd.prog.nodes[2];
    const rdfgpu_regex& rx = regex_text[e.u];
    std::string key(1, (char)e.op);
    key.append(reinterpret_cast<const char*>(&e.lo), sizeof e.lo);
    key.append(rx.flags ? std::string(rx.flags, rx.flags_len) : std::string()).push_back('\0');
    key.append(rx.pattern ? std::string(rx.pattern, rx.pattern_len) : std::string());
    const u64 n_ids = std::min<u64>(store->n_ids, store->n_str_ids);
    unsigned char* verdict = nullptr;
    if (!opt.on(RDFGPU_OPT_NO_STRING_VERDICTS) && n_ids > 0) {
      std::unique_lock<std::mutex> building(store->slice_build_mu);
      { std::lock_guard<std::mutex> l(store->slice_mu); auto it = store->string_verdicts.find(key); if (it != store->string_verdicts.end()) verdict = it->second; }
      if (!verdict && in.cap * 4 >= n_ids) {
        // bounded cache: a workload of ever-changing patterns must not pile up one table per pattern — beyond 64
        // entries the table is this execution's scratch
        bool cache_it;
        { std::lock_guard<std::mutex> l(store->slice_mu); cache_it = store->string_verdicts.size() < 64; }
        if (cache_it) RDFGPU_HIP(hipMalloc((void**)&verdict, n_ids)); else verdict = scratch<unsigned char>(n_ids);
        const int64_t lang = e.op == RDFGPU_EX_REGEX ? -1 : (e.lo < 0 ? 0 : e.lo);
        timed(KC_REGEX_VERDICTS, 0, n_ids, nullptr, 16 + 8 + 1, nullptr, 0, 0, [&] { TypedTable vt = a.tt; vt.rt_error = nullptr;   // a verdict pass covers the whole dictionary: what it cannot answer is verdict 3, an error only for a row that reads it
                                                                                               launch_regex_verdicts(regex_dev + e.u, vt, lang, verdict, n_ids, stream); });
        if (cache_it) {
          RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;   // complete before other plans may see it
          std::lock_guard<std::mutex> l(store->slice_mu);
          store->string_verdicts[key] = verdict;
        }
      }
    }
    if (verdict) { a.verdict = verdict; a.n_verdict = n_ids; } else shape = 0;
  }
  // a typed comparison on the sorted column of a big store slice with few distinct ids: the qualifying runs are copied, the
  // predicate column is not streamed at all
  if (shape == 2 && !opt.on(RDFGPU_OPT_NO_VALUE_VERDICTS) && !opt.on(RDFGPU_OPT_NO_RUN_COPY) && in.sorted_col >= 0 && (u32)in.sorted_col == nd.prog.nodes[0].u &&
      in.key_max >= in.key_min && !in.n_dev && in.cap >= (1ull << 20) && in.cap < (1ull << 32) && nd.n_proj <= 2) {
    const u64 span = (u64)in.key_max - in.key_min + 1;
    if (span <= kRunCopyMaxIds && span * 1024 <= in.cap) {
      a.value_min = in.key_min; a.value_span = span;
      a.stream_bits = reinterpret_cast<unsigned short*>(scratch<u32>(1)); a.stream_counts = scratch<u32>(1); a.stream_offs = a.stream_counts;   // (the argument block wants them non-null)
      // Where every id's run starts is a function of the slice alone: kept with the slice's other tables per store version (the searches
      // that find them are five dependent HBM round trips per id: 10 of the operator's 68 us); with the table at hand the comparison of
      // every id is answered in the scan kernel — one launch plans the copy.
      const u32* pcol = in.cols[in.sorted_col];
      const bool cacheable = in.stable_id != 0 && !opt.on(RDFGPU_OPT_NO_TABLE_CACHE);
      u32* cached_lo = nullptr;
      SliceTable* vst = nullptr;
      std::unique_lock<std::mutex> building(store->slice_build_mu, std::defer_lock);
      if (cacheable) {
        SliceKey sk; sk.n_keys = 1; sk.rows = in.cap; sk.key[0] = pcol;
        vst = store->slice_table(sk);
        building.lock();
        for (const auto& v : vst->value_starts) if (v.first == in.key_min && v.span == span) cached_lo = v.lo;
      }
      const bool own = cacheable && !cached_lo && vst->value_starts.size() < 4;
      u32* run_lo = cached_lo ? cached_lo : own ? store->table_alloc<u32>(span + 1) : scratch<u32>(span + 1);   // (one entry past the last id)
      RunCopyBuffers b{run_lo, scratch<u32>(span), scratch<u32>(span + 1), scratch<u32>(span + 1), scratch<u32>(span + 1), scratch<u32>(1)};
      if (!cached_lo) timed(KC_VALUE_RUNS, 0, span, nullptr, 16, nullptr, 0, 0, [&] { launch_value_runs(a, b, stream); });
      // (one launch for scan + copy — 2048 workgroups that each scan the run lengths in LDS and copy an equal share — was tried: 68 us
      //  against 6 + 51: the big chunks do not hide their memory latency the way 16 K small workgroups do)
      timed(KC_RUN_SCAN, 0, span, nullptr, cached_lo ? 16 + 8 : 8, nullptr, 0, 0, [&] { launch_run_scan(a, b, cached_lo != nullptr, stream); });
      if (own) {   // publish only when complete
        RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++; metrics.tables_built++;
        vst->value_starts.push_back(SliceTable::ValueStarts{in.key_min, span, run_lo});
      }
      if (building.owns_lock()) building.unlock();
      timed(KC_RUN_COPY, 0, 0, nullptr, 0, a.n_out_dev, 0, 8ull * nd.n_proj, [&] { launch_run_copy(a, b, stream); });
      t.cap = in.cap; t.n_dev = a.n_out_dev;
      return t;
    }
  }
  // FilterExec bytes: 4·c_r·N + t·N + 4·c_w·σN with t = 9 B per typed gather (tag + i64); shape 3: t = 1 B
  if (filter_streams(a, shape)) {   // two passes without atomics: verdict bits + tile counts, device scan, ordered write
    const u64 tiles = filter_stream_tiles(a);
    a.stream_bits = scratch<unsigned short>(tiles * 256);
    a.stream_counts = scratch<u32>(tiles + 1); a.stream_offs = scratch<u32>(tiles + 1);
    a.stream_temp_bytes = scan_temp_bytes(tiles + 1);
    a.stream_temp = scratch<unsigned char>(a.stream_temp_bytes);
    RDFGPU_HIP(hipMemsetAsync(a.stream_counts + tiles, 0, sizeof(u32), stream));
    // compulsory bytes: pass 1 streams the predicate column (its typed-value gathers hit a table that is cache-resident or
    // not: not counted) and writes one bit per row; pass 2 reads the bits and the output columns and writes the survivors
    // a typed comparison over the sorted column of a store slice: answered once per id of the slice's id range when that
    // range is small next to the rows (a GPOS slice of one predicate: its objects), then one bit per row
    if (shape == 2 && !opt.on(RDFGPU_OPT_NO_VALUE_VERDICTS) && in.sorted_col >= 0 && (u32)in.sorted_col == nd.prog.nodes[0].u && in.key_max >= in.key_min) {
      const u64 span = (u64)in.key_max - in.key_min + 1;
      if (span * 4 <= in.cap) {
        u32* words = scratch<u32>(((span + 63) / 64) * 2);
        a.value_bits = words; a.value_min = in.key_min; a.value_span = span;
        timed(KC_VALUE_VERDICTS, 0, span, nullptr, 16, nullptr, 0, 0, [&] { launch_value_verdicts(a, stream); });
        shape = 4;
      }
    }
    const int kc1 = shape == 1 ? KC_FILTER_BITS_ID : shape == 2 ? KC_FILTER_BITS_TV : shape == 4 ? KC_FILTER_BITS_VALUE : KC_FILTER_BITS_VERDICT;
    timed(kc1, tiles * 4, in.cap, in.n_dev, 4, nullptr, 0, 0, [&] { launch_filter_bits(a, shape, stream); });
    timed(scan_class(tiles + 1), 0, tiles + 1, nullptr, 8, nullptr, 0, 0, [&] { exclusive_scan_u32(a.stream_counts, a.stream_offs, tiles + 1, a.stream_temp, a.stream_temp_bytes, stream); });
    timed(KC_FILTER_WRITE, tiles * 8, in.cap, in.n_dev, 4ull * nd.n_proj, a.n_out_dev, 0, 4ull * nd.n_proj, [&] { launch_filter_write(a, shape, stream); });
    t.cap = in.cap; t.n_dev = a.n_out_dev;
    return t;
  }
  const int kc = shape == 1 ? KC_FILTER_ID : shape == 2 ? KC_FILTER_TV : shape == 3 ? KC_FILTER_VERDICT : KC_FILTER_VM;
  timed(kc, 0, in.cap, in.n_dev, shape == 3 ? 4ull * nd.n_cols_read + 1 : 4ull * nd.n_cols_read + 9ull * nd.n_enc_tv, a.n_out_dev, 0, 4ull * nd.n_proj,
        [&] { launch_filter(a, shape, stream); });
  t.cap = in.cap; t.n_dev = a.n_out_dev;
  return t;
}

static u32 pow2_at_least(u64 v) { u64 p = 1024; while (p < v && p < (1ull << 31)) p <<= 1; return (u32)p; }

// DISTINCT + ORDER BY keys LIMIT k (per group): counting sort of row ids by group, one wave per group selecting the k
// smallest distinct key tuples, compaction by offsets.  Two host round trips (largest group id; final count + the
// "unsupported kind in a SORT_BY_TERM column" flag): this operator ends a query, it is not inside the join pipeline.
DevTable Plan::exec_topk(NodeInfo& nd) {
  const DevTable in = exec_node((u32)nd.d.left);
  DevTable t;
  t.n_cols = nd.n_proj;
  if (in.cap == 0) { t.cap = 0; return t; }
  if (in.cap >= (1ull << 32)) fail(RDFGPU_ERR_UNSUPPORTED, "TopK over %llu rows", (unsigned long long)in.cap);
  TopkArgs a{};
  for (u32 c = 0; c < in.n_cols; c++) a.in[c] = in.cols[c];
  a.n_in_dev = in.n_dev; a.n_in_cap = in.cap;
  a.has_group = nd.d.table_slot != 0; a.group_col = a.has_group ? nd.d.table_slot - 1 : 0;
  a.n_keys = nd.d.n_keys;
  for (u32 k = 0; k < a.n_keys; k++) { a.key_col[k] = nd.d.left_keys[k]; a.key_by_term[k] = nd.d.right_keys[k]; }   // RDFGPU_SORT_BY_*
  a.k = nd.d.table_cols;
  a.tt = typed_table();
  u64* n_out = new_counter();
  u32* flags = reinterpret_cast<u32*>(new_counter());   // {largest group id, unsupported-kind flag}
  a.n_out_dev = n_out; a.bad = flags + 1;
  a.n_groups = 1;
  if (a.has_group) {
    timed(KC_TOPK_MAX, 0, in.cap, in.n_dev, 4, nullptr, 0, 0, [&] { launch_topk_max(a.in[a.group_col], in.n_dev, in.cap, flags, stream); });
    u32 mx = 0;
    RDFGPU_HIP(hipMemcpyAsync(&mx, flags, sizeof(u32), hipMemcpyDeviceToHost, stream));
    RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
    if (mx >= (1u << 24)) fail(RDFGPU_ERR_UNSUPPORTED, "TopK: group ids up to %u (dense ids below 2^24 expected)", mx);
    a.n_groups = mx + 1;
  }
  const u64 ng = a.n_groups;
  a.counts = scratch<u32>(ng + 1); a.offsets = scratch<u32>(ng + 1); a.cursor = scratch<u32>(ng);
  a.perm = scratch<u32>(in.cap); a.picked = scratch<u32>(ng * a.k);
  a.out_counts = scratch<u32>(ng + 1); a.out_offsets = scratch<u32>(ng + 1);
  const u64 out_cap = std::min<u64>(in.cap, ng * a.k);
  a.n_out_cols = nd.n_proj;
  for (u32 c = 0; c < nd.n_proj; c++) { a.proj[c] = nd.proj[c]; a.out[c] = scratch<u32>(out_cap); t.cols[c] = a.out[c]; }
  RDFGPU_HIP(hipMemsetAsync(a.counts, 0, (ng + 1) * sizeof(u32), stream));
  RDFGPU_HIP(hipMemsetAsync(a.out_counts, 0, (ng + 1) * sizeof(u32), stream));
  const size_t tb = scan_temp_bytes(ng + 1);
  void* temp = scratch<unsigned char>(tb);
  u32 key_bytes = 4 * a.n_keys;
  for (u32 k = 0; k < a.n_keys; k++) key_bytes += a.key_by_term[k] ? 16 : 0;
  timed(KC_TOPK_HIST, 0, in.cap, in.n_dev, 4, nullptr, 0, 0, [&] { launch_topk_hist(a, stream); });
  exclusive_scan_u32(a.counts, a.offsets, ng + 1, temp, tb, stream);
  RDFGPU_HIP(hipMemcpyAsync(a.cursor, a.offsets, ng * sizeof(u32), hipMemcpyDeviceToDevice, stream));
  timed(KC_TOPK_SCATTER, 0, in.cap, in.n_dev, 8, nullptr, 0, 0, [&] { launch_topk_scatter(a, stream); });
  timed(KC_TOPK_SELECT, 0, in.cap, in.n_dev, (4ull + key_bytes) * a.k, nullptr, 0, 0, [&] { launch_topk_select(a, stream); });
  exclusive_scan_u32(a.out_counts, a.out_offsets, ng + 1, temp, tb, stream);
  timed(KC_TOPK_WRITE, 0, 0, nullptr, 0, n_out, 0, 8ull * nd.n_proj, [&] { launch_topk_write(a, stream); });
  const u32 i0 = (u32)(n_out - counters);
  RDFGPU_HIP(hipMemcpyAsync(ctx->counters_host + i0, counters + i0, 2 * sizeof(u64), hipMemcpyDeviceToHost, stream));
  RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
  if ((ctx->counters_host[i0 + 1] >> 32) != 0) fail(RDFGPU_ERR_UNSUPPORTED, "TopK: SORT_BY_TERM over a column that is not all strings / IRIs / blank nodes");
  t.cap = std::min<u64>(out_cap, ctx->counters_host[i0]);
  t.n_dev = n_out;
  return t;
}

// Which input the hash join builds on.  A left join must build on the preserved (left) side.  An inner join builds
// on the smaller input — unless exactly one input is a pure slice of the store (the same rows on every execution
// until the store changes) and the other one is no larger: the slice's join table (direct-address / CSR / hash) is
// built once and cached on the node, so building there costs nothing per execution and the probe is the SMALL side
// (an index nested-loop join against the store's own permutation: `PARAMS JOIN (?s p ?o)` touches |PARAMS| rows,
// not the 5 M-row predicate partition).
bool Plan::choose_build_left(const NodeInfo& nd, const DevTable& L, const DevTable& R, bool left_join, bool lf, bool rf, bool lpost, bool rpost) const {
  if (left_join) {
    // OPTIONAL: HashJoinExec(Left) builds on its left input and probes with the right one — every row of the right input is read, however few
    // left rows there are.  When the right input is a store slice (its join table is cached per store version) and much larger than the left, the
    // join is run the other way round and PRESERVES ITS PROBE SIDE: build = the slice's table, probe = the left rows, a probe row without a match is
    // emitted once with a null right side (LdsJoinArgs::probe_outer).  Same multiset of rows.  Needs: no join filter, no fused filter.
    const bool rs = R.stable_id != 0 && R.n_dev == nullptr && !rf && !lf;
    if (rs && nd.d.kind == RDFGPU_NODE_HASH_JOIN && nd.shape == 0 && !opt.on(RDFGPU_OPT_NO_TABLE_CACHE) && !opt.on(RDFGPU_OPT_NO_INDEX_JOIN) &&
        !opt.on(RDFGPU_OPT_NO_PROBE_OUTER_JOIN) && R.cap > 1024 && L.cap * 4 <= R.cap && L.cap < (1ull << 31))
      return false;
    return true;
  }
  const bool smaller_left = L.cap <= R.cap;
  if (nd.d.kind != RDFGPU_NODE_HASH_JOIN || opt.on(RDFGPU_OPT_NO_TABLE_CACHE) || opt.on(RDFGPU_OPT_NO_INDEX_JOIN)) return smaller_left;
  // (a slice under a `col <=|!=> literal` FilterExec still counts: that filter can run as a conjunct of the join filter)
  const bool ls = L.stable_id != 0 && L.n_dev == nullptr && (!lf || lpost), rs = R.stable_id != 0 && R.n_dev == nullptr && (!rf || rpost);
  if (!ls && !rs) return smaller_left;
  // candidate: build on the slice (the larger one when both inputs are slices), probe with the other input
  const bool slice_left = ls && rs ? !smaller_left : ls;
  const DevTable& S = slice_left ? L : R; const DevTable& O = slice_left ? R : L;
  if (O.cap > S.cap) return smaller_left;                   // the slice is already the smaller side
  if (ls && rs && O.cap * 8 > S.cap) return smaller_left;   // two slices of similar size: nothing to gain
  if (S.cap <= 1024) return smaller_left;                   // LDS-table territory
  // A slice whose keys are not one dense id range (several key columns, or an earlier attempt on this very slice said
  // so) gets a cached HASH table: built once per store version, probed with the fewer rows.  It loses only to a small
  // table built on the other side that stays in L2 while the slice's would not (measured on LUBM Q9's two-key join,
  // 24.6 M rows against a 57 M-row slice: 5.9 ms building on the smaller side every run, 1.3 ms on the cached slice).
  bool hash_only = nd.d.n_keys != 1;
  if (!hash_only) {
    SliceKey sk; sk.n_keys = 1; sk.rows = S.cap; sk.key[0] = S.cols[slice_left ? nd.d.left_keys[0] : nd.d.right_keys[0]];
    const SliceTable* st = store->find_slice_table(sk);
    hash_only = st && st->dense_failed;
  }
  if (hash_only) {
    // The slice is sorted by one of the join keys and the other input is large: building on THAT input inside the step — partition passes
    // over its rows, the slice read in place as id-range partitions (part_join.hip) — beats probing the slice's cached hash table, whose
    // probes are random 64-byte reads of a table far larger than the caches.  Measured on LUBM-8000 Q9's closing two-key join (98 M rows
    // against the 229 M-row takesCourse slice): 3.0 ms with the build in the step, 4.2 - 5.0 ms on the cached table.  The partitioned join
    // walks the whole slice, so it pays only when the other input is a good fraction of it (break-even near a fifth, from those numbers).
    bool sorted_by_key = false;
    for (u32 k = 0; k < nd.d.n_keys; k++) sorted_by_key = sorted_by_key || (S.sorted_col >= 0 && (u32)S.sorted_col == (slice_left ? nd.d.left_keys[k] : nd.d.right_keys[k]));
    if (sorted_by_key && !(ls && rs) && !lf && !rf && nd.d.n_keys <= 2 && S.key_min >= 1 && S.key_max >= S.key_min &&
        !opt.on(RDFGPU_OPT_NO_PARTITIONED_JOIN) && !opt.on(RDFGPU_OPT_NO_RANGE_PARTITION) && !opt.on(RDFGPU_OPT_NO_OWN_PARTITION_PASS) &&
        O.cap >= opt.v[RDFGPU_OPT_PARTITION_MIN_BUILD] && O.cap * 5 >= S.cap && O.cap < (1ull << 31) && S.cap < (1ull << 31))
      return !slice_left;
    if (S.cap > (256ull << 20)) return smaller_left;                             // 32 B per row: keep the footprint sane
    if (O.cap * 8 > S.cap && O.cap <= (1ull << 20)) return smaller_left;
  }
  return slice_left;
}

// Fused lookup chain (R4): walks down from `top` through inner single-key hash joins whose one input is a pure store
// slice with a cached DIRECT-address table and whose other input is a hash join consumed only here.  Only tried on
// speculative re-executions (cardinalities and tables known from the first run).
bool Plan::plan_chain(NodeInfo& top, ChainRequest& req) {
  if (!speculative || !top.has_last || opt.on(RDFGPU_OPT_NO_CHAIN_FUSION) || opt.on(RDFGPU_OPT_NO_TABLE_CACHE)) return false;
  req.top = &top;
  std::vector<ChainLink> down;
  NodeInfo* cur = &top;
  while ((int)down.size() < kMaxChain) {
    const rdfgpu_plan_node& d = cur->d;
    if (d.kind != RDFGPU_NODE_HASH_JOIN || d.join_type != RDFGPU_JOIN_INNER || d.n_keys != 1) break;
    if (cur->prog.n != 0 && cur->shape != 2 && cur->shape != 3) break;
    bool found = false;
    for (int side = 0; side < 2 && !found; side++) {
      const u32 cs = (u32)(side == 0 ? d.left : d.right), co = (u32)(side == 0 ? d.right : d.left);
      const NodeInfo& sn = nodes[cs]; const NodeInfo& on = nodes[co];
      if (sn.d.kind != RDFGPU_NODE_DATA_SOURCE || sources[sn.source].has_residual) continue;
      if (on.d.kind != RDFGPU_NODE_HASH_JOIN || on.refs != 1) continue;
      const DevTable S = exec_node(cs);   // a slice: no launch
      if (S.cap == 0 || S.stable_id == 0) continue;
      SliceKey sk; sk.n_keys = 1; sk.rows = S.cap; sk.key[0] = S.cols[side == 0 ? d.left_keys[0] : d.right_keys[0]];
      const SliceTable* st = store->find_slice_table(sk);
      if ((!st || !st->dense_tried) && S.n_dev == nullptr && S.cap > std::min<u64>(opt.v[RDFGPU_OPT_LDS_MAX_BUILD], kLdsJoinMaxBuild) && !opt.on(RDFGPU_OPT_NO_DIRECT_TABLE)) {
        // the store changed since this chain last ran (its history is still good): the slice's table is built here, inside
        // the execution, and the chain stays fused — no un-fused execution just to get the tables back
        SliceTable* fresh = store->slice_table(sk);
        std::unique_lock<std::mutex> building(store->slice_build_mu);
        if (!fresh->dense_tried) build_dense_table(fresh, sk.key[0], S.cap);
        st = fresh;
      }
      if (!st || !st->direct) continue;
      down.push_back(ChainLink{cur, side == 0, S, st});
      cur = &nodes[co];
      found = true;
    }
    if (!found) break;
  }
  if (down.empty()) return false;
  req.links.assign(down.rbegin(), down.rend());   // bottom-up: links[0] sits directly above the base join
  return true;
}

// Resolves the chain against the base join's inputs; false (nothing changed in `a` that matters) if some column cannot
// be addressed the way the kernel needs.
bool Plan::apply_chain(const ChainRequest& req, NodeInfo& base, const DevTable& L, const DevTable& R, bool build_left, LdsJoinArgs& a, u64& stage_bytes, BandArgs* band, bool* use_band) {
  std::vector<ColRef> cur(base.n_proj);
  for (u32 k = 0; k < base.n_proj; k++) {
    const u32 c = base.proj[k];
    const bool from_left = c < L.n_cols;
    cur[k] = ColRef{from_left ? L.cols[c] : R.cols[c - L.n_cols], (from_left == build_left) ? 1u : 0u, 0u};
  }
  ChainStage stages[kMaxChain];
  const SliceTable::ValueColumn* stage_vc[kMaxChain] = {nullptr, nullptr, nullptr};
  stage_bytes = 0;
  for (size_t t = 0; t < req.links.size(); t++) {
    const ChainLink& ln = req.links[t];
    const NodeInfo& N = *ln.node;
    const u32 wl = nodes[N.d.left].width;
    const u32 prev_w = ln.slice_is_left ? nodes[N.d.right].width : wl;
    if (prev_w != cur.size()) return false;
    bool bad = false;
    auto resolve = [&](u32 c) -> ColRef {
      const bool in_left = c < wl; const u32 local = in_left ? c : c - wl;
      if (in_left == ln.slice_is_left) { if (local >= ln.slice.n_cols) { bad = true; return ColRef{}; } return ColRef{ln.slice.cols[local], 2u + (u32)t, 0u}; }
      if (local >= cur.size()) { bad = true; return ColRef{}; }
      return cur[local];
    };
    ChainStage& st = stages[t];
    std::memset(&st, 0, sizeof st);
    const u32 prev_key = ln.slice_is_left ? N.d.right_keys[0] : N.d.left_keys[0];
    if (prev_key >= cur.size()) return false;
    st.key = cur[prev_key];
    if (st.key.src > 1) return false;                       // the kernel looks a stage up from a BASE column
    st.direct = ln.table->direct; st.kmin = ln.table->kmin; st.kn = ln.table->kn;
    u32 n_fcols = 0;
    if (N.prog.n == 0) st.fs = 0;
    else if (N.shape == 2) {
      st.fs = 2; n_fcols = 2;
      st.f[0] = resolve(N.prog.nodes[0].u); st.f[1] = resolve(N.prog.nodes[1].u);
      st.is_eq = N.prog.nodes[2].op == RDFGPU_EX_ID_EQ;
    } else if (N.shape == 3) {
      const rdfgpu_expr_node* e = N.prog.nodes;
      auto lit = [&](u32 o) { TvLiteral l{}; l.lo = e[o + 4].lo; l.hi = e[o + 4].hi; l.aux = e[o + 4].u; l.tag = e[o + 4].tag; l.flags = e[o + 4].flags;
                              l.arith_sub = e[o + 5].op == RDFGPU_EX_SUB; l.cmp_op = e[o + 6].op; return l; };
      st.fs = 3; n_fcols = 4;
      st.f[0] = resolve(e[0].u); st.f[1] = resolve(e[2].u); st.f[2] = resolve(e[8].u); st.f[3] = resolve(e[10].u);
      st.l0 = lit(0); st.l1 = lit(8);
    } else return false;
    for (u32 q = 0; q < n_fcols; q++) if (st.f[q].src > 1 && st.f[q].src != 2u + (u32)t) return false;   // base columns or this stage's
    // integer window whose x operand is a column of this stage's slice and whose y operands are base columns: use the
    // slice's decoded value table (built once per store version, kept with the direct table)
    if (st.fs == 3 && st.f[0].src == 2u + (u32)t && st.f[2].src == st.f[0].src && st.f[2].ptr == st.f[0].ptr && st.f[1].src <= 1 && st.f[3].src <= 1 &&
        !opt.on(RDFGPU_OPT_NO_VALUE_TABLES)) {
      SliceTable* tab = const_cast<SliceTable*>(ln.table);
      std::unique_lock<std::mutex> building(store->slice_build_mu);
      SliceTable::ValueColumn* vc = nullptr;
      for (auto& v : tab->values) if (v.col == st.f[0].ptr) vc = &v;
      if (!vc) {
        const u32 key_local = ln.slice_is_left ? N.d.left_keys[0] : N.d.right_keys[0];
        long long* val = store->table_alloc<long long>(tab->kn);
        metrics.tables_built++;
        u32* bad = reinterpret_cast<u32*>(new_counter());
        launch_fill_i64(val, INT64_MIN, tab->kn, stream);
        launch_direct_values(ln.slice.cols[key_local], st.f[0].ptr, ln.slice.cap, tab->kmin, tab->kn, typed_table(), val, bad, stream);
        u32 is_bad = 0;
        RDFGPU_HIP(hipMemcpyAsync(&is_bad, bad, sizeof(u32), hipMemcpyDeviceToHost, stream));
        RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
        if (is_bad) { store->table_free(val); val = nullptr; }
        SliceTable::ValueColumn fresh{st.f[0].ptr, val, val != nullptr};
        if (val) {   // value range: the bias of the band join's 32-bit window intervals
          long long* mm = reinterpret_cast<long long*>(new_counter()); (void)new_counter();
          const long long init[2] = {INT64_MAX, INT64_MIN + 1};
          RDFGPU_HIP(hipMemcpyAsync(mm, init, sizeof init, hipMemcpyHostToDevice, stream));
          launch_val_minmax(val, tab->kn, mm, stream);
          long long got[2];
          RDFGPU_HIP(hipMemcpyAsync(got, mm, sizeof got, hipMemcpyDeviceToHost, stream));
          RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
          fresh.vmin = got[0]; fresh.vmax = got[1];
        }
        tab->values.push_back(fresh);
        vc = &tab->values.back();
      }
      if (vc->usable) { st.val = vc->val; stage_vc[t] = vc; }
    }
    std::vector<ColRef> next(N.n_proj);
    for (u32 k = 0; k < N.n_proj; k++) next[k] = resolve(N.proj[k]);
    if (bad) return false;
    cur.swap(next);
    stage_bytes += base.last_rows * (8ull + 4ull * n_fcols);   // per candidate: key + table slot + filter operands (estimate)
  }
  if (cur.size() != req.top->n_proj || cur.size() > (size_t)kMaxCols) return false;
  // Band join (band_join.hip): when the groups of the CSR base are small, every stage hangs off a BUILD column and the
  // stage filters are integer windows between a stage value and probe columns, the chain runs group by group — both
  // sides partitioned by the key, the pair tests in registers — instead of probe row by probe row.
  *use_band = false;
  if (band && a.csr_off && cur_build_table && !opt.on(RDFGPU_OPT_NO_BAND_JOIN) && arg_slots_used < ExecContext::kArgSlots) {
    const DevTable& B = build_left ? L : R; const DevTable& Pp = build_left ? R : L;
    auto from_build = [&](u32 c) { return (c < L.n_cols) == build_left; };
    auto range_op = [](u8 op) { return op == RDFGPU_EX_GT || op == RDFGPU_EX_LT || op == RDFGPU_EX_GEQ || op == RDFGPU_EX_LEQ; };
    BandArgs b{};
    bool ok = a.n_keys == 1 && (a.has_filter == 0 || a.has_filter == 2) && a.has_probe_filter == 0 && a.visited == nullptr;
    if (ok && a.has_filter == 2) {
      const bool ab = from_build(a.idp.a), bb = from_build(a.idp.b);
      ok = ab != bb;
      if (ok) { b.has_neq = 1; b.neq_is_eq = a.idp.is_eq; b.neq_build = a.cols[ab ? a.idp.a : a.idp.b]; b.neq_probe = a.cols[ab ? a.idp.b : a.idp.a]; }
    }
    if (ok && a.has_post) {
      ok = from_build(a.post.col);
      b.has_post = 1; b.post_col = a.cols[a.post.col]; b.post_lit = a.post.lit; b.post_is_eq = a.post.is_eq;
    }
    b.n_stages = (u32)req.links.size();
    bool pack16 = !opt.on(RDFGPU_OPT_NO_BAND_PACK16);
    for (size_t t = 0; ok && t < req.links.size(); t++) {
      const ChainStage& st = stages[t];
      ok = st.key.src == 1 && (st.fs == 0 || st.fs == 3);
      if (!ok) break;
      b.stage[t] = BandStage{st.key.ptr, st.direct, st.kmin, st.kn};
      if (st.fs == 0) continue;
      const SliceTable::ValueColumn* vc = stage_vc[t];
      ok = st.val != nullptr && vc && b.n_win < 2 && st.f[1].src == 0 && st.f[3].src == 0 && range_op(st.l0.cmp_op) && range_op(st.l1.cmp_op) &&
           vc->vmin <= vc->vmax && (unsigned long long)(vc->vmax - vc->vmin) < 0xFFFFFFE0ull;
      if (!ok) break;
      BandWin& w = b.win[b.n_win++];
      if ((unsigned long long)(vc->vmax - vc->vmin) > 65530ull) pack16 = false;   // biased values 1 .. range + 1 have to fit 16 bits
      w.key_col = st.key.ptr; w.val = st.val; w.vkmin = st.kmin; w.vkn = st.kn; w.vbase = vc->vmin;
      w.y0 = st.f[1].ptr; w.y1 = st.f[3].ptr; w.l0 = st.l0; w.l1 = st.l1; w.stage = (u32)t;
    }
    b.pack16 = pack16 ? 1u : 0u;
    for (size_t k = 0; ok && k < cur.size(); k++) {
      if (cur[k].src == 0) { ok = b.n_row_cols < kBandMaxRowCols; if (ok) { b.out_from_row[k] = 1; b.out_sel[k] = (u8)b.n_row_cols; b.row_col[b.n_row_cols++] = cur[k].ptr; } }
      else { ok = b.n_entry_cols < kBandMaxSideCols; if (ok) { b.out_from_row[k] = 0; b.out_sel[k] = (u8)(2 + b.n_entry_cols); b.entry_col[b.n_entry_cols++] = cur[k]; } }
    }
    if (ok) {   // group sizes: the largest decides (one wave joins a whole group), measured once per table
      SliceTable* tab = cur_build_table;
      std::unique_lock<std::mutex> building(store->slice_build_mu);
      if (tab->csr_max_group == 0) {
        u32* mx = reinterpret_cast<u32*>(new_counter());
        launch_csr_max_group(a.csr_off, a.direct_n, mx, stream);
        u32 got = 0;
        RDFGPU_HIP(hipMemcpyAsync(&got, mx, sizeof got, hipMemcpyDeviceToHost, stream));
        RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
        tab->csr_max_group = got ? got : 1;
      }
      ok = tab->csr_max_group <= kBandMaxGroup && B.cap >= 4ull * a.direct_n && Pp.cap * 4 >= a.direct_n && Pp.cap < (1ull << 31);
    }
    if (ok) { *band = b; *use_band = true; }
  }
  // Range index: a CSR base whose first stage is an integer window (GT / LT / GEQ / LEQ) between the stage's decoded
  // value and probe-side columns expands, per probe row, only the part of the key's group whose value can pass —
  // the group is kept sorted by that value (built once per store version, kept with the CSR table).
  if (!*use_band) {
    const ChainStage& s0 = stages[0];
    auto range_op = [](u8 op) { return op == RDFGPU_EX_GT || op == RDFGPU_EX_LT || op == RDFGPU_EX_GEQ || op == RDFGPU_EX_LEQ; };
    const DevTable& B = build_left ? L : R;
    if (a.csr_off && cur_build_table && s0.val && s0.fs == 3 && s0.key.src == 1 && s0.f[1].src == 0 && s0.f[3].src == 0 &&
        range_op(s0.l0.cmp_op) && range_op(s0.l1.cmp_op) && !opt.on(RDFGPU_OPT_NO_RANGE_INDEX)) {
      SliceTable* tab = cur_build_table;
      std::unique_lock<std::mutex> building(store->slice_build_mu);
      SliceTable::RangeIndex* ri = nullptr;
      for (auto& r : tab->ranges) if (r.val == s0.val && r.link_col == s0.key.ptr) ri = &r;
      if (!ri) {
        SliceTable::RangeIndex fresh{s0.val, s0.key.ptr, nullptr, nullptr, 0, nullptr, false};
        const u64 n = B.cap;
        long long* mm = reinterpret_cast<long long*>(new_counter()); (void)new_counter();   // {min, max}: two slots
        const long long init[2] = {INT64_MAX, INT64_MIN + 1};
        RDFGPU_HIP(hipMemcpyAsync(mm, init, sizeof init, hipMemcpyHostToDevice, stream));
        launch_range_minmax(s0.key.ptr, a.csr_rows, n, s0.val, s0.kmin, s0.kn, mm, stream);
        long long got[2];
        RDFGPU_HIP(hipMemcpyAsync(got, mm, sizeof got, hipMemcpyDeviceToHost, stream));
        RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
        if (got[0] <= got[1] && (unsigned long long)(got[1] - got[0]) < 0xFFFFFFF0ull && n < (1ull << 32)) {
          u64* key_in = scratch<u64>(n); u64* key_out = scratch<u64>(n); u32* rows_in = scratch<u32>(n);
          fresh.rows = store->table_alloc<u32>(n);
          fresh.vals = store->table_alloc<u32>(n);
          metrics.tables_built++;
          fresh.vbase = got[0];
          launch_range_keys(a.build_key[0], a.direct_min, s0.key.ptr, a.csr_rows, n, s0.val, s0.kmin, s0.kn, got[0], key_in, rows_in, stream);
          const size_t tb = sort_temp_bytes(n);
          void* temp = scratch<unsigned char>(tb);
          sort_pairs_u64_u32(key_in, key_out, rows_in, fresh.rows, n, temp, tb, stream);
          launch_range_decode(key_out, n, fresh.vals, stream);
          fresh.link = store->table_alloc<u32>(n);
          launch_gather_u32(s0.key.ptr, fresh.rows, fresh.link, n, stream);   // the link column in index order
          RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
          fresh.usable = true;
        }
        tab->ranges.push_back(fresh);
        ri = &tab->ranges.back();
      }
      if (ri->usable) { a.range_rows = ri->rows; a.range_vals = ri->vals; a.range_vbase = ri->vbase; a.range_link = ri->link; a.range_link_col = ri->link_col; }
    }
  }
  a.n_chain = (u32)req.links.size();
  for (size_t t = 0; t < req.links.size(); t++) a.chain[t] = stages[t];
  for (size_t k = 0; k < cur.size(); k++) a.chain_out[k] = cur[k];
  a.n_out_cols = (u32)cur.size();
  return true;
}

// The held-back write pass of an ordered slice join, run after all: its consumer turned out not to take the band join's records.
void Plan::flush_pending_oj() {
  if (!pending_oj.active) return;
  pending_oj.active = false;
  const OrderedJoinArgs& o = pending_oj.o;
  timed(KC_OJ_WRITE, 0, pending_oj.n_build, nullptr, 4, o.n_out_dev, 0, 8ull * o.n_out_cols, [&] { launch_ordered_join_write(o, stream); });
}

DevTable Plan::exec_join(NodeInfo& nd) {
  const bool left_join = nd.d.join_type == RDFGPU_JOIN_LEFT;
  if (!pending_chain) {
    ChainRequest req;
    if (plan_chain(nd, req)) {
      NodeInfo* base = req.links.front().slice_is_left ? &nodes[req.links.front().node->d.right] : &nodes[req.links.front().node->d.left];
      req.base = base;
      pending_chain = &req;
      const DevTable fused = exec_node((u32)(base - nodes.data()));
      pending_chain = nullptr;
      if (req.consumed) return fused;      // `fused` already has this node's schema
      // not taken (the base join ran normally and is memoised): continue the ordinary way
    }
  }
  // Pipeline fusion: a FilterExec child (identity projection, consumed by this join only) is not
  // materialised when it ends up on the probe side of the LDS join — its predicate runs inside the probe.
  auto fusable = [&](int32_t ci) {
    if (nd.d.kind != RDFGPU_NODE_HASH_JOIN || opt.on(RDFGPU_OPT_NO_FILTER_FUSION) || opt.on(RDFGPU_OPT_NO_LDS_JOIN)) return false;
    const NodeInfo& c = nodes[ci];
    if (c.d.kind != RDFGPU_NODE_FILTER || c.prog.n == 0 || c.refs != 1 || c.n_proj != nodes[c.d.left].width) return false;
    for (u32 k = 0; k < c.n_proj; k++) if (c.proj[k] != k) return false;
    return true;
  };
  bool lf = !left_join && fusable(nd.d.left), rf = fusable(nd.d.right);
  // the inputs are sub-plans of their own: a chain request pending for THIS join must not keep the joins below from
  // planning theirs (the batched Q5 has two: the constants' look-ups by X, and window 1 / window 2 / label above the
  // candidate join)
  ChainRequest* const for_this_join = pending_chain;
  pending_chain = nullptr;
  DevTable L = lf ? exec_node((u32)nodes[nd.d.left].d.left) : exec_node((u32)nd.d.left);
  DevTable R = rf ? exec_node((u32)nodes[nd.d.right].d.left) : exec_node((u32)nd.d.right);
  pending_chain = for_this_join;
  if (pending_oj.active && !(nd.band_takes_records && nd.d.kind == RDFGPU_NODE_HASH_JOIN && !left_join && !lf && !rf)) flush_pending_oj();
  const NodeInfo* post = nullptr;   // a build-side FilterExec kept as a conjunct of the join filter (see below)
  if (lf || rf) {
    // `col <=|!=> literal` over a store slice: if the join builds on that slice (index join through the slice's cached
    // table) the FilterExec is neither materialised nor fused into the probe — it becomes one more conjunct of the
    // join filter, evaluated on the candidate pairs
    auto postable = [&](bool has, int32_t ci, const DevTable& in) {
      return has && nodes[ci].shape == 1 && in.stable_id != 0 && in.n_dev == nullptr && in.cap > 1024 && nd.d.n_keys == 1 && !opt.on(RDFGPU_OPT_NO_INDEX_JOIN);
    };
    const bool lpost = postable(lf, nd.d.left, L), rpost = postable(rf, nd.d.right, R);
    const bool build_left = choose_build_left(nd, L, R, left_join, lf, rf, lpost, rpost);
    const bool lds = ((build_left ? L.cap : R.cap) <= kLdsJoinMaxBuild || !opt.on(RDFGPU_OPT_NO_GLOBAL_TABLE_JOIN)) && L.cap && R.cap;
    if (lds && lf && build_left && lpost) { post = &nodes[nd.d.left]; lf = false; }
    else if (lds && rf && !build_left && rpost) { post = &nodes[nd.d.right]; rf = false; }
    // a fused filter survives only on the probe side of the LDS join; anything else is materialised now
    if (lf && (!lds || build_left)) { L = apply_filter(nodes[nd.d.left], L); lf = false; }
    if (rf && (!lds || !build_left)) { R = apply_filter(nodes[nd.d.right], R); rf = false; }
  }
  DevTable t;
  t.n_cols = nd.n_proj;

  if (nd.d.kind == RDFGPU_NODE_CROSS_JOIN) {
    flush_pending_oj();
    const u64 cap = L.cap * R.cap;
    if (cap == 0) { t.cap = 0; return t; }
    if (cap >= (1ull << 40)) fail(RDFGPU_ERR_UNSUPPORTED, "cross join of %llu x %llu rows", (unsigned long long)L.cap, (unsigned long long)R.cap);
    CrossArgs a{};
    for (u32 c = 0; c < L.n_cols; c++) a.left[c] = L.cols[c];
    for (u32 c = 0; c < R.n_cols; c++) a.right[c] = R.cols[c];
    a.n_left_cols = L.n_cols; a.n_right_cols = R.n_cols; a.n_out_cols = nd.n_proj;
    for (u32 c = 0; c < nd.n_proj; c++) { a.proj[c] = nd.proj[c]; a.out[c] = scratch<u32>(cap); t.cols[c] = a.out[c]; }
    a.n_left_dev = L.n_dev; a.n_left_cap = L.cap; a.n_right_dev = R.n_dev; a.n_right_cap = R.cap;
    const bool dyn = L.n_dev || R.n_dev;
    a.n_out_dev = dyn ? new_counter() : nullptr;
    // CrossJoinExec bytes: 4·c·N (both inputs read once) + 4·c_o·m·N (every output cell written)
    timed(KC_CROSS, 4ull * R.n_cols * R.cap, L.cap, L.n_dev, 4ull * L.n_cols, a.n_out_dev, dyn ? 0 : cap, 4ull * nd.n_proj,
          [&] { launch_cross(a, stream); });
    t.cap = cap; t.n_dev = a.n_out_dev;
    return t;
  }

  // HashJoinExec / NestedLoopJoinExec
  if (L.cap == 0 || (R.cap == 0 && !left_join)) { t.cap = 0; return t; }
  const bool hash = nd.d.kind == RDFGPU_NODE_HASH_JOIN;
  if (hash && !opt.on(RDFGPU_OPT_NO_LDS_JOIN)) {
    const bool build_left = choose_build_left(nd, L, R, left_join, lf, rf);
    if ((build_left ? L.cap : R.cap) <= kLdsJoinMaxBuild || !opt.on(RDFGPU_OPT_NO_GLOBAL_TABLE_JOIN)) {
      const NodeInfo* pf = lf ? &nodes[nd.d.left] : rf ? &nodes[nd.d.right] : nullptr;
      if (post && build_left != (post == &nodes[nd.d.left])) fail(RDFGPU_ERR_DEVICE, "join: build side changed under a residual filter");
      return exec_lds_join(nd, L, R, build_left, pf, post);
    }
  }
  flush_pending_oj();
  JoinArgs a{};
  for (u32 c = 0; c < L.n_cols; c++) a.left[c] = L.cols[c];
  for (u32 c = 0; c < R.n_cols; c++) a.right[c] = R.cols[c];
  a.n_left_cols = L.n_cols; a.n_right_cols = R.n_cols; a.n_out_cols = nd.n_proj;
  for (u32 c = 0; c < nd.n_proj; c++) a.proj[c] = nd.proj[c];
  a.n_keys = hash ? nd.d.n_keys : 0;
  for (u32 k = 0; k < a.n_keys; k++) { a.left_keys[k] = nd.d.left_keys[k]; a.right_keys[k] = nd.d.right_keys[k]; }
  a.n_left_dev = L.n_dev; a.n_left_cap = L.cap; a.n_right_dev = R.n_dev; a.n_right_cap = R.cap;
  a.has_filter = nd.prog.n ? 1 : 0;
  a.prog = nd.prog;
  a.tt = typed_table();
  if (L.cap >= 0xFFFFFFF0ull) fail(RDFGPU_ERR_UNSUPPORTED, "build side of %llu rows", (unsigned long long)L.cap);
  if (left_join) { a.visited = scratch<u8>(L.cap); if (!hash) RDFGPU_HIP(hipMemsetAsync(a.visited, 0, L.cap, stream)); }
  if (hash) {
    const u32 nb = pow2_at_least(2 * L.cap);
    a.heads = scratch<u32>(nb); a.bucket_mask = nb - 1;
    a.next = scratch<u32>(L.cap);
    RDFGPU_HIP(hipMemsetAsync(a.heads, 0xFF, (size_t)nb * 4, stream));
    // build: 4·k·N_b keys read + 8·N_b (one head/next slot written per row)   (build half)
    timed(KC_JOIN_BUILD, 0, L.cap, L.n_dev, 4ull * a.n_keys + 8, nullptr, 0, 0, [&] { launch_join_build(a, stream); });
  }
  // columns of the probe side the write pass has to read: keys ∪ projected right columns ∪ filter columns
  u32 probe_cols = 0;
  {
    bool used[kMaxCols] = {};
    for (u32 k = 0; k < a.n_keys; k++) used[a.right_keys[k]] = true;
    for (u32 c = 0; c < nd.n_proj; c++) if (nd.proj[c] >= L.n_cols) used[nd.proj[c] - L.n_cols] = true;
    for (u32 i = 0; i < nd.prog.n; i++) if (nd.prog.nodes[i].op == RDFGPU_EX_COLUMN && nd.prog.nodes[i].u >= L.n_cols) used[nd.prog.nodes[i].u - L.n_cols] = true;
    for (bool b : used) probe_cols += b;
  }
  u64 total = 0;
  u32* offs = nullptr;
  if (R.cap) {
    u32* counts = scratch<u32>(R.cap);
    offs = scratch<u32>(R.cap);
    const size_t tb = scan_temp_bytes(R.cap);
    void* temp = scratch<u8>(tb);
    a.counts = counts;
    // count pass: 4·k·N_p keys + 8·N_p (head + first chain slot read per probe row)
    timed(hash ? KC_JOIN_COUNT : KC_NLJ_COUNT, 0, R.cap, R.n_dev, 4ull * a.n_keys + 8, nullptr, 0, 0,
          [&] { if (hash) launch_join_count(a, stream); else launch_nlj_count(a, stream); });
    timed(scan_class(R.cap), 0, R.cap, nullptr, 8, nullptr, 0, 0, [&] { inclusive_scan_u32(counts, offs, R.cap, temp, tb, stream); });
    u32 tot32 = 0;
    RDFGPU_HIP(hipMemcpyAsync(&tot32, offs + R.cap - 1, 4, hipMemcpyDeviceToHost, stream));
    RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
    total = tot32;
  }
  const u64 cap = total + (left_join ? L.cap : 0);
  if (cap == 0) { t.cap = 0; return t; }
  for (u32 c = 0; c < nd.n_proj; c++) { a.out[c] = scratch<u32>(cap); t.cols[c] = a.out[c]; }
  if (total) {
    a.counts = offs;
    // write pass: 4·(k+p_p)·N_p + 8·N_p + 4·c_o·N_o   (probe half)
    timed(hash ? KC_JOIN_WRITE : KC_NLJ_WRITE, 0, R.cap, R.n_dev, 4ull * probe_cols + 8, nullptr, total, 4ull * nd.n_proj,
          [&] { if (hash) launch_join_write(a, stream); else launch_nlj_write(a, stream); });
  }
  t.cap = cap;
  if (left_join) {
    a.n_out_dev = new_counter();
    RDFGPU_HIP(hipMemcpyAsync(a.n_out_dev, &total, sizeof(u64), hipMemcpyHostToDevice, stream));
    RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;   // `total` is a stack variable
    timed(KC_LEFT_TAIL, 0, L.cap, L.n_dev, 1, nullptr, 0, 0, [&] { launch_join_left_unmatched(a, stream); });
    t.n_dev = a.n_out_dev;
  }
  return t;
}

// The dense join tables of a single-key store slice (decided once per slice and store version, with `slice_build_mu` held):
// direct-address if the keys are unique, CSR (offsets + row ids grouped by key; the identity when the slice is sorted by the
// key) if not; `dense_failed` when the id range is not worth a 4-byte-per-id table.  Costs a few small kernels and host
// syncs at that time, nothing afterwards.
void Plan::build_dense_table(SliceTable* st, const u32* key, u64 n) {
  u32* mm = reinterpret_cast<u32*>(new_counter());     // {min, max}
  u32* flags = reinterpret_cast<u32*>(new_counter());  // {duplicate seen, unsorted seen}
  const u32 init[2] = {0xFFFFFFFFu, 0u};
  RDFGPU_HIP(hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, stream));
  timed(KC_MINMAX, 4ull * n, 0, nullptr, 0, nullptr, 0, 0, [&] { launch_minmax_u32(key, n, mm, stream); });
  u32 got[2];
  RDFGPU_HIP(hipMemcpyAsync(got, mm, sizeof(got), hipMemcpyDeviceToHost, stream));
  RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
  // "dense" = the id range is worth a 4-B-per-id table: up to 4 ids per row outright; up to 64 ids per row while
  // the table stays small (16 M ids = 64 MB) — a subject-hash shard of a slice keeps the slice's id range with
  // 1/G of its rows, and must not fall off the index-join path for that
  const u64 range = got[0] <= got[1] ? (u64)(got[1] - got[0]) + 1 : ~0ull;
  const bool dense = range <= 4 * n + 1024 || (range <= 64 * n + 1024 && range <= (16ull << 20));
  if (!dense) { st->dense_failed = true; st->dense_tried = true; return; }
  metrics.tables_built++;
  const u32 kmin = got[0], kn = got[1] - got[0] + 1;
  st->kmin = kmin; st->kn = kn;
  // more rows than ids in the range: some key repeats (pigeonhole: null keys only make it more so when they are few; with many nulls the
  // attempt below would have succeeded — then the CSR form is merely the more general table for the same join) — no direct-address attempt
  if (n <= (u64)kn) {
    u32* direct = store->table_alloc<u32>(kn);
    RDFGPU_HIP(hipMemsetAsync(direct, 0xFF, (size_t)kn * sizeof(u32), stream));
    timed(KC_GDIRECT_BUILD, 0, n, nullptr, 8, nullptr, 0, 0, [&] { launch_gdirect_build(key, n, direct, kmin, kn, flags, stream); });
    u32 is_dup = 0;
    RDFGPU_HIP(hipMemcpyAsync(&is_dup, flags, sizeof(u32), hipMemcpyDeviceToHost, stream));
    RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
    if (!is_dup) { st->direct = direct; st->dense_tried = true; return; }
    store->table_free(direct);
  }
  // duplicates: CSR (offsets + row ids grouped by key) — by boundary searches when the slice is sorted by the key, by one radix sort otherwise
  u32* off = store->table_alloc<u32>((u64)kn + 2); u32* rows = nullptr;
  if (n >= (1ull << 32)) fail(RDFGPU_ERR_UNSUPPORTED, "CSR table of %llu rows", (unsigned long long)n);
  u32* rel = scratch<u32>(n);
  timed(KC_CSR_HIST, 0, n, nullptr, 8, nullptr, 0, 0, [&] { launch_csr_rel_keys(key, n, kmin, kn, rel, flags + 1, stream); });
  u32 unsorted = 0;
  RDFGPU_HIP(hipMemcpyAsync(&unsorted, flags + 1, sizeof(u32), hipMemcpyDeviceToHost, stream));
  RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
  const u32* grouped = rel;
  if (unsorted) {   // else the slice is sorted by the key: rows[] is the identity and is never materialised
    rows = store->table_alloc<u32>(n);
    u32 bits = 1;
    while ((1ull << bits) <= kn) bits++;             // keys 0 .. kn (kn = joins nothing: sorts to the end)
    u32* rel_s = scratch<u32>(n); u32* iota = scratch<u32>(n);
    const size_t stb = sort_u32_temp_bytes(n, bits);
    void* stemp = scratch<unsigned char>(stb);
    launch_iota_u32(iota, n, stream);
    timed(KC_CSR_SCATTER, 0, n, nullptr, 12, nullptr, 0, 0, [&] { sort_pairs_u32_u32(rel, rel_s, iota, rows, n, bits, stemp, stb, stream); });
    grouped = rel_s;
  }
  launch_sorted_bounds(grouped, n, kn, off, stream);   // off[k] = first position with rel >= k, k = 0 .. kn (off[kn] = the rows that join something)
  RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;   // complete before other plans may see it
  st->csr_rows = rows; st->csr_off = off;
  st->dense_tried = true;
}

// The direct-address form for a build side that is NOT cached (RDFGPU_OPT_NO_TABLE_CACHE, or an intermediate with a host-known row
// count): built inside this execution, in scratch memory, when the single key turns out unique over a dense id range — the same two
// kernels as the cached form (min / max, then one store per row with a duplicate flag) and two host round trips.  4 bytes per ID
// instead of 8 bytes per SLOT at load <= 0.5: the 285 k-row property slices of BSBM-100M are 1.1 MB (resident in every XCD's L2)
// instead of an 8 MB hash table that 0.54 G random probes fetch from the Infinity Cache line by line.  false: not unique / not dense.
bool Plan::build_transient_direct(LdsJoinArgs& a, u64 n) {
  const u32* key = a.build_key[0];
  u32* mm = reinterpret_cast<u32*>(new_counter());     // {min, max}
  u32* flags = reinterpret_cast<u32*>(new_counter());  // {duplicate seen, -}
  const u32 init[2] = {0xFFFFFFFFu, 0u};
  RDFGPU_HIP(hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, stream));
  timed(KC_MINMAX, 4ull * n, 0, nullptr, 0, nullptr, 0, 0, [&] { launch_minmax_u32(key, n, mm, stream); });
  u32 got[2];
  RDFGPU_HIP(hipMemcpyAsync(got, mm, sizeof(got), hipMemcpyDeviceToHost, stream));
  RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
  const u64 range = got[0] <= got[1] ? (u64)(got[1] - got[0]) + 1 : ~0ull;
  if (range > 4 * n + 1024 || n > range) return false;   // sparse ids, or more rows than ids (some key repeats)
  const u32 kmin = got[0], kn = (u32)range;
  u32* direct = scratch<u32>(kn);
  RDFGPU_HIP(hipMemsetAsync(direct, 0xFF, (size_t)kn * sizeof(u32), stream));
  timed(KC_GDIRECT_BUILD, 0, n, nullptr, 8, nullptr, 0, 0, [&] { launch_gdirect_build(key, n, direct, kmin, kn, flags, stream); });
  u32 is_dup = 0;
  RDFGPU_HIP(hipMemcpyAsync(&is_dup, flags, sizeof(u32), hipMemcpyDeviceToHost, stream));
  RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
  if (is_dup) return false;
  metrics.tables_built++;
  a.direct = direct; a.direct_min = kmin; a.direct_n = kn;
  return true;
}

// HashJoinExec whose build side fits one workgroup's LDS: one fused kernel, optimistic output capacity.
DevTable Plan::exec_lds_join(NodeInfo& nd, const DevTable& L, const DevTable& R, bool build_left, const NodeInfo* probe_filter, const NodeInfo* post_filter) {
  // a LEFT join built on its right input preserves its PROBE side (choose_build_left): for everything below it is an inner join whose
  // kernel adds one null-extended row per unmatched probe row — no visited flags, no tail pass
  const bool probe_outer = nd.d.join_type == RDFGPU_JOIN_LEFT && !build_left;
  const bool left_join = nd.d.join_type == RDFGPU_JOIN_LEFT && build_left;
  const DevTable& B = build_left ? L : R;
  const DevTable& P = build_left ? R : L;
  DevTable t;
  t.n_cols = nd.n_proj;
  LdsJoinArgs a{};
  cur_build_table = nullptr;
  for (u32 c = 0; c < L.n_cols; c++) a.cols[c] = L.cols[c];
  for (u32 c = 0; c < R.n_cols; c++) a.cols[L.n_cols + c] = R.cols[c];
  a.n_left_cols = L.n_cols; a.n_out_cols = nd.n_proj;
  for (u32 c = 0; c < nd.n_proj; c++) a.proj[c] = nd.proj[c];
  a.build_is_left = build_left ? 1 : 0;
  a.probe_outer = probe_outer ? 1u : 0u;
  a.n_keys = nd.d.n_keys;
  u32 build_keys[RDFGPU_MAX_KEYS] = {}, probe_keys[RDFGPU_MAX_KEYS] = {};
  for (u32 k = 0; k < a.n_keys; k++) {
    build_keys[k] = build_left ? nd.d.left_keys[k] : nd.d.right_keys[k];
    probe_keys[k] = build_left ? nd.d.right_keys[k] : nd.d.left_keys[k];
    a.build_key[k] = B.cols[build_keys[k]];
    a.probe_key[k] = P.cols[probe_keys[k]];
  }
  a.n_build_dev = B.n_dev; a.n_build_cap = B.cap; a.n_probe_dev = P.n_dev; a.n_probe_cap = P.cap;
  if (B.cap >= (1ull << 30)) fail(RDFGPU_ERR_UNSUPPORTED, "build side of %llu rows", (unsigned long long)B.cap);
  u32 slots = 64;
  while (slots < 2 * B.cap) slots <<= 1;
  bool use_part = false; PartArgs part{};
  // LDS copy per workgroup vs ONE table in HBM/L2: the LDS form pays the build once per workgroup and, above
  // ~16 KiB of table, costs occupancy (a 128 KiB table = one workgroup per CU = latency-bound probes).
  const u64 lds_limit = std::min<u64>(opt.v[RDFGPU_OPT_LDS_MAX_BUILD], kLdsJoinMaxBuild);
  const bool global_table = B.cap > lds_limit;
  // Every lane of a wave waits for the longest chain among its 64 probes, so short chains matter more than a
  // small table: LDS tables get load <= 0.25 and at least 2048 slots (16 KiB), HBM tables under 1 MiB load <= 0.125.
  if (!global_table) { while ((slots < 4 * B.cap || slots < 2048) && slots < 2 * kLdsJoinMaxBuild) slots <<= 1; }
  else { while (slots < 8 * B.cap && (u64)slots * sizeof(uint2) < (1u << 20)) slots <<= 1; }
  a.tbl_mask = slots - 1;
  // The HBM table of a build side that is a pure slice of the store (a param-free scan: label, simProperty…) is the
  // same for every plan until the store changes: it is built once per store version and kept on the store.
  bool build_now = false;
  if (global_table) {
    const bool cacheable = B.stable_id != 0 && B.n_dev == nullptr && !opt.on(RDFGPU_OPT_NO_TABLE_CACHE);
    if (cacheable) {
      SliceKey sk; sk.n_keys = a.n_keys; sk.rows = B.cap;
      for (u32 k = 0; k < a.n_keys; k++) sk.key[k] = a.build_key[k];
      SliceTable* st = store->slice_table(sk);
      cur_build_table = st;
      std::unique_lock<std::mutex> building(store->slice_build_mu);
      // Dense forms first (one single key over a dense id range): direct-address if the keys are unique, CSR if not.
      // Decided once per slice; costs a few small kernels and host syncs at that time, nothing afterwards.
      if (!st->dense_tried && a.n_keys == 1 && !opt.on(RDFGPU_OPT_NO_DIRECT_TABLE)) build_dense_table(st, a.build_key[0], B.cap);
      if (st->csr_off) {
        a.csr_off = st->csr_off; a.csr_rows = st->csr_rows; a.direct_min = st->kmin; a.direct_n = st->kn;
        // lanes per probe row: a small probe side with a large fan-out is spread over the chip
        // (first execution: the table's mean rows per key stands in for the unknown fan-out)
        const u64 fan = nd.has_last ? nd.last_rows / (P.cap ? P.cap : 1) : B.cap / (st->kn ? st->kn : 1);
        // measured on the BSBM candidate join (fan-out 111): 8 lanes per row is best at 75 k and at 1.2 M probe rows alike
        // (a tiny probe side — a single query's constants — is latency-bound instead: spread each row over up to a whole wave)
        const bool tiny = P.cap < 4096;
        u32 rl = 0;
        while (rl < (tiny ? 6u : 3u) && ((tiny ? 2ull : 16ull) << rl) <= fan && (P.cap << (rl + 1)) <= (1ull << 25)) rl++;
        if (opt.v[RDFGPU_OPT_CSR_ROW_LANES_LOG2]) rl = (u32)std::min<u64>(6, opt.v[RDFGPU_OPT_CSR_ROW_LANES_LOG2] - 1);
        if (probe_outer) rl = 0;   // (one lane per probe row: the row's null-extended candidate is produced once)
        a.row_lanes_log2 = rl;
      } else if (st->direct) {
        a.direct = st->direct; a.direct_min = st->kmin; a.direct_n = st->kn;
      } else {
        if (!st->slots || st->mask != a.tbl_mask) {   // build the hash form now, under the lock, and publish it only when complete
          if (st->slots) { store->table_free(st->slots); st->slots = nullptr; }
          void* mem = store->table_alloc<uint2>(slots);
          metrics.tables_built++;
          a.gslots = static_cast<uint2*>(mem);
          RDFGPU_HIP(hipMemsetAsync(a.gslots, 0xFF, (size_t)slots * sizeof(uint2), stream));
          timed(KC_GJOIN_BUILD, 0, B.cap, B.n_dev, 4ull * a.n_keys + 8, nullptr, 0, 0, [&] { launch_gjoin_build(a, stream); });
          RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
          st->slots = mem; st->mask = a.tbl_mask;
          }
        a.gslots = static_cast<uint2*>(st->slots);
      }
    } else if (a.n_keys <= 2 && !probe_filter && !post_filter && !opt.on(RDFGPU_OPT_NO_PARTITIONED_JOIN) && B.cap >= opt.v[RDFGPU_OPT_PARTITION_MIN_BUILD] &&
               B.cap < (1ull << 31) && P.cap < (1ull << 31)) {
      use_part = true;     // radix-partition both sides; every partition's table is built in LDS (part_join.hip)
    } else {
      // a large probe side pays for two host round trips with every probe that hits a 4-byte entry in L2 instead of an 8-byte slot beyond it
      bool direct_built = false;
      if (a.n_keys == 1 && B.n_dev == nullptr && B.cap >= 4096 && P.cap >= (1ull << 22) && !nd.transient_direct_failed && !opt.on(RDFGPU_OPT_NO_DIRECT_TABLE)) {
        direct_built = build_transient_direct(a, B.cap);
        nd.transient_direct_failed = !direct_built;
      }
      if (!direct_built) {
        a.gslots = scratch<uint2>(slots);
        build_now = true;
        RDFGPU_HIP(hipMemsetAsync(a.gslots, 0xFF, (size_t)slots * sizeof(uint2), stream));
      }
    }
  }
  if (P.cap >= (1ull << 32)) fail(RDFGPU_ERR_UNSUPPORTED, "probe side of %llu rows", (unsigned long long)P.cap);
  {
    // Candidate queue per wave: a full queue costs one output reservation (a same-address atomic, ~88 per
    // microsecond chip-wide), an oversized one costs occupancy (8 queues x 8 B x entries of LDS per workgroup).
    // Sized from the matches a wave can expect out of one tile (64 x rows-per-lane probe rows), which is what a
    // workgroup of an HBM-table join sees in its whole life; "expected" = the previous execution's cardinality
    // when known, else one match per probe row.  A direct-address table has at most one match per row.
    const u32 q_env = (u32)opt.v[RDFGPU_OPT_JOIN_WAVE_Q];
    const u64 per_tile = 64ull * (u64)lds_join_items(P.cap << a.row_lanes_log2, global_table);
    const u64 expect = nd.has_last ? nd.last_rows : P.cap;
    u64 want = a.direct ? per_tile : (expect * per_tile * 3 / 2) / (P.cap ? P.cap : 1);
    u32 q = 256;
    while (q < want && q < 1024) q <<= 1;
    if (!global_table && (size_t)slots * sizeof(uint2) > 64 * 1024) q = 256;
    a.wave_q = q_env ? q_env : q;
    // partitioned join: a sparse output (the previous execution found less than one match per 8 probe rows) needs no deep queues —
    // 52 KB of LDS per workgroup instead of 64: three workgroups per CU instead of two
    if (use_part) a.wave_q = nd.has_last && !nd.last_scaled && nd.last_rows * 8 < P.cap ? 64 : 256;
  }
  a.probe_col_base = build_left ? L.n_cols : 0;
  a.has_filter = (u32)nd.shape;   // 0 none / 1 generic VM / 3 window
  if (nd.shape == 1) a.prog = upload_program(nd.prog);
  if (nd.shape == 2) { a.idp.a = nd.prog.nodes[0].u; a.idp.b = nd.prog.nodes[1].u; a.idp.is_eq = nd.prog.nodes[2].op == RDFGPU_EX_ID_EQ; }
  if (nd.shape == 3) {
    const rdfgpu_expr_node* e = nd.prog.nodes;
    auto lit = [&](u32 o) { TvLiteral l{}; l.lo = e[o + 4].lo; l.hi = e[o + 4].hi; l.aux = e[o + 4].u; l.tag = e[o + 4].tag; l.flags = e[o + 4].flags;
                            l.arith_sub = e[o + 5].op == RDFGPU_EX_SUB; l.cmp_op = e[o + 6].op; return l; };
    a.win.x0 = e[0].u; a.win.y0 = e[2].u; a.win.x1 = e[8].u; a.win.y1 = e[10].u;
    a.win.l0 = lit(0); a.win.l1 = lit(8);
  }
  a.has_probe_filter = probe_filter ? (probe_filter->shape == 1 ? 1u : 2u) : 0u;
  if (probe_filter) {
    if (probe_filter->shape == 1) {
      const rdfgpu_expr_node* e = probe_filter->prog.nodes;
      a.pid.col = a.probe_col_base + e[0].u; a.pid.lit = e[1].u; a.pid.is_eq = e[2].op == RDFGPU_EX_ID_EQ;
    } else a.probe_prog = upload_program(probe_filter->prog);
  }
  if (post_filter) {   // FilterExec of the build side's input, columns relative to that input
    const rdfgpu_expr_node* e = post_filter->prog.nodes;
    a.has_post = 1;
    a.post.col = (build_left ? 0 : L.n_cols) + e[0].u; a.post.lit = e[1].u; a.post.is_eq = e[2].op == RDFGPU_EX_ID_EQ;
  }
  a.tt = typed_table();
  a.stream_direct = opt.on(RDFGPU_OPT_NO_STREAM_JOIN) ? 0u : 1u;
  if (left_join) a.visited = scratch<u8>(L.cap);
  u64* n_out = new_counter();
  u32* overflow = reinterpret_cast<u32*>(new_counter());
  a.n_out_dev = n_out; a.overflow = overflow;
  const u64 tail = left_join ? L.cap : 0;

  // columns the kernel reads on the probe side: keys ∪ projected ∪ filter columns
  u32 probe_cols = 0, build_payload = 0;
  {
    bool pu[kMaxCols] = {}, bu[kMaxCols] = {};
    const u32 nl = L.n_cols;
    auto mark = [&](u32 c) { const bool from_left = c < nl; const u32 local = from_left ? c : c - nl; ((from_left == build_left) ? bu : pu)[local] = true; };
    for (u32 k = 0; k < a.n_keys; k++) { pu[probe_keys[k]] = true; }
    for (u32 c = 0; c < nd.n_proj; c++) mark(nd.proj[c]);
    for (u32 i = 0; i < nd.prog.n; i++) if (nd.prog.nodes[i].op == RDFGPU_EX_COLUMN) mark(nd.prog.nodes[i].u);
    if (probe_filter) for (u32 i = 0; i < probe_filter->prog.n; i++) if (probe_filter->prog.nodes[i].op == RDFGPU_EX_COLUMN) pu[probe_filter->prog.nodes[i].u] = true;
    for (u32 k = 0; k < a.n_keys; k++) bu[build_keys[k]] = false;
    for (bool b : pu) probe_cols += b;
    for (bool b : bu) build_payload += b;
  }
  // SURVEY §8d hash join: 4(k+p_b)N_b + 8N_b + 4(k+p_p)N_p + 8N_p + 4 c_o N_o  (the 8-byte slot lives in LDS here)
  const u64 fixed = (4ull * (a.n_keys + build_payload) + 8) * B.cap;

  if (use_part) {
    flush_pending_oj();
    // output of the previous execution (none: single pass): above ~50 M rows the reservations of a single pass (one
    // same-address atomic per 256 rows, ~88 per microsecond) cost more than walking every partition twice
    const u64 expect_out = nd.has_last ? nd.last_rows : 0;
    part.two_pass = expect_out >= opt.v[RDFGPU_OPT_PARTITION_TWO_PASS_ROWS] ? 1u : 0u;
    if (part.two_pass && nd.shape == 2 && !left_join) {   // `build column <=|!=> probe column`: decided during the walk (part_join.hip, INL)
      auto from_build = [&](u32 c) { return (c < L.n_cols) == build_left; };
      const u32 ca = nd.prog.nodes[0].u, cb = nd.prog.nodes[1].u;
      if (from_build(ca) != from_build(cb)) { part.inl_build = a.cols[from_build(ca) ? ca : cb]; part.inl_probe = a.cols[from_build(ca) ? cb : ca]; }
    }
    prepare_partitions(a, B, P, part);   // the build half of this HashJoinExec: inside the operator, every execution
  }
  // SURVEY 8d hash-join bytes of a partitioned join: both sides' key + payload columns and one 8-byte slot per row, the output;
  // the partition passes are in the time of the operator, not in its bytes
  const u64 part_fixed = (4ull * (a.n_keys + build_payload) + 8) * B.cap;
  bool stream_values_tried = false;
  auto launch_join = [&](int kc_lds, u64 fixed_bytes, u64 out_bytes_per_row) {
    if (use_part) timed(KC_PART_JOIN, part_fixed, P.cap, P.n_dev, 4ull * probe_cols + 8, n_out, 0, out_bytes_per_row, [&] { launch_part_join(a, part, stream); });
    else {
      const bool streamed = a.stream_direct && direct_stream_join_ok(a);
      // the streaming form over a direct table, window filter on ONE build column against probe columns, a probe side large enough to pay
      // for a kernel and a host round trip: that column decoded per KEY (a.key_vals), once per execution
      if (streamed && !stream_values_tried && a.direct && a.has_filter == 3 && a.has_probe_filter == 0 && B.n_dev == nullptr && a.tt.n_ids != 0 &&
          direct_stream_join_items(P.cap) == 8 && P.cap >= (1ull << 22) && !opt.on(RDFGPU_OPT_NO_VALUE_TABLES)) {
        stream_values_tried = true;
        auto from_build = [&](u32 c) { return (c < L.n_cols) == build_left; };
        if (a.win.x0 == a.win.x1 && from_build(a.win.x0) && !from_build(a.win.y0) && !from_build(a.win.y1)) {
          long long* vals = scratch<long long>(a.direct_n);
          u32* bad = reinterpret_cast<u32*>(new_counter());
          launch_fill_i64(vals, INT64_MIN, a.direct_n, stream);
          launch_direct_values(a.build_key[0], a.cols[a.win.x0], B.cap, a.direct_min, a.direct_n, a.tt, vals, bad, stream);
          u32 is_bad = 0;
          RDFGPU_HIP(hipMemcpyAsync(&is_bad, bad, sizeof(u32), hipMemcpyDeviceToHost, stream));
          RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
          if (!is_bad) {
            a.key_vals = vals; metrics.tables_built++;
            a.stream_need_build_row = (a.has_post && from_build(a.post.col)) ? 1u : 0u;
            for (u32 c = 0; c < nd.n_proj; c++) if (from_build(nd.proj[c])) a.stream_need_build_row = 1u;
          }
        }
      }
      timed(streamed ? (int)KC_STREAM_JOIN : kc_lds, fixed_bytes, P.cap, P.n_dev, 4ull * probe_cols + 8, n_out, 0, out_bytes_per_row, [&] { launch_lds_join(a, stream); });
    }
  };
  if (build_now)   // build pass: keys read + one 8-byte slot written per build row
    timed(KC_GJOIN_BUILD, 0, B.cap, B.n_dev, 4ull * a.n_keys + 8, nullptr, 0, 0, [&] { launch_gjoin_build(a, stream); });
  // Speculative mode (re-execution of a plan whose previous run is known): the output is sized from the
  // previous cardinality of this operator and NOTHING is waited for — the exact count stays on the device,
  // the overflow flag is checked once at the end of the plan (Plan::execute), which re-runs exactly if any
  // speculation failed.
  // First execution of a plan (DataFusion compiles a fresh plan per query): no history, but the table form bounds or
  // estimates the output — a direct-address table yields at most one match per probe row (exact bound), a CSR table
  // about its mean rows per key, a hash table is assumed unique-ish — so the join can run without a host round
  // trip as well; the overflow flag at the end of the plan catches a wrong guess (exact re-run).
  u64 first_guess = 0;
  if (speculative && !nd.has_last && !left_join && !opt.on(RDFGPU_OPT_NO_FIRST_RUN_SPECULATION)) {
    if (a.direct) first_guess = P.cap;
    else if (a.csr_off) first_guess = 2 * P.cap * ((B.cap + a.direct_n - 1) / (a.direct_n ? a.direct_n : 1)) + 1024;
    else first_guess = P.cap + 1024;
  }
  if (speculative && (nd.has_last || first_guess)) {
    // a fusable run of follow-up lookups above this join (Plan::plan_chain) executes inside this join's resolve
    // phase: the output is then the TOP node's, sized from the top node's history
    NodeInfo* size_node = &nd;
    u64 stage_bytes = 0;
    BandArgs band{}; bool use_band = false;
    if (pending_chain && pending_chain->base == &nd && !pending_chain->consumed && global_table && !use_part && !left_join && !probe_outer && !probe_filter && nd.shape != 1 &&
        apply_chain(*pending_chain, nd, L, R, build_left, a, stage_bytes, &band, &use_band)) {
      pending_chain->consumed = true;
      size_node = pending_chain->top;
      t.n_cols = a.n_out_cols;
    }
    if (!use_band) flush_pending_oj();   // (only the band join takes an ordered slice join's held-back write pass)
    const bool chained = a.n_chain != 0;
    const u64 spec_cap = nd.has_last ? std::max<u64>(1024, size_node->last_rows + size_node->last_rows / (size_node->last_scaled ? 2 : 4) + 256)   // 25 % head room over the previous run (50 % over an extrapolation)
                                     : std::max<u64>(1024, first_guess);
    a.out_cap = spec_cap;
    for (u32 c = 0; c < a.n_out_cols; c++) { a.out[c] = scratch<u32>(spec_cap + tail); t.cols[c] = a.out[c]; }
    if (left_join) RDFGPU_HIP(hipMemsetAsync(a.visited, 0, L.cap, stream));
    // A small table against a CSR slice that is sorted by another column, with an output about as large as the slice:
    // the matches are emitted in the slice's order (ordered_join.hip) — what consumes them partitioned by that column
    // (the band join above) then has nothing to sort; the chain's look-ups by table columns run once per table row.
    bool use_ordered = false;
    if (!use_band && !probe_outer && !opt.on(RDFGPU_OPT_NO_ORDERED_JOIN) && a.csr_off && !a.range_rows && !left_join && a.has_filter == 0 && !a.has_probe_filter && !a.has_post &&
        a.n_keys == 1 && B.sorted_col >= 0 && (u32)B.sorted_col != build_keys[0] && !B.n_dev && B.stable_id && B.cap < (1ull << 32) && P.cap <= (1ull << 24) &&
        size_node->has_last && size_node->last_rows * 8 >= B.cap) {
      use_ordered = true;
      for (u32 s = 0; s < a.n_chain; s++) if (a.chain[s].key.src != 0 || a.chain[s].fs != 0) use_ordered = false;
      u32 from_table = 0;   // output columns taken from the table row or a stage row travel in its packed record: at most 8
      for (u32 c = 0; c < a.n_out_cols; c++) {
        if (chained) from_table += a.chain_out[c].src != 1;
        else { const u32 pc = a.proj[c]; from_table += ((pc < a.n_left_cols) == (a.build_is_left != 0)) ? 0u : 1u; }
      }
      if (from_table > 8 || a.n_out_cols > kOjMaxOutCols) use_ordered = false;
    }
    if (use_ordered) {
      OrderedJoinArgs o{};
      o.build_key = a.build_key[0]; o.n_build = B.cap;
      o.probe_key = a.probe_key[0]; o.n_probe_dev = P.n_dev; o.n_probe_cap = P.cap;
      o.kmin = a.direct_min; o.kn = a.direct_n;
      o.head = scratch<uint2>(a.direct_n); o.next = scratch<u32>(P.cap);
      RDFGPU_HIP(hipMemsetAsync(o.head, 0xFF, (size_t)a.direct_n * sizeof(uint2), stream));
      o.n_stages = a.n_chain;
      for (u32 s = 0; s < a.n_chain; s++) o.stage[s] = OrderedJoinStage{a.chain[s].key.ptr, a.chain[s].direct, a.chain[s].kmin, a.chain[s].kn, scratch<u32>(P.cap)};
      o.n_out_cols = a.n_out_cols;
      u32 n_words = 0;
      for (u32 c = 0; c < a.n_out_cols; c++) {
        if (chained) o.out_ref[c] = a.chain_out[c];
        else { const u32 pc = a.proj[c]; const bool from_left = pc < a.n_left_cols; o.out_ref[c] = ColRef{a.cols[pc], (from_left == (a.build_is_left != 0)) ? 1u : 0u, 0u}; }
        o.out[c] = a.out[c];
        o.out_slot[c] = o.out_ref[c].src == 1 ? (u8)0xFF : (u8)n_words++;
        if (o.out_ref[c].src == 1 && o.out_ref[c].ptr == B.cols[B.sorted_col] && t.sorted_col < 0) { t.sorted_col = (int)c; t.key_min = B.key_min; t.key_max = B.key_max; }
      }
      o.n_rec = n_words > 4 ? 2u : 1u;
      o.trec = scratch<uint4>(P.cap * o.n_rec);
      o.out_cap = spec_cap; o.n_out_dev = n_out; o.overflow = overflow;
      const u64 tiles = ordered_join_tiles(B.cap);
      o.tile_count = scratch<u32>(tiles + 1); o.tile_off = scratch<u32>(tiles + 1);
      o.row_head = scratch<u32>(B.cap); o.row_cnt = scratch<unsigned char>(B.cap);
      const size_t tb = scan_temp_bytes(tiles + 1);
      void* temp = scratch<unsigned char>(tb);
      timed(KC_OJ_PROBE, 0, P.cap, P.n_dev, 8 + 12ull * a.n_chain, nullptr, 0, 0, [&] { launch_ordered_join_probe(o, stream); });
      timed(KC_OJ_COUNT, 0, B.cap, nullptr, 4, nullptr, 0, 0, [&] { launch_ordered_join_count(o, stream); });
      timed(scan_class(tiles + 1), 0, tiles + 1, nullptr, 8, nullptr, 0, 0, [&] { exclusive_scan_u32(o.tile_count, o.tile_off, tiles + 1, temp, tb, stream); });
      // the consumer is a band join that (last time) found this output sorted by its key and needed nothing else of it: the write
      // pass is held back — that join has it write its row records instead of this table (exec_band_join), anything else flushes it
      const int consumer = size_node->parent;
      if (consumer >= 0 && nodes[consumer].band_takes_records && !pending_oj.active && t.sorted_col >= 0) {
        pending_oj.active = true; pending_oj.o = o; pending_oj.first_col = a.out[0]; pending_oj.n_build = B.cap; pending_oj.n_chain = a.n_chain;
      } else timed(KC_OJ_WRITE, 0, B.cap, nullptr, 4, n_out, 0, 8ull * a.n_out_cols, [&] { launch_ordered_join_write(o, stream); });
    }
    else if (use_band) { cur_band_node = &nd; exec_band_join(a, band, B, P, 4ull * (1 + build_payload), 4ull * probe_cols); cur_band_node = nullptr; }
    else launch_join(lds_join_class(a.has_filter, a.has_probe_filter, lds_join_items(P.cap << a.row_lanes_log2, global_table), use_part ? kJoinTableLds : lds_join_mode(a), chained),
                     (global_table ? 0 : fixed) + stage_bytes, 4ull * a.n_out_cols);
    spec_checks.push_back({size_node, (u32)(n_out - counters), left_join});
    t.cap = spec_cap + tail; t.n_dev = n_out;
    if (left_join) {
      JoinArgs ja{};
      for (u32 c = 0; c < L.n_cols; c++) ja.left[c] = L.cols[c];
      ja.n_left_cols = L.n_cols; ja.n_right_cols = R.n_cols; ja.n_out_cols = nd.n_proj;
      for (u32 c = 0; c < nd.n_proj; c++) { ja.proj[c] = nd.proj[c]; ja.out[c] = a.out[c]; }
      ja.n_left_dev = L.n_dev; ja.n_left_cap = L.cap;
      ja.visited = a.visited; ja.n_out_dev = n_out; ja.matched_total = spec_cap + tail;
      timed(KC_LEFT_TAIL, 0, L.cap, L.n_dev, 1, nullptr, 0, 0, [&] { launch_join_left_unmatched(ja, stream); });
    }
    return t;
  }
  flush_pending_oj();
  u64 out_cap = P.cap < 1024 ? 1024 : P.cap;   // optimistic: at most one match per probe row on average
  if (a.csr_off) out_cap = std::max<u64>(out_cap, 2 * P.cap * ((B.cap + a.direct_n - 1) / (a.direct_n ? a.direct_n : 1)) + 1024);   // CSR: twice the mean rows per key
  u64 total = 0;
  for (int attempt = 0; attempt < 2; attempt++) {
    a.out_cap = out_cap;
    for (u32 c = 0; c < nd.n_proj; c++) { a.out[c] = scratch<u32>(out_cap + tail); t.cols[c] = a.out[c]; }
    if (left_join) RDFGPU_HIP(hipMemsetAsync(a.visited, 0, L.cap, stream));
    launch_join(lds_join_class(a.has_filter, a.has_probe_filter, lds_join_items(P.cap << a.row_lanes_log2, global_table), use_part ? kJoinTableLds : lds_join_mode(a)),
                global_table ? 0 : fixed, 4ull * nd.n_proj);
    const u32 i0 = (u32)(n_out - counters);
    RDFGPU_HIP(hipMemcpyAsync(ctx->counters_host + i0, counters + i0, 2 * sizeof(u64), hipMemcpyDeviceToHost, stream));
    RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
    total = ctx->counters_host[i0];
    const bool ovf = (ctx->counters_host[i0 + 1] & 0xFFFFFFFFull) != 0;
    if (!ovf) break;
    if (attempt == 1) fail(RDFGPU_ERR_DEVICE, "LDS join overflowed its exact-size output");
    out_cap = total;   // the count is exact even when the writes did not fit: run again with room for all
    RDFGPU_HIP(hipMemsetAsync(n_out, 0, 2 * sizeof(u64), stream));
  }
  nd.last_rows = total; nd.has_last = true; nd.last_scaled = false;   // history for the next (speculative) execution
  t.cap = total + tail;
  if (!left_join) { if (total == 0) t.cap = 0; return t; }
  // left join tail: unmatched build rows, nulls on the right
  JoinArgs ja{};
  for (u32 c = 0; c < L.n_cols; c++) ja.left[c] = L.cols[c];
  ja.n_left_cols = L.n_cols; ja.n_right_cols = R.n_cols; ja.n_out_cols = nd.n_proj;
  for (u32 c = 0; c < nd.n_proj; c++) { ja.proj[c] = nd.proj[c]; ja.out[c] = a.out[c]; }
  ja.n_left_dev = L.n_dev; ja.n_left_cap = L.cap;
  ja.visited = a.visited; ja.n_out_dev = n_out;
  timed(KC_LEFT_TAIL, 0, L.cap, L.n_dev, 1, nullptr, 0, 0, [&] { launch_join_left_unmatched(ja, stream); });
  t.n_dev = n_out;
  return t;
}

// Radix partitioning of both sides of a HashJoinExec by the top bits of the key hash (part_join.hip): per side one pass
// that computes (partition, {row, key0, key1}) per row, one rocPRIM radix sort moving the 12-byte records, one pass that
// finds the partition boundaries.  Rows with a null key (NullEqualsNothing) or beyond the live row count ride in the last
// partition, marked (row = kNil) so that the join skips them.
void Plan::prepare_partitions(const LdsJoinArgs& a, const DevTable& B, const DevTable& P, PartArgs& pa) {
  u32 bits = 0;
  u32 target_rows = kPartTargetRows, part_slots = kPartSlots;
  // a join with a large output is bound by the latency of its gathers (part_join.hip, BIG): a 2048-slot table (24 KB) lets three
  // workgroups share a CU instead of two — 8.25 -> 7.5 ms on the 0.54 G-row candidate join of BSBM Q5 (profiles/tools/nc_variants.py)
  if (pa.two_pass) { part_slots = 2048; target_rows = 512; }
  if (opt.v[RDFGPU_OPT_PARTITION_SLOTS]) {
    const u64 v = opt.v[RDFGPU_OPT_PARTITION_SLOTS];
    if (v < 1024 || v > 8192 || (v & (v - 1))) fail(RDFGPU_ERR_INVALID, "PARTITION_SLOTS = %llu (a power of two from 1024 to 8192)", (unsigned long long)v);
    part_slots = (u32)v; target_rows = part_slots / 4;
  }
  if (opt.v[RDFGPU_OPT_PARTITION_ROWS]) target_rows = (u32)std::min<u64>(opt.v[RDFGPU_OPT_PARTITION_ROWS], part_slots / 2);
  while (bits < 16 && (B.cap >> bits) > target_rows) bits++;   // <= 16 bits = two radix passes; larger partitions are joined chunk by chunk
  const u32 n_parts = 1u << bits;
  pa.n_parts = n_parts; pa.chunk = part_slots / 2; pa.tbl_mask = part_slots - 1;
  // The probe side is a store slice sorted by one of the join keys (and every row of it is live): its partitions are KEY RANGES
  // of that column — contiguous pieces of the slice, found by one binary search per partition and read in place; only the
  // build side goes through the partition sort (LUBM Q9's closing join: 98 M of 327 M rows).
  PartKeyRange kr{-1, 0u, 0u, 0u, 0u, nullptr};
  if (!opt.on(RDFGPU_OPT_NO_RANGE_PARTITION) && P.sorted_col >= 0 && !P.n_dev && P.cap && P.key_min >= 1 && P.key_max >= P.key_min && n_parts >= 4)
    for (u32 k = 0; k < a.n_keys && kr.range < 0; k++)
      if (P.cols[P.sorted_col] == a.probe_key[k]) {
        // id ranges are not row ranges (LUBM: undergraduate and graduate courses share the slice, at different rows per id):
        // a coarse directory over the id range hands every bucket partitions in proportion to the slice rows in it
        const u64 span = (u64)P.key_max - P.key_min + 1;
        const u32 n_coarse = std::min<u32>(4096u, n_parts / 4);
        u32 cshift = 0;
        while (((span - 1) >> cshift) >= n_coarse) cshift++;
        uint2* dir = scratch<uint2>((u64)n_coarse + 1);
        kr = PartKeyRange{(int)k, P.key_min, P.key_max, cshift, n_coarse, dir};
        timed(KC_BAND_BOUNDS, 0, n_coarse, nullptr, 0, nullptr, 0, 0, [&] { launch_part_equalise(a.probe_key[k], P.cap, kr, n_parts, dir, stream); });
      }
  auto side = [&](const DevTable& T, const u32* const* keys, const PartRec*& recs, const u32*& start, PartKeyRange r) {
    const u64 n = T.cap;
    u32* st = scratch<u32>((u64)n_parts + 2);
    if (!opt.on(RDFGPU_OPT_NO_OWN_PARTITION_PASS)) {   // hand-written MSD passes that recompute the partition from the keys (part_pass.hip)
      const PartPassPlan pl = part_pass_plan(n, bits);
      PartPassBuffers w{};
      w.recs = scratch<PartRec>(n); w.recs_a = pl.two ? scratch<PartRec>(n) : nullptr;
      w.pid16 = scratch<unsigned short>(n); w.digit = pl.two ? scratch<unsigned char>(n) : nullptr;
      w.hist_a = scratch<u32>(pl.hist_a);
      w.total = scratch<u32>((u64)std::max<u32>(pl.nb_a, n_parts) + 2); w.base_a = scratch<u32>((u64)pl.nb_a + 2);
      if (pl.two) {
        w.hist_b = scratch<u32>(pl.hist_b);
        w.tiles_b = scratch<unsigned char>(pl.tile_desc_bytes); w.n_tiles_b = scratch<u32>(1);
      }
      w.tb = scratch<u32>((u64)pl.nb_a + 2);
      w.start = st;
      const size_t tb = scan_temp_bytes((u64)n_parts + 2);
      void* temp = scratch<unsigned char>(tb);
      timed(KC_PART_PASS, 0, n, T.n_dev, 0, nullptr, 0, 0, [&] { part_pass_run(w, pl, keys[0], a.n_keys > 1 ? keys[1] : nullptr, a.n_keys, T.n_dev, n, bits, n_parts, r, temp, tb, stream); });
      recs = w.recs; start = st;
      return;
    }
    u32* skey_in = scratch<u32>(n); u32* skey = scratch<u32>(n);
    PartRec* sval_in = scratch<PartRec>(n); PartRec* sval = scratch<PartRec>(n);
    const size_t tb = part_sort_temp_bytes(n, bits ? bits : 1);
    void* temp = scratch<unsigned char>(tb);
    timed(KC_PART_KEYS, 0, n, T.n_dev, 0, nullptr, 0, 0, [&] { launch_part_keys(keys[0], a.n_keys > 1 ? keys[1] : nullptr, a.n_keys, T.n_dev, n, bits, n_parts, r, skey_in, sval_in, stream); });
    timed(KC_RADIX_SORT, 0, n, nullptr, 0, nullptr, 0, 0, [&] { part_sort(skey_in, skey, sval_in, sval, n, bits ? bits : 1, temp, tb, stream); });
    timed(KC_BAND_BOUNDS, 0, n_parts, nullptr, 0, nullptr, 0, 0, [&] { launch_sorted_bounds(skey, n, n_parts, st, stream); });
    recs = sval; start = st;
  };
  side(B, a.build_key, pa.bpart, pa.bstart, kr);
  if (kr.range >= 0) {
    u32* st = scratch<u32>((u64)n_parts + 2);
    timed(KC_BAND_BOUNDS, 0, n_parts, nullptr, 0, nullptr, 0, 0, [&] { launch_part_range_bounds(a.probe_key[kr.range], P.cap, kr, n_parts, st, stream); });
    pa.ppart = nullptr; pa.pstart = st; pa.pcol0 = a.probe_key[0]; pa.pcol1 = a.n_keys > 1 ? a.probe_key[1] : nullptr;
  } else side(P, a.probe_key, pa.ppart, pa.pstart, kr);
}

// The fused chain as a key-partitioned band join (band_join.hip).  `a` is complete (chain, output columns, out_cap,
// counters); everything allocated here is scratch of this execution.  Bytes recorded per kernel = what that kernel has
// to move once (compulsory): decode reads the probe columns and writes the records, the mask kernel reads the records
// and the group entries with their stage look-ups and writes one bit per pair, the emit kernel reads the bits and
// writes the output.
void Plan::exec_band_join(LdsJoinArgs& a, BandArgs& b, const DevTable& B, const DevTable& P, u64 build_bytes_per_row, u64 probe_bytes_per_row) {
  const u32 kn = a.direct_n;
  const u64 np = P.cap, nb = B.cap;
  // the probe side arrives sorted by the key (an ordered slice join below, ordered_join.hip): nothing to partition.  Keys
  // below the table's range would map to "joins nothing" (= kn, the largest) out of order: the slice's id range rules that out.
  bool presorted = false;
  if (P.sorted_col >= 0 && P.cols[P.sorted_col] == a.probe_key[0] && P.key_min >= std::max<u32>(1u, a.direct_min) && !opt.on(RDFGPU_OPT_NO_ORDERED_JOIN)) presorted = true;
  b.presorted = presorted ? 1u : 0u;
  // the full-semantics pass is launched when the previous execution met a row that needed it (or there was none); a row that
  // needs it after all is caught at the end of the plan like any failed speculation
  const bool skip_slow = speculative && cur_band_node && cur_band_node->band_ran && cur_band_node->band_slow_rows == 0;
  // The probe side is the held-back output of an ordered slice join (Plan::pending_oj): if everything this join reads of it
  // travels in that join's packed table record — the window operands, the id operand, the rows' output values — and its key is
  // the slice's sorted column, that join writes this join's row records itself (OjBandFuse) and the table in between is never
  // written.  Otherwise its write pass runs now.
  OjBandFuse fuse{}; bool fused = false;
  if (pending_oj.active) {
    const OrderedJoinArgs& o = pending_oj.o;
    bool ok = presorted && skip_slow && P.cols[0] == pending_oj.first_col && P.n_cols == o.n_out_cols;
    auto slot_of = [&](const u32* col, u8& slot) {      // the word of the packed record that holds output column `col`
      slot = 0xFFu;
      for (u32 c = 0; c < o.n_out_cols; c++) if (o.out[c] == col && o.out_slot[c] != 0xFFu) { slot = o.out_slot[c]; return true; }
      return false;
    };
    fuse.y0_slot[0] = fuse.y0_slot[1] = fuse.y1_slot[0] = fuse.y1_slot[1] = fuse.neq_slot = fuse.row_slot[0] = fuse.row_slot[1] = 0xFFu;
    for (u32 w = 0; ok && w < b.n_win; w++) ok = slot_of(b.win[w].y0, fuse.y0_slot[w]) && slot_of(b.win[w].y1, fuse.y1_slot[w]);
    if (ok && b.has_neq) ok = slot_of(b.neq_probe, fuse.neq_slot);
    for (u32 u = 0; ok && u < b.n_row_cols; u++) ok = slot_of(b.row_col[u], fuse.row_slot[u]);
    fuse.key_col = nullptr;
    for (u32 c = 0; ok && c < o.n_out_cols; c++) if (o.out[c] == a.probe_key[0] && o.out_slot[c] == 0xFFu && o.out_ref[c].src == 1) fuse.key_col = o.out_ref[c].ptr;
    ok = ok && fuse.key_col != nullptr;
    if (ok) fused = true; else flush_pending_oj();
  }
  if (cur_band_node) cur_band_node->band_takes_records = presorted && skip_slow && !opt.on(RDFGPU_OPT_NO_ORDERED_JOIN);
  // small probe side, not sorted: counting sort on the key (band_scatter_kernel) instead of rocPRIM's radix sort; a larger one
  // when the previous execution found it piecewise sorted (>= 4 rows per run of equal neighbouring keys: the N sorted runs
  // a repartition delivers) — one atomic per run, rows of a run scattered together
  const u64 runs_seen = cur_band_node ? (cur_band_node->band_run_stats & 0xFFFFFFFFull) : 0, run_rows = cur_band_node ? (cur_band_node->band_run_stats >> 32) : 0;
  const bool counting = !presorted && !opt.on(RDFGPU_OPT_NO_ORDERED_JOIN) && (np <= (1ull << 21) || (runs_seen && run_rows >= 4 * runs_seen));
  if (counting) {
    b.key_hist = scratch<u32>((u64)kn + 2); b.key_cursor = scratch<u32>((u64)kn + 2);
    RDFGPU_HIP(hipMemsetAsync(b.key_hist, 0, ((size_t)kn + 2) * sizeof(u32), stream));
  }
  b.csr_off = a.csr_off; b.csr_rows = a.csr_rows; b.kmin = a.direct_min; b.kn = kn;
  b.probe_key = a.probe_key[0]; b.n_probe_dev = P.n_dev; b.n_probe_cap = np;
  b.tt = a.tt;
  b.n_out_cols = a.n_out_cols; b.out_cap = a.out_cap; b.n_out_dev = a.n_out_dev; b.overflow = a.overflow;
  for (u32 c = 0; c < a.n_out_cols; c++) b.out[c] = a.out[c];
  u32 bits = 1;
  while ((1ull << bits) <= kn) bits++;            // keys 0 .. kn (kn = joins nothing)
  u32* skey = scratch<u32>(np); u32* perm = scratch<u32>(np);
  if (presorted) { b.skey_in = skey; b.sval_in = perm; b.rec = nullptr; }
  // the packed pair test reads 8 bytes of window, the id operand and at most one output value per row: 16 bytes per row instead of 32
  // whenever no full-semantics pass will want the flags (decode pass and ordered-join records alike)
  b.compact = (b.pack16 && b.n_row_cols <= 1 && skip_slow && !opt.on(RDFGPU_OPT_NO_BAND_COMPACT)) ? 1u : 0u;
  if (!presorted) { b.skey_in = scratch<u32>(np); b.sval_in = scratch<u32>(np); b.rec = scratch<uint4>((b.compact ? 1 : 2) * np); }
  b.skey = skey; b.perm = perm;
  b.rec_s = scratch<uint4>(np); b.aux_s = scratch<uint4>(np);
  b.slow_rows = reinterpret_cast<u32*>(new_counter());
  b.run_stats = reinterpret_cast<unsigned long long*>(new_counter());
  const size_t stb = sort_u32_temp_bytes(np, bits);
  void* stemp = scratch<unsigned char>(stb);
  u64 entry_bytes = build_bytes_per_row + (a.csr_rows ? 4 : 0) + (a.has_post ? 4 : 0);
  for (u32 t = 0; t < a.n_chain; t++) entry_bytes += 4 + (a.chain[t].val ? 8 : 0);
  // The decoded entries depend on the store's slices and the chain's constants only: kept with the slice's CSR table
  // (per store version, like every other join table), so a steady-state step does not decode 5.4 M build rows again.
  std::string ekey;
  {
    auto put = [&](const void* p, size_t n) { ekey.append(reinterpret_cast<const char*>(p), n); };
    put(&b.csr_off, sizeof b.csr_off); put(&b.csr_rows, sizeof b.csr_rows); put(&b.kmin, 4); put(&b.kn, 4); put(&b.n_stages, 4); put(&b.n_win, 4);
    for (u32 t = 0; t < b.n_stages; t++) put(&b.stage[t], sizeof(BandStage));
    for (u32 w = 0; w < b.n_win; w++) { put(&b.win[w].key_col, sizeof(void*)); put(&b.win[w].val, sizeof(void*)); put(&b.win[w].vkmin, 4); put(&b.win[w].vkn, 4); put(&b.win[w].vbase, 8); }
    put(&b.has_post, 4); put(&b.post_lit, 4); put(&b.post_is_eq, 4); put(&b.post_col, sizeof(void*)); put(&b.has_neq, 4); put(&b.neq_build, sizeof(void*));
    put(&b.n_entry_cols, 4);
    for (u32 u = 0; u < b.n_entry_cols; u++) { put(&b.entry_col[u].ptr, sizeof(void*)); put(&b.entry_col[u].src, 4); }
    put(&nb, 8);
  }
  bool have_entries = false;
  const bool cache_entries = cur_build_table && B.stable_id != 0 && !opt.on(RDFGPU_OPT_NO_TABLE_CACHE);
  std::unique_lock<std::mutex> entries_lock(store->slice_build_mu, std::defer_lock);
  if (cache_entries) {
    entries_lock.lock();
    for (const auto& e : cur_build_table->band_entries) if (e.key == ekey) { b.et = e.et; for (u32 u = 0; u < b.n_entry_cols; u++) b.eo[u] = e.eo[u]; have_entries = true; }
  }
  b.n_entries = nb;
  if (!have_entries) {
  // stages all keyed by one build column: their look-ups once per distinct key value instead of once per entry
  {
    const u32* kc = a.n_chain ? b.stage[0].key_col : nullptr;
    bool same = kc != nullptr;
    u64 lo = ~0ull, hi = 0;
    for (u32 t = 0; t < b.n_stages; t++) { same = same && b.stage[t].key_col == kc; lo = std::min<u64>(lo, b.stage[t].kmin); hi = std::max<u64>(hi, (u64)b.stage[t].kmin + b.stage[t].kn); }
    for (u32 w = 0; w < b.n_win; w++) same = same && b.win[w].key_col == kc;
    if (same && hi > lo && hi - lo <= (64ull << 20) && hi - lo <= 8 * nb + 1024) {
      b.pt_min = (u32)lo; b.pt_n = (u32)(hi - lo); b.pt_key_col = kc;
      b.pt = scratch<uint4>(2ull * b.pt_n);
      timed(KC_BAND_PT, 0, b.pt_n, nullptr, 4ull * b.n_stages + 8ull * b.n_win + 4ull * b.n_entry_cols + 32, nullptr, 0, 0, [&] { launch_band_pt(b, stream); });
    }
  }
  // the build side, once: per row its columns + stage look-ups read, 16 B of operands + the output values written
  const bool own_entries = cache_entries && cur_build_table->band_entries.size() < 8;
  if (own_entries) {
    b.et = store->table_alloc<uint4>(nb + 64);
    for (u32 u = 0; u < b.n_entry_cols; u++) b.eo[u] = store->table_alloc<u32>(nb);
    metrics.tables_built++;
  } else {
    b.et = scratch<uint4>(nb + 64);                  // padded: the pair test reads whole groups of 8 entries
    for (u32 u = 0; u < b.n_entry_cols; u++) b.eo[u] = scratch<u32>(nb);
  }
  timed(KC_BAND_ENTRIES, 0, nb, B.n_dev, entry_bytes + 4ull * b.n_entry_cols + 16 + 4ull * b.n_entry_cols, nullptr, 0, 0, [&] { launch_band_entries(b, stream); });
    if (cache_entries && cur_build_table->band_entries.size() < 8) {   // publish only when complete
      RDFGPU_HIP(hipStreamSynchronize(stream)); metrics.host_syncs++;
      SliceTable::BandEntries e{ekey, b.et, {nullptr, nullptr, nullptr, nullptr}};
      for (u32 u = 0; u < b.n_entry_cols; u++) e.eo[u] = b.eo[u];
      cur_build_table->band_entries.push_back(e);
    }
  }
  if (entries_lock.owns_lock()) entries_lock.unlock();
  // per probe row: key + the window operands + the id operand read, 24 B of record + 8 B of sort pair written
  b.poff = scratch<u32>((u64)kn + 2);
  // blocks: sum over keys of ceil(E/64) * ceil(R/64) <= cmax * (rows / 64) + sum of ceil(E/64) over the keys
  const u64 cmax = (cur_build_table->csr_max_group + 63) / 64;
  const u64 max_blocks = cmax * (np / 64 + 1) + nb / 64 + kn + 1;
  if (max_blocks >= (1ull << 31)) fail(RDFGPU_ERR_UNSUPPORTED, "band join of %llu blocks", (unsigned long long)max_blocks);
  b.max_blocks = (u32)max_blocks;
  b.bcount = scratch<u32>(max_blocks + 1); b.bofs = scratch<u32>(max_blocks + 1);   // (bcount is zeroed by the decode pass: a memset is two more launches, ~10 us of launch gap each on this part)
  if (fused) {
    const OrderedJoinArgs& o = pending_oj.o;
    pending_oj.active = false;
    // the packed pair test reads 8 bytes of window, the id operand and at most one output value per row: 16 bytes per match instead of 32
    fuse.compact = (b.pack16 && b.n_row_cols <= 1 && skip_slow && !opt.on(RDFGPU_OPT_NO_BAND_COMPACT)) ? 1 : 0;
    b.compact = fuse.compact;
    // `entry id != row id` by entry index: the band join's groups are the rows of the very slice the ordered join streamed (same sorted column, same
    // rows, identity CSR), the entry's id is that join's build key, the row's id its probe key — equal keys are what made the match, and a store slice
    // holds every (key, sorted column) pair once: the only entry of the group whose id equals the row's is the slice row the match came from
    fuse.self_index = 0;
    if (fuse.compact && b.has_neq && !b.neq_is_eq && b.csr_rows == nullptr && a.build_key[0] == fuse.key_col && B.cap == pending_oj.n_build &&
        b.neq_build == o.build_key && B.stable_id != 0 && !opt.on(RDFGPU_OPT_NO_BAND_COMPACT))
      for (u32 c = 0; c < o.n_out_cols; c++)
        if (o.out[c] == b.neq_probe && o.out_slot[c] != 0xFFu && o.out_ref[c].src == 0 && o.out_ref[c].ptr == o.probe_key) fuse.self_index = 1;
    b.neq_self = fuse.self_index;
    fuse.brec = scratch<uint4>((fuse.compact ? 1 : 2) * o.n_probe_cap);
    fuse.rec_s = b.rec_s; fuse.aux_s = b.aux_s; fuse.poff = b.poff; fuse.bcount = b.bcount; fuse.max_blocks = b.max_blocks;
    fuse.kmin = b.kmin; fuse.kn = b.kn;
    // per table row: its packed record read + two typed-value gathers + 32 B written; per slice row the count pass's 5 bytes + its
    // key; per match a 32-byte record gathered and stored
    const u64 rec_bytes = fuse.compact ? 16 : 32;
    timed(KC_OJ_BAND_RECORDS, 0, o.n_probe_cap, o.n_probe_dev, 16ull * o.n_rec + 9ull * b.n_win + rec_bytes, nullptr, 0, 0, [&] { launch_oj_band_records(o, b, fuse, stream); });
    timed(KC_OJ_WRITE_BAND, 0, pending_oj.n_build, nullptr, 4 + 1 + 4, a.n_probe_dev, 0, 2 * rec_bytes, [&] { launch_ordered_join_write_band(o, fuse, stream); });
  } else
  timed(KC_BAND_DECODE, 0, np, P.n_dev, 4 + 4ull * (b.n_win + b.has_neq) + 9ull * b.n_win + 24 + 8, nullptr, 0, 0, [&] { launch_band_decode(b, stream); });
  // the partition pass: in the time, not in the algorithmic bytes
  if (!presorted && !counting) timed(KC_RADIX_SORT, 0, np, nullptr, 0, nullptr, 0, 0, [&] { sort_pairs_u32_u32(b.skey_in, skey, b.sval_in, perm, np, bits, stemp, stb, stream); });
  b.boff = scratch<u32>((u64)kn + 1);
  // the block kernels launch one wave per block: sized from the previous execution's count (+ 25 %), not from the upper bound
  b.n_blocks_out = new_counter();
  band_block_counters.push_back({cur_band_node, (u32)(b.n_blocks_out - counters), (u32)(reinterpret_cast<u64*>(b.slow_rows) - counters), (u32)(reinterpret_cast<u64*>(b.run_stats) - counters), skip_slow});
  const u64 hist = cur_band_node ? cur_band_node->band_blocks : 0;
  b.launch_blocks = (u32)std::min<u64>(max_blocks, hist ? hist + hist / 4 + 1024 : max_blocks);
  b.bdesc = scratch<uint4>(max_blocks);
  b.masks = scratch<u64>(max_blocks * 64);
  const size_t tb = std::max(scan_temp_bytes(std::max<u64>((u64)kn + 1, max_blocks + 1)), band_blocks_scan_temp_bytes(kn));
  void* temp = scratch<unsigned char>(tb);
  if (counting) {   // poff = exclusive scan of the rows per key (entry kn = the rows that join something); then the scatter
    timed(scan_class((u64)kn + 1), 0, (u64)kn + 1, nullptr, 8, nullptr, 0, 0, [&] { exclusive_scan_u32(b.key_hist, b.poff, (u64)kn + 1, temp, tb, stream); });
    RDFGPU_HIP(hipMemcpyAsync(b.key_cursor, b.poff, ((size_t)kn + 1) * sizeof(u32), hipMemcpyDeviceToDevice, stream));
    timed(KC_BAND_ROWS, 0, np, P.n_dev, 8 + 32 + 32, nullptr, 0, 0, [&] { launch_band_scatter(b, stream); });
  } else if (!presorted) timed(KC_BAND_BOUNDS, 0, np, nullptr, 4, nullptr, 0, 0, [&] { launch_band_bounds(skey, np, kn, b.poff, stream); });   // (presorted: the decode pass wrote poff)
  timed(scan_class((u64)kn + 1), 12ull * kn, (u64)kn + 1, nullptr, 4, nullptr, 0, 0, [&] { band_blocks_scan(a.csr_off, b.poff, kn, b.boff, temp, tb, stream); });   // (blocks per key: the scan's input iterator)
  timed(KC_BAND_DESC, 12ull * kn, 0, nullptr, 0, nullptr, 0, 0, [&] { launch_band_desc(b, stream); });
  if (!presorted && !counting) timed(KC_BAND_ROWS, 0, np, P.n_dev, 4 + 32 + 32, nullptr, 0, 0, [&] { launch_band_rows(b, stream); });
  // per probe row 4 (sorted position) + 24 (record) read, per entry 16 B read, per pair one bit written; the pair count
  // is not known on the host
  timed(KC_BAND_MASK, 16ull * nb, np, P.n_dev, b.compact ? 16 : 4 + 24, nullptr, 0, 0, [&] { launch_band_mask(b, stream); });
  if (!skip_slow) {
    // the full-semantics pass needs the chain's literals and columns: the fused join kernel's argument block, by pointer
    static_assert(sizeof(LdsJoinArgs) <= ExecContext::kArgBytes, "argument staging slot too small");
    const u32 slot = arg_slots_used++;
    LdsJoinArgs* a_host = reinterpret_cast<LdsJoinArgs*>(ctx->args_host + (size_t)slot * ExecContext::kArgBytes);
    LdsJoinArgs* a_dev = reinterpret_cast<LdsJoinArgs*>(ctx->args_dev + (size_t)slot * ExecContext::kArgBytes);
    *a_host = a;
    RDFGPU_HIP(hipMemcpyAsync(a_dev, a_host, sizeof(LdsJoinArgs), hipMemcpyHostToDevice, stream));
    timed(KC_BAND_SLOW, 0, 0, nullptr, 0, nullptr, 0, 0, [&] { launch_band_slow(a_dev, b, stream); });
  }
  timed(scan_class(max_blocks + 1), 0, max_blocks + 1, nullptr, 8, nullptr, 0, 0, [&] { exclusive_scan_u32(b.bcount, b.bofs, max_blocks + 1, temp, tb, stream); });
  timed(KC_BAND_EMIT, 4ull * b.n_entry_cols * nb, np, P.n_dev, 4 + 4ull * b.n_row_cols, a.n_out_dev, 0, 4ull * a.n_out_cols, [&] { launch_band_emit(b, stream); });
}

void Plan::ensure_host_copy() {
  if (host_valid) return;
  if (!executed) fail(RDFGPU_ERR_INVALID, "plan has not been executed");
  store->activate();
  host_cols.assign(result.n_cols, std::vector<u32>());
  for (u32 c = 0; c < result.n_cols; c++) {
    host_cols[c].resize(result_rows);
    if (result_rows) RDFGPU_HIP(hipMemcpyAsync(host_cols[c].data(), result.cols[c], result_rows * 4, hipMemcpyDeviceToHost, stream));
  }
  RDFGPU_HIP(hipStreamSynchronize(stream));
  host_valid = true;
}

}  // namespace rdfgpu
