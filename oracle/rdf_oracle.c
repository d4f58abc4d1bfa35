/*
 * rdf_oracle.c — CPU restatement of the reference's scan / join / FILTER path.
 * TEST INFRASTRUCTURE ONLY (see rdf_oracle.h).  Each function cites the reference
 * file:line (relative to the rdf-fusion tree) whose behaviour it restates.
 */
#define _GNU_SOURCE
#include "rdf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __int128 i128;
typedef uint32_t u32;
typedef uint64_t u64;

static __thread char g_err[256];
const char* orc_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

/* ------------------------------------------------------------------------------------ */
/* Index components (lib/storage/src/index/components.rs:63-85)                          */
/* PERM[c][k] = GSPO position stored at level k of index c                               */
/* ------------------------------------------------------------------------------------ */
static const int PERM[3][4] = {
    {0, 1, 2, 3}, /* GSPO */
    {0, 2, 3, 1}, /* GPOS */
    {0, 3, 1, 2}, /* GOSP */
};

typedef struct { u32 c[4]; } quad;

static int quad_cmp(const void* a, const void* b) {
  const quad* x = (const quad*)a; const quad* y = (const quad*)b;
  for (int k = 0; k < 4; k++) { if (x->c[k] != y->c[k]) return x->c[k] < y->c[k] ? -1 : 1; }
  return 0;
}

typedef struct {
  u64 n;
  u32* col[4]; /* index-order columns, lexicographically sorted, unique; value 0 == null */
} perm_index;

struct orc_store {
  u32 batch_size; /* row-group size, MemIndexConfiguration::batch_size quad_index.rs:16-25 */
  perm_index idx[3];
  rdfgpu_typed_value* tv; u64 n_ids;
  i128* dec; u64 n_dec;
  int faithful_decode;
  u32* hm_keys; u32* hm_vals; u64 hm_cap; /* id -> id hash map for faithful decode */
  u64* str_off; unsigned char* heap; u64 n_str_ids; /* lexical forms of string ids (object_id_mapping.rs:48-59) */
};

/* REGEX patterns of the plan being executed (rdfgpu_plan_desc.regexes) */
static __thread const rdfgpu_regex* g_regexes; static __thread u32 g_n_regexes;

orc_store* orc_store_new(u32 batch_size) {
  orc_store* s = (orc_store*)calloc(1, sizeof *s);
  s->batch_size = batch_size ? batch_size : 8192;
  return s;
}
static void perm_free(perm_index* p) { for (int k = 0; k < 4; k++) { free(p->col[k]); p->col[k] = NULL; } p->n = 0; }
void orc_store_clear(orc_store* s) { for (int i = 0; i < 3; i++) perm_free(&s->idx[i]); }
void orc_store_free(orc_store* s) {
  if (!s) return;
  orc_store_clear(s); free(s->tv); free(s->dec); free(s->hm_keys); free(s->hm_vals); free(s);
}
u64 orc_store_len(const orc_store* s) { return s->idx[0].n; }
void orc_store_set_faithful_decode(orc_store* s, int on) { s->faithful_decode = on; }

static quad* make_sorted(const u32* g, const u32* su, const u32* p, const u32* o, u64 n, int comp, u64* n_out) {
  const u32* src[4] = {g, su, p, o};
  quad* q = (quad*)malloc((n ? n : 1) * sizeof(quad));
  for (u64 i = 0; i < n; i++) for (int k = 0; k < 4; k++) q[i].c[k] = src[PERM[comp][k]][i];
  qsort(q, n, sizeof(quad), quad_cmp);
  u64 m = 0;
  for (u64 i = 0; i < n; i++) if (i == 0 || quad_cmp(&q[i], &q[m - 1]) != 0) q[m++] = q[i];
  *n_out = m;
  return q;
}

/* IndexPermutations::insert permutations.rs:102-118 + MemIndexData::insert quad_index_data.rs:287-332
   (bulk form: sorted set merge; duplicates ignored). */
u64 orc_store_extend(orc_store* s, const u32* g, const u32* su, const u32* p, const u32* o, u64 n) {
  u64 inserted = 0;
  for (int comp = 0; comp < 3; comp++) {
    u64 m; quad* q = make_sorted(g, su, p, o, n, comp, &m);
    perm_index* ix = &s->idx[comp];
    u64 cap = ix->n + m;
    u32* out[4]; for (int k = 0; k < 4; k++) out[k] = (u32*)malloc((cap ? cap : 1) * sizeof(u32));
    u64 i = 0, j = 0, w = 0;
    while (i < ix->n || j < m) {
      int take_old;
      if (i >= ix->n) take_old = 0; else if (j >= m) take_old = 1; else {
        quad a; for (int k = 0; k < 4; k++) a.c[k] = ix->col[k][i];
        int c = quad_cmp(&a, &q[j]);
        if (c == 0) { j++; continue; } /* already contained: skip (quad_index_data.rs:301-304) */
        take_old = c < 0;
      }
      if (take_old) { for (int k = 0; k < 4; k++) out[k][w] = ix->col[k][i]; i++; }
      else { for (int k = 0; k < 4; k++) out[k][w] = q[j].c[k]; j++; }
      w++;
    }
    inserted = w - ix->n;
    for (int k = 0; k < 4; k++) { free(ix->col[k]); ix->col[k] = out[k]; }
    ix->n = w;
    free(q);
  }
  return inserted;
}

/* IndexPermutations::remove permutations.rs:120-128 + MemIndexData::remove quad_index_data.rs:338-376 */
u64 orc_store_remove(orc_store* s, const u32* g, const u32* su, const u32* p, const u32* o, u64 n) {
  u64 removed = 0;
  for (int comp = 0; comp < 3; comp++) {
    u64 m; quad* q = make_sorted(g, su, p, o, n, comp, &m);
    perm_index* ix = &s->idx[comp];
    u64 i = 0, j = 0, w = 0;
    while (i < ix->n) {
      quad a; for (int k = 0; k < 4; k++) a.c[k] = ix->col[k][i];
      while (j < m && quad_cmp(&q[j], &a) < 0) j++;
      if (j < m && quad_cmp(&q[j], &a) == 0) { i++; continue; }
      for (int k = 0; k < 4; k++) ix->col[k][w] = ix->col[k][i];
      w++; i++;
    }
    removed = ix->n - w;
    ix->n = w;
    free(q);
  }
  return removed;
}

int orc_store_adopt_sorted(orc_store* s, u32 comp, const u32* c0, const u32* c1, const u32* c2, const u32* c3, u64 n) {
  if (comp >= 3) FAIL("bad components");
  const u32* src[4] = {c0, c1, c2, c3};
  perm_index* ix = &s->idx[comp];
  perm_free(ix);
  for (int k = 0; k < 4; k++) { ix->col[k] = (u32*)malloc((n ? n : 1) * sizeof(u32)); memcpy(ix->col[k], src[k], n * sizeof(u32)); }
  ix->n = n;
  return 0;
}

int orc_store_read_index(const orc_store* s, u32 comp, u32* c0, u32* c1, u32* c2, u32* c3, u64 cap, u64* n) {
  if (comp >= 3) FAIL("bad components");
  const perm_index* ix = &s->idx[comp];
  u32* dst[4] = {c0, c1, c2, c3};
  u64 m = ix->n < cap ? ix->n : cap;
  for (int k = 0; k < 4; k++) if (dst[k]) memcpy(dst[k], ix->col[k], m * sizeof(u32));
  if (n) *n = ix->n;
  return 0;
}

static inline u64 mix64(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

void orc_store_set_typed_values(orc_store* s, const rdfgpu_typed_value* values, u64 n_ids, const int64_t* decimals, u64 n_dec) {
  free(s->tv); free(s->dec); free(s->hm_keys); free(s->hm_vals);
  s->tv = (rdfgpu_typed_value*)malloc((n_ids ? n_ids : 1) * sizeof *s->tv);
  memcpy(s->tv, values, n_ids * sizeof *s->tv); s->n_ids = n_ids;
  s->dec = (i128*)malloc((n_dec ? n_dec : 1) * sizeof(i128)); s->n_dec = n_dec;
  for (u64 i = 0; i < n_dec; i++) s->dec[i] = ((i128)decimals[2 * i + 1] << 64) | (i128)(u64)decimals[2 * i];
  /* id -> slot hash map mirroring the DashMap probe of object_id_mapping.rs:376-399 */
  u64 cap = 16; while (cap < 2 * n_ids) cap <<= 1;
  s->hm_cap = cap; s->hm_keys = (u32*)malloc(cap * sizeof(u32)); s->hm_vals = (u32*)malloc(cap * sizeof(u32));
  memset(s->hm_keys, 0xff, cap * sizeof(u32));
  for (u64 id = 1; id < n_ids; id++) {
    u64 h = mix64(id) & (cap - 1);
    while (s->hm_keys[h] != 0xffffffffu) h = (h + 1) & (cap - 1);
    s->hm_keys[h] = (u32)id; s->hm_vals[h] = (u32)id;
  }
}

void orc_store_set_strings(orc_store* s, const u64* offsets, u64 n_ids, const unsigned char* heap, u64 heap_bytes) {
  free(s->str_off); free(s->heap);
  s->str_off = (u64*)malloc((n_ids + 1) * sizeof(u64));
  memcpy(s->str_off, offsets, (n_ids + 1) * sizeof(u64));
  s->heap = (unsigned char*)malloc(heap_bytes ? heap_bytes : 1);
  memcpy(s->heap, heap, heap_bytes);
  s->n_str_ids = n_ids;
}

/* ------------------------------------------------------------------------------------ */
/* Predicates and instructions                                                           */
/* ------------------------------------------------------------------------------------ */

/* MemIndexPruningPredicate from a scan predicate (scan_instructions.rs:399-423):
   returns 1 and (from,to) if the predicate can prune. */
static int pruning_predicate(const rdfgpu_scan_instruction* in, const u32* pool, u32* from, u32* to) {
  if (in->pred == RDFGPU_PRED_IN) {
    if (in->b == 0) return 0;
    *from = pool[in->a]; *to = pool[in->a + in->b - 1]; /* sorted set: first / last */
    return 1;
  }
  if (in->pred == RDFGPU_PRED_BETWEEN) { *from = in->a; *to = in->b; return 1; }
  return 0;
}

/* MemQuadIndex::compute_scan_score quad_index.rs:100-130 */
static u64 scan_score_pool(const rdfgpu_scan_instruction instr[4], const u32* pool) {
  u64 score = 0;
  for (int i = 0; i < 4; i++) {
    u32 from, to;
    int is_between;
    if (instr[i].pred == RDFGPU_PRED_IN) {
      if (instr[i].b == 0) break;
      if (instr[i].b == 1) { from = to = 0; is_between = 0; }           /* EqualTo(id) */
      else { from = pool ? pool[instr[i].a] : 0; to = pool ? pool[instr[i].a + instr[i].b - 1] : 1; is_between = 1; }
    } else if (instr[i].pred == RDFGPU_PRED_BETWEEN) {
      from = instr[i].a; to = instr[i].b;
      is_between = from != to;  /* Between(x,x) is turned into EqualTo (scan_instructions.rs:413-419) */
    } else break;
    u64 potent = (u64)(4 - i) * 2;
    u64 reward = is_between ? (from == to ? 2 : 1) : 2;
    score += reward << potent;
    if (is_between) break; /* inner levels cannot be pruned below a range */
  }
  return score;
}
u64 orc_scan_score(const rdfgpu_scan_instruction instr[4]) { return scan_score_pool(instr, NULL); }

/* MemIndexScanInstructions::reorder scan_instructions.rs:137-152 */
static void reorder(const rdfgpu_scan_instruction gspo[4], int comp, rdfgpu_scan_instruction out[4]) {
  for (int k = 0; k < 4; k++) out[k] = gspo[PERM[comp][k]];
}

/* IndexPermutations::choose_index permutations.rs:81-96: max score, ties -> first listed */
static u32 choose_index_pool(const rdfgpu_scan_instruction gspo[4], u32 available, const u32* pool) {
  int best = -1; u64 best_score = 0;
  for (int comp = 0; comp < 3; comp++) {
    if (!(available & (1u << comp))) continue;
    rdfgpu_scan_instruction r[4]; reorder(gspo, comp, r);
    u64 sc = scan_score_pool(r, pool);
    if (best < 0 || sc > best_score) { best = comp; best_score = sc; }
  }
  return (u32)best;
}
u32 orc_choose_index(const rdfgpu_scan_instruction gspo[4], u32 available) { return choose_index_pool(gspo, available, NULL); }

/* MemIndexScanPredicate::try_and_with scan_instructions.rs:170-210 */
int orc_predicate_and(const rdfgpu_predicate* a, const rdfgpu_predicate* b, rdfgpu_predicate* out, u32* out_ids) {
  memset(out, 0, sizeof *out);
  if (a->pred == RDFGPU_PRED_FALSE || b->pred == RDFGPU_PRED_FALSE) { out->pred = RDFGPU_PRED_FALSE; return 1; }
  if (a->pred == RDFGPU_PRED_IN && b->pred == RDFGPU_PRED_IN) {
    u32 i = 0, j = 0, w = 0;
    while (i < a->n_ids && j < b->n_ids) {
      if (a->ids[i] == b->ids[j]) { out_ids[w++] = a->ids[i]; i++; j++; }
      else if (a->ids[i] < b->ids[j]) i++; else j++;
    }
    if (!w) { out->pred = RDFGPU_PRED_FALSE; return 1; }
    out->pred = RDFGPU_PRED_IN; out->ids = out_ids; out->n_ids = w; return 1;
  }
  if ((a->pred == RDFGPU_PRED_IN && b->pred == RDFGPU_PRED_BETWEEN) || (a->pred == RDFGPU_PRED_BETWEEN && b->pred == RDFGPU_PRED_IN)) {
    const rdfgpu_predicate* in = a->pred == RDFGPU_PRED_IN ? a : b;
    const rdfgpu_predicate* bt = a->pred == RDFGPU_PRED_IN ? b : a;
    u32 w = 0;
    for (u32 i = 0; i < in->n_ids; i++) if (in->ids[i] >= bt->from && in->ids[i] <= bt->to) out_ids[w++] = in->ids[i];
    if (!w) { out->pred = RDFGPU_PRED_FALSE; return 1; }
    out->pred = RDFGPU_PRED_IN; out->ids = out_ids; out->n_ids = w; return 1;
  }
  if (a->pred == RDFGPU_PRED_BETWEEN && b->pred == RDFGPU_PRED_BETWEEN) {
    u32 from = a->from > b->from ? a->from : b->from;
    u32 to = a->to < b->to ? a->to : b->to;
    if (from > to) { out->pred = RDFGPU_PRED_FALSE; return 1; }
    out->pred = RDFGPU_PRED_BETWEEN; out->from = from; out->to = to; return 1;
  }
  return 0; /* not combinable (EqualTo) */
}

/* MemStoragePredicateExpr::to_scan_predicate predicate_pushdown.rs:120-157 */
int orc_pushdown_to_scan_predicate(u32 op, u32 value, rdfgpu_predicate* out) {
  memset(out, 0, sizeof *out);
  switch (op) {
    case RDFGPU_OP_GT:
      if (value == 0xffffffffu) { out->pred = RDFGPU_PRED_FALSE; return 1; }
      out->pred = RDFGPU_PRED_BETWEEN; out->from = value + 1; out->to = 0xffffffffu; return 1;
    case RDFGPU_OP_GTEQ: out->pred = RDFGPU_PRED_BETWEEN; out->from = value; out->to = 0xffffffffu; return 1;
    case RDFGPU_OP_LT:
      if (value == 0) { out->pred = RDFGPU_PRED_FALSE; return 1; }
      out->pred = RDFGPU_PRED_BETWEEN; out->from = 0; out->to = value - 1; return 1;
    case RDFGPU_OP_LTEQ: out->pred = RDFGPU_PRED_BETWEEN; out->from = 0; out->to = value; return 1;
    case RDFGPU_OP_EQ: out->pred = RDFGPU_PRED_IN; out->from = value; out->to = value; out->n_ids = 1; return 1;
    default: FAIL("unsupported operator");
  }
}

/* ------------------------------------------------------------------------------------ */
/* MemColumnChunk::find_range_between quad_index_data.rs:600-650                          */
/* Null slots hold the value 0 (UInt32Array::from(Vec<Option<u32>>)), sorted first.       */
/* ------------------------------------------------------------------------------------ */
int orc_find_range_between(const u32* values, u64 n, u32 from, u32 to, u64* lo, u64* hi) {
  *lo = *hi = 0;
  if (to < values[0]) return ORC_FR_BEFORE;
  if (values[n - 1] < from) return ORC_FR_AFTER;
  if (from == 0 && to == 0) {
    u64 nulls = 0; while (nulls < n && values[nulls] == 0) nulls++;
    if (nulls == 0) return ORC_FR_BEFORE;
    *lo = 0; *hi = nulls; return ORC_FR_CONTAINED;
  }
  u64 pos = 0; while (pos < n && values[pos] < from) pos++;           /* linear `position` :627 */
  if (values[pos] > to) { *lo = pos; return ORC_FR_NOT_CONTAINED; }
  u64 cnt = 0; while (pos + cnt < n && values[pos + cnt] <= to) cnt++; /* linear `take_while` :641 */
  *lo = pos; *hi = pos + cnt;
  return ORC_FR_CONTAINED;
}

/* ------------------------------------------------------------------------------------ */
/* MemIndexData::prune_relevant_row_groups quad_index_data.rs:155-284                     */
/* Row groups of a bulk-loaded index are consecutive chunks of batch_size rows (:324-329). */
/* ------------------------------------------------------------------------------------ */
typedef struct { u64 start, end; } rg_slice;

static int prune_impl(const orc_store* s, int comp, const rdfgpu_scan_instruction instr[4], const u32* pool,
                      rg_slice** out_slices, u32* out_n, u32* dropped_mask) {
  const perm_index* ix = &s->idx[comp];
  u64 B = s->batch_size;
  u32 ng = (u32)((ix->n + B - 1) / B);
  rg_slice* cur = (rg_slice*)malloc((ng ? ng : 1) * sizeof *cur);   /* `self.row_groups.clone()` :160 */
  for (u32 i = 0; i < ng; i++) { cur[i].start = (u64)i * B; cur[i].end = cur[i].start + B < ix->n ? cur[i].start + B : ix->n; }
  u32 ncur = ng;
  *dropped_mask = 0;

  for (int column = 0; column < 4; column++) {
    u32 from, to;
    if (!pruning_predicate(&instr[column], pool, &from, &to)) break;
    /* :403-411: In with one id => EqualTo; with several => Between(first,last) */
    const u32* v = ix->col[column];
    u32 first = 0; int first_res = ORC_FR_AFTER; u64 flo = 0, fhi = 0; int found = 0;
    for (u32 g = 0; g < ncur; g++) {
      int r = orc_find_range_between(v + cur[g].start, cur[g].end - cur[g].start, from, to, &flo, &fhi);
      if (r != ORC_FR_AFTER) { first = g; first_res = r; found = 1; break; }
    }
    if (!found || first_res != ORC_FR_CONTAINED) { free(cur); *out_slices = NULL; *out_n = 0; return 0; }
    rg_slice* nxt = (rg_slice*)malloc(ncur * sizeof *nxt); u32 nn = 0;
    nxt[nn].start = cur[first].start + flo; nxt[nn].end = cur[first].start + fhi; nn++;
    if (fhi == cur[first].end - cur[first].start) {
      for (u32 g = first + 1; g < ncur; g++) {
        u64 lo, hi;
        int r = orc_find_range_between(v + cur[g].start, cur[g].end - cur[g].start, from, to, &lo, &hi);
        if (r == ORC_FR_BEFORE) break;
        if (r != ORC_FR_CONTAINED || lo != 0) { free(cur); free(nxt); FAIL("column is not sorted"); }
        if (hi < cur[g].end - cur[g].start) { nxt[nn].start = cur[g].start; nxt[nn].end = cur[g].start + hi; nn++; break; }
        nxt[nn++] = cur[g];
      }
    }
    free(cur); cur = nxt; ncur = nn;
    if (from != to) break; /* :240 */
  }

  /* :245-265 which predicates are now redundant */
  for (int i = 0; i < 4; i++) {
    if (instr[i].pred == RDFGPU_PRED_IN) { if (instr[i].b == 1) *dropped_mask |= 1u << i; else break; }
    else if (instr[i].pred == RDFGPU_PRED_BETWEEN) { *dropped_mask |= 1u << i; if (instr[i].a != instr[i].b) break; }
    else break;
  }
  *out_slices = cur; *out_n = ncur;
  return 0;
}

int orc_prune(const orc_store* s, u32 comp, const rdfgpu_scan_instruction instr[4], const u32* pool,
              u64* starts, u64* ends, u32 cap, u32* dropped_mask) {
  rg_slice* sl; u32 n;
  if (comp >= 3) FAIL("bad components");
  if (s->idx[comp].n == 0) { *dropped_mask = 0; return 0; }
  if (prune_impl(s, (int)comp, instr, pool, &sl, &n, dropped_mask)) return -1;
  for (u32 i = 0; i < n && i < cap; i++) { starts[i] = sl[i].start; ends[i] = sl[i].end; }
  free(sl);
  return (int)n;
}

/* ------------------------------------------------------------------------------------ */
/* MemQuadIndexScanIterator::next scan.rs:104-212; compute_selection_vector :264-280;     */
/* apply_predicate :292-340; reorder_result :394-409                                      */
/* ------------------------------------------------------------------------------------ */

/* MemIndexScanInstructions::new scan_instructions.rs:19-46: a variable bound twice turns the
   second occurrence into Traverse(EqualTo(var)). */
static void normalize(const rdfgpu_scan_instruction in[4], rdfgpu_scan_instruction out[4]) {
  for (int i = 0; i < 4; i++) {
    out[i] = in[i];
    if (in[i].kind == RDFGPU_SCAN) {
      for (int j = 0; j < i; j++) if (out[j].kind == RDFGPU_SCAN && out[j].var == in[i].var) {
        out[i].kind = RDFGPU_TRAVERSE; out[i].pred = RDFGPU_PRED_EQUAL_TO; out[i].a = in[i].var; out[i].b = 0;
        break;
      }
    }
  }
}

void orc_scan_result_free(orc_scan_result* r) {
  for (int k = 0; k < 4; k++) { free(r->cols[k]); r->cols[k] = NULL; }
  free(r->batch_rows); r->batch_rows = NULL;
}

int orc_scan(const orc_store* s, const rdfgpu_scan_instruction gspo_in[4], const u32* pool, int force_index, orc_scan_result* out) {
  memset(out, 0, sizeof *out);
  rdfgpu_scan_instruction gspo[4]; normalize(gspo_in, gspo);
  /* output schema: bound variables in G,S,P,O order (patterns/mod.rs:68-107) */
  for (int i = 0; i < 4; i++) if (gspo[i].kind == RDFGPU_SCAN) out->vars[out->n_cols++] = gspo[i].var;

  int comp = force_index >= 0 ? force_index : (int)choose_index_pool(gspo, 7u, pool);
  out->chosen_index = (u32)comp;
  rdfgpu_scan_instruction instr[4]; reorder(gspo, comp, instr);
  const perm_index* ix = &s->idx[comp];
  if (ix->n == 0) return 0;

  rg_slice* sl; u32 nsl; u32 dropped;
  if (prune_impl(s, comp, instr, pool, &sl, &nsl, &dropped)) return -1;
  if (nsl == 0) { free(sl); return 0; }
  /* `new_instructions`: predicates proven by pruning are removed (:267-278) */
  for (int i = 0; i < 4; i++) if (dropped & (1u << i)) { instr[i].pred = RDFGPU_PRED_NONE; }

  u64 cap = 0; for (u32 g = 0; g < nsl; g++) cap += sl[g].end - sl[g].start;
  u32* outcol[4] = {0, 0, 0, 0}; int level_of_col[4]; int ncol = 0;
  /* map output column -> index level that scans that variable */
  for (u32 c = 0; c < out->n_cols; c++) {
    for (int k = 0; k < 4; k++) if (instr[k].kind == RDFGPU_SCAN && instr[k].var == out->vars[c]) level_of_col[c] = k;
    outcol[c] = (u32*)malloc((cap ? cap : 1) * sizeof(u32)); ncol++;
  }
  out->batch_rows = (u32*)malloc((nsl ? nsl : 1) * sizeof(u32));
  uint8_t* mask = (uint8_t*)malloc(s->batch_size + 1);
  u64 w = 0;
  int any_pred = 0; for (int k = 0; k < 4; k++) if (instr[k].pred != RDFGPU_PRED_NONE) any_pred = 1;

  for (u32 g = 0; g < nsl; g++) { /* `data.remove(0)` per batch :138 */
    u64 st = sl[g].start, len = sl[g].end - sl[g].start;
    int have_mask = 0;
    if (any_pred) {
      for (int k = 0; k < 4; k++) {
        const rdfgpu_scan_instruction* in = &instr[k];
        if (in->pred == RDFGPU_PRED_NONE) continue;
        const u32* d = ix->col[k] + st;
        int this_mask = 1;
        /* one boolean pass per predicate, AND-combined (:268-279) */
        if (in->pred == RDFGPU_PRED_IN) {
          if (in->b == 0) this_mask = 0; /* empty set: reduce() of nothing => None */
          else for (u64 i = 0; i < len; i++) { uint8_t m = 0; for (u32 q = 0; q < in->b; q++) m |= d[i] == pool[in->a + q]; mask[i] = have_mask ? (mask[i] & m) : m; }
        } else if (in->pred == RDFGPU_PRED_BETWEEN) {
          for (u64 i = 0; i < len; i++) { uint8_t m = d[i] >= in->a && d[i] <= in->b; mask[i] = have_mask ? (mask[i] & m) : m; }
        } else if (in->pred == RDFGPU_PRED_EQUAL_TO) {
          int other = -1; for (int j = 0; j < 4; j++) if (instr[j].kind == RDFGPU_SCAN && instr[j].var == in->a) { other = j; break; }
          if (other < 0) this_mask = 0; /* `position(..)?` => no mask (:310-313) */
          else { const u32* e = ix->col[other] + st; for (u64 i = 0; i < len; i++) { uint8_t m = e[i] == d[i]; mask[i] = have_mask ? (mask[i] & m) : m; } }
        } else if (in->pred == RDFGPU_PRED_FALSE) {
          for (u64 i = 0; i < len; i++) mask[i] = 0;
        }
        if (this_mask) have_mask = 1;
      }
    }
    if (!have_mask) { /* hand the slices through untouched (:146-170) */
      for (int c = 0; c < ncol; c++) memcpy(outcol[c] + w, ix->col[level_of_col[c]] + st, len * sizeof(u32));
      w += len; out->batch_rows[out->n_batches++] = (u32)len;
    } else {
      u64 cnt = 0;
      for (u64 i = 0; i < len; i++) if (mask[i]) { for (int c = 0; c < ncol; c++) outcol[c][w + cnt] = ix->col[level_of_col[c]][st + i]; cnt++; }
      if (cnt) { w += cnt; out->batch_rows[out->n_batches++] = (u32)cnt; } /* never emit empty batches (:195-198) */
    }
  }
  free(mask); free(sl);
  for (int c = 0; c < ncol; c++) out->cols[c] = outcol[c];
  out->n_rows = w;
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Typed values and the FILTER expression semantics                                       */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  uint8_t kind; /* 0 ID, 1 TV, 2 BOOL */
  uint8_t tag, flags, b; /* b: BOOL 0/1/2(null) */
  u32 aux; u32 id;
  int64_t lo; i128 dec;
  /* a COMPUTED string (STR / SUBSTR / UCASE / LCASE / a constant with bytes): materialised in the per-row arena; it has no
     rank in the dictionary's order, so comparisons with it go by bytes */
  const unsigned char* sp; u32 sl; uint8_t computed;
} val;
static __thread unsigned char g_arena[1 << 16]; static __thread size_t g_arena_used;
static unsigned char* arena_take(size_t n) {
  if (g_arena_used + n > sizeof g_arena) return NULL;
  unsigned char* p = g_arena + g_arena_used; g_arena_used += n; return p;
}

static const i128 DEC_POW = (i128)1000000000000000000LL; /* decimal.rs:9-11 */

static int is_timestamp(uint8_t tag) { return tag == RDFGPU_TV_DATE_TIME || tag == RDFGPU_TV_TIME || tag == RDFGPU_TV_DATE; }
static val tv_null(void) { val v; memset(&v, 0, sizeof v); v.kind = 1; v.tag = RDFGPU_TV_NULL; return v; }
static val tv_bool(int b) { val v = tv_null(); v.tag = RDFGPU_TV_BOOLEAN; v.lo = b ? 1 : 0; return v; }

/* ENC_TV: with_typed_value_encoding.rs:71-79 -> decode_array_to_typed_value object_id_mapping.rs:376-399 */
static val enc_tv(const orc_store* s, u32 id) {
  val v = tv_null();
  if (id == 0 || id >= s->n_ids) return v;
  u32 slot = id;
  if (s->faithful_decode) { /* one hash-map probe per row, like the DashMap lookup */
    u64 h = mix64(id) & (s->hm_cap - 1);
    while (s->hm_keys[h] != id) { if (s->hm_keys[h] == 0xffffffffu) return v; h = (h + 1) & (s->hm_cap - 1); }
    slot = s->hm_vals[h];
  }
  const rdfgpu_typed_value* t = &s->tv[slot];
  v.tag = t->tag; v.flags = t->flags; v.aux = t->aux; v.lo = t->lo; v.id = id;
  if (t->tag == RDFGPU_TV_DECIMAL || is_timestamp(t->tag)) { if ((u64)t->lo >= s->n_dec) return tv_null(); v.dec = s->dec[t->lo]; }
  return v;
}

static inline float f32_of(int64_t lo) { u32 b = (u32)lo; float f; memcpy(&f, &b, 4); return f; }
static inline double f64_of(int64_t lo) { double d; memcpy(&d, &lo, 8); return d; }
static inline int64_t bits_f32(float f) { u32 b; memcpy(&b, &f, 4); return (int64_t)b; }
static inline int64_t bits_f64(double d) { int64_t b; memcpy(&b, &d, 8); return b; }

/* From<Decimal> for Double decimal.rs:445-462 */
static double dec_to_f64(i128 value) {
  i128 shift = DEC_POW;
  if (value != 0) while (shift != 1 && value % 10 == 0) { value /= 10; shift /= 10; }
  return (double)value / (double)shift;
}

enum { NK_INT, NK_INTEGER, NK_FLOAT, NK_DOUBLE, NK_DECIMAL, NK_NONE };
static int num_kind(uint8_t tag) {
  switch (tag) {
    case RDFGPU_TV_INT: return NK_INT; case RDFGPU_TV_INTEGER: return NK_INTEGER; case RDFGPU_TV_FLOAT: return NK_FLOAT;
    case RDFGPU_TV_DOUBLE: return NK_DOUBLE; case RDFGPU_TV_DECIMAL: return NK_DECIMAL; default: return NK_NONE;
  }
}
/* NumericPair::with_casts_from numeric.rs:127-201 */
static int pair_kind(int a, int b) {
  if (a == NK_DOUBLE || b == NK_DOUBLE) return NK_DOUBLE;
  if (a == NK_FLOAT || b == NK_FLOAT) return NK_FLOAT;
  if (a == NK_DECIMAL || b == NK_DECIMAL) return NK_DECIMAL;
  if (a == NK_INTEGER || b == NK_INTEGER) return NK_INTEGER;
  return NK_INT;
}
static double to_f64(const val* v, int k) {
  switch (k) {
    case NK_INT: case NK_INTEGER: return (double)v->lo;                 /* double.rs:189-201 */
    case NK_FLOAT: return (double)f32_of(v->lo);
    case NK_DOUBLE: return f64_of(v->lo);
    default: return dec_to_f64(v->dec);
  }
}
static float to_f32(const val* v, int k) {
  switch (k) {
    case NK_INT: return (float)(int32_t)v->lo;                          /* float.rs:156-161 */
    case NK_INTEGER: return (float)v->lo;                               /* float.rs:164-169 */
    case NK_FLOAT: return f32_of(v->lo);
    case NK_DECIMAL: return (float)dec_to_f64(v->dec);                  /* decimal.rs:437-443 */
    default: return (float)f64_of(v->lo);
  }
}
static i128 to_dec(const val* v, int k) { return k == NK_DECIMAL ? v->dec : (i128)v->lo * DEC_POW; }

#define ORD_NONE 2
/* PartialOrd for TypedValueRef typed_value.rs:162-261 ; returns -1/0/1 or ORD_NONE */
static int tv_partial_cmp(const val* a, const val* b) {
  if (a->tag == RDFGPU_TV_NULL || b->tag == RDFGPU_TV_NULL) return ORD_NONE;
  if (a->tag == RDFGPU_TV_BLANK_NODE) { if (b->tag == RDFGPU_TV_BLANK_NODE) return a->lo < b->lo ? -1 : a->lo > b->lo; return -1; }
  if (a->tag == RDFGPU_TV_NAMED_NODE) {
    if (b->tag == RDFGPU_TV_BLANK_NODE) return 1;
    if (b->tag == RDFGPU_TV_NAMED_NODE) return a->lo < b->lo ? -1 : a->lo > b->lo;
    return -1;
  }
  if (b->tag == RDFGPU_TV_NAMED_NODE || b->tag == RDFGPU_TV_BLANK_NODE) return 1;
  /* partial_cmp_literals :184-261 */
  if (a->tag == RDFGPU_TV_STRING) {
    if (b->tag != RDFGPU_TV_STRING) return ORD_NONE;
    if (a->aux != b->aux) return ORD_NONE; /* simple vs lang, or different languages (language_string.rs:44-52) */
    return a->lo < b->lo ? -1 : a->lo > b->lo;
  }
  if (a->tag == RDFGPU_TV_BOOLEAN) { if (b->tag != RDFGPU_TV_BOOLEAN) return ORD_NONE; return (a->lo != 0) - (b->lo != 0); }
  int ka = num_kind(a->tag), kb = num_kind(b->tag);
  if (ka != NK_NONE) {
    if (kb == NK_NONE) return ORD_NONE;
    switch (pair_kind(ka, kb)) { /* numeric.rs:90-100 */
      case NK_INT: case NK_INTEGER: return a->lo < b->lo ? -1 : a->lo > b->lo;
      case NK_FLOAT: { float x = to_f32(a, ka), y = to_f32(b, kb); if (x < y) return -1; if (x > y) return 1; if (x == y) return 0; return ORD_NONE; }
      case NK_DOUBLE: { double x = to_f64(a, ka), y = to_f64(b, kb); if (x < y) return -1; if (x > y) return 1; if (x == y) return 0; return ORD_NONE; }
      default: { i128 x = to_dec(a, ka), y = to_dec(b, kb); return x < y ? -1 : x > y; }
    }
  }
  if (a->tag == RDFGPU_TV_OTHER) { /* :253-258 */
    if (b->tag == RDFGPU_TV_OTHER && a->aux == b->aux && a->lo == b->lo) return 0;
    return ORD_NONE;
  }
  if (is_timestamp(a->tag)) { /* PartialOrd for Timestamp, lib/model/src/xsd/date_time.rs:1617-1654 */
    if (b->tag != a->tag) return ORD_NONE;                       /* typed_value.rs:222-242: same kind only */
    int ta = a->aux & 1, tb = b->aux & 1;                         /* timezone_offset.is_some() */
    if (ta == tb) return a->dec < b->dec ? -1 : a->dec > b->dec;
    const i128 shift = (i128)(14 * 3600) * DEC_POW;
    i128 other = ta ? b->dec : a->dec, plus_v, minus_v;
    if (__builtin_add_overflow(other, shift, &plus_v)) return ORD_NONE;   /* checked_add(..).ok()? */
    if (__builtin_sub_overflow(other, shift, &minus_v)) return ORD_NONE;
    int plus, minus;
    if (ta) { plus = a->dec < plus_v ? -1 : a->dec > plus_v; minus = a->dec < minus_v ? -1 : a->dec > minus_v; }
    else { plus = plus_v < b->dec ? -1 : plus_v > b->dec; minus = minus_v < b->dec ? -1 : minus_v > b->dec; }
    return plus == minus ? plus : ORD_NONE;
  }
  /* durations: opaque in this build => incomparable (documented gap; duration.rs:271-310 is calendar arithmetic) */
  return ORD_NONE;
}

/* ADD add.rs:40-86, SUB sub.rs (same shape) */
static val tv_arith(const val* a, const val* b, int sub) {
  int ka = num_kind(a->tag), kb = num_kind(b->tag);
  if (ka == NK_NONE || kb == NK_NONE) return tv_null();
  val r = tv_null();
  switch (pair_kind(ka, kb)) {
    case NK_INT: { int32_t x = (int32_t)a->lo, y = (int32_t)b->lo, z;
      if (sub ? __builtin_sub_overflow(x, y, &z) : __builtin_add_overflow(x, y, &z)) return tv_null();
      r.tag = RDFGPU_TV_INT; r.lo = z; return r; }
    case NK_INTEGER: { int64_t z;
      if (sub ? __builtin_sub_overflow(a->lo, b->lo, &z) : __builtin_add_overflow(a->lo, b->lo, &z)) return tv_null(); /* integer.rs:39-58 */
      r.tag = RDFGPU_TV_INTEGER; r.lo = z; return r; }
    case NK_FLOAT: { float x = to_f32(a, ka), y = to_f32(b, kb); float z = sub ? x - y : x + y; r.tag = RDFGPU_TV_FLOAT; r.lo = bits_f32(z); return r; }
    case NK_DOUBLE: { double x = to_f64(a, ka), y = to_f64(b, kb); double z = sub ? x - y : x + y; r.tag = RDFGPU_TV_DOUBLE; r.lo = bits_f64(z); return r; }
    default: { i128 x = to_dec(a, ka), y = to_dec(b, kb), z;
      if (sub ? __builtin_sub_overflow(x, y, &z) : __builtin_add_overflow(x, y, &z)) return tv_null(); /* decimal.rs:66-88 */
      r.tag = RDFGPU_TV_DECIMAL; r.dec = z; return r; }
  }
}

/* EBV effective_boolean_value.rs:99-119 ; returns 0/1/2(null) */
static uint8_t tv_ebv(const val* v) {
  switch (v->tag) {
    case RDFGPU_TV_BOOLEAN: case RDFGPU_TV_INT: case RDFGPU_TV_INTEGER: return v->lo != 0;
    case RDFGPU_TV_FLOAT: return f32_of(v->lo) != 0.0f;
    case RDFGPU_TV_DOUBLE: return f64_of(v->lo) != 0.0;
    case RDFGPU_TV_DECIMAL: return v->dec != 0;
    case RDFGPU_TV_STRING: if (v->aux != 0) return 2; return !(v->flags & RDFGPU_TVF_EMPTY_STRING);
    default: return 2;
  }
}

/* the bytes of a string value: a computed string's own, a dictionary string's lexical form from the heap */
static int str_bytes(const orc_store* s, const val* v, const unsigned char** p, size_t* n) {
  if (v->tag != RDFGPU_TV_STRING) return 0;
  if (v->computed) { *p = v->sp; *n = v->sl; return 1; }
  if (v->id == 0 || v->id >= s->n_str_ids) return 0;
  *p = s->heap + s->str_off[v->id]; *n = (size_t)(s->str_off[v->id + 1] - s->str_off[v->id]);
  return 1;
}
static val tv_computed_string(const unsigned char* p, size_t n, u32 lang) {
  val v = tv_null();
  v.tag = RDFGPU_TV_STRING; v.computed = 1; v.sp = p; v.sl = (u32)n; v.aux = lang; v.flags = n == 0 ? RDFGPU_TVF_EMPTY_STRING : 0;
  return v;
}
/* length in bytes of the UTF-8 sequence a lead byte starts (the heap holds valid UTF-8) */
static size_t utf8_len(unsigned char b) { return b < 0x80 ? 1 : (b >> 5) == 6 ? 2 : (b >> 4) == 14 ? 3 : 4; }

typedef struct { const u32* const* lc; u32 nl; u64 li; const u32* const* rc; u32 nr; u64 ri; } rowctx;
static inline u32 row_col(const rowctx* r, u32 c) { return c < r->nl ? r->lc[c][r->li] : r->rc[c - r->nl][r->ri]; }

#define STK 32
static int eval_prog(const orc_store* s, const rdfgpu_expr_node* p, u32 n, const rowctx* row, val* result) {
  val st[STK]; int sp = 0;
  g_arena_used = 0;
  for (u32 i = 0; i < n; i++) {
    const rdfgpu_expr_node* e = &p[i];
    val v; memset(&v, 0, sizeof v);
    switch (e->op) {
      case RDFGPU_EX_COLUMN: if (e->u >= row->nl + row->nr) FAIL("column %u out of range", e->u); v.kind = 0; v.id = row_col(row, e->u); break;
      case RDFGPU_EX_LIT_ID: v.kind = 0; v.id = e->u; break;
      case RDFGPU_EX_LIT_TV: v = tv_null(); v.tag = e->tag; v.flags = e->flags; v.aux = e->u; v.lo = e->lo;
        if (e->tag == RDFGPU_TV_DECIMAL || is_timestamp(e->tag)) { v.dec = ((i128)e->hi << 64) | (i128)(u64)e->lo; }
        break;
      case RDFGPU_EX_LIT_BOOL: v.kind = 2; v.b = (uint8_t)(e->u > 2 ? 2 : e->u); break;
      case RDFGPU_EX_ENC_TV: if (sp < 1 || st[sp - 1].kind != 0) FAIL("ENC_TV needs an id"); v = enc_tv(s, st[--sp].id); break;
      case RDFGPU_EX_GT: case RDFGPU_EX_LT: case RDFGPU_EX_GEQ: case RDFGPU_EX_LEQ: case RDFGPU_EX_EQ: case RDFGPU_EX_NEQ: {
        if (sp < 2 || st[sp - 1].kind != 1 || st[sp - 2].kind != 1) FAIL("comparison needs two typed values");
        val b = st[--sp], a = st[--sp];
        int o;
        if (a.tag == RDFGPU_TV_STRING && b.tag == RDFGPU_TV_STRING && (a.computed || b.computed)) {
          /* partial_cmp_literals typed_value.rs:184-196: simple with simple, language-tagged with the same language: the values, as `str` */
          const unsigned char *pa, *pb; size_t na, nb;
          if (a.aux != b.aux) o = ORD_NONE;
          else {
            if (!str_bytes(s, &a, &pa, &na) || !str_bytes(s, &b, &pb, &nb)) FAIL("a computed string is compared with a string that has no bytes here (pass the literal as RDFGPU_EX_LIT_STR)");
            int c = memcmp(pa, pb, na < nb ? na : nb);
            o = c < 0 ? -1 : c > 0 ? 1 : na < nb ? -1 : na > nb;
          }
        } else o = tv_partial_cmp(&a, &b);
        if (o == ORD_NONE) { v = tv_null(); break; } /* ThinError => null (greater_than.rs:52-60) */
        int r = e->op == RDFGPU_EX_GT ? o > 0 : e->op == RDFGPU_EX_LT ? o < 0 : e->op == RDFGPU_EX_GEQ ? o >= 0 :
                e->op == RDFGPU_EX_LEQ ? o <= 0 : e->op == RDFGPU_EX_EQ ? o == 0 : o != 0;
        v = tv_bool(r); break; }
      case RDFGPU_EX_ADD: case RDFGPU_EX_SUB: {
        if (sp < 2 || st[sp - 1].kind != 1 || st[sp - 2].kind != 1) FAIL("arithmetic needs two typed values");
        val b = st[--sp], a = st[--sp]; v = tv_arith(&a, &b, e->op == RDFGPU_EX_SUB); break; }
      case RDFGPU_EX_REGEX: {   /* regex.rs:47-141: simple / language strings are searched, anything else is an error */
        if (sp < 1 || st[sp - 1].kind != 1) FAIL("REGEX needs a typed value");
        if (e->u >= g_n_regexes) FAIL("REGEX pattern %u out of range", e->u);
        val a = st[--sp];
        v = tv_null();
        const unsigned char* subj; size_t subj_n;
        if (!str_bytes(s, &a, &subj, &subj_n)) break;
        const rdfgpu_regex* rx = &g_regexes[e->u];
        int m = orc_regex_is_match(rx->pattern, rx->pattern_len, rx->flags ? rx->flags : "", rx->flags ? rx->flags_len : 0, subj, subj_n);
        if (m == -2) FAIL("REGEX with \\d \\w \\s or \\b over a string with non-ASCII characters needs the regex crate's Unicode tables");
        if (m >= 0) v = tv_bool(m);
        break; }
      case RDFGPU_EX_REGEX_VAR: {
        /* regex.rs:59-76: the pattern is the row's second argument — a simple literal, compiled for this row */
        if (sp < 2 || st[sp - 1].kind != 1 || st[sp - 2].kind != 1) FAIL("REGEX needs two typed values");
        if (e->u >= g_n_regexes) FAIL("REGEX pattern table %u out of range", e->u);
        val pat = st[--sp]; val a = st[--sp];
        v = tv_null();
        if (pat.tag != RDFGPU_TV_STRING || pat.aux != 0 || pat.computed || pat.id == 0 || pat.id >= s->n_str_ids) break;
        const unsigned char* subj; size_t subj_n;
        if (!str_bytes(s, &a, &subj, &subj_n)) break;
        const rdfgpu_regex* rx = &g_regexes[e->u];      /* the flags are the plan's constant third argument */
        int m = orc_regex_is_match((const char*)(s->heap + s->str_off[pat.id]), (size_t)(s->str_off[pat.id + 1] - s->str_off[pat.id]),
                                   rx->flags ? rx->flags : "", rx->flags ? rx->flags_len : 0, subj, subj_n);
        if (m == -2) FAIL("REGEX with \\d \\w \\s or \\b over a string with non-ASCII characters needs the regex crate's Unicode tables");
        if (m >= 0) v = tv_bool(m);
        break; }
      case RDFGPU_EX_CONTAINS: case RDFGPU_EX_STRSTARTS: case RDFGPU_EX_STRENDS: {
        /* contains.rs / str_starts.rs / str_ends.rs: both arguments string literals, compatible per
           string_literal.rs:80-95 (the constant has no language, or the value's); then str::contains / starts_with / ends_with */
        if (sp < 1 || st[sp - 1].kind != 1) FAIL("string function needs a typed value");
        if (e->u >= g_n_regexes) FAIL("string constant %u out of range", e->u);
        val a = st[--sp];
        v = tv_null();
        const unsigned char* hay; size_t hl;
        if (!str_bytes(s, &a, &hay, &hl)) break;
        if (e->lo > 0 && (int64_t)a.aux != e->lo) break;
        const rdfgpu_regex* rx = &g_regexes[e->u];
        const size_t nl = rx->pattern_len;
        int r = 0;
        if (nl <= hl) {
          if (e->op == RDFGPU_EX_STRSTARTS) r = memcmp(hay, rx->pattern, nl) == 0;
          else if (e->op == RDFGPU_EX_STRENDS) r = memcmp(hay + hl - nl, rx->pattern, nl) == 0;
          else { for (size_t k = 0; k + nl <= hl && !r; k++) r = memcmp(hay + k, rx->pattern, nl) == 0; }
        }
        v = tv_bool(r);
        break; }
      case RDFGPU_EX_LANG_IN: {   /* LANGMATCHES(LANG(v), range): lang.rs:45-58 then the host-resolved verdict of lang_matches.rs:52-69 */
        if (sp < 1 || st[sp - 1].kind != 1) FAIL("LANGMATCHES(LANG()) needs a typed value");
        if (e->u >= g_n_regexes) FAIL("language table %u out of range", e->u);
        val a = st[--sp];
        v = tv_null();
        if (a.tag == RDFGPU_TV_NULL || a.tag == RDFGPU_TV_NAMED_NODE || a.tag == RDFGPU_TV_BLANK_NODE) break;   /* LANG of a non-literal: error */
        u32 lang = a.tag == RDFGPU_TV_STRING ? a.aux : 0;   /* every other literal has the empty tag */
        const rdfgpu_regex* rx = &g_regexes[e->u];
        if (lang >= rx->pattern_len) break;
        v = tv_bool(rx->pattern[lang] != 0);
        break; }
      case RDFGPU_EX_STR: {   /* str.rs:42 in the plain-term encoding (the encoding an object-id argument is converted to): the lexical form as written */
        if (sp < 1 || st[sp - 1].kind != 0) FAIL("STR needs an id");
        u32 id = st[--sp].id;
        v = tv_null();
        if (id == 0 || id >= s->n_str_ids) break;
        size_t len = (size_t)(s->str_off[id + 1] - s->str_off[id]);
        unsigned char* buf = arena_take(len);
        if (!buf) FAIL("string arena exhausted");
        memcpy(buf, s->heap + s->str_off[id], len);
        v = tv_computed_string(buf, len, 0);
        break; }
      case RDFGPU_EX_LIT_STR:
        if (e->u >= g_n_regexes) FAIL("string constant %u out of range", e->u);
        v = tv_computed_string((const unsigned char*)g_regexes[e->u].pattern, g_regexes[e->u].pattern_len, e->lo < 0 ? 0 : (u32)e->lo);
        break;
      case RDFGPU_EX_STRLEN: {   /* strlen.rs: chars().count() */
        if (sp < 1 || st[sp - 1].kind != 1) FAIL("STRLEN needs a typed value");
        val a = st[--sp]; const unsigned char* q; size_t qn;
        v = tv_null();
        if (!str_bytes(s, &a, &q, &qn)) break;
        int64_t chars = 0;
        for (size_t i = 0; i < qn; i += utf8_len(q[i])) chars++;
        v.tag = RDFGPU_TV_INTEGER; v.lo = chars;
        break; }
      case RDFGPU_EX_SUBSTR: {   /* sub_str.rs:83-121 */
        if (e->u != 2 && e->u != 3) FAIL("SUBSTR takes 2 or 3 operands");
        if (sp < (int)e->u) FAIL("SUBSTR: stack underflow");
        val len_v = tv_null(); if (e->u == 3) len_v = st[--sp];
        val from = st[--sp], a = st[--sp];
        v = tv_null();
        const unsigned char* q; size_t qn;
        if (!str_bytes(s, &a, &q, &qn)) break;
        /* Integer::try_from(typed value): numerics only; float / double / decimal go through Decimal (not restated: refused) */
        if (num_kind(from.tag) != NK_NONE && from.tag != RDFGPU_TV_INT && from.tag != RDFGPU_TV_INTEGER) FAIL("SUBSTR with a float / double / decimal position is not restated");
        if (e->u == 3 && num_kind(len_v.tag) != NK_NONE && len_v.tag != RDFGPU_TV_INT && len_v.tag != RDFGPU_TV_INTEGER) FAIL("SUBSTR with a float / double / decimal length is not restated");
        if (num_kind(from.tag) == NK_NONE || (e->u == 3 && num_kind(len_v.tag) == NK_NONE)) break;
        if (from.lo < 1 || (e->u == 3 && len_v.lo < 0)) break;           /* usize::try_from(negative) / index 0: errors */
        size_t starts[4096]; size_t nch = 0;                              /* byte offset of every character */
        for (size_t i = 0; i < qn; i += utf8_len(q[i])) { if (nch >= 4096) FAIL("SUBSTR over more than 4096 characters"); starts[nch++] = i; }
        size_t c0 = (size_t)(from.lo - 1), b0, b1;
        if (c0 >= nch) { b0 = b1 = qn; }
        else { b0 = starts[c0]; size_t c1 = e->u == 3 ? c0 + (size_t)len_v.lo : nch; b1 = c1 >= nch ? qn : starts[c1]; }
        unsigned char* buf = arena_take(b1 - b0);
        if (!buf) FAIL("string arena exhausted");
        memcpy(buf, q + b0, b1 - b0);
        v = tv_computed_string(buf, b1 - b0, a.aux);
        break; }
      case RDFGPU_EX_STRBEFORE: case RDFGPU_EX_STRAFTER: {   /* str_before.rs / str_after.rs; argument compatibility string_literal.rs:80-95 */
        if (sp < 2 || st[sp - 1].kind != 1 || st[sp - 2].kind != 1) FAIL("STRBEFORE / STRAFTER need two typed values");
        val b = st[--sp], a = st[--sp];
        const unsigned char *q, *r; size_t qn, rn;
        v = tv_null();
        if (a.tag != RDFGPU_TV_STRING || b.tag != RDFGPU_TV_STRING) break;
        if (b.aux != 0 && b.aux != a.aux) break;
        if (!str_bytes(s, &a, &q, &qn) || !str_bytes(s, &b, &r, &rn)) break;
        size_t pos = (size_t)-1;
        for (size_t i = 0; rn <= qn && i + rn <= qn; i++) if (memcmp(q + i, r, rn) == 0) { pos = i; break; }
        if (pos == (size_t)-1) { v = tv_computed_string((const unsigned char*)"", 0, 0); break; }   /* "" without a language */
        size_t b0 = e->op == RDFGPU_EX_STRAFTER ? pos + rn : 0, b1 = e->op == RDFGPU_EX_STRAFTER ? qn : pos;
        unsigned char* buf = arena_take(b1 - b0);
        if (!buf) FAIL("string arena exhausted");
        memcpy(buf, q + b0, b1 - b0);
        v = tv_computed_string(buf, b1 - b0, a.aux);
        break; }
      case RDFGPU_EX_UCASE: case RDFGPU_EX_LCASE: {   /* ucase.rs / lcase.rs: str::to_uppercase / to_lowercase, language kept */
        if (sp < 1 || st[sp - 1].kind != 1) FAIL("UCASE / LCASE needs a typed value");
        val a = st[--sp]; const unsigned char* q; size_t qn;
        v = tv_null();
        if (!str_bytes(s, &a, &q, &qn)) break;
        unsigned char* buf = arena_take(qn);
        if (!buf) FAIL("string arena exhausted");
        for (size_t i = 0; i < qn; i++) {
          if (q[i] >= 0x80) FAIL("UCASE / LCASE over a string with non-ASCII characters needs Unicode case tables (not restated)");
          buf[i] = e->op == RDFGPU_EX_UCASE ? (unsigned char)((q[i] >= 'a' && q[i] <= 'z') ? q[i] - 32 : q[i]) : (unsigned char)((q[i] >= 'A' && q[i] <= 'Z') ? q[i] + 32 : q[i]);
        }
        v = tv_computed_string(buf, qn, a.aux);
        break; }
      case RDFGPU_EX_EBV: if (sp < 1 || st[sp - 1].kind != 1) FAIL("EBV needs a typed value"); { val a = st[--sp]; v.kind = 2; v.b = tv_ebv(&a); } break;
      case RDFGPU_EX_ID_EQ: case RDFGPU_EX_ID_NEQ: {
        if (sp < 2 || st[sp - 1].kind != 0 || st[sp - 2].kind != 0) FAIL("id comparison needs two ids");
        u32 b = st[--sp].id, a = st[--sp].id; v.kind = 2;
        v.b = (a == 0 || b == 0) ? 2 : (uint8_t)((a == b) == (e->op == RDFGPU_EX_ID_EQ)); break; }
      case RDFGPU_EX_AND: case RDFGPU_EX_OR: { /* SQL three-valued logic expr_builder_context.rs:393-434 */
        if (sp < 2 || st[sp - 1].kind != 2 || st[sp - 2].kind != 2) FAIL("AND/OR need two booleans");
        uint8_t b = st[--sp].b, a = st[--sp].b; v.kind = 2;
        if (e->op == RDFGPU_EX_AND) v.b = (a == 0 || b == 0) ? 0 : (a == 2 || b == 2) ? 2 : 1;
        else v.b = (a == 1 || b == 1) ? 1 : (a == 2 || b == 2) ? 2 : 0;
        break; }
      case RDFGPU_EX_NOT: if (sp < 1 || st[sp - 1].kind != 2) FAIL("NOT needs a boolean"); { uint8_t a = st[--sp].b; v.kind = 2; v.b = a == 2 ? 2 : !a; } break;
      case RDFGPU_EX_IS_COMPATIBLE: { /* is_compatible.rs:97-136 */
        if (sp < 2 || st[sp - 1].kind != 0 || st[sp - 2].kind != 0) FAIL("IS_COMPATIBLE needs two ids");
        u32 b = st[--sp].id, a = st[--sp].id; v.kind = 2; v.b = (a == 0 || b == 0 || a == b); break; }
      case RDFGPU_EX_BOUND: if (sp < 1 || st[sp - 1].kind != 0) FAIL("BOUND needs an id"); { u32 a = st[--sp].id; v.kind = 2; v.b = a != 0; } break;
      case RDFGPU_EX_BOOL_AS_TV: if (sp < 1 || st[sp - 1].kind != 2) FAIL("BOOLEAN_AS_TERM needs a boolean"); { uint8_t a = st[--sp].b; v = a == 2 ? tv_null() : tv_bool(a); } break;
      default: FAIL("unknown expression op %u", e->op);
    }
    if (sp >= STK) FAIL("expression stack overflow");
    st[sp++] = v;
  }
  if (sp != 1) FAIL("expression leaves %d values", sp);
  *result = st[0];
  return 0;
}

int orc_eval_bool(const orc_store* s, const rdfgpu_expr_node* prog, u32 n, const u32* const* cols, u32 n_cols, u64 n_rows, uint8_t* out) {
  for (u64 i = 0; i < n_rows; i++) {
    rowctx r = {cols, n_cols, i, NULL, 0, 0}; val v;
    if (eval_prog(s, prog, n, &r, &v)) return -1;
    if (v.kind != 2) FAIL("program does not yield a boolean");
    out[i] = v.b;
  }
  return 0;
}
int orc_eval_tv(const orc_store* s, const rdfgpu_expr_node* prog, u32 n, const u32* const* cols, u32 n_cols, u64 n_rows, rdfgpu_typed_value* out, int64_t* out_hi) {
  for (u64 i = 0; i < n_rows; i++) {
    rowctx r = {cols, n_cols, i, NULL, 0, 0}; val v;
    if (eval_prog(s, prog, n, &r, &v)) return -1;
    if (v.kind != 1) FAIL("program does not yield a typed value");
    memset(&out[i], 0, sizeof out[i]);
    out[i].tag = v.tag; out[i].flags = v.flags; out[i].aux = v.aux; out[i].lo = v.lo; out_hi[i] = 0;
    if (v.tag == RDFGPU_TV_DECIMAL) { out[i].lo = (int64_t)(u64)v.dec; out_hi[i] = (int64_t)(v.dec >> 64); }
  }
  return 0;
}

/* the string table (REGEX patterns / string constants) the next orc_eval_* calls of this thread refer to */
void orc_eval_set_table(const rdfgpu_regex* regexes, u32 n_regexes) { g_regexes = regexes; g_n_regexes = n_regexes; }
int orc_eval_str(const orc_store* s, const rdfgpu_expr_node* prog, u32 n, const rdfgpu_regex* regexes, u32 n_regexes,
                 const u32* const* cols, u32 n_cols, u64 n_rows, uint8_t* out_state, u32* out_lang, u64* out_off, uint8_t* out_bytes, u64 cap) {
  g_regexes = regexes; g_n_regexes = n_regexes;
  u64 at = 0;
  for (u64 i = 0; i < n_rows; i++) {
    rowctx r = {cols, n_cols, i, NULL, 0, 0}; val v;
    out_off[i] = at;
    if (eval_prog(s, prog, n, &r, &v)) return -1;
    if (v.kind != 1) FAIL("program does not yield a typed value");
    const unsigned char* p; size_t len;
    out_state[i] = 0; out_lang[i] = 0;
    if (!str_bytes(s, &v, &p, &len)) continue;
    if (at + len > cap) FAIL("orc_eval_str: output buffer of %llu bytes is too small", (unsigned long long)cap);
    memcpy(out_bytes + at, p, len); at += len;
    out_state[i] = 1; out_lang[i] = v.aux;
  }
  out_off[n_rows] = at;
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Operators: FilterExec, HashJoinExec(CollectLeft), CrossJoinExec, NestedLoopJoinExec    */
/* Semantics: SURVEY.md Appendix B 1-6 (join/rewrite.rs:71-221, join/logical.rs:251-338)  */
/* ------------------------------------------------------------------------------------ */
void orc_table_free(orc_table* t) { for (u32 c = 0; c < RDFGPU_MAX_COLUMNS; c++) { free(t->cols[c]); t->cols[c] = NULL; } t->n_rows = 0; t->n_cols = 0; }

typedef struct { orc_table t; u64 cap; } tbuilder;
static void tb_init(tbuilder* b, u32 n_cols, u64 cap) {
  memset(b, 0, sizeof *b); b->t.n_cols = n_cols; b->cap = cap ? cap : 16;
  for (u32 c = 0; c < n_cols; c++) b->t.cols[c] = (u32*)malloc(b->cap * sizeof(u32));
}
static inline void tb_reserve(tbuilder* b) {
  if (b->t.n_rows < b->cap) return;
  b->cap *= 2; for (u32 c = 0; c < b->t.n_cols; c++) b->t.cols[c] = (u32*)realloc(b->t.cols[c], b->cap * sizeof(u32));
}

typedef struct {
  const orc_store* s; const rdfgpu_plan_desc* d; const orc_bound_table* tables; u32 n_tables; rdfgpu_metrics* m;
} pctx;

static int out_width(const pctx* c, const rdfgpu_plan_node* nd, u32 full, const u32** proj, u32* np) {
  if (nd->n_proj == RDFGPU_NO_PROJECTION) {
    if (full > RDFGPU_MAX_COLUMNS) FAIL("too many columns");   /* orc_table holds RDFGPU_MAX_COLUMNS */
    *proj = NULL; *np = full; return 0;
  }
  if ((u64)nd->proj_off + nd->n_proj > c->d->n_pool) FAIL("projection out of pool range");
  *proj = c->d->pool + nd->proj_off; *np = nd->n_proj;
  for (u32 i = 0; i < *np; i++) if ((*proj)[i] >= full) FAIL("projection column out of range");
  if (*np > RDFGPU_MAX_COLUMNS) FAIL("too many columns");
  return 0;
}
static inline void emit_row(tbuilder* b, const rowctx* r, const u32* proj, u32 np, int right_null) {
  tb_reserve(b);
  for (u32 k = 0; k < np; k++) {
    u32 c = proj ? proj[k] : k;
    u32 v = (c < r->nl) ? r->lc[c][r->li] : (right_null ? 0u : r->rc[c - r->nl][r->ri]);
    b->t.cols[k][b->t.n_rows] = v;
  }
  b->t.n_rows++;
}

static inline u64 key_hash(const u32* const* cols, const u32* keys, u32 nk, u64 row) {
  u64 h = 0x9e3779b97f4a7c15ULL;
  for (u32 k = 0; k < nk; k++) h = mix64(h ^ cols[keys[k]][row]);
  return h;
}

static int exec_node(const pctx* c, u32 idx, orc_table* out);

static int exec_filter(const pctx* c, const rdfgpu_plan_node* nd, orc_table* out) {
  orc_table in; memset(&in, 0, sizeof in);
  if (exec_node(c, (u32)nd->left, &in)) return -1;
  const u32* proj; u32 np; if (out_width(c, nd, in.n_cols, &proj, &np)) { orc_table_free(&in); return -1; }
  tbuilder b; tb_init(&b, np, in.n_rows);
  const u32* const* cols = (const u32* const*)in.cols;
  /* FilterExec works batch by batch; row order inside is preserved */
  for (u64 i = 0; i < in.n_rows; i++) {
    rowctx r = {cols, in.n_cols, i, NULL, 0, 0}; val v;
    if (nd->expr_len) {
      if (eval_prog(c->s, c->d->exprs + nd->expr_off, nd->expr_len, &r, &v)) { orc_table_free(&in); orc_table_free(&b.t); return -1; }
      if (v.kind != 2) { orc_table_free(&in); orc_table_free(&b.t); FAIL("filter predicate is not boolean"); }
      if (v.b != 1) continue; /* keep only `true` (logical_plan_builder.rs:114-129) */
    }
    emit_row(&b, &r, proj, np, 0);
  }
  orc_table_free(&in);
  *out = b.t;
  return 0;
}

static int exec_join(const pctx* c, const rdfgpu_plan_node* nd, orc_table* out) {
  orc_table L, R; memset(&L, 0, sizeof L); memset(&R, 0, sizeof R);
  if (exec_node(c, (u32)nd->left, &L)) return -1;
  if (exec_node(c, (u32)nd->right, &R)) { orc_table_free(&L); return -1; }
  int rc = -1;
  const u32* proj; u32 np;
  tbuilder b; memset(&b, 0, sizeof b);
  u32* heads = NULL; u32* next = NULL; uint8_t* visited = NULL;
  if (out_width(c, nd, L.n_cols + R.n_cols, &proj, &np)) goto done;
  if (L.n_cols + R.n_cols > 2 * RDFGPU_MAX_COLUMNS) { snprintf(g_err, sizeof g_err, "too many columns"); goto done; }
  tb_init(&b, np, L.n_rows > R.n_rows ? L.n_rows : R.n_rows);
  const u32* const* lc = (const u32* const*)L.cols; const u32* const* rcols = (const u32* const*)R.cols;
  const rdfgpu_expr_node* prog = nd->expr_len ? c->d->exprs + nd->expr_off : NULL;
  int left_join = nd->join_type == RDFGPU_JOIN_LEFT;
  if (left_join) visited = (uint8_t*)calloc(L.n_rows ? L.n_rows : 1, 1);

  if (nd->kind == RDFGPU_NODE_HASH_JOIN) {
    if (nd->n_keys == 0 || nd->n_keys > RDFGPU_MAX_KEYS) { snprintf(g_err, sizeof g_err, "bad key count"); goto done; }
    for (u32 k = 0; k < nd->n_keys; k++) if (nd->left_keys[k] >= L.n_cols || nd->right_keys[k] >= R.n_cols) { snprintf(g_err, sizeof g_err, "join key out of range"); goto done; }
    /* build (CollectLeft): chained hash table over the whole left side */
    u64 nb = 16; while (nb < 2 * L.n_rows) nb <<= 1;
    heads = (u32*)malloc(nb * sizeof(u32)); memset(heads, 0xff, nb * sizeof(u32));
    next = (u32*)malloc((L.n_rows ? L.n_rows : 1) * sizeof(u32));
    for (u64 i = 0; i < L.n_rows; i++) {
      int has_null = 0; for (u32 k = 0; k < nd->n_keys; k++) if (lc[nd->left_keys[k]][i] == 0) has_null = 1;
      if (has_null) { next[i] = 0xffffffffu; continue; } /* NullEqualsNothing join/rewrite.rs:89,217 */
      u64 h = key_hash(lc, nd->left_keys, nd->n_keys, i) & (nb - 1);
      next[i] = heads[h]; heads[h] = (u32)i;
    }
    /* probe, one right batch (8192 rows) at a time — order of output is irrelevant (multiset) */
    for (u64 j = 0; j < R.n_rows; j++) {
      int has_null = 0; for (u32 k = 0; k < nd->n_keys; k++) if (rcols[nd->right_keys[k]][j] == 0) has_null = 1;
      if (has_null) continue;
      u64 h = key_hash(rcols, nd->right_keys, nd->n_keys, j) & (nb - 1);
      for (u32 i = heads[h]; i != 0xffffffffu; i = next[i]) {
        int eq = 1; for (u32 k = 0; k < nd->n_keys; k++) if (lc[nd->left_keys[k]][i] != rcols[nd->right_keys[k]][j]) { eq = 0; break; }
        if (!eq) continue;
        rowctx r = {lc, L.n_cols, i, rcols, R.n_cols, j};
        if (prog) { val v; if (eval_prog(c->s, prog, nd->expr_len, &r, &v)) goto done; if (v.kind != 2) { snprintf(g_err, sizeof g_err, "join filter is not boolean"); goto done; } if (v.b != 1) continue; }
        if (visited) visited[i] = 1;
        emit_row(&b, &r, proj, np, 0);
      }
    }
  } else { /* CrossJoinExec / NestedLoopJoinExec */
    for (u64 i = 0; i < L.n_rows; i++) for (u64 j = 0; j < R.n_rows; j++) {
      rowctx r = {lc, L.n_cols, i, rcols, R.n_cols, j};
      if (prog) { val v; if (eval_prog(c->s, prog, nd->expr_len, &r, &v)) goto done; if (v.kind != 2) { snprintf(g_err, sizeof g_err, "join filter is not boolean"); goto done; } if (v.b != 1) continue; }
      if (visited) visited[i] = 1;
      emit_row(&b, &r, proj, np, 0);
    }
  }
  if (left_join) for (u64 i = 0; i < L.n_rows; i++) if (!visited[i]) { rowctx r = {lc, L.n_cols, i, rcols, R.n_cols, 0}; emit_row(&b, &r, proj, np, 1); }
  *out = b.t; memset(&b, 0, sizeof b);
  rc = 0;
done:
  free(heads); free(next); free(visited); orc_table_free(&L); orc_table_free(&R); if (rc) orc_table_free(&b.t);
  return rc;
}

/* DISTINCT + TopK per group (..Q5 (Execution Plan).snap:5-9: AggregateExec gby = sort keys with first_value, then
   SortExec TopK(fetch)): sort rows by (group, keys), drop adjacent duplicates, keep the first k of every group. */
typedef struct { u32 g; u64 k[4]; u64 row; } topk_row;
static int topk_cmp(const void* a, const void* b) {
  const topk_row* x = (const topk_row*)a; const topk_row* y = (const topk_row*)b;
  if (x->g != y->g) return x->g < y->g ? -1 : 1;
  for (int i = 0; i < 4; i++) if (x->k[i] != y->k[i]) return x->k[i] < y->k[i] ? -1 : 1;
  return 0;
}
static int exec_topk(const pctx* c, const rdfgpu_plan_node* nd, orc_table* out) {
  orc_table in; memset(&in, 0, sizeof in);
  if (exec_node(c, (u32)nd->left, &in)) return -1;
  const u32* proj; u32 np; if (out_width(c, nd, in.n_cols, &proj, &np)) { orc_table_free(&in); return -1; }
  if (nd->n_keys < 1 || nd->n_keys > 4) { orc_table_free(&in); FAIL("TopK needs 1 to 4 sort keys"); }
  const int has_group = nd->table_slot != 0; const u32 gcol = has_group ? nd->table_slot - 1 : 0;
  for (u32 i = 0; i < nd->n_keys; i++) if (nd->left_keys[i] >= in.n_cols) { orc_table_free(&in); FAIL("TopK: key column out of range"); }
  if (has_group && gcol >= in.n_cols) { orc_table_free(&in); FAIL("TopK: group column out of range"); }
  for (u32 q = 0; q < np; q++) {
    const u32 pc = proj ? proj[q] : q;
    int covered = has_group && pc == gcol;
    for (u32 i = 0; i < nd->n_keys; i++) covered = covered || (pc == nd->left_keys[i] && nd->right_keys[i] == RDFGPU_SORT_BY_ID);
    if (!covered) { orc_table_free(&in); FAIL("TopK: output column %u is neither the group nor a sort key by id", pc); }
  }
  topk_row* rows = (topk_row*)malloc((in.n_rows ? in.n_rows : 1) * sizeof(topk_row));
  for (u64 r = 0; r < in.n_rows; r++) {
    rows[r].g = has_group ? in.cols[gcol][r] : 0; rows[r].row = r; rows[r].k[1] = 0; rows[r].k[2] = 0; rows[r].k[3] = 0;
    for (u32 i = 0; i < nd->n_keys; i++) {
      const u32 id = in.cols[nd->left_keys[i]][r];
      if (nd->right_keys[i] == RDFGPU_SORT_BY_ID) { rows[r].k[i] = id; continue; }
      const val v = enc_tv(c->s, id);   /* NULLS FIRST: tag 0 sorts before everything */
      if (nd->right_keys[i] == RDFGPU_SORT_BY_DOUBLE) {   /* sortable_term/builder.rs:36-39: numerics order by Double::from(Numeric); f64 total order */
        const int nk = num_kind(v.tag);
        if (nk == NK_NONE) { rows[r].k[i] = 0; continue; }
        double d = to_f64(&v, nk); u64 bits; memcpy(&bits, &d, 8);
        rows[r].k[i] = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
        continue;
      }
      if (v.tag != RDFGPU_TV_NULL && v.tag != RDFGPU_TV_STRING && v.tag != RDFGPU_TV_NAMED_NODE && v.tag != RDFGPU_TV_BLANK_NODE) {
        free(rows); orc_table_free(&in); FAIL("TopK: sort by term over a typed value of tag %u", v.tag);
      }
      rows[r].k[i] = ((u64)v.tag << 56) | ((u64)v.lo & 0x00FFFFFFFFFFFFFFull);
    }
  }
  qsort(rows, in.n_rows, sizeof(topk_row), topk_cmp);
  tbuilder b; tb_init(&b, np, in.n_rows);
  u64 taken = 0;
  for (u64 r = 0; r < in.n_rows; r++) {
    if (r > 0 && rows[r].g != rows[r - 1].g) taken = 0;
    if (r > 0 && topk_cmp(&rows[r], &rows[r - 1]) == 0) continue;   /* DISTINCT */
    if (taken >= nd->table_cols) continue;
    taken++;
    tb_reserve(&b);
    for (u32 q = 0; q < np; q++) b.t.cols[q][b.t.n_rows] = in.cols[proj ? proj[q] : q][rows[r].row];
    b.t.n_rows++;
  }
  free(rows); orc_table_free(&in);
  *out = b.t;
  return 0;
}

/* KleenePlusClosureExec, lib/physical/src/paths/kleene_plus/physical.rs:246-384.  The reference keeps the inner paths in
   `initial_paths_map` (graph -> set of (start, end)), every path seen in the set `all_paths`, and extends the paths of
   `current_delta` by the initial paths that start where they end — of the same graph (compute_new_single_graph_paths,
   :362-384) or of every graph (compute_new_cross_graph_paths, :344-360) — until an iteration adds nothing.  Sets are
   sorted arrays here. */
typedef struct { u32 g, s, e; } cpath;
static int cpath_cmp(const void* a, const void* b) {
  const cpath* x = (const cpath*)a; const cpath* y = (const cpath*)b;
  if (x->g != y->g) return x->g < y->g ? -1 : 1;
  if (x->s != y->s) return x->s < y->s ? -1 : 1;
  if (x->e != y->e) return x->e < y->e ? -1 : 1;
  return 0;
}
static int cpath_by_start(const void* a, const void* b) {
  const cpath* x = (const cpath*)a; const cpath* y = (const cpath*)b;
  if (x->s != y->s) return x->s < y->s ? -1 : 1;
  return cpath_cmp(a, b);
}
static u64 cpath_unique(cpath* p, u64 n) {
  if (n == 0) return 0;
  qsort(p, n, sizeof(cpath), cpath_cmp);
  u64 m = 1;
  for (u64 i = 1; i < n; i++) if (cpath_cmp(&p[i], &p[m - 1]) != 0) p[m++] = p[i];
  return m;
}
static int exec_closure(const pctx* c, const rdfgpu_plan_node* nd, orc_table* out) {
  orc_table in;
  if (nd->left < 0) FAIL("KleenePlusClosureExec needs an input");
  if (exec_node(c, (u32)nd->left, &in)) return -1;
  if (in.n_cols != 3) { orc_table_free(&in); FAIL("inner paths are (graph, start, end)"); }
  const int cross = nd->join_type == 1;
  u64 n = in.n_rows;
  cpath* all = (cpath*)malloc((n ? n : 1) * sizeof(cpath));
  for (u64 i = 0; i < n; i++) {
    all[i].g = in.cols[0][i]; all[i].s = in.cols[1][i]; all[i].e = in.cols[2][i];
    if (all[i].s == 0 || all[i].e == 0) { free(all); orc_table_free(&in); FAIL("Could not obtain start / end value from inner paths."); }   /* :321-326 */
  }
  orc_table_free(&in);
  u64 na = cpath_unique(all, n);                       /* collect_next_batch :309-341: every inner path is in the closure ... */
  const u64 ni = na;
  cpath* init = (cpath*)malloc((ni ? ni : 1) * sizeof(cpath));   /* ... the initial paths, ordered for look-up by (graph,) start */
  memcpy(init, all, ni * sizeof(cpath));
  if (cross) qsort(init, ni, sizeof(cpath), cpath_by_start);
  cpath* delta = (cpath*)malloc((na ? na : 1) * sizeof(cpath)); u64 nd_ = na;   /* ... and in the first delta */
  memcpy(delta, all, na * sizeof(cpath));
  while (nd_ > 0) {
    u64 cap = 1024, nn = 0;
    cpath* next = (cpath*)malloc(cap * sizeof(cpath));
    for (u64 i = 0; i < nd_; i++) {
      const cpath p = delta[i];
      u64 b = 0, e2 = ni;                              /* first initial path (of graph p.g) starting at p.e */
      while (b < e2) {
        const u64 m = b + (e2 - b) / 2;
        const int less = cross ? init[m].s < p.e : (init[m].g < p.g || (init[m].g == p.g && init[m].s < p.e));
        if (less) b = m + 1; else e2 = m;
      }
      for (u64 k = b; k < ni && init[k].s == p.e && (cross || init[k].g == p.g); k++) {
        cpath q = {p.g, p.s, init[k].e};                /* path_ac :369-373: the graph of the path being extended */
        if (bsearch(&q, all, na, sizeof(cpath), cpath_cmp)) continue;    /* all_paths.insert() == false */
        if (nn == cap) { cap *= 2; next = (cpath*)realloc(next, cap * sizeof(cpath)); }
        next[nn++] = q;
      }
    }
    nn = cpath_unique(next, nn);                        /* a path reached twice in one round enters the set once */
    if (nn == 0) { free(next); break; }
    all = (cpath*)realloc(all, (na + nn) * sizeof(cpath));
    memcpy(all + na, next, nn * sizeof(cpath));
    na += nn;
    qsort(all, na, sizeof(cpath), cpath_cmp);
    free(delta); delta = next; nd_ = nn;
  }
  free(delta); free(init);
  const u32* proj = nd->n_proj == RDFGPU_NO_PROJECTION ? NULL : c->d->pool + nd->proj_off;
  const u32 np = proj ? nd->n_proj : 3;
  out->n_cols = np; out->n_rows = na;
  for (u32 q = 0; q < np; q++) {
    const u32 src = proj ? proj[q] : q;
    if (src >= 3) { free(all); FAIL("closure projection column %u out of range", src); }
    out->cols[q] = (u32*)malloc((na ? na : 1) * sizeof(u32));
    for (u64 i = 0; i < na; i++) out->cols[q][i] = src == 0 ? all[i].g : src == 1 ? all[i].s : all[i].e;
  }
  free(all);
  return 0;
}

static int exec_node(const pctx* c, u32 idx, orc_table* out) {
  if (idx >= c->d->n_nodes) FAIL("node index out of range");
  const rdfgpu_plan_node* nd = &c->d->nodes[idx];
  memset(out, 0, sizeof *out);
  int rc = 0;
  switch (nd->kind) {
    case RDFGPU_NODE_DATA_SOURCE: {
      orc_scan_result r;
      if (orc_scan(c->s, nd->scan, c->d->pool, -1, &r)) return -1;
      out->n_cols = r.n_cols; out->n_rows = r.n_rows;
      for (u32 k = 0; k < r.n_cols; k++) { out->cols[k] = r.cols[k]; r.cols[k] = NULL; }
      orc_scan_result_free(&r);
      break; }
    case RDFGPU_NODE_FILTER: rc = exec_filter(c, nd, out); break;
    case RDFGPU_NODE_PROJECTION: { rdfgpu_plan_node f = *nd; f.expr_len = 0; rc = exec_filter(c, &f, out); break; }
    case RDFGPU_NODE_HASH_JOIN: case RDFGPU_NODE_CROSS_JOIN: case RDFGPU_NODE_NESTED_LOOP_JOIN: rc = exec_join(c, nd, out); break;
    case RDFGPU_NODE_TOPK: rc = exec_topk(c, nd, out); break;
    case RDFGPU_NODE_CLOSURE: rc = exec_closure(c, nd, out); break;
    case RDFGPU_NODE_UNION: {   /* UnionExec: every row of the left input, then every row of the right one */
      if (nd->left < 0 || nd->right < 0) FAIL("UNION needs two inputs");
      orc_table l, r;
      if (exec_node(c, (u32)nd->left, &l)) return -1;
      if (exec_node(c, (u32)nd->right, &r)) { orc_table_free(&l); return -1; }
      if (l.n_cols != r.n_cols) { orc_table_free(&l); orc_table_free(&r); FAIL("UNION inputs have %u and %u columns", l.n_cols, r.n_cols); }
      const u32* proj = nd->n_proj == RDFGPU_NO_PROJECTION ? NULL : c->d->pool + nd->proj_off;
      const u32 np = proj ? nd->n_proj : l.n_cols;
      out->n_cols = np; out->n_rows = l.n_rows + r.n_rows;
      for (u32 q = 0; q < np; q++) {
        const u32 src = proj ? proj[q] : q;
        if (src >= l.n_cols) { orc_table_free(&l); orc_table_free(&r); FAIL("UNION projection column %u out of range", src); }
        out->cols[q] = (u32*)malloc((out->n_rows ? out->n_rows : 1) * sizeof(u32));
        memcpy(out->cols[q], l.cols[src], l.n_rows * sizeof(u32));
        memcpy(out->cols[q] + l.n_rows, r.cols[src], r.n_rows * sizeof(u32));
      }
      orc_table_free(&l); orc_table_free(&r);
      break; }
    case RDFGPU_NODE_TABLE: {
      if (nd->table_slot >= c->n_tables) FAIL("table slot %u not bound", nd->table_slot);
      const orc_bound_table* t = &c->tables[nd->table_slot];
      if (t->n_cols != nd->table_cols) FAIL("bound table has %u columns, node declares %u", t->n_cols, nd->table_cols);
      out->n_cols = t->n_cols; out->n_rows = t->n_rows;
      for (u32 k = 0; k < t->n_cols; k++) { out->cols[k] = (u32*)malloc((t->n_rows ? t->n_rows : 1) * sizeof(u32)); memcpy(out->cols[k], t->cols[k], t->n_rows * sizeof(u32)); }
      break; }
    default: FAIL("unknown node kind %u", nd->kind);
  }
  if (rc == 0 && c->m && idx != c->d->root) c->m->intermediate_rows += out->n_rows;
  return rc;
}

int orc_plan_execute(const orc_store* s, const rdfgpu_plan_desc* desc, const orc_bound_table* tables, u32 n_tables, orc_table* out, rdfgpu_metrics* metrics) {
  if (metrics) memset(metrics, 0, sizeof *metrics);
  pctx c = {s, desc, tables, n_tables, metrics};
  g_regexes = desc->regexes; g_n_regexes = desc->n_regexes;
  if (exec_node(&c, desc->root, out)) return -1;
  if (metrics) metrics->output_rows = out->n_rows;
  return 0;
}
