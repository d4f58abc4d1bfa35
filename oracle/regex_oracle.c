/* regex_oracle.c — CPU restatement of SPARQL REGEX for the ORACLE (test infrastructure only; nothing under
 * rdf-fusion_amd/ links or calls this).
 *
 * Follows lib/functions/src/scalar/strings/regex.rs:47-141: `compile_pattern` (flags s m i x q; `q` escapes the
 * pattern; an unknown flag or an invalid pattern is the error value) and `Regex::is_match` = unanchored search.
 * The `regex` crate itself (1.12.2, Cargo.lock:3834-3835) is a third-party dependency absent from /root/reference;
 * this restates its documented syntax for the subset the device supports and a little more (anchors anywhere):
 *   literals, `.`, `[...]`, `( )`, `(?: )`, `(?P<n> )`, `|`, `* + ? {m} {m,} {m,n}` (+ lazy suffix), `^ $ \A \z`,
 *   escaped punctuation, `\n \r \t \f \v \a \xHH`, the Perl classes `\d \w \s \D \W \S` (also inside `[...]`) and the word
 *   boundaries `\b \B` with their ASCII members — the crate's Unicode classes agree with those on all-ASCII subjects; a
 *   subject with a non-ASCII character then yields -2 ("needs the crate's Unicode tables") instead of a guess;
 *   inline flags `(?imsxU-imsxU)` / `(?flags:...)`; nested classes, ASCII POSIX classes `[[:alpha:]]`, the class set operations
 *   `&& -- ~~` (a class is kept as an expression TREE over leaf sets and evaluated per code point — the device folds it into
 *   one set at compile time); Unicode general categories `\p{..}` with their ASCII members under the same -2 rule.
 * Deliberately a DIFFERENT algorithm from the device's bit-parallel Glushkov automaton over bytes: a Thompson
 * program (char / split / jmp / assert / match) run by a Pike VM over decoded CODE POINTS, so UTF-8 expansion,
 * follow-set or anchoring mistakes on the device side do not cancel out.  tests/ cross-check both against Python's
 * `re` as a third opinion.
 * Returns: 1 match, 0 no match, -1 = the error value (bad flag / unsupported or invalid pattern), -2 = not answerable
 * without the Unicode tables (see above).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { I_CHAR, I_SPLIT, I_JMP, I_MATCH, I_BOL, I_EOL, I_BOT, I_EOT, I_WB, I_NWB };
typedef struct {
  int op;
  int x, y;            /* SPLIT: both targets; JMP: x */
  uint64_t ascii[2];   /* CHAR: members among U+0000..U+007F */
  uint32_t extra[2];   /* CHAR: up to two non-ASCII members (case folds) */
  int n_extra;
  int neg;             /* CHAR: negated set */
  int any_non_ascii;   /* CHAR: (non-negated) also every code point >= 0x80 */
  int tree;            /* CHAR: > 0 = root of a class expression (cnode index + 1) instead of the flat set */
} inst;
/* class expression: leaf set, complement, union, intersection, difference, symmetric difference */
enum { C_LEAF, C_NOT, C_OR, C_AND, C_DIFF, C_XOR };
typedef struct { int op, l, r; uint64_t ascii[2]; uint32_t extra[2]; int n_extra; } cnode;

typedef struct {
  const unsigned char* p; size_t n, i;
  int f_i, f_s, f_m, f_x;
  inst* prog; int n_inst, cap;
  int bad;
  int ascii_only;      /* \d \w \s \b used: restated with their ASCII members; a non-ASCII subject is then "needs the Unicode tables" (-2) */
  cnode* cn; int n_cn, cap_cn;
} rx;
static int cnew(rx* r, int op, int l, int rr) {
  if (r->n_cn == r->cap_cn) { r->cap_cn = r->cap_cn ? 2 * r->cap_cn : 32; r->cn = (cnode*)realloc(r->cn, (size_t)r->cap_cn * sizeof(cnode)); }
  memset(&r->cn[r->n_cn], 0, sizeof(cnode));
  r->cn[r->n_cn].op = op; r->cn[r->n_cn].l = l; r->cn[r->n_cn].r = rr;
  return r->n_cn++;
}
static int ceval(const rx* r, int k, uint32_t cp) {
  const cnode* c = &r->cn[k];
  switch (c->op) {
    case C_LEAF: { if (cp < 0x80) return (int)((c->ascii[cp >> 6] >> (cp & 63)) & 1); for (int e = 0; e < c->n_extra; e++) if (c->extra[e] == cp) return 1; return 0; }
    case C_NOT: return !ceval(r, c->l, cp);
    case C_OR: return ceval(r, c->l, cp) || ceval(r, c->r, cp);
    case C_AND: return ceval(r, c->l, cp) && ceval(r, c->r, cp);
    case C_DIFF: return ceval(r, c->l, cp) && !ceval(r, c->r, cp);
    default: return ceval(r, c->l, cp) != ceval(r, c->r, cp);
  }
}

static int emit(rx* r, int op) {
  if (r->n_inst == r->cap) { r->cap = r->cap ? 2 * r->cap : 64; r->prog = (inst*)realloc(r->prog, (size_t)r->cap * sizeof(inst)); }
  memset(&r->prog[r->n_inst], 0, sizeof(inst));
  r->prog[r->n_inst].op = op;
  return r->n_inst++;
}
static void set_add(inst* c, unsigned ch) { c->ascii[ch >> 6] |= 1ull << (ch & 63); }
static int set_has(const inst* c, unsigned ch) { return (int)((c->ascii[ch >> 6] >> (ch & 63)) & 1); }
static void fold(rx* r, inst* c) {   /* `i`: simple case folding of ASCII letters + the two non-ASCII partners */
  if (!r->f_i) return;
  for (unsigned ch = 'a'; ch <= 'z'; ch++) if (set_has(c, ch) || set_has(c, ch - 32)) { set_add(c, ch); set_add(c, ch - 32); }
  if (set_has(c, 'k')) c->extra[c->n_extra++] = 0x212A;   /* KELVIN SIGN */
  if (set_has(c, 's')) c->extra[c->n_extra++] = 0x017F;   /* LATIN SMALL LETTER LONG S */
}
static void skip_x(rx* r) {
  if (!r->f_x) return;
  for (;;) {
    while (r->i < r->n && strchr(" \t\n\r\f\v", r->p[r->i])) r->i++;
    if (r->i < r->n && r->p[r->i] == '#') { while (r->i < r->n && r->p[r->i] != '\n') r->i++; continue; }
    break;
  }
}
static int decode(const unsigned char* s, size_t n, size_t* i, uint32_t* cp) {
  unsigned c = s[*i];
  int len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
  if (!len || *i + (size_t)len > n) return -1;
  uint32_t v = len == 1 ? c : len == 2 ? (c & 0x1F) : len == 3 ? (c & 0x0F) : (c & 0x07);
  for (int k = 1; k < len; k++) { if ((s[*i + k] & 0xC0) != 0x80) return -1; v = (v << 6) | (s[*i + k] & 0x3F); }
  *i += (size_t)len; *cp = v;
  return 0;
}
static int escape(rx* r, unsigned* out) {   /* after '\\' */
  if (r->i >= r->n) return -1;
  unsigned e = r->p[r->i++];
  switch (e) {
    case 'n': *out = '\n'; return 0; case 'r': *out = '\r'; return 0; case 't': *out = '\t'; return 0;
    case 'f': *out = '\f'; return 0; case 'v': *out = '\v'; return 0; case 'a': *out = 7; return 0;
    case 'x': {
      if (r->i + 2 > r->n) return -1;
      unsigned v = 0;
      for (int k = 0; k < 2; k++) {
        unsigned h = r->p[r->i++];
        int d = h >= '0' && h <= '9' ? (int)(h - '0') : h >= 'a' && h <= 'f' ? (int)(h - 'a' + 10) : h >= 'A' && h <= 'F' ? (int)(h - 'A' + 10) : -1;
        if (d < 0) return -1;
        v = v * 16 + (unsigned)d;
      }
      if (v >= 0x80) return -1;
      *out = v; return 0;
    }
    default:
      /* regex-syntax is_escapeable_character: any ASCII character that is not a letter, a digit, '<' or '>' */
      if (e < 0x80 && e != '<' && e != '>' && !(e >= '0' && e <= '9') && !((e | 32) >= 'a' && (e | 32) <= 'z')) { *out = e; return 0; }
      return -1;   /* \d \w \s \b \p{..}: not restated */
  }
}

/* `\d \w \s` / `\D \W \S` at p[i] (after the backslash): adds the ASCII members to `c`; 1 if consumed */
static int perl_class(rx* r, inst* c) {
  if (r->i >= r->n) return 0;
  unsigned e = r->p[r->i];
  uint64_t m[2] = {0, 0};
#define M_ADD(ch) (m[(ch) >> 6] |= 1ull << ((ch) & 63))
  if (e == 'd' || e == 'D') { for (unsigned k = '0'; k <= '9'; k++) M_ADD(k); }
  else if (e == 'w' || e == 'W') { for (unsigned k = '0'; k <= '9'; k++) M_ADD(k); for (unsigned k = 'a'; k <= 'z'; k++) { M_ADD(k); M_ADD(k - 32); } M_ADD('_'); }
  else if (e == 's' || e == 'S') { M_ADD('\t'); M_ADD('\n'); M_ADD('\v'); M_ADD('\f'); M_ADD('\r'); M_ADD(' '); }
  else return 0;
#undef M_ADD
  r->i++;
  r->ascii_only = 1;
  const int neg = e == 'D' || e == 'W' || e == 'S';
  for (unsigned k = 0; k < 0x80; k++) { const int in = (int)((m[k >> 6] >> (k & 63)) & 1); if (in != neg) set_add(c, k); }
  return 1;
}

/* Fragments: the parser emits code for a sub-expression into [start, n_inst) and every fragment falls through at its
 * end, so concatenation is emission order; repetition and alternation copy / patch with relative fix-ups. */
static int parse_alt(rx* r);

static void shift_targets(rx* r, int from, int by, int lo) {   /* instructions moved up by `by`: fix absolute targets >= lo */
  for (int k = from; k < r->n_inst; k++) {
    inst* in = &r->prog[k];
    if (in->op == I_SPLIT) { if (in->x >= lo) in->x += by; if (in->y >= lo) in->y += by; }
    else if (in->op == I_JMP) { if (in->x >= lo) in->x += by; }
  }
}
static void insert_at(rx* r, int at, int op) {   /* open a slot at `at` */
  emit(r, op);
  memmove(&r->prog[at + 1], &r->prog[at], (size_t)(r->n_inst - 1 - at) * sizeof(inst));
  memset(&r->prog[at], 0, sizeof(inst));
  r->prog[at].op = op;
  shift_targets(r, at + 1, 1, at);
}
static void copy_frag(rx* r, int start, int end) {   /* append a copy of [start, end) */
  const int delta = r->n_inst - start;
  for (int k = start; k < end; k++) {
    int id = emit(r, 0);
    r->prog[id] = r->prog[k];
    inst* in = &r->prog[id];
    if (in->op == I_SPLIT) { in->x += delta; in->y += delta; }
    else if (in->op == I_JMP) in->x += delta;
  }
}

/* the ASCII members of a POSIX class (1) / a Unicode general category or Any / ASCII / Alphabetic (2) into m; 0 = unknown name */
static void m_range(uint64_t* m, unsigned a, unsigned b) { for (unsigned k = a; k <= b; k++) m[k >> 6] |= 1ull << (k & 63); }
static void m_chars(uint64_t* m, const char* cs) { for (; *cs; cs++) m[(unsigned char)*cs >> 6] |= 1ull << ((unsigned char)*cs & 63); }
static int posix_members(const char* name, size_t n, uint64_t* m) {
#define IS(x) (n == strlen(x) && !memcmp(name, x, n))
  if (IS("alnum")) { m_range(m, '0', '9'); m_range(m, 'A', 'Z'); m_range(m, 'a', 'z'); }
  else if (IS("alpha")) { m_range(m, 'A', 'Z'); m_range(m, 'a', 'z'); }
  else if (IS("ascii")) m_range(m, 0, 127);
  else if (IS("blank")) m_chars(m, " \t");
  else if (IS("cntrl")) { m_range(m, 0, 31); m_range(m, 127, 127); }
  else if (IS("digit")) m_range(m, '0', '9');
  else if (IS("graph")) m_range(m, '!', '~');
  else if (IS("lower")) m_range(m, 'a', 'z');
  else if (IS("print")) m_range(m, ' ', '~');
  else if (IS("punct")) { m_range(m, '!', '/'); m_range(m, ':', '@'); m_range(m, '[', '`'); m_range(m, '{', '~'); }
  else if (IS("space")) m_chars(m, " \t\n\v\f\r");
  else if (IS("upper")) m_range(m, 'A', 'Z');
  else if (IS("word")) { m_range(m, '0', '9'); m_range(m, 'A', 'Z'); m_range(m, 'a', 'z'); m_chars(m, "_"); }
  else if (IS("xdigit")) { m_range(m, '0', '9'); m_range(m, 'A', 'F'); m_range(m, 'a', 'f'); }
  else return 0;
  return 1;
}
static int unicode_members(const char* raw, size_t rn, uint64_t* m) {
  char name[40]; size_t n = 0;
  for (size_t k = 0; k < rn; k++) { char ch = raw[k]; if (ch == '_' || ch == ' ' || ch == '-') continue; if (n >= sizeof name - 1) return 0; name[n++] = (char)(ch >= 'A' && ch <= 'Z' ? ch + 32 : ch); }
  /* general categories that have ASCII members, by every name the crate accepts for them; the others are empty on ASCII */
  static const struct { const char* names; const char* ranges; } T[] = {
    {"any ascii", "\x01\x7f"}, {"l letter lc casedletter alphabetic alpha", "AZaz"}, {"lu uppercaseletter", "AZ"}, {"ll lowercaseletter", "az"},
    {"n number nd decimalnumber", "09"}, {"p punctuation", "!#%*,/:;?@[]__{{}}"}, {"pc connectorpunctuation", "__"}, {"pd dashpunctuation", "--"},
    {"ps openpunctuation", "(([[{{"}, {"pe closepunctuation", "))]]}}"}, {"po otherpunctuation", "!#%\x27**,,./:;?@\\\\"},
    {"s symbol", "$$++<>^^``||~~"}, {"sm mathsymbol", "++<>||~~"}, {"sc currencysymbol", "$$"}, {"sk modifiersymbol", "^^``"},
    {"z separator zs spaceseparator", "  "}, {"c other cc control", "\x01\x1f\x7f\x7f"},
    {"lt titlecaseletter lm modifierletter lo otherletter m mark mn nonspacingmark mc spacingmark me enclosingmark nl letternumber no othernumber "
     "pi initialpunctuation pf finalpunctuation so othersymbol zl lineseparator zp paragraphseparator cf format cs surrogate co privateuse cn unassigned", ""},
  };
  for (size_t t = 0; t < sizeof T / sizeof T[0]; t++) {
    const char* q = T[t].names;
    while (*q) {
      const char* e = q; while (*e && *e != ' ') e++;
      if ((size_t)(e - q) == n && !memcmp(q, name, n)) {
        const unsigned char* g = (const unsigned char*)T[t].ranges;
        for (; g[0]; g += 2) m_range(m, g[0] == 1 && (t == 0 || t == 16) ? 0 : g[0], g[1]);   /* (\x01 stands for U+0000: a NUL cannot sit in the table's strings) */
        return 1;
      }
      q = *e ? e + 1 : e;
    }
  }
  return 0;
}
static void cfold(rx* r, int leaf) {   /* `i` on a leaf set (folding distributes over the set operations' operands, which are unions of leaves) */
  if (!r->f_i) return;
  cnode* c = &r->cn[leaf];
  for (unsigned ch = 'a'; ch <= 'z'; ch++) {
    const int lo = (int)((c->ascii[ch >> 6] >> (ch & 63)) & 1), up = (int)((c->ascii[(ch - 32) >> 6] >> ((ch - 32) & 63)) & 1);
    if (lo || up) { c->ascii[ch >> 6] |= 1ull << (ch & 63); c->ascii[(ch - 32) >> 6] |= 1ull << ((ch - 32) & 63); }
  }
  if ((c->ascii['k' >> 6] >> ('k' & 63)) & 1) c->extra[c->n_extra++] = 0x212A;
  if ((c->ascii['s' >> 6] >> ('s' & 63)) & 1) c->extra[c->n_extra++] = 0x017F;
}
/* `\p{..}` / `\P{..}` / `\pL` at p[i] ('p' / 'P'): members into a fresh leaf; returns the leaf index, -1 = not a \p escape, -2 = bad / needs tables */
static int unicode_leaf(rx* r) {
  if (r->i >= r->n || (r->p[r->i] != 'p' && r->p[r->i] != 'P')) return -1;
  int neg = r->p[r->i] == 'P';
  r->i++;
  if (r->i >= r->n) return -2;
  const char* name; size_t nn;
  if (r->p[r->i] == '{') {
    r->i++;
    if (r->i < r->n && r->p[r->i] == '^') { neg = !neg; r->i++; }
    name = (const char*)r->p + r->i;
    size_t j = r->i; while (j < r->n && r->p[j] != '}') j++;
    if (j >= r->n) return -2;
    nn = j - r->i; r->i = j + 1;
  } else { name = (const char*)r->p + r->i; nn = 1; r->i++; }
  const char* eq = (const char*)memchr(name, '=', nn);
  if (eq) {
    char prop[24]; size_t pn = 0;
    for (const char* q = name; q < eq; q++) { char ch = *q; if (ch == '_' || ch == ' ' || ch == '-') continue; if (pn >= sizeof prop - 1) return -2; prop[pn++] = (char)(ch >= 'A' && ch <= 'Z' ? ch + 32 : ch); }
    if (!((pn == 2 && !memcmp(prop, "gc", 2)) || (pn == 15 && !memcmp(prop, "generalcategory", 15)))) return -2;
    nn -= (size_t)(eq + 1 - name); name = eq + 1;
  }
  uint64_t m[2] = {0, 0};
  if (!unicode_members(name, nn, m)) return -2;
  r->ascii_only = 1;
  int leaf = cnew(r, C_LEAF, 0, 0);
  r->cn[leaf].ascii[0] = m[0]; r->cn[leaf].ascii[1] = m[1];
  cfold(r, leaf);                                   /* regex-syntax folds the class, THEN negates it: (?i)\P{Lu} excludes the lower-case letters too */
  return neg ? cnew(r, C_NOT, leaf, 0) : leaf;
}
/* after '[': returns the index of the class expression, -1 on error */
static int parse_class_expr(rx* r) {
  int neg = 0;
  if (r->i < r->n && r->p[r->i] == '^') { neg = 1; r->i++; }
  int acc = -1, op = 0;                 /* finished operands so far, the operator waiting for its right side */
  int cur = -1;                         /* union of the items of the operand being read */
  int leaf = cnew(r, C_LEAF, 0, 0);     /* its plain characters and ranges */
  int first = 1;
#define UNITE(x) (cur = cur < 0 ? (x) : cnew(r, C_OR, cur, (x)))
#define CLOSE() do { cfold(r, leaf); UNITE(leaf); acc = acc < 0 ? cur : cnew(r, op == 1 ? C_AND : op == 2 ? C_DIFF : C_XOR, acc, cur); cur = -1; leaf = cnew(r, C_LEAF, 0, 0); } while (0)
  for (;;) {
    if (r->i >= r->n) return -1;
    unsigned ch = r->p[r->i];
    if (ch == ']' && !first) { r->i++; break; }
    if (ch == '[') {
      if (r->i + 1 < r->n && r->p[r->i + 1] == ':') {
        size_t j = r->i + 2; int pneg = 0;
        if (j < r->n && r->p[j] == '^') { pneg = 1; j++; }
        size_t j0 = j; while (j < r->n && r->p[j] != ':' && r->p[j] != ']') j++;
        if (j + 1 < r->n && r->p[j] == ':' && r->p[j + 1] == ']') {
          uint64_t m[2] = {0, 0};
          if (!posix_members((const char*)r->p + j0, j - j0, m)) return -1;
          int pl = cnew(r, C_LEAF, 0, 0);
          r->cn[pl].ascii[0] = m[0]; r->cn[pl].ascii[1] = m[1];
          cfold(r, pl);
          if (pneg) pl = cnew(r, C_NOT, pl, 0);
          UNITE(pl);
          r->i = j + 2; first = 0;
          continue;
        }
      }
      r->i++;
      int inner = parse_class_expr(r);
      if (inner < 0) return -1;
      UNITE(inner);
      first = 0;
      continue;
    }
    if ((ch == '&' || ch == '-' || ch == '~') && r->i + 1 < r->n && r->p[r->i + 1] == ch && !(ch == '-' && first)) {
      if (r->i + 2 < r->n && r->p[r->i + 2] == ']') return -1;
      CLOSE();
      op = ch == '&' ? 1 : ch == '-' ? 2 : 3;
      r->i += 2; first = 0;
      continue;
    }
    first = 0;
    if (ch >= 0x80) return -1;
    r->i++;
    if (ch == '\\') {
      { inst tmp; memset(&tmp, 0, sizeof tmp); if (perl_class(r, &tmp)) { r->cn[leaf].ascii[0] |= tmp.ascii[0]; r->cn[leaf].ascii[1] |= tmp.ascii[1]; continue; } }
      { int ul = unicode_leaf(r); if (ul == -2) return -1; if (ul >= 0) { UNITE(ul); continue; } }
      if (escape(r, &ch)) return -1;
    }
    unsigned hi = ch;
    if (r->i + 1 < r->n && r->p[r->i] == '-' && r->p[r->i + 1] != ']' && r->p[r->i + 1] != '-') {
      r->i++;
      hi = r->p[r->i++];
      if (hi >= 0x80 || hi == '[') return -1;
      if (hi == '\\' && escape(r, &hi)) return -1;
      if (hi < ch) return -1;
    }
    for (unsigned k = ch; k <= hi; k++) r->cn[leaf].ascii[k >> 6] |= 1ull << (k & 63);
  }
  CLOSE();
  return neg ? cnew(r, C_NOT, acc, 0) : acc;
#undef UNITE
#undef CLOSE
}
static int parse_class(rx* r) {
  int root = parse_class_expr(r);
  if (root < 0) return -1;
  int id = emit(r, I_CHAR);
  r->prog[id].tree = root + 1;
  return 0;
}

static int parse_atom(rx* r) {
  unsigned ch = r->p[r->i];
  if (ch == '(') {
    r->i++;
    if (r->i < r->n && r->p[r->i] == '?') {
      if (r->i + 1 < r->n && r->p[r->i + 1] == ':') r->i += 2;
      else if (r->i + 1 < r->n && (r->p[r->i + 1] == 'P' || r->p[r->i + 1] == '<')) {
        r->i += r->p[r->i + 1] == 'P' ? 2 : 1;
        if (r->i >= r->n || r->p[r->i] != '<') return -1;
        while (r->i < r->n && r->p[r->i] != '>') r->i++;
        if (r->i >= r->n) return -1;
        r->i++;
      } else {   /* (?flags) for the rest of the enclosing group / (?flags:...) for this group */
        r->i++;
        int on = 1, any = 0;
        const int s_i = r->f_i, s_s = r->f_s, s_m = r->f_m, s_x = r->f_x;
        for (;; r->i++) {
          if (r->i >= r->n) return -1;
          unsigned fc = r->p[r->i];
          if (fc == ')' || fc == ':') break;
          if (fc == '-') { if (!on) return -1; on = 0; continue; }
          if (fc == 'i') r->f_i = on; else if (fc == 's') r->f_s = on; else if (fc == 'm') r->f_m = on; else if (fc == 'x') r->f_x = on;
          else if (fc != 'U') return -1;
          any = 1;
        }
        if (!any) return -1;
        if (r->p[r->i] == ')') { r->i++; return 1; }     /* a directive: no code, no repetition may follow it */
        r->i++;
        if (parse_alt(r)) return -1;
        r->f_i = s_i; r->f_s = s_s; r->f_m = s_m; r->f_x = s_x;
        if (r->i >= r->n || r->p[r->i] != ')') return -1;
        r->i++;
        return 0;
      }
    }
    {
      const int s_i = r->f_i, s_s = r->f_s, s_m = r->f_m, s_x = r->f_x;
      if (parse_alt(r)) return -1;
      r->f_i = s_i; r->f_s = s_s; r->f_m = s_m; r->f_x = s_x;
    }
    if (r->i >= r->n || r->p[r->i] != ')') return -1;
    r->i++;
    return 0;
  }
  if (ch == '[') { r->i++; return parse_class(r); }
  if (ch == '.') {
    r->i++;
    int id = emit(r, I_CHAR);
    inst* c = &r->prog[id];
    c->neg = 1;                         /* everything ... */
    if (!r->f_s) set_add(c, '\n');      /* ... except a line feed */
    return 0;
  }
  if (ch == '^') { r->i++; emit(r, r->f_m ? I_BOL : I_BOT); return 0; }
  if (ch == '$') { r->i++; emit(r, r->f_m ? I_EOL : I_EOT); return 0; }
  if (ch == '\\') {
    r->i++;
    if (r->i < r->n && r->p[r->i] == 'A') { r->i++; emit(r, I_BOT); return 0; }
    if (r->i < r->n && r->p[r->i] == 'z') { r->i++; emit(r, I_EOT); return 0; }
    if (r->i < r->n && (r->p[r->i] == 'b' || r->p[r->i] == 'B')) { emit(r, r->p[r->i] == 'b' ? I_WB : I_NWB); r->i++; r->ascii_only = 1; return 0; }
    { int id = emit(r, I_CHAR); inst c = r->prog[id]; if (perl_class(r, &c)) { fold(r, &c); r->prog[id] = c; return 0; } r->n_inst--; }
    { int ul = unicode_leaf(r); if (ul == -2) return -1; if (ul >= 0) { int id = emit(r, I_CHAR); r->prog[id].tree = ul + 1; return 0; } }
    unsigned b;
    if (escape(r, &b)) return -1;
    int id = emit(r, I_CHAR);
    set_add(&r->prog[id], b);
    fold(r, &r->prog[id]);
    return 0;
  }
  if (ch == '*' || ch == '+' || ch == '?' || ch == '{' || ch == ')' || ch == '|') return -1;
  uint32_t cp;
  if (decode(r->p, r->n, &r->i, &cp)) return -1;
  int id = emit(r, I_CHAR);
  if (cp < 0x80) { set_add(&r->prog[id], cp); fold(r, &r->prog[id]); }
  else { if (r->f_i) return -1; r->prog[id].extra[0] = cp; r->prog[id].n_extra = 1; }
  return 0;
}

static void wrap_star(rx* r, int start) {   /* L1: split L2, L3; L2: e; jmp L1; L3: */
  insert_at(r, start, I_SPLIT);
  int j = emit(r, I_JMP);
  r->prog[j].x = start;
  r->prog[start].x = start + 1; r->prog[start].y = r->n_inst;
}
static void wrap_opt(rx* r, int start) {    /* split L1, L2; L1: e; L2: */
  insert_at(r, start, I_SPLIT);
  r->prog[start].x = start + 1; r->prog[start].y = r->n_inst;
}
static int parse_uint(rx* r, unsigned* v) {
  if (r->i >= r->n || r->p[r->i] < '0' || r->p[r->i] > '9') return -1;
  *v = 0;
  while (r->i < r->n && r->p[r->i] >= '0' && r->p[r->i] <= '9') { *v = *v * 10 + (r->p[r->i] - '0'); if (*v > 1000) return -2; r->i++; }
  return 0;
}
static int parse_repeat(rx* r) {
  int start = r->n_inst;
  { const int a = parse_atom(r); if (a < 0) return -1; if (a == 1) return 0; }
  for (;;) {
    skip_x(r);
    if (r->i >= r->n) break;
    unsigned ch = r->p[r->i];
    if (ch != '*' && ch != '+' && ch != '?' && ch != '{') break;
    if (ch == '{') {
      r->i++;
      unsigned lo = 0, hi = 0; int open = 0;
      if (parse_uint(r, &lo)) return -1;
      if (r->i < r->n && r->p[r->i] == ',') { r->i++; int e = parse_uint(r, &hi); if (e == -2) return -1; if (e) open = 1; }
      else hi = lo;
      if (r->i >= r->n || r->p[r->i] != '}') return -1;
      r->i++;
      if (!open && hi < lo) return -1;
      if (lo + (open ? 1 : hi - lo) > 64) return -1;
      /* e{lo,hi} = e x lo, then (e)? x (hi - lo)   |   e{lo,} = e x lo, then e* */
      const int end = r->n_inst, len = end - start;
      inst* body = (inst*)malloc((size_t)(len ? len : 1) * sizeof(inst));
      memcpy(body, &r->prog[start], (size_t)len * sizeof(inst));
      r->n_inst = start;                                 /* re-emit from scratch */
      const unsigned copies = lo + (open ? 1 : hi - lo);
      for (unsigned k = 0; k < copies; k++) {
        const int at = r->n_inst;
        for (int q = 0; q < len; q++) {
          int id = emit(r, 0);
          r->prog[id] = body[q];
          inst* in = &r->prog[id];
          if (in->op == I_SPLIT) { in->x += at - start; in->y += at - start; }
          else if (in->op == I_JMP) in->x += at - start;
        }
        if (k >= lo) { if (open) wrap_star(r, at); else wrap_opt(r, at); }
      }
      free(body);
    } else {
      r->i++;
      if (ch == '*') wrap_star(r, start);
      else if (ch == '?') wrap_opt(r, start);
      else {   /* e+ = e e* */
        const int end = r->n_inst;
        copy_frag(r, start, end);
        wrap_star(r, end);
      }
    }
    if (r->i < r->n && r->p[r->i] == '?') r->i++;   /* lazy: same language */
  }
  return 0;
}
static int parse_cat(rx* r) {
  for (;;) {
    skip_x(r);
    if (r->i >= r->n || r->p[r->i] == '|' || r->p[r->i] == ')') return 0;
    if (parse_repeat(r)) return -1;
  }
}
static int parse_alt(rx* r) {   /* split L1, L2; L1: a; jmp END; L2: b ... */
  int start = r->n_inst;
  if (parse_cat(r)) return -1;
  int jumps[256], n_j = 0;
  skip_x(r);
  while (r->i < r->n && r->p[r->i] == '|') {
    r->i++;
    insert_at(r, start, I_SPLIT);
    for (int k = 0; k < n_j; k++) if (jumps[k] >= start) jumps[k]++;   /* jmp slots behind the insertion moved up by one */
    if (n_j == 255) return -1;
    jumps[n_j++] = emit(r, I_JMP);
    r->prog[start].x = start + 1; r->prog[start].y = r->n_inst;
    start = r->n_inst;
    if (parse_cat(r)) return -1;
    skip_x(r);
  }
  for (int k = 0; k < n_j; k++) r->prog[jumps[k]].x = r->n_inst;
  return 0;
}

/* ---- Pike VM over code points ---- */
typedef struct { int* pc; int n; } tlist;
static void add_thread(const rx* r, tlist* l, int* mark, int gen, int pc, const uint32_t* s, size_t i, size_t len) {
  if (mark[pc] == gen) return;
  mark[pc] = gen;
  const inst* in = &r->prog[pc];
  switch (in->op) {
    case I_JMP: add_thread(r, l, mark, gen, in->x, s, i, len); break;
    case I_SPLIT: add_thread(r, l, mark, gen, in->x, s, i, len); add_thread(r, l, mark, gen, in->y, s, i, len); break;
    case I_BOT: if (i == 0) add_thread(r, l, mark, gen, pc + 1, s, i, len); break;
    case I_EOT: if (i == len) add_thread(r, l, mark, gen, pc + 1, s, i, len); break;
    case I_BOL: if (i == 0 || s[i - 1] == '\n') add_thread(r, l, mark, gen, pc + 1, s, i, len); break;
    case I_EOL: if (i == len || s[i] == '\n') add_thread(r, l, mark, gen, pc + 1, s, i, len); break;
    case I_WB: case I_NWB: {   /* ASCII word characters (the subject is all-ASCII when these occur) */
      const int a = i > 0 && s[i - 1] < 0x80 && (s[i - 1] == '_' || (s[i - 1] >= '0' && s[i - 1] <= '9') || ((s[i - 1] | 32) >= 'a' && (s[i - 1] | 32) <= 'z'));
      const int b = i < len && s[i] < 0x80 && (s[i] == '_' || (s[i] >= '0' && s[i] <= '9') || ((s[i] | 32) >= 'a' && (s[i] | 32) <= 'z'));
      if ((a != b) == (in->op == I_WB)) add_thread(r, l, mark, gen, pc + 1, s, i, len);
      break; }
    default: l->pc[l->n++] = pc;
  }
}
static int char_ok(const rx* r, const inst* in, uint32_t cp) {
  if (in->tree) return ceval(r, in->tree - 1, cp);
  int member = 0;
  if (cp < 0x80) member = set_has(in, cp);
  else { for (int k = 0; k < in->n_extra; k++) member = member || in->extra[k] == cp; if (in->any_non_ascii) member = 1; }
  return in->neg ? !member : member;
}

int orc_regex_is_match(const char* pattern, size_t pattern_len, const char* flags, size_t flags_len, const unsigned char* subject, size_t subject_len) {
  rx r; memset(&r, 0, sizeof r);
  int q = 0;
  for (size_t k = 0; k < flags_len; k++) {
    switch (flags[k]) {
      case 's': r.f_s = 1; break; case 'm': r.f_m = 1; break; case 'i': r.f_i = 1; break; case 'x': r.f_x = 1; break; case 'q': q = 1; break;
      default: return -1;   /* regex.rs:137 */
    }
  }
  r.p = (const unsigned char*)pattern; r.n = pattern_len;
  int bad = 0;
  if (q) {   /* regex::escape: every character stands for itself */
    if (r.f_x) bad = 1;   /* `x` would still strip the whitespace that escape() leaves alone: outside the restated subset */
    while (!bad && r.i < r.n) {
      uint32_t cp;
      if (decode(r.p, r.n, &r.i, &cp)) { bad = 1; break; }
      int id = emit(&r, I_CHAR);
      if (cp < 0x80) { set_add(&r.prog[id], cp); fold(&r, &r.prog[id]); }
      else { if (r.f_i) { bad = 1; break; } r.prog[id].extra[0] = cp; r.prog[id].n_extra = 1; }
    }
  } else {
    if (parse_alt(&r) || r.i != r.n) bad = 1;
  }
  if (bad) { free(r.prog); free(r.cn); return -1; }
  emit(&r, I_MATCH);

  uint32_t* cps = (uint32_t*)malloc((subject_len + 1) * sizeof(uint32_t));
  size_t len = 0, at = 0;
  while (at < subject_len) { if (decode(subject, subject_len, &at, &cps[len])) { free(cps); free(r.prog); free(r.cn); return -1; } len++; }
  if (r.ascii_only) for (size_t k = 0; k < len; k++) if (cps[k] >= 0x80) { free(cps); free(r.prog); free(r.cn); return -2; }

  /* raw = successor pcs of the previous character; clist = their epsilon closure at this position */
  int* raw = (int*)malloc((size_t)r.n_inst * sizeof(int)); int n_raw = 0;
  int* next_raw = (int*)malloc((size_t)r.n_inst * sizeof(int));
  tlist clist = {(int*)malloc((size_t)r.n_inst * sizeof(int)), 0};
  int* mark = (int*)calloc((size_t)r.n_inst, sizeof(int));
  int gen = 0, matched = 0;
  for (size_t i = 0; i <= len && !matched; i++) {
    gen++;
    clist.n = 0;
    for (int k = 0; k < n_raw; k++) add_thread(&r, &clist, mark, gen, raw[k], cps, i, len);
    add_thread(&r, &clist, mark, gen, 0, cps, i, len);   /* unanchored search: a match may start here */
    int n_next = 0;
    for (int k = 0; k < clist.n; k++) {
      const inst* in = &r.prog[clist.pc[k]];
      if (in->op == I_MATCH) { matched = 1; break; }
      if (i < len && char_ok(&r, in, cps[i])) next_raw[n_next++] = clist.pc[k] + 1;
    }
    int* t = raw; raw = next_raw; next_raw = t; n_raw = n_next;
  }
  free(raw); free(next_raw); free(clist.pc); free(mark); free(cps); free(r.prog); free(r.cn);
  return matched;
}
