// regex_compile.hpp — host-side compiler of SPARQL REGEX patterns into a bit-parallel position automaton
// (Glushkov construction, <= 64 byte positions) that the device simulates per row with one u64 of state.
//
// Replaces `compile_pattern` + `Regex::is_match` (lib/functions/src/scalar/strings/regex.rs:47-141; the `regex`
// crate 1.12, Cargo.lock:3834-3835).  is_match is an unanchored search, so greedy / lazy makes no difference and
// no captures are needed.  Supported syntax — everything else is reported as UNSUPPORTED at plan compile, never
// answered differently from the crate:
//   literals (any UTF-8), `.`, `[...]` / `[^...]` with ASCII members and ranges, `( )`, `(?: )`, `(?P<n> )`, `|`,
//   `* + ? {m} {m,} {m,n}` (+ lazy suffix), escaped punctuation, `\n \r \t \f \v \xHH`, a leading `^` / `\A` and a
//   trailing `$` / `\z` of a pattern without top-level alternation;
//   `\d \w \s` and `\D \W \S` (also inside classes) and the word boundaries `\b \B` outside repetitions — with their
//   ASCII members: on an all-ASCII subject that IS the crate's Unicode class; a subject with a non-ASCII byte raises the
//   plan's run-time error (RegexProg::ascii_only) instead of being answered without the Unicode tables;
//   flags: `i` (ASCII letters, incl. the two non-ASCII simple folds K <-> U+212A and s <-> U+017F), `s`, `m`, `x`, `q` — as the
//   SPARQL flags argument and INLINE: `(?i)`, `(?s-i)`, `(?imsx-imsx:...)` (a flag group applies to the rest of the enclosing
//   group, `(?flags:...)` to its own; `U` = swap greediness is accepted and irrelevant for is_match);
//   classes: nested `[a[bc]]`, ASCII POSIX classes `[[:alpha:]]` / `[[:^digit:]]`, the set operations `&&` `--` `~~`
//   (left to right, operands unions — regex-syntax's precedence), negation of the whole;
//   Unicode general categories `\p{L}` `\pL` `\p{Lu}` `\P{N}` `\p{^P}` ... (short and long names) with their ASCII members,
//   under the same rule as the Perl classes (ASCII assignments are the same in every Unicode version; a non-ASCII subject
//   raises the run-time error).
// Not supported: scripts / binary properties other than Any / ASCII / Alphabetic, `(?-u)` / `(?R)`, non-ASCII class members,
// non-ASCII letters under `i`, anchors elsewhere, `\b` under a repetition, > 64 positions.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "regex_prog.hpp"

namespace rdfgpu {

namespace regex_detail {

struct ByteSet {
  uint64_t w[4] = {0, 0, 0, 0};
  void add(unsigned b) { w[b >> 6] |= 1ull << (b & 63); }
  void add_range(unsigned a, unsigned b) { for (unsigned c = a; c <= b; c++) add(c); }
  bool has(unsigned b) const { return (w[b >> 6] >> (b & 63)) & 1; }
};

enum Kind { EMPTY, LEAF, CAT, ALT, STAR, PLUS, OPT, A_START, A_END, A_WORDB, A_NWORDB };
struct Node {
  Kind kind = EMPTY;
  ByteSet set;
  std::vector<std::shared_ptr<Node>> kids;
};
using NodeP = std::shared_ptr<Node>;
inline NodeP mk(Kind k) { auto n = std::make_shared<Node>(); n->kind = k; return n; }
inline NodeP leaf(const ByteSet& s) { auto n = mk(LEAF); n->set = s; return n; }
inline NodeP leaf_range(unsigned a, unsigned b) { ByteSet s; s.add_range(a, b); return leaf(s); }
inline NodeP cat(std::vector<NodeP> k) { if (k.empty()) return mk(EMPTY); if (k.size() == 1) return k[0]; auto n = mk(CAT); n->kids = std::move(k); return n; }
inline NodeP alt(std::vector<NodeP> k) { if (k.size() == 1) return k[0]; auto n = mk(ALT); n->kids = std::move(k); return n; }
inline NodeP un(Kind k, NodeP a) { auto n = mk(k); n->kids.push_back(std::move(a)); return n; }

// any non-ASCII Unicode scalar as UTF-8 (loose on surrogates / overlongs: the subject strings are valid UTF-8)
inline NodeP non_ascii_char() {
  return alt({cat({leaf_range(0xC2, 0xDF), leaf_range(0x80, 0xBF)}),
              cat({leaf_range(0xE0, 0xEF), leaf_range(0x80, 0xBF), leaf_range(0x80, 0xBF)}),
              cat({leaf_range(0xF0, 0xF4), leaf_range(0x80, 0xBF), leaf_range(0x80, 0xBF), leaf_range(0x80, 0xBF)})});
}
inline NodeP bytes_seq(const unsigned char* b, size_t n) {
  std::vector<NodeP> k;
  for (size_t i = 0; i < n; i++) { ByteSet s; s.add(b[i]); k.push_back(leaf(s)); }
  return cat(std::move(k));
}

struct Parser {
  const unsigned char* p; size_t n, i = 0;
  bool f_i = false, f_s = false, f_x = false, f_m = false;
  bool ascii_only = false;          // a Perl / Unicode class or a word boundary was used
  std::string err;
  bool fail(const char* m) { if (err.empty()) err = m; return false; }
  bool eof() const { return i >= n; }

  void skip_x() {   // `x`: whitespace and #-comments are ignored outside classes
    if (!f_x) return;
    for (;;) {
      while (!eof() && (p[i] == ' ' || p[i] == '\t' || p[i] == '\n' || p[i] == '\r' || p[i] == '\f' || p[i] == '\v')) i++;
      if (!eof() && p[i] == '#') { while (!eof() && p[i] != '\n') i++; continue; }
      break;
    }
  }
  static bool is_alpha(unsigned c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

  // one ASCII set (after case folding) -> node, adding the two non-ASCII simple case folds
  NodeP ascii_set_node(ByteSet s) {
    if (!f_i) return leaf(s);
    for (unsigned c = 'a'; c <= 'z'; c++) if (s.has(c) || s.has(c - 32)) { s.add(c); s.add(c - 32); }
    std::vector<NodeP> alts{leaf(s)};
    static const unsigned char kelvin[3] = {0xE2, 0x84, 0xAA}, long_s[2] = {0xC5, 0xBF};
    if (s.has('k')) alts.push_back(bytes_seq(kelvin, 3));
    if (s.has('s')) alts.push_back(bytes_seq(long_s, 2));
    return alt(std::move(alts));
  }

  bool utf8_len(size_t at, size_t& len) {
    const unsigned c = p[at];
    len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
    if (len == 0 || at + len > n) return fail("pattern is not valid UTF-8");
    for (size_t k = 1; k < len; k++) if ((p[at + k] & 0xC0) != 0x80) return fail("pattern is not valid UTF-8");
    return true;
  }
  NodeP literal_at(size_t at, size_t len) {
    if (len == 1) { ByteSet s; s.add(p[at]); return ascii_set_node(s); }
    if (f_i) { fail("non-ASCII literal under the `i` flag"); return nullptr; }
    return bytes_seq(p + at, len);
  }

  // `\d \w \s` / `\D \W \S` at p[i]: their ASCII members (the crate's Unicode classes restricted to ASCII)
  bool perl_class(ByteSet& s) {
    if (eof()) return false;
    const unsigned e = p[i];
    ByteSet m;
    if (e == 'd' || e == 'D') m.add_range('0', '9');
    else if (e == 'w' || e == 'W') { m.add_range('0', '9'); m.add_range('a', 'z'); m.add_range('A', 'Z'); m.add('_'); }
    else if (e == 's' || e == 'S') { m.add('\t'); m.add('\n'); m.add('\v'); m.add('\f'); m.add('\r'); m.add(' '); }
    else return false;
    i++;
    ascii_only = true;
    if (e == 'D' || e == 'W' || e == 'S') { for (unsigned c = 0; c < 0x80; c++) if (!m.has(c)) s.add(c); }
    else for (unsigned c = 0; c < 0x80; c++) if (m.has(c)) s.add(c);
    return true;
  }

  // escape after '\\': returns a single byte value in `c` (is_byte) or fails
  bool escape_byte(unsigned& c) {
    if (eof()) return fail("trailing backslash");
    const unsigned e = p[i++];
    switch (e) {
      case 'n': c = '\n'; return true;   case 'r': c = '\r'; return true;   case 't': c = '\t'; return true;
      case 'f': c = '\f'; return true;   case 'v': c = '\v'; return true;   case 'a': c = 7; return true;
      case 'x': {
        if (i + 2 > n) return fail("bad \\x escape");
        unsigned v = 0;
        for (int k = 0; k < 2; k++) {
          const unsigned h = p[i++];
          const int d = h >= '0' && h <= '9' ? (int)(h - '0') : h >= 'a' && h <= 'f' ? (int)(h - 'a' + 10) : h >= 'A' && h <= 'F' ? (int)(h - 'A' + 10) : -1;
          if (d < 0) return fail("bad \\x escape");
          v = v * 16 + (unsigned)d;
        }
        if (v >= 0x80) return fail("non-ASCII \\x escape");
        c = v; return true;
      }
      default:
        // escapable = any ASCII character that is not a letter, a digit, '<' or '>' (the crate's is_escapeable_character)
        if (e < 0x80 && !is_alpha(e) && !(e >= '0' && e <= '9') && e != '<' && e != '>') { c = e; return true; }
        return fail("escape class needs Unicode tables or is not an escape");
    }
  }

  // A class as a set over ASCII + the two non-ASCII partners of simple case folding (KELVIN SIGN, LATIN SMALL LETTER LONG S)
  // + "every other non-ASCII scalar": the universe in which union / intersection / difference / complement are exact.
  struct CSet {
    ByteSet a; bool kelvin = false, long_s = false, other = false;
    void fold() {   // simple case folding, as regex-syntax applies it to both operands of a set operation and to the finished class
      for (unsigned c = 'a'; c <= 'z'; c++) if (a.has(c) || a.has(c - 32)) { a.add(c); a.add(c - 32); }
      if (a.has('k') || kelvin) { a.add('k'); a.add('K'); kelvin = true; }
      if (a.has('s') || long_s) { a.add('s'); a.add('S'); long_s = true; }
    }
    void complement() { for (int w = 0; w < 2; w++) a.w[w] = ~a.w[w]; a.w[2] = a.w[3] = 0; kelvin = !kelvin; long_s = !long_s; other = !other; }
    void unite(const CSet& o) { for (int w = 0; w < 2; w++) a.w[w] |= o.a.w[w]; kelvin |= o.kelvin; long_s |= o.long_s; other |= o.other; }
    void intersect(const CSet& o) { for (int w = 0; w < 2; w++) a.w[w] &= o.a.w[w]; kelvin &= o.kelvin; long_s &= o.long_s; other &= o.other; }
    void subtract(const CSet& o) { for (int w = 0; w < 2; w++) a.w[w] &= ~o.a.w[w]; kelvin &= !o.kelvin; long_s &= !o.long_s; other &= !o.other; }
    void sym_diff(const CSet& o) { for (int w = 0; w < 2; w++) a.w[w] ^= o.a.w[w]; kelvin ^= o.kelvin; long_s ^= o.long_s; other ^= o.other; }
  };
  NodeP cset_node(const CSet& c) {
    std::vector<NodeP> alts;
    ByteSet ascii = c.a; ascii.w[2] = ascii.w[3] = 0;
    if (ascii.w[0] | ascii.w[1]) alts.push_back(leaf(ascii));
    static const unsigned char kelvin[3] = {0xE2, 0x84, 0xAA}, long_s[2] = {0xC5, 0xBF};
    if (c.other) {
      if (!c.kelvin || !c.long_s) { fail("class that excludes the non-ASCII case partner of k / s but keeps other non-ASCII characters"); return nullptr; }
      alts.push_back(non_ascii_char());
    } else {
      if (c.kelvin) alts.push_back(bytes_seq(kelvin, 3));
      if (c.long_s) alts.push_back(bytes_seq(long_s, 2));
    }
    if (alts.empty()) { ByteSet none; return leaf(none); }   // the empty class: a position no byte enters
    return alt(std::move(alts));
  }
  // ASCII POSIX classes `[:name:]` (regex-syntax: ASCII-only by definition)
  static bool posix_class(const std::string& name, ByteSet& m) {
    auto r = [&](unsigned a, unsigned b) { m.add_range(a, b); };
    if (name == "alnum") { r('0', '9'); r('A', 'Z'); r('a', 'z'); }
    else if (name == "alpha") { r('A', 'Z'); r('a', 'z'); }
    else if (name == "ascii") r(0, 0x7F);
    else if (name == "blank") { m.add(' '); m.add('\t'); }
    else if (name == "cntrl") { r(0, 0x1F); m.add(0x7F); }
    else if (name == "digit") r('0', '9');
    else if (name == "graph") r('!', '~');
    else if (name == "lower") r('a', 'z');
    else if (name == "print") r(' ', '~');
    else if (name == "punct") { r('!', '/'); r(':', '@'); r('[', '`'); r('{', '~'); }
    else if (name == "space") { m.add('\t'); m.add('\n'); m.add('\v'); m.add('\f'); m.add('\r'); m.add(' '); }
    else if (name == "upper") r('A', 'Z');
    else if (name == "word") { r('0', '9'); r('A', 'Z'); r('a', 'z'); m.add('_'); }
    else if (name == "xdigit") { r('0', '9'); r('A', 'F'); r('a', 'f'); }
    else return false;
    return true;
  }
  // The ASCII members of a Unicode general category (or of Any / ASCII / Alphabetic), by short or long name.  The ASCII block's
  // assignments are the same in every version of the Unicode tables.
  static bool unicode_class_ascii(std::string name, ByteSet& m) {
    std::string k;
    for (char ch : name) if (ch != '_' && ch != ' ' && ch != '-') k.push_back((char)(ch >= 'A' && ch <= 'Z' ? ch + 32 : ch));   // loose matching (UAX44-LM3)
    auto r = [&](unsigned a, unsigned b) { m.add_range(a, b); };
    auto add = [&](const char* cs) { for (; *cs; cs++) m.add((unsigned char)*cs); };
    auto Lu = [&] { r('A', 'Z'); }; auto Ll = [&] { r('a', 'z'); }; auto Nd = [&] { r('0', '9'); };
    auto Pc = [&] { m.add('_'); }; auto Pd = [&] { m.add('-'); }; auto Ps = [&] { add("([{"); }; auto Pe = [&] { add(")]}"); };
    auto Po = [&] { add("!\"#%&'*,./:;?@\\"); };
    auto Sm = [&] { add("+<=>|~"); }; auto Sc = [&] { m.add('$'); }; auto Sk = [&] { add("^`"); };
    auto Zs = [&] { m.add(' '); }; auto Cc = [&] { r(0, 0x1F); m.add(0x7F); };
    if (k == "any") r(0, 0x7F);
    else if (k == "ascii") r(0, 0x7F);
    else if (k == "alphabetic" || k == "alpha") { Lu(); Ll(); }
    else if (k == "l" || k == "letter") { Lu(); Ll(); }
    else if (k == "lu" || k == "uppercaseletter") Lu();
    else if (k == "ll" || k == "lowercaseletter") Ll();
    else if (k == "lc" || k == "casedletter") { Lu(); Ll(); }
    else if (k == "lt" || k == "titlecaseletter" || k == "lm" || k == "modifierletter" || k == "lo" || k == "otherletter") {}
    else if (k == "m" || k == "mark" || k == "mn" || k == "nonspacingmark" || k == "mc" || k == "spacingmark" || k == "me" || k == "enclosingmark") {}
    else if (k == "n" || k == "number" || k == "nd" || k == "decimalnumber") Nd();
    else if (k == "nl" || k == "letternumber" || k == "no" || k == "othernumber") {}
    else if (k == "p" || k == "punctuation") { Pc(); Pd(); Ps(); Pe(); Po(); }
    else if (k == "pc" || k == "connectorpunctuation") Pc();
    else if (k == "pd" || k == "dashpunctuation") Pd();
    else if (k == "ps" || k == "openpunctuation") Ps();
    else if (k == "pe" || k == "closepunctuation") Pe();
    else if (k == "pi" || k == "initialpunctuation" || k == "pf" || k == "finalpunctuation") {}
    else if (k == "po" || k == "otherpunctuation") Po();
    else if (k == "s" || k == "symbol") { Sm(); Sc(); Sk(); }
    else if (k == "sm" || k == "mathsymbol") Sm();
    else if (k == "sc" || k == "currencysymbol") Sc();
    else if (k == "sk" || k == "modifiersymbol") Sk();
    else if (k == "so" || k == "othersymbol") {}
    else if (k == "z" || k == "separator" || k == "zs" || k == "spaceseparator") Zs();
    else if (k == "zl" || k == "lineseparator" || k == "zp" || k == "paragraphseparator") {}
    else if (k == "c" || k == "other" || k == "cc" || k == "control") Cc();
    else if (k == "cf" || k == "format" || k == "cs" || k == "surrogate" || k == "co" || k == "privateuse" || k == "cn" || k == "unassigned") {}
    else return false;
    return true;
  }
  // `\p{..}` / `\P{..}` / `\pL` at p[i] (the 'p' / 'P'): the class's ASCII members into `s` (complemented for \P / {^..});
  // like the Perl classes, exact on ASCII subjects only
  bool unicode_class(ByteSet& s) {
    if (eof() || (p[i] != 'p' && p[i] != 'P')) return false;
    bool neg = p[i] == 'P';
    i++;
    std::string name;
    if (eof()) { fail("bad \\p escape"); return true; }
    if (p[i] == '{') {
      i++;
      if (!eof() && p[i] == '^') { neg = !neg; i++; }
      while (!eof() && p[i] != '}') name.push_back((char)p[i++]);
      if (eof()) { fail("bad \\p escape"); return true; }
      i++;
    } else name.push_back((char)p[i++]);
    const size_t eq = name.find('=');
    if (eq != std::string::npos) {   // gc=Lu / General_Category=Lu; any other property (scripts ...) needs the tables
      std::string prop;
      for (char ch : name.substr(0, eq)) if (ch != '_' && ch != ' ' && ch != '-') prop.push_back((char)(ch >= 'A' && ch <= 'Z' ? ch + 32 : ch));
      if (prop != "gc" && prop != "generalcategory") { fail("Unicode property other than the general category"); return true; }
      name = name.substr(eq + 1);
    }
    ByteSet m;
    if (!unicode_class_ascii(name, m)) { fail("Unicode class that needs the Unicode tables (script / property)"); return true; }
    ascii_only = true;
    if (f_i) for (unsigned c = 'a'; c <= 'z'; c++) if (m.has(c) || m.has(c - 32)) { m.add(c); m.add(c - 32); }   // regex-syntax folds the class, THEN negates it
    for (unsigned c = 0; c < 0x80; c++) if (m.has(c) != neg) s.add(c);
    return true;
  }

  // after '[': items and set operations up to the matching ']' (regex-syntax precedence: ranges, then union, then && -- ~~ left to right)
  bool parse_class_set(CSet& out) {
    bool neg = false;
    if (!eof() && p[i] == '^') { neg = true; i++; }
    CSet acc; bool have_acc = false; int pending_op = 0;   // 0 none, 1 &&, 2 --, 3 ~~
    CSet cur; bool first = true;
    auto close_operand = [&]() {
      if (f_i) cur.fold();
      if (!have_acc) { acc = cur; have_acc = true; }
      else if (pending_op == 1) acc.intersect(cur);
      else if (pending_op == 2) acc.subtract(cur);
      else acc.sym_diff(cur);
      cur = CSet();
    };
    for (;;) {
      if (eof()) return fail("unclosed class");
      unsigned c = p[i];
      if (c == ']' && !first) { i++; break; }
      if (c == '[') {
        if (i + 1 < n && p[i + 1] == ':') {          // POSIX class [:name:] / [:^name:]
          size_t j = i + 2; bool pneg = false;
          if (j < n && p[j] == '^') { pneg = true; j++; }
          std::string name;
          while (j < n && p[j] != ':' && p[j] != ']') name.push_back((char)p[j++]);
          if (j + 1 < n && p[j] == ':' && p[j + 1] == ']') {
            ByteSet m;
            if (!posix_class(name, m)) return fail("unknown POSIX class");
            CSet item; item.a = m;
            if (pneg) item.complement();
            cur.unite(item);
            i = j + 2; first = false;
            continue;
          }
        }
        i++;                                          // nested class
        CSet inner;
        if (!parse_class_set(inner)) return false;
        cur.unite(inner);
        first = false;
        continue;
      }
      if ((c == '&' || c == '-' || c == '~') && i + 1 < n && p[i + 1] == c && !(c == '-' && first)) {
        // `--` right before the closing bracket is two literal dashes in regex-syntax? It is a set operation with an empty right side there; refused here
        if (i + 2 < n && p[i + 2] == ']') return fail("set operation without a right operand");
        close_operand();
        pending_op = c == '&' ? 1 : c == '-' ? 2 : 3;
        i += 2; first = false;
        continue;
      }
      first = false;
      if (c >= 0x80) return fail("non-ASCII class member");
      i++;
      if (c == '\\') {
        if (perl_class(cur.a)) continue;              // `[\d_-]`: a class inside the class (no range from / to it)
        { const size_t before = i; if (unicode_class(cur.a)) { if (!err.empty()) return false; continue; } i = before; }
        if (!escape_byte(c)) return false;
      }
      unsigned hi = c;
      if (!eof() && p[i] == '-' && i + 1 < n && p[i + 1] != ']' && !(p[i + 1] == '-')) {   // range (a dash before `--` is not one)
        i++;
        hi = p[i];
        if (hi >= 0x80 || hi == '[') return fail("non-ASCII class member");
        i++;
        if (hi == '\\') { if (!escape_byte(hi)) return false; }
        if (hi < c) return fail("invalid class range");
      }
      cur.a.add_range(c, hi);
    }
    close_operand();
    if (f_i) acc.fold();
    if (neg) acc.complement();
    out = acc;
    return true;
  }
  NodeP parse_class() {   // after '['
    CSet c;
    if (!parse_class_set(c)) return nullptr;
    return cset_node(c);
  }

  NodeP parse_atom() {
    const unsigned c = p[i];
    if (c == '(') {
      i++;
      if (!eof() && p[i] == '?') {
        if (i + 1 < n && p[i + 1] == ':') i += 2;
        else if (i + 1 < n && (p[i + 1] == 'P' || p[i + 1] == '<')) {   // named group: the name is irrelevant for is_match
          i += p[i + 1] == 'P' ? 2 : 1;
          if (eof() || p[i] != '<') { fail("bad group"); return nullptr; }
          while (!eof() && p[i] != '>') i++;
          if (eof()) { fail("bad group name"); return nullptr; }
          i++;
        } else {   // inline flags: (?flags) for the rest of the enclosing group, (?flags:...) for this one
          i++;
          bool on = true, any = false;
          const bool s_i = f_i, s_s = f_s, s_x = f_x, s_m = f_m;
          for (;; i++) {
            if (eof()) { fail("unclosed flag group"); return nullptr; }
            const unsigned fc = p[i];
            if (fc == ')' || fc == ':') break;
            if (fc == '-') { if (!on) { fail("bad flag group"); return nullptr; } on = false; continue; }
            if (fc == 'i') f_i = on; else if (fc == 's') f_s = on; else if (fc == 'x') f_x = on; else if (fc == 'm') f_m = on;
            else if (fc == 'U') {}   // swap greediness: the same language
            else { fail("inline flag other than i m s x U"); return nullptr; }
            any = true;
          }
          if (!any) { fail("empty flag group"); return nullptr; }
          if (p[i] == ')') { i++; auto e = mk(EMPTY); e->set.add(3); return e; }   // (set bit 3: a flag directive, not an expression)
          i++;                                          // ':' — a group with its own flags
          NodeP inner = parse_alt();
          f_i = s_i; f_s = s_s; f_x = s_x; f_m = s_m;
          if (!inner) return nullptr;
          if (eof() || p[i] != ')') { fail("unclosed group"); return nullptr; }
          i++;
          return inner;
        }
      }
      const bool s_i = f_i, s_s = f_s, s_x = f_x, s_m = f_m;   // a (?flags) directive inside ends with its group
      NodeP inner = parse_alt();
      f_i = s_i; f_s = s_s; f_x = s_x; f_m = s_m;
      if (!inner) return nullptr;
      if (eof() || p[i] != ')') { fail("unclosed group"); return nullptr; }
      i++;
      return inner;
    }
    if (c == '[') { i++; return parse_class(); }
    if (c == '.') {
      i++;
      ByteSet s; s.add_range(0, 0x7F);
      if (!f_s) s.w[0] &= ~(1ull << '\n');
      return alt({leaf(s), non_ascii_char()});
    }
    if (c == '^') { i++; auto a = mk(A_START); if (f_m) a->set.add(2); return a; }   // set bit 2: under `m` (also matches after a line feed)
    if (c == '$') { i++; auto a = mk(A_END); if (f_m) a->set.add(2); return a; }
    if (c == '\\') {
      i++;
      if (!eof() && p[i] == 'A') { i++; auto a = mk(A_START); a->set.add(1); return a; }   // set bit 1: not affected by `m`
      if (!eof() && p[i] == 'z') { i++; auto a = mk(A_END); a->set.add(1); return a; }
      if (!eof() && (p[i] == 'b' || p[i] == 'B')) { const bool nb = p[i] == 'B'; i++; ascii_only = true; return mk(nb ? A_NWORDB : A_WORDB); }
      { ByteSet ps_; if (perl_class(ps_)) return ascii_set_node(ps_); }
      { ByteSet pu_; if (unicode_class(pu_)) { if (!err.empty()) return nullptr; return ascii_set_node(pu_); } }
      unsigned b;
      if (!escape_byte(b)) return nullptr;
      ByteSet s; s.add(b);
      return ascii_set_node(s);
    }
    if (c == '*' || c == '+' || c == '?' || c == '{') { fail("repetition operator without an expression"); return nullptr; }
    if (c == ')' || c == '|') { fail("unexpected character"); return nullptr; }
    size_t len;
    if (!utf8_len(i, len)) return nullptr;
    NodeP l = literal_at(i, len);
    i += len;
    return l;
  }

  static NodeP clone(const NodeP& a) {
    auto n = std::make_shared<Node>(*a);
    for (auto& k : n->kids) k = clone(k);
    return n;
  }
  bool parse_uint(unsigned& v) {
    if (eof() || p[i] < '0' || p[i] > '9') return false;
    v = 0;
    while (!eof() && p[i] >= '0' && p[i] <= '9') { v = v * 10 + (p[i] - '0'); if (v > 1000) return fail("repetition count too large"); i++; }
    return true;
  }
  NodeP parse_repeat() {
    NodeP a = parse_atom();
    if (!a) return nullptr;
    for (;;) {
      skip_x();
      if (eof()) break;
      const unsigned c = p[i];
      if (c != '*' && c != '+' && c != '?' && c != '{') break;
      if (a->kind == A_START || a->kind == A_END || a->kind == A_WORDB || a->kind == A_NWORDB) { fail("repeated anchor"); return nullptr; }
      if (c == '{') {
        i++;
        unsigned lo = 0, hi = 0; bool open = false;
        if (!parse_uint(lo)) { fail("bad repetition"); return nullptr; }
        if (!eof() && p[i] == ',') { i++; if (!parse_uint(hi)) { if (!err.empty()) return nullptr; open = true; } }
        else hi = lo;
        if (eof() || p[i] != '}') { fail("bad repetition"); return nullptr; }
        i++;
        if (!open && hi < lo) { fail("bad repetition range"); return nullptr; }
        std::vector<NodeP> k;
        for (unsigned r = 0; r < lo; r++) k.push_back(clone(a));
        if (open) k.push_back(un(STAR, clone(a)));
        else for (unsigned r = lo; r < hi; r++) k.push_back(un(OPT, clone(a)));
        if (k.size() > 64) { fail("more than 64 positions"); return nullptr; }
        a = cat(std::move(k));
      } else {
        i++;
        a = un(c == '*' ? STAR : c == '+' ? PLUS : OPT, a);
      }
      if (!eof() && p[i] == '?') i++;   // lazy: irrelevant for is_match
    }
    return a;
  }
  NodeP parse_cat() {
    std::vector<NodeP> k;
    for (;;) {
      skip_x();
      if (eof() || p[i] == '|' || p[i] == ')') break;
      NodeP r = parse_repeat();
      if (!r) return nullptr;
      if (r->kind == EMPTY && r->set.has(3)) continue;   // a (?flags) directive
      k.push_back(r);
    }
    return cat(std::move(k));
  }
  NodeP parse_alt() {
    std::vector<NodeP> k;
    for (;;) {
      NodeP c = parse_cat();
      if (!c) return nullptr;
      k.push_back(c);
      skip_x();
      if (!eof() && p[i] == '|') { i++; continue; }
      break;
    }
    return alt(std::move(k));
  }
};

// Glushkov sets with word-boundary conditions: index 0 = unconditional, 1 = only across a word boundary (`\b`), 2 = only
// where there is none (`\B`).  A condition on `first` holds at the point before the position's character, on `last` at the
// point after it, on `nullable` at the single point of the empty match.
struct Info { bool nullable[3]; uint64_t first[3], last[3]; };
inline int both(int a, int b) { return a == 0 ? b : b == 0 ? a : a == b ? a : -1; }   // -1: `\b` and `\B` at one point

struct Builder {
  RegexProg* out; uint32_t n_pos = 0; bool too_big = false; bool stray_anchor = false; bool assert_in_loop = false;
  uint64_t* fol(int c) { return c == 0 ? out->follow : c == 1 ? out->follow_b : out->follow_nb; }
  void edges(const Info& a, const Info& b) {   // every last of a -> every first of b, conditions combined
    for (int ca = 0; ca < 3; ca++) for (int cb = 0; cb < 3; cb++) {
      const int c = both(ca, cb);
      if (c < 0 || !a.last[ca] || !b.first[cb]) continue;
      for (uint64_t l = a.last[ca]; l; l &= l - 1) fol(c)[__builtin_ctzll(l)] |= b.first[cb];
    }
  }
  Info build(const NodeP& a) {
    Info z{{false, false, false}, {0, 0, 0}, {0, 0, 0}};
    switch (a->kind) {
      case EMPTY: z.nullable[0] = true; return z;
      case A_START: case A_END: stray_anchor = true; z.nullable[0] = true; return z;
      case A_WORDB: z.nullable[1] = true; return z;
      case A_NWORDB: z.nullable[2] = true; return z;
      case LEAF: {
        if (n_pos >= 64) { too_big = true; return z; }
        const uint32_t id = n_pos++;
        for (unsigned b = 0; b < 256; b++) if (a->set.has(b)) out->byte_mask[b] |= 1ull << id;
        z.first[0] = z.last[0] = 1ull << id;
        return z;
      }
      case CAT: {
        Info acc = z; acc.nullable[0] = true;
        for (const NodeP& k : a->kids) {
          const Info b = build(k);
          edges(acc, b);
          Info r = z;
          for (int c = 0; c < 3; c++) { r.first[c] = acc.first[c]; r.last[c] = b.last[c]; }
          for (int ca = 0; ca < 3; ca++) for (int cb = 0; cb < 3; cb++) {
            const int c = both(ca, cb);
            if (c < 0) continue;
            if (acc.nullable[ca]) r.first[c] |= b.first[cb];
            if (b.nullable[cb]) r.last[c] |= acc.last[ca];
            if (acc.nullable[ca] && b.nullable[cb]) r.nullable[c] = true;
          }
          acc = r;
        }
        return acc;
      }
      case ALT: {
        Info acc = z;
        for (const NodeP& k : a->kids) { const Info b = build(k); for (int c = 0; c < 3; c++) { acc.nullable[c] = acc.nullable[c] || b.nullable[c]; acc.first[c] |= b.first[c]; acc.last[c] |= b.last[c]; } }
        return acc;
      }
      default: {   // STAR, PLUS, OPT
        Info b = build(a->kids[0]);
        if (a->kind != OPT) {
          if (b.nullable[1] || b.nullable[2]) assert_in_loop = true;   // (\b|x)*: an iteration that is only an assertion
          edges(b, b);
        }
        if (a->kind != PLUS) b.nullable[0] = true;
        return b;
      }
    }
  }
};

}  // namespace regex_detail

enum RegexStatus { REGEX_OK = 0, REGEX_UNSUPPORTED = 1 };

// Compiles `pattern` with SPARQL `flags` (regex.rs:107-141).  REGEX_UNSUPPORTED + `why` for anything outside the
// subset above (which includes patterns the crate itself rejects: those would be the error value on every row,
// but telling them apart needs the crate's whole grammar, so they are refused as well).
inline RegexStatus regex_compile(const char* pattern, size_t n, const char* flags, size_t n_flags, RegexProg& out, std::string& why) {
  using namespace regex_detail;
  std::memset(&out, 0, sizeof out);
  bool q = false, multiline = false; (void)multiline;
  Parser ps{reinterpret_cast<const unsigned char*>(pattern), n};
  for (size_t k = 0; k < n_flags; k++) {
    switch (flags[k]) {
      case 's': ps.f_s = true; break;
      case 'm': multiline = true; ps.f_m = true; break;
      case 'i': ps.f_i = true; break;
      case 'x': ps.f_x = true; break;
      case 'q': q = true; break;
      default: out.always_error = 1; return REGEX_OK;   // invalid option => error on every row (regex.rs:137)
    }
  }
  NodeP root;
  if (q) {   // regex::escape: the whole pattern is a literal (the `x` flag then has nothing left to strip but
             // whitespace, which `regex::escape` does not protect: refuse that corner)
    if (ps.f_x) { why = "flags q and x together"; return REGEX_UNSUPPORTED; }
    std::vector<NodeP> k;
    size_t at = 0;
    while (at < n) {
      size_t len;
      if (!ps.utf8_len(at, len)) { why = ps.err; return REGEX_UNSUPPORTED; }
      NodeP l = ps.literal_at(at, len);
      if (!l) { why = ps.err; return REGEX_UNSUPPORTED; }
      k.push_back(l);
      at += len;
    }
    root = cat(std::move(k));
  } else {
    root = ps.parse_alt();
    if (root && !ps.eof()) { ps.fail("unbalanced parenthesis"); root = nullptr; }
    if (!root) { why = ps.err; return REGEX_UNSUPPORTED; }
  }
  // anchors: only the first / last element of a top-level concatenation
  if (root->kind == A_START || root->kind == A_END) { auto c = mk(CAT); c->kids.push_back(root); root = c; }
  if (root->kind == CAT) {
    if (!root->kids.empty() && root->kids.front()->kind == A_START) {
      out.anchor_start = 1; out.ml_start = root->kids.front()->set.has(2) && !root->kids.front()->set.has(1);
      root->kids.erase(root->kids.begin());
    }
    if (!root->kids.empty() && root->kids.back()->kind == A_END) {
      out.anchor_end = 1; out.ml_end = root->kids.back()->set.has(2) && !root->kids.back()->set.has(1);
      root->kids.pop_back();
    }
  }
  Builder b{&out};
  const Info info = b.build(root);
  if (b.stray_anchor) { why = "anchor that is not the first / last element of the pattern"; return REGEX_UNSUPPORTED; }
  if (b.too_big) { why = "more than 64 positions"; return REGEX_UNSUPPORTED; }
  if (b.assert_in_loop) { why = "word boundary under a repetition"; return REGEX_UNSUPPORTED; }
  out.first = info.first[0]; out.last = info.last[0]; out.nullable = info.nullable[0] ? 1 : 0; out.n_pos = b.n_pos;
  out.first_b = info.first[1]; out.first_nb = info.first[2]; out.last_b = info.last[1]; out.last_nb = info.last[2];
  out.nullable_b = info.nullable[1] ? 1 : 0; out.nullable_nb = info.nullable[2] ? 1 : 0;
  out.ascii_only = ps.ascii_only ? 1 : 0;
  uint64_t any = out.first_b | out.first_nb | out.last_b | out.last_nb;
  for (int k = 0; k < 64; k++) any |= out.follow_b[k] | out.follow_nb[k];
  out.has_assert = (any || out.nullable_b || out.nullable_nb) ? 1 : 0;
  return REGEX_OK;
}

}  // namespace rdfgpu
