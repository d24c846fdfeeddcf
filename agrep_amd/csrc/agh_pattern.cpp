// agh_pattern.cpp -- the pattern compiler of libagrep_hip.so: agrep's non-regex pattern language to the
// tables the device engines consume.  Takes the place of preprocess() (preproce.c:137-332: delimiter +
// separator prefix, meta characters, -w / -x wrapping) and maskgen() (maskgen.c:68-266: one byte class per
// position, Mask[256], Init[0], Init1, NO_ERR_MASK, endposition, D_endpos) for callers that do not link the
// reference's front end -- with it, agh_query_from_maskgen's input can be produced by the library itself.
// Host-only code: no HIP call, usable (and tested: tests/test_pattern_compiler.py against the reference's own
// tables in tests/golden/) without a GPU.
//
// Language (docs of the reference; preproce.c:238-332 is where it is decided):
//   c        the byte c           \c    the byte c, whatever it is
//   [a-fxyz] one of these bytes   [^...] any byte but these      .   any byte but newline
//   #        any run of bytes (zero or more): the position in front of it stays reachable once reached
//   <...>    no error may touch these positions     ^  $   a newline (an anchor), never touched by an error
//   a;b      AND: both parts in the record          a,b    OR: one of the parts
//   -w / -x  (qflags) a non-alphanumeric byte / a record boundary on both sides, never touched by an error
// Regular expressions ( * | ( ) ) belong to another automaton (re(), agrep.c:468-1917: out of scope) and are
// refused, as are unescaped meta characters inside [] (the reference turns them into its internal symbol
// codes 129..145 there, i.e. into classes of bytes nobody meant).
//
// Layout of the result (maskgen.c:218-257, kept so that the tables are interchangeable with the reference's):
// positions 1..M = delimiter bytes, one separator, the pattern; position p lives at bit M - p of a 32-bit
// word; M <= 31 (maskgen.c:201-208), i.e. 29 - |delimiter| pattern positions.
#include <errno.h>
#include <string.h>

#include <vector>

#include "agh_internal.h"

namespace {

struct byte_class {
    uint32_t w[8];
    byte_class() { memset(w, 0, sizeof(w)); }
    void add(unsigned c) { w[c >> 5] |= 1u << (c & 31u); }
    void add_range(unsigned lo, unsigned hi) { for (unsigned c = lo; c <= hi && c < 256u; ++c) add(c); }
    bool has(unsigned c) const { return (w[c >> 5] >> (c & 31u)) & 1u; }
    void invert() { for (uint32_t &x : w) x = ~x; }
};

struct position {
    byte_class cls;
    bool no_error = false;      // an error transition may not enter this position (maskgen.c:80-95, 171-193)
    bool separator = false;     // ';' / ',' / the one behind the delimiter: empty class, an end position
};

// internal symbol the reference feeds for "-x: newline in front" (agrep.h NNLINE = 131): the class holds the
// newline and that byte, and so do the reference's tables
const unsigned NNLINE_BYTE = 131;

bool is_upper(unsigned c) { return c >= 'A' && c <= 'Z'; }

struct compiler {
    std::vector<position> pos;  // pos[0] unused: positions count from 1 as in maskgen.c
    uint32_t wild = 0, ends = 0, noerr = 0;     // in Bit[] space: position j <-> bit 32 - j
    bool and_seen = false, or_seen = false, fancy = false;
    int angle = 0;              // '<' minus '>' so far (maskgen.c's EVEN): has to end at 0, may not go below
    bool exact = false;         // maskgen.c's No_error: ON behind a '<', OFF behind a '>' -- not a depth: <<ab>c> leaves c free
    bool nocase = false;

    static uint32_t bit(int j) { return j >= 1 && j <= 32 ? 1u << (32 - j) : 0u; }
    int next() const { return (int)pos.size(); }        // index the next position will get

    int add(const byte_class &c, bool no_error, bool sep = false)
    {
        // M <= 31 positions (delimiter + separator + pattern): maskgen.c:201-208 tests j > WORD after its j++
        if (next() > 31) return fail("pattern too long (has > 31 positions with its delimiter)");
        position p;
        p.cls = c;
        p.no_error = no_error;
        p.separator = sep;
        if (no_error) noerr |= bit(next());
        if (sep) ends |= bit(next());
        pos.push_back(p);
        return 0;
    }
    int add_byte(unsigned c, bool no_error)
    {
        byte_class k;
        if (nocase && is_upper(c)) c += 32;             // maskgen.c:52-59; the upper-case rows are aliased at the end
        k.add(c);
        return add(k, no_error || c == '\n');           // a newline position is never entered by an error (:171-175)
    }
    int add_word_boundary()                              // maskgen.c:176-187: the non-alphanumeric ASCII bytes
    {
        byte_class k;
        k.add_range(1, 47);
        k.add_range(58, 64);
        k.add_range(91, 96);
        k.add_range(123, 127);
        return add(k, true);
    }
};

int parse_class(compiler &C, const unsigned char *p, int len, int &i)
{
    // p[i] is the byte behind '['
    byte_class k;
    bool complement = false;
    if (i < len && p[i] == '^') { complement = true; ++i; }
    bool closed = false;
    const char *const unmatched = "unmatched '[', ']' (use \\[, \\] to search for [, ])";
    // A '-' that is not between two bytes of its own -- first or last in the class, or behind a finished range
    // ([a-c-e]) -- is read by the reference in ways nobody would guess ([a-e-c] is a..c, [-a] is empty, [a-] is
    // "unmatched"): refused here, so that whatever this compiler accepts it compiles like the reference.
    const char *const stray = "'-' inside [] that is not between two bytes must be written \\-";
    while (i < len) {
        unsigned c = p[i];
        if (c == ']') { closed = true; ++i; break; }
        if (c == '-') return fail(stray);
        if (c == '\\') {
            if (i + 1 >= len) return fail(unmatched);
            c = p[i + 1];
            i += 2;
        } else {
            if (strchr(".#,;*|()<>^${}~", (int)c))
                return fail("'%c' inside [] must be written \\%c", (int)c, (int)c);
            ++i;
        }
        if (C.nocase && is_upper(c)) c += 32;
        // lo-hi: the bytes from lo to hi and nothing else -- a range that runs backwards is empty, its first
        // byte included (maskgen.c:246-250 tests lo <= c <= hi)
        if (i < len && p[i] == '-') {
            if (i + 1 >= len || p[i + 1] == ']' || p[i + 1] == '-') return fail(stray);
            unsigned hi = p[i + 1];
            i += 2;
            if (hi == '\\') {
                if (i >= len) return fail(unmatched);
                hi = p[i++];
            }
            if (C.nocase && is_upper(hi)) hi += 32;
            k.add_range(c, hi);
            if (i < len && p[i] == '-') return fail(stray);
            continue;
        }
        k.add(c);
    }
    if (!closed) return fail(unmatched);
    if (complement) k.invert();                          // over all 256 bytes, the newline included (maskgen.c:252)
    C.fancy = true;
    return C.add(k, C.exact);
}

}   // namespace

extern "C" int agh_compile_pattern(const unsigned char *pat, int len, unsigned qflags, const unsigned char *delim,
                                   int dlen, agh_pattern_tables *out)
{
    if (!pat || !out || len < 1) return fail("empty pattern");
    if (!delim || dlen < 1 || dlen > AGH_MAX_DELIM) return fail("delimiter length %d outside 1..%d", dlen, AGH_MAX_DELIM);
    if ((qflags & AGH_Q_WORD) && (qflags & AGH_Q_WHOLELINE)) return fail("illegal option combination (-x and -w)");
    if ((qflags & AGH_Q_WHOLELINE) && !(dlen == 1 && delim[0] == '\n'))
        return fail("-d and -x are not compatible");                      // compat.c:89-96
    compiler C;
    C.nocase = (qflags & AGH_Q_NOCASE) != 0;
    C.pos.emplace_back();                               // (index 0)
    // the delimiter (agrep.c:2265-2316 wraps it into '<' '>': no error inside) and its separator
    for (int i = 0; i < dlen; ++i)
        if (C.add_byte(delim[i], true)) return -1;
    if (C.add(byte_class(), false, true)) return -1;
    // -x / -w in front (preproce.c:148-166)
    if (qflags & AGH_Q_WHOLELINE) {
        byte_class k;
        k.add('\n');
        k.add(NNLINE_BYTE);
        if (C.add(k, true)) return -1;
    } else if ((qflags & AGH_Q_WORD) && C.add_word_boundary()) {
        return -1;
    }
    const int first_pattern_pos = C.next();
    for (int i = 0; i < len;) {
        const unsigned c = pat[i];
        switch (c) {
        case '\\':
            if (i + 1 >= len) return fail("a pattern cannot end in a backslash");
            if (pat[i + 1] == '\n') C.fancy = true;     // (not a plain literal: the literal builder would let an error touch it)
            if (C.add_byte(pat[i + 1], C.exact)) return -1;
            i += 2;
            break;
        case '#':                                       // maskgen.c:68-78: the position in front becomes sticky
            C.wild |= compiler::bit(C.next() - 1);
            C.fancy = true;
            ++i;
            break;
        case '[':
            ++i;
            if (parse_class(C, pat, len, i)) return -1;
            break;
        case ']':
            return fail("unmatched '[', ']' (use \\[, \\] to search for [, ])");
        case '<':
            ++C.angle;
            C.exact = true;
            C.fancy = true;
            ++i;
            break;
        case '>':
            if (--C.angle < 0) return fail("unmatched '<', '>' (use \\<, \\> to search for <, >)");
            C.exact = false;
            ++i;
            break;
        case ';':                                       // maskgen.c:150-163
            if (C.or_seen) return fail("illegal pattern: cannot handle AND (';') and OR (',') simultaneously");
            C.and_seen = true;
            C.fancy = true;
            if (C.add(byte_class(), false, true)) return -1;
            ++i;
            break;
        case ',':                                       // maskgen.c:136-149
            if (C.and_seen) return fail("illegal pattern: cannot handle OR (',') and AND (';') simultaneously");
            C.or_seen = true;
            C.fancy = true;
            if (C.add(byte_class(), false, true)) return -1;
            ++i;
            break;
        case '^':
            // preproce.c:282 takes a '^' behind ANY '[' for the class complement, an escaped \[ included: what the
            // reference makes of "\[^x" is not a '[' followed by an anchor -- refused like its other surprises
            if (i > 0 && pat[i - 1] == '[') return fail("'^' directly behind \\[ is read as a class complement by the reference: write \\[\\^ or move it");
            /* fall through */
        case '$':                                       // preproce.c:284-293: an anchor is a newline position
            C.fancy = true;
            if (C.add_byte('\n', true)) return -1;
            ++i;
            break;
        case '.': {                                     // any byte but the newline (maskgen.c:243-247, no regex)
            byte_class k;
            k.add_range(0, 255);
            k.w['\n' >> 5] &= ~(1u << ('\n' & 31));
            C.fancy = true;
            if (C.add(k, C.exact)) return -1;
            ++i;
            break;
        }
        case '*':
        case '|':
        case '(':
        case ')':
            return fail("'%c': regular expressions are outside the scan path of this library (escape it as \\%c for "
                        "the byte itself)", (int)c, (int)c);
        case '{':
        case '}':
        case '~':                               // asplit.c:139-262: grouping and NOT of the boolean-pattern parser
            return fail("'%c': boolean pattern expressions are outside the scan path of this library (escape it as "
                        "\\%c for the byte itself)", (int)c, (int)c);
        default:
            if (c == '\n') C.fancy = true;              // a newline byte in the pattern: a no-error position, like ^ and $
            if (C.add_byte(c, C.exact)) return -1;
            ++i;
            break;
        }
    }
    if (C.angle != 0) return fail("unmatched '<', '>' (use \\<, \\> to search for <, >)");
    if (C.next() == first_pattern_pos) return fail("empty pattern");
    // -x / -w behind (preproce.c:157-175)
    if (qflags & AGH_Q_WHOLELINE) {
        if (C.add_byte('\n', true)) return -1;
    } else if ((qflags & AGH_Q_WORD) && C.add_word_boundary()) {
        return -1;
    }
    // ---- the tables, in maskgen's layout (maskgen.c:218-266) -----------------------------------------
    const int M = C.next() - 1, base = 32 - M, D_length = dlen + 1;
    memset(out, 0, sizeof(*out));
    out->M = M;
    out->AND = C.and_seen ? 1 : 0;
    out->simple = C.fancy ? 0 : 1;
    uint32_t wildmask = C.wild >> base, endposition = C.ends >> base;
    uint32_t no_err = (C.noerr >> 1) & ~compiler::bit(1);
    // maskgen.c:19 declares NO_ERR_MASK with implicit int, so the reference's shift at :223 is arithmetic: the
    // bits above the pattern come out set (asearch only ever ANDs the word with states below them)
    no_err = base >= 1 ? (uint32_t)((int32_t)~no_err >> (base - 1)) : ~no_err;     // (base == 0: M = 32)
    uint32_t init0 = endposition;
    for (int i = 1; i <= base; ++i) init0 |= compiler::bit(i);
    endposition = (endposition << 1) + 1u;
    out->Init0 = init0;
    out->Init1 = init0 | wildmask | endposition;
    out->D_endpos = (endposition >> (M - D_length)) << (M - D_length);
    out->endposition = endposition ^ out->D_endpos;
    out->NO_ERR_MASK = no_err;
    out->wildmask = wildmask;
    for (int k = 1; k <= M; ++k) {
        const uint32_t b = compiler::bit(base + k);
        for (unsigned c = 0; c < 256u; ++c)
            if (C.pos[(size_t)k].cls.has(c)) out->Mask[c] |= b;
    }
    if (C.nocase)                                       // maskgen.c:259-266
        for (unsigned c = 'A'; c <= 'Z'; ++c) out->Mask[c] = out->Mask[c + 32];
    return 0;
}

extern "C" agh_query *agh_query_pattern(const unsigned char *pat, int len, int D, unsigned qflags,
                                        const unsigned char *delim, int dlen)
{
    agh_pattern_tables t;
    if (agh_compile_pattern(pat, len, qflags, delim, dlen, &t)) return nullptr;
    // A plain literal goes to the literal builder (sample filter).  With a -w / -x guard only without errors: there
    // the reference takes its simple-pattern engines (also for patterns with \\c or '-': probed, `agrep -w w-l`
    // matches "\xc3\xa9w-l"), whose word test is !isalnum() over all 256 bytes (bm(), sgrep.c:750-756) -- what
    // agh_query_literal_ex implements.  With errors the reference goes through maskgen() (checksg.c:133-134), whose
    // word-boundary class is 1..47, 58..64, 91..96, 123..127 (maskgen.c:176-187: 0x00 and 0x80..0xFF are NOT
    // boundaries, "\xc3\xa9car" is no word "car"): those queries take the compiled tables, which are maskgen's bit
    // for bit.  (-n sends the reference through maskgen() at k = 0 as well; a caller that wants that reading of -w
    // next to bytes >= 0x80 passes the tables of agh_compile_pattern to agh_query_from_maskgen itself.)
    const bool guarded = (qflags & (AGH_Q_WORD | AGH_Q_WHOLELINE)) != 0;
    if (t.simple && (!guarded || D == 0)) {
        std::vector<unsigned char> lit;
        for (int i = 0; i < len; ++i) {
            if (pat[i] == '\\' && i + 1 < len) ++i;
            lit.push_back(pat[i]);
        }
        return agh_query_literal_ex(lit.data(), (int)lit.size(), D, qflags, delim, dlen);
    }
    // (the delimiter as asearch.c:54 wants it: the bytes themselves; lower-cased under -i like the tables)
    unsigned char d[AGH_MAX_DELIM];
    for (int i = 0; i < dlen; ++i) d[i] = ((qflags & AGH_Q_NOCASE) && is_upper(delim[i])) ? (unsigned char)(delim[i] + 32) : delim[i];
    return agh_query_from_maskgen(t.Mask, t.Init0, t.Init1, t.NO_ERR_MASK, t.endposition, t.D_endpos, t.M, d, dlen, D,
                                  t.AND);
}
