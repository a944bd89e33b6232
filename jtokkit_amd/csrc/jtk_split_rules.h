// jtk_split_rules.h -- "does a pre-token piece start at byte p?" as a function of p's neighbourhood.
//
// The reference walks a java.util.regex Matcher over the text (GptBytePairEncoding.java:77-80) with
// the patterns of EncodingFactory.java:63 (r50k/p50k/p50k_edit) and :105 (cl100k), compiled with
// UNICODE_CHARACTER_CLASS (:129).  find() yields consecutive, gap-free matches; alternatives are
// tried leftmost-first with greedy, backtracking quantifiers.  Because every alternative is a run
// of ONE character class (plus a one-character prefix / CR-LF suffix), whether a match starts at a
// given character is decided by the class of that character, of one or two characters before it,
// and by three properties of the run it sits in.  That makes the split data parallel: one lane per
// byte evaluates jtk_is_piece_start() below, no sequential matcher state.
//
// Classes: L = \p{L}, N = \p{N}, W = \s, O = everything else.  "ms(i)" = a match starts at i.
//
//  both patterns
//   - the first character of a document starts a match.
//   - O at i:  ms(i) <=> previous char is not O and is not U+0020          (" ?O+" glues one space;
//              an O-run is never split: "!!abc" -> "!!", "abc")
//   - contraction: an apostrophe with ms(') followed by s|t|m|d|re|ve|ll (cl100k: case-insensitive,
//              incl. U+017F for s) is a piece of its own, so a match also starts right after it
//              when a letter follows ("'sat" -> "'s", "at").
//  r50k family
//   - L/N run start at i: ms(i) <=> previous char is not U+0020 (and i is not the letter of a
//              contraction); never inside a run (except after a contraction).
//   - W at i:  ms(i) <=> i starts its whitespace run, or i is the last char of the run and the run
//              is followed by a non-whitespace char ("\s+(?!\S)" gives back exactly one char).
//  cl100k
//   - L run start at i: previous N or CR/LF -> ms; previous other whitespace -> glued (not ms);
//              previous O char c -> ms(i) = !ms(c)    ("[^\r\n\p{L}\p{N}]?\p{L}+")
//   - N at i:  ms(i) <=> (number of N chars before i in its run) % 3 == 0          ("\p{N}{1,3}")
//   - W at i, run [a,e), P = class before a, X = class at e (EOT at document end):
//        leading CR/LF of a run that follows an O char belong to the O piece (" ?O+[\r\n]*");
//        s = first char not swallowed that way;  k = last CR/LF in [s,e);  t = k+1 (or s);
//        matches start at s, at t ("\s*[\r\n]+" ends on the LAST CR/LF), and at the last char of
//        the run when X != EOT and it is not a CR/LF ("\s+(?!\S)" / glue to the next piece).
#ifndef JTK_SPLIT_RULES_H
#define JTK_SPLIT_RULES_H

#include "jtk_common.h"

// Win must provide:  idx_t (integer type of positions), kMaxWalk (0 = unbounded run walks),
//                    uint32_t cb(idx_t p)  (class byte, JTK_CB_DS set for p >= n_bytes) and uint32_t byte(idx_t p).
// cb(p-1) is only ever read when !(cb(p) & JTK_CB_DS), i.e. p-1 lies in the same document.
// With kMaxWalk > 0 a run walk longer than that sets `unresolved` (the caller re-evaluates the
// position with an unbounded window); every other access stays within 8 bytes of p.

template <class Win>
JTK_HD bool jtk_other_is_match_start(const Win& w, typename Win::idx_t j) {
    const uint32_t c = w.cb(j);
    if (c & JTK_CB_DS) return true;
    const uint32_t pb = w.cb(j - 1);
    return (pb & JTK_CB_CLS) != JTK_CLS_O && !(pb & JTK_CB_SP);
}

JTK_HD uint32_t jtk_fold_ascii(uint32_t b, bool ci) {
    return (ci && (b - 'A') < 26u) ? (b | 0x20u) : b;
}

// Length in bytes (2 or 3) of the contraction alternative matching at apostrophe position j, else 0.
// Bytes after j must belong to the same document (checked through the DS flag).
template <class Win>
JTK_HD int jtk_contraction_len(const Win& w, typename Win::idx_t j, bool ci) {
    if (w.byte(j) != '\'') return 0;
    if (w.cb(j + 1) & JTK_CB_DS) return 0;
    const uint32_t raw1 = w.byte(j + 1);
    const uint32_t c1 = jtk_fold_ascii(raw1, ci);
    if (c1 == 's' || c1 == 't' || c1 == 'm' || c1 == 'd') return 2;
    if (w.cb(j + 2) & JTK_CB_DS) return 0;
    const uint32_t raw2 = w.byte(j + 2);
    const uint32_t c2 = jtk_fold_ascii(raw2, ci);
    if ((c1 == 'r' || c1 == 'v') && c2 == 'e') return 3;
    if (c1 == 'l' && c2 == 'l') return 3;
    if (ci && raw1 == 0xC5u && raw2 == 0xBFu) return 3;      // U+017F folds to 's' under UNICODE_CASE
    return 0;
}

template <int KIND, class Win>
JTK_HD bool jtk_is_piece_start_t(const Win& w, typename Win::idx_t p, bool& unresolved) {
    typedef typename Win::idx_t idx_t;
    const uint32_t c = w.cb(p);
    if (c & JTK_CB_CONT) return false;
    if (c & JTK_CB_DS) return true;
    constexpr bool cl = (KIND == JTK_PAT_CL100K);
    const uint32_t pb = w.cb(p - 1);
    const uint32_t cls = c & JTK_CB_CLS, pc = pb & JTK_CB_CLS;

    if (cls == JTK_CLS_O) return pc != JTK_CLS_O && !(pb & JTK_CB_SP);

    if (cls == JTK_CLS_L) {
        if (pc == JTK_CLS_L) {
            // inside a letter run: only the end of a contraction piece starts a match here
            if (pb & JTK_CB_DS) return false;
            if (jtk_contraction_len(w, p - 2, cl) == 2 && jtk_other_is_match_start(w, p - 2)) return true;
            if (!(w.cb(p - 2) & JTK_CB_DS) && jtk_contraction_len(w, p - 3, cl) == 3
                && jtk_other_is_match_start(w, p - 3)) return true;
            return false;
        }
        if (pc == JTK_CLS_O) {
            idx_t j = p - 1;                                   // lead byte of the previous (O) char
            if (w.cb(j) & JTK_CB_CONT) { j--; if (w.cb(j) & JTK_CB_CONT) { j--; if (w.cb(j) & JTK_CB_CONT) j--; } }
            const bool prev_ms = jtk_other_is_match_start(w, j);
            if (cl) return !prev_ms;                           // glued as the one-char prefix, or contraction
            return !(prev_ms && jtk_contraction_len(w, j, false) != 0);
        }
        if (cl) return pc == JTK_CLS_N || (pb & JTK_CB_NL);   // W: only CR/LF cannot be the prefix
        return !(pb & JTK_CB_SP);
    }

    if (cls == JTK_CLS_N) {
        if (!cl) return pc != JTK_CLS_N && !(pb & JTK_CB_SP);
        uint32_t cnt = 0;                                      // N chars before p in this run
        idx_t k = p;
        int steps = 0;
        while (!(w.cb(k) & JTK_CB_DS) && (w.cb(k - 1) & JTK_CB_CLS) == JTK_CLS_N) {
            k--;
            cnt += (w.cb(k) & JTK_CB_CONT) ? 0u : 1u;
            if (Win::kMaxWalk && ++steps >= Win::kMaxWalk) { unresolved = true; return false; }
        }
        return cnt % 3u == 0u;
    }

    // ---- whitespace ----
    idx_t nx = p + 1;                                          // first byte after this char
    while ((w.cb(nx) & (JTK_CB_CONT | JTK_CB_DS)) == JTK_CB_CONT && nx < p + 4) nx++;
    const uint32_t nb = w.cb(nx);
    const bool last_in_run = (nb & JTK_CB_DS) || (nb & JTK_CB_CLS) != JTK_CLS_W;
    const bool followed_by_text = last_in_run && !(nb & JTK_CB_DS);
    const bool run_start = pc != JTK_CLS_W;
    if (!cl) return run_start || followed_by_text;

    // walk back over the CR/LF bytes directly before p
    idx_t k = p;
    int steps = 0;
    while (!(w.cb(k) & JTK_CB_DS) && (w.cb(k - 1) & JTK_CB_NL)) {
        k--;
        if (Win::kMaxWalk && ++steps >= Win::kMaxWalk) { unresolved = true; return false; }
    }
    bool at_run_start, after_other;
    if (w.cb(k) & JTK_CB_DS) { at_run_start = true; after_other = false; }
    else {
        const uint32_t q = w.cb(k - 1) & JTK_CB_CLS;
        at_run_start = q != JTK_CLS_W;
        after_other = q == JTK_CLS_O;
    }
    const bool swallowed_prefix = at_run_start && after_other;   // [a,p) is all CR/LF and follows an O char
    const bool is_nl = (c & JTK_CB_NL) != 0;
    if (swallowed_prefix && is_nl) return false;                  // part of the O piece's [\r\n]*
    if (swallowed_prefix) return true;                            // p == s
    if (k == p && at_run_start) return true;                      // p == a == s
    if (k != p) {
        // previous byte is a CR/LF of this run at or after s: p == t iff no CR/LF remains in [p,e)
        idx_t q = p;
        steps = 0;
        for (;;) {
            const uint32_t cq = w.cb(q);
            if ((cq & JTK_CB_DS) && q != p) return true;
            if ((cq & JTK_CB_CLS) != JTK_CLS_W) return true;
            if (cq & JTK_CB_NL) return false;
            q++;
            if (Win::kMaxWalk && ++steps >= Win::kMaxWalk) { unresolved = true; return false; }
        }
    }
    return followed_by_text && !is_nl;
}

template <class Win>
JTK_HD bool jtk_is_piece_start(const Win& w, typename Win::idx_t p, int kind) {
    bool unresolved = false;
    return kind == JTK_PAT_CL100K ? jtk_is_piece_start_t<JTK_PAT_CL100K>(w, p, unresolved)
                                  : jtk_is_piece_start_t<JTK_PAT_R50K>(w, p, unresolved);
}

#endif
