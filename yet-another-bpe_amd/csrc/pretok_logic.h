// pretok_logic.h -- per-position rules of the pre-tokeniser, shared by the HIP kernels (yabpe_pretok_kernels.h) and by
// the CPU unit-test model (tests/hostmodel/pretok_model.cpp).
//
// What is reproduced: reference trainer.py:136-214 -- every chunk of a file is decoded as UTF-8 and split with
//     sp_1|...|sp_k|'(?:[sdmt]|ll|ve|re)| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+          (:163-167)
// by regex.findall: leftmost match at the current position, alternatives tried in order, then continue after the match.
// Every character is matched by some alternative, so the pre-tokens PARTITION the chunk: the result is fully described
// by one flag per byte, "a pre-token starts here".  The rules below compute that flag for every position independently:
//
//   classes   L = \p{L}, N = \p{N}, S = \s, O = other (table generated from the regex module, unicode_classes.inc)
//   runs      a non-space character continues the token of the character before it iff both have the same class;
//             a single U+0020 directly before a non-space run belongs to that run (" ?" of the three run alternatives)
//   spaces    a whitespace run is one token (\s+(?!\S) backs off one character when a non-space follows): its first
//             character starts a token, and so does its LAST character when the run is longer than one character and a
//             non-space character follows (that last character is then either the " " prefix of the next run or a
//             one-character \s+ token)
//   'x        an apostrophe AT A TOKEN START followed by s d m t ll ve re is a token of its own (alternative 1 comes first);
//             the character after it starts a token whatever its class.  An apostrophe is at a token start iff the
//             character before it is a letter, a number or a whitespace other than U+0020 (or the text starts there):
//             this never depends on another contraction, so contractions are decided independently too
//   specials  literal strings, tried first, in configuration order, but only where a token starts (findall never
//             looks inside a match).  Whether a token starts at an occurrence can depend on the occurrences just before
//             it (overlap, or a match ending right there), so occurrences closer than (length + 3) bytes form a CHAIN
//             that is resolved left to right by the thread of its first occurrence (pt_special_walk); chains are
//             independent of each other.  The +3 is the reach of a contraction that a special's end can enable.
//
// Positions are BYTE offsets into the UTF-8 text; only lead bytes can carry a flag.
#pragma once
#include <stdint.h>

#include "tile_logic.h" // YB_HD

enum : uint8_t {
    PT_L = 0,
    PT_N = 1,
    PT_S = 2,
    PT_O = 3,
    PT_CLS = 3,    // mask of the class bits
    PT_CONT = 4,   // UTF-8 continuation byte (not a character start)
    PT_CHUNK0 = 8, // first byte of a chunk: every chunk is a text of its own (trainer.py:172-198)
};

struct PtView {
    const uint8_t *text; // text[j - org] = byte j of the corpus (org = 0: the arrays are the whole corpus; the fused kernel
    const uint8_t *meta; // works on a window staged in LDS and sets org to the window's first position)
    uint64_t n;          // length of the corpus (positions >= n do not exist)
    uint64_t org;
    YB_HD uint8_t T(uint64_t j) const { return text[j - org]; }
    YB_HD uint8_t M(uint64_t j) const { return meta[j - org]; } // per byte: class | PT_CONT | PT_CHUNK0
};

// ---------------------------------------------------------------- UTF-8 (Python's strict decoder)
YB_HD int pt_lead_len(uint8_t b) {
    if (b < 0x80) return 1;
    if (b >= 0xC2 && b <= 0xDF) return 2;
    if (b >= 0xE0 && b <= 0xEF) return 3;
    if (b >= 0xF0 && b <= 0xF4) return 4;
    return 0; // continuation byte, overlong lead (C0, C1) or > U+10FFFF (F5..FF)
}

// Decodes the character whose lead byte is at i; `end` = end of the chunk.  Returns its length, or 0 when the sequence is
// malformed (truncated, bad continuation byte, overlong form, surrogate, > U+10FFFF): UnicodeDecodeError.start == i.
YB_HD int pt_decode(const PtView &v, uint64_t i, uint64_t end, uint32_t *cp) {
    const uint8_t b0 = v.T(i);
    const int len = pt_lead_len(b0);
    if (len == 0 || i + (uint64_t)len > end) return 0;
    if (len == 1) {
        *cp = b0;
        return 1;
    }
    const uint8_t b1 = v.T(i + 1);
    uint8_t lo = 0x80, hi = 0xBF;
    if (b0 == 0xE0) lo = 0xA0;       // no overlong 3-byte forms
    else if (b0 == 0xED) hi = 0x9F;  // no surrogates
    else if (b0 == 0xF0) lo = 0x90;  // no overlong 4-byte forms
    else if (b0 == 0xF4) hi = 0x8F;  // <= U+10FFFF
    if (b1 < lo || b1 > hi) return 0;
    uint32_t c = len == 2 ? (b0 & 0x1Fu) : len == 3 ? (b0 & 0x0Fu) : (b0 & 0x07u);
    c = (c << 6) | (b1 & 0x3Fu);
    for (int k = 2; k < len; ++k) {
        const uint8_t bk = v.T(i + k);
        if ((bk & 0xC0) != 0x80) return 0;
        c = (c << 6) | (bk & 0x3Fu);
    }
    *cp = c;
    return len;
}

// First pass, one call per byte: the meta byte (class / continuation), and whether Python's decoder would stop HERE
// (the smallest such position over the text is UnicodeDecodeError.start).  v.meta holds only the PT_CHUNK0 marks at this
// point.  chunk_end: end of this byte's chunk when it is closer than 4 bytes, else any value >= i + 4.
YB_HD uint8_t pt_classify(const PtView &v, uint64_t i, uint64_t chunk_end, const uint8_t *cls_table, bool *bad) {
    const uint8_t b = v.T(i);
    *bad = false;
    if ((b & 0xC0) == 0x80) {
        // a continuation byte is fine iff a lead byte at most 3 bytes back (inside this chunk) announces a sequence
        // that reaches it; a malformed sequence is reported by its lead byte (a smaller position)
        bool stray = true;
        if (!(v.M(i) & PT_CHUNK0)) {
            for (int d = 1; d <= 3 && (uint64_t)d <= i; ++d) {
                const uint8_t p = v.T(i - d);
                if ((p & 0xC0) != 0x80) {
                    stray = pt_lead_len(p) <= d;
                    break;
                }
                if (v.M(i - d) & PT_CHUNK0) break; // the chunk starts with continuation bytes
            }
        }
        *bad = stray;
        return PT_CONT | PT_O;
    }
    uint32_t cp = 0;
    if (pt_decode(v, i, chunk_end, &cp) == 0) {
        *bad = true;
        return PT_O;
    }
    return cls_table[cp];
}

// ---------------------------------------------------------------- token starts
YB_HD bool pt_text_start(const PtView &v, uint64_t j, int64_t forced) { return (v.M(j) & PT_CHUNK0) || (int64_t)j == forced; }

// length of the character at j (valid text)
YB_HD int pt_char_len(const PtView &v, uint64_t j) { return pt_lead_len(v.T(j)); }

// Contraction suffix after an apostrophe at p: 1 for s d m t, 2 for ll ve re, 0 for none (case-sensitive, inside the chunk).
YB_HD int pt_contraction_len(const PtView &v, uint64_t p) {
    if (v.T(p) != '\'' || p + 1 >= v.n || (v.M(p + 1) & PT_CHUNK0)) return 0;
    const uint8_t x = v.T(p + 1);
    if (x == 's' || x == 'd' || x == 'm' || x == 't') return 1;
    if (p + 2 >= v.n || (v.M(p + 2) & PT_CHUNK0)) return 0;
    const uint8_t y = v.T(p + 2);
    if ((x == 'l' && y == 'l') || (x == 'v' && y == 'e') || (x == 'r' && y == 'e')) return 2;
    return 0;
}

// The run rules alone (no contraction, no special): j is a character start and not the start of the text.
YB_HD bool pt_base_start(const PtView &v, uint64_t j) {
    uint64_t prev = j - 1;
    for (int k = 0; k < 3 && prev > 0 && (v.M(prev) & PT_CONT); ++k) --prev; // (bounded: malformed text is reported, not followed)
    const uint8_t c = v.M(j) & PT_CLS, pc = v.M(prev) & PT_CLS;
    if (c != PT_S) {
        if (pc == c) return false;               // continues the run
        return v.T(prev) != ' ';              // a U+0020 right before the run starts the token instead
    }
    if (pc != PT_S) return true;                 // first character of a whitespace run
    const uint64_t nx = j + (uint64_t)pt_char_len(v, j);
    if (nx >= v.n || (v.M(nx) & PT_CHUNK0)) return false; // the run reaches the end of the text: one token
    return (v.M(nx) & PT_CLS) != PT_S;        // last character of a run of >= 2, a non-space follows
}

// Is the apostrophe-contraction at p taken?  `forced`: a position where a token is known to start (a special token ended
// there), or -1.  Callers guarantee that no text start lies in (p, j] for the j they are deciding.
YB_HD bool pt_contraction_taken(const PtView &v, uint64_t p, int64_t forced) {
    if (pt_contraction_len(v, p) == 0) return false;
    return pt_text_start(v, p, forced) || pt_base_start(v, p);
}

// THE rule: does a pre-token start at byte j?  (forced = -1, or the end of a special token that was matched: the text
// before it is consumed, a token starts there, and only apostrophes at or after it can open a contraction.)
YB_HD bool pt_is_start(const PtView &v, uint64_t j, int64_t forced) {
    if (v.M(j) & PT_CONT) return false;
    if (pt_text_start(v, j, forced)) return true;
    // j - 1 exists and belongs to the same text.  Contractions reach at most 3 bytes to the right of their apostrophe.
    if (pt_contraction_taken(v, j - 1, forced)) return false; // first letter of the suffix
    if (j >= 2 && !pt_text_start(v, j - 1, forced)) {
        if (pt_contraction_taken(v, j - 2, forced)) return pt_contraction_len(v, j - 2) == 1; // after 's / inside 'll
        if (j >= 3 && !pt_text_start(v, j - 2, forced) && pt_contraction_taken(v, j - 3, forced) &&
            pt_contraction_len(v, j - 3) == 2)
            return true; // after 'll 've 're
    }
    return pt_base_start(v, j);
}

// ---------------------------------------------------------------- special tokens
struct PtSpecials {
    const uint8_t *bytes;
    const uint32_t *off; // n + 1 offsets into bytes
    uint32_t n;
    uint32_t max_len;
};

// 1 + index of the first special (configuration order) whose bytes stand at i inside one chunk, 0 for none.
YB_HD uint32_t pt_special_at(const PtView &v, const PtSpecials &sp, uint64_t i) {
    for (uint32_t s = 0; s < sp.n; ++s) {
        const uint32_t len = sp.off[s + 1] - sp.off[s];
        if (len == 0 || i + len > v.n) continue;
        const uint8_t *w = sp.bytes + sp.off[s];
        bool eq = true;
        for (uint32_t k = 0; k < len && eq; ++k) eq = v.T(i + k) == w[k] && (k == 0 || !(v.M(i + k) & PT_CHUNK0));
        if (eq) return s + 1;
    }
    return 0;
}

YB_HD uint32_t pt_special_len(const PtSpecials &sp, uint32_t occ) { return sp.off[occ] - sp.off[occ - 1]; }

// Occurrences are looked up on demand (occ(q) = pt_special_at at q, usually behind a first-byte filter): they are rare,
// so no per-byte occurrence array is kept.

// An occurrence at i heads its chain iff no earlier occurrence (of the same chunk) reaches it: q + len(q) + 3 >= i.
template <class OccF>
YB_HD bool pt_special_is_head(const PtView &v, const PtSpecials &sp, OccF occ, uint64_t i) {
    if (v.M(i) & PT_CHUNK0) return true;
    const uint64_t window = (uint64_t)sp.max_len + 3;
    for (uint64_t d = 1; d <= window && d <= i; ++d) {
        const uint64_t q = i - d;
        const uint32_t o = occ(q);
        if (o && (uint64_t)pt_special_len(sp, o) + 3 >= d) return false;
        if (v.M(q) & PT_CHUNK0) break; // nothing before the chunk's first byte matters
    }
    return true;
}

// Resolves the chain headed by the occurrence `o0` at i, left to right, and writes the flags it changes.
template <class OccF>
YB_HD void pt_special_walk(const PtView &v, const PtSpecials &sp, OccF occ, uint8_t *flags, uint64_t i, uint32_t o0) {
    int64_t cover = -1;        // end of the last special that was taken
    uint64_t q = i;
    uint32_t oq = o0;
    uint64_t reach = i;        // last byte an occurrence of this chain can influence
    while (true) {
        const uint32_t len = pt_special_len(sp, oq);
        if (q + len + 3 > reach) reach = q + len + 3;
        bool taken;
        if (cover >= 0 && (int64_t)q < cover) taken = false;            // inside the previous match
        else taken = pt_is_start(v, q, cover >= 0 && (int64_t)q <= cover + 3 ? cover : -1);
        if (taken) {
            flags[q] = 1;
            for (uint32_t k = 1; k < len; ++k) flags[q + k] = 0;
            cover = (int64_t)(q + len);
            // a token starts right after the match; redo the (at most 3) positions a contraction there can reach
            for (uint64_t j = (uint64_t)cover; j < (uint64_t)cover + 4 && j < v.n; ++j) {
                if (v.M(j) & PT_CHUNK0) break; // the next chunk is a text of its own
                flags[j] = pt_is_start(v, j, cover) ? 1 : 0;
            }
        }
        // next occurrence of the chain
        uint64_t nq = q + 1;
        uint32_t no = 0;
        for (; nq <= reach && nq < v.n; ++nq) {
            if (v.M(nq) & PT_CHUNK0) break;
            no = occ(nq);
            if (no) break;
        }
        if (!no) return;
        q = nq;
        oq = no;
    }
}
