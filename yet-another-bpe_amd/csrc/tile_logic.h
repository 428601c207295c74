// tile_logic.h -- per-position rules of the merge-apply pass, shared by the HIP kernels
// (yabpe_kernels.hip) and by the CPU unit-test model (tests/hostmodel/tile_model.cpp).
//
// Token stream format (DESIGN.md "Data layout"): u16 ids; every word is followed by SEP; PAD fills
// tile lead/tail regions.  Ids are < PAD, so SEP/PAD never match a pair.
//
// The rules implement trainer.py:276-285 (greedy left-to-right, non-overlapping replacement) and the net
// effect of the incremental recount trainer.py:264-273 + :290-294 on the pair table, expressed per merge
// site so that all sites can be processed in parallel:
//   site p (tok[p]==a, tok[p+1]==b, selected by the greedy rule) rewrites (a,b) -> c and changes the
//   pair to its left and the pair to its right.  Where two sites touch (…a b a b…), the pair between
//   them is accounted for by the RIGHT site only, so nothing is counted twice.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define YB_HD __host__ __device__ __forceinline__
#else
#define YB_HD inline
#endif

#define YB_SEP 0xFFFFu
#define YB_PAD 0xFFFEu
#define YB_MAX_TOKENS 0xFFFEu /* ids 0 .. 0xFFFD */

// key of the pair table: (left << 16) | right
YB_HD uint32_t yb_pairkey(uint32_t left, uint32_t right) { return (left << 16) | right; }
// the same pair as it appears in memory (little endian: first element in the low half)
YB_HD uint32_t yb_memkey(uint32_t left, uint32_t right) { return (right << 16) | left; }

struct YbDeltas {
    // fixed slots (no runtime-indexed arrays, so the device keeps them in registers):
    //   left:  lo -1, ln +1     right: ro -1, rn +1
    uint32_t lo, ln, ro, rn;
    bool left, right;
};

// Neighbour deltas of merge site p (the site's own (a,b) -1 is accounted separately).
//   T(q): token at position q of the tile (PAD outside), M(q): 1 if q is a merge site (0 outside).
template <class TokF, class MrgF>
YB_HD void yb_site_deltas(int p, uint32_t a, uint32_t b, uint32_t c, TokF T, MrgF M, YbDeltas &d) {
    const uint32_t L = T(p - 1);
    d.left = L < YB_PAD;
    if (M(p - 2)) { // left neighbour is the b of the site at p-2: old pair (b,a), new pair (c,c)
        d.lo = yb_pairkey(b, a);
        d.ln = yb_pairkey(c, c);
    } else {
        d.lo = yb_pairkey(L, a);
        d.ln = yb_pairkey(L, c);
    }
    const uint32_t R = T(p + 2);
    d.right = R < YB_PAD && !M(p + 2); // if p+2 is a site, that site's left side covers this boundary
    d.ro = yb_pairkey(b, R);
    d.rn = yb_pairkey(c, R);
}

// What position p of the old tile contributes to the rewritten tile.
// Returns 1 and sets `out` when the element is kept.
// drop_dead (flat layout only): words that are reduced to a single token can never produce a pair again
// (trainer.py:231 iterates range(len(word)-1)), so the token and its SEP are removed from the stream; empty
// words (SEP after SEP/PAD/tile start) likewise.
template <class TokF, class MrgF>
YB_HD int yb_keep(int p, uint32_t c, bool drop_dead, TokF T, MrgF M, uint32_t &out) {
    uint32_t v = T(p);
    if (v == YB_PAD) return 0;
    if (M(p - 1)) return 0; // consumed as the b of the site at p-1
    if (v < YB_PAD) {
        out = M(p) ? c : v;
        if (!drop_dead) return 1;
        bool prev_boundary = M(p - 2) ? false : (T(p - 1) >= YB_PAD);
        int nx = M(p) ? p + 2 : p + 1;
        bool next_sep = !M(nx) && T(nx) == YB_SEP;
        return !(prev_boundary && next_sep);
    }
    // v == SEP
    out = YB_SEP;
    if (!drop_dead) return 1;
    if (M(p - 2)) { // word ends with the c written at p-2
        bool pb = M(p - 4) ? false : (T(p - 3) >= YB_PAD);
        return !pb;
    }
    uint32_t u = T(p - 1);
    if (u >= YB_PAD) return 0; // empty word
    bool pb = M(p - 3) ? false : (T(p - 2) >= YB_PAD);
    return !pb;
}

// Site-driven form of the same rewrite, used by k_apply (work proportional to the number of sites):
// the only elements that leave a tile are the b of every site, PAD, and -- in the flat layout -- the two
// remaining slots (c, SEP) of a word that was exactly (a b): it is now a single token and can never produce
// a pair again.  This equals yb_keep() on a stream that holds no single-token / empty words, which the flat
// loader guarantees (it never emits them) and this rule preserves.
template <class TokF>
YB_HD bool yb_site_word_dies(int p, TokF T) {
    return T(p - 1) >= YB_PAD && T(p + 2) == YB_SEP;
}

