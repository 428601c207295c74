// tile_model.cpp -- TEST-ONLY CPU model of the tile-stream merge algorithm.
//
// Purpose: exercise the per-position rules in yet-another-bpe_amd/csrc/tile_logic.h (the exact header the
// HIP kernels include) against the oracle on this GPU-less container.  It mirrors the device data layout
// (tiles of CAP u16 slots, words never straddle tiles, SEP/PAD markers, flat vs weighted layout, in-tile
// compaction, incremental pair table driven by per-site deltas) with plain sequential loops.
// It is NOT part of the product and is never loaded by it.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../yet-another-bpe_amd/csrc/tile_logic.h"

namespace {

struct Model {
    int CAP, LMAX, S;
    bool weighted;
    std::vector<std::vector<uint16_t>> tiles;   // live prefix of each tile
    std::vector<uint32_t> wbase;                // weighted: index of the first word starting in the tile
    std::vector<uint64_t> freq;                 // weighted: per word
    std::vector<std::vector<uint16_t>> longw;   // long words (no SEP)
    std::vector<uint64_t> longf;
    std::unordered_map<uint32_t, int64_t> table;
    std::vector<std::string> tok;
    std::unordered_map<std::string, uint32_t> vocab;
    std::vector<uint32_t> m_left, m_right, m_merged;
    std::vector<uint64_t> m_count;
    bool keep_mismatch = false;
    uint32_t cur_key = 0xFFFFFFFFu;
};

void add_delta(Model &m, uint32_t key, int64_t d) {
    if (key == m.cur_key) return;  // the merged pair's count becomes exactly 0: zeroed at selection, never updated
    m.table[key] += d;
}

void count_all(Model &m) {
    m.table.clear();
    for (size_t t = 0; t < m.tiles.size(); t++) {
        auto &v = m.tiles[t];
        uint32_t wi = m.weighted ? m.wbase[t] : 0;
        for (size_t p = 0; p < v.size(); p++) {
            if (v[p] == YB_SEP) { wi++; continue; }
            if (v[p] < YB_PAD && p + 1 < v.size() && v[p + 1] < YB_PAD)
                add_delta(m, yb_pairkey(v[p], v[p + 1]), m.weighted ? (int64_t)m.freq[wi] : 1);
        }
    }
    for (size_t i = 0; i < m.longw.size(); i++)
        for (size_t p = 0; p + 1 < m.longw[i].size(); p++)
            add_delta(m, yb_pairkey(m.longw[i][p], m.longw[i][p + 1]), (int64_t)m.longf[i]);
}

void apply_tile(Model &m, size_t t, uint32_t a, uint32_t b, uint32_t c, bool verify_keep = true) {
    auto &v = m.tiles[t];
    int n = (int)v.size();
    auto T = [&](int q) -> uint32_t { return (q < 0 || q >= n) ? YB_PAD : v[q]; };
    // merged flags: greedy left to right; for a==b the parity rule used on the device
    std::vector<uint8_t> mr(n + 8, 0);
    if (a != b) {
        for (int p = 0; p + 1 < n; p++) mr[p] = (v[p] == a && v[p + 1] == b);
    } else {
        int last_non = -1;
        for (int p = 0; p < n; p++) {
            if (v[p] != a) { last_non = p; continue; }
            bool match = p + 1 < n && v[p + 1] == a;
            mr[p] = match && (((p - last_non - 1) & 1) == 0);
        }
    }
    auto M = [&](int q) -> int { return (q < 0 || q >= n) ? 0 : mr[q]; };
    bool any = false;
    for (int p = 0; p < n; p++) any |= mr[p];
    if (!any) return;
    // deltas
    uint32_t wi = m.weighted ? m.wbase[t] : 0;
    for (int p = 0; p < n; p++) {
        if (v[p] == YB_SEP) { wi++; continue; }
        if (!mr[p]) continue;
        int64_t w = m.weighted ? (int64_t)m.freq[wi] : 1;
        YbDeltas d;
        yb_site_deltas(p, a, b, c, T, M, d);
        if (d.left) { add_delta(m, d.lo, -w); add_delta(m, d.ln, +w); }
        if (d.right) { add_delta(m, d.ro, -w); add_delta(m, d.rn, +w); }
        add_delta(m, yb_pairkey(a, b), -w);
    }
    // compaction, site-driven (what k_apply does): drop b of every site, PAD, and dead (a b) words in flat mode
    std::vector<uint8_t> drop(n + 4, 0);
    for (int p = 0; p < n; p++) {
        if (v[p] == YB_PAD) drop[p] = 1;
        if (!mr[p]) continue;
        drop[p + 1] = 1;
        if (!m.weighted && yb_site_word_dies(p, T)) { drop[p] = 1; drop[p + 2] = 1; }
    }
    std::vector<uint16_t> out;
    for (int p = 0; p < n; p++) {
        if (drop[p]) continue;
        out.push_back(mr[p] ? (uint16_t)c : v[p]);
    }
    if (verify_keep) {  // the position-wise rule (used by retile) must agree
        std::vector<uint16_t> out2;
        for (int p = 0; p < n; p++) {
            uint32_t o;
            if (yb_keep(p, c, !m.weighted, T, M, o)) out2.push_back((uint16_t)o);
        }
        if (out2 != out) m.keep_mismatch = true;
    }
    v.swap(out);
}

void apply_long(Model &m, size_t i, uint32_t a, uint32_t b, uint32_t c) {
    auto &t = m.longw[i];
    int64_t w = (int64_t)m.longf[i];
    size_t len = t.size(), j = 0, o = 0;
    bool have_prev = false;
    uint32_t prev_old = 0, prev_new = 0;
    while (j < len) {
        if (j + 1 < len && t[j] == a && t[j + 1] == b) {
            if (have_prev) {
                add_delta(m, yb_pairkey(prev_old, a), -w);
                add_delta(m, yb_pairkey(prev_new, c), +w);
            }
            add_delta(m, yb_pairkey(a, b), -w);
            if (j + 2 < len) {
                bool next_site = (j + 3 < len) && t[j + 2] == a && t[j + 3] == b;
                if (!next_site) {
                    add_delta(m, yb_pairkey(b, t[j + 2]), -w);
                    add_delta(m, yb_pairkey(c, t[j + 2]), +w);
                }
            }
            t[o++] = (uint16_t)c;
            prev_old = b; prev_new = c; have_prev = true;
            j += 2;
        } else {
            uint16_t x = t[j];
            t[o++] = x;
            prev_old = prev_new = x; have_prev = true;
            j += 1;
        }
    }
    t.resize(o);
}

int tokcmp(const std::string &x, const std::string &y) {
    size_t n = std::min(x.size(), y.size());
    int c = memcmp(x.data(), y.data(), n);
    if (c) return c;
    return (x.size() > y.size()) - (x.size() < y.size());
}

}  // namespace

extern "C" {

// Returns number of merges; outputs ids.  verify: after every iteration the incremental table must equal a
// full recount (returns -1 - iteration on mismatch).
int tile_model_train(const uint8_t *bytes, const uint64_t *off, const uint64_t *wfreq, uint64_t n_words,
                     const uint8_t *tok_bytes, const uint32_t *tok_off, uint32_t n_tokens, uint32_t num_merges,
                     uint64_t min_frequency, int cap, int lmax, int verify, uint32_t *out_left, uint32_t *out_right,
                     uint32_t *out_merged, uint64_t *out_count) {
    Model m;
    m.CAP = cap; m.LMAX = lmax; m.S = cap - lmax + 1;
    m.weighted = wfreq != nullptr;
    for (uint32_t i = 0; i < n_tokens; i++) {
        m.tok.emplace_back((const char *)tok_bytes + tok_off[i], tok_off[i + 1] - tok_off[i]);
        m.vocab[m.tok.back()] = i;
    }
    uint64_t total = off[n_words] + n_words;
    size_t n_tiles = (size_t)((total + m.S - 1) / m.S);
    std::vector<std::vector<uint16_t>> slots(n_tiles, std::vector<uint16_t>(m.CAP, YB_PAD));
    std::vector<uint32_t> tlen(n_tiles, 0);
    m.wbase.assign(n_tiles, 0xFFFFFFFFu);
    for (uint64_t w = 0; w < n_words; w++) {
        uint64_t P = off[w] + w, L = off[w + 1] - off[w];
        size_t k = (size_t)(P / m.S);
        int slot = (int)(P - (uint64_t)k * m.S);
        if (m.weighted) { m.freq.push_back(wfreq[w]); m.wbase[k] = std::min(m.wbase[k], (uint32_t)w); }
        if (!m.weighted && L <= 1) continue;  // can never form a pair (trainer.py:231)
        if (L + 1 > (uint64_t)m.LMAX) {
            std::vector<uint16_t> lw(L);
            for (uint64_t j = 0; j < L; j++) lw[j] = bytes[off[w] + j];
            m.longw.push_back(lw);
            m.longf.push_back(m.weighted ? wfreq[w] : 1);
            if (m.weighted) {
                slots[k][slot] = YB_SEP;  // placeholder keeps word indices aligned
                tlen[k] = std::max<uint32_t>(tlen[k], slot + 1);
            }
            continue;
        }
        for (uint64_t j = 0; j < L; j++) slots[k][slot + j] = bytes[off[w] + j];
        slots[k][slot + L] = YB_SEP;
        tlen[k] = std::max<uint32_t>(tlen[k], slot + L + 1);
        if (slot + (int)L + 1 > m.CAP) return -1000000;
    }
    m.tiles.resize(n_tiles);
    for (size_t k = 0; k < n_tiles; k++) m.tiles[k].assign(slots[k].begin(), slots[k].begin() + tlen[k]);
    slots.clear();
    count_all(m);

    uint32_t done = 0;
    for (uint32_t it = 0; it < num_merges; it++) {
        // argmax with the byte-lexicographic tie-break (trainer.py:246)
        bool have = false; uint32_t bk = 0; int64_t bc = 0;
        for (auto &kv : m.table) {
            if (kv.second <= 0) continue;
            if (!have || kv.second > bc) { have = true; bk = kv.first; bc = kv.second; continue; }
            if (kv.second == bc) {
                int c = tokcmp(m.tok[kv.first >> 16], m.tok[bk >> 16]);
                if (c == 0) c = tokcmp(m.tok[kv.first & 0xffff], m.tok[bk & 0xffff]);
                if (c > 0) bk = kv.first;
            }
        }
        if (!have || (uint64_t)bc < min_frequency) break;
        uint32_t a = bk >> 16, b = bk & 0xffff;
        std::string mb = m.tok[a] + m.tok[b];
        uint32_t c;
        auto f = m.vocab.find(mb);
        if (f != m.vocab.end()) c = f->second;
        else { c = (uint32_t)m.tok.size(); m.tok.push_back(mb); m.vocab[mb] = c; }
        m.cur_key = bk;
        m.table[bk] = 0;
        for (size_t t = 0; t < m.tiles.size(); t++) apply_tile(m, t, a, b, c);
        for (size_t i = 0; i < m.longw.size(); i++) apply_long(m, i, a, b, c);
        m.cur_key = 0xFFFFFFFFu;
        out_left[done] = a; out_right[done] = b; out_merged[done] = c; out_count[done] = (uint64_t)bc;
        done++;
        if (m.keep_mismatch) return -2000000 - (int)it;
        if (verify) {
            auto inc = m.table;
            count_all(m);
            for (auto &kv : inc) {
                int64_t r = 0; auto g = m.table.find(kv.first); if (g != m.table.end()) r = g->second;
                if (r != kv.second) return -1 - (int)it;
            }
            for (auto &kv : m.table) {
                int64_t r = 0; auto g = inc.find(kv.first); if (g != inc.end()) r = g->second;
                if (r != kv.second) return -1 - (int)it;
            }
        }
    }
    return (int)done;
}
}
