// batch_sim.cpp -- design study (CPU, not part of the product): how many consecutive merges of sequential BPE can be
// decided from ONE look at the pair table?  Runs exact sequential BPE on pooled words (its own small implementation) and,
// at every batch start, walks the candidates in selection order (count, bytes(left), bytes(right)) under the batching rule
// the device selection uses (see select_body in csrc/yabpe_kernels.h):
//
//   accept candidate j after the accepted merges i < j iff
//     (1) right(j) != a_i and left(j) != b_i             (its own count cannot move: only (x,a_i), (b_i,y), (a_i,b_i) fall)
//     (2) no pair with count >= n_j -- other than accepted merges -- has right token a_i or left token b_i or is (b_i,a_i)
//         (every pair a merge creates contains its new token and is bounded by count(x,a_i) / count(b_i,y) / count(b_i,a_i))
//     (3) the merged bytes of i are a new token (an existing token's pairs already have counts)
//   and stop at the first candidate that fails.
//
// The prediction is then checked against the sequence sequential BPE really takes (any mismatch is printed: the rule must
// be exact), and the batch-size histogram is printed by segment of the job.
//
//   g++ -O2 -o /tmp/batch_sim tools/batch_sim.cpp && /tmp/batch_sim words.bin 32000 [maxbatch]
//   words.bin: u64 n_words, u64 n_bytes, u64 off[n_words+1], u8 bytes[n_bytes]
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <set>
#include <unordered_map>
#include <vector>

struct Word { std::vector<uint32_t> t; uint64_t f; };
struct PairInfo { int64_t cnt = 0; std::vector<uint32_t> words; };
static std::vector<std::string> tok;
static std::unordered_map<std::string, uint32_t> vocab;
static std::unordered_map<uint64_t, PairInfo> table;
static std::vector<Word> words;

static inline uint64_t K(uint32_t a, uint32_t b) { return ((uint64_t)a << 32) | b; }
static bool lex_gt(uint64_t x, uint64_t y) { // (bytes(l), bytes(r)) of x > of y
    const std::string &xl = tok[x >> 32], &yl = tok[y >> 32];
    if (xl != yl) return xl > yl;
    return tok[(uint32_t)x] > tok[(uint32_t)y];
}
struct Cand { uint64_t key; int64_t cnt; };
static bool cand_before(const Cand &x, const Cand &y) { return x.cnt != y.cnt ? x.cnt > y.cnt : lex_gt(x.key, y.key); }
struct CandLess { bool operator()(const Cand &x, const Cand &y) const { return x.key != y.key && cand_before(x, y); } };
static std::set<Cand, CandLess> order;               // every pair with count > 0, in selection order
static std::unordered_map<uint64_t, int64_t> touched; // key -> count before the current merge
static inline void bump(uint64_t key, int64_t d) {
    PairInfo &p = table[key];
    touched.emplace(key, p.cnt);
    p.cnt += d;
}
static void commit_touched() {
    for (auto &kv : touched) {
        const int64_t now = table[kv.first].cnt;
        if (now == kv.second) continue;
        if (kv.second > 0) order.erase(Cand{kv.first, kv.second});
        if (now > 0) order.insert(Cand{kv.first, now});
    }
    touched.clear();
}

// applies merge (a,b) -> c exactly; returns sites
static uint64_t apply_merge(uint32_t a, uint32_t b, uint32_t c) {
    auto it = table.find(K(a, b));
    std::vector<uint32_t> aff;
    aff.swap(it->second.words);
    std::sort(aff.begin(), aff.end());
    aff.erase(std::unique(aff.begin(), aff.end()), aff.end());
    uint64_t sites = 0;
    std::vector<uint32_t> nt;
    for (uint32_t wi : aff) {
        Word &w = words[wi];
        bool has = false;
        for (size_t j = 0; j + 1 < w.t.size(); ++j) if (w.t[j] == a && w.t[j + 1] == b) { has = true; break; }
        if (!has) continue;
        for (size_t j = 0; j + 1 < w.t.size(); ++j) bump(K(w.t[j], w.t[j + 1]), -(int64_t)w.f);
        nt.clear();
        for (size_t j = 0; j < w.t.size();) {
            if (j + 1 < w.t.size() && w.t[j] == a && w.t[j + 1] == b) { nt.push_back(c); j += 2; sites += w.f; }
            else { nt.push_back(w.t[j]); j += 1; }
        }
        w.t = nt;
        for (size_t j = 0; j + 1 < w.t.size(); ++j) {
            bump(K(w.t[j], w.t[j + 1]), (int64_t)w.f);
            if (w.t[j] == c || w.t[j + 1] == c) table[K(w.t[j], w.t[j + 1])].words.push_back(wi);
        }
    }
    commit_touched();
    return sites;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: batch_sim words.bin n_merges [maxbatch] [variant]\n"); return 2; }
    const uint32_t n_merges = (uint32_t)atoi(argv[2]);
    const uint32_t maxbatch = argc > 3 ? (uint32_t)atoi(argv[3]) : 16;
    const int variant = argc > 4 ? atoi(argv[4]) : 0;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    uint64_t nw, nb;
    if (fread(&nw, 8, 1, f) != 1 || fread(&nb, 8, 1, f) != 1) return 1;
    std::vector<uint64_t> off(nw + 1);
    std::vector<uint8_t> bytes(nb);
    if (fread(off.data(), 8, nw + 1, f) != nw + 1 || fread(bytes.data(), 1, nb, f) != nb) return 1;
    fclose(f);
    for (int i = 0; i < 256; ++i) { tok.push_back(std::string(1, (char)i)); vocab[tok.back()] = i; }
    tok.push_back("<|endoftext|>"); vocab[tok.back()] = 256;
    {
        std::unordered_map<std::string, uint64_t> pool;
        pool.reserve(nw / 16);
        for (uint64_t i = 0; i < nw; ++i) pool[std::string((const char *)&bytes[off[i]], off[i + 1] - off[i])]++;
        for (auto &kv : pool) {
            Word w; w.f = kv.second;
            for (unsigned char ch : kv.first) w.t.push_back(ch);
            words.push_back(std::move(w));
        }
    }
    fprintf(stderr, "unique words %zu\n", words.size());
    for (uint32_t wi = 0; wi < words.size(); ++wi) {
        const Word &w = words[wi];
        for (size_t j = 0; j + 1 < w.t.size(); ++j) { PairInfo &p = table[K(w.t[j], w.t[j + 1])]; p.cnt += (int64_t)w.f; p.words.push_back(wi); }
    }
    for (auto &kv : table) if (kv.second.cnt > 0) order.insert(Cand{kv.first, kv.second.cnt});
    std::vector<uint64_t> hist(maxbatch + 1, 0);
    uint64_t seg_batches = 0, seg_merges = 0, mism = 0;
    uint32_t done = 0, seg_start = 0;
    uint64_t why[8] = {0};
    while (done < n_merges) {
        // candidates: everything with count >= 0.5 * best (a superset of what the walk needs)
        if (order.empty()) break;
        const int64_t best = order.begin()->cnt;
        std::vector<Cand> cl;
        for (auto it = order.begin(); it != order.end() && cl.size() < 4096 && it->cnt * 2 >= best; ++it) cl.push_back(*it);
        // walk
        std::vector<uint64_t> acc;
        uint32_t stop_reason = 0;
        for (size_t j = 0; j < cl.size() && acc.size() < maxbatch && done + acc.size() < n_merges; ++j) {
            const uint32_t p = cl[j].key >> 32, q = (uint32_t)cl[j].key;
            const int64_t nj = cl[j].cnt;
            bool ok = true;
            for (uint64_t ak : acc) {
                const uint32_t a = ak >> 32, b = (uint32_t)ak;
                if (variant == 0 ? (p == a || p == b || q == a || q == b) : (q == a || p == b)) { ok = false; stop_reason = 1; break; }
            }
            if (ok && !acc.empty()) {
                // (2): any non-accepted listed pair with count >= nj whose right token is some a_i or whose left token is some b_i
                for (size_t k = 0; k < cl.size() && cl[k].cnt >= nj && ok; ++k) {
                    if (std::find(acc.begin(), acc.end(), cl[k].key) != acc.end()) continue;
                    const uint32_t l = cl[k].key >> 32, r = (uint32_t)cl[k].key;
                    for (uint64_t ak : acc) {
                        const uint32_t a = ak >> 32, b = (uint32_t)ak;
                        if (!(r == a || l == b)) continue;
                        if (variant >= 2 && cl[k].cnt == nj) {
                            // tie: the pairs this one can turn into reach at most nj; they block only if they would be selected before j.
                            // left token of a new pair: l itself, or the new token of an accepted merge whose b is l; right token likewise
                            // variant 2: exact byte comparison; variant 3: what the device can decide from 8-byte prefixes + lengths
                            auto cmp3 = [&](const std::string &x, const std::string &y) -> int { // -1 / 0 / +1, 2 = unknown (variant 3)
                                if (variant == 2) return x < y ? -1 : x > y ? 1 : 0;
                                std::string xp = x.substr(0, 8), yp = y.substr(0, 8);
                                xp.resize(8, '\0'); yp.resize(8, '\0');
                                if (xp != yp) return xp < yp ? -1 : 1;
                                if (x.size() <= 8 && y.size() <= 8) return x.size() < y.size() ? -1 : x.size() > y.size() ? 1 : 0;
                                return 2;
                            };
                            auto new_gt_j = [&](const std::string &nl, const std::string &nr) { // may (nl, nr) be selected before (p, q)?
                                const int cl_ = cmp3(nl, tok[p]);
                                if (cl_ == 2 || cl_ > 0) return true;
                                if (cl_ < 0) return false;
                                const int cr = cmp3(nr, tok[q]);
                                return cr == 2 || cr > 0 || cr == 0;
                            };
                            bool blocks = false;
                            std::vector<std::string> lefts, rights;
                            lefts.push_back(tok[l]); rights.push_back(tok[r]);
                            for (uint64_t a2 : acc) {
                                if ((uint32_t)a2 == l) lefts.push_back(tok[a2 >> 32] + tok[(uint32_t)a2]);
                                if ((uint32_t)(a2 >> 32) == r) rights.push_back(tok[a2 >> 32] + tok[(uint32_t)a2]);
                            }
                            for (size_t li = 0; li < lefts.size(); ++li)
                                for (size_t ri = 0; ri < rights.size(); ++ri)
                                    if ((li || ri) && new_gt_j(lefts[li], rights[ri])) blocks = true;
                            if (!blocks) continue;
                            ok = false; stop_reason = 5; break;
                        }
                        ok = false; stop_reason = 2; break;
                    }
                }
            }
            if (!ok) break;
            // (3) the previous accepted merge must create a NEW token -- checked when it is accepted: an old token ends the batch
            acc.push_back(cl[j].key);
            const std::string m = tok[p] + tok[q];
            if (vocab.count(m) || p == q) { stop_reason = 3; break; }
            // a merge that creates the same bytes as an earlier one in the batch: ends the batch as well
            bool dup = false;
            for (size_t e = 0; e + 1 < acc.size(); ++e) if (tok[acc[e] >> 32] + tok[(uint32_t)acc[e]] == m) dup = true;
            if (dup) { stop_reason = 3; break; }
        }
        if (cl.size() && acc.size() == cl.size()) stop_reason = 4;
        why[stop_reason]++;
        // run the batch sequentially and compare with what sequential BPE selects
        for (size_t e = 0; e < acc.size(); ++e) {
            if (order.empty()) { done = n_merges; break; }
            const int64_t bc = order.begin()->cnt; const uint64_t bk = order.begin()->key;
            if (bk != acc[e]) {
                ++mism;
                if (mism < 10) fprintf(stderr, "MISMATCH at merge %u (batch pos %zu): predicted (%u,%u) real (%u,%u) cnt %lld\n", done, e, (uint32_t)(acc[e] >> 32), (uint32_t)acc[e], (uint32_t)(bk >> 32), (uint32_t)bk, (long long)bc);
                acc.resize(e); // what was right so far stays; redo from here
                break;
            }
            const uint32_t a = bk >> 32, b = (uint32_t)bk;
            const std::string m = tok[a] + tok[b];
            uint32_t c;
            auto vit = vocab.find(m);
            if (vit == vocab.end()) { c = (uint32_t)tok.size(); tok.push_back(m); vocab[m] = c; } else c = vit->second;
            apply_merge(a, b, c);
            ++done;
        }
        hist[std::min<size_t>(acc.size(), maxbatch)]++;
        seg_batches++;
        seg_merges += acc.size();
        if (done - seg_start >= (done < 400 ? 50u : 2000u) || done >= n_merges) {
            printf("merges %6u..%6u  batches %6llu  mean batch %.2f  best %lld  table %zu  stops: conflict1 %llu bound2 %llu tie5 %llu oldtok %llu full %llu cap %llu\n", seg_start, done,
                   (unsigned long long)seg_batches, seg_batches ? (double)seg_merges / seg_batches : 0.0, (long long)best, table.size(),
                   (unsigned long long)why[1], (unsigned long long)why[2], (unsigned long long)why[5], (unsigned long long)why[3], (unsigned long long)why[4], (unsigned long long)why[0]);
            fflush(stdout);
            seg_start = done; seg_batches = seg_merges = 0;
            memset(why, 0, sizeof why);
            // drop dead entries now and then
            for (auto it = table.begin(); it != table.end();) { if (it->second.cnt <= 0 && it->second.words.empty()) it = table.erase(it); else ++it; }
        }
    }
    printf("mismatches %llu\nhistogram:", (unsigned long long)mism);
    for (size_t k = 0; k <= maxbatch; ++k) printf(" %zu:%llu", k, (unsigned long long)hist[k]);
    printf("\n");
    return 0;
}
