/*
 * bpe_oracle.c -- CPU restatement of the reference BPE training hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the timed CPU baseline.  The product path
 * (yet-another-bpe_amd/) never imports, links or calls it.
 *
 * What it restates (reference = DreamOneX/yet-another-bpe, pure Python):
 *   _init_base_vocab    src/yet_another_bpe/trainer.py:119-134
 *   _merge_loop         src/yet_another_bpe/trainer.py:216-302
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against
 *   - the reference's own fixture tests/fixtures_gpt2/train-bpe-reference-merges.txt
 *     (carried as tests/golden/g2_reference_merges_243.hex), and
 *   - vectors produced by running the reference itself in the build container
 *     (tests/golden/make_golden.py, committed next to its outputs).
 *
 * Structure follows the reference: word-frequency dedup (:221-225), pair counts
 * plus a pair -> words inverted index (:227-235), argmax with byte-lexicographic
 * tie-break (:246), rewrite of the affected words only (:254-294), vocab/merges
 * bookkeeping where token identity is the byte string (:296-300).
 *
 * Tokens are canonical ids: one id per distinct byte string, so id equality is
 * byte-string equality, which is what the reference's tuple-of-bytes words compare.
 */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    uint8_t *ptr;
    uint32_t len;
} tok_t;

typedef struct {
    uint32_t *tok; /* canonical token ids, in-place shrinking */
    uint32_t len;
    uint64_t freq;
    uint32_t stamp; /* iteration in which this word was last rewritten */
} word_t;

typedef struct {
    uint64_t key; /* (left << 32) | right ; UINT64_MAX = empty */
    int64_t count;
    uint32_t *words; /* inverted index; may hold stale/duplicate ids (reference keeps a superset too) */
    uint32_t n_words, cap_words;
    uint64_t touched; /* iteration + 1 in which the count last changed (see the argmax) */
} pair_t;

typedef struct oracle_result {
    tok_t *toks;
    uint32_t n_toks, cap_toks;
    /* bytes -> id map (open addressing over ids) */
    uint32_t *vmap;
    uint32_t vmap_cap;
    /* merges */
    uint32_t *m_left, *m_right, *m_merged;
    uint64_t *m_count;
    uint32_t n_merges;
    /* stats */
    uint64_t n_unique_words;
    uint64_t n_pairs_initial;
} oracle_result;

#define EMPTY_KEY UINT64_MAX

static uint64_t hash_bytes(const uint8_t *p, uint64_t n) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint64_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ULL;
    }
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ULL;
    h ^= h >> 32;
    return h;
}

static uint64_t hash_u64(uint64_t x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}

/* ---------- vocab: bytes -> id, identity is the byte string (trainer.py:130, :298) ---------- */

static void vmap_insert_raw(oracle_result *r, uint32_t id) {
    uint64_t h = hash_bytes(r->toks[id].ptr, r->toks[id].len);
    uint32_t m = r->vmap_cap - 1;
    uint32_t s = (uint32_t)h & m;
    while (r->vmap[s] != UINT32_MAX) s = (s + 1) & m;
    r->vmap[s] = id;
}

static void vmap_grow(oracle_result *r) {
    free(r->vmap);
    r->vmap_cap *= 2;
    r->vmap = (uint32_t *)malloc(sizeof(uint32_t) * r->vmap_cap);
    memset(r->vmap, 0xff, sizeof(uint32_t) * r->vmap_cap);
    for (uint32_t i = 0; i < r->n_toks; i++) vmap_insert_raw(r, i);
}

static uint32_t vocab_find(const oracle_result *r, const uint8_t *p, uint32_t n) {
    uint64_t h = hash_bytes(p, n);
    uint32_t m = r->vmap_cap - 1;
    uint32_t s = (uint32_t)h & m;
    while (r->vmap[s] != UINT32_MAX) {
        const tok_t *t = &r->toks[r->vmap[s]];
        if (t->len == n && memcmp(t->ptr, p, n) == 0) return r->vmap[s];
        s = (s + 1) & m;
    }
    return UINT32_MAX;
}

static uint32_t vocab_add(oracle_result *r, const uint8_t *p, uint32_t n) {
    if (r->n_toks == r->cap_toks) {
        r->cap_toks *= 2;
        r->toks = (tok_t *)realloc(r->toks, sizeof(tok_t) * r->cap_toks);
    }
    uint32_t id = r->n_toks++;
    r->toks[id].ptr = (uint8_t *)malloc(n ? n : 1);
    memcpy(r->toks[id].ptr, p, n);
    r->toks[id].len = n;
    if ((uint64_t)r->n_toks * 2 > r->vmap_cap)
        vmap_grow(r);
    else
        vmap_insert_raw(r, id);
    return id;
}

/* Python bytes ordering: unsigned bytewise, a proper prefix sorts lower. */
static int tok_cmp(const tok_t *a, const tok_t *b) {
    uint32_t n = a->len < b->len ? a->len : b->len;
    int c = memcmp(a->ptr, b->ptr, n);
    if (c) return c;
    return (a->len > b->len) - (a->len < b->len);
}

/* ---------- pair table ---------- */

typedef struct {
    pair_t *slots;
    uint64_t cap, used;
} ptab_t;

static void ptab_init(ptab_t *t, uint64_t cap) {
    t->cap = cap;
    t->used = 0;
    t->slots = (pair_t *)malloc(sizeof(pair_t) * cap);
    for (uint64_t i = 0; i < cap; i++) {
        t->slots[i].key = EMPTY_KEY;
        t->slots[i].count = 0;
        t->slots[i].words = NULL;
        t->slots[i].n_words = t->slots[i].cap_words = 0;
        t->slots[i].touched = 0;
    }
}

static pair_t *ptab_get(ptab_t *t, uint64_t key);

static void ptab_grow(ptab_t *t) {
    ptab_t n;
    ptab_init(&n, t->cap * 2);
    for (uint64_t i = 0; i < t->cap; i++) {
        if (t->slots[i].key == EMPTY_KEY) continue;
        pair_t *d = ptab_get(&n, t->slots[i].key);
        *d = t->slots[i];
    }
    free(t->slots);
    *t = n;
}

static pair_t *ptab_get(ptab_t *t, uint64_t key) {
    if (t->used * 2 >= t->cap) ptab_grow(t);
    uint64_t m = t->cap - 1;
    uint64_t s = hash_u64(key) & m;
    while (t->slots[s].key != EMPTY_KEY) {
        if (t->slots[s].key == key) return &t->slots[s];
        s = (s + 1) & m;
    }
    t->slots[s].key = key;
    t->used++;
    return &t->slots[s];
}

static void pair_index_add(pair_t *p, uint32_t w) {
    if (p->n_words && p->words[p->n_words - 1] == w) return; /* cheap dedupe of repeats in one word */
    if (p->n_words == p->cap_words) {
        p->cap_words = p->cap_words ? p->cap_words * 2 : 4;
        p->words = (uint32_t *)realloc(p->words, sizeof(uint32_t) * p->cap_words);
    }
    p->words[p->n_words++] = w;
}

/* ---------- argmax (trainer.py:246) ----------
 * max(pair_counts.items(), key=lambda x: (x[1], x[0])) is a function of the table's contents only: the pair with the
 * greatest (count, bytes(left), bytes(right)).  Scanning the whole table for it every iteration (what the reference's max()
 * does) makes a job with millions of distinct pairs take hours on one core, so the SAME maximum is kept available in a
 * binary heap of (count, pair) snapshots ordered by that very key: every pair is pushed when its count changes (once per
 * iteration, with the count it ends the iteration on), a snapshot whose count is no longer the pair's is stale and
 * dropped when it surfaces.  Every pair with a positive count has its current snapshot in the heap, so the first snapshot
 * that is not stale is the table's maximum.  tests/test_oracle_golden.py pins the result on the reference's vectors,
 * among them the 8,199-merge run whose iterations mostly have ties for the top count. */
typedef struct {
    int64_t count;
    uint64_t key;
} snap_t;
typedef struct {
    snap_t *a;
    uint64_t n, cap;
} heap_t;
static const tok_t *g_toks; /* (the heap orders by token bytes) */
static int snap_gt(const snap_t *x, const snap_t *y) {
    if (x->count != y->count) return x->count > y->count;
    int c = tok_cmp(&g_toks[x->key >> 32], &g_toks[y->key >> 32]);
    if (c == 0) c = tok_cmp(&g_toks[(uint32_t)x->key], &g_toks[(uint32_t)y->key]);
    return c > 0;
}
static void heap_push(heap_t *h, snap_t v) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 1024;
        h->a = (snap_t *)realloc(h->a, sizeof(snap_t) * h->cap);
    }
    uint64_t i = h->n++;
    while (i) {
        uint64_t p = (i - 1) >> 1;
        if (!snap_gt(&v, &h->a[p])) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = v;
}
static void heap_pop(heap_t *h) {
    snap_t v = h->a[--h->n];
    uint64_t i = 0;
    for (;;) {
        uint64_t l = 2 * i + 1, r = l + 1, m;
        if (l >= h->n) break;
        m = (r < h->n && snap_gt(&h->a[r], &h->a[l])) ? r : l;
        if (!snap_gt(&h->a[m], &v)) break;
        h->a[i] = h->a[m];
        i = m;
    }
    if (h->n) h->a[i] = v;
}

/* ---------- word dedup table (trainer.py:221-225) ---------- */

typedef struct {
    uint64_t *hash;
    uint32_t *wid; /* UINT32_MAX = empty */
    uint64_t cap;
} wtab_t;

/* freq: occurrences of each input word (NULL: 1 each).  Equal words are pooled and their counts added, exactly what
   word_freq[word] += 1 (trainer.py:221-225) gives when the occurrences come one by one: a caller that has already pooled
   its pre-tokens (an 8 GiB text does not fit as a list of occurrences) passes the pooled words with their counts. */
oracle_result *bpe_oracle_train_weighted(const uint8_t *bytes, const uint64_t *off, const uint64_t *freq, uint64_t n_words,
                                         const uint8_t *sp_bytes, const uint32_t *sp_off, uint32_t n_specials,
                                         uint64_t vocab_size, uint64_t min_frequency, double max_seconds);

oracle_result *bpe_oracle_train(const uint8_t *bytes, const uint64_t *off, uint64_t n_words,
                                const uint8_t *sp_bytes, const uint32_t *sp_off, uint32_t n_specials,
                                uint64_t vocab_size, uint64_t min_frequency, double max_seconds) {
    return bpe_oracle_train_weighted(bytes, off, NULL, n_words, sp_bytes, sp_off, n_specials, vocab_size, min_frequency, max_seconds);
}

oracle_result *bpe_oracle_train_weighted(const uint8_t *bytes, const uint64_t *off, const uint64_t *freq, uint64_t n_words,
                                         const uint8_t *sp_bytes, const uint32_t *sp_off, uint32_t n_specials,
                                         uint64_t vocab_size, uint64_t min_frequency, double max_seconds) {
    /* max_seconds > 0 bounds the merge loop's wall time (bench.py's cpu_baseline leg); 0 = run to the end */
    struct timespec ts0;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    oracle_result *r = (oracle_result *)calloc(1, sizeof(oracle_result));
    r->cap_toks = 1024;
    r->toks = (tok_t *)malloc(sizeof(tok_t) * r->cap_toks);
    r->vmap_cap = 4096;
    r->vmap = (uint32_t *)malloc(sizeof(uint32_t) * r->vmap_cap);
    memset(r->vmap, 0xff, sizeof(uint32_t) * r->vmap_cap);

    /* _init_base_vocab (trainer.py:119-134): 256 single bytes, then each special token's
       UTF-8 bytes in config order, skipped when those bytes are already a key. */
    for (uint32_t b = 0; b < 256; b++) {
        uint8_t c = (uint8_t)b;
        vocab_add(r, &c, 1);
    }
    for (uint32_t s = 0; s < n_specials; s++) {
        const uint8_t *p = sp_bytes + sp_off[s];
        uint32_t n = sp_off[s + 1] - sp_off[s];
        if (vocab_find(r, p, n) == UINT32_MAX) vocab_add(r, p, n);
    }
    uint32_t base_vocab = r->n_toks;

    /* word_freq (trainer.py:221-225): equal byte strings pooled with a count */
    wtab_t wt;
    wt.cap = 1024;
    while (wt.cap < n_words * 2 + 16) wt.cap <<= 1;
    wt.hash = (uint64_t *)malloc(sizeof(uint64_t) * wt.cap);
    wt.wid = (uint32_t *)malloc(sizeof(uint32_t) * wt.cap);
    memset(wt.wid, 0xff, sizeof(uint32_t) * wt.cap);
    uint64_t cap_w = 1024, nw = 0;
    word_t *words = (word_t *)malloc(sizeof(word_t) * cap_w);
    uint64_t *wstart = (uint64_t *)malloc(sizeof(uint64_t) * cap_w); /* byte offset of representative */
    for (uint64_t i = 0; i < n_words; i++) {
        const uint8_t *p = bytes + off[i];
        uint64_t n = off[i + 1] - off[i];
        uint64_t h = hash_bytes(p, n);
        uint64_t m = wt.cap - 1, s = h & m;
        for (;;) {
            if (wt.wid[s] == UINT32_MAX) {
                if (nw == cap_w) {
                    cap_w *= 2;
                    words = (word_t *)realloc(words, sizeof(word_t) * cap_w);
                    wstart = (uint64_t *)realloc(wstart, sizeof(uint64_t) * cap_w);
                }
                wt.hash[s] = h;
                wt.wid[s] = (uint32_t)nw;
                words[nw].len = (uint32_t)n;
                words[nw].freq = freq ? freq[i] : 1;
                words[nw].stamp = UINT32_MAX;
                words[nw].tok = NULL;
                wstart[nw] = off[i];
                nw++;
                break;
            }
            if (wt.hash[s] == h) {
                word_t *w = &words[wt.wid[s]];
                if (w->len == n && memcmp(bytes + wstart[wt.wid[s]], p, n) == 0) {
                    w->freq += freq ? freq[i] : 1;
                    break;
                }
            }
            s = (s + 1) & m;
        }
    }
    free(wt.hash);
    free(wt.wid);
    r->n_unique_words = nw;
    for (uint64_t i = 0; i < nw; i++) {
        words[i].tok = (uint32_t *)malloc(sizeof(uint32_t) * (words[i].len ? words[i].len : 1));
        const uint8_t *p = bytes + wstart[i];
        for (uint32_t j = 0; j < words[i].len; j++) words[i].tok[j] = p[j]; /* byte b has id b */
    }
    free(wstart);

    /* initial pair counts + inverted index (trainer.py:227-235); every adjacent position counts */
    ptab_t pt;
    ptab_init(&pt, 1 << 16);
    for (uint64_t i = 0; i < nw; i++) {
        word_t *w = &words[i];
        for (uint32_t j = 0; j + 1 < w->len; j++) {
            uint64_t key = ((uint64_t)w->tok[j] << 32) | w->tok[j + 1];
            pair_t *p = ptab_get(&pt, key);
            p->count += (int64_t)w->freq;
            pair_index_add(p, (uint32_t)i);
        }
    }
    r->n_pairs_initial = pt.used;
    heap_t hp = {NULL, 0, 0};
    for (uint64_t s2 = 0; s2 < pt.cap; s2++)
        if (pt.slots[s2].key != EMPTY_KEY && pt.slots[s2].count > 0) {
            g_toks = r->toks;
            heap_push(&hp, (snap_t){pt.slots[s2].count, pt.slots[s2].key});
        }
    uint64_t *touched = NULL; /* keys whose count changed in this iteration */
    uint64_t n_touched = 0, cap_touched = 0;

    /* num_merges = max(0, vocab_size - len(vocab)) (trainer.py:238) */
    uint64_t num_merges = vocab_size > base_vocab ? vocab_size - base_vocab : 0;
    uint64_t cap_m = num_merges < 1024 ? num_merges + 1 : 1024;
    r->m_left = (uint32_t *)malloc(sizeof(uint32_t) * cap_m);
    r->m_right = (uint32_t *)malloc(sizeof(uint32_t) * cap_m);
    r->m_merged = (uint32_t *)malloc(sizeof(uint32_t) * cap_m);
    r->m_count = (uint64_t *)malloc(sizeof(uint64_t) * cap_m);

    uint8_t *mbuf = NULL;
    uint32_t mbuf_cap = 0;
    uint32_t *scratch = NULL;
    uint32_t scratch_cap = 0;

    for (uint64_t it = 0; it < num_merges; it++) {
        if (max_seconds > 0 && (it & 15) == 0) {
            struct timespec ts1;
            clock_gettime(CLOCK_MONOTONIC, &ts1);
            if ((ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec) > max_seconds) break;
        }
        /* argmax (trainer.py:246): max over (count, (bytes(p0), bytes(p1))); only count > 0 exist */
        pair_t *best = NULL;
        g_toks = r->toks; /* (the token array may have been reallocated) */
        while (hp.n) {
            pair_t *p = ptab_get(&pt, hp.a[0].key);
            if (p->count == hp.a[0].count && p->count > 0) {
                best = p;
                break;
            }
            heap_pop(&hp); /* stale: the pair's count has moved on since this snapshot */
        }
        if (!best) break;                                      /* trainer.py:242-243 */
        if ((uint64_t)best->count < min_frequency) break;      /* trainer.py:247-248 */

        uint32_t x = (uint32_t)(best->key >> 32), y = (uint32_t)best->key;
        uint64_t best_count = (uint64_t)best->count;

        /* merged = p0 + p1 (trainer.py:251); id is reused when the bytes already exist (:298-300) */
        uint32_t mlen = r->toks[x].len + r->toks[y].len;
        if (mlen > mbuf_cap) {
            mbuf_cap = mlen * 2;
            mbuf = (uint8_t *)realloc(mbuf, mbuf_cap);
        }
        memcpy(mbuf, r->toks[x].ptr, r->toks[x].len);
        memcpy(mbuf + r->toks[x].len, r->toks[y].ptr, r->toks[y].len);
        uint32_t z = vocab_find(r, mbuf, mlen);
        int is_new = (z == UINT32_MAX);
        if (is_new) z = vocab_add(r, mbuf, mlen);

        /* affected words = snapshot of the pair's index (trainer.py:254) */
        uint32_t n_aff = best->n_words;
        uint32_t *aff = best->words;
        best->words = NULL;
        best->n_words = best->cap_words = 0;

        for (uint32_t a = 0; a < n_aff; a++) {
            word_t *w = &words[aff[a]];
            if (w->stamp == (uint32_t)it) continue; /* already rewritten this iteration */
            int has = 0;
            for (uint32_t j = 0; j + 1 < w->len; j++)
                if (w->tok[j] == x && w->tok[j + 1] == y) {
                    has = 1;
                    break;
                }
            if (!has) continue; /* stale index entry (reference: freq == 0 -> continue, :257-259) */
            w->stamp = (uint32_t)it;

            /* decrement every old pair (trainer.py:264-273) */
            for (uint32_t j = 0; j + 1 < w->len; j++) {
                pair_t *p = ptab_get(&pt, ((uint64_t)w->tok[j] << 32) | w->tok[j + 1]);
                p->count -= (int64_t)w->freq;
                if (p->touched != it + 1) {
                    p->touched = it + 1;
                    if (n_touched == cap_touched) {
                        cap_touched = cap_touched ? cap_touched * 2 : 1024;
                        touched = (uint64_t *)realloc(touched, sizeof(uint64_t) * cap_touched);
                    }
                    touched[n_touched++] = p->key;
                }
            }
            /* greedy left-to-right, non-overlapping rewrite (trainer.py:276-285) */
            if (w->len > scratch_cap) {
                scratch_cap = w->len * 2;
                scratch = (uint32_t *)realloc(scratch, sizeof(uint32_t) * scratch_cap);
            }
            uint32_t n = 0, j = 0;
            while (j < w->len) {
                if (j + 1 < w->len && w->tok[j] == x && w->tok[j + 1] == y) {
                    scratch[n++] = z;
                    j += 2;
                } else {
                    scratch[n++] = w->tok[j];
                    j += 1;
                }
            }
            memcpy(w->tok, scratch, sizeof(uint32_t) * n);
            w->len = n;
            /* increment every new pair and index it (trainer.py:290-294) */
            for (uint32_t k = 0; k + 1 < w->len; k++) {
                pair_t *p = ptab_get(&pt, ((uint64_t)w->tok[k] << 32) | w->tok[k + 1]);
                p->count += (int64_t)w->freq;
                if (w->tok[k] == z || w->tok[k + 1] == z) pair_index_add(p, aff[a]);
                if (p->touched != it + 1) {
                    p->touched = it + 1;
                    if (n_touched == cap_touched) {
                        cap_touched = cap_touched ? cap_touched * 2 : 1024;
                        touched = (uint64_t *)realloc(touched, sizeof(uint64_t) * cap_touched);
                    }
                    touched[n_touched++] = p->key;
                }
            }
        }
        free(aff);
        /* the pairs whose count changed get a fresh snapshot (the count they end this iteration on) */
        g_toks = r->toks;
        for (uint64_t q = 0; q < n_touched; q++) {
            pair_t *p = ptab_get(&pt, touched[q]);
            if (p->count > 0) heap_push(&hp, (snap_t){p->count, p->key});
        }
        n_touched = 0;

        if (r->n_merges == cap_m) {
            cap_m *= 2;
            r->m_left = (uint32_t *)realloc(r->m_left, sizeof(uint32_t) * cap_m);
            r->m_right = (uint32_t *)realloc(r->m_right, sizeof(uint32_t) * cap_m);
            r->m_merged = (uint32_t *)realloc(r->m_merged, sizeof(uint32_t) * cap_m);
            r->m_count = (uint64_t *)realloc(r->m_count, sizeof(uint64_t) * cap_m);
        }
        r->m_left[r->n_merges] = x;   /* merges.append(best_pair) always (trainer.py:296) */
        r->m_right[r->n_merges] = y;
        r->m_merged[r->n_merges] = z; /* new id only if the bytes were new (trainer.py:298-300) */
        r->m_count[r->n_merges] = best_count;
        r->n_merges++;
    }

    for (uint64_t i = 0; i < nw; i++) free(words[i].tok);
    free(words);
    for (uint64_t s = 0; s < pt.cap; s++) free(pt.slots[s].words);
    free(pt.slots);
    free(mbuf);
    free(scratch);
    free(hp.a);
    free(touched);
    return r;
}

uint32_t bpe_oracle_n_merges(const oracle_result *r) { return r->n_merges; }
uint32_t bpe_oracle_n_tokens(const oracle_result *r) { return r->n_toks; }
uint64_t bpe_oracle_n_unique_words(const oracle_result *r) { return r->n_unique_words; }
uint64_t bpe_oracle_n_pairs_initial(const oracle_result *r) { return r->n_pairs_initial; }

void bpe_oracle_get_merges(const oracle_result *r, uint32_t *left, uint32_t *right, uint32_t *merged,
                           uint64_t *count) {
    memcpy(left, r->m_left, sizeof(uint32_t) * r->n_merges);
    memcpy(right, r->m_right, sizeof(uint32_t) * r->n_merges);
    memcpy(merged, r->m_merged, sizeof(uint32_t) * r->n_merges);
    if (count) memcpy(count, r->m_count, sizeof(uint64_t) * r->n_merges);
}

uint32_t bpe_oracle_token_len(const oracle_result *r, uint32_t id) { return r->toks[id].len; }

void bpe_oracle_token_bytes(const oracle_result *r, uint32_t id, uint8_t *out) {
    memcpy(out, r->toks[id].ptr, r->toks[id].len);
}

void bpe_oracle_free(oracle_result *r) {
    if (!r) return;
    for (uint32_t i = 0; i < r->n_toks; i++) free(r->toks[i].ptr);
    free(r->toks);
    free(r->vmap);
    free(r->m_left);
    free(r->m_right);
    free(r->m_merged);
    free(r->m_count);
    free(r);
}
