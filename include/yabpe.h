/*
 * yabpe.h -- C ABI of the MI355X-native BPE training hot path (libyabpe.so).
 *
 * The reference (DreamOneX/yet-another-bpe) is pure Python and has no FFI: its seam for this path is the
 * method BBPETrainer._merge_loop (src/yet_another_bpe/trainer.py:216-302), fed by train() (:63-92).
 * These entry points are what a binding for that method needs; each one names the reference lines whose
 * RESULT it reproduces.  Plain pointers and sizes only (no torch / numpy types); every pointer the caller
 * passes stays owned by the caller; the library owns only its opaque context and the device buffers it
 * hands out through yabpe_synth_generate().
 *
 * All functions return 0 (YABPE_OK) or a negative error code; yabpe_last_error() gives the message.
 * A context is bound to one GPU and must not be used from two threads at once.
 * There is NO CPU fallback: without a usable gfx950 device yabpe_create() fails with YABPE_E_NODEVICE.
 */
#ifndef YABPE_H
#define YABPE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YABPE_ABI_VERSION 2

enum {
    YABPE_OK = 0,
    YABPE_E_INVALID = -1,  /* bad argument / call order */
    YABPE_E_NODEVICE = -2, /* no HIP device */
    YABPE_E_HIP = -3,      /* HIP runtime error (message has the call) */
    YABPE_E_CAPACITY = -4, /* a documented limit was hit (token ids are u16: at most 65534 tokens) */
    YABPE_E_INTERNAL = -5, /* device-side invariant violated */
    YABPE_E_COMM = -6,     /* RCCL error */
    YABPE_E_UTF8 = -7      /* yabpe_pretokenize: the text is not valid UTF-8 (position reported) */
};

/* yabpe_load_words flags */
#define YABPE_LOAD_DEDUP 0x1u /* pool equal words on the device first (trainer.py:221-225) */

typedef struct yabpe_ctx yabpe_ctx;

/* Library / device ---------------------------------------------------------------------------------- */
int yabpe_abi_version(void);
int yabpe_device_count(void);

/* Creates a context on HIP device `device_id`.  One context per _merge_loop call (or reuse via yabpe_reset). */
int yabpe_create(yabpe_ctx **out, int device_id);
void yabpe_destroy(yabpe_ctx *ctx);
/* Message of the last error on this context (ctx == NULL: of the last failed yabpe_create). Never NULL. */
const char *yabpe_last_error(const yabpe_ctx *ctx);

/* Tunables by name (see DESIGN.md "Tunables"): "check_interval", "retile_pct", "apply_blocks", "event_sample",
   "table_min_log2", "skip_index", "cand_argmax", "verify" ... */
int yabpe_set_option(yabpe_ctx *ctx, const char *name, int64_t value);

/* Base vocabulary ------------------------------------------------------------------------------------
 * Result of _init_base_vocab (trainer.py:119-134): ids 0..n_tokens-1, token i = tok_bytes[tok_off[i]..tok_off[i+1]).
 * The first 256 entries must be the single bytes 0..255 in order (trainer.py:123-125).  The host computes the
 * list (dedup rule of :130 included); the device needs the bytes for the byte-lexicographic tie-break (:246)
 * and for "merged not in vocab" (:298). */
int yabpe_set_vocab(yabpe_ctx *ctx, const uint8_t *tok_bytes, const uint32_t *tok_off, uint32_t n_tokens);

/* Corpus ---------------------------------------------------------------------------------------------
 * The `sequences` argument of _merge_loop (trainer.py:216) as flat buffers: word i = bytes[word_off[i]..word_off[i+1]).
 * word_freq == NULL means every word counts once (flat layout: every occurrence resident in HBM).
 * With word_freq the caller has already pooled equal words (trainer.py:221-225) and passes their counts.
 * Pointers may be host or device memory (detected); device buffers are read in place, host buffers are staged. */
int yabpe_load_words(yabpe_ctx *ctx, const uint8_t *bytes, const uint64_t *word_off, const uint64_t *word_freq,
                     uint64_t n_words, uint32_t flags);

/* Merge loop -----------------------------------------------------------------------------------------
 * Runs trainer.py:238-300: at most `num_merges` iterations (the host computes max(0, vocab_size - len(vocab)),
 * :238), stops early when no pair is left (:242-243) or the best count < min_frequency (:247-248).
 * Outputs, one entry per merge in selection order (arrays of capacity num_merges, may be NULL):
 *   out_left/out_right   ids of the merged pair            (merges.append(best_pair), :296)
 *   out_merged           id of left+right: a fresh id, or the existing id when those bytes are already a
 *                        token (no id consumed, :298-300)
 *   out_count            the pair's count when it was selected
 * May be called again to continue training with more merges. */
int yabpe_train(yabpe_ctx *ctx, uint32_t num_merges, uint64_t min_frequency, uint32_t *out_left,
                uint32_t *out_right, uint32_t *out_merged, uint64_t *out_count, uint32_t *out_n_merges);

/* Token bytes after training: token id -> bytes (vocab of trainer.py:302, inverted). */
int yabpe_n_tokens(yabpe_ctx *ctx, uint32_t *out_n_tokens);
int yabpe_token_bytes(yabpe_ctx *ctx, uint32_t id, uint8_t *out, uint32_t cap, uint32_t *out_len);

/* Measurement ---------------------------------------------------------------------------------------- */
typedef struct yabpe_stats_t {
    uint64_t n_words;          /* W: resident words (after optional dedup) */
    uint64_t n_words_input;    /* words passed to yabpe_load_words */
    uint64_t n_long_words;     /* words handled by the long-word path */
    uint64_t tokens_initial;   /* T_0 */
    uint64_t tokens_now;       /* T_i = T_0 - sum of merged sites */
    uint64_t merges_done;
    uint64_t n_tiles;
    uint64_t live_slots;       /* u16 slots the apply kernel currently reads */
    uint64_t table_capacity;
    uint64_t table_entries;
    uint64_t retiles, table_rebuilds;
    double load_ms;            /* yabpe_load_words device time */
    double train_ms;           /* device time of all yabpe_train calls (event-timed) */
    double apply_ms_sampled;   /* sum of the event-timed apply phases (k_apply, or k_scan + k_slow) */
    uint64_t apply_launches_sampled;
    uint64_t apply_algo_bytes_sampled;   /* sum of 2*(T_i + W) over the sampled launches */
    uint64_t apply_actual_bytes_sampled; /* sum of 2*live_slots over the sampled launches */
    uint64_t algo_bytes_total;           /* sum over all iterations of 2*(T_i + W) */
    /* split form only: the streaming scan kernel (k_scan) by itself */
    double scan_ms_sampled;
    uint64_t scan_launches_sampled;
    uint64_t scan_algo_bytes_sampled;
    uint64_t scan_actual_bytes_sampled;
    /* skip index: launches of k_scan_skip and the tiles they actually read (the rest was skipped by signature) */
    uint64_t scan_skip_launches;
    uint64_t scan_skip_tiles_read;
    /* candidate argmax: rebuilds of the candidate list (scans of the table) and batches that fell back to the full table scan */
    uint64_t cand_rebuilds;
    uint64_t cand_rescans;
    /* fused per-merge launches: apply of merge i + selection of merge i+1 in one kernel (the production form) */
    uint64_t fused_launches;
    /* the streaming phase: event-timed launches of the fused k_apply (whole iteration when the selection is fused in) */
    double dense_ms_sampled;
    uint64_t dense_launches_sampled;
    uint64_t dense_algo_bytes_sampled;   /* sum of 2*(T_i + W) over them */
    uint64_t dense_actual_bytes_sampled; /* sum of the bytes of live slots + tile lengths they read */
    /* the sparse phase (skip index): device time from the switch to the end of the last yabpe_train call, merges applied in it */
    double sparse_ms;
    uint64_t sparse_merges;
    /* the second half of each yabpe_train call's merges (few sites per merge: the latency-bound regime) */
    double tail_ms;
    uint64_t tail_merges;
    /* multi-GPU: all-gathers of [header | records] buffers (one per merge), bytes every rank receives per exchange, record
       capacity per rank, times the buffers had to grow (a merge produced more records than fit: global recount) */
    uint64_t exchanges, exchange_bytes, exchange_cap_records, exchange_growths, exchange_max_records;
    /* launches of the sparse phase: one launch applies a BATCH of merges (sparse_merges / sparse_launches = mean batch) */
    uint64_t sparse_launches;
    /* ... and the launches of the second half of each yabpe_train call's merges (tail_merges / tail_launches = mean batch there) */
    uint64_t tail_launches;
    /* peer-to-peer exchange: device time of the event-timed push-and-wait launches (one per round of launches) */
    double exchange_ms_sampled;
    uint64_t exchanges_sampled;
    uint64_t exchange_p2p;     /* 1: the exchanges go peer to peer (0: through the attached transport's all-gather) */
    /* the streaming phase: its launches and the merges they applied (a launch applies a batch of up to "batch_max_stream"
       merges in one pass over the stream); dense_algo_bytes_sampled / dense_actual_bytes_sampled are the phase's totals
       scaled to the event-timed launches */
    uint64_t dense_launches, dense_merges;
} yabpe_stats_t;
int yabpe_stats(yabpe_ctx *ctx, yabpe_stats_t *out);
/* Per-iteration log of the last yabpe_train call: sites merged M_i and live slots read by iteration i. */
int yabpe_iter_log(yabpe_ctx *ctx, uint64_t *out_sites, uint64_t *out_live_slots, uint32_t cap, uint32_t *out_n);

/* Per-launch HIP-event timings of k_apply from the last yabpe_train call (option "event_sample" = N times every
   Nth launch): iteration index (relative to the call), duration of the whole apply phase and of its streaming
   kernel (k_scan / k_scan_skip, or the fused k_apply) in microseconds. */
int yabpe_event_log(yabpe_ctx *ctx, uint32_t *out_iter, float *out_us, float *out_scan_us, uint32_t cap, uint32_t *out_n);

/* Latency pieces of one sparse merge, measured on the (otherwise idle) device: what the floor of the per-merge launch is
   built from (bench.py prints the model).  All in microseconds. */
typedef struct yabpe_latency_t {
    double launch_gap_us;       /* back-to-back dependent launches of an empty kernel on one stream, per launch */
    double load_trip_us;        /* one dependent global load that misses the caches (pointer chase, one lane) */
    double coherent_trip_us;    /* the same with device-scope loads (hand-offs inside a launch) */
    double atomic_trip_us;      /* one dependent returning device-scope atomic */
} yabpe_latency_t;
int yabpe_latency_probe(yabpe_ctx *ctx, yabpe_latency_t *out);

/* Debug / self-check: recount every pair from the token stream into a scratch table and compare with the
   incrementally maintained table.  *out_mismatches = number of differing keys. */
int yabpe_verify_table(yabpe_ctx *ctx, uint64_t *out_mismatches);
/* Debug: decode the resident token stream back to bytes and return an order-independent checksum over
   (word bytes, segmentation) plus the number of words/tokens it saw. */
int yabpe_stream_checksum(yabpe_ctx *ctx, uint64_t *out_sum, uint64_t *out_words, uint64_t *out_tokens);

/* Synthetic corpus of SURVEY.md 8(d), generated on the device (bit-identical to yet_another_bpe/synth.py).
   Returns device pointers owned by the library (freed by yabpe_synth_free or yabpe_destroy). */
int yabpe_synth_generate(yabpe_ctx *ctx, uint64_t target_bytes, uint32_t n_types, uint64_t seed,
                         const uint8_t *alphabet, uint32_t alphabet_len, int space_prefix,
                         uint8_t **out_dev_bytes, uint64_t **out_dev_off, uint64_t *out_n_words, uint64_t *out_n_bytes);
/* The same Zipf draw (w_j = floor(2^40 / (j+1)), u = rnd(seed,3,i) % sum w) over a lexicon the caller supplies (host arrays:
   bytes + n_types+1 offsets, every entry 1..65,535 bytes): the drawn entries are concatenated until target_bytes is reached.
   For synthetic TEXT (yet_another_bpe/synth.py text_lexicon: multi-byte UTF-8 words, digits, punctuation, whitespace runs,
   long letter runs) whose pre-tokens yabpe_pretokenize then finds; out_dev_off are the piece boundaries (not pre-tokens). */
int yabpe_synth_generate_lex(yabpe_ctx *ctx, uint64_t target_bytes, uint32_t n_types, uint64_t seed,
                             const uint8_t *lex_bytes, const uint64_t *lex_off,
                             uint8_t **out_dev_bytes, uint64_t **out_dev_off, uint64_t *out_n_pieces, uint64_t *out_n_bytes);
int yabpe_synth_free(yabpe_ctx *ctx);
/* Copy `n` bytes device->host / host->device (for fixtures and the CPU-baseline sample). */
int yabpe_memcpy_d2h(yabpe_ctx *ctx, void *dst_host, const void *src_dev, uint64_t n);
int yabpe_memcpy_h2d(yabpe_ctx *ctx, void *dst_dev, const void *src_host, uint64_t n);

/* Pre-tokeniser (the step before the path) -----------------------------------------
 * Reproduces _preprocess_corpus (trainer.py:136-214) on the device: every chunk [chunk_off[k], chunk_off[k+1]) of `text`
 * (the last one ends at n_bytes; chunk_off == NULL: one chunk) is decoded as UTF-8 and split with the GPT-2 pattern of
 * trainer.py:163, preceded by the special tokens in the given order (:165-167), exactly as regex.findall does.  The host
 * keeps what it does in the reference: reading files and choosing the chunk cuts (:172-198).
 * Results: *out_dev_text = the text in device memory (`text` itself when it already is a device pointer, else a staged
 * copy owned by the library) and *out_dev_word_off = n_words + 1 offsets into it (device memory, owned by the library,
 * released by yabpe_pretokenize_free / yabpe_destroy): pre-token i = text[off[i], off[i+1]).  Both can be passed
 * straight to yabpe_load_words (no copy; add YABPE_LOAD_DEDUP to pool equal pre-tokens, trainer.py:221-225).
 * Malformed UTF-8: returns YABPE_E_UTF8 and *out_bad_pos = UnicodeDecodeError.start of the first bad chunk (:156-161).
 * The character classes (\p{L}, \p{N}, \s) are those of the third-party `regex` module the reference uses
 * (csrc/unicode_classes.inc, generated by tools/gen_unicode_classes.py). */
int yabpe_pretokenize(yabpe_ctx *ctx, const uint8_t *text, uint64_t n_bytes, const uint64_t *chunk_off, uint32_t n_chunks,
                      const uint8_t *special_bytes, const uint32_t *special_off, uint32_t n_special,
                      const uint8_t **out_dev_text, uint64_t **out_dev_word_off, uint64_t *out_n_words, int64_t *out_bad_pos);
int yabpe_pretokenize_free(yabpe_ctx *ctx);

/* Multi-GPU (one process per GPU; words are sharded by the caller) -----------------
 * Every rank holds its shard of the words and a replica of the pair table.  After each apply pass the ranks
 * exchange their aggregated (pair, delta) records with ONE all-gather on the compute stream and every rank adds all
 * of them to its replica (integer sums: order-independent, so all ranks select the same merge).
 * Attach the communicator after yabpe_create / yabpe_set_vocab and BEFORE yabpe_load_words; then every rank calls
 * yabpe_load_words (its shard: word_off may point into the middle of a larger offsets array) and yabpe_train
 * collectively with the same arguments.  All ranks return the same merges.
 * unique_id: the 128-byte ncclUniqueId made by yabpe_comm_unique_id on rank 0 and broadcast by the caller. */
int yabpe_comm_unique_id(uint8_t out_id[128]);
int yabpe_comm_init(yabpe_ctx *ctx, int rank, int n_ranks, const uint8_t unique_id[128]);
/* Same protocol over a caller-supplied transport instead of RCCL (other fabrics; the repo's tests use it to run
 * 2 ranks on one GPU).  fn must gather `nbytes` from every rank's DEVICE buffer send_dev into recv_dev (rank
 * order) and return 0; it is called with the context's stream idle. */
typedef int (*yabpe_allgather_fn)(void *user, const void *send_dev, void *recv_dev, uint64_t nbytes);
int yabpe_comm_init_custom(yabpe_ctx *ctx, int rank, int n_ranks, yabpe_allgather_fn fn, void *user);
/* Peer-to-peer exchange instead of the all-gather (after yabpe_comm_init / yabpe_comm_init_custom, before yabpe_load_words;
 * collective: every rank calls it).  Each rank exports its receive area as a hipIpc handle (the attached transport carries
 * the 64-byte handles once), maps its peers' areas, and from then on every exchange is ONE small launch that pushes this
 * rank's records into the peers' memory (xGMI between GPUs of a node; two processes on one GPU work the same way), raises a
 * flag there and waits for the peers' flags -- no collective kernel, no host in the loop.  The transport stays attached
 * for the rare small agreements (lockstep check, buffer growth).  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 where the host
 * driver only supports dmabuf IPC. */
int yabpe_comm_enable_p2p(yabpe_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* YABPE_H */
