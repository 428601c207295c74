"""Pure-Python restatement of the reference's merge loop with the reference's own data structures -- TEST
INFRASTRUCTURE ONLY, like the rest of oracle/: imported by tests/ and by bench.py's cpu_baseline leg (the CPU number the
north star names: "the reference's pure-Python trainer timed on the box's host cores"), never by the product package.

What it restates (DreamOneX/yet-another-bpe, src/yet_another_bpe/trainer.py):
  _init_base_vocab  :119-134   256 single bytes, then the special tokens' UTF-8 bytes in order, duplicates skipped
  _merge_loop       :216-302   words pooled as tuples of byte strings with a frequency (:221-225); a dict of pair counts
                               and a dict pair -> set of words that contain it (:227-235); every iteration takes
                               max over ALL live pairs by (count, pair) (:246), rewrites only the words in the pair's set
                               (:254-294: old pairs decremented and dropped at <= 0, greedy left-to-right replacement,
                               new pairs incremented), appends the merge and gives the merged bytes an id unless those
                               bytes are a token already (:296-300)
The cost profile is therefore the reference's: O(live pairs) Python work per iteration for the max, dict/set updates per
affected word.  Pinned against the golden vectors made by running the reference (tests/test_oracle_golden.py)."""
from __future__ import annotations

import time
from collections import defaultdict


def base_vocab(special_tokens) -> dict[bytes, int]:
    vocab = {bytes([b]): b for b in range(256)}  # :123-125
    for tok in special_tokens:  # :128-132
        raw = tok.encode("utf-8") if isinstance(tok, str) else bytes(tok)
        if raw not in vocab:
            vocab[raw] = len(vocab)
    return vocab


def _replace_pair(word: tuple, left: bytes, right: bytes, fused: bytes) -> tuple:
    """Greedy left-to-right, non-overlapping (:276-285)."""
    out = []
    i, n = 0, len(word)
    while i < n:
        if i + 1 < n and word[i] == left and word[i + 1] == right:
            out.append(fused)
            i += 2
        else:
            out.append(word[i])
            i += 1
    return tuple(out)


def merge_loop(sequences, vocab_size: int, min_frequency: int, special_tokens, max_seconds: float = 0.0):
    """-> (vocab: dict[bytes, int], merges: list[(bytes, bytes)]) as BBPETrainer._merge_loop returns them (:302).
    `sequences`: iterable of byte strings / int lists.  max_seconds > 0 stops the loop early (bench.py's bounded sample)."""
    vocab = base_vocab(special_tokens)
    freq_of: dict[tuple, int] = defaultdict(int)  # :221-225
    for seq in sequences:
        freq_of[tuple(bytes([b]) for b in seq)] += 1
    count_of: dict[tuple, int] = defaultdict(int)  # :227-235
    words_with: dict[tuple, set] = defaultdict(set)
    for word, f in freq_of.items():
        for pair in zip(word, word[1:]):
            count_of[pair] += f
            words_with[pair].add(word)
    merges: list[tuple[bytes, bytes]] = []
    deadline = time.time() + max_seconds if max_seconds > 0 else None
    for _ in range(max(0, vocab_size - len(vocab))):  # :238-241
        if not count_of or (deadline is not None and time.time() > deadline):
            break
        best = max(count_of.items(), key=lambda kv: (kv[1], kv[0]))[0]  # :246
        if count_of[best] < min_frequency:  # :247-248
            break
        left, right = best
        fused = left + right
        for word in list(words_with.get(best, ())):  # :254
            f = freq_of.get(word, 0)
            if f == 0:
                continue
            del freq_of[word]
            for pair in zip(word, word[1:]):  # :264-273
                count_of[pair] -= f
                if count_of[pair] <= 0:
                    del count_of[pair]
                    words_with.pop(pair, None)
                else:
                    words_with[pair].discard(word)
            new_word = _replace_pair(word, left, right, fused)
            freq_of[new_word] = freq_of.get(new_word, 0) + f  # :288
            for pair in zip(new_word, new_word[1:]):  # :290-294
                count_of[pair] += f
                words_with[pair].add(new_word)
        merges.append(best)  # :296
        if fused not in vocab:  # :298-300
            vocab[fused] = len(vocab)
    return vocab, merges
