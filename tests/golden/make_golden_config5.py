#!/usr/bin/env python3
"""G8 -- the shape of BASELINE configs[4] at a size one GPU holds: TEXT in -> GPT-2 pre-tokenisation -> pooled words ->
50,000 merges (reference path: trainer.py:63-92 train(), :136-214 _preprocess_corpus, :216-302 _merge_loop).

Two 1 GiB synthetic texts (SURVEY 8d generator, so the GPU box regenerates them bit-identically with `yabpe_synth_generate`
and nothing large is committed):
  config2_text   50,000 lowercase word types, every word prefixed with one space (the config-2 generator at 1 GiB)
  mixed_ascii    200,000 types over letters / digits / punctuation / apostrophes / spaces / newlines, space-prefixed:
                 contractions, digit runs, punctuation runs and whitespace runs with look-ahead all occur
The oracle is oracle/pretok.py's pattern (the `regex` module = the reference's own dependency on this step) run over the
WHOLE text as one chunk, then oracle/bpe_oracle.c on the pre-tokens.  Writes tests/golden/g8_config5_meta.json: digests of
the text, of the pre-token lengths (u32 little-endian) and of the merges list at several prefixes.

    python tests/golden/make_golden_config5.py            # ~30 min, ~30 GiB of memory, one core
"""
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
from oracle import oracle, pretok  # noqa: E402
from yet_another_bpe import synth  # noqa: E402

MIXED = b"abcdefghijklmnopqrstuvwxyzetaoinshrETAOIN0123456789  \n''.,-!?stdm"
SPECS = {
    "config2_text": dict(target=1 << 30, n_types=50_000, seed=2, alphabet=b"abcdefghijklmnopqrstuvwxyz", space_prefix=True),
    "mixed_ascii": dict(target=1 << 30, n_types=200_000, seed=5, alphabet=MIXED, space_prefix=True),
}
N_MERGES = 50_000
SP = ["<|endoftext|>"]
PREFIXES = (1000, 10000, 32000, 50000)


def one(name: str, s: dict) -> dict:
    t0 = time.time()
    flat, off = synth.generate(synth.SynthSpec(s["target"], s["n_types"], s["seed"], s["alphabet"], s["space_prefix"]))
    n_gen_words = len(off) - 1
    del off
    data = flat.tobytes()
    print(f"[{name}] text: {len(data)} bytes, {n_gen_words} generated words ({time.time() - t0:.0f} s)", flush=True)
    text = data.decode("utf-8")  # ASCII: character index == byte index
    pat = pretok.split_pattern(SP)
    ends = np.empty(len(data) // 2 + 16, dtype=np.uint64)  # grown if needed
    n = 0
    t1 = time.time()
    for m in pat.finditer(text):
        e = m.end()
        if e == m.start():
            continue  # trainer.py:169: empty matches are dropped
        if n == len(ends):
            ends = np.concatenate([ends, np.empty(len(ends), dtype=np.uint64)])
        ends[n] = e
        n += 1
    del text
    poff = np.zeros(n + 1, dtype=np.uint64)
    poff[1:] = ends[:n]
    del ends
    assert int(poff[-1]) == len(data), "the pattern covers every character"
    lens = np.diff(poff).astype(np.uint32)
    print(f"[{name}] regex: {n} pre-tokens, longest {int(lens.max())} ({time.time() - t1:.0f} s)", flush=True)
    t2 = time.time()
    vocab, merges, ids = oracle.train_flat(flat, poff, 257 + N_MERGES, 1, SP, return_ids=True)
    print(f"[{name}] oracle: {len(merges)} merges, {ids['unique_words']} unique words ({time.time() - t2:.0f} s)", flush=True)
    lines = oracle.merges_hex(merges).splitlines(keepends=True)
    return {
        "generator": {"target_bytes": s["target"], "n_types": s["n_types"], "seed": s["seed"], "alphabet_hex": s["alphabet"].hex(),
                      "space_prefix": s["space_prefix"]},
        "special_tokens": SP, "min_frequency": 1,
        "text_bytes": len(data), "text_sha256": hashlib.sha256(data).hexdigest(), "generated_words": n_gen_words,
        "pretokens": n, "pretoken_lengths_u32_sha256": hashlib.sha256(lens.tobytes()).hexdigest(), "longest_pretoken": int(lens.max()),
        "unique_words": int(ids["unique_words"]), "n_merges": len(merges), "vocab_size": len(vocab),
        "merges_sha256": {str(k): hashlib.sha256("".join(lines[:k]).encode()).hexdigest() for k in PREFIXES if k <= len(lines)},
        "first_count": int(ids["count"][0]), "last_count": int(ids["count"][-1]),
        "id_triples_sha256": hashlib.sha256(ids["left"].tobytes() + ids["right"].tobytes() + ids["merged"].tobytes()).hexdigest(),
        "regex_version": __import__("regex").__version__,
    }


def main() -> None:
    out = HERE / "g8_config5_meta.json"
    meta = json.loads(out.read_text()) if out.exists() else {}
    for name in (sys.argv[1:] or list(SPECS)):
        meta[name] = one(name, SPECS[name])
        out.write_text(json.dumps(meta, indent=1))
        print(json.dumps(meta[name], indent=1), flush=True)


if __name__ == "__main__":
    main()
