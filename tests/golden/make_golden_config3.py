#!/usr/bin/env python3
"""Full-size pin (BASELINE configs[2]): the 1 GiB synthetic byte corpus of SURVEY 8d (config 3, generated on the host by
yet_another_bpe/synth.py -- bit-identical to the device generator) through the CPU oracle (oracle/bpe_oracle.c, itself
pinned against the reference by the other fixtures) for 32,000 merges.  Writes tests/golden/g7_config3_meta.json with
SHA-256 digests of the merges list (hex-line serialisation of oracle.merges_hex) at several prefixes.  About 10 minutes
and 6 GiB of memory on one core; run once, commit the JSON.

    python tests/golden/make_golden_config3.py
"""
import hashlib
import json
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
from oracle import oracle  # noqa: E402
from yet_another_bpe import synth  # noqa: E402

t0 = time.time()
spec = synth.SynthSpec.config3(1024 << 20)
flat, off = synth.generate(spec)
print(f"corpus: {flat.size} bytes, {len(off) - 1} words ({time.time() - t0:.0f} s)", flush=True)
sp = ["<|endoftext|>"]
t1 = time.time()
vocab, merges, ids = oracle.train_flat(flat, off, 257 + 32000, 1, sp, return_ids=True)
print(f"oracle: {len(merges)} merges ({time.time() - t1:.0f} s)", flush=True)
text = oracle.merges_hex(merges)
lines = text.splitlines(keepends=True)
meta = {"config": "SURVEY 8d config 3: target 1 GiB, alphabet 0..255, 1,000,000 Zipf types, seed 3; specials ['<|endoftext|>'], min_frequency 1",
        "corpus_bytes": int(flat.size), "n_words": int(len(off) - 1), "corpus_sha256": hashlib.sha256(flat.tobytes()).hexdigest(),
        "unique_words": int(ids["unique_words"]), "n_merges": len(merges), "vocab_size": len(vocab),
        "merges_sha256": {str(k): hashlib.sha256("".join(lines[:k]).encode()).hexdigest() for k in (100, 1000, 5000, 10000, 20000, 32000) if k <= len(lines)},
        "first_count": int(ids["count"][0]), "last_count": int(ids["count"][-1]),
        "id_triples_sha256": hashlib.sha256(ids["left"].tobytes() + ids["right"].tobytes() + ids["merged"].tobytes()).hexdigest()}
(HERE / "g7_config3_meta.json").write_text(json.dumps(meta, indent=1))
print(json.dumps(meta, indent=1))
