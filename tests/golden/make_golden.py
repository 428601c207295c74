#!/usr/bin/env python3
"""Generates the golden vectors in this directory by RUNNING THE REFERENCE (DreamOneX/yet-another-bpe,
mounted read-only at /root/reference) in the build container.  The reference never travels to the GPU box;
only this script and its outputs (data: inputs + expected outputs) are committed.

    cd /root/repo && python tests/golden/make_golden.py

Outputs (all under tests/golden/):
  corpus.en, g2_reference_merges_243.txt      data files the reference's own tests hold
                                               (tests/fixtures_gpt2/corpus.en, train-bpe-reference-merges.txt)
  snapshot_special_tokens.pkl                 tests/_snapshots/test_train_bpe_special_tokens.pkl (G7; primitive opcodes only)
  data/*.txt                                  tests/data/{simple,empty,unicode,multiline,sample}.txt
  g1_corpus_en_exhaustive.hex                 reference merges on corpus.en to exhaustion (min_frequency=1), hex lines
  g1_corpus_en_vocab_1000.json                reference vocab (hex -> id) at vocab_size=1000
  g345_cases.json                             G3 (Sennrich), G4 (edge cases), random small trials: inputs + reference outputs
  g5_config2_merges_1000.hex                  reference merges on the 10 MiB synthetic corpus of SURVEY §8(d) config 2
  g6_pretokens.json                           sha256/counts of reference _preprocess_corpus outputs
"""
from __future__ import annotations

import hashlib
import json
import random
import shutil
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF / "src"))
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))

from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig  # noqa: E402  (the REFERENCE)

assert "/root/reference/" in sys.modules["yet_another_bpe.trainer"].__file__

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("yabpe_synth", REPO / "yet-another-bpe_amd/yet_another_bpe/synth.py")
synth = importlib.util.module_from_spec(_spec)
sys.modules["yabpe_synth"] = synth
_spec.loader.exec_module(synth)


def hexlines(merges) -> str:
    return "".join(f"{a.hex()} {b.hex()}\n" for a, b in merges)


def ref_merge_loop(sequences, vocab_size, min_frequency, special_tokens):
    cfg = BBPETrainerConfig(vocab_size=vocab_size, min_frequency=min_frequency, max_workers=1,
                            special_tokens=list(special_tokens))
    return BBPETrainer(cfg)._merge_loop(sequences)


def ref_pretokens(path: Path, special_tokens, chunk_size_bytes=1 << 30, max_workers=1):
    cfg = BBPETrainerConfig(max_workers=max_workers, chunk_size_bytes=chunk_size_bytes,
                            special_tokens=list(special_tokens))
    return BBPETrainer(cfg)._preprocess_corpus([path])


def case(name, words, special_tokens, vocab_size, min_frequency):
    seqs = [list(w) for w in words]
    vocab, merges = ref_merge_loop(seqs, vocab_size, min_frequency, special_tokens)
    return {
        "name": name,
        "words": [bytes(w).hex() for w in words],
        "special_tokens": list(special_tokens),
        "vocab_size": vocab_size,
        "min_frequency": min_frequency,
        "merges": [[a.hex(), b.hex()] for a, b in merges],
        "vocab": {k.hex(): v for k, v in vocab.items() if v >= 256},
        "vocab_len": len(vocab),
    }


def main() -> None:
    t0 = time.time()
    # ---- data files held by the reference's own tests
    shutil.copyfile(REF / "tests/fixtures_gpt2/corpus.en", HERE / "corpus.en")
    shutil.copyfile(REF / "tests/fixtures_gpt2/train-bpe-reference-merges.txt", HERE / "g2_reference_merges_243.txt")
    shutil.copyfile(REF / "tests/_snapshots/test_train_bpe_special_tokens.pkl", HERE / "snapshot_special_tokens.pkl")
    (HERE / "data").mkdir(exist_ok=True)
    for n in ("simple", "empty", "unicode", "multiline", "sample"):
        shutil.copyfile(REF / f"tests/data/{n}.txt", HERE / "data" / f"{n}.txt")

    # ---- G1: corpus.en to exhaustion
    sp = ["<|endoftext|>"]
    seqs = ref_pretokens(REF / "tests/fixtures_gpt2/corpus.en", sp)
    vocab, merges = ref_merge_loop(seqs, 10 ** 6, 1, sp)
    (HERE / "g1_corpus_en_exhaustive.hex").write_text(hexlines(merges))
    print("G1 merges", len(merges), "vocab", len(vocab), f"{time.time()-t0:.1f}s")
    v1000, m1000 = ref_merge_loop(seqs, 1000, 1, sp)
    assert m1000 == merges[:743]
    (HERE / "g1_corpus_en_vocab_1000.json").write_text(json.dumps({k.hex(): v for k, v in v1000.items()}, indent=0))
    v2, m2 = ref_merge_loop(seqs, 10 ** 6, 2, sp)
    g1_meta = {"n_exhaustive": len(merges), "n_min_frequency_2": len(m2),
               "sha256": {str(k): hashlib.sha256(hexlines(merges[:k]).encode()).hexdigest()
                          for k in (243, 256, 512, 743, 1000, 2000, len(merges))}}
    assert m2 == merges[:len(m2)]
    (HERE / "g1_meta.json").write_text(json.dumps(g1_meta, indent=1))

    # ---- G3/G4 + random small trials
    cases = []
    sennrich = [b"low"] * 5 + [b"lower"] * 2 + [b"widest"] * 3 + [b"newest"] * 6
    cases.append(case("g3_sennrich", sennrich, ["<|endoftext|>"], 257 + 12, 1))
    cases.append(case("g4_ties", [b"ab", b"cd", b"\xff\x00", b"zz"], ["<|endoftext|>"], 259, 1))
    cases.append(case("g4_runs", [b"aaaa", b"aaa"], ["<|endoftext|>"], 300, 1))
    cases.append(case("g4_runs_long", [b"a" * 37, b"a" * 12, b"ba" * 9 + b"a" * 5], ["<|endoftext|>"], 300, 1))
    cases.append(case("g4_abab", [b"abab"] * 3 + [b"abc"] * 3, ["<|endoftext|>"], 300, 1))
    cases.append(case("g4_special_as_word", [b"<|x|>"] * 5 + [b"ab"] * 2, ["<|x|>"], 257 + 6, 1))
    cases.append(case("g4_single_byte_special", [b"ab", b"ab", b"a"], ["a", "[X]"], 300, 1))
    cases.append(case("g4_min_freq", [b"ab"], ["<|endoftext|>"], 300, 2))
    cases.append(case("g4_empty", [], ["[PAD]", "[UNK]", "[BOS]", "[EOS]"], 300, 1))
    cases.append(case("g4_single_tokens_only", [b"a", b"b", b"c"], ["<|endoftext|>"], 300, 1))
    cases.append(case("g4_vocab_limit", [b"AB", b"CD", b"EF", b"GH", b"IJ"] * 10, ["[PAD]", "[UNK]", "[BOS]", "[EOS]"], 262, 1))
    cases.append(case("g4_vocab_smaller_than_base", [b"abab"] * 3, ["<|endoftext|>"], 100, 1))
    cases.append(case("g4_long_word", [bytes((i * 7 + i // 5) % 5 + 97 for i in range(700))] * 2 + [b"abcde" * 3], ["<|endoftext|>"], 257 + 60, 1))
    cases.append(case("g4_long_run", [b" " * 1500, b"  ", b"x" + b" " * 301 + b"y"], ["<|endoftext|>"], 257 + 14, 1))
    cases.append(case("g4_highbytes", [bytes([250, 251, 252, 253, 254, 255] * 3), bytes([255, 255, 254, 254])] * 2, ["<|endoftext|>"], 290, 1))
    rng = random.Random(20260116)
    alphabets = [b"ab", b"abc", b"ab<|>x", bytes([250, 251, 252, 253, 254, 255, 97]), b"abcdefgh "]
    specials_opts = [[], ["<|x|>"], ["a", "<|>"], ["[PAD]", "[UNK]"], ["<|endoftext|>"]]
    for t in range(400):
        al = rng.choice(alphabets)
        ntypes = rng.randint(0, 8)
        words = []
        for _ in range(ntypes):
            w = bytes(rng.choice(al) for _ in range(rng.randint(1, 9)))
            words += [w] * rng.randint(1, 5)
        rng.shuffle(words)
        spc = rng.choice(specials_opts)
        base = 256 + len({s.encode() for s in spc if len(s.encode()) > 1})
        cases.append(case(f"rand_{t:03d}", words, spc, base + rng.randint(0, 25), rng.randint(1, 3)))
    (HERE / "g345_cases.json").write_text(json.dumps(cases, indent=0))
    print("cases", len(cases), f"{time.time()-t0:.1f}s")

    # ---- G5: config 2 synthetic (10 MiB, 1k merges) through the reference
    flat, off = synth.generate(synth.SynthSpec.config2())
    fb = flat.tobytes()
    o = off.tolist()
    seqs2 = [list(fb[o[i]:o[i + 1]]) for i in range(len(o) - 1)]
    t1 = time.time()
    v5, m5 = ref_merge_loop(seqs2, 257 + 1000, 1, ["<|endoftext|>"])
    t_ref = time.time() - t1
    (HERE / "g5_config2_merges_1000.hex").write_text(hexlines(m5))
    (HERE / "g5_meta.json").write_text(json.dumps({
        "n_words": len(seqs2), "n_bytes": len(fb), "corpus_sha256": hashlib.sha256(fb).hexdigest(),
        "offsets_sha256": hashlib.sha256(off.tobytes()).hexdigest(),
        "merges_sha256": hashlib.sha256(hexlines(m5).encode()).hexdigest(),
        "reference_merge_loop_seconds_here": round(t_ref, 2), "n_merges": len(m5)}, indent=1))
    print("G5", len(m5), f"ref loop {t_ref:.1f}s", f"{time.time()-t0:.1f}s")
    del seqs2

    # ---- G6: pre-tokenization pins
    g6 = {}
    files = {"corpus.en": REF / "tests/fixtures_gpt2/corpus.en"}
    for n in ("simple", "empty", "unicode", "multiline", "sample"):
        files[f"data/{n}.txt"] = REF / f"tests/data/{n}.txt"
    for label, path in files.items():
        for spname, spc in (("endoftext", ["<|endoftext|>"]), ("default4", ["[PAD]", "[UNK]", "[BOS]", "[EOS]"]), ("none", [])):
            toks = ref_pretokens(path, spc)
            h = hashlib.sha256()
            for tkn in toks:
                h.update(len(tkn).to_bytes(4, "little"))
                h.update(bytes(tkn))
            g6[f"{label}|{spname}"] = {"n": len(toks), "n_unique": len({bytes(x) for x in toks}), "sha256": h.hexdigest()}
    # chunked variant (tiny chunks) on the unicode/sample files: results depend on chunk_size_bytes (trainer.py:183-197)
    for label in ("data/unicode.txt", "data/sample.txt", "data/multiline.txt"):
        for cs in (5, 16, 64):
            toks = ref_pretokens(files[label], ["[PAD]", "[UNK]", "[BOS]", "[EOS]"], chunk_size_bytes=cs, max_workers=2)
            h = hashlib.sha256()
            for tkn in toks:
                h.update(len(tkn).to_bytes(4, "little"))
                h.update(bytes(tkn))
            g6[f"{label}|default4|chunk{cs}"] = {"n": len(toks), "sha256": h.hexdigest()}
    (HERE / "g6_pretokens.json").write_text(json.dumps(g6, indent=1))
    print("done", f"{time.time()-t0:.1f}s")


if __name__ == "__main__":
    main()
