"""CPU: the two steps after the hot path against vectors made by RUNNING THE REFERENCE (tests/golden/make_golden_tokenizer.py,
G9): the on-disk model format of BBPETrainer.save() (reference trainer.py:94-117) by file digest, and
BBPETokenizer.encode / decode (tokenizer.py:152-349) id for id -- through from_file (the lossy first-space reload of
tokenizer.py:137 included), in memory, with special tokens split longest first, and with a vocabulary that lacks bytes
([UNK] / id-0 fallback).  Plus the lossless format offered next to the reference's."""
from __future__ import annotations

import hashlib
import json

import pytest

from oracle import oracle
from tests import helpers
from yet_another_bpe.tokenizer import BBPETokenizer
from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig


@pytest.fixture(scope="module")
def g9(golden_dir):
    return json.loads((golden_dir / "g9_tokenizer.json").read_text())


def _trainer_with_model(name: str, golden_dir) -> BBPETrainer:
    """A trainer object holding the model the reference trained (from committed fixtures / the pinned oracle; no GPU)."""
    if name == "corpus_en_1000":
        cfg = BBPETrainerConfig(vocab_size=1000, min_frequency=1, max_workers=1, special_tokens=["<|endoftext|>"])
        t = BBPETrainer(cfg)
        t._vocab = {bytes.fromhex(k): v for k, v in json.loads((golden_dir / "g1_corpus_en_vocab_1000.json").read_text()).items()}
        t._merges = helpers.read_hex_merges(golden_dir / "g1_corpus_en_exhaustive.hex")[:743]
        return t
    cfg = BBPETrainerConfig(vocab_size=300, min_frequency=1, max_workers=1)
    t = BBPETrainer(cfg)
    words = [bytes(s) for s in t._preprocess_corpus([golden_dir / "data" / "sample.txt"])]
    t._vocab, t._merges = oracle.merge_loop(words, 300, 1, list(cfg.special_tokens))
    return t


@pytest.mark.parametrize("idx", [0, 1])
def test_save_format_digests_and_encode_vectors(g9, golden_dir, tmp_path, idx):
    m = g9["models"][idx]
    t = _trainer_with_model(m["name"], golden_dir)
    assert len(t._vocab) == m["n_vocab"] and len(t._merges) == m["n_merges"]
    t.save(tmp_path / "model")
    for fname, digest in m["save_sha256"].items():  # byte for byte what the reference's save() wrote
        data = (tmp_path / "model" / fname).read_bytes()
        assert len(data) == m["save_bytes"][fname], fname
        assert hashlib.sha256(data).hexdigest() == digest, fname
    toks = {"from_file": BBPETokenizer.from_file(tmp_path / "model"),
            "in_memory": BBPETokenizer(vocab=dict(t._vocab), merges=list(t._merges), special_tokens=list(t.config.special_tokens))}
    assert len(toks["from_file"]._merges) == m["n_merges_reloaded"]
    assert sum(1 for a, b in zip(toks["from_file"]._merges, t._merges) if a != b) == m["merges_reloaded_differ"]  # the lossy reload, reproduced
    if m["name"] == "corpus_en_1000":
        va = dict(t._vocab)
        for s, i in m["longest_first_specials_extra_ids"].items():
            va[s.encode()] = i
        toks["longest_first_specials"] = BBPETokenizer(vocab=va, merges=list(t._merges), special_tokens=["<|endoftext|>", "<|x|>", "<|x|><|y|>", "<|y|>"])
    else:
        removed = {bytes.fromhex(h) for h in m["lacking_removed"]}
        lacking = {k: v for k, v in t._vocab.items() if k not in removed}
        toks["lacking_bytes_with_unk"] = BBPETokenizer(vocab=lacking, merges=list(t._merges), special_tokens=list(t.config.special_tokens))
        toks["lacking_bytes_no_unk"] = BBPETokenizer(vocab={k: v for k, v in lacking.items() if k != b"[UNK]"}, merges=list(t._merges), special_tokens=[])
    assert set(toks) == set(m["encode"])
    for name, tok in toks.items():
        assert tok.special_tokens == m["tokenizer_specials"][name] and tok.vocab_size == m["tokenizer_vocab_size"][name]
        for text, row in zip(g9["texts"], m["encode"][name]):
            ids = tok.encode(text)
            assert ids == row["ids"], (name, text)
            assert tok.decode(ids) == row["decoded"], (name, text)
        assert tok.encode_batch(g9["texts"][:9]) == [r["ids"] for r in m["encode"][name][:9]]
        assert tok.decode_batch([r["ids"] for r in m["encode"][name][:9]]) == [r["decoded"] for r in m["encode"][name][:9]]


def test_lossless_format_round_trips_what_the_reference_format_loses(golden_dir, tmp_path):
    t = _trainer_with_model("corpus_en_1000", golden_dir)
    t.save(tmp_path / "ref")
    t.save_lossless(tmp_path / "hex")
    lossy, exact = BBPETokenizer.from_file(tmp_path / "ref"), BBPETokenizer.from_file_lossless(tmp_path / "hex")
    assert exact._merges == t._merges and exact._vocab == t._vocab and exact.special_tokens == ["<|endoftext|>"]
    assert lossy._merges != t._merges  # every merge whose left token holds a space comes back split elsewhere (tokenizer.py:137)
    text = "the merger of the two companies was announced"
    mem = BBPETokenizer(vocab=dict(t._vocab), merges=list(t._merges), special_tokens=["<|endoftext|>"])
    assert exact.encode(text) == mem.encode(text)
    assert len(lossy.encode(text)) > len(exact.encode(text))  # merges lost -> longer encodings, same text back
    assert lossy.decode(lossy.encode(text)) == exact.decode(exact.encode(text)) == text
    with pytest.raises(ValueError, match="not been trained"):
        BBPETrainer(BBPETrainerConfig()).save_lossless(tmp_path / "x")
