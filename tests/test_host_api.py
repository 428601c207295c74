"""CPU: host-side logic that mirrors the reference interface -- config defaults, pre-tokenisation (pinned against
reference outputs, G6), chunking, error behaviour, save format, tokenizer.  No GPU calls."""
from __future__ import annotations

import hashlib
import json
from pathlib import Path

import pytest

from oracle import oracle
from tests import helpers
from yet_another_bpe import BBPEModel, BBPETokenizer, BBPETrainer, BBPETrainerConfig

DATA = helpers.GOLDEN / "data"
SPS = {"endoftext": ["<|endoftext|>"], "default4": ["[PAD]", "[UNK]", "[BOS]", "[EOS]"], "none": []}


def test_config_defaults():
    c = BBPETrainerConfig()  # reference trainer.py:31-38
    assert (c.vocab_size, c.min_frequency, c.max_workers, c.chunk_size_bytes, c.seed) == (32000, 2, 8, 8 * 1024 * 1024, 42)
    assert list(c.special_tokens) == ["[PAD]", "[UNK]", "[BOS]", "[EOS]"]
    t = BBPETrainer()
    assert t._vocab == {} and t._merges == [] and isinstance(t.config, BBPETrainerConfig)


def test_base_vocab_rules():
    t = BBPETrainer(BBPETrainerConfig(special_tokens=["a", "[X]", "[X]"]))
    v = t._init_base_vocab()
    assert len(v) == 257 and v[b"[X]"] == 256 and v[b"a"] == 97  # existing bytes get no new id (trainer.py:130)


def test_pretokenizer_matches_reference_pins(golden_dir):
    g6 = json.loads((golden_dir / "g6_pretokens.json").read_text())
    for key, pin in g6.items():
        parts = key.split("|")
        chunk = int(parts[2][5:]) if len(parts) > 2 else 1 << 30
        t = BBPETrainer(BBPETrainerConfig(max_workers=2 if len(parts) > 2 else 1, chunk_size_bytes=chunk, special_tokens=SPS[parts[1]]))
        toks = t._preprocess_corpus([golden_dir / parts[0]])
        h = hashlib.sha256()
        for s in toks:
            assert isinstance(s, list) and all(isinstance(b, int) and 0 <= b <= 255 for b in s)
            h.update(len(s).to_bytes(4, "little"))
            h.update(bytes(s))
        assert len(toks) == pin["n"] and h.hexdigest() == pin["sha256"], key


def test_preprocess_byte_conservation_and_workers():
    for name in ("simple", "unicode", "multiline", "sample"):
        raw = (DATA / f"{name}.txt").read_bytes()
        a = BBPETrainer(BBPETrainerConfig(max_workers=1))._preprocess_corpus([DATA / f"{name}.txt"])
        b = BBPETrainer(BBPETrainerConfig(max_workers=4))._preprocess_corpus([DATA / f"{name}.txt"])
        assert b"".join(bytes(s) for s in a) == raw and a == b
    assert BBPETrainer()._preprocess_corpus([DATA / "empty.txt"]) == []


def test_errors(tmp_path):
    t = BBPETrainer()
    with pytest.raises(ValueError, match="At least one file"):
        t.train([])
    with pytest.raises(FileNotFoundError):
        t._preprocess_corpus([tmp_path / "nope.txt"])
    bad = tmp_path / "bad.txt"
    bad.write_bytes(b"ok \xff\xfe bad")
    with pytest.raises(ValueError, match="invalid UTF-8"):
        t._preprocess_corpus([bad])
    with pytest.raises(ValueError, match="not been trained"):
        t.save(tmp_path / "out")


def test_empty_inputs_need_no_gpu():
    t = BBPETrainer(BBPETrainerConfig(vocab_size=300))
    vocab, merges = t._merge_loop([])
    assert len(vocab) == 260 and merges == []
    m = t.train([DATA / "empty.txt"])
    assert isinstance(m, BBPEModel) and len(m.vocab) == 260 and m.merges == []
    t2 = BBPETrainer(BBPETrainerConfig(vocab_size=100))  # vocab_size below the base vocab: zero merges
    assert t2._merge_loop([[65, 66]])[1] == []


def _trained_by_oracle(vocab_size=400, specials=("<|endoftext|>",)):
    vocab, merges = oracle.merge_loop(helpers.corpus_en_words(), vocab_size, 1, list(specials))
    t = BBPETrainer(BBPETrainerConfig(vocab_size=vocab_size, special_tokens=list(specials)))
    t._vocab, t._merges = vocab, merges
    return t


def test_save_format_and_tokenizer_roundtrip(tmp_path):
    t = _trained_by_oracle()
    t.save(tmp_path / "model")
    vj = json.loads((tmp_path / "model" / "vocab.json").read_text(encoding="utf-8"))
    assert vj == {k.decode("latin-1"): v for k, v in t._vocab.items()}
    lines = (tmp_path / "model" / "merges.txt").read_text(encoding="utf-8").split("\n")
    assert lines[0] == f"{t._merges[0][0].decode('latin-1')} {t._merges[0][1].decode('latin-1')}"
    assert json.loads((tmp_path / "model" / "special_tokens.json").read_text()) == ["<|endoftext|>"]
    tok = BBPETokenizer.from_file(tmp_path / "model")
    assert tok.vocab_size == len(t._vocab) and tok.special_tokens == ["<|endoftext|>"]
    direct = BBPETokenizer(vocab=t._vocab, merges=t._merges, special_tokens=["<|endoftext|>"])
    text = "The quick brown fox<|endoftext|>jumps over the lazy dog. Ünïcödé ok"
    ids = direct.encode(text)
    assert direct.decode(ids) == text
    assert t._vocab[b"<|endoftext|>"] in ids
    assert direct.encode_batch([text, ""]) == [ids, []]
    assert direct.decode_batch([ids]) == [text]
    assert direct.decode([10 ** 9]) == ""  # unknown ids are skipped
    assert "hits=" in direct.cache_info()
    direct.clear_cache()
    assert direct._encode_word(" the") == list(direct._word_ids(" the"))
    assert set(direct.get_vocab().values()) == set(t._vocab.values())


def test_tokenizer_applies_merges_in_rank_order():
    vocab = {bytes([b]): b for b in range(256)}
    merges = [(b"a", b"a"), (b"aa", b"a"), (b"b", b"c")]
    for l, r in merges:
        vocab[l + r] = len(vocab)
    tok = BBPETokenizer(vocab=vocab, merges=merges)
    assert tok._encode_word("aaaa") == [vocab[b"aa"], vocab[b"aa"]]
    assert tok._encode_word("aaa") == [vocab[b"aaa"]]
    assert tok._encode_word("abc") == [97, vocab[b"bc"]]
