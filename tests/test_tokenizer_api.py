"""CPU: the reference's tokenizer tests (reference tests/test_tokenizer.py:16-522), restated against this package's
BBPETokenizer.  The reference trains its fixtures with BBPETrainer.train(); training is the GPU path here, so on the CPU the
same models come from the pinned oracle (host pre-tokenisation + oracle/bpe_oracle.c) and go through this package's save() /
from_file() -- tests/test_gpu_api.py::TestTokenizerOnTrainedModel repeats the round trips on a model the GPU trained.
Exact ids are pinned separately, against vectors made by the reference (tests/test_tokenizer_golden.py)."""
from __future__ import annotations

import json

import pytest

from oracle import oracle
from tests import helpers
from yet_another_bpe.tokenizer import BBPETokenizer
from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig

DATA = helpers.GOLDEN / "data"


def saved_model(tmp_path, source: str, vocab_size: int, special_tokens=None):
    """What `trainer.train([file]); trainer.save(dir)` leaves on disk (reference tests' fixture, e.g. :138-147)."""
    kw = {} if special_tokens is None else {"special_tokens": special_tokens}
    t = BBPETrainer(BBPETrainerConfig(vocab_size=vocab_size, min_frequency=1, max_workers=1, **kw))
    words = [bytes(s) for s in t._preprocess_corpus([DATA / source])]
    t._vocab, t._merges = oracle.merge_loop(words, vocab_size, 1, list(t.config.special_tokens))
    t.save(tmp_path / "model")
    return tmp_path / "model"


@pytest.fixture
def trained_tokenizer(tmp_path):
    return BBPETokenizer.from_file(saved_model(tmp_path, "multiline.txt", 300))


class TestInit:  # reference :16-47
    def test_empty(self):
        tok = BBPETokenizer()
        assert tok.vocab_size == 0 and tok.special_tokens == []

    def test_vocab_merges_specials(self):
        vocab = {b"a": 0, b"b": 1, b"ab": 2}
        assert BBPETokenizer(vocab=vocab).get_vocab() == {"a": 0, "b": 1, "ab": 2}
        assert BBPETokenizer(vocab=vocab, merges=[(b"a", b"b")]).vocab_size == 3
        tok = BBPETokenizer(special_tokens=["[PAD]", "[UNK]"])
        assert tok.special_tokens == ["[PAD]", "[UNK]"]
        tok.special_tokens.append("x")  # a copy is handed out (tokenizer.py special_tokens property)
        assert tok.special_tokens == ["[PAD]", "[UNK]"]


class TestFromFile:  # reference :50-132
    def test_trained_model_loads(self, tmp_path):
        d = saved_model(tmp_path, "simple.txt", 270)
        tok = BBPETokenizer.from_file(d)
        assert tok.vocab_size >= 260 and isinstance(tok.special_tokens, list)
        assert tok.get_vocab() == json.loads((d / "vocab.json").read_text(encoding="utf-8"))

    def test_special_tokens_file(self, tmp_path):
        tok = BBPETokenizer.from_file(saved_model(tmp_path, "simple.txt", 270, ["<|endoftext|>", "<|pad|>"]))
        assert "<|endoftext|>" in tok.special_tokens and "<|pad|>" in tok.special_tokens

    def test_without_special_tokens_file(self, tmp_path):
        d = tmp_path / "m"
        d.mkdir()
        (d / "vocab.json").write_text(json.dumps({chr(i): i for i in range(256)}), encoding="utf-8")
        (d / "merges.txt").write_text("")
        assert BBPETokenizer.from_file(d).special_tokens == []

    def test_missing_directory(self):
        with pytest.raises(FileNotFoundError):
            BBPETokenizer.from_file("/nonexistent/path")


class TestEncodeDecode:  # reference :135-249
    def test_empty(self, trained_tokenizer):
        assert trained_tokenizer.encode("") == [] and trained_tokenizer.decode([]) == ""

    @pytest.mark.parametrize("text", ["hello", "Hello, world!", "你好世界", "hello world", "Hello! How are you? Fine, thanks.", "Line 1\nLine 2\nLine 3"])
    def test_ids_are_valid(self, trained_tokenizer, text):
        ids = trained_tokenizer.encode(text)
        assert isinstance(ids, list) and len(ids) > 0 and all(isinstance(i, int) and 0 <= i < trained_tokenizer.vocab_size for i in ids)

    def test_unknown_id_is_skipped(self, trained_tokenizer):
        assert isinstance(trained_tokenizer.decode([999999]), str)
        assert trained_tokenizer.decode([999999] + trained_tokenizer.encode("a")) == "a"


class TestRoundtrip:  # reference :252-320, :462-522
    @pytest.mark.parametrize("text", ["a", "hello", "Hello, world!", "你好世界 Hello 안녕하세요", "The answer is 42.", "Hello! How are you? I'm fine, thanks.",
                                      "Line 1\nLine 2\nLine 3", "Hello   world\t\ttabs", "", "Hello world! " * 1000, "   \t\t\n\n   ", "aaaaaaaaaa",
                                      "Hello 你好 مرحبا Привет", "Hello \U0001f44b World \U0001f30d"])
    def test_text_comes_back(self, trained_tokenizer, text):
        assert trained_tokenizer.decode(trained_tokenizer.encode(text)) == text

    def test_single_characters(self, trained_tokenizer):
        for ch in "abcXYZ123!@#":
            assert trained_tokenizer.decode(trained_tokenizer.encode(ch)) == ch


class TestBatch:  # reference :324-388
    def test_batches(self, trained_tokenizer):
        assert trained_tokenizer.encode_batch([]) == [] and trained_tokenizer.decode_batch([]) == []
        texts = ["hello", "world", "test"]
        enc = trained_tokenizer.encode_batch(texts)
        assert enc == [trained_tokenizer.encode(t) for t in texts]
        assert trained_tokenizer.encode_batch(["hello"]) == [trained_tokenizer.encode("hello")]
        assert trained_tokenizer.decode_batch([trained_tokenizer.encode("hello")]) == ["hello"]
        assert trained_tokenizer.decode_batch(enc) == texts
        other = ["Hello!", "World!", "Test 123"]
        assert trained_tokenizer.decode_batch(trained_tokenizer.encode_batch(other)) == other


class TestSpecialTokens:  # reference :391-459
    def test_special_token_is_one_id_and_round_trips(self, tmp_path):
        tok = BBPETokenizer.from_file(saved_model(tmp_path, "simple.txt", 270, ["<|endoftext|>"]))
        sid = tok.get_vocab()["<|endoftext|>"]
        ids = tok.encode("Hello<|endoftext|>World")
        assert ids.count(sid) == 1
        assert tok.decode(ids) == "Hello<|endoftext|>World"


class TestCache:  # tokenizer.py: clear_cache / cache_info / _encode_word
    def test_cache_counts_and_word_encoder(self, trained_tokenizer):
        trained_tokenizer.clear_cache()
        trained_tokenizer.encode("hello hello hello")
        assert trained_tokenizer.cache_info().startswith("hits=")
        assert "size=" in trained_tokenizer.cache_info() and "/8192" in trained_tokenizer.cache_info()
        assert trained_tokenizer._encode_word(" hello") == trained_tokenizer.encode(" hello")
        trained_tokenizer.clear_cache()
        assert "size=0/" in trained_tokenizer.cache_info()
