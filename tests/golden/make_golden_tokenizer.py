#!/usr/bin/env python3
"""G9 -- pins for the two steps AFTER the hot path, made by RUNNING THE REFERENCE in the build container:
  * the on-disk model format: SHA-256 of vocab.json / merges.txt / special_tokens.json as the reference's
    BBPETrainer.save() writes them (trainer.py:94-117) for two models it trained here;
  * the tokenizer: BBPETokenizer.encode / decode (tokenizer.py:152-349) on ~70 texts, through from_file (so the lossy
    first-space reload of tokenizer.py:137 is part of the expected ids) and through the in-memory constructor, with
    special tokens (split longest first, :100-102), a vocabulary that lacks bytes (the [UNK] / id-0 fallback, :216, :298),
    CRLF, emoji, whitespace runs, contractions.
Only this script and its output (data: inputs + expected outputs) are committed.

    cd /root/repo && python tests/golden/make_golden_tokenizer.py     # writes tests/golden/g9_tokenizer.json
"""
from __future__ import annotations

import hashlib
import json
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF / "src"))

from yet_another_bpe.tokenizer import BBPETokenizer  # noqa: E402  (the REFERENCE)
from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig  # noqa: E402

assert "/root/reference/" in sys.modules["yet_another_bpe.tokenizer"].__file__

TEXTS = [
    "", " ", "  ", "a", "Hello world!", "Hello, world! How are you?", "the quick brown fox jumps over the lazy dog",
    "I'm sure it's they'll we've you're he'd 'tis", "numbers 12345 and 3.14159, 1e-9; 2026-10-04", "tabs\tand\nnewlines\n\nand  double  spaces   ",
    "line one\r\nline two\r\n\r\nline four", "trailing space ", " leading space", "\n", "\r\n", "   \n   ", "a b non-breaking",
    "naïve café über straße", "你好，世界", "こんにちは", "emoji \U0001f600\U0001f680 and \U0001f469‍\U0001f4bb zwj",
    "mixedé\U0001f600x", "<|endoftext|>", "a<|endoftext|>b", "<|endoftext|><|endoftext|>", "x <|endoftext|> y", "<|endoftext", "endoftext|>", "<|endoftext|> <|endoftext|>",
    "[PAD]", "[UNK] unknown [BOS]start[EOS]", "brackets [not special] [UNK]x", "price: $12.50 (approx.) -- ok?", "e-mail: someone@example.com; url https://example.org/a?b=c&d=e",
    "CamelCaseWordsAndsnake_case_words", "ALL CAPS TEXT", "rep rep rep rep rep rep rep rep", "aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa", "abababababababab", "zzz", "z", "qqqqzqqqq",
    "The merger of the two companies was announced on Monday.", "In the beginning the Universe was created. This has made a lot of people very angry.",
    "don't can't won't shouldn't", "'s 't 're 've 'm 'll 'd", " 's", "x's", "'", "''", "a'b", "1st 2nd 3rd 4th", "100%", "#hashtag @mention", "½ ² ① numerals",
    "السلام عليكم", "Привет, мир!", "한국어 텍스트", "control \x00\x01\x02 chars", "\x7f del", "tab\tseparated\tvalues",
    "a" * 200, " " * 50 + "x", "word " * 40, "\n".join(f"line {i}" for i in range(12)), "<|x|><|y|>", "<|x|><|y|> then <|x|> then <|y|>", "<|x|", "|x|>", "<|y|><|x|>",
]


def sha(p: Path) -> str:
    return hashlib.sha256(p.read_bytes()).hexdigest()


def model_block(name: str, trainer: BBPETrainer, extra_tokenizers: dict) -> dict:
    with tempfile.TemporaryDirectory() as d:
        out = Path(d) / "model"
        trainer.save(str(out))
        files = {n: sha(out / n) for n in ("vocab.json", "merges.txt", "special_tokens.json")}
        sizes = {n: (out / n).stat().st_size for n in files}
        tok_file = BBPETokenizer.from_file(out)
    tok_mem = BBPETokenizer(vocab=dict(trainer._vocab), merges=list(trainer._merges), special_tokens=list(trainer.config.special_tokens))
    toks = {"from_file": tok_file, "in_memory": tok_mem, **extra_tokenizers}
    block = {"name": name, "save_sha256": files, "save_bytes": sizes, "n_vocab": len(trainer._vocab), "n_merges": len(trainer._merges),
             "special_tokens": list(trainer.config.special_tokens),
             "merges_reloaded_differ": sum(1 for a, b in zip(tok_file._merges, trainer._merges) if a != b),
             "n_merges_reloaded": len(tok_file._merges), "encode": {}}
    for tname, tok in toks.items():
        rows = []
        for t in TEXTS:
            ids = tok.encode(t)
            rows.append({"ids": ids, "decoded": tok.decode(ids)})
        block["encode"][tname] = rows
        block.setdefault("tokenizer_specials", {})[tname] = tok.special_tokens
        block.setdefault("tokenizer_vocab_size", {})[tname] = tok.vocab_size
    return block


def main() -> None:
    out = {"texts": TEXTS, "models": []}
    # model A: corpus.en, vocab 1000, ["<|endoftext|>"] (the G1@1000 model)
    cfg = BBPETrainerConfig(vocab_size=1000, min_frequency=1, max_workers=1, chunk_size_bytes=1 << 30, special_tokens=["<|endoftext|>"])
    ta = BBPETrainer(cfg)
    ta.train([REF / "tests/fixtures_gpt2/corpus.en"])
    # + the same vocabulary with extra special tokens where one is a prefix of another (longest first), ids appended
    va = dict(ta._vocab)
    for s in ("<|x|>", "<|x|><|y|>", "<|y|>"):
        va[s.encode()] = len(va)
    longest = BBPETokenizer(vocab=va, merges=list(ta._merges), special_tokens=["<|endoftext|>", "<|x|>", "<|x|><|y|>", "<|y|>"])
    out["models"].append(model_block("corpus_en_1000", ta, {"longest_first_specials": longest}))
    out["models"][-1]["longest_first_specials_extra_ids"] = {s: va[s.encode()] for s in ("<|x|>", "<|x|><|y|>", "<|y|>")}
    # model B: the reference's default special tokens ([PAD] [UNK] [BOS] [EOS]) on tests/data/sample.txt, vocab 300
    cfg = BBPETrainerConfig(vocab_size=300, min_frequency=1, max_workers=1)
    tb = BBPETrainer(cfg)
    tb.train([REF / "tests/data/sample.txt"])
    # + a vocabulary that LACKS some bytes: with [UNK] present its id is the fallback; without it, id 0
    lacking = {k: v for k, v in tb._vocab.items() if k not in (b"z", b"q", b"\xc3", b"\xf0")}
    unk = BBPETokenizer(vocab=lacking, merges=list(tb._merges), special_tokens=list(cfg.special_tokens))
    no_unk = BBPETokenizer(vocab={k: v for k, v in lacking.items() if k != b"[UNK]"}, merges=list(tb._merges), special_tokens=[])
    out["models"].append(model_block("sample_300_default_specials", tb, {"lacking_bytes_with_unk": unk, "lacking_bytes_no_unk": no_unk}))
    out["models"][-1]["lacking_removed"] = [b.hex() for b in (b"z", b"q", b"\xc3", b"\xf0")]
    (HERE / "g9_tokenizer.json").write_text(json.dumps(out, ensure_ascii=True, indent=0))
    for m in out["models"]:
        print(m["name"], m["save_sha256"], "reloaded merges that differ:", m["merges_reloaded_differ"], "of", m["n_merges"], "->", m["n_merges_reloaded"])


if __name__ == "__main__":
    main()
