"""GPU: the device pre-tokeniser (yabpe_pretokenize) through the C ABI against regex.findall with the
reference's pattern (trainer.py:163-167), the reference's pinned pre-token hashes (G6), Python's UnicodeDecodeError
positions, and train() end to end with the pre-tokeniser on the device vs on the host."""
from __future__ import annotations

import hashlib
import json
import os
import random

import numpy as np
import pytest
import regex

from tests import helpers
from tests.test_pretok_model import EDGE, GPT2, SPECIALS, regex_split

pytestmark = pytest.mark.gpu


def device_split(ctx, data: bytes, specials=(), chunk_starts=(0,)):
    dt, do, nw = ctx.pretokenize(data, chunk_starts=list(chunk_starts), special_tokens=specials)
    off = ctx.d2h(do, (nw + 1) * 8).view(np.uint64).tolist()
    ctx.pretokenize_free()
    assert off[-1] == len(data) and (nw == 0 or off[0] == 0)
    return [data[a:b] for a, b in zip(off[:-1], off[1:])]


def batch_check(ctx, strings, specials):
    """All strings in ONE buffer, each as a chunk of its own (chunks are separate texts)."""
    blobs = [s.encode("utf-8") for s in strings if s]
    starts, pos = [], 0
    for b in blobs:
        starts.append(pos)
        pos += len(b)
    data = b"".join(blobs)
    got = device_split(ctx, data, specials, starts)
    exp = regex_split(data, specials, starts)
    if got != exp:  # find the first differing string for the message
        for s in strings:
            g = device_split(ctx, s.encode("utf-8"), specials) if s else []
            assert g == regex_split(s.encode("utf-8"), specials), (s, specials, g)
    assert got == exp


def test_edge_cases_and_chains():
    from yet_another_bpe import _native

    chain_texts = ["<<<", "<<<<<", "a<<b", "<s>'sx", "<s>sx", "x<s>'s", "<s><s>", "it's", "it'sx", "abcab", "a b c", "<a>b<a>b>", "<a><a>b", "éeé",
                   "x  x x ", "''''s", "\n\n\nx", "1212 121", "it'it's s", "sxsx<s>sx", "<s>'s's", "<|endoftext|>a<|endoftext|><|endoftext|> b"]
    with _native.Context() as ctx:
        assert device_split(ctx, b"") == []
        for sp in SPECIALS:
            batch_check(ctx, EDGE + chain_texts, sp)


def test_random_strings():
    from yet_another_bpe import _native

    rng = random.Random(21)
    alphabets = ["ab '", "a1 .'s\n", "'stdmlvre x", " \t\n\r\x0b\x0c\x85  a1.", "<|endoftext|> a's", "<s>x' ", "é中\U0001F600a 1'", "[PAD][UNK] ab", "it's ", "<>"]
    with _native.Context() as ctx:
        for sp in SPECIALS + [["'s", "'"], ["<", "<<", "<<<"], ["aaa", "aa", "a"], [" ", "  "], ["a'll"]]:
            strings = ["".join(rng.choice(al) for _ in range(rng.randint(1, 40))) for al in (rng.choice(alphabets) for _ in range(1500))]
            batch_check(ctx, strings, sp)


def test_reference_pins_g6(golden_dir):
    from yet_another_bpe import _native
    from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig

    sps = {"endoftext": ["<|endoftext|>"], "default4": ["[PAD]", "[UNK]", "[BOS]", "[EOS]"], "none": []}
    g6 = json.loads((golden_dir / "g6_pretokens.json").read_text())
    with _native.Context() as ctx:
        for key, pin in g6.items():
            parts = key.split("|")
            path = golden_dir / parts[0]
            chunk = int(parts[2][5:]) if len(parts) > 2 else 1 << 30
            ranges = BBPETrainer(BBPETrainerConfig(chunk_size_bytes=chunk))._chunk_ranges(path)
            data = path.read_bytes()
            got = device_split(ctx, data, sps[parts[1]], [a for a, _ in ranges] or [0]) if data else []
            h = hashlib.sha256()
            for t in got:
                h.update(len(t).to_bytes(4, "little"))
                h.update(t)
            assert len(got) == pin["n"] and h.hexdigest() == pin["sha256"], key


def test_every_code_point():
    from yet_another_bpe import _native

    chars = [chr(c) for c in range(0x110000) if not 0xD800 <= c <= 0xDFFF]
    with _native.Context() as ctx:
        for lo in range(0, len(chars), 140000):
            s = "".join(f"a{ch}1 {ch}" for ch in chars[lo:lo + 140000]).encode("utf-8")
            assert device_split(ctx, s) == regex_split(s), lo


def test_invalid_utf8_position():
    from yet_another_bpe import _native

    bad = [b"\x80", b"a\x80", b"\xc3", b"a\xc3", b"\xc3(", b"\xe2\x82", b"\xe2\x82a", b"\xe2(\xa1", b"\xc0\xaf", b"\xe0\x80\xaf", b"\xed\xa0\x80",
           b"\xf0\x80\x80\x80", b"\xf4\x90\x80\x80", b"\xf5\x80\x80\x80", b"ok\xc3\xa9\xa9", b"\xf0\x9f\x98", b"abc\xff", b"\xc3\xa9\xc3", b"\xe4\xb8\xad\x80x"]
    rng = random.Random(3)
    bad += [bytes(rng.choice([0x61, 0x20, 0x80, 0xbf, 0xc2, 0xc3, 0xe0, 0xe2, 0xed, 0xf0, 0xf4, 0xa0, 0x90, 0x9f]) for _ in range(rng.randint(1, 8))) for _ in range(300)]
    with _native.Context() as ctx:
        for b in bad:
            try:
                b.decode("utf-8")
                exp = -1
            except UnicodeDecodeError as e:
                exp = e.start
            if exp < 0:
                assert b"".join(device_split(ctx, b)) == b
                continue
            with pytest.raises(_native.Utf8Error) as e:
                ctx.pretokenize(b)
            assert e.value.position == exp, (b, e.value.position, exp)
        # a long valid prefix, then the damage: the SMALLEST bad position is reported
        big = ("héllo wörld " * 100000).encode("utf-8")
        broken = big[:700001] + b"\xff" + big[700001:900000] + b"\xc3" + big[900000:]
        try:
            broken.decode("utf-8")
        except UnicodeDecodeError as err:
            with pytest.raises(_native.Utf8Error) as e:
                ctx.pretokenize(broken)
            assert e.value.position == err.start


def test_megabytes_of_text_and_byte_conservation():
    from yet_another_bpe import _native

    rng = random.Random(5)
    words = ["the", "of", "and", "it's", "don't", "we'll", "naïve", "東京", "42", "3.14", "e-mail", "<|endoftext|>", "\n", "\n\n", "  ", "\t", "!?", "'", "“quoted”", "I'm"]
    text = " ".join(rng.choice(words) for _ in range(700_000)).encode("utf-8")  # ~4 MB
    with _native.Context() as ctx:
        got = device_split(ctx, text, ["<|endoftext|>"])
        assert got == regex_split(text, ["<|endoftext|>"])
        cuts = [0, 1_000_003, 2_000_001, 3_000_000]
        cuts = [c if (text[c] & 0xC0) != 0x80 else c + 1 + ((text[c + 1] & 0xC0) == 0x80) for c in cuts]
        assert device_split(ctx, text, ["<|endoftext|>"], cuts) == regex_split(text, ["<|endoftext|>"], cuts)


def test_train_with_device_pretokenizer_equals_host(golden_dir, tmp_path, monkeypatch):
    from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig

    expected = helpers.read_hex_merges(golden_dir / "g1_corpus_en_exhaustive.hex")[:743]
    for chunk in (1 << 30, 4096):
        cfg = BBPETrainerConfig(vocab_size=1000, min_frequency=1, special_tokens=["<|endoftext|>"], chunk_size_bytes=chunk)
        monkeypatch.setenv("YABPE_PRETOKENIZE", "gpu")
        dev = BBPETrainer(cfg).train([golden_dir / "corpus.en"])
        monkeypatch.setenv("YABPE_PRETOKENIZE", "host")
        host = BBPETrainer(cfg).train([golden_dir / "corpus.en"])
        assert dev.merges == host.merges and dev.vocab == host.vocab
        if chunk == 1 << 30:
            assert dev.merges == expected  # reference-made golden merges (tests/golden/g1_*)
    # several files, one of them empty; flat layout
    monkeypatch.setenv("YABPE_PRETOKENIZE", "gpu")
    monkeypatch.setenv("YABPE_LAYOUT", "flat")
    files = [golden_dir / "data/sample.txt", golden_dir / "data/empty.txt", golden_dir / "data/unicode.txt"]
    cfg = BBPETrainerConfig(vocab_size=300, min_frequency=1)
    dev = BBPETrainer(cfg).train(files)
    monkeypatch.setenv("YABPE_PRETOKENIZE", "host")
    host = BBPETrainer(cfg).train(files)
    assert dev.merges == host.merges and dev.vocab == host.vocab
    # errors keep the reference's wording (trainer.py:158-161)
    badf = tmp_path / "bad.txt"
    badf.write_bytes(b"fine text " * 10 + b"\xff\xfe oops")
    monkeypatch.setenv("YABPE_PRETOKENIZE", "gpu")
    with pytest.raises(ValueError, match=r"contains invalid UTF-8 at position 100\."):
        BBPETrainer(cfg).train([badf])
    with pytest.raises(FileNotFoundError):
        BBPETrainer(cfg).train([tmp_path / "missing.txt"])
    empty = BBPETrainer(cfg).train([golden_dir / "data/empty.txt"])
    assert empty.merges == [] and len(empty.vocab) == 260
