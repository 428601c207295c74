"""CPU: the pre-tokeniser's per-position rules (yet-another-bpe_amd/csrc/pretok_logic.h, the functions the HIP kernels
call) against regex.findall with the reference's pattern (trainer.py:163-167) -- on the reference's pinned pre-token
fixtures (tests/golden/g6_pretokens.json), on hand-written edge cases and on random / adversarial strings."""
from __future__ import annotations

import ctypes
import hashlib
import json
import random
import subprocess
from pathlib import Path

import numpy as np
import pytest
import regex

HM = Path(__file__).resolve().parent / "hostmodel"
GOLD = Path(__file__).resolve().parent / "golden"
GPT2 = r"""'(?:[sdmt]|ll|ve|re)| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""


@pytest.fixture(scope="module")
def model():
    so, src = HM / "libpretok_model.so", HM / "pretok_model.cpp"
    csrc = HM.parent.parent / "yet-another-bpe_amd/csrc"
    deps = [src, csrc / "pretok_logic.h", csrc / "unicode_classes.inc"]
    if not so.exists() or so.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    lib.pretok_model.restype = ctypes.c_int
    return lib


def model_split(lib, data: bytes, specials=(), chunk_starts=(0,)):
    """-> (list of pre-token byte strings, error position or -1)"""
    text = np.frombuffer(data, dtype=np.uint8).copy() if data else np.zeros(1, np.uint8)
    n = len(data)
    ch = np.asarray(list(chunk_starts) + [n], dtype=np.uint64)
    sb = [s.encode("utf-8") for s in specials]
    spb = np.frombuffer(b"".join(sb) or b"\0", dtype=np.uint8).copy()
    spo = np.zeros(len(sb) + 1, dtype=np.uint32)
    if sb:
        spo[1:] = np.cumsum([len(x) for x in sb])
    flags = np.zeros(max(n, 1), dtype=np.uint8)
    err = ctypes.c_int64(-1)
    vp = ctypes.c_void_p
    lib.pretok_model(vp(text.ctypes.data), ctypes.c_uint64(n), vp(ch.ctypes.data), ctypes.c_uint32(len(ch) - 1), vp(spb.ctypes.data),
                     vp(spo.ctypes.data), ctypes.c_uint32(len(sb)), vp(flags.ctypes.data), ctypes.byref(err))
    if err.value >= 0:
        return None, err.value
    cuts = np.flatnonzero(flags[:n]).tolist() + [n]
    return [data[a:b] for a, b in zip(cuts[:-1], cuts[1:])], -1


def regex_split(data: bytes, specials=(), chunk_starts=(0,)):
    """The oracle of this step: regex.findall with the reference's pattern (oracle/pretok.py)."""
    from oracle import pretok

    return pretok.pretokenize(data, specials, chunk_starts)


EDGE = [
    "", "a", " ", "  ", "a b", "a  b", "a   b", "a \n b", "a\tb", "a \tb", "a\t b", "a  ", "\n\nhello", "  hello", "hello world  ",
    "don't", "don'ts", "it's", "'tis", "'llama", "we'll've're", " 's", "!'s", "x's'd", "'s's's", "'S", "a'b", "1's", "\n's", " \n's",
    "abc123", "abc 123", "abc  123", "3.14", "a.b", "a. b", "a .b", " !b", "!!", " !!", "a b", "a  b", "a  b",
    "été l'été", "中文 测试", "१२३", "x́y", "\U0001F600 ok", "a\U0001F600b", "²³",
    "'", "''", "'s", "'ll", "'l", "'lx", "'ve", "'v", "'re'd", "a'", "a 'll", "\t'll", "e'er", "'ll'll", "'t't", " '", "  '",
    "\r\n", "a\r\nb", "a\r\n\r\nb", "\x0bx", "\x1cx", "\x85x", " \x85 x", "x \n", "x\n ", "tab\there", "end.\n", "0 1  2   3",
]
SPECIALS = [[], ["<|endoftext|>"], ["[PAD]", "[UNK]", "[BOS]", "[EOS]"], ["<<"], ["<s>", "sx", "'s"], ["ab", "abc", "b"], ["it'", "s"],
            ["a b", " "], ["<a>", "<a>b", ">"], ["é", "e"], ["x ", " x"], ["''"], ["\n"], ["12", "1"]]


def test_edge_cases(model):
    for s in EDGE:
        for sp in SPECIALS[:3]:
            got, err = model_split(model, s.encode("utf-8"), sp)
            assert err == -1 and got == regex_split(s.encode("utf-8"), sp), (s, sp, got)


def test_special_token_chains(model):
    texts = ["<<<", "<<<<", "<<<<<", "a<<b", "<s>'sx", "<s>sx", "x<s>'s", "<s><s>", "<s> <s>", "it's", "it'sx", "xit's", "abcab", "ab abc",
             "a b c", "a  b", "<a>b<a>b>", "<a><a>b", "éeé", "x  x x ", "''''s", "'''", "\n\n\nx", "a\nb", "1212 121", "it'it's s",
             "sxsx<s>sx", "<s>'s's", "abcabcb", "bab", "a bb", " a b", "x x", "<|endoftext|>a<|endoftext|><|endoftext|> b"]
    for s in texts:
        for sp in SPECIALS:
            got, err = model_split(model, s.encode("utf-8"), sp)
            assert err == -1 and got == regex_split(s.encode("utf-8"), sp), (s, sp, got, regex_split(s.encode("utf-8"), sp))


def test_random_strings(model):
    rng = random.Random(7)
    alphabets = ["ab '", "a1 .'s\n", "'stdmlvre x", " \t\n\r\x0b\x0c\x85  a1.", "<|endoftext|> a's", "<s>x' ", "é中\U0001F600a 1'",
                 "[PAD][UNK] ab", "ab c", "it's "]
    for trial in range(4000):
        al = rng.choice(alphabets)
        s = "".join(rng.choice(al) for _ in range(rng.randint(0, 24)))
        sp = rng.choice(SPECIALS)
        got, err = model_split(model, s.encode("utf-8"), sp)
        assert err == -1 and got == regex_split(s.encode("utf-8"), sp), (s, sp, got, regex_split(s.encode("utf-8"), sp))


def test_chunks_are_separate_texts(model):
    rng = random.Random(9)
    for trial in range(500):
        s = "".join(rng.choice("ab '\n<s>xé") for _ in range(rng.randint(2, 40))).encode("utf-8")
        cuts = sorted({0} | {c for c in (rng.randrange(1, len(s)) for _ in range(rng.randint(0, 3))) if (s[c] & 0xC0) != 0x80})
        sp = rng.choice([[], ["<s>"], ["ab", "'s"]])
        got, err = model_split(model, s, sp, cuts)
        assert err == -1 and got == regex_split(s, sp, cuts), (s, sp, cuts)


def test_unicode_classes_everywhere(model):
    """Every code point next to a letter, a digit and a space: pins the generated class table against the regex module."""
    chars = [chr(c) for c in range(0x110000) if not 0xD800 <= c <= 0xDFFF]
    for lo in range(0, len(chars), 20000):
        s = "".join(f"a{ch}1 {ch}" for ch in chars[lo:lo + 20000]).encode("utf-8")
        got, err = model_split(model, s)
        assert err == -1 and got == regex_split(s), lo


def test_invalid_utf8_position(model):
    bad = [b"\x80", b"a\x80", b"\xc3", b"a\xc3", b"\xc3(", b"\xe2\x82", b"\xe2\x82a", b"\xe2(\xa1", b"\xc0\xaf", b"\xe0\x80\xaf", b"\xed\xa0\x80",
           b"\xf0\x80\x80\x80", b"\xf4\x90\x80\x80", b"\xf5\x80\x80\x80", b"ok\xc3\xa9\xa9", b"\xf0\x9f\x98", b"abc\xff", b"\xc3\xa9\xc3", b"\xe4\xb8\xad\x80x"]
    for b in bad:
        with pytest.raises(UnicodeDecodeError) as e:
            b.decode("utf-8")
        got, err = model_split(model, b)
        assert got is None and err == e.value.start, (b, err, e.value.start)
    rng = random.Random(3)
    for trial in range(3000):
        b = bytes(rng.choice([0x61, 0x20, 0x80, 0xbf, 0xc2, 0xc3, 0xe0, 0xe2, 0xed, 0xf0, 0xf4, 0xa0, 0x90, 0x9f]) for _ in range(rng.randint(1, 8)))
        try:
            b.decode("utf-8")
            exp = -1
        except UnicodeDecodeError as e:
            exp = e.start
        got, err = model_split(model, b)
        assert err == exp, (b, err, exp)


def test_reference_pretoken_fixtures(model):
    """G6: hashes of the reference's own _preprocess_corpus output (tests/golden/make_golden.py), chunked variants included."""
    import sys
    sys.path.insert(0, str(HM.parent.parent / "yet-another-bpe_amd"))
    from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig

    sps = {"endoftext": ["<|endoftext|>"], "default4": ["[PAD]", "[UNK]", "[BOS]", "[EOS]"], "none": []}
    g6 = json.loads((GOLD / "g6_pretokens.json").read_text())
    for key, pin in g6.items():
        parts = key.split("|")
        path = GOLD / parts[0]
        chunk = int(parts[2][5:]) if len(parts) > 2 else 1 << 30
        ranges = BBPETrainer(BBPETrainerConfig(chunk_size_bytes=chunk))._chunk_ranges(path)
        data = path.read_bytes()
        assert [a for a, _ in ranges] == sorted({a for a, _ in ranges}) and (not ranges or ranges[-1][1] == len(data))
        got, err = model_split(model, data, sps[parts[1]], [a for a, _ in ranges] or [0])
        assert err == -1
        h = hashlib.sha256()
        for t in got:
            h.update(len(t).to_bytes(4, "little"))
            h.update(t)
        assert len(got) == pin["n"] and h.hexdigest() == pin["sha256"], key
