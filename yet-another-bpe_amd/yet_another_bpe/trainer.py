"""Byte-level BPE trainer: host-side mirror of the reference interface over the MI355X HIP hot path.

Same names, arguments, defaults and error behaviour as DreamOneX/yet-another-bpe
`src/yet_another_bpe/trainer.py` (BBPETrainerConfig :17-38, BBPEModel :41-52, BBPETrainer :55-302);
the merge loop itself (:216-302) runs on the GPU through the C ABI in include/yabpe.h (libyabpe.so,
loaded with ctypes by `_native`).  There is no CPU fallback: without the built library and a GPU,
`_merge_loop` raises.

Host-only steps (file chunking, save format) stay in Python as in the reference.  Pre-tokenisation (UTF-8 decode +
GPT-2 regex, trainer.py:136-214) exists twice with identical results: `_preprocess_corpus` runs the `regex` module on
the host and returns Python lists, as the reference's method does; `train()` on a corpus of 1 MiB or more (or with
YABPE_PRETOKENIZE=gpu) hands the raw file bytes to `yabpe_pretokenize` instead and never builds Python objects per
pre-token (YABPE_PRETOKENIZE=host forces the host path).
"""
from __future__ import annotations

import json
import os
from collections import Counter
from collections.abc import Mapping, Sequence
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np
import regex

# GPT-2 pre-tokenisation pattern (reference trainer.py:163)
_GPT2_SPLIT = r"""'(?:[sdmt]|ll|ve|re)| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""


@dataclass
class BBPETrainerConfig:
    """Configuration of a BBPE trainer (fields and defaults of reference trainer.py:31-38).

    Attributes:
        vocab_size: target vocabulary size, special tokens included.
        min_frequency: a pair is merged only while its count is at least this.
        max_workers: worker threads for file-chunk pre-tokenisation.
        chunk_size_bytes: logical chunk size when splitting large files.
        seed: unused (kept for compatibility, as in the reference).
        special_tokens: strings that get vocabulary ids right after the 256 bytes.
    """

    vocab_size: int = 32000
    min_frequency: int = 2
    max_workers: int = 8
    chunk_size_bytes: int = 8 * 1024 * 1024
    seed: int = 42
    special_tokens: Sequence[str] = field(default_factory=lambda: ["[PAD]", "[UNK]", "[BOS]", "[EOS]"])


class BBPEModel:
    """Result container (reference trainer.py:41-52): copies of vocab, merges, special tokens."""

    def __init__(self, vocab: Mapping[bytes, int], merges: Sequence[tuple[bytes, bytes]],
                 special_tokens: Sequence[str]) -> None:
        self.vocab: dict[bytes, int] = dict(vocab)
        self.merges: list[tuple[bytes, bytes]] = list(merges)
        self.special_tokens: list[str] = list(special_tokens)


def _utf8_cut(window: bytes, pos: int) -> int:
    """Largest cut <= pos inside `window` that is not in the middle of a UTF-8 sequence
    (continuation bytes are 10xxxxxx); mirrors find_utf8_boundary of reference trainer.py:139-144."""
    if pos >= len(window):
        return len(window)
    while pos > 0 and (window[pos] & 0xC0) == 0x80:
        pos -= 1
    return pos


def chunk_ranges(size: int, step: int, read) -> list[tuple[int, int]]:
    """The reference's chunk cuts (get_chunks, trainer.py:172-198) for `size` bytes that `read(offset, n) -> bytes` gives
    access to (a file, or text that lives on the device): a cut every `step` bytes, moved back to a UTF-8 boundary."""
    if size == 0:
        return []
    if size <= step:
        return [(0, size)]
    ranges: list[tuple[int, int]] = []
    start = 0
    while start < size:
        stop = min(start + step, size)
        if stop < size:  # back off to a UTF-8 boundary using a 5-byte window (trainer.py:183-190)
            w0 = max(0, stop - 4)
            window = read(w0, stop + 1 - w0)
            stop = w0 + _utf8_cut(window, stop - w0)
        if stop > start:
            ranges.append((start, stop))
            start = stop
        else:
            start += 1  # no progress possible: the reference skips one byte (trainer.py:196-197)
    return ranges


class BBPETrainer:
    """Byte-level BPE trainer with the reference's API; the merge loop runs on the GPU."""

    def __init__(self, config: BBPETrainerConfig | None = None) -> None:
        self.config: BBPETrainerConfig = config or BBPETrainerConfig()
        self._vocab: dict[bytes, int] = {}
        self._merges: list[tuple[bytes, bytes]] = []
        self.last_stats: dict | None = None  # yabpe_stats of the last merge loop (not in the reference)

    # ------------------------------------------------------------------ train / save (trainer.py:63-117)
    def train(self, files: Sequence[str | Path]) -> BBPEModel:
        if not files:
            raise ValueError("At least one file must be provided")
        paths = [Path(f) if isinstance(f, str) else f for f in files]
        mode = os.environ.get("YABPE_PRETOKENIZE", "auto")
        if mode == "gpu" or (mode == "auto" and sum(p.stat().st_size for p in paths if p.exists()) >= (1 << 20)):
            return self._train_device(paths)
        pretokens = self._pretokenize(paths)
        specials = list(self.config.special_tokens)
        if not pretokens:  # empty corpus: base vocab only (trainer.py:81-85)
            self._vocab = self._init_base_vocab()
            self._merges = []
            return BBPEModel(vocab=self._vocab, merges=[], special_tokens=specials)
        # word-frequency pooling (trainer.py:221-225) on the host: the pre-tokens are Python strings here anyway
        if os.environ.get("YABPE_LAYOUT", "dedup") == "flat":
            words = [t.encode("utf-8") for t in pretokens]
            freq = None
        else:
            pooled = Counter(pretokens)
            words = [t.encode("utf-8") for t in pooled]
            freq = np.fromiter(pooled.values(), dtype=np.uint64, count=len(pooled))
        vocab, merges = self._merge_loop_words(words, freq)
        self._vocab = vocab
        self._merges = merges
        return BBPEModel(vocab=vocab, merges=merges, special_tokens=specials)

    def _train_device(self, paths: Sequence[Path]) -> BBPEModel:
        """train() with the pre-tokeniser on the GPU: file bytes -> yabpe_pretokenize -> word offsets in HBM ->
        yabpe_load_words (equal pre-tokens pooled on the device) -> merge loop.  Same results as the host path."""
        from . import _native  # fails loudly when libyabpe.so / a GPU is missing

        specials = list(self.config.special_tokens)
        pieces: list[np.ndarray] = []
        chunks: list[tuple[int, Path, int]] = []  # (start in the joined buffer, file, start in the file)
        total = 0
        for path in paths:
            if not path.exists():
                raise FileNotFoundError(f"File not found: {path}")
            ranges = self._chunk_ranges(path)
            if not ranges:
                continue
            data = np.fromfile(path, dtype=np.uint8)
            for start, stop in ranges:  # (the reference can skip bytes between chunks, trainer.py:196-197)
                chunks.append((total, path, start))
                pieces.append(data[start:stop])
                total += stop - start
        base = self._base_tokens()
        num_merges = max(0, self.config.vocab_size - len(base))
        empty = BBPEModel(vocab={t: i for i, t in enumerate(base)}, merges=[], special_tokens=specials)
        if total == 0:
            self._vocab, self._merges = dict(empty.vocab), []
            return empty
        text = pieces[0] if len(pieces) == 1 else np.concatenate(pieces)
        with _native.Context() as ctx:
            try:
                dev_text, dev_off, n_words = ctx.pretokenize(text, chunk_starts=[c[0] for c in chunks], special_tokens=specials)
            except _native.Utf8Error as e:
                k = max(i for i, c in enumerate(chunks) if c[0] <= e.position)
                g0, path, f0 = chunks[k]
                raise ValueError(f"File {path} contains invalid UTF-8 at position {f0 + e.position - g0}.") from e
            if n_words == 0 or num_merges == 0:
                self._vocab, self._merges = dict(empty.vocab), []
                return empty
            ctx.set_vocab(base)
            ctx.load_words_ptr(dev_text, dev_off, n_words, dedup=os.environ.get("YABPE_LAYOUT", "dedup") != "flat")
            left, right, merged, _count = ctx.train(num_merges, max(0, int(self.config.min_frequency)))  # (<= 0: merge to exhaustion, as the reference does)
            self.last_stats = ctx.stats()
        vocab, merges = self._decode_merges(base, left, right, merged)
        self._vocab = vocab
        self._merges = merges
        return BBPEModel(vocab=vocab, merges=merges, special_tokens=specials)

    @staticmethod
    def _decode_merges(base: Sequence[bytes], left, right, merged):
        toks = list(base)
        merges: list[tuple[bytes, bytes]] = []
        for l, r, m in zip(left.tolist(), right.tolist(), merged.tolist()):
            merges.append((toks[l], toks[r]))
            if m == len(toks):  # a fresh id; otherwise the bytes already existed (trainer.py:298-300)
                toks.append(toks[l] + toks[r])
        return {t: i for i, t in enumerate(toks)}, merges

    def save(self, output_dir: str | Path) -> None:
        """vocab.json / merges.txt / special_tokens.json in the reference's format (trainer.py:94-117)."""
        if not self._vocab:
            raise ValueError("Model has not been trained yet. Call train() first.")
        out = Path(output_dir)
        out.mkdir(parents=True, exist_ok=True)
        with open(out / "vocab.json", "w", encoding="utf-8") as f:
            json.dump({tok.decode("latin-1"): idx for tok, idx in self._vocab.items()}, f, ensure_ascii=False, indent=2)
        with open(out / "merges.txt", "w", encoding="utf-8") as f:
            for left, right in self._merges:
                f.write(f"{left.decode('latin-1')} {right.decode('latin-1')}\n")
        with open(out / "special_tokens.json", "w", encoding="utf-8") as f:
            json.dump(list(self.config.special_tokens), f, ensure_ascii=False, indent=2)

    def save_lossless(self, output_dir: str | Path) -> None:
        """The same model in a format that survives a reload byte for byte: `vocab.hex.json` ({token hex: id}),
        `merges.hex` (one `left_hex right_hex` line per merge, the serialisation of the golden files) and
        special_tokens.json.  The reference's own format (save()) loses every merge whose LEFT token contains a space --
        its loader splits each line at the first space (tokenizer.py:137) -- and every token with a line break; it is kept
        as it is for compatibility, this one is offered next to it (BBPETokenizer.from_file_lossless)."""
        if not self._vocab:
            raise ValueError("Model has not been trained yet. Call train() first.")
        out = Path(output_dir)
        out.mkdir(parents=True, exist_ok=True)
        with open(out / "vocab.hex.json", "w", encoding="ascii") as f:
            json.dump({tok.hex(): idx for tok, idx in self._vocab.items()}, f, indent=0)
        with open(out / "merges.hex", "w", encoding="ascii") as f:
            for left, right in self._merges:
                f.write(f"{left.hex()} {right.hex()}\n")
        with open(out / "special_tokens.json", "w", encoding="utf-8") as f:
            json.dump(list(self.config.special_tokens), f, ensure_ascii=False, indent=2)

    # ------------------------------------------------------------------ base vocab (trainer.py:119-134)
    def _base_tokens(self) -> list[bytes]:
        toks = [bytes([b]) for b in range(256)]
        seen = set(toks)
        for s in self.config.special_tokens:
            tb = s.encode("utf-8")
            if tb not in seen:  # a special whose bytes already exist gets no id (trainer.py:130)
                seen.add(tb)
                toks.append(tb)
        return toks

    def _init_base_vocab(self) -> dict[bytes, int]:
        return {t: i for i, t in enumerate(self._base_tokens())}

    # ------------------------------------------------------------------ pre-tokenisation (trainer.py:136-214)
    def _chunk_ranges(self, path: Path) -> list[tuple[int, int]]:
        size = path.stat().st_size
        with open(path, "rb") as f:
            def read(off: int, n: int) -> bytes:
                f.seek(off)
                return f.read(n)

            return chunk_ranges(size, self.config.chunk_size_bytes, read)

    def _split_pattern(self) -> "regex.Pattern[str]":
        pat = _GPT2_SPLIT
        if self.config.special_tokens:  # specials first, in config order, kept as words (trainer.py:165-167)
            pat = "|".join(regex.escape(t) for t in self.config.special_tokens) + "|" + pat
        return regex.compile(pat)

    def _pretokenize(self, files: Sequence[Path]) -> list[str]:
        """All non-empty pre-tokens as strings, in file / chunk order."""
        pattern = self._split_pattern()

        def run(path: Path, start: int, stop: int) -> list[str]:
            with open(path, "rb") as f:
                f.seek(start)
                raw = f.read(stop - start)
            try:
                text = raw.decode("utf-8")
            except UnicodeDecodeError as e:
                raise ValueError(f"File {path} contains invalid UTF-8 at position {start + e.start}.") from e
            return [t for t in pattern.findall(text) if t]

        out: list[str] = []
        with ThreadPoolExecutor(max_workers=self.config.max_workers) as pool:
            jobs = []
            for path in files:
                if not path.exists():
                    raise FileNotFoundError(f"File not found: {path}")
                for start, stop in self._chunk_ranges(path):
                    jobs.append(pool.submit(run, path, start, stop))
            for job in jobs:  # submission order => deterministic output order (trainer.py:209)
                out.extend(job.result())
        return out

    def _preprocess_corpus(self, files: Sequence[Path]) -> list[list[int]]:
        """Pre-tokens as lists of byte values, as the reference returns them (trainer.py:136-214)."""
        return [list(t.encode("utf-8")) for t in self._pretokenize(files)]

    # ------------------------------------------------------------------ merge loop (trainer.py:216-302)
    def _merge_loop(self, sequences: list[list[int]]) -> tuple[dict[bytes, int], list[tuple[bytes, bytes]]]:
        """Runs the BPE merge loop on the GPU.  `sequences`: one list of byte values (0..255) per pre-token."""
        words = [bytes(s) for s in sequences]
        return self._merge_loop_words(words, None)

    def _merge_loop_words(self, words: Sequence[bytes], freq: np.ndarray | None):
        base = self._base_tokens()
        num_merges = max(0, self.config.vocab_size - len(base))  # trainer.py:238
        if not words or num_merges == 0:
            return {t: i for i, t in enumerate(base)}, []
        from . import _native  # fails loudly when libyabpe.so / a GPU is missing

        lens = np.fromiter((len(w) for w in words), dtype=np.uint64, count=len(words))
        off = np.zeros(len(words) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        flat = np.frombuffer(b"".join(words), dtype=np.uint8)
        with _native.Context() as ctx:
            ctx.set_vocab(base)
            ctx.load_words(flat, off, freq)
            left, right, merged, _count = ctx.train(num_merges, max(0, int(self.config.min_frequency)))  # (<= 0: merge to exhaustion, as the reference does)
            self.last_stats = ctx.stats()
        return self._decode_merges(base, left, right, merged)
