"""CPU: the batch rule of DESIGN (c) -- which consecutive merges one look at the pair table may fix -- replayed against
sequential BPE by tools/batch_sim.cpp (its own small exact implementation of trainer.py:241-300 on pooled words): at every
batch start it walks the candidates under the rule, with the byte comparisons cut to what the device can decide (8-byte
prefixes + lengths: variant 3), then runs sequential BPE and compares merge by merge.  Any mismatch fails.  The GPU parity
tests check the device's implementation of the rule; this one checks the rule."""
from __future__ import annotations

import re
import struct
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))


@pytest.fixture(scope="module")
def batch_sim(tmp_path_factory):
    exe = tmp_path_factory.mktemp("batch_sim") / "batch_sim"
    subprocess.run(["g++", "-O2", "-o", str(exe), str(REPO / "tools" / "batch_sim.cpp")], check=True)
    return exe


def _result(r) -> tuple[int, int, int]:
    """-> (merges, batches, mismatches) from batch_sim's last lines: "mismatches N" and "histogram: k:batches_of_k ..."."""
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"^mismatches (\d+)\nhistogram:(.*)$", r.stdout, re.M)
    assert m, r.stdout[-2000:]
    hist = [tuple(int(v) for v in kv.split(":")) for kv in m.group(2).split()]
    return sum(k * n for k, n in hist), sum(n for _k, n in hist), int(m.group(1))


def _words_file(path: Path, flat: np.ndarray, off: np.ndarray) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("<QQ", len(off) - 1, len(flat)))
        f.write(np.ascontiguousarray(off, dtype=np.uint64).tobytes())
        f.write(np.ascontiguousarray(flat, dtype=np.uint8).tobytes())


@pytest.mark.parametrize("name,mib,types,seed,alphabet,prefix,merges", [
    ("ascii_words", 4, 20_000, 21, b"abcdefghijklmnopqrstuvwxyz", True, 3000),
    ("few_types_many_ties", 2, 400, 22, b"abcdef", True, 1500),     # long chains inside word types, counts tie all the time
    ("all_bytes", 4, 100_000, 23, bytes(range(256)), False, 3000),
])
def test_batches_equal_sequential_bpe(batch_sim, tmp_path, name, mib, types, seed, alphabet, prefix, merges):
    from yet_another_bpe import synth

    flat, off = synth.generate(synth.SynthSpec(mib << 20, types, seed, alphabet, prefix))
    wf = tmp_path / f"{name}.bin"
    _words_file(wf, flat, off)
    r = subprocess.run([str(batch_sim), str(wf), str(merges), "16", "3"], capture_output=True, text=True, timeout=600)
    n_merges, n_batches, mismatches = _result(r)
    assert mismatches == 0, r.stderr[-2000:]
    assert 0 < n_merges <= merges and n_batches < n_merges  # (a small corpus runs out of pairs first) batches do form


def test_golden_corpus_batches_equal_sequential_bpe(batch_sim, tmp_path, golden_dir):
    """The reference's own test corpus (pre-tokenised as the trainer does), 743 merges as in its golden model."""
    from oracle import pretok

    text = (golden_dir / "corpus.en").read_text(encoding="utf-8")
    words = [m.group().encode("utf-8") for m in pretok.split_pattern(["<|endoftext|>"]).finditer(text) if m.group()]
    flat = np.frombuffer(b"".join(words), dtype=np.uint8)
    off = np.zeros(len(words) + 1, dtype=np.uint64)
    np.cumsum([len(w) for w in words], out=off[1:])
    wf = tmp_path / "corpus_en.bin"
    _words_file(wf, flat, off)
    r = subprocess.run([str(batch_sim), str(wf), "743", "16", "3"], capture_output=True, text=True, timeout=600)
    n_merges, n_batches, mismatches = _result(r)
    assert mismatches == 0 and n_merges == 743 and n_batches < 743, (r.stdout[-1000:], r.stderr[-1000:])
