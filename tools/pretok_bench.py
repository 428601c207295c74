#!/usr/bin/env python3
"""Dev tool: throughput of the device pre-tokeniser (yabpe_pretokenize) on synthetic text resident in HBM, and of the
host path (`regex`, what the reference runs, trainer.py:163-170) on a sample of the same text.
   python tools/pretok_bench.py [MiB]"""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
import numpy as np
import regex
from yet_another_bpe import _native, synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
GPT2 = r"""'(?:[sdmt]|ll|ve|re)| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""
SP = ["<|endoftext|>"]
# text = the config-2 style corpus (space + lower-case word, Zipf over 50k types), generated on the device
with _native.Context() as ctx:
    pb, po, nw, nb = ctx.synth_generate(mib << 20, 50_000, 2, b"abcdefghijklmnopqrstuvwxyz", True)
    for rep in range(3):
        t0 = time.perf_counter()
        dt, do, n_words = ctx.pretokenize(pb, n_bytes=nb, special_tokens=SP)
        dt_s = time.perf_counter() - t0
        print(f"device: {nb / 2**20:.0f} MiB -> {n_words} pre-tokens in {dt_s * 1e3:.1f} ms = {nb / dt_s / 1e9:.2f} GB/s (wall, incl. allocation)", flush=True)
        assert n_words == nw
        ctx.pretokenize_free()
    sample = ctx.d2h(pb, min(nb, 16 << 20)).tobytes()
sample = sample[: sample.rfind(b" ")]
pat = regex.compile("|".join(regex.escape(t) for t in SP) + "|" + GPT2)
t0 = time.perf_counter()
toks = pat.findall(sample.decode("utf-8"))
dt_h = time.perf_counter() - t0
print(f"host regex: {len(sample) / 2**20:.1f} MiB -> {len(toks)} pre-tokens in {dt_h:.2f} s = {len(sample) / dt_h / 1e6:.2f} MB/s (1 core)")
