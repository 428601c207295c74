#!/usr/bin/env python3
"""Timing driver with the fields of the reference's tests/benchmark_trainer.py (:13-94): BBPETrainer.train() end to end
(pre-tokenisation on the host + merge loop on the GPU) on tests/golden/corpus.en at vocab 500 and 1000, three runs
each, and on a 5 MB synthetic ASCII text (the reference's third case needs a TinyStories sample that is not shipped).
Prints mean / min time, final vocab size and merges learned per case, then one JSON line with all of it.

    python tools/benchmark_trainer.py            # needs an MI355X (no CPU fallback)
"""
from __future__ import annotations

import json
import statistics
import sys
import tempfile
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
from yet_another_bpe import synth  # noqa: E402
from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig  # noqa: E402


def timed_runs(corpus: Path, vocab_size: int, runs: int) -> dict:
    secs, model = [], None
    for _ in range(runs):
        trainer = BBPETrainer(BBPETrainerConfig(vocab_size=vocab_size, min_frequency=1, max_workers=1, special_tokens=["[EOS]"]))
        t0 = time.perf_counter()
        model = trainer.train([corpus])
        secs.append(time.perf_counter() - t0)
    return {"corpus": corpus.name, "vocab_size_target": vocab_size, "runs": runs, "mean_time_s": statistics.mean(secs),
            "min_time_s": min(secs), "max_time_s": max(secs), "vocab_size": len(model.vocab), "merges_count": len(model.merges)}


def synthetic_text(path: Path, target_bytes: int) -> None:
    """Space-separated lower-case words, Zipf over 50k types (the config-2 generator of SURVEY 8d), as a text file."""
    flat, off = synth.generate(synth.SynthSpec(target_bytes, 50_000, 2, b"abcdefghijklmnopqrstuvwxyz", True))
    path.write_bytes(flat.tobytes())


def main() -> dict:
    cases = [("Small corpus (corpus.en)", REPO / "tests/golden/corpus.en", 500, 3),
             ("Small corpus, larger vocab", REPO / "tests/golden/corpus.en", 1000, 3)]
    out = []
    with tempfile.TemporaryDirectory() as td:
        big = Path(td) / "synthetic_5M.txt"
        synthetic_text(big, 5 << 20)
        cases.append(("Large corpus (5 MB synthetic text)", big, 1000, 1))
        print("=" * 60 + "\nBBPETrainer benchmark (MI355X merge loop)\n" + "=" * 60)
        for title, path, vs, runs in cases:
            if not path.exists():
                print(f"skipping {title}: {path} not found")
                continue
            r = timed_runs(path, vs, runs)
            out.append(r)
            print(f"\n{title}\n" + "-" * 40)
            print(f"  Vocab size target: {vs}\n  Final vocab size: {r['vocab_size']}\n  Merges learned: {r['merges_count']}")
            print(f"  Mean time: {r['mean_time_s']:.3f}s\n  Min time:  {r['min_time_s']:.3f}s")
    line = {"benchmark": "BBPETrainer.train", "cases": out}
    print("\n" + json.dumps(line))
    return line


if __name__ == "__main__":
    main()
