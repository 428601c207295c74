"""GPU, the shape of BASELINE.json's configs[4] at a size one GPU holds (G8): 1 GiB of TEXT -> `yabpe_pretokenize` (GPT-2
split on the device) -> `YABPE_LOAD_DEDUP` (word pooling on the device, trainer.py:221-225) -> `yabpe_train(50000)`, against
the digests the CPU oracle (regex over the whole text + oracle/bpe_oracle.c) produced for exactly these texts
(tests/golden/g8_config5_meta.json, made by tests/golden/make_golden_config5.py).  Reference path: trainer.py:63-92
(train), :136-214 (_preprocess_corpus), :216-302 (_merge_loop).  The flat layout (every pre-token occurrence resident)
must give the same merges as the pooled one."""
from __future__ import annotations

import hashlib
import json

import numpy as np
import pytest

from oracle import oracle
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["config2_text", "mixed_ascii"])
def test_text_in_50k_merges_out(golden_dir, name):
    from yet_another_bpe import _native

    meta = json.loads((golden_dir / "g8_config5_meta.json").read_text())[name]
    g = meta["generator"]
    sp = meta["special_tokens"]
    base = helpers.base_tokens(sp)
    with _native.Context() as gen:
        tb, _to, tw, tn = gen.synth_generate(g["target_bytes"], g["n_types"], g["seed"], bytes.fromhex(g["alphabet_hex"]), g["space_prefix"])
        assert (tn, tw) == (meta["text_bytes"], meta["generated_words"])
        assert hashlib.sha256(gen.d2h(tb, tn).tobytes()).hexdigest() == meta["text_sha256"]  # the oracle's text, bit for bit
        dt, do, nw = gen.pretokenize(tb, n_bytes=tn, special_tokens=sp)
        assert nw == meta["pretokens"]
        off = gen.d2h(do, (nw + 1) * 8, dtype=np.uint64)
        lens = np.diff(off).astype(np.uint32)
        del off
        assert int(lens.max()) == meta["longest_pretoken"]
        assert hashlib.sha256(lens.tobytes()).hexdigest() == meta["pretoken_lengths_u32_sha256"]  # every boundary where regex put it
        del lens
        triples = []
        for dedup in (True, False):
            with _native.Context() as ctx:
                ctx.set_vocab(base)
                ctx.load_words_ptr(dt, do, nw, dedup=dedup)
                left, right, merged, count = ctx.train(meta["n_merges"], meta["min_frequency"])
                st = ctx.stats()
                assert ctx.verify_table() == 0  # incremental table == recount of the final stream
            assert len(left) == meta["n_merges"] == 50000
            if dedup:
                assert st["n_words"] == meta["unique_words"]
            assert int(count[0]) == meta["first_count"] and int(count[-1]) == meta["last_count"]
            assert bool(np.all(count[:-1] >= count[1:]))
            triples.append(hashlib.sha256(left.astype(np.uint32).tobytes() + right.astype(np.uint32).tobytes() + merged.astype(np.uint32).tobytes()).hexdigest())
            if dedup:  # the byte-level merges list at several prefixes
                toks = list(base)
                merges = []
                for l, r, m in zip(left.tolist(), right.tolist(), merged.tolist()):
                    merges.append((toks[l], toks[r]))
                    if m == len(toks):
                        toks.append(toks[l] + toks[r])
                lines = oracle.merges_hex(merges).splitlines(keepends=True)
                for k, digest in meta["merges_sha256"].items():
                    assert hashlib.sha256("".join(lines[: int(k)]).encode()).hexdigest() == digest, f"first {k} merges differ from the oracle"
                assert len(toks) == meta["vocab_size"]
        assert triples[0] == triples[1] == meta["id_triples_sha256"]  # pooled and flat layouts, and the oracle, agree
