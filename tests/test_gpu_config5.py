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


G10 = ["g10_config5_0.25gib_meta.json", "g10_config5_8gib_meta.json"]


@pytest.mark.parametrize("meta_name", G10)
def test_config5_text_utf8_long_tokens(golden_dir, meta_name):
    """BASELINE configs[4] at its size (G10; the quarter-GiB twin runs the same path in a second): 8 GiB of synthetic TEXT
    with multi-byte UTF-8 (Cyrillic, CJK, Gothic, emoji), whitespace runs (U+00A0, U+3000), contractions, the special token
    as text and letter / space / digit runs of 64..300 bytes (the long-word path), cut into the reference's chunks
    (trainer.py:172-198) -> `yabpe_pretokenize` -> `YABPE_LOAD_DEDUP` -> 50,000 merges, against what chunked `regex` + the C
    oracle on the pooled words gave in the build container (tests/golden/make_golden_config5_8gib.py)."""
    from yet_another_bpe import _native, synth
    from yet_another_bpe.trainer import chunk_ranges

    path = golden_dir / meta_name
    if not path.exists():
        pytest.skip(f"{meta_name} has not been generated")
    meta = json.loads(path.read_text())
    g, sp = meta["generator"], meta["special_tokens"]
    base = helpers.base_tokens(sp)
    lb, lo = synth.text_lexicon(g["n_types"], g["seed"])
    assert hashlib.sha256(lb.tobytes() + lo.tobytes()).hexdigest() == g["lexicon_sha256"]
    with _native.Context() as gen:
        tb, _to, n_pieces, tn = gen.synth_generate_lex(g["target_bytes"], g["seed"], lb, lo)
        assert (tn, n_pieces) == (meta["text_bytes"], meta["pieces"])
        sha = hashlib.sha256()
        for a in range(0, tn, 1 << 28):
            sha.update(gen.d2h(tb + a, min(1 << 28, tn - a)).tobytes())
        assert sha.hexdigest() == meta["text_sha256"]  # the oracle's text, bit for bit
        ranges = chunk_ranges(tn, meta["chunk_size_bytes"], lambda off, n: gen.d2h(tb + off, n).tobytes())
        assert len(ranges) == meta["chunks"] and all(ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1))
        dt, do, nw = gen.pretokenize(tb, n_bytes=tn, chunk_starts=[a for a, _ in ranges], special_tokens=sp)
        assert nw == meta["pretokens"]
        sha, longest, step, prev = hashlib.sha256(), 0, 1 << 26, 0
        for a in range(0, nw, step):  # every boundary where regex put it: the byte lengths, in order
            off = gen.d2h(do + 8 * a, 8 * (min(step, nw - a) + 1), dtype=np.uint64)
            lens = np.diff(off).astype(np.uint32)
            sha.update(lens.tobytes())
            longest = max(longest, int(lens.max()))
        assert sha.hexdigest() == meta["pretoken_lengths_u32_sha256"] and longest == meta["longest_pretoken"]
        with _native.Context() as ctx:
            ctx.set_vocab(base)
            ctx.load_words_ptr(dt, do, nw, dedup=True)
            left, right, merged, count = ctx.train(meta["n_merges"], meta["min_frequency"])
            st = ctx.stats()
            assert ctx.verify_table() == 0
        assert st["n_words"] == meta["unique_words"] and st["n_long_words"] == meta["unique_long_words"] > 0
        assert len(left) == meta["n_merges"] and int(count[0]) == meta["first_count"] and int(count[-1]) == meta["last_count"]
        assert hashlib.sha256(left.astype(np.uint32).tobytes() + right.astype(np.uint32).tobytes() + merged.astype(np.uint32).tobytes()).hexdigest() == meta["id_triples_sha256"]
        toks, merges = list(base), []
        for l, r, m in zip(left.tolist(), right.tolist(), merged.tolist()):
            merges.append((toks[l], toks[r]))
            if m == len(toks):
                toks.append(toks[l] + toks[r])
        lines = oracle.merges_hex(merges).splitlines(keepends=True)
        for k, digest in meta["merges_sha256"].items():
            assert hashlib.sha256("".join(lines[: int(k)]).encode()).hexdigest() == digest, f"first {k} merges differ from the oracle"
        assert len(toks) == meta["vocab_size"] and sum(1 for t in toks if any(b >= 0x80 for b in t)) == meta["non_ascii_tokens_in_vocab"]


@pytest.mark.parametrize("meta_name", G10)
def test_config5_text_two_ranks(golden_dir, meta_name):
    """The same job over two ranks (sharing the test box's GPU; records exchanged peer to peer): each pre-tokenises and pools its chunks."""
    from tests import dist_workers

    path = golden_dir / meta_name
    if not path.exists():
        pytest.skip(f"{meta_name} has not been generated")
    meta = json.loads(path.read_text())
    n_merges = meta["n_merges"]
    outs = dist_workers.spawn(dist_workers.gpu_device_text_sharded, 2, meta_name, n_merges, timeout=1500)
    for digest, got, n_pre, n_words, n_long, merges_digest in outs:
        assert got == n_merges and merges_digest == meta["merges_sha256"][str(n_merges)]
        if n_merges == meta["n_merges"]:
            assert digest == meta["id_triples_sha256"]
        assert n_pre > 0 and n_words > 0 and n_long > 0
    assert sum(o[2] for o in outs) == meta["pretokens"]  # the ranks' chunks cover the text exactly once
