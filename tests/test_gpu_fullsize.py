"""GPU, BASELINE.json's full size (configs[2]): the 1 GiB synthetic byte corpus, 32,000 merges, against the pin the CPU
oracle produced for exactly this job (tests/golden/g7_config3_meta.json, made by tests/golden/make_golden_config3.py),
plus the size-independent properties of the path: the best count never increases, the incrementally maintained pair
table equals a recount of the final stream, tokens only disappear through recorded merge sites, flat and pooled layouts
and a second run give the same merges."""
from __future__ import annotations

import hashlib
import json

import numpy as np
import pytest

from oracle import oracle
from tests import helpers

pytestmark = pytest.mark.gpu
SP = ["<|endoftext|>"]


def test_config3_one_gib_32k_merges(golden_dir):
    from yet_another_bpe import _native, synth

    meta = json.loads((golden_dir / "g7_config3_meta.json").read_text())
    spec = synth.SynthSpec.config3(1024 << 20)
    base = helpers.base_tokens(SP)
    with _native.Context() as gen:
        pb, po, nw, nb = gen.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
        assert (nb, nw) == (meta["corpus_bytes"], meta["n_words"])
        assert hashlib.sha256(gen.d2h(pb, nb).tobytes()).hexdigest() == meta["corpus_sha256"]  # same corpus as the oracle's
        triples = []
        for dedup in (False, True, False):
            with _native.Context() as ctx:
                ctx.set_vocab(base)
                ctx.load_words_ptr(pb, po, nw, dedup=dedup)
                left, right, merged, count = ctx.train(32000, 1)
                st = ctx.stats()
                assert ctx.verify_table() == 0                                  # incremental table == recount of the stream
                sites, _live = ctx.iter_log()
                assert len(left) == meta["n_merges"] and st["n_long_words"] == 0
                assert int(count[0]) == meta["first_count"] and int(count[-1]) == meta["last_count"]
                assert bool(np.all(count[:-1] >= count[1:]))                    # best count never increases
                if not dedup:                                                   # flat: every site removes exactly one token
                    assert st["tokens_initial"] - st["tokens_now"] == int(np.asarray(sites, dtype=np.uint64).sum())
                else:
                    assert st["n_words"] == meta["unique_words"]
                h = hashlib.sha256(left.astype(np.uint32).tobytes() + right.astype(np.uint32).tobytes() + merged.astype(np.uint32).tobytes()).hexdigest()
                triples.append(h)
                if len(triples) == 1:  # the byte-level merges list, as the golden files serialise it, at several prefixes
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
        assert triples[0] == triples[1] == triples[2] == meta["id_triples_sha256"]  # layouts and repeated runs agree with the oracle


@pytest.mark.parametrize("world,transport", [(2, "torch"), (2, "torch+p2p"), (4, "torch+p2p")])
def test_config4_one_gib_word_sharded(golden_dir, world, transport):
    """BASELINE configs[3]: the same job word-sharded over 2 and 4 ranks (they share the test box's one GPU and exchange
    through the custom transport; the exchange protocol, the replicated table and the batch selection are what RCCL ranks
    run too).  Every rank must return the oracle's id triples (G7), hold a consistent table and a share of the words."""
    from tests import dist_workers

    meta = json.loads((golden_dir / "g7_config3_meta.json").read_text())
    # ("torch": every exchange is an all-gather through host memory (gloo); "+p2p": the ranks push their records into each
    # other's hipIpc-mapped buffers from a kernel -- what GPUs of one node do over xGMI)
    outs = dist_workers.spawn(dist_workers.gpu_fullsize_sharded, world, transport, timeout=1500)
    for digest, n_merges, nw, nb, mism, my_words, exchanges, growths, xus, train_ms in outs:
        assert (nb, nw) == (meta["corpus_bytes"], meta["n_words"])
        assert n_merges == meta["n_merges"] and digest == meta["id_triples_sha256"]
        assert mism == 0 and my_words > 0 and exchanges > 0
    assert sum(o[5] for o in outs) == meta["n_words"]  # the shards cover the corpus exactly once
    print("world", world, transport, "exchanges", outs[0][6], "buffer growths", outs[0][7], "exchange launch us (sampled)", round(outs[0][8], 2), "merge loop ms", round(outs[0][9], 1))
