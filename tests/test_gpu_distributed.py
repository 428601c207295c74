"""GPU (-m gpu): the multi-rank path of libyabpe.so with TWO ranks sharing the one GPU of the test box.
RCCL refuses two ranks on one device, so the ranks exchange through the custom transport (yabpe_comm_init_custom,
gloo underneath); everything else -- sharded load, per-rank delta tables, extract / all-gather / apply kernels,
in-band halts, global recounts -- is the code the RCCL transport runs too.  Merges must equal the oracle's."""
from __future__ import annotations

import pytest

from oracle import oracle
from tests import dist_workers, helpers

pytestmark = pytest.mark.gpu
SP = ["<|endoftext|>"]


def _expect(scenario):
    from yet_another_bpe import synth

    if scenario.startswith("corpus_en"):
        return oracle.merge_loop(helpers.corpus_en_words(), 257 + 700, 1, SP)[1]
    if scenario == "synthetic_small_buffers":
        flat, off = synth.generate(synth.SynthSpec(3 << 20, 20_000, 9, bytes(range(256)), False))
        return oracle.train_flat(flat, off, 257 + 400, 1, SP)[1]
    if scenario == "synthetic_medium":
        flat, off = synth.generate(synth.SynthSpec(24 << 20, 100_000, 13, bytes(range(256)), False))
        return oracle.train_flat(flat, off, 257 + 2000, 1, SP)[1]
    if scenario == "synthetic_tight_exchange":
        flat, off = synth.generate(synth.SynthSpec(6 << 20, 30_000, 17, bytes(range(256)), False))
        return oracle.train_flat(flat, off, 257 + 600, 1, SP)[1]
    words = [b" " * 900, b"ab" * 700, b"xyz" * 50, b"abcabc", b"  ", b"aaa"] * 3 + [b"hello world"] * 5
    return oracle.merge_loop(words, 257 + 120, 1, SP)[1]


@pytest.mark.parametrize("scenario", ["corpus_en_flat", "corpus_en_weighted", "synthetic_small_buffers", "synthetic_medium", "synthetic_tight_exchange", "long_words"])
def test_two_ranks_one_gpu(scenario):
    exp = _expect(scenario)
    outs = dist_workers.spawn(dist_workers.gpu_sharded, 2, scenario, timeout=900)
    for merges, n_words, rebuilds, retiles in outs:
        assert [(bytes.fromhex(a), bytes.fromhex(b)) for a, b in merges] == exp
    assert outs[0][1] > 0 and outs[1][1] > 0  # both ranks held words
    if scenario in ("synthetic_small_buffers", "synthetic_tight_exchange"):
        assert outs[0][2] >= 1  # the overflow recovery (global recount) ran


@pytest.mark.parametrize("world,scenario", [(2, "corpus_en_flat"), (2, "corpus_en_weighted"), (2, "synthetic_small_buffers"), (2, "synthetic_tight_exchange"),
                                            (2, "long_words"), (4, "synthetic_medium"), (4, "corpus_en_flat_direct_store")])
def test_peer_to_peer_exchange(world, scenario):
    """The same scenarios with the exchange going peer to peer (yabpe_comm_enable_p2p: hipIpc-mapped receive buffers, one
    push-and-wait launch per batch instead of an all-gather; the areas are mapped once, at a fixed size -- growing exchange
    buffers fit inside them): result = the oracle's."""
    exp = _expect(scenario if scenario != "corpus_en_flat_direct_store" else "corpus_en_flat")
    outs = dist_workers.spawn(dist_workers.gpu_sharded, world, scenario, "torch+p2p", timeout=900)
    for merges, n_words, rebuilds, retiles in outs:
        assert [(bytes.fromhex(a), bytes.fromhex(b)) for a, b in merges] == exp
    if scenario in ("synthetic_small_buffers", "synthetic_tight_exchange"):
        assert outs[0][2] >= 1  # the overflow recovery ran (the buffers grew; the mapped areas stayed)


@pytest.mark.parametrize("scenario", ["corpus_en_flat", "synthetic_medium"])
def test_four_ranks_one_gpu(scenario):
    """World size 4 through the same transport (the box allows 6 processes on its card): shards, exchange buffers and the
    fused delta-apply + selection launch with four contributors; result = the oracle's."""
    exp = _expect(scenario)
    outs = dist_workers.spawn(dist_workers.gpu_sharded, 4, scenario, timeout=900)
    for merges, n_words, rebuilds, retiles in outs:
        assert [(bytes.fromhex(a), bytes.fromhex(b)) for a, b in merges] == exp
    assert all(o[1] > 0 for o in outs)


def test_four_ranks_direct_store_flushes_into_the_send_buffer():
    """Named regression case: four ranks, streaming form only, direct-indexed delta store -- hist_flush calls the
    record sink of flush_entries once per role, back to back; the sink's LDS scratch (s_wtot / s_sbase) was rewritten under slower
    waves until the helper ended on a barrier (commit e85c033: an intermittent wrong count with 4 ranks)."""
    exp = _expect("corpus_en_flat")
    outs = dist_workers.spawn(dist_workers.gpu_sharded, 4, "corpus_en_flat_direct_store", timeout=900)
    for merges, n_words, rebuilds, retiles in outs:
        assert [(bytes.fromhex(a), bytes.fromhex(b)) for a, b in merges] == exp


def test_rccl_transport_single_rank_smoke():
    """The real RCCL calls (dlopen, ncclGetUniqueId, ncclCommInitRank, ncclAllGather on the compute stream) with a
    1-rank communicator: the exchange path runs end to end through librccl and must not change the result."""
    from yet_another_bpe import _native

    words = helpers.corpus_en_words()
    base = helpers.base_tokens(SP)
    flat, off = helpers.flatten(words)
    exp = oracle.merge_loop(words, 257 + 300, 1, SP)[1]
    with _native.Context(0) as ctx:
        ctx.set_option("force_comm", 1)
        ctx.set_option("verify", 1)
        ctx.set_vocab(base)
        uid = _native.Context.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        ctx.comm_init(0, 1, uid)
        ctx.load_words(flat, off)
        left, right, merged, _ = ctx.train(300, 1)
    toks = list(base)
    got = []
    for l, r, m in zip(left.tolist(), right.tolist(), merged.tolist()):
        got.append((toks[l], toks[r]))
        if m == len(toks):
            toks.append(toks[l] + toks[r])
    assert got == exp


def test_text_in_model_out_two_ranks(golden_dir, monkeypatch, tmp_path):
    """BASELINE configs[4] shape on one GPU: every rank pre-tokenises its chunks of the text on the device, pools its
    pre-tokens and joins the collective merge loop; result = the single-process trainer with the same chunk size."""
    from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig

    from oracle import pretok

    data = (golden_dir / "corpus.en").read_bytes()
    for chunk in (16384, 1 << 30):  # 9 chunks over 2 ranks / one chunk: the second rank holds no words at all
        # expected: the ORACLE on the reference's chunking (trainer.py:172-198: cuts every chunk_size bytes, moved back to a
        # UTF-8 boundary; corpus.en is ASCII, so the cuts are the multiples of the chunk size) -- regex split per chunk, C oracle
        starts = list(range(0, len(data), chunk))
        exp_vocab, exp_merges = oracle.merge_loop(pretok.pretokenize(data, SP, chunk_starts=starts), 700, 1, SP)
        cfg = BBPETrainerConfig(vocab_size=700, min_frequency=1, special_tokens=SP, chunk_size_bytes=chunk)
        assert BBPETrainer(cfg)._chunk_ranges(golden_dir / "corpus.en") == [(s0, min(s0 + chunk, len(data))) for s0 in starts]  # the product cuts there too
        outs = dist_workers.spawn(dist_workers.gpu_text_sharded, 2, str(golden_dir / "corpus.en"), chunk, 700, SP, timeout=900)
        for merges, nv in outs:
            assert [(bytes.fromhex(a), bytes.fromhex(b)) for a, b in merges] == exp_merges and nv == len(exp_vocab)
    bad = tmp_path / "bad.txt"
    bad.write_bytes(b"good " * 5000 + b"\xff" + b" more" * 5000)
    with pytest.raises(AssertionError, match="contains invalid UTF-8 at position 25000"):
        dist_workers.spawn(dist_workers.gpu_text_sharded, 2, str(bad), 8192, 300, SP, timeout=300)
