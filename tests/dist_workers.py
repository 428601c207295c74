"""Worker functions for the multi-rank tests (spawned with torch.multiprocessing, gloo on 127.0.0.1)."""
from __future__ import annotations

import os
import sys
import traceback
from collections import Counter
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / "yet-another-bpe_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def _init(rank: int, world: int, port: int):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _run(fn, rank, world, port, q, *args):
    import faulthandler
    import signal

    faulthandler.register(signal.SIGUSR1, all_threads=True)  # (kill -USR1 <worker>: where is a stuck rank?)
    if os.environ.get("YABPE_TEST_DUMP_AFTER"):  # (debugging a stuck rank: Python stacks of all threads after N seconds, repeated)
        faulthandler.dump_traceback_later(float(os.environ["YABPE_TEST_DUMP_AFTER"]), repeat=True)
    try:
        dist = _init(rank, world, port)
        out = fn(rank, world, dist, *args)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", out))
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


def spawn(fn, world: int, *args, timeout: float = 600.0):
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(fn, r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            rank, status, out = q.get(timeout=timeout)
            assert status == "ok", f"rank {rank} failed:\n{out}"
            results[rank] = out
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    return [results[r] for r in range(world)]


# ---------------------------------------------------------------- CPU: transport plumbing + the distributed algorithm
def cpu_transport_roundtrip(rank, world, dist):
    """The yabpe_allgather_fn callback path with host memory standing in for device memory."""
    import ctypes

    import numpy as np

    from yet_another_bpe import _native
    from yet_another_bpe.distributed import host_memory_transport

    n = 1000
    send = np.full(n, rank + 1, dtype=np.uint8)
    send[0] = 7 * rank
    recv = np.zeros(n * world, dtype=np.uint8)
    tr = host_memory_transport()
    cb = _native.ALLGATHER_FN(lambda _u, s, r, nb: tr(s, r, nb))  # exactly how Context.comm_init_custom wraps it
    rc = cb(None, ctypes.c_void_p(send.ctypes.data), ctypes.c_void_p(recv.ctypes.data), n)
    assert rc == 0
    for r in range(world):
        assert recv[r * n] == 7 * r and (recv[r * n + 1:(r + 1) * n] == r + 1).all()
    return int(recv.sum())


def cpu_sharded_reference(rank, world, dist, words_hex, specials, vocab_size, min_frequency):
    """Pure-Python model of the multi-GPU protocol: word shards, replicated pair table, per-merge exchange of
    aggregated (pair, delta) records, identical argmax on every rank.  Returns the merges as hex pairs."""
    import numpy as np

    from yet_another_bpe.distributed import plan_shards

    words = [bytes.fromhex(w) for w in words_hex]
    lens = np.array([len(w) for w in words], dtype=np.uint64)
    off = np.zeros(len(words) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    w0, w1 = plan_shards(off, world)[rank]
    mine = [tuple(bytes([b]) for b in w) for w in words[w0:w1]]
    vocab = {bytes([b]): b for b in range(256)}
    for s in specials:
        if s.encode() not in vocab:
            vocab[s.encode()] = len(vocab)

    def local_counts(ws):
        c = Counter()
        for w in ws:
            for j in range(len(w) - 1):
                c[(w[j], w[j + 1])] += 1
        return c

    def exchange(delta: dict):
        parts = [None] * world
        dist.all_gather_object(parts, sorted(delta.items()))
        return parts

    table = Counter()
    for part in exchange(local_counts(mine)):  # initial count: sum of the shards' histograms
        for k, v in part:
            table[k] += v
    merges = []
    for _ in range(max(0, vocab_size - len(vocab))):
        live = {k: v for k, v in table.items() if v > 0}
        if not live:
            break
        best = max(live.items(), key=lambda kv: (kv[1], kv[0]))[0]
        if live[best] < min_frequency:
            break
        x, y = best
        z = x + y
        before = local_counts(mine)
        new = []
        for w in mine:
            out, j = [], 0
            while j < len(w):
                if j + 1 < len(w) and w[j] == x and w[j + 1] == y:
                    out.append(z)
                    j += 2
                else:
                    out.append(w[j])
                    j += 1
            new.append(tuple(out))
        mine = new
        after = local_counts(mine)
        delta = {k: after.get(k, 0) - before.get(k, 0) for k in set(before) | set(after)}
        delta = {k: v for k, v in delta.items() if v != 0 and k != best}  # the merged pair is zeroed, not updated
        table[best] = 0
        for part in exchange(delta):
            for k, v in part:
                table[k] += v
        merges.append(best)
        if z not in vocab:
            vocab[z] = len(vocab)
    return [(a.hex(), b.hex()) for a, b in merges], len(vocab)


# ---------------------------------------------------------------- GPU: 2 ranks on one GPU through the custom transport
def gpu_sharded(rank, world, dist, scenario, transport="torch"):
    import numpy as np

    from tests import helpers
    from yet_another_bpe import _native, synth
    from yet_another_bpe.distributed import train_sharded

    sp = ["<|endoftext|>"]
    base = helpers.base_tokens(sp)
    opts = {"verify": 1}
    if scenario in ("corpus_en_flat", "corpus_en_flat_direct_store"):
        flat, off = helpers.flatten(helpers.corpus_en_words())
        freq, merges = None, 700
        if scenario == "corpus_en_flat_direct_store":  # the streaming form throughout: its direct-indexed delta store flushes into the
            opts.update({"split": 0, "hist": 1})      # send buffer four times back to back per launch (the race fixed in e85c033)
    elif scenario == "corpus_en_weighted":
        uw, fq = helpers.pooled(helpers.corpus_en_words())
        flat, off = helpers.flatten(uw)
        freq, merges = fq, 700
    elif scenario == "synthetic_small_buffers":  # forces DELTA_FULL / TABLE_FULL recoveries and retiles
        flat, off = synth.generate(synth.SynthSpec(3 << 20, 20_000, 9, bytes(range(256)), False))
        freq, merges = None, 400
        opts.update({"delta_cap": 4, "delta_table_log2": 6, "table_min_log2": 10, "check_interval": 5,
                     "retile_pct": 95, "retile_min_tiles": 8})
    elif scenario == "synthetic_medium":  # big enough for the split forms, signatures, candidate argmax, a retile
        flat, off = synth.generate(synth.SynthSpec(24 << 20, 100_000, 13, bytes(range(256)), False))
        freq, merges = None, 2000
        opts = {}
    elif scenario == "synthetic_tight_exchange":  # the exchange transmits LESS than merges need (half the recent maximum): overflows
        flat, off = synth.generate(synth.SynthSpec(6 << 20, 30_000, 17, bytes(range(256)), False))  # that grow it again without reallocating
        freq, merges = None, 600
        opts.update({"check_interval": 4, "delta_headroom_pct": 50, "delta_margin": 0, "delta_floor": 8, "delta_granule": 8})
    elif scenario == "long_words":
        words = [b" " * 900, b"ab" * 700, b"xyz" * 50, b"abcabc", b"  ", b"aaa"] * 3 + [b"hello world"] * 5
        flat, off = helpers.flatten(words)
        freq, merges = None, 120
    else:
        raise ValueError(scenario)
    left, right, merged, count, stats = train_sharded(lambda: _native.Context(0), flat, off, freq, base, merges, 1, rank, world,
                                                      transport=transport, options=opts)
    if transport.endswith("+p2p"):
        assert stats["exchange_p2p"] == 1 and stats["exchanges_sampled"] > 0  # the exchanges really went peer to peer
    toks = list(base)
    out = []
    for l, r, m in zip(left.tolist(), right.tolist(), merged.tolist()):
        out.append((toks[l].hex(), toks[r].hex()))
        if m == len(toks):
            toks.append(toks[l] + toks[r])
    return out, int(stats["n_words"]), int(stats["table_rebuilds"]), int(stats["retiles"])


def gpu_text_sharded(rank, world, dist, path, chunk_size, vocab_size, specials):
    """Text in, merges out: each rank pre-tokenises its chunks of the file on the (shared) GPU."""
    from yet_another_bpe import _native
    from yet_another_bpe.distributed import train_text_sharded
    from yet_another_bpe.trainer import BBPETrainerConfig

    cfg = BBPETrainerConfig(vocab_size=vocab_size, min_frequency=1, special_tokens=specials, chunk_size_bytes=chunk_size)
    model = train_text_sharded(lambda: _native.Context(0), [path], cfg, rank, world, transport="torch", options={"verify": 1})
    return [(a.hex(), b.hex()) for a, b in model.merges], len(model.vocab)


def gpu_fullsize_sharded(rank, world, dist, transport="torch"):
    """BASELINE configs[3]: the 1 GiB / 32,000-merge job word-sharded over `world` ranks that share the test box's one GPU
    (custom transport, gloo underneath).  Every rank generates the corpus on the device, trains on its word range and
    returns the digest of its (left, right, merged) id triples, its merge count and the corpus digest inputs."""
    import hashlib

    import numpy as np

    from tests import helpers
    from yet_another_bpe import _native, synth
    from yet_another_bpe.distributed import ShardedRunner

    spec = synth.SynthSpec.config3(1024 << 20)
    base = helpers.base_tokens(["<|endoftext|>"])
    with _native.Context(0) as gen:
        pb, po, nw, nb = gen.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
        runner = ShardedRunner(gen, pb, po, nw, nb, base, rank, world, 0, transport=transport)
        try:
            res = runner.run(32000, 1)
            mism = runner._ctx.verify_table()
        finally:
            runner.close()
    h = hashlib.sha256(res["left"].astype(np.uint32).tobytes() + res["right"].astype(np.uint32).tobytes() + res["merged"].astype(np.uint32).tobytes()).hexdigest()
    st = res["stats"]
    xus = 1e3 * st["exchange_ms_sampled"] / max(1, st["exchanges_sampled"]) if st["exchange_p2p"] else -1.0
    return h, int(res["n_merges"]), int(nw), int(nb), int(mism), int(st["n_words"]), int(st["exchanges"]), int(st["exchange_growths"]), float(xus), float(st["train_ms"])


def gpu_device_text_sharded(rank, world, dist, meta_name, n_merges=None):
    """BASELINE configs[4] shape at the size of tests/golden/<meta_name>: every rank regenerates the synthetic text on the
    (shared) GPU, pre-tokenises its chunks, pools them and joins the merge loop.  -> (id-triples digest, merges, pre-tokens here, words here)."""
    import hashlib
    import json

    import numpy as np

    from yet_another_bpe import _native, synth
    from yet_another_bpe.distributed import train_device_text_sharded
    from yet_another_bpe.trainer import BBPETrainerConfig

    meta = json.loads((REPO / "tests" / "golden" / meta_name).read_text())
    g = meta["generator"]
    lb, lo = synth.text_lexicon(g["n_types"], g["seed"])

    def make_text(ctx):
        tb, _to, _np, tn = ctx.synth_generate_lex(g["target_bytes"], g["seed"], lb, lo)
        assert tn == meta["text_bytes"]
        return tb, tn

    cfg = BBPETrainerConfig(vocab_size=257 + (n_merges or meta["n_merges"]), min_frequency=meta["min_frequency"], special_tokens=meta["special_tokens"],
                            chunk_size_bytes=meta["chunk_size_bytes"])
    # (the exchanges go peer to peer: through host memory -- gloo -- the 50,000-merge job spends a minute in the transport alone)
    left, right, merged, count, st, n_pre = train_device_text_sharded(lambda: _native.Context(0), make_text, cfg, rank, world, transport="torch+p2p")
    assert st["exchange_p2p"] == 1
    h = hashlib.sha256(left.astype(np.uint32).tobytes() + right.astype(np.uint32).tobytes() + merged.astype(np.uint32).tobytes()).hexdigest()
    # the byte-level merges list as the golden files serialise it (for a prefix of the job only the prefix digests apply)
    toks = [bytes([b]) for b in range(256)] + [t.encode() for t in meta["special_tokens"]]
    lines = []
    for l, r, m in zip(left.tolist(), right.tolist(), merged.tolist()):
        lines.append(f"{toks[l].hex()} {toks[r].hex()}\n")
        if m == len(toks):
            toks.append(toks[l] + toks[r])
    hm = hashlib.sha256("".join(lines).encode()).hexdigest()
    return h, len(left), int(n_pre), int(st["n_words"]), int(st["n_long_words"]), hm
