"""Multi-GPU driver: one process per GPU, words sharded by contiguous ranges, pair-count deltas exchanged with one
all-gather per merge (RCCL over xGMI from inside libyabpe.so; see include/yabpe.h "Multi-GPU").

The reference has no distributed code at all; this is new design.  Words are independent units --
pairs never cross a word (reference trainer.py:232) -- so any partition of the words gives the same global counts.
`torch.distributed` is used only for the rendezvous (broadcast of the 128-byte RCCL id) and, in tests, as a
stand-in transport.
"""
from __future__ import annotations

import os

import ctypes
from typing import Callable, Sequence

import numpy as np


def plan_shards(off: np.ndarray, world: int) -> list[tuple[int, int]]:
    """Contiguous word ranges [w0, w1) per rank, balanced by resident slots (bytes + one separator per word).

    off: u64 offsets (n_words + 1), host array."""
    n = len(off) - 1
    base = int(off[0])
    total = int(off[n]) - base + n
    bounds = [0]
    packed = off.astype(np.int64) - base + np.arange(n + 1, dtype=np.int64)  # monotone
    for r in range(1, world):
        bounds.append(int(np.searchsorted(packed, (total * r) // world, side="left")))
    bounds.append(n)
    bounds = [min(max(b, 0), n) for b in bounds]
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def plan_shards_device(read_off: Callable[[int], int], n_words: int, world: int) -> list[tuple[int, int]]:
    """Same plan for an offsets array that lives on the device: read_off(i) returns off[i] (a few dozen reads)."""
    base = read_off(0)
    total = read_off(n_words) - base + n_words
    bounds = [0]
    for r in range(1, world):
        target = (total * r) // world
        lo, hi = 0, n_words
        while lo < hi:  # first w with (off[w] - base + w) >= target
            mid = (lo + hi) // 2
            if read_off(mid) - base + mid >= target:
                hi = mid
            else:
                lo = mid + 1
        bounds.append(max(lo, bounds[-1]))
    bounds.append(n_words)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


class TorchTransport:
    """yabpe_allgather_fn over torch.distributed (any backend).  Used by the tests to run several ranks on one GPU
    (gloo); production uses RCCL inside the library instead.  to_host/to_dev move bytes between the "device"
    buffers the library hands out and host memory."""

    def __init__(self, to_host: Callable[[int, int], np.ndarray], to_dev: Callable[[int, np.ndarray], None], group=None):
        self.to_host, self.to_dev, self.group = to_host, to_dev, group
        self.calls = 0
        self.bytes = 0

    def __call__(self, send_ptr: int, recv_ptr: int, nbytes: int) -> int:
        import torch
        import torch.distributed as dist

        world = dist.get_world_size(self.group)
        mine = torch.from_numpy(self.to_host(send_ptr, nbytes).copy())
        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, mine, group=self.group)
        self.to_dev(recv_ptr, torch.cat(parts).numpy())
        self.calls += 1
        self.bytes += nbytes * world
        return 0


def host_memory_transport(group=None) -> TorchTransport:
    """Transport whose "device" pointers are plain host addresses (CPU-only tests of the callback plumbing)."""
    def to_host(ptr, n):
        return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(ptr)).copy()

    def to_dev(ptr, arr):
        ctypes.memmove(ptr, arr.ctypes.data, arr.nbytes)

    return TorchTransport(to_host, to_dev, group)


def attach(ctx, rank: int, world: int, transport: str = "rccl") -> None:
    """Attach a communicator to a _native.Context (before load_words).  transport: "rccl" | "torch", with "+p2p" appended (or
    YABPE_P2P=1): the per-batch exchange then goes peer to peer through hipIpc-mapped buffers (yabpe_comm_enable_p2p) and the
    named transport only carries the handles and the rare small agreements."""
    if world == 1:
        return
    p2p = transport.endswith("+p2p") or os.environ.get("YABPE_P2P") == "1"
    transport = transport[:-4] if transport.endswith("+p2p") else transport
    _attach_transport(ctx, rank, world, transport)
    if p2p:
        ctx.comm_enable_p2p()


def _attach_transport(ctx, rank: int, world: int, transport: str) -> None:
    import torch
    import torch.distributed as dist

    if transport == "rccl":
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid = torch.tensor(list(ctx.comm_unique_id()), dtype=torch.uint8, device=dev)
        dist.broadcast(uid, src=0)
        ctx.comm_init(rank, world, bytes(uid.cpu().tolist()))
    elif transport == "torch":
        ctx.comm_init_custom(rank, world, TorchTransport(lambda p, n: ctx.d2h(p, n), lambda p, a: ctx.h2d(p, a)))
    else:
        raise ValueError(transport)


def train_sharded(ctx_factory, flat: np.ndarray, off: np.ndarray, freq, base_tokens: Sequence[bytes], num_merges: int,
                  min_frequency: int, rank: int, world: int, transport: str = "rccl", options: dict | None = None):
    """Host arrays: every rank passes the SAME full corpus; each loads its own shard.  Returns
    (left, right, merged, count, stats)."""
    shards = plan_shards(off, world)
    w0, w1 = shards[rank]
    with ctx_factory() as ctx:
        for k, v in (options or {}).items():
            ctx.set_option(k, v)
        ctx.set_vocab(list(base_tokens))
        attach(ctx, rank, world, transport)
        sub_off = off[w0:w1 + 1]
        sub_freq = None if freq is None else freq[w0:w1]
        ctx.load_words(flat, sub_off, sub_freq)
        left, right, merged, count = ctx.train(num_merges, min_frequency)
        return left, right, merged, count, ctx.stats()


def plan_chunk_shards(sizes: Sequence[int], world: int) -> list[tuple[int, int]]:
    """Contiguous chunk ranges [c0, c1) per rank, balanced by bytes (chunks are the units the pre-tokeniser treats as
    separate texts, reference trainer.py:172-198, so any assignment of whole chunks gives the same pre-tokens)."""
    cum = np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))])
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world):
        bounds.append(max(bounds[-1], int(np.searchsorted(cum, (total * r) // world, side="left"))))
    bounds.append(len(sizes))
    return [(min(bounds[r], len(sizes)), min(bounds[r + 1], len(sizes))) for r in range(world)]


def train_text_sharded(ctx_factory, files, config, rank: int, world: int, transport: str = "rccl", options: dict | None = None):
    """BBPETrainer.train() over several GPUs, text in, model out (BASELINE configs[4] shape): every rank reads ITS chunks
    of the files, pre-tokenises them on its GPU (yabpe_pretokenize), pools equal pre-tokens locally and joins the
    collective merge loop.  Chunk cuts are the reference's (config.chunk_size_bytes), so the result equals the
    single-process train() with the same config.  Returns a BBPEModel (the same on every rank)."""
    from pathlib import Path

    from . import _native
    from .trainer import BBPEModel, BBPETrainer

    tr = BBPETrainer(config)
    paths = [Path(f) for f in files]
    for p in paths:
        if not p.exists():
            raise FileNotFoundError(f"File not found: {p}")
    chunks = [(p, a, b) for p in paths for a, b in tr._chunk_ranges(p)]
    c0, c1 = plan_chunk_shards([b - a for _, a, b in chunks], world)[rank] if chunks else (0, 0)
    pieces, starts, total = [], [], 0
    for p, a, b in chunks[c0:c1]:
        starts.append(total)
        pieces.append(np.fromfile(p, dtype=np.uint8, count=b - a, offset=a))
        total += b - a
    base = tr._base_tokens()
    specials = list(config.special_tokens)
    num_merges = max(0, config.vocab_size - len(base))
    with ctx_factory() as ctx:
        for k, v in (options or {}).items():
            ctx.set_option(k, v)
        ctx.set_vocab(base)
        attach(ctx, rank, world, transport)
        err = None
        n_words = 0
        if total:
            text = pieces[0] if len(pieces) == 1 else np.concatenate(pieces)
            try:
                dev_text, dev_off, n_words = ctx.pretokenize(text, chunk_starts=starts, special_tokens=specials)
            except _native.Utf8Error as e:
                k = max(i for i, s0 in enumerate(starts) if s0 <= e.position)
                p, a, _ = chunks[c0 + k]
                err = (c0 + k, f"File {p} contains invalid UTF-8 at position {a + e.position - starts[k]}.")
        if world > 1:  # every rank must learn about a bad chunk anywhere, or the others would wait in the merge loop
            import torch.distributed as dist

            errs = [None] * world
            dist.all_gather_object(errs, err)
            err = min((e for e in errs if e), default=None)
        if err:
            raise ValueError(err[1])
        if n_words:
            ctx.load_words_ptr(dev_text, dev_off, n_words, dedup=True)
        else:
            ctx.load_words(np.zeros(0, np.uint8), np.zeros(1, np.uint64), None, dedup=True)  # no words here: same layout as the peers
        left, right, merged, _count = ctx.train(num_merges, int(config.min_frequency))
    vocab, merges = BBPETrainer._decode_merges(base, left, right, merged)
    return BBPEModel(vocab=vocab, merges=merges, special_tokens=specials)


def train_device_text_sharded(ctx_factory, make_text, config, rank: int, world: int, transport: str = "rccl", options: dict | None = None):
    """train_text_sharded for text that is produced ON the device (synthetic corpora at sizes a test cannot keep on disk):
    `make_text(ctx) -> (dev_ptr, n_bytes)` runs on every rank; the chunk cuts are the reference's (config.chunk_size_bytes),
    this rank pre-tokenises ITS chunks, pools the pre-tokens and joins the collective merge loop.
    Returns (left, right, merged, count, stats, n_pretokens_here)."""
    from .trainer import BBPETrainer, chunk_ranges

    tr = BBPETrainer(config)
    base = tr._base_tokens()
    specials = list(config.special_tokens)
    num_merges = max(0, config.vocab_size - len(base))
    with ctx_factory() as ctx:
        for k, v in (options or {}).items():
            ctx.set_option(k, v)
        ctx.set_vocab(base)
        attach(ctx, rank, world, transport)
        ptr, n_bytes = make_text(ctx)
        ranges = chunk_ranges(n_bytes, config.chunk_size_bytes, lambda off, n: ctx.d2h(ptr + off, n).tobytes())
        c0, c1 = plan_chunk_shards([b - a for a, b in ranges], world)[rank]
        mine = ranges[c0:c1]
        n_words = 0
        if mine:
            a0 = mine[0][0]
            assert all(mine[i][1] == mine[i + 1][0] for i in range(len(mine) - 1)), "chunks of valid UTF-8 follow one another"
            dev_text, dev_off, n_words = ctx.pretokenize(ptr + a0, n_bytes=mine[-1][1] - a0, chunk_starts=[a - a0 for a, _ in mine],
                                                         special_tokens=specials)
        if n_words:
            ctx.load_words_ptr(dev_text, dev_off, n_words, dedup=True)
        else:
            ctx.load_words(np.zeros(0, np.uint8), np.zeros(1, np.uint64), None, dedup=True)  # no words here: same layout as the peers
        left, right, merged, count = ctx.train(num_merges, int(config.min_frequency))
        return left, right, merged, count, ctx.stats(), n_words


class ShardedRunner:
    """bench.py helper: the corpus is already on every rank's GPU; each rank trains on its word range."""

    def __init__(self, gen_ctx, bytes_ptr: int, off_ptr: int, n_words: int, n_bytes: int, base_tokens, rank: int,
                 world: int, local_rank: int, transport: str = "rccl"):
        self.bytes_ptr, self.off_ptr, self.base = bytes_ptr, off_ptr, list(base_tokens)
        self.rank, self.world, self.local_rank, self.transport = rank, world, local_rank, transport

        def read_off(i: int) -> int:
            return int(gen_ctx.d2h(off_ptr + 8 * i, 8, dtype=np.uint64)[0])

        self.w0, self.w1 = plan_shards_device(read_off, n_words, world)[rank]

    def _context(self):
        """One context + communicator for all jobs (RCCL communicator creation is not part of a training job)."""
        from . import _native

        if getattr(self, "_ctx", None) is None:
            self._ctx = _native.Context(self.local_rank)
            self._ctx.set_vocab(self.base)
            attach(self._ctx, self.rank, self.world, self.transport)
        return self._ctx

    def run(self, num_merges: int, min_frequency: int, dedup: bool = False, event_sample: int = 0,
            options: dict | None = None) -> dict:
        ctx = self._context()
        ctx.set_option("event_sample", event_sample)
        for k, v in (options or {}).items():
            ctx.set_option(k, v)
        ctx.set_vocab(self.base)  # resets tokens / merge state; the communicator stays attached
        ctx.load_words_ptr(self.bytes_ptr, self.off_ptr + 8 * self.w0, self.w1 - self.w0, dedup=dedup)
        left, right, merged, count = ctx.train(num_merges, min_frequency)
        return {"n_merges": len(left), "stats": ctx.stats(), "left": left, "right": right, "merged": merged}

    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None:
            self._ctx.close()
            self._ctx = None
