"""ctypes binding of libyabpe.so (C ABI: include/yabpe.h) -- the only way the Python host reaches the GPU.

No fallback of any kind: if the library is not built, or no MI355X is visible, this raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int64, c_uint8, c_uint32, c_uint64, c_void_p
from pathlib import Path

import numpy as np

_CSRC = Path(__file__).resolve().parent.parent / "csrc"
LIB_PATH = Path(os.environ.get("YABPE_LIB", _CSRC / "libyabpe.so"))


class YabpeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"yabpe error {code}: {msg}")
        self.code = code


class Utf8Error(ValueError):
    """yabpe_pretokenize found malformed UTF-8; .position = UnicodeDecodeError.start inside the buffer."""

    def __init__(self, position: int):
        super().__init__(f"invalid UTF-8 at byte {position}")
        self.position = position


E_UTF8 = -7


class Stats(ctypes.Structure):
    _fields_ = [(n, c_uint64) for n in (
        "n_words", "n_words_input", "n_long_words", "tokens_initial", "tokens_now", "merges_done", "n_tiles",
        "live_slots", "table_capacity", "table_entries", "retiles", "table_rebuilds")] + [
        ("load_ms", c_double), ("train_ms", c_double), ("apply_ms_sampled", c_double)] + [(n, c_uint64) for n in (
            "apply_launches_sampled", "apply_algo_bytes_sampled", "apply_actual_bytes_sampled", "algo_bytes_total")] + [
        ("scan_ms_sampled", c_double)] + [(n, c_uint64) for n in (
            "scan_launches_sampled", "scan_algo_bytes_sampled", "scan_actual_bytes_sampled", "scan_skip_launches", "scan_skip_tiles_read", "cand_rebuilds", "cand_rescans", "fused_launches")] + [
        ("dense_ms_sampled", c_double)] + [(n, c_uint64) for n in ("dense_launches_sampled", "dense_algo_bytes_sampled", "dense_actual_bytes_sampled")] + [
        ("sparse_ms", c_double), ("sparse_merges", c_uint64), ("tail_ms", c_double), ("tail_merges", c_uint64)] + [
        (n, c_uint64) for n in ("exchanges", "exchange_bytes", "exchange_cap_records", "exchange_growths", "exchange_max_records", "sparse_launches", "tail_launches")] + [("exchange_ms_sampled", c_double), ("exchanges_sampled", c_uint64), ("exchange_p2p", c_uint64),
        ("dense_launches", c_uint64), ("dense_merges", c_uint64)]


class Latency(ctypes.Structure):
    _fields_ = [(n, c_double) for n in ("launch_gap_us", "load_trip_us", "coherent_trip_us", "atomic_trip_us")]


_lib = None
# int fn(void *user, const void *send_dev, void *recv_dev, uint64_t nbytes)  (include/yabpe.h: yabpe_allgather_fn)
ALLGATHER_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_uint64)

# every symbol include/yabpe.h declares (tests/test_abi.py checks the library exports all of them)
SYMBOLS = [
    "yabpe_abi_version", "yabpe_device_count", "yabpe_create", "yabpe_destroy", "yabpe_last_error", "yabpe_set_option",
    "yabpe_set_vocab", "yabpe_load_words", "yabpe_train", "yabpe_n_tokens", "yabpe_token_bytes", "yabpe_stats",
    "yabpe_iter_log", "yabpe_event_log", "yabpe_latency_probe", "yabpe_verify_table", "yabpe_stream_checksum", "yabpe_synth_generate", "yabpe_synth_generate_lex", "yabpe_synth_free",
    "yabpe_memcpy_d2h", "yabpe_memcpy_h2d", "yabpe_pretokenize", "yabpe_pretokenize_free",
    "yabpe_comm_unique_id", "yabpe_comm_init", "yabpe_comm_init_custom", "yabpe_comm_enable_p2p",
]


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C {_CSRC}` (hipcc --offload-arch=gfx950). "
                "The BPE hot path has no CPU fallback.")
        L = ctypes.CDLL(str(LIB_PATH))
        L.yabpe_abi_version.restype = c_int
        L.yabpe_device_count.restype = c_int
        L.yabpe_create.argtypes = [POINTER(c_void_p), c_int]
        L.yabpe_destroy.argtypes = [c_void_p]
        L.yabpe_destroy.restype = None
        L.yabpe_last_error.argtypes = [c_void_p]
        L.yabpe_last_error.restype = c_char_p
        L.yabpe_set_option.argtypes = [c_void_p, c_char_p, c_int64]
        L.yabpe_set_vocab.argtypes = [c_void_p, c_void_p, c_void_p, c_uint32]
        L.yabpe_load_words.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_uint32]
        L.yabpe_train.argtypes = [c_void_p, c_uint32, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_uint32)]
        L.yabpe_n_tokens.argtypes = [c_void_p, POINTER(c_uint32)]
        L.yabpe_token_bytes.argtypes = [c_void_p, c_uint32, c_void_p, c_uint32, POINTER(c_uint32)]
        L.yabpe_stats.argtypes = [c_void_p, POINTER(Stats)]
        L.yabpe_iter_log.argtypes = [c_void_p, c_void_p, c_void_p, c_uint32, POINTER(c_uint32)]
        L.yabpe_event_log.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, POINTER(c_uint32)]
        L.yabpe_verify_table.argtypes = [c_void_p, POINTER(c_uint64)]
        L.yabpe_stream_checksum.argtypes = [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]
        L.yabpe_synth_generate.argtypes = [c_void_p, c_uint64, c_uint32, c_uint64, c_void_p, c_uint32, c_int,
                                           POINTER(c_void_p), POINTER(c_void_p), POINTER(c_uint64), POINTER(c_uint64)]
        L.yabpe_synth_generate_lex.argtypes = [c_void_p, c_uint64, c_uint32, c_uint64, c_void_p, c_void_p,
                                               POINTER(c_void_p), POINTER(c_void_p), POINTER(c_uint64), POINTER(c_uint64)]
        L.yabpe_synth_free.argtypes = [c_void_p]
        L.yabpe_pretokenize.argtypes = [c_void_p, c_void_p, c_uint64, c_void_p, c_uint32, c_void_p, c_void_p, c_uint32,
                                        POINTER(c_void_p), POINTER(c_void_p), POINTER(c_uint64), POINTER(ctypes.c_int64)]
        L.yabpe_pretokenize_free.argtypes = [c_void_p]
        L.yabpe_memcpy_d2h.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64]
        L.yabpe_memcpy_h2d.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64]
        L.yabpe_comm_unique_id.argtypes = [c_void_p]
        L.yabpe_comm_init.argtypes = [c_void_p, c_int, c_int, c_void_p]
        L.yabpe_comm_init_custom.argtypes = [c_void_p, c_int, c_int, ALLGATHER_FN, c_void_p]
        L.yabpe_comm_enable_p2p.argtypes = [c_void_p]
        if L.yabpe_abi_version() != 2:
            raise ImportError("libyabpe.so ABI version mismatch")
        _lib = L
    return _lib


LOAD_DEDUP = 0x1


class Context:
    """One GPU context == one merge-loop run (or several yabpe_train continuations)."""

    def __init__(self, device: int | None = None):
        L = lib()
        if device is None:
            device = int(os.environ.get("YABPE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self._h = c_void_p()
        rc = L.yabpe_create(byref(self._h), device)
        if rc != 0:
            raise YabpeError(rc, L.yabpe_last_error(None).decode())
        self.device = device
        for k, v in os.environ.items():  # YABPE_OPT_check_interval=16 etc.
            if k.startswith("YABPE_OPT_"):
                self.set_option(k[len("YABPE_OPT_"):], int(v))

    # -- lifetime
    def close(self) -> None:
        if self._h:
            lib().yabpe_destroy(self._h)
            self._h = c_void_p()

    def __enter__(self) -> "Context":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int) -> None:
        if rc != 0:
            raise YabpeError(rc, lib().yabpe_last_error(self._h).decode())

    # -- API
    def set_option(self, name: str, value: int) -> None:
        self._chk(lib().yabpe_set_option(self._h, name.encode(), int(value)))

    def set_vocab(self, tokens: list[bytes]) -> None:
        blob = np.frombuffer(b"".join(tokens), dtype=np.uint8)
        off = np.zeros(len(tokens) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(t) for t in tokens])
        self._base = list(tokens)
        self._chk(lib().yabpe_set_vocab(self._h, blob.ctypes.data, off.ctypes.data, len(tokens)))

    def load_words(self, flat, off, freq=None, dedup: bool = False) -> None:
        """flat: u8 bytes, off: u64 offsets (n+1), freq: optional u64 counts.  numpy arrays (host) or
        integer device addresses wrapped as (ptr, n) via load_words_ptr."""
        flat = np.ascontiguousarray(flat, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        fq = None
        if freq is not None:
            fq = np.ascontiguousarray(freq, dtype=np.uint64)
            assert len(fq) == n
        self._keep = (flat, off, fq)
        self._chk(lib().yabpe_load_words(self._h, flat.ctypes.data if flat.size else None, off.ctypes.data,
                                         fq.ctypes.data if fq is not None else None, n, LOAD_DEDUP if dedup else 0))

    def load_words_ptr(self, bytes_ptr: int, off_ptr: int, n_words: int, freq_ptr: int = 0, dedup: bool = False) -> None:
        """Device (or host) addresses, e.g. from synth_generate() or torch tensors' data_ptr()."""
        self._chk(lib().yabpe_load_words(self._h, c_void_p(bytes_ptr), c_void_p(off_ptr),
                                         c_void_p(freq_ptr) if freq_ptr else None, n_words, LOAD_DEDUP if dedup else 0))

    def train(self, num_merges: int, min_frequency: int):
        left = np.zeros(max(num_merges, 1), dtype=np.uint32)
        right = np.zeros_like(left)
        merged = np.zeros_like(left)
        count = np.zeros(max(num_merges, 1), dtype=np.uint64)
        n = c_uint32(0)
        self._chk(lib().yabpe_train(self._h, num_merges, min_frequency, left.ctypes.data, right.ctypes.data,
                                    merged.ctypes.data, count.ctypes.data, byref(n)))
        k = n.value
        return left[:k], right[:k], merged[:k], count[:k]

    def n_tokens(self) -> int:
        n = c_uint32(0)
        self._chk(lib().yabpe_n_tokens(self._h, byref(n)))
        return n.value

    def token_bytes(self, tid: int) -> bytes:
        ln = c_uint32(0)
        self._chk(lib().yabpe_token_bytes(self._h, tid, None, 0, byref(ln)))
        buf = (c_uint8 * max(ln.value, 1))()
        self._chk(lib().yabpe_token_bytes(self._h, tid, buf, ln.value, byref(ln)))
        return bytes(buf[:ln.value])

    def stats(self) -> dict:
        s = Stats()
        self._chk(lib().yabpe_stats(self._h, byref(s)))
        return {f: getattr(s, f) for f, _ in Stats._fields_}

    def iter_log(self):
        n = c_uint32(0)
        self._chk(lib().yabpe_iter_log(self._h, None, None, 0, byref(n)))
        sites = np.zeros(max(n.value, 1), dtype=np.uint64)
        live = np.zeros(max(n.value, 1), dtype=np.uint64)
        self._chk(lib().yabpe_iter_log(self._h, sites.ctypes.data, live.ctypes.data, n.value, byref(n)))
        return sites[:n.value], live[:n.value]

    def event_log(self):
        n = c_uint32(0)
        self._chk(lib().yabpe_event_log(self._h, None, None, None, 0, byref(n)))
        it = np.zeros(max(n.value, 1), dtype=np.uint32)
        us = np.zeros(max(n.value, 1), dtype=np.float32)
        scan = np.zeros(max(n.value, 1), dtype=np.float32)
        self._chk(lib().yabpe_event_log(self._h, it.ctypes.data, us.ctypes.data, scan.ctypes.data, n.value, byref(n)))
        return it[:n.value], us[:n.value], scan[:n.value]

    def latency_probe(self) -> dict:
        """Measured latency pieces of one sparse merge on this device (include/yabpe.h yabpe_latency_t), microseconds."""
        s = Latency()
        self._chk(lib().yabpe_latency_probe(self._h, byref(s)))
        return {f: getattr(s, f) for f, _ in Latency._fields_}

    def verify_table(self) -> int:
        m = c_uint64(0)
        self._chk(lib().yabpe_verify_table(self._h, byref(m)))
        return m.value

    def stream_checksum(self) -> tuple[int, int, int]:
        a, b, c = c_uint64(0), c_uint64(0), c_uint64(0)
        self._chk(lib().yabpe_stream_checksum(self._h, byref(a), byref(b), byref(c)))
        return a.value, b.value, c.value

    def synth_generate(self, target_bytes: int, n_types: int, seed: int, alphabet: bytes, space_prefix: bool):
        """-> (dev_bytes_ptr, dev_off_ptr, n_words, n_bytes); buffers live until close()/synth_free()."""
        al = np.frombuffer(alphabet, dtype=np.uint8)
        pb, po, nw, nb = c_void_p(), c_void_p(), c_uint64(0), c_uint64(0)
        self._chk(lib().yabpe_synth_generate(self._h, target_bytes, n_types, seed, al.ctypes.data, len(al),
                                             1 if space_prefix else 0, byref(pb), byref(po), byref(nw), byref(nb)))
        return pb.value, po.value, nw.value, nb.value

    def synth_generate_lex(self, target_bytes: int, seed: int, lex_bytes: np.ndarray, lex_off: np.ndarray):
        """Synthetic text from a lexicon (synth.text_lexicon): -> (dev_bytes_ptr, dev_piece_off_ptr, n_pieces, n_bytes)."""
        lb = np.ascontiguousarray(lex_bytes, dtype=np.uint8)
        lo = np.ascontiguousarray(lex_off, dtype=np.uint64)
        pb, po, nw, nb = c_void_p(), c_void_p(), c_uint64(0), c_uint64(0)
        self._chk(lib().yabpe_synth_generate_lex(self._h, c_uint64(target_bytes), c_uint32(len(lo) - 1), c_uint64(seed), c_void_p(lb.ctypes.data),
                                                 c_void_p(lo.ctypes.data), byref(pb), byref(po), byref(nw), byref(nb)))
        return pb.value, po.value, nw.value, nb.value

    def synth_free(self) -> None:
        self._chk(lib().yabpe_synth_free(self._h))

    def pretokenize(self, text, n_bytes: int | None = None, chunk_starts=None, special_tokens=()):
        """GPT-2 pre-tokenisation on the device (reference trainer.py:136-214).  `text`: bytes / u8 array (staged) or a
        device address (then n_bytes is required).  chunk_starts: ascending chunk starts, first one 0.
        -> (dev_text_ptr, dev_word_off_ptr, n_words); raises Utf8Error(position) on malformed UTF-8."""
        keep = None
        if isinstance(text, int):
            ptr, n = c_void_p(text), int(n_bytes)
        else:
            keep = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, dtype=np.uint8)
            ptr, n = c_void_p(keep.ctypes.data if keep.size else 0), int(keep.size)
        ch = np.ascontiguousarray(chunk_starts if chunk_starts is not None and len(chunk_starts) else [0], dtype=np.uint64)
        sb = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in special_tokens]
        spb = np.frombuffer(b"".join(sb) or b"\0", dtype=np.uint8)
        spo = np.zeros(len(sb) + 1, dtype=np.uint32)
        if sb:
            spo[1:] = np.cumsum([len(x) for x in sb])
        dt, do, nw, bad = c_void_p(), c_void_p(), c_uint64(0), ctypes.c_int64(-1)
        rc = lib().yabpe_pretokenize(self._h, ptr, n, ch.ctypes.data, len(ch), spb.ctypes.data, spo.ctypes.data, len(sb),
                                     byref(dt), byref(do), byref(nw), byref(bad))
        if rc == E_UTF8:
            raise Utf8Error(bad.value)
        self._chk(rc)
        return dt.value or 0, do.value, nw.value

    def pretokenize_free(self) -> None:
        self._chk(lib().yabpe_pretokenize_free(self._h))

    def h2d(self, dev_ptr: int, arr: np.ndarray) -> None:
        arr = np.ascontiguousarray(arr)
        self._chk(lib().yabpe_memcpy_h2d(self._h, c_void_p(dev_ptr), arr.ctypes.data, arr.nbytes))

    # -- multi-GPU
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (c_uint8 * 128)()
        rc = lib().yabpe_comm_unique_id(buf)
        if rc != 0:
            raise YabpeError(rc, lib().yabpe_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, rank: int, n_ranks: int, unique_id: bytes) -> None:
        """RCCL transport (one process per GPU).  Call before load_words."""
        assert len(unique_id) == 128
        buf = (c_uint8 * 128).from_buffer_copy(unique_id)
        self._chk(lib().yabpe_comm_init(self._h, rank, n_ranks, buf))

    def comm_init_custom(self, rank: int, n_ranks: int, allgather) -> None:
        """Custom transport: allgather(send_dev_ptr, recv_dev_ptr, nbytes) -> 0 on success."""
        def _cb(_user, send, recv, nbytes):
            try:
                return int(allgather(send, recv, nbytes) or 0)
            except Exception as e:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return -1
        self._ag_cb = ALLGATHER_FN(_cb)  # keep alive
        self._chk(lib().yabpe_comm_init_custom(self._h, rank, n_ranks, self._ag_cb, None))

    def comm_enable_p2p(self) -> None:
        """Peer-to-peer exchange (hipIpc-mapped receive buffers) instead of the all-gather; collective."""
        self._chk(lib().yabpe_comm_enable_p2p(self._h))

    def d2h(self, dev_ptr: int, nbytes: int, dtype=np.uint8) -> np.ndarray:
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._chk(lib().yabpe_memcpy_d2h(self._h, out.ctypes.data, c_void_p(dev_ptr), nbytes))
        return out


def train_words(words_flat, words_off, freq, base_tokens: list[bytes], num_merges: int, min_frequency: int,
                dedup: bool = False, options: dict | None = None, want_stats: bool = False):
    """Convenience used by tests/bench: returns (vocab, merges[, stats]) like _merge_loop."""
    with Context() as ctx:
        for k, v in (options or {}).items():
            ctx.set_option(k, v)
        ctx.set_vocab(base_tokens)
        ctx.load_words(words_flat, words_off, freq, dedup=dedup)
        left, right, merged, count = ctx.train(num_merges, min_frequency)
        stats = ctx.stats() if want_stats else None
    toks = list(base_tokens)
    merges = []
    for l, r, m in zip(left.tolist(), right.tolist(), merged.tolist()):
        merges.append((toks[l], toks[r]))
        if m == len(toks):
            toks.append(toks[l] + toks[r])
        else:
            assert toks[m] == toks[l] + toks[r], "merged id does not name left+right"
    vocab = {t: i for i, t in enumerate(toks)}
    return (vocab, merges, stats) if want_stats else (vocab, merges)
