import sys
sys.path.insert(0, "yet-another-bpe_amd")
from yet_another_bpe import _native, synth
spec = synth.SynthSpec.config3(1024 << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    with _native.Context() as ctx:
        ctx.set_vocab(base); ctx.load_words_ptr(pb, po, nw)
        prev = None; done = 0
        for m in (6000, 7000, 12000, 13000, 20000, 21000, 30000, 31000):
            ctx.train(m - done, 1); done = m
            st = ctx.stats()
            if prev and m - prev[0] == 1000:
                dl = st["scan_skip_launches"] - prev[1]["scan_skip_launches"]
                dr = st["scan_skip_tiles_read"] - prev[1]["scan_skip_tiles_read"]
                ds = prev[1]["tokens_now"] - st["tokens_now"]
                print(f"merges {prev[0]}..{m}: tiles {st['n_tiles']}  read/launch {dr/dl:9.1f} ({dr/dl/st['n_tiles']*100:.2f} %)  sites/launch {ds/dl:9.1f}  false positives/launch {(dr-ds)/dl:9.1f} ({(dr-ds)/dl/st['n_tiles']*100:.3f} %)")
            prev = (m, st)
