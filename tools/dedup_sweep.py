import sys, time
sys.path.insert(0, "yet-another-bpe_amd")
from yet_another_bpe import _native, synth
spec = synth.SynthSpec.config3(1024 << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    for cfg in sys.argv[1:] or [""]:
        with _native.Context() as ctx:
            for kv in filter(None, cfg.split(",")):
                k, v = kv.split("="); ctx.set_option(k, int(v))
            ctx.set_vocab(base); ctx.load_words_ptr(pb, po, nw, dedup=True)
            l, r, m, c = ctx.train(32000, 1); st = ctx.stats()
            print(f"dedup {cfg or '(defaults)':30s} train {st['train_ms']:.1f} ms load {st['load_ms']:.1f} ms  tiles {st['n_tiles']}", flush=True)
