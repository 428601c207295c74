import sys
sys.path.insert(0, "/root/repo/yet-another-bpe_amd"); sys.path.insert(0, "/root/repo")
from yet_another_bpe import _native, synth
spec = synth.SynthSpec.config3(256 << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    with _native.Context() as ctx:
        ctx.set_option("force_comm", 1)
        ctx.set_vocab(base)
        ctx.comm_init(0, 1, _native.Context.comm_unique_id())
        ctx.load_words_ptr(pb, po, nw)
        for k in range(12):
            l, r, m, c = ctx.train(1, 1)
            st = ctx.stats()
            print(k, int(c[0]), "records max so far", st["exchange_max_records"], "cap", st["exchange_cap_records"], "growths", st["exchange_growths"], flush=True)
