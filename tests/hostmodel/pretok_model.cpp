// CPU model of the pre-tokeniser: runs the per-position rules of yet-another-bpe_amd/csrc/pretok_logic.h (the same
// functions the HIP kernels call) over every byte of a text, sequentially.  Test infrastructure only.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../yet-another-bpe_amd/csrc/pretok_logic.h"
#include "../../yet-another-bpe_amd/csrc/unicode_classes.inc"

static std::vector<uint8_t> g_cls;

static void build_table() {
    if (!g_cls.empty()) return;
    g_cls.assign(0x110000, PT_O);
    for (unsigned r = 0; r < YB_UNICODE_CLASS_NRUNS; ++r) {
        const unsigned lo = YB_UNICODE_CLASS_RUNS[r][0];
        const unsigned hi = r + 1 < YB_UNICODE_CLASS_NRUNS ? YB_UNICODE_CLASS_RUNS[r + 1][0] : 0x110000;
        memset(g_cls.data() + lo, (int)YB_UNICODE_CLASS_RUNS[r][1], hi - lo);
    }
}

// flags_out[i] = 1 iff a pre-token starts at byte i.  *err_pos = first malformed byte (UnicodeDecodeError.start) or -1.
extern "C" int pretok_model(const uint8_t *text, uint64_t n, const uint64_t *chunk_off, uint32_t n_chunks,
                            const uint8_t *sp_bytes, const uint32_t *sp_off, uint32_t n_sp, uint8_t *flags_out,
                            int64_t *err_pos) {
    build_table();
    std::vector<uint8_t> meta(n, 0);
    for (uint32_t c = 0; c < n_chunks; ++c)
        if (chunk_off[c] < n) meta[chunk_off[c]] |= PT_CHUNK0;
    *err_pos = -1;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t end = n;
        for (uint64_t k = i + 1; k < i + 4 && k < n; ++k)
            if (meta[k] & PT_CHUNK0) {
                end = k;
                break;
            }
        bool bad = false;
        const PtView v0{text, meta.data(), n, 0};
        const uint8_t m = pt_classify(v0, i, end, g_cls.data(), &bad);
        meta[i] = (uint8_t)((meta[i] & PT_CHUNK0) | m);
        if (bad && *err_pos < 0) *err_pos = (int64_t)i;
    }
    if (*err_pos >= 0) return 0;
    PtView v{text, meta.data(), n, 0};
    for (uint64_t i = 0; i < n; ++i) flags_out[i] = pt_is_start(v, i, -1) ? 1 : 0;
    if (n_sp) {
        uint32_t max_len = 0;
        for (uint32_t s = 0; s < n_sp; ++s) max_len = sp_off[s + 1] - sp_off[s] > max_len ? sp_off[s + 1] - sp_off[s] : max_len;
        PtSpecials sp{sp_bytes, sp_off, n_sp, max_len};
        auto occ = [&](uint64_t q) -> uint32_t { return pt_special_at(v, sp, q); };
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t o = occ(i);
            if (o && pt_special_is_head(v, sp, occ, i)) pt_special_walk(v, sp, occ, flags_out, i, o);
        }
    }
    return 0;
}
