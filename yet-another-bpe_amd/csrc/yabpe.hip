// yabpe.hip -- C ABI (include/yabpe.h) over the HIP kernels in yabpe_kernels.h / yabpe_aux_kernels.h.
// Built only for gfx950:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC yabpe.hip -o libyabpe.so
//
// Host control flow of yabpe_train (reference trainer.py:238-300): per merge the host enqueues ONE launch on one stream
// with NO host round trip -- k_apply (streaming phase) or k_scan_skip (sparse phase) applies the merge that DevState names
// and its last workgroup selects the next one (fused_select_tail); every kernel reads the current merge / stop flags from
// DevState in HBM.  The host looks at DevState every `check_interval` merges to stop early (done), to service a halt, to
// grow the pair table, to refresh signatures / the candidate list and to retile the shrinking token stream.
// Multi-GPU: apply launch (updates leave as records) -> one all-gather -> k_delta_apply (+ the selection).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is dlopen()ed when yabpe_comm_* is first used

#include <algorithm>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/yabpe.h"
#include "yabpe_kernels.h"
#include "yabpe_aux_kernels.h"
#include "yabpe_pretok_kernels.h"
#include "unicode_classes.inc"

using namespace yb;

namespace {

struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        // A process can hold two ROCm stacks (the system one and the copy bundled with PyTorch).  RCCL must be the
        // one that sits next to the HIP runtime THIS library is bound to, or it will not see our device.
        std::vector<std::string> names;
        Dl_info info;
        if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash);
                names.push_back(dir + "/librccl.so.1");
                names.push_back(dir + "/librccl.so");
            }
        }
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        for (const auto &name : names) {
            r.h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
        }
        if (r.h) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
            r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
            r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
            if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.AllReduce || !r.CommDestroy) r.h = nullptr;
        }
    }
    return r.h ? &r : nullptr;
}

constexpr uint32_t MAX_APPLY_BLOCKS = 8192;
constexpr uint32_t MAX_LISTS = 2048;  // the sparse launch's scan workgroups at most (blk_read slots)
thread_local std::string g_create_error;

struct EventPair {
    hipEvent_t e0, e1, e2;  // before the apply phase, after its streaming kernel, after its last kernel
    uint32_t iter_rel;
    bool split;
};

}  // namespace

struct yabpe_ctx {
    int device = 0;
    int n_cu = 256;
    int apply_occ = 4;  // workgroups of the streaming-phase kernels that one CU holds at a time (asked of the runtime at creation)
    hipStream_t stream = nullptr;
    std::string err;
    std::map<std::string, int64_t> opt;

    // vocab
    bool have_vocab = false;
    uint32_t base_tokens = 0;
    TokTable tt{};
    // corpus
    bool have_words = false;
    bool weighted = false;
    uint64_t n_words = 0, n_words_input = 0, tokens_initial = 0;
    uint32_t n_tiles = 0;
    uint16_t *tiles = nullptr, *tiles_alt = nullptr;
    uint32_t tiles_cap = 0, tiles_alt_cap = 0;  // capacities in tiles
    uint32_t *tile_len = nullptr, *tile_len_alt = nullptr;
    uint32_t *tile_wbase = nullptr, *tile_wbase_alt = nullptr;
    uint32_t *wfreq = nullptr;
    // long words
    uint32_t n_long = 0;
    uint16_t *long_tok = nullptr;
    unsigned long long *long_sig = nullptr;  // one blocked-Bloom signature per long word (k_apply_long launches only what may hold the pair)
    uint32_t long_sig_stride = 0;
    unsigned long long *long_off = nullptr;
    uint32_t *long_len = nullptr, *long_freq = nullptr;
    uint64_t long_tokens = 0;
    // pair table
    PairTable table{};
    uint64_t table_cap = 0;
    // state
    DevState *st = nullptr;
    DevState *st_host = nullptr;  // pinned
    Best *partials = nullptr;
    uint32_t n_partials = 0, n_partials_cap = 0;
    // records of the last train call
    uint32_t rec_cap = 0, rec_n = 0;
    uint32_t *rec_left = nullptr, *rec_right = nullptr, *rec_merged = nullptr;
    unsigned long long *rec_count = nullptr, *rec_sites = nullptr, *rec_live = nullptr;
    std::vector<uint64_t> log_sites, log_live;
    std::vector<uint32_t> ev_iter;
    std::vector<float> ev_us;
    // stats
    yabpe_stats_t stats{};
    std::vector<EventPair> events;
    // synth buffers
    std::vector<void *> synth_bufs;
    std::vector<void *> pretok_bufs;          // staged text / word offsets handed out by yabpe_pretokenize
    uint8_t *pt_cls = nullptr;                // class per code point (unicode_classes.inc expanded), built on first use
    // misc device scratch
    unsigned long long *scratch64 = nullptr;  // 16 x u64: [0] live sum [1] freq overflow [2,3] long words [4,5,6] verify/checksum
                                              // [7] comm_max [8] exchange record count [9] local count-table entries [10..12] comm_max3
    unsigned long long *blk_stats = nullptr;  // 2 x MAX_APPLY_BLOCKS per-workgroup counters of k_apply
    uint32_t blk_used = 0;                    // largest grid that wrote blk_stats since they were last folded
    bool split_mode = false;         // the sparse form (skip index + batches of merges per launch) instead of the streaming one
    uint32_t kmax_now = 1;           // DevState::kmax as last written
    uint32_t rec_n_live = 0;         // record entries to keep when the record arrays grow
    std::vector<float> ev_scan_us;
    // skip index
    unsigned long long *sig = nullptr;
    uint32_t sig_stride = 0;
    unsigned long long *blk_read = nullptr;  // [MAX_LISTS] tiles read by k_scan_skip, accumulated
    uint64_t scan_skip_launches = 0;
    bool sig_valid = false;
    uint32_t sig_built_at = 0;
    uint64_t sig_tokens_at_build = 0;  // T when the signatures were last built (stale bits grow with the sites merged since)
    uint64_t sig_builds = 0;
    // candidate argmax
    CandState *cand_state = nullptr;
    uint32_t cand_built_at = 0;      // merge index (of this yabpe_train call) at which cand[] was last rebuilt
    unsigned long long cand_best_at_build = 0;
    uint32_t *sel_ticket = nullptr;  // finished-workgroup counters (TICKET_WORDS; the last workgroup selects; they reset themselves)
    unsigned long long *cand = nullptr;
    bool use_cand = false;
    uint64_t cand_rebuilds = 0, cand_rescans = 0;
    double cand_margin = 0.2;        // T = best count x (1 - margin); adapted so that the list stays short
    uint32_t cand_n_at_build = 0;
    // fused per-merge launch: the apply kernel of merge i ends with the selection of merge i + 1
    bool pending = false;            // a merge has been selected (recorded) and not applied yet
    uint64_t fused_launches = 0;
    // multi-GPU
    int rank = 0, n_ranks = 1;
    bool multi = false;  // exchange path active (n_ranks > 1, or a 1-rank communicator forced for testing)
    ncclComm_t comm = nullptr;
    yabpe_allgather_fn ag_fn = nullptr;  // custom transport (tests / other fabrics) instead of RCCL
    void *ag_user = nullptr;
    unsigned long long *xsmall = nullptr;  // n_ranks x u64 receive slots for small agreements
    uint8_t *xsend = nullptr, *xrecv = nullptr;  // [DeltaHdr | xcap DeltaRec] of this rank / of every rank
    uint64_t exchange_growths = 0;  // times the exchange buffers had to grow (a merge produced more records than fit)
    uint64_t exchange_max_records = 0;  // largest record count of one rank seen at a batch end
    uint32_t xcap = 0;      // records per rank the exchange buffers hold
    uint64_t xstride = 0;   // bytes per rank buffer
    uint32_t xeff = 0;      // records per rank an exchange TRANSMITS (<= xcap): follows what the merges really produce, so
                            // that an all-gather moves a few dozen KiB per rank late in a job, not the whole 256 KiB buffer
    uint64_t xeff_sum = 0;  // (statistics: sum of xeff over the exchanges)
    uint32_t xseen[4] = {0, 0, 0, 0};  // the largest record counts of the last four batches
    uint64_t exchanges = 0;
    // peer-to-peer exchange (yabpe_comm_enable_p2p): receive areas mapped through hipIpc handles
    bool p2p = false;
    uint8_t *p2p_area = nullptr;               // this rank's receive area: 2 halves of n_ranks slots + the flag words (plain hipMalloc: exported)
    std::vector<uint8_t *> p2p_peer;           // every rank's area as mapped here (own rank: p2p_area)
    uint64_t p2p_half = 0, p2p_flags_off = 0, p2p_seq = 0, p2p_stride = 0;  // p2p_stride: bytes of one sender's slot in an area
    uint32_t p2p_cap = 0;                      // records a slot holds (fixed for the job: the areas are mapped once)
    hipEvent_t p2p_e0 = nullptr, p2p_e1 = nullptr;  // (statistics: every 64th exchange is timed)
    bool p2p_pending = false;                  // an event pair was recorded and not read yet
};

namespace {

int fail(yabpe_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c)
        c->err = buf;
    else
        g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fail((c), YABPE_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

#define NCCLCHK(c, call)                                                                            \
    do {                                                                                           \
        ncclResult_t r__ = (call);                                                                 \
        if (r__ != ncclSuccess)                                                                    \
            return fail((c), YABPE_E_COMM, "%s failed: %s (%s:%d)", #call,                         \
                        rccl() && rccl()->GetErrorString ? rccl()->GetErrorString(r__) : "?", __FILE__, __LINE__); \
    } while (0)

static bool trace_alloc_on() {
    static const bool on = [] { const char *e = getenv("YABPE_TRACE_ALLOC"); return e && *e == '1'; }();
    return on;
}

// Device allocations go through a small per-process cache: a training job allocates a few large buffers (tiles,
// signatures, worklists, retile targets) and frees them at the end, and the next job asks for the same sizes again.
// hipMalloc / hipFree of multi-GB buffers are host-side stalls of anywhere from 1 ms to 100+ ms depending on the state of
// the driver's page tables (measured: 14 ms vs 340 ms for the same yabpe_load_words on two boxes) -- time the GPU idles.
// Freed blocks are kept (up to YABPE_POOL_MAX_GIB, default 24) and handed out again to requests of the same rounded size.
struct DevPool {
    std::mutex m;
    std::multimap<std::pair<int, size_t>, void *> free_blocks;  // (device, bytes) -> block
    std::map<void *, std::pair<int, size_t>> live;              // block -> (device, bytes)
    size_t held = 0;
    size_t cap = [] { const char *e = getenv("YABPE_POOL_MAX_GIB"); return (size_t)(e ? atoll(e) : 24) << 30; }();
};
static DevPool &pool() {
    static DevPool *p = new DevPool();  // (never destroyed: device memory is released by the runtime at process exit)
    return *p;
}
static size_t pool_round(size_t bytes) { return bytes >= (1u << 20) ? (bytes + ((2u << 20) - 1)) & ~(size_t)((2u << 20) - 1) : (bytes + 255) & ~(size_t)255; }
static void pool_trim(int dev, size_t need_free) {  // give cached blocks back to the runtime (largest first)
    DevPool &P = pool();
    size_t freed = 0;
    while (freed < need_free && !P.free_blocks.empty()) {
        auto it = std::prev(P.free_blocks.end());
        (void)dev;
        freed += it->first.second;
        P.held -= it->first.second;
        (void)hipFree(it->second);
        P.free_blocks.erase(it);
    }
}
static hipError_t pool_alloc(int dev, void **out, size_t bytes) {
    DevPool &P = pool();
    const size_t rb = pool_round(bytes);
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.free_blocks.find({dev, rb});
    if (it != P.free_blocks.end()) {
        *out = it->second;
        P.held -= rb;
        P.free_blocks.erase(it);
        P.live[*out] = {dev, rb};
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, rb);
    if (e != hipSuccess) {  // out of memory with blocks in the cache: release them and try once more
        (void)hipGetLastError();
        pool_trim(dev, ~(size_t)0);
        e = hipMalloc(out, rb);
    }
    if (e == hipSuccess) P.live[*out] = {dev, rb};
    return e;
}
static void pool_free(void *p) {
    DevPool &P = pool();
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.live.find(p);
    if (it == P.live.end()) {  // not ours (allocated with hipMalloc directly)
        (void)hipFree(p);
        return;
    }
    const auto key = it->second;
    P.live.erase(it);
    if (key.second > P.cap) {
        (void)hipFree(p);
        return;
    }
    if (P.held + key.second > P.cap) pool_trim(key.first, P.held + key.second - P.cap);
    P.free_blocks.insert({key, p});
    P.held += key.second;
}

// YABPE_TRACE_ALLOC=1: every device allocation of the library goes to stderr (address range, element size) -- the map
// that tells which buffer a "Memory access fault ... on address X" belongs to or lies next to.
template <class T>
int dmalloc(yabpe_ctx *c, T **p, uint64_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    HIPCHK(c, pool_alloc(c ? c->device : 0, (void **)p, n * sizeof(T)));
    if (trace_alloc_on())
        fprintf(stderr, "[yabpe alloc r%d] %p .. %p  %llu x %zu B\n", c ? c->rank : -1, (void *)*p, (void *)((char *)*p + n * sizeof(T)),
                (unsigned long long)n, sizeof(T));
    return 0;
}
#define TRY(x)               \
    do {                     \
        int r__ = (x);       \
        if (r__ != 0) return r__; \
    } while (0)

void dfree(void *p) {
    if (p && trace_alloc_on()) fprintf(stderr, "[yabpe free] %p\n", p);
    if (p) pool_free(p);
}

int64_t optv(yabpe_ctx *c, const char *k, int64_t dflt) {
    auto it = c->opt.find(k);
    return it == c->opt.end() ? dflt : it->second;
}

bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

inline uint32_t cdiv64(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

int fill_u16(yabpe_ctx *c, uint16_t *p, uint64_t n, uint16_t v) {
    if (!n) return 0;
    uint64_t th = (n + 7) / 8;
    hipLaunchKernelGGL(k_fill_u16, dim3(cdiv64(th, 256)), dim3(256), 0, c->stream, p, (unsigned long long)n, v);
    HIPCHK(c, hipGetLastError());
    return 0;
}
int fill_u32(yabpe_ctx *c, uint32_t *p, uint64_t n, uint32_t v) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_fill_u32, dim3(cdiv64(n, 256)), dim3(256), 0, c->stream, p, (unsigned long long)n, v);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int state_pull(yabpe_ctx *c) {
    HIPCHK(c, hipMemcpyAsync(c->st_host, c->st, sizeof(DevState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
int state_push(yabpe_ctx *c) {
    HIPCHK(c, hipMemcpyAsync(c->st, c->st_host, sizeof(DevState), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---------------------------------------------------------------- pair table
void table_free(PairTable &t) {
    dfree(t.keys);
    dfree(t.cnt);
    t.keys = nullptr;
    t.cnt = nullptr;
    t.cand_cs = nullptr;
    t.cand_list = nullptr;
    t.cand_T = 0;
    t.sink_rec = nullptr;
    t.sink_hdr = nullptr;
    t.sink_cap = 0;
}

int table_alloc(yabpe_ctx *c, PairTable &t, uint64_t cap, unsigned long long *entries_ctr) {
    TRY(dmalloc(c, &t.keys, cap));
    TRY(dmalloc(c, &t.cnt, cap));
    t.cap = (uint32_t)cap;
    // (multi-GPU: a replica must never give up on a probe chain on its own -- its layout differs from its peers' -- so the
    //  chain may run over the whole table; the deterministic 80 % rule of the selection stops all ranks long before that)
    t.max_probe = c->multi ? (uint32_t)cap : (uint32_t)std::min<uint64_t>(cap, 2048);
    t.sink_rec = nullptr;
    t.sink_hdr = nullptr;
    t.sink_cap = 0;
    t.entries = entries_ctr;
    // (the candidate list is attached to the main table only, once it is built: cand_attach / cand_rebuild)
    t.cand_cs = nullptr;
    t.cand_list = nullptr;
    t.cand_T = 0;
    HIPCHK(c, hipMemsetAsync(t.keys, 0xFF, cap * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipMemsetAsync(t.cnt, 0, cap * sizeof(unsigned long long), c->stream));
    return 0;
}

// bitmap of the candidate argmax for the main table (cleared; the candidate list is rebuilt by the caller's next check)
int cand_attach(yabpe_ctx *c) {
    c->use_cand = false;
    PairTable &t = c->table;
    t.cand_cs = nullptr;
    t.cand_list = nullptr;
    if (!optv(c, "cand_argmax", 1)) return 0;
    if (!c->sel_ticket) {
        TRY(dmalloc(c, &c->sel_ticket, TICKET_WORDS));
        HIPCHK(c, hipMemsetAsync(c->sel_ticket, 0, TICKET_WORDS * 4, c->stream));
    }
    if (!c->cand_state) {
        TRY(dmalloc(c, &c->cand_state, 1));
        TRY(dmalloc(c, &c->cand, CAND_CAP));
    }
    return 0;
}

// list = every slot with count >= T, T = best_count x (1 - margin); enables the candidate argmax from here on.  While the
// list is attached to c->table, every kernel that raises a count appends to it (cand_note), so it stays exact for as long
// as the maximum stays >= T.  The margin adapts: the fused selection is ONE workgroup reading the whole list.
int cand_rebuild(yabpe_ctx *c, unsigned long long best_count) {
    c->use_cand = false;
    c->table.cand_cs = nullptr;
    c->table.cand_list = nullptr;
    if (!optv(c, "cand_argmax", 1) || !c->cand_state || best_count < (unsigned long long)optv(c, "cand_min_count", 16)) return 0;
    const uint32_t target = (uint32_t)std::max<int64_t>(64, optv(c, "cand_target", 768));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, c->table_cap / (BLOCK * 8)));
    CandState h{};
    for (int attempt = 0; attempt < 12; ++attempt) {
        unsigned long long margin = (unsigned long long)((double)best_count * c->cand_margin);
        if (margin < 1) margin = 1;
        h = CandState{best_count - std::min(margin, best_count - 1), 0u, 0u, 0u, 0u};
        HIPCHK(c, hipMemcpyAsync(c->cand_state, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
        PairTable t = c->table;
        t.cand_list = c->cand;
        t.cand_cs = c->cand_state;
        t.cand_T = h.T;
        {
            CandParams P{t, c->tt.rec, c->partials, c->st, c->cand_state, nullptr, SelectParams{}};
            hipLaunchKernelGGL(k_cand_rebuild, dim3(grid), dim3(BLOCK), 0, c->stream, P);
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&h, c->cand_state, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->cand_rebuilds++;
        if ((h.overflow || h.n > target) && margin > 1) { // too many pairs that close to the top: halve the margin, again
            c->cand_margin *= 0.5;
            continue;
        }
        break;
    }
    // room to spare: rebuild less often next time -- and a batch can reach further down (early in a job the best counts lie
    // far apart: a selection can only batch pairs that are on the list)
    if (h.n < target / 8) c->cand_margin = std::min((double)optv(c, "cand_margin_max_pct", 60) / 100.0, c->cand_margin * 2.0);
    c->use_cand = !h.overflow && h.n < CAND_CAP / 2;
    c->cand_n_at_build = h.n;
    if (c->use_cand) {
        c->table.cand_cs = c->cand_state;
        c->table.cand_list = c->cand;
        c->table.cand_T = h.T;
    }
    return 0;
}

uint32_t count_grid(yabpe_ctx *c) {
    uint32_t want = (c->n_tiles + WPB - 1) / WPB;
    uint32_t cap = (uint32_t)optv(c, "apply_blocks", (int64_t)c->n_cu * c->apply_occ);
    return std::max(1u, std::min(std::min(want, cap), MAX_APPLY_BLOCKS));
}

int fold_stats(yabpe_ctx *c) {
    FoldParams F{c->st, c->blk_stats, std::max(c->blk_used, 1u)};
    hipLaunchKernelGGL(k_fold_stats, dim3(1), dim3(BLOCK), 0, c->stream, F);
    HIPCHK(c, hipGetLastError());
    c->blk_used = 0;
    return 0;
}

// THE collective: every rank contributes `nbytes` from device memory, every rank receives all contributions in
// rank order.  RCCL all-gather on the compute stream (asynchronous), or the caller's transport (synchronous).
int comm_allgather(yabpe_ctx *c, const void *send, void *recv, uint64_t nbytes) {
    if (c->comm) {
        NCCLCHK(c, rccl()->AllGather(send, recv, nbytes, ncclUint8, c->comm, c->stream));
        return 0;
    }
    if (!c->ag_fn) return fail(c, YABPE_E_COMM, "no transport attached");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc = c->ag_fn(c->ag_user, send, recv, nbytes);
    if (rc != 0) return fail(c, YABPE_E_COMM, "custom all-gather transport failed (%d)", rc);
    return 0;
}

int comm_max(yabpe_ctx *c, unsigned long long v, unsigned long long *out);

// All ranks: dump the local table `lt` as records, all-gather them (padded to the largest rank), add every rank's
// records into `t`.  Used for counts that cannot use the dense byte histogram (weighted words, recounts).
int exchange_table(yabpe_ctx *c, PairTable lt, uint64_t lcap, PairTable t) {
    unsigned long long *d_cnt = &c->scratch64[8];
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, 8, c->stream));
    // first pass only counts (cap 0), so that every rank learns the padded size
    const uint32_t grid = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, lcap / BLOCK));
    DumpParams D0{lt, nullptr, d_cnt, 0};
    hipLaunchKernelGGL(k_table_dump, dim3(grid), dim3(BLOCK), 0, c->stream, D0);
    HIPCHK(c, hipGetLastError());
    unsigned long long mine = 0, maxn = 0;
    HIPCHK(c, hipMemcpyAsync(&mine, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    TRY(comm_max(c, mine, &maxn));
    if (maxn == 0) return 0;
    uint8_t *send = nullptr, *recv = nullptr;
    const uint64_t stride = 16 + maxn * sizeof(DeltaRec);
    TRY(dmalloc(c, &send, stride));
    TRY(dmalloc(c, &recv, stride * c->n_ranks));
    HIPCHK(c, hipMemsetAsync(send, 0, 16, c->stream));
    DumpParams D1{lt, reinterpret_cast<DeltaRec *>(send + 16), reinterpret_cast<unsigned long long *>(send), maxn};
    hipLaunchKernelGGL(k_table_dump, dim3(grid), dim3(BLOCK), 0, c->stream, D1);
    HIPCHK(c, hipGetLastError());
    TRY(comm_allgather(c, send, recv, stride));
    std::vector<unsigned long long> counts(c->n_ranks);
    for (int r = 0; r < c->n_ranks; ++r)
        HIPCHK(c, hipMemcpyAsync(&counts[r], recv + (uint64_t)r * stride, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int r = 0; r < c->n_ranks; ++r) {
        if (!counts[r]) continue;
        RecApplyParams A{reinterpret_cast<const DeltaRec *>(recv + (uint64_t)r * stride + 16), counts[r], t, c->st};
        hipLaunchKernelGGL(k_records_apply, dim3(cdiv64(counts[r], BLOCK)), dim3(BLOCK), 0, c->stream, A);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dfree(send);
    dfree(recv);
    return 0;
}

// max over ranks of a small integer (host-visible); identity on one GPU
int comm_max(yabpe_ctx *c, unsigned long long v, unsigned long long *out) {
    *out = v;
    if (!c->multi) return 0;
    HIPCHK(c, hipMemcpyAsync(&c->scratch64[7], &v, 8, hipMemcpyHostToDevice, c->stream));
    TRY(comm_allgather(c, &c->scratch64[7], c->xsmall, 8));
    std::vector<unsigned long long> all(c->n_ranks);
    HIPCHK(c, hipMemcpyAsync(all.data(), c->xsmall, 8 * (size_t)c->n_ranks, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto x : all) *out = std::max(*out, x);
    return 0;
}

// max over ranks of three small integers in ONE exchange (host-visible); identity on one GPU
int comm_max3(yabpe_ctx *c, unsigned long long v[3]) {
    if (!c->multi) return 0;
    HIPCHK(c, hipMemcpyAsync(&c->scratch64[10], v, 24, hipMemcpyHostToDevice, c->stream));
    TRY(comm_allgather(c, &c->scratch64[10], c->xsmall, 24));
    std::vector<unsigned long long> all(3 * (size_t)c->n_ranks);
    HIPCHK(c, hipMemcpyAsync(all.data(), c->xsmall, 24 * (size_t)c->n_ranks, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int r = 0; r < c->n_ranks; ++r)
        for (int k = 0; k < 3; ++k) v[k] = std::max(v[k], all[3 * (size_t)r + k]);
    return 0;
}

// pair counts of THIS rank's shard into `t` (zeroed by the caller), generic tokens
int launch_count_local(yabpe_ctx *c, PairTable t) {
    if (c->n_tiles) {
        CountParams P{c->tiles, c->tile_len, c->tile_wbase, c->wfreq, c->n_tiles, t, c->st};
        if (c->weighted)
            hipLaunchKernelGGL(k_count<true>, dim3(count_grid(c)), dim3(BLOCK), 0, c->stream, P);
        else
            hipLaunchKernelGGL(k_count<false>, dim3(count_grid(c)), dim3(BLOCK), 0, c->stream, P);
        HIPCHK(c, hipGetLastError());
    }
    if (c->n_long) {
        LongParams L{c->long_tok, c->long_off, c->long_len, c->long_freq, c->n_long, t, c->st, nullptr, 0u};
        hipLaunchKernelGGL(k_count_long, dim3(c->n_long), dim3(BLOCK), 0, c->stream, L);
        HIPCHK(c, hipGetLastError());
    }
    return 0;
}

// Full recount of the resident stream -- of ALL ranks' shards when a communicator is attached -- into `t`
// (zeroed by the caller).  all_bytes: every token is still a byte (initial count of the flat layout).
int launch_count(yabpe_ctx *c, PairTable t, bool all_bytes = false) {
    if (all_bytes && !c->weighted && optv(c, "dense_count", 1)) {
        // trainer.py:227-235 over 256 x 256 keys: dense LDS histograms, summed across ranks by one all-reduce
        unsigned long long *dense = nullptr;
        TRY(dmalloc(c, &dense, 65536));
        HIPCHK(c, hipMemsetAsync(dense, 0, 65536 * 8, c->stream));
        if (c->n_tiles) {
            const uint32_t bpp = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(c->n_cu / 2, (c->n_tiles + CB_WAVES - 1) / CB_WAVES));
            CountBytesParams P{c->tiles, c->tile_len, c->n_tiles, dense, bpp};
            hipLaunchKernelGGL(k_count_bytes, dim3(4 * bpp), dim3(CB_BLOCK), 0, c->stream, P);
        }
        if (c->multi) {  // sum the per-rank histograms: gather all, add rows
            unsigned long long *all = nullptr;
            TRY(dmalloc(c, &all, 65536ull * c->n_ranks));
            TRY(comm_allgather(c, dense, all, 65536 * 8));
            hipLaunchKernelGGL(k_sum_rows, dim3(65536 / BLOCK), dim3(BLOCK), 0, c->stream, all, dense, 65536u, (uint32_t)c->n_ranks);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(c->stream));
            dfree(all);
        }
        DenseToTableParams D{dense, t, c->st};
        hipLaunchKernelGGL(k_dense_to_table, dim3(65536 / BLOCK), dim3(BLOCK), 0, c->stream, D);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dfree(dense);
        if (c->n_long) {  // long words of this rank (bytes too) -- their pairs go through the generic exchange
            if (!c->multi) {
                LongParams L{c->long_tok, c->long_off, c->long_len, c->long_freq, c->n_long, t, c->st, nullptr, 0u};
                hipLaunchKernelGGL(k_count_long, dim3(c->n_long), dim3(BLOCK), 0, c->stream, L);
                HIPCHK(c, hipGetLastError());
            }
        }
        if (!c->multi) return 0;
        // multi-GPU: any rank may hold long words; exchange their pairs generically (usually nothing)
        unsigned long long any_long = 0;
        TRY(comm_max(c, c->n_long, &any_long));
        if (!any_long) return 0;
        PairTable lt{};
        HIPCHK(c, hipMemsetAsync(&c->scratch64[9], 0, 8, c->stream));
        TRY(table_alloc(c, lt, 1ull << 18, &c->scratch64[9]));
        if (c->n_long) {
            LongParams L{c->long_tok, c->long_off, c->long_len, c->long_freq, c->n_long, lt, c->st, nullptr, 0u};
            hipLaunchKernelGGL(k_count_long, dim3(c->n_long), dim3(BLOCK), 0, c->stream, L);
            HIPCHK(c, hipGetLastError());
        }
        int rc = exchange_table(c, lt, 1ull << 18, t);
        table_free(lt);
        return rc;
    }
    if (!c->multi) return launch_count_local(c, t);
    // generic multi-GPU count: local table -> records -> all-gather -> sum into t
    PairTable lt{};
    const uint64_t lcap = (uint64_t)t.cap;
    HIPCHK(c, hipMemsetAsync(&c->scratch64[9], 0, 8, c->stream));
    TRY(table_alloc(c, lt, lcap, &c->scratch64[9]));
    TRY(launch_count_local(c, lt));
    int rc = exchange_table(c, lt, lcap, t);
    table_free(lt);
    return rc;
}

// (Re)build the pair table from the token stream with at least `min_cap` slots; grows until the load is <= 1/2.
int table_rebuild(yabpe_ctx *c, uint64_t min_cap, bool all_bytes = false) {
    uint64_t cap = std::max<uint64_t>(min_cap, 1ull << optv(c, "table_min_log2", 16));
    bool shrunk = false;
    for (int attempt = 0; attempt < 16; ++attempt) {
        table_free(c->table);
        HIPCHK(c, hipMemsetAsync(&c->st->table_entries, 0, sizeof(unsigned long long), c->stream));
        HIPCHK(c, hipMemsetAsync(&c->st->halt_req, 0, sizeof(uint32_t), c->stream));
        TRY(table_alloc(c, c->table, cap, &c->st->table_entries));
        c->table_cap = cap;
        TRY(launch_count(c, c->table, all_bytes));
        TRY(state_pull(c));
        unsigned long long bad = (c->st_host->halt_req != 0 || c->st_host->table_entries * 2 > cap) ? 1 : 0, any_bad = 0;
        TRY(comm_max(c, bad, &any_bad));  // replicas differ in layout: all ranks retry together
        if (!any_bad) {
            if (!shrunk && c->st_host->table_entries * 32 < cap && cap > (1ull << 16)) {
                // far too roomy (scans of the table -- candidate rebuilds, the fallback argmax -- read every slot): count
                // once more at ~8x the entries (the entry count is the same on every rank, so all ranks take this branch together)
                shrunk = true;
                cap = std::max<uint64_t>(c->st_host->table_entries * 8, 1ull << 16);
                continue;
            }
            c->stats.table_rebuilds++;
            TRY(cand_attach(c));
            return 0;
        }
        cap *= 4;
        if (cap > (1ull << 31)) break;
    }
    return fail(c, YABPE_E_CAPACITY, "pair table does not fit (more than 2^29 distinct pairs)");
}

// Grow the pair table by re-inserting its live entries (entries whose count fell to 0 are dropped).
int table_grow(yabpe_ctx *c, uint64_t new_cap) {
    for (int attempt = 0; attempt < 8; ++attempt, new_cap = new_cap * 3 / 2) {
        PairTable nt{};
        HIPCHK(c, hipMemsetAsync(&c->st->table_entries, 0, sizeof(unsigned long long), c->stream));
        TRY(table_alloc(c, nt, new_cap, &c->st->table_entries));
        RehashParams R{c->table, nt, c->st};
        const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, std::max<uint64_t>(1, c->table_cap / BLOCK));
        hipLaunchKernelGGL(k_rehash, dim3(grid), dim3(BLOCK), 0, c->stream, R);
        HIPCHK(c, hipGetLastError());
        TRY(state_pull(c));
        if (c->st_host->halt_req == 0 && c->st_host->table_entries * 10 <= new_cap * 6) {
            table_free(c->table);
            c->table = nt;
            c->table_cap = new_cap;
            c->stats.table_rebuilds++;
            TRY(cand_attach(c));
            return 0;
        }
        table_free(nt);
        c->st_host->halt_req = 0;
        TRY(state_push(c));
    }
    return fail(c, YABPE_E_CAPACITY, "pair table does not fit");
}

#define YB_TRACE_C(c, ...) do { if (optv(c, "trace_host", 0)) { fprintf(stderr, "[yabpe r%d] ", (c)->rank); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } } while (0)
// The peer-to-peer receive areas are exported and mapped ONCE per job, at their final size, and never freed while the job
// lives.  Taking them down and up again when the exchange buffers grow (unmap, free, allocate, export, map) does not survive
// on this runtime: about one time in five hipIpcOpenMemHandle of the re-exported area failed on one rank and never returned
// on the other (two processes, one MI355X, the 8 GiB job's second growth) -- whether the ranks map one at a time or together,
// with or without a barrier between unmapping and freeing.  A first mapping has never failed.
void p2p_close(yabpe_ctx *c) {  // (yabpe_destroy: no collective here -- a peer may be gone)
    for (size_t r = 0; r < c->p2p_peer.size(); ++r)
        if (c->p2p_peer[r] && (int)r != c->rank) (void)hipIpcCloseMemHandle(c->p2p_peer[r]);
    c->p2p_peer.clear();
    // the own area: a peer may still have it mapped -- left to the process's exit (the driver drops it with its last mapping)
    if (c->p2p_area && c->n_ranks <= 1) (void)hipFree(c->p2p_area);
    c->p2p_area = nullptr;
}

// Build the peer-to-peer receive areas: `p2p_cap` records per sender and half (option "p2p_cap_records", 4 Mi = 64 MiB per
// slot: 1 GiB of a rank's 288 GB at 8 ranks), exchange the IPC handles through the attached transport (one all-gather of
// 64 bytes per rank), map the peers' areas.  Collective: every rank calls it at the same point and runs the same sequence of
// collectives whatever fails locally; if ANY rank could not export or map a buffer, all ranks drop the peer-to-peer path
// together and keep exchanging through the transport (c->p2p = false, no error).
int p2p_setup(yabpe_ctx *c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->p2p_area) return 0;  // (once)
    unsigned long long failed = 0;
    std::string why;
    if (c->n_ranks > XCHG_MAX_RANKS) { failed = 1; why = "too many ranks"; }
    c->p2p_cap = (uint32_t)std::max<int64_t>((int64_t)c->xcap, std::min<int64_t>(optv(c, "p2p_cap_records", 4ll << 20), 1ll << 28));
    c->p2p_stride = 16 + (uint64_t)c->p2p_cap * sizeof(DeltaRec);
    c->p2p_half = c->p2p_stride * (uint64_t)c->n_ranks;
    c->p2p_flags_off = (2 * c->p2p_half + 255) & ~255ull;
    const uint64_t bytes = c->p2p_flags_off + 2ull * c->n_ranks * 8 + 256;
    hipIpcMemHandle_t mine;
    memset(&mine, 0, sizeof mine);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (!failed) {
        YB_TRACE_C(c, "p2p setup: allocate %llu bytes", (unsigned long long)bytes);
        hipError_t e = hipMalloc((void **)&c->p2p_area, bytes);
        // (only the flag words have to start at zero: a slot is read up to the count its header announces)
        if (e == hipSuccess) e = hipMemset(c->p2p_area + c->p2p_flags_off, 0, bytes - c->p2p_flags_off);
        if (e == hipSuccess) e = hipIpcGetMemHandle(&mine, c->p2p_area);
        if (e != hipSuccess) { failed = 1; why = std::string("export: ") + hipGetErrorString(e); (void)hipGetLastError(); }
    }
    uint8_t *d_h = nullptr, *d_all = nullptr;
    TRY(dmalloc(c, &d_h, 64));
    TRY(dmalloc(c, &d_all, 64ull * c->n_ranks));
    HIPCHK(c, hipMemcpy(d_h, &mine, 64, hipMemcpyHostToDevice));
    TRY(comm_allgather(c, d_h, d_all, 64));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<hipIpcMemHandle_t> all(c->n_ranks);
    HIPCHK(c, hipMemcpy(all.data(), d_all, 64ull * c->n_ranks, hipMemcpyDeviceToHost));
    dfree(d_h);
    dfree(d_all);
    c->p2p_peer.assign(c->n_ranks, nullptr);
    for (int r = 0; r < c->n_ranks && !failed; ++r) {
        if (r == c->rank) { c->p2p_peer[r] = c->p2p_area; continue; }
        void *q = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&q, all[r], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { failed = 1; why = std::string("map: ") + hipGetErrorString(e); (void)hipGetLastError(); break; }
        c->p2p_peer[r] = (uint8_t *)q;
    }
    c->p2p_seq = 0;
    YB_TRACE_C(c, "p2p setup: mapped (failed %llu)", failed);
    // nobody pushes before every rank has mapped everything -- and everybody learns whether somebody could not
    unsigned long long any_failed = 0;
    TRY(comm_max(c, failed, &any_failed));
    if (any_failed) {
        c->p2p = false;  // (what is mapped stays mapped until yabpe_destroy: nothing is taken down while peers may be in the runtime)
        if (failed && optv(c, "trace_exchange", 0)) fprintf(stderr, "[yabpe r%d] peer-to-peer exchange not available here (%s): exchanging through the transport\n", c->rank, why.c_str());
    }
    return 0;
}

// (re)allocate the per-iteration exchange buffers for `cap` records per rank
int comm_buffers(yabpe_ctx *c, uint32_t cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dfree(c->xsend);
    dfree(c->xrecv);
    c->xsend = c->xrecv = nullptr;
    c->xcap = cap;
    c->xeff = cap;
    c->xstride = 16 + (uint64_t)cap * sizeof(DeltaRec);
    TRY(dmalloc(c, &c->xsend, c->xstride));
    TRY(dmalloc(c, &c->xrecv, c->xstride * c->n_ranks));
    HIPCHK(c, hipMemsetAsync(c->xsend, 0, c->xstride, c->stream));
    HIPCHK(c, hipMemsetAsync(c->xrecv, 0, c->xstride * c->n_ranks, c->stream));
    // (the peer-to-peer areas keep their size: a buffer beyond it -- the same on every rank -- goes through the transport from here on)
    if (c->p2p && c->p2p_area && cap > c->p2p_cap) {
        c->p2p = false;
        if (optv(c, "trace_exchange", 0)) fprintf(stderr, "[yabpe r%d] exchange buffers of %u records exceed the peer-to-peer areas (%u): exchanging through the transport\n", c->rank, cap, c->p2p_cap);
    }
    return 0;
}

int refresh_live_slots(yabpe_ctx *c) {
    HIPCHK(c, hipMemsetAsync(&c->scratch64[0], 0, 8, c->stream));
    if (c->n_tiles) {
        hipLaunchKernelGGL(k_sum_u32, dim3(std::min<uint32_t>(1024, cdiv64(c->n_tiles, BLOCK))), dim3(BLOCK), 0, c->stream,
                           c->tile_len, (unsigned long long)c->n_tiles, &c->scratch64[0]);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(&c->st->live_slots, &c->scratch64[0], 8, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

void free_corpus(yabpe_ctx *c) {
    dfree(c->tiles); dfree(c->tiles_alt); dfree(c->tile_len); dfree(c->tile_len_alt);
    dfree(c->tile_wbase); dfree(c->tile_wbase_alt); dfree(c->wfreq); dfree(c->sig);
    c->sig = nullptr; c->sig_stride = 0;
    dfree(c->long_tok); dfree(c->long_off); dfree(c->long_len); dfree(c->long_freq); dfree(c->long_sig);
    c->long_sig = nullptr; c->long_sig_stride = 0;
    c->tiles = c->tiles_alt = nullptr;
    c->tile_len = c->tile_len_alt = c->tile_wbase = c->tile_wbase_alt = c->wfreq = nullptr;
    c->long_tok = nullptr; c->long_off = nullptr; c->long_len = c->long_freq = nullptr;
    c->n_long = 0; c->n_tiles = 0;
    table_free(c->table);
    c->have_words = false;
}

void free_records(yabpe_ctx *c) {
    dfree(c->rec_left); dfree(c->rec_right); dfree(c->rec_merged);
    dfree(c->rec_count); dfree(c->rec_sites); dfree(c->rec_live);
    c->rec_left = c->rec_right = c->rec_merged = nullptr;
    c->rec_count = c->rec_sites = c->rec_live = nullptr;
    c->rec_cap = 0;
}


// (Re)compute every tile's signature from the resident stream (after load and after a retile).
int build_signatures(yabpe_ctx *c) {
    if (!optv(c, "skip_index", 1) || c->n_tiles == 0) return 0;
    if (!c->sig || c->sig_stride < c->n_tiles) {
        dfree(c->sig);
        c->sig = nullptr;
        c->sig_stride = (uint32_t)((c->n_tiles + 31u) & ~31u);  // rows start 128-B aligned
        TRY(dmalloc(c, &c->sig, (uint64_t)SIG_ROWS * c->sig_stride));
    }
    SigParams P{c->tiles, c->tile_len, c->n_tiles, c->sig, c->sig_stride};
    const uint32_t n_groups = (uint32_t)((c->n_tiles + SIG_TILES - 1) / SIG_TILES);
    hipLaunchKernelGGL(k_build_sig, dim3(std::min<uint32_t>(n_groups, 256 * 3)), dim3(BLOCK), 0, c->stream, P);
    c->sig_tokens_at_build = c->st_host->tokens_now;
    c->sig_builds++;
    HIPCHK(c, hipGetLastError());
    return 0;
}

// Repack the live words of the flat layout into fresh, densely filled tiles (drops dead/empty words and PAD).
// The apply kernel reads tile prefixes only, so this is what keeps its traffic proportional to live tokens.
int retile_flat(yabpe_ctx *c) {
    if (c->n_tiles == 0) return 0;
    uint32_t *kept = nullptr;
    unsigned long long *base = nullptr;
    TRY(dmalloc(c, &kept, c->n_tiles));
    TRY(dmalloc(c, &base, (uint64_t)c->n_tiles + 1));
    const uint32_t grid = count_grid(c);
    RetileParams P{c->tiles, c->tile_len, c->n_tiles, kept, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_retile<false>, dim3(grid), dim3(BLOCK), 0, c->stream, P);
    HIPCHK(c, hipGetLastError());
    if (exclusive_scan<uint32_t>(c->stream, kept, c->n_tiles, base, (unsigned long long)c->n_tiles + 1) != 0)
        return fail(c, YABPE_E_HIP, "retile scan failed: %s", hipGetErrorString(hipGetLastError()));
    unsigned long long total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, base + c->n_tiles, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const uint32_t new_n = std::max<uint32_t>(1, (uint32_t)((total + SPAN - 1) / SPAN));
    if (!c->tiles_alt || c->tiles_alt_cap < new_n) {
        dfree(c->tiles_alt);
        dfree(c->tile_len_alt);
        c->tiles_alt = nullptr;
        c->tile_len_alt = nullptr;
        TRY(dmalloc(c, &c->tiles_alt, (uint64_t)new_n * CAP));
        TRY(dmalloc(c, &c->tile_len_alt, new_n));
        c->tiles_alt_cap = new_n;
    }
    TRY(fill_u16(c, c->tiles_alt, (uint64_t)new_n * CAP, YB_PAD));
    HIPCHK(c, hipMemsetAsync(c->tile_len_alt, 0, (size_t)new_n * 4, c->stream));
    P.base = base;
    P.new_tiles = c->tiles_alt;
    P.new_len = c->tile_len_alt;
    hipLaunchKernelGGL(k_retile<true>, dim3(grid), dim3(BLOCK), 0, c->stream, P);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::swap(c->tiles, c->tiles_alt);
    std::swap(c->tile_len, c->tile_len_alt);
    std::swap(c->tiles_cap, c->tiles_alt_cap);
    c->n_tiles = new_n;
    dfree(kept);
    dfree(base);
    TRY(refresh_live_slots(c));
    if (c->sig_valid) TRY(build_signatures(c));
    c->stats.retiles++;
    return 0;
}

}  // namespace

// =================================================================================================== C ABI
extern "C" {

int yabpe_abi_version(void) { return YABPE_ABI_VERSION; }

int yabpe_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char *yabpe_last_error(const yabpe_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int yabpe_create(yabpe_ctx **out, int device_id) {
    if (!out) return fail(nullptr, YABPE_E_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, YABPE_E_NODEVICE, "no HIP device available (%s); the BPE hot path has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    if (device_id < 0 || device_id >= n) return fail(nullptr, YABPE_E_INVALID, "device_id %d out of range [0,%d)", device_id, n);
    yabpe_ctx *c = new yabpe_ctx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) {
        delete c;
        return fail(nullptr, YABPE_E_HIP, "hipSetDevice(%d) failed", device_id);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        // The streaming-phase grid must be resident all at once: its workgroups stride over the tiles and all finish together, so
        // a workgroup that has to wait for a free slot runs alone afterwards (5 per CU asked, 4 resident: +25 % on every launch).
        int occ = 0, lo = 1 << 30;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_apply<false, false>, BLOCK, 0) == hipSuccess && occ > 0) lo = std::min(lo, occ);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_apply<false, true>, BLOCK, 0) == hipSuccess && occ > 0) lo = std::min(lo, occ);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_apply<true, false>, BLOCK, 0) == hipSuccess && occ > 0) lo = std::min(lo, occ);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_count<false>, BLOCK, 0) == hipSuccess && occ > 0) lo = std::min(lo, occ);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_count<true>, BLOCK, 0) == hipSuccess && occ > 0) lo = std::min(lo, occ);
        (void)hipGetLastError();
        c->apply_occ = lo == (1 << 30) ? 4 : lo;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&c->st, sizeof(DevState)) != hipSuccess ||
        hipHostMalloc((void **)&c->st_host, sizeof(DevState), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&c->scratch64, 16 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void **)&c->blk_stats, 2 * MAX_APPLY_BLOCKS * sizeof(unsigned long long)) != hipSuccess) {
        int code = fail(nullptr, YABPE_E_HIP, "context allocation failed: %s", hipGetErrorString(hipGetLastError()));
        yabpe_destroy(c);
        return code;
    }
    (void)hipMemset(c->st, 0, sizeof(DevState));
    (void)hipMemset(c->blk_stats, 0, 2 * MAX_APPLY_BLOCKS * sizeof(unsigned long long));
    memset(c->st_host, 0, sizeof(DevState));
    *out = c;
    return YABPE_OK;
}

void yabpe_destroy(yabpe_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    yabpe_synth_free(c);
    yabpe_pretokenize_free(c);
    dfree(c->pt_cls);
    free_corpus(c);
    free_records(c);
    dfree(c->tt.pool); dfree(c->tt.off); dfree(c->tt.len); dfree(c->tt.rec); dfree(c->tt.vset);
    dfree(c->partials);
    dfree(c->st);
    dfree(c->scratch64);
    dfree(c->blk_stats);
    dfree(c->blk_read);
    dfree(c->cand_state);
    dfree(c->sel_ticket);
    dfree(c->cand);
    dfree(c->xsend);
    dfree(c->xrecv);
    dfree(c->xsmall);
    p2p_close(c);
    if (c->p2p_e0) (void)hipEventDestroy(c->p2p_e0);
    if (c->p2p_e1) (void)hipEventDestroy(c->p2p_e1);
    if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    if (c->st_host) (void)hipHostFree(c->st_host);
    for (auto &e : c->events) {
        (void)hipEventDestroy(e.e0);
        (void)hipEventDestroy(e.e1);
        (void)hipEventDestroy(e.e2);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int yabpe_set_option(yabpe_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return YABPE_E_INVALID;
    c->opt[name] = value;
    return YABPE_OK;
}

// ---------------------------------------------------------------------------------------------- vocab
int yabpe_set_vocab(yabpe_ctx *c, const uint8_t *tok_bytes, const uint32_t *tok_off, uint32_t n_tokens) {
    if (!c) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!tok_bytes || !tok_off || n_tokens < 256) return fail(c, YABPE_E_INVALID, "vocab needs the 256 byte tokens first");
    for (uint32_t b = 0; b < 256; ++b)
        if (tok_off[b + 1] - tok_off[b] != 1 || tok_bytes[tok_off[b]] != b)
            return fail(c, YABPE_E_INVALID, "token %u must be the single byte %u (trainer.py:123-125)", b, b);
    if (n_tokens >= YB_MAX_TOKENS) return fail(c, YABPE_E_CAPACITY, "base vocabulary too large for u16 token ids");
    const uint32_t pool_cap = (uint32_t)optv(c, "pool_bytes", 64 << 20) & ~3u;
    const uint32_t vset_cap = 1u << 18;
    // every token starts at a multiple of 4 in the pool (new tokens' bytes are written four at a time)
    std::vector<uint32_t> off(n_tokens), len(n_tokens), order(n_tokens);
    std::vector<uint8_t> pool;
    for (uint32_t i = 0; i < n_tokens; ++i) {
        len[i] = tok_off[i + 1] - tok_off[i];
        off[i] = (uint32_t)pool.size();
        pool.insert(pool.end(), tok_bytes + tok_off[i], tok_bytes + tok_off[i + 1]);
        pool.resize((pool.size() + 3) & ~(size_t)3, 0);
        order[i] = i;
    }
    if (pool.size() > pool_cap / 2) return fail(c, YABPE_E_CAPACITY, "base vocabulary bytes exceed the token pool");
    dfree(c->tt.pool); dfree(c->tt.off); dfree(c->tt.len); dfree(c->tt.rec); dfree(c->tt.vset);
    TRY(dmalloc(c, &c->tt.pool, pool_cap));
    TRY(dmalloc(c, &c->tt.off, YB_MAX_TOKENS));
    TRY(dmalloc(c, &c->tt.len, YB_MAX_TOKENS));
    TRY(dmalloc(c, &c->tt.rec, YB_MAX_TOKENS));
    TRY(dmalloc(c, &c->tt.vset, vset_cap));
    c->tt.pool_cap = pool_cap;
    c->tt.vset_mask = vset_cap - 1;
    std::vector<TokRec> rec(n_tokens);
    std::vector<unsigned long long> vset(vset_cap, VSET_EMPTY);
    auto cmp = [&](uint32_t x, uint32_t y) {
        uint32_t n = std::min(len[x], len[y]);
        int d = memcmp(pool.data() + off[x], pool.data() + off[y], n);
        if (d) return d < 0;
        return len[x] < len[y];
    };
    std::sort(order.begin(), order.end(), cmp);
    for (uint32_t r = 0; r < n_tokens; ++r) {
        if (r && !cmp(order[r - 1], order[r])) return fail(c, YABPE_E_INVALID, "duplicate token bytes in the base vocabulary (trainer.py:130)");
        rec[order[r]].rank = r;
    }
    for (uint32_t i = 0; i < n_tokens; ++i) {
        rec[i].len = len[i];
        rec[i].hash = yb_hash_bytes(pool.data() + off[i], len[i]);
        rec[i].pre8 = yb_pre8(pool.data() + off[i], len[i]);
        rec[i].pad = 0;
        uint32_t s = yb_vset_home(rec[i].hash, len[i]) & c->tt.vset_mask;
        while (vset[s] != VSET_EMPTY) s = (s + 1) & c->tt.vset_mask;
        vset[s] = yb_vset_entry(i, rec[i].hash);
    }
    HIPCHK(c, hipMemset(c->tt.pool, 0, pool_cap));
    HIPCHK(c, hipMemcpy(c->tt.pool, pool.data(), pool.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->tt.off, off.data(), n_tokens * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->tt.len, len.data(), n_tokens * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->tt.rec, rec.data(), n_tokens * sizeof(TokRec), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->tt.vset, vset.data(), (size_t)vset_cap * 8, hipMemcpyHostToDevice));
    TRY(state_pull(c));
    DevState *h = c->st_host;
    memset(h, 0, sizeof(DevState));
    h->n_tokens = n_tokens;
    h->pool_used = (uint32_t)pool.size();
    TRY(state_push(c));
    c->base_tokens = n_tokens;
    c->have_vocab = true;
    return YABPE_OK;
}

// ---------------------------------------------------------------------------------------------- corpus
int yabpe_load_words(yabpe_ctx *c, const uint8_t *bytes, const uint64_t *word_off, const uint64_t *word_freq,
                     uint64_t n_words, uint32_t flags) {
    if (!c) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->have_vocab) return fail(c, YABPE_E_INVALID, "call yabpe_set_vocab first");
    if (!word_off) return fail(c, YABPE_E_INVALID, "word_off is NULL");
    if (n_words >= 0xFFFFFFFFull) return fail(c, YABPE_E_CAPACITY, "more than 2^32-2 words per context");
    free_corpus(c);
    hipEvent_t ev0, ev1;
    HIPCHK(c, hipEventCreate(&ev0));
    HIPCHK(c, hipEventCreate(&ev1));

    // ---- make inputs device-resident
    const unsigned long long *d_off = nullptr;
    const uint8_t *d_bytes = nullptr;
    const unsigned long long *d_freq = nullptr;
    void *own_off = nullptr, *own_bytes = nullptr, *own_freq = nullptr;
    uint64_t total_bytes = 0, off_base = 0;
    if (is_device_ptr(word_off)) {
        d_off = (const unsigned long long *)word_off;
        HIPCHK(c, hipMemcpy(&total_bytes, word_off + n_words, 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(&off_base, word_off, 8, hipMemcpyDeviceToHost));
        total_bytes -= off_base;
    } else {
        off_base = word_off[0];
        total_bytes = word_off[n_words] - off_base;
        HIPCHK(c, hipMalloc(&own_off, (n_words + 1) * 8));
        HIPCHK(c, hipMemcpy(own_off, word_off, (n_words + 1) * 8, hipMemcpyHostToDevice));
        d_off = (const unsigned long long *)own_off;
    }
    if (total_bytes && !bytes) return fail(c, YABPE_E_INVALID, "bytes is NULL");
    if (is_device_ptr(bytes) || total_bytes == 0) {
        d_bytes = bytes;
    } else {
        HIPCHK(c, hipMalloc(&own_bytes, total_bytes));
        HIPCHK(c, hipMemcpy(own_bytes, bytes + off_base, total_bytes, hipMemcpyHostToDevice));
        d_bytes = (const uint8_t *)own_bytes - off_base;  // offsets stay absolute
    }
    if (word_freq) {
        if (is_device_ptr(word_freq)) {
            d_freq = (const unsigned long long *)word_freq;
        } else {
            HIPCHK(c, hipMalloc(&own_freq, std::max<uint64_t>(n_words, 1) * 8));
            HIPCHK(c, hipMemcpy(own_freq, word_freq, n_words * 8, hipMemcpyHostToDevice));
            d_freq = (const unsigned long long *)own_freq;
        }
    }
    auto cleanup_inputs = [&]() { dfree(own_off); dfree(own_bytes); dfree(own_freq); own_off = own_bytes = own_freq = nullptr; };
    HIPCHK(c, hipEventRecord(ev0, c->stream));

    c->n_words_input = n_words;
    // ---- optional device-side pooling of equal words (trainer.py:221-225)
    DedupOut dd{};
    if ((flags & YABPE_LOAD_DEDUP) && n_words > 0) {
        int r = dedup_words(c->stream, d_bytes, d_off, d_freq, n_words, total_bytes, &dd);
        if (r != 0) { cleanup_inputs(); return fail(c, YABPE_E_HIP, "device dedup failed (%d): %s", r, hipGetErrorString(hipGetLastError())); }
        cleanup_inputs();
        own_bytes = dd.bytes; own_off = dd.off; own_freq = dd.freq;
        d_bytes = dd.bytes; d_off = dd.off; d_freq = dd.freq;
        n_words = dd.n_unique;
        total_bytes = dd.total_bytes;
        off_base = 0;
    }
    // (a rank that holds no words must still run the layout its peers run: the collectives differ between the layouts)
    c->weighted = d_freq != nullptr || (flags & YABPE_LOAD_DEDUP) != 0;
    c->n_words = n_words;
    c->tokens_initial = total_bytes;

    // ---- tiles
    const uint64_t packed = total_bytes + n_words;
    const uint64_t n_tiles64 = (packed + SPAN - 1) / SPAN;
    if (n_tiles64 >= 0xFFFFFFF0ull) { cleanup_inputs(); return fail(c, YABPE_E_CAPACITY, "corpus too large for one context"); }
    c->n_tiles = (uint32_t)n_tiles64;
    c->tiles_cap = c->n_tiles;
    c->tiles_alt_cap = 0;
    TRY(dmalloc(c, &c->tiles, (uint64_t)c->n_tiles * CAP));
    TRY(dmalloc(c, &c->tile_len, c->n_tiles));
    if (c->weighted) {
        TRY(dmalloc(c, &c->tile_wbase, c->n_tiles));
        TRY(dmalloc(c, &c->wfreq, n_words));
        HIPCHK(c, hipMemsetAsync(&c->scratch64[1], 0, 8, c->stream));
        if (n_words) {
            hipLaunchKernelGGL(k_freq64_to_32, dim3(cdiv64(n_words, 256)), dim3(256), 0, c->stream, d_freq, c->wfreq,
                               (unsigned long long)n_words, (uint32_t *)&c->scratch64[1]);
            HIPCHK(c, hipGetLastError());
        }
    }
    uint32_t long_cap = (uint32_t)optv(c, "long_cap", 1 << 16);
    uint32_t *d_long_word = nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        TRY(fill_u16(c, c->tiles, (uint64_t)c->n_tiles * CAP, YB_PAD));
        HIPCHK(c, hipMemsetAsync(c->tile_len, 0, (size_t)c->n_tiles * 4, c->stream));
        if (c->weighted) TRY(fill_u32(c, c->tile_wbase, c->n_tiles, 0xFFFFFFFFu));
        HIPCHK(c, hipMemsetAsync(&c->scratch64[2], 0, 16, c->stream));  // [2] = long count (u32), [3] = long tokens
        dfree(d_long_word);
        TRY(dmalloc(c, &d_long_word, long_cap));
        if (n_words) {
            LoadParams P{d_bytes, d_off, (unsigned long long)off_base, (unsigned long long)n_words, c->tiles, c->tile_len, c->tile_wbase,
                         (uint32_t *)&c->scratch64[2], &c->scratch64[3], d_long_word, long_cap};
            hipLaunchKernelGGL(k_load_words, dim3(cdiv64(n_words, BLOCK)), dim3(BLOCK), 0, c->stream, P);
            HIPCHK(c, hipGetLastError());
        }
        unsigned long long host2[2];
        HIPCHK(c, hipMemcpyAsync(host2, &c->scratch64[2], 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->n_long = (uint32_t)(host2[0] & 0xFFFFFFFFu);
        c->long_tokens = host2[1];
        if (c->n_long <= long_cap) break;
        long_cap = c->n_long;
        if (attempt == 1) { cleanup_inputs(); return fail(c, YABPE_E_INTERNAL, "long-word list overflow"); }
    }
    if (c->weighted) {
        unsigned long long ov = 0;
        HIPCHK(c, hipMemcpy(&ov, &c->scratch64[1], 8, hipMemcpyDeviceToHost));
        if (ov & 0xFFFFFFFFu) { cleanup_inputs(); return fail(c, YABPE_E_CAPACITY, "a word frequency exceeds 2^32-1"); }
    }
    // ---- long words: own buffer, one workgroup per word
    if (c->n_long) {
        std::vector<uint32_t> lw(c->n_long);
        HIPCHK(c, hipMemcpy(lw.data(), d_long_word, (size_t)c->n_long * 4, hipMemcpyDeviceToHost));
        std::sort(lw.begin(), lw.end());  // deterministic layout
        HIPCHK(c, hipMemcpy(d_long_word, lw.data(), (size_t)c->n_long * 4, hipMemcpyHostToDevice));
        // where each long word goes: prefix sum of their lengths, on the device
        uint32_t *d_ll = nullptr;
        TRY(dmalloc(c, &d_ll, c->n_long));
        TRY(dmalloc(c, &c->long_off, c->n_long + 1));
        HIPCHK(c, hipMemsetAsync(&c->scratch64[2], 0, 8, c->stream));
        hipLaunchKernelGGL(k_long_lengths, dim3(cdiv64(c->n_long, 256)), dim3(256), 0, c->stream, d_off, d_long_word, c->n_long, d_ll, (uint32_t *)&c->scratch64[2]);
        HIPCHK(c, hipGetLastError());
        if (exclusive_scan<uint32_t>(c->stream, d_ll, c->n_long, c->long_off, (unsigned long long)c->n_long + 1) != 0) {
            cleanup_inputs();
            return fail(c, YABPE_E_HIP, "long-word scan failed: %s", hipGetErrorString(hipGetLastError()));
        }
        unsigned long long long_total = 0, too_long = 0;
        // (on the context's stream: it does not synchronise with the null stream, and a one-block scan returns without waiting)
        HIPCHK(c, hipMemcpyAsync(&long_total, c->long_off + c->n_long, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&too_long, &c->scratch64[2], 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dfree(d_ll);
        if (too_long & 0xFFFFFFFFull) { cleanup_inputs(); return fail(c, YABPE_E_CAPACITY, "a single word longer than 2^32-1 bytes"); }
        TRY(dmalloc(c, &c->long_tok, long_total));
        TRY(dmalloc(c, &c->long_len, c->n_long));
        if (c->weighted) TRY(dmalloc(c, &c->long_freq, c->n_long));
        LoadLongParams L{d_bytes, d_off, d_freq, d_long_word, c->long_off, c->long_tok, c->long_len, c->long_freq, c->n_long};
        hipLaunchKernelGGL(k_load_long, dim3(c->n_long), dim3(BLOCK), 0, c->stream, L);
        HIPCHK(c, hipGetLastError());
        if (optv(c, "long_sig", 1)) {  // signatures: a merge whose pair is in no long word then costs 8 bytes per long word
            c->long_sig_stride = (c->n_long + 31u) & ~31u;
            TRY(dmalloc(c, &c->long_sig, (uint64_t)SIG_ROWS * c->long_sig_stride));
            LongParams S{c->long_tok, c->long_off, c->long_len, c->long_freq, c->n_long, PairTable{}, c->st, c->long_sig, c->long_sig_stride};
            hipLaunchKernelGGL(k_build_sig_long, dim3(c->n_long), dim3(BLOCK), 0, c->stream, S);
            HIPCHK(c, hipGetLastError());
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dfree(d_long_word);
    cleanup_inputs();

    // ---- state + initial pair count (trainer.py:227-235)
    TRY(state_pull(c));
    DevState *h = c->st_host;
    h->iter = 0; h->done = 0; h->halt = 0; h->halt_req = 0; h->sites = 0;
    h->tokens_now = total_bytes;
    h->table_entries = 0;
    TRY(state_push(c));
    TRY(refresh_live_slots(c));
    c->sig_valid = false;
    TRY(table_rebuild(c, 1ull << 18, /*all_bytes=*/true));  // 65,536 possible byte pairs: start at load <= 1/4
    c->stats.table_rebuilds = 0;
    HIPCHK(c, hipEventRecord(ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    c->stats.load_ms = ms;
    c->stats.retiles = 0;
    c->stats.train_ms = 0;
    c->stats.apply_ms_sampled = 0;
    c->stats.apply_launches_sampled = 0;
    c->stats.apply_algo_bytes_sampled = 0;
    c->stats.apply_actual_bytes_sampled = 0;
    c->stats.dense_ms_sampled = 0;
    c->stats.dense_launches_sampled = 0;
    c->stats.dense_launches = 0;
    c->stats.dense_merges = 0;
    c->stats.dense_algo_bytes_sampled = 0;
    c->stats.dense_actual_bytes_sampled = 0;
    c->stats.sparse_ms = 0;
    c->stats.sparse_merges = 0;
    c->stats.sparse_launches = 0;
    c->stats.tail_ms = 0;
    c->stats.tail_merges = 0;
    c->stats.tail_launches = 0;
    c->stats.exchange_ms_sampled = 0;
    c->stats.exchanges_sampled = 0;
    c->stats.scan_ms_sampled = 0;
    c->stats.scan_launches_sampled = 0;
    c->stats.scan_algo_bytes_sampled = 0;
    c->stats.scan_actual_bytes_sampled = 0;
    c->scan_skip_launches = 0;
    c->cand_rebuilds = c->cand_rescans = 0;
    c->fused_launches = 0;
    c->exchanges = 0;
    c->xeff = c->xcap;  // (a new job starts with whole buffers: its first merges are its densest)
    c->xseen[0] = c->xseen[1] = c->xseen[2] = c->xseen[3] = 0;
    c->xeff_sum = 0;
    c->exchange_growths = 0;
    c->exchange_max_records = 0;
    if (c->blk_read) HIPCHK(c, hipMemsetAsync(c->blk_read, 0, MAX_LISTS * 8, c->stream));
    c->stats.algo_bytes_total = 0;
    c->have_words = true;
    return YABPE_OK;
}

// ---------------------------------------------------------------------------------------------- train
static SelectParams select_params(yabpe_ctx *c, uint32_t rec_base, uint32_t n_part, uint32_t n_blk) {
    return SelectParams{c->partials, n_part, c->tt, c->st, c->rec_left, c->rec_right, c->rec_merged,
                        c->rec_count, c->rec_sites, c->rec_live, rec_base, c->table,
                        c->multi ? reinterpret_cast<DeltaHdr *>(c->xsend) : nullptr, c->blk_stats, std::max(n_blk, 1u),
                        c->use_cand ? c->cand_state : nullptr, (!c->weighted && !c->multi) ? 1u : 0u};
}

// The selection as launches of its own (trainer.py:241-251, 296-300): exact argmax over the candidate list, or over the
// whole table when there is no list, then stop rules and merged-token creation.  Selects ONE merge.
static int launch_select(yabpe_ctx *c, uint32_t rec_base) {
    const uint32_t n_part = c->use_cand ? std::min<uint32_t>(c->n_partials_cap, 64) : c->n_partials;
    const SelectParams S = select_params(c, rec_base, n_part, c->blk_used);
    c->blk_used = 0;  // the selection folds and clears them; what follows counts the next merge's grids
    bool selected = false;
    if (c->use_cand) {
        const bool fuse = optv(c, "fuse_select", 1) != 0;
        CandParams CP{c->table, c->tt.rec, c->partials, c->st, c->cand_state, fuse ? c->sel_ticket : nullptr, S};
        hipLaunchKernelGGL(k_argmax_cand, dim3(n_part), dim3(BLOCK), 0, c->stream, CP);
        selected = fuse;
    } else {
        ArgmaxParams A{c->table, c->tt.rec, c->partials, c->st};
        hipLaunchKernelGGL(k_argmax_partial, dim3(c->n_partials), dim3(BLOCK), 0, c->stream, A);
    }
    if (!selected) hipLaunchKernelGGL(k_select, dim3(1), dim3(BLOCK), 0, c->stream, S);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// Can the apply launch of the merges that are selected now end with the selection of the next ones?
static bool can_fuse(yabpe_ctx *c) {
    if (!c->use_cand || !optv(c, "fused", 1)) return false;
    if (c->multi) return true;        // the selection rides on the launch that applies the exchanged records (k_delta_apply)
    if (!c->n_tiles) return false;
    if (!c->split_mode) return true;  // k_apply
    return c->sig && c->sig_valid;    // k_scan_skip
}

// Applies the selected merges (DevState::batch) to the token stream and the pair table (trainer.py:254-294).  fuse: the
// launch that finishes the apply also selects the next merges (its last workgroup; see fused_select_tail) -- one launch per batch.
static int launch_apply(yabpe_ctx *c, uint32_t rec_base, uint32_t tokens_upper, uint32_t apply_grid, EventPair *ev, bool fuse) {
    RankParams R{c->tt, c->st};
    const uint32_t rank_blocks = cdiv64(std::min<uint32_t>(tokens_upper, YB_MAX_TOKENS), BLOCK);
    // lexrank maintenance rides on the apply launch (extra workgroups) whenever there is one over the tiles
    const bool sparse = c->split_mode && c->sig && c->sig_valid;
    const bool rank_rides = c->n_tiles && (fuse || optv(c, "rank_rides", 1)) && (c->split_mode ? sparse : true);
    if (!rank_rides) hipLaunchKernelGGL(k_rank_update, dim3(rank_blocks), dim3(BLOCK), 0, c->stream, R);
    // where the apply pass puts its pair-count updates: the pair table, or (multi-GPU) this rank's send buffer as records
    PairTable out_table = c->table;
    if (c->multi) {
        out_table = PairTable{};
        out_table.sink_hdr = reinterpret_cast<DeltaHdr *>(c->xsend);
        out_table.sink_rec = reinterpret_cast<DeltaRec *>(c->xsend + 16);
        out_table.sink_cap = c->xeff;  // (what this exchange transmits: a rank with more records than that reports it by its count)
    }
    const bool fuse_kernel = fuse && !c->multi;  // the apply launch itself ends with the selection
    if (ev) {
        ev->split = c->split_mode;
        HIPCHK(c, hipEventRecord(ev->e0, c->stream));
    }
    // the long words (more than 63 tokens: outside the tile stream) ride on the fused launch as extra workgroups, like the
    // lexrank maintenance: one signature word per long word and merge, the few words that pass are rewritten
    // (words per workgroup: about two workgroups per CU while there are few long words, BLOCK when there are many)
    const uint32_t long_group = (uint32_t)std::min<uint64_t>(BLOCK, std::max<uint64_t>(32, cdiv64(c->n_long, 2ull * std::max(1, c->n_cu))));
    const LongParams LW{c->long_tok, c->long_off, c->long_len, c->long_freq, c->n_long, out_table, c->st, c->long_sig, c->long_sig_stride, long_group};
    const bool long_rides = fuse_kernel && c->n_long && c->n_tiles && rank_rides;
    const uint32_t long_blocks = long_rides ? cdiv64(c->n_long, long_group) : 0u;
    if (fuse_kernel && c->n_long && !long_rides)  // (no tile launch to ride on: on their own, first -- the fused launch must be the last one to touch the table)
        hipLaunchKernelGGL(k_apply_long, dim3(cdiv64(c->n_long, long_group)), dim3(BLOCK), 0, c->stream, LW);
    // the fused selection folds the per-workgroup counters of THIS launch too: it must know the larger grid
    auto fuse_params = [&](uint32_t grid_now) {
        FuseParams F{};
        if (fuse_kernel) {
            F.ticket = c->sel_ticket;
            F.sel = select_params(c, rec_base, 0u, std::max(c->blk_used, grid_now));
        }
        return F;
    };
    if (c->n_tiles) {
        ApplyParams P{c->tiles, c->tile_len, c->tile_wbase, c->wfreq, c->n_tiles, out_table, c->st, c->blk_stats,
                      c->split_mode ? c->sig : nullptr, c->sig_stride, (uint32_t)AGG_N - 1u, 1u};  // signatures are maintained in the sparse form only
        if (!c->split_mode) {
            // streaming form: one merge per launch, one coalesced pass over the live stream
            const FuseParams F = fuse_params(apply_grid);
            const uint32_t rr_blocks = rank_rides ? rank_blocks : 0u;
            c->blk_used = std::max(c->blk_used, apply_grid);
            if (c->weighted)
                hipLaunchKernelGGL(k_apply<true>, dim3(apply_grid + rr_blocks + long_blocks), dim3(BLOCK), 0, c->stream, P, apply_grid, R, F, LW, apply_grid + rr_blocks);
            else if (optv(c, "hist", 1) && tokens_upper <= (uint32_t)HIST_V)  // (every token id this launch can meet indexes the direct store)
                hipLaunchKernelGGL((k_apply<false, true>), dim3(apply_grid + rr_blocks + long_blocks), dim3(BLOCK), 0, c->stream, P, apply_grid, R, F, LW, apply_grid + rr_blocks);
            else
                hipLaunchKernelGGL(k_apply<false>, dim3(apply_grid + rr_blocks + long_blocks), dim3(BLOCK), 0, c->stream, P, apply_grid, R, F, LW, apply_grid + rr_blocks);
            if (ev) HIPCHK(c, hipEventRecord(ev->e1, c->stream));
        } else {
            // sparse form: skip index + rewrite of the tiles that pass, a batch of merges per launch.
            // A grid that is resident all at once (no second round of workgroups behind the first): each thread tests kt
            // tiles' signatures per merge, a workgroup owns `chunk` consecutive tiles at a time.  Wide workgroups
            // (full_wpb waves, 16 waves per CU either way): a merge late in a job has its sites in a few word types, so
            // every workgroup with a site adds to the SAME few table counts -- same-address atomics are served one after
            // the other, and the queue is as long as there are workgroups; a wider workgroup also evens out how many
            // matched tiles a wave has to rewrite.  (0 = by the merge: 16 waves once a merge has so few sites that the
            // queue on the hot counts is what is left -- and the whole stream fits one round of such workgroups -- else 8)
            if (!sparse) return fail(c, YABPE_E_INTERNAL, "sparse form without signatures");
            int64_t wpb_opt = optv(c, "full_wpb", 0);
            if (wpb_opt <= 0)
                wpb_opt = (c->st_host->best_count <= (unsigned long long)optv(c, "wide_sites_per_cu", 20) * c->n_cu && (c->n_tiles + 2047u) / 2048u <= c->n_cu) ? 16 : 8;
            const uint32_t nw = wpb_opt >= 16 ? 16u : 8u;  // (a 4-wave form existed until round 3: never chosen, pruned)
            const uint32_t nt = nw * 64u;
            const uint32_t target = (uint32_t)std::max<int64_t>(1, optv(c, "full_skip_blocks", (int64_t)c->n_cu * (16 / nw)));
            // tiles per workgroup: what fills `target` workgroups, in whole waves of signature tests, at most kt_max per thread
            const uint32_t ktm = (uint32_t)scan_kt_max((int)nw);
            const uint32_t chunk = std::min<uint32_t>(nt * ktm, std::max<uint32_t>(64u, (uint32_t)(((uint64_t)(c->n_tiles + target - 1) / target + 63u) / 64u * 64u)));
            const uint32_t kt = (chunk + nt - 1) / nt;
            const uint32_t n_chunks = (c->n_tiles + chunk - 1) / chunk;
            const uint32_t scan_grid = std::max(1u, std::min<uint32_t>(std::min<uint32_t>(n_chunks, target), MAX_LISTS));
            if (!c->blk_read) {
                TRY(dmalloc(c, &c->blk_read, MAX_LISTS));
                HIPCHK(c, hipMemsetAsync(c->blk_read, 0, MAX_LISTS * 8, c->stream));
            }
            // (a sparse merge leaves a workgroup a few dozen deltas: a quarter of the aggregator per merge of the batch is
            // plenty -- less to initialise and to flush; the count of the LAST batch's merges bounds this batch's)
            if (optv(c, "agg_small", 1) && !c->weighted && c->st_host->best_count * 4 < 48ull * std::max<uint32_t>(1u, c->n_cu * 3))
                P.agg_mask = (uint32_t)(c->kmax_now > 2 ? AGG_N : nw >= 16 ? AGG_N / 2 : AGG_N / 4) - 1u;
            const uint32_t rr_blocks = rank_rides ? rank_blocks : 0u;
            ScanSkipParams SQ{P, c->blk_read, scan_grid, kt, chunk, R, fuse_params(scan_grid), LW, scan_grid + rr_blocks};
            c->blk_used = std::max(c->blk_used, scan_grid);
            const uint32_t grid = scan_grid + rr_blocks + long_blocks;
#define YB_LAUNCH_SPARSE(NW_)                                                                                        \
    do {                                                                                                             \
        if (c->weighted)                                                                                             \
            hipLaunchKernelGGL((k_scan_skip<true, NW_>), dim3(grid), dim3(NW_ * 64), 0, c->stream, c->st, SQ);      \
        else                                                                                                         \
            hipLaunchKernelGGL((k_scan_skip<false, NW_>), dim3(grid), dim3(NW_ * 64), 0, c->stream, c->st, SQ);     \
    } while (0)
            if (nw == 16) YB_LAUNCH_SPARSE(16);
            else YB_LAUNCH_SPARSE(8);
#undef YB_LAUNCH_SPARSE
            c->scan_skip_launches++;
            if (ev) HIPCHK(c, hipEventRecord(ev->e1, c->stream));
        }
    } else if (ev) {
        HIPCHK(c, hipEventRecord(ev->e1, c->stream));
    }
    if (ev) HIPCHK(c, hipEventRecord(ev->e2, c->stream));
    if (!fuse_kernel && c->n_long) hipLaunchKernelGGL(k_apply_long, dim3(cdiv64(c->n_long, long_group)), dim3(BLOCK), 0, c->stream, LW);
    if (c->multi) {
        // Per batch: the apply launch above left this rank's updates as [header | records] in its send buffer (the
        // aggregator flush writes them there: no delta table, no extraction pass); ONE all-gather; ONE launch adds every
        // rank's records to the replica and -- fused form -- selects the next merges in its last workgroup.
        const uint64_t xbytes = 16 + (uint64_t)c->xeff * sizeof(DeltaRec);  // [header | xeff records] per rank, packed at that stride
        const uint8_t *recv = c->xrecv;
        uint64_t rstride = xbytes;
        if (c->p2p) {  // push into the peers' memory, wait for theirs: one small launch, no collective
            XchgParams X{};
            X.send = c->xsend;
            for (int r = 0; r < c->n_ranks; ++r) X.peer[r] = c->p2p_peer[r];
            X.flags_off = c->p2p_flags_off;
            X.half_bytes = c->p2p_half;
            X.stride = c->p2p_stride;
            X.seq = ++c->p2p_seq;
            X.rank = (uint32_t)c->rank;
            X.n_ranks = (uint32_t)c->n_ranks;
            X.cap = c->xeff;
            X.timeout_ticks = (unsigned long long)optv(c, "p2p_timeout_ms", 20000) * 100000ull;
            X.st = c->st;
            const bool timed = !c->p2p_pending && c->p2p_e0;  // (one exchange per round of launches is timed)
            if (timed) HIPCHK(c, hipEventRecord(c->p2p_e0, c->stream));
            hipLaunchKernelGGL(k_xchg_push, dim3((uint32_t)c->n_ranks), dim3(BLOCK), 0, c->stream, X);
            if (timed) {
                HIPCHK(c, hipEventRecord(c->p2p_e1, c->stream));
                c->p2p_pending = true;  // (read at the end of this round of launches)
            }
            recv = c->p2p_area + (X.seq & 1ull) * c->p2p_half;
            rstride = c->p2p_stride;
        } else {
            TRY(comm_allgather(c, c->xsend, c->xrecv, xbytes));
        }
        c->xeff_sum += c->xeff;
        DeltaApplyParams DA{recv, (uint32_t)c->n_ranks, c->xeff, rstride, c->table, c->st, FuseParams{}, c->p2p ? 1u : 0u};
        if (fuse) {
            DA.F.ticket = c->sel_ticket;
            DA.F.sel = select_params(c, rec_base, 0u, c->blk_used);
        }
        hipLaunchKernelGGL(k_delta_apply, dim3(cdiv64((uint64_t)c->n_ranks * c->xeff, DELTA_RPW)), dim3(BLOCK), 0, c->stream, DA);
        c->exchanges++;
    }
    if (fuse) {
        c->blk_used = 0;  // folded by the selection at the end of that launch
        c->fused_launches++;
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

namespace {
// the timing events of one yabpe_train call: destroyed on every way out
struct TrainEvents {
    hipEvent_t t0 = nullptr, t1 = nullptr, t_split = nullptr, t_tail = nullptr;
    int create() {
        return (hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess && hipEventCreate(&t_split) == hipSuccess &&
                hipEventCreate(&t_tail) == hipSuccess) ? 0 : -1;
    }
    ~TrainEvents() {
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
        if (t_split) (void)hipEventDestroy(t_split);
        if (t_tail) (void)hipEventDestroy(t_tail);
    }
};
}  // namespace

int yabpe_train(yabpe_ctx *c, uint32_t num_merges, uint64_t min_frequency, uint32_t *out_left, uint32_t *out_right,
                uint32_t *out_merged, uint64_t *out_count, uint32_t *out_n_merges) {
    if (!c) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->have_words) return fail(c, YABPE_E_INVALID, "call yabpe_load_words first");
    if (out_n_merges) *out_n_merges = 0;
    TRY(state_pull(c));
    DevState *h = c->st_host;
    // The u16 id space bounds the vocabulary at YB_MAX_TOKENS; the reference has no such bound (trainer.py:238, 298-300).  A
    // merge that re-creates the bytes of an existing token takes no id (trainer.py:298), so the number of merges a job can run
    // is not bounded by the ids that are left: the request stands as it is, and only when the loop needs an id past the last
    // one does the selection stop it (HALT_VOCAB_FULL -> YABPE_E_CAPACITY below) -- reported, never silently truncated.
    // Most such jobs end long before (no pairs left, min_frequency); the per-merge records grow as the loop proceeds.
    const uint32_t rec_base = h->iter;
    if ((uint64_t)rec_base + num_merges > 0xFFFFFFF0ull) return fail(c, YABPE_E_CAPACITY, "more than 2^32 merges in one context");
    h->num_merges = rec_base + num_merges;
    h->min_freq = min_frequency;
    h->done = 0;
    h->halt = 0;
    h->halt_req = 0;
    h->n_batch = 0;
    h->n_select = 0;
    h->batch_others = 0;
    h->kmax = 1;
    h->win_shift = 3;
    TRY(state_push(c));
    c->kmax_now = 1;
    c->rec_n = 0;
    c->log_sites.clear();
    c->log_live.clear();
    c->ev_iter.clear();
    c->ev_us.clear();
    c->ev_scan_us.clear();
    c->split_mode = false;
    c->sig_valid = false;
    if (num_merges == 0) return YABPE_OK;

    // per-merge records: a job cannot run more merges than it has ids for plus the (rare) merges that reuse one; room for the
    // ids, grown below if a job really gets past that
    auto ensure_records = [&](uint32_t need) -> int {
        if (c->rec_cap >= need) return 0;
        const uint32_t ncap = std::max<uint32_t>(need, c->rec_cap + c->rec_cap / 2);
        uint32_t *nl = nullptr, *nr = nullptr, *nm = nullptr;
        unsigned long long *nc = nullptr, *ns = nullptr, *nv = nullptr;
        TRY(dmalloc(c, &nl, ncap)); TRY(dmalloc(c, &nr, ncap)); TRY(dmalloc(c, &nm, ncap));
        TRY(dmalloc(c, &nc, ncap)); TRY(dmalloc(c, &ns, ncap)); TRY(dmalloc(c, &nv, ncap));
        HIPCHK(c, hipMemsetAsync(ns, 0, (size_t)ncap * 8, c->stream));
        if (c->rec_cap && c->rec_n_live) {
            const size_t n = c->rec_n_live;
            HIPCHK(c, hipMemcpyAsync(nl, c->rec_left, n * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(nr, c->rec_right, n * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(nm, c->rec_merged, n * 4, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(nc, c->rec_count, n * 8, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(ns, c->rec_sites, n * 8, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(nv, c->rec_live, n * 8, hipMemcpyDeviceToDevice, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        free_records(c);
        c->rec_left = nl; c->rec_right = nr; c->rec_merged = nm; c->rec_count = nc; c->rec_sites = ns; c->rec_live = nv;
        c->rec_cap = ncap;
        return 0;
    };
    c->rec_n_live = 0;
    const uint32_t rec_first = (uint32_t)std::min<uint64_t>(num_merges, (uint64_t)YB_MAX_TOKENS + 4096);
    if (c->rec_cap < rec_first) { free_records(c); }
    TRY(ensure_records(rec_first));
    HIPCHK(c, hipMemsetAsync(c->rec_sites, 0, (size_t)c->rec_cap * 8, c->stream));

    const uint32_t check = (uint32_t)std::max<int64_t>(1, optv(c, "check_interval", 32));
    const uint32_t ev_sample = (uint32_t)std::max<int64_t>(0, optv(c, "event_sample", 0));
    const uint32_t ev_sample_dense = (uint32_t)std::max<int64_t>(0, optv(c, "event_sample_dense", (int64_t)ev_sample));  // fused k_apply launches
    const double retile_frac = (double)optv(c, "retile_pct", 60) / 100.0;
    const uint64_t retile_min_tiles = (uint64_t)optv(c, "retile_min_tiles", 4096);
    size_t ev_next = 0;

    TrainEvents T;
    if (T.create() != 0) return fail(c, YABPE_E_HIP, "event creation failed");
    HIPCHK(c, hipEventRecord(T.t0, c->stream));
    bool split_marked = false, tail_marked = false;
    uint32_t split_at = 0, tail_at = 0;

    uint32_t i = 0;  // merges selected so far in this call (read back from the device between batches of launches)
    const bool trace_host = optv(c, "trace_host", 0) != 0;  // (debugging a stuck multi-GPU job: where is this rank's host?)
#define YB_TRACE(...) do { if (trace_host) { fprintf(stderr, "[yabpe r%d] ", c->rank); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); fflush(stderr); } } while (0)
    bool finished = false;
    bool skip_cand_once = false;
    bool first_round = true;
    unsigned long long prev_best = 0;
    uint64_t launches_sparse = 0, launches_at_tail = 0, launches_dense = 0, launch_no = 0;
    uint32_t dense_applied = 0;  // merges the streaming launches applied (what was selected when the form changed, less the pending batch)
    uint32_t cand_n_all = 0;  // length of the candidate list at the last read (multi-GPU: of the longest replica's)
    bool sparse_global = false;  // the sparse form has been called for (by any rank)
    c->use_cand = false;
    c->pending = false;
    while (!finished) {
        // (re)size the argmax grid to the table
        uint32_t want_partials = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, c->table_cap / (BLOCK * 8)));
        if (want_partials != c->n_partials) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            dfree(c->partials);
            c->partials = nullptr;
            c->n_partials_cap = std::max<uint32_t>(want_partials, 64);
            TRY(dmalloc(c, &c->partials, c->n_partials_cap));
            c->n_partials = want_partials;
        }
        const uint32_t apply_grid = count_grid(c);
        // sites per merge never increase (the best count is monotone): once they are sparse relative to the
        // number of tiles, switch from the streaming kernel to the skip index + rewrite for good
        const int64_t split_opt = optv(c, "split", -1);  // -1 auto (see want_sparse below: for good, and for all ranks), 0 never, 1 always
        const bool sparse_now = split_opt >= 0 ? split_opt == 1 : sparse_global;
        c->split_mode = sparse_now && c->n_tiles;  // (a rank without tiles launches no apply kernel, but selects like the others)
        if (!tail_marked && i >= num_merges / 2) {  // (measurement: the second half of this call's merges)
            tail_marked = true;
            tail_at = i;
            launches_at_tail = launches_sparse;
            HIPCHK(c, hipEventRecord(T.t_tail, c->stream));
        }
        if (c->split_mode && !split_marked) {  // (measurement: where the streaming phase ends and the sparse phase begins)
            split_marked = true;
            split_at = i;
            dense_applied = i - std::min<uint32_t>(i, c->pending ? h->n_batch : 0u);
            launches_dense = h->n_select - std::min<uint32_t>(h->n_select, c->pending ? 1u : 0u);  // (selections that committed a batch = launches with work; launches queued behind a stop flag return at once)
            HIPCHK(c, hipEventRecord(T.t_split, c->stream));
        }
        if (skip_cand_once) {  // the last batch hit HALT_RESCAN: finish it with the full scan
            c->use_cand = false;
            skip_cand_once = false;
        } else if (h->iter > rec_base) {
            // The list stays exact for as long as the maximum stays >= its threshold T (slots that rise are appended by
            // the argmax itself), so it is rebuilt -- a scan of the whole table -- only now and then: when there is none
            // (start, table rebuilt or grown, fallback), and every `cand_rebuild_every` merges to keep it short.
            const uint32_t every = (uint32_t)std::max<int64_t>(1, optv(c, "cand_rebuild_every", 512));
            // ... and before the maximum can reach T within the next two batches at the pace of the last one: a fallback
            // costs the rest of a batch.  Early in a run the best count falls fast, so the list is rebuilt every batch
            // there; late in the run hardly ever.  A list that has grown long (appended slots) is rebuilt too: the fused
            // selection is one workgroup reading all of it.
            const unsigned long long drop = prev_best > h->best_count ? prev_best - h->best_count : 0;
            const bool near_T = h->best_count < c->table.cand_T + 2 * drop + 1;
            // (multi-GPU: a pair can be listed more often on one replica than on another -- the longest list decides for all,
            // every rank must rebuild at the same merges or their thresholds T drift apart)
            const bool long_list = cand_n_all > std::max<uint32_t>(4u * BLOCK - 64u, (uint32_t)optv(c, "cand_target", 768));  // (one pass of the fused selection: 4 entries per thread)
            if (!c->use_cand || i - c->cand_built_at >= every || i < c->cand_built_at || near_T || long_list) {
                YB_TRACE("cand_rebuild at %u", i);
                TRY(cand_rebuild(c, h->best_count));
                c->cand_built_at = i;
                c->cand_best_at_build = h->best_count;
            }
            prev_best = h->best_count;
        }
        if (c->split_mode) {
            // signatures are built when the sparse form starts and refreshed now and then
            // (rewrites only ever ADD bits: every merged site leaves up to two new pairs and three stale ones behind)
            const uint32_t every = (uint32_t)std::max<int64_t>(1, optv(c, "sig_rebuild_every", 16384));
            const uint64_t merged_since = c->sig_tokens_at_build > h->tokens_now ? c->sig_tokens_at_build - h->tokens_now : 0;
            const bool stale = merged_since * 100 > c->sig_tokens_at_build * (uint64_t)optv(c, "sig_rebuild_pct", 20);
            if (!c->sig_valid || i - c->sig_built_at >= every || i < c->sig_built_at || stale) {
                YB_TRACE("build_signatures at %u", i);
                TRY(build_signatures(c));
                c->sig_valid = true;
                c->sig_built_at = i;
            }
        }
        // merges one selection may take: a batch in the sparse form.  The streaming form applies one merge per pass: a pass that
        // applies a batch was built and measured (DESIGN (c)) -- a tenth slower per pass, and the first merges of a job hardly ever
        // batch (they are the consecutive pairs of the most frequent words: rule (1))
        const uint32_t kmax = sparse_now ? (uint32_t)std::max<int64_t>(1, std::min<int64_t>(KMAX, optv(c, "batch_max", KMAX))) : 1u;
        if (kmax != c->kmax_now) {
            HIPCHK(c, hipMemcpyAsync(&c->st->kmax, &kmax, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            c->kmax_now = kmax;
            h->kmax = kmax;
        }
        // records for everything the launches of this round can select
        const uint32_t n_launch = (first_round && !sparse_now) ? std::min<uint32_t>(check, 8) : check;  // (the same on every rank: a launch is an exchange)
        first_round = false;
        c->rec_n_live = i;
        TRY(ensure_records((uint32_t)std::min<uint64_t>(num_merges, (uint64_t)i + (uint64_t)(n_launch + 2) * kmax)));
        auto sample = [&](uint32_t iter_rel) -> EventPair * {  // (every Nth LAUNCH; iter_rel: the merges selected so far, a lower bound in the fused forms)
            const uint32_t every = c->split_mode ? ev_sample : ev_sample_dense;
            if (!every || (launch_no++ % every) != 0) return nullptr;
            if (ev_next == c->events.size()) {
                EventPair n{};
                if (hipEventCreate(&n.e0) != hipSuccess || hipEventCreate(&n.e1) != hipSuccess || hipEventCreate(&n.e2) != hipSuccess) return nullptr;
                c->events.push_back(n);
            }
            c->events[ev_next].iter_rel = iter_rel;
            return &c->events[ev_next++];
        };
        // Fused form: one launch applies the pending merges and selects the next ones; otherwise a merge is a selection
        // launch followed by its apply launch(es).  `i` counts selected merges exactly while one launch selects one merge
        // (streaming form, unfused); in the sparse form a launch selects a batch and `i` is a lower bound until the next read.
        const bool fuse = can_fuse(c);
        const uint32_t tok_upper0 = h->n_tokens;
        uint32_t launched = 0;
        if (c->pending && !fuse) {  // leave the fused form: the selected merges are applied on their own
            TRY(launch_apply(c, rec_base, tok_upper0 + kmax, apply_grid, sample(i ? i - 1 : 0), false));
            c->pending = false;
        }
        if (fuse && !c->pending && i < num_merges) {  // enter it: a selection on its own
            TRY(launch_select(c, rec_base));
            c->pending = true;
            ++i;
        }
        for (uint32_t l = 0; l < n_launch && i < num_merges + (fuse ? 1u : 0u); ++l) {
            ++launched;
            const uint32_t tok_upper = tok_upper0 + (launched + 1u) * kmax + 1u;
            if (fuse) {
                // (i == num_merges: every merge is selected, the last ones are pending -- this launch applies them and its
                // selection finds the limit reached: done.  An apply launch must never run twice on one selection: in the
                // multi-GPU form the send buffer would still hold the records of the first time.)
                const bool last = i >= num_merges;
                TRY(launch_apply(c, rec_base, tok_upper, apply_grid, sample(i - 1), true));
                if (kmax == 1) ++i; else if (sparse_now) ++launches_sparse;  // (a batch per launch: i stays a lower bound until the next read)
                if (last) break;
            } else {
                TRY(launch_select(c, rec_base));
                TRY(launch_apply(c, rec_base, tok_upper, apply_grid, sample(i), false));
                ++i;
            }
        }
        YB_TRACE("round of %u launches queued (p2p seq %llu), waiting", launched, (unsigned long long)c->p2p_seq);
        TRY(fold_stats(c));
        TRY(state_pull(c));
        i = h->iter - rec_base;                              // what the device really selected
        YB_TRACE("round done: %u merges, halt %u req %u done %u", i, h->halt, h->halt_req, h->done);
        if (c->p2p_pending) {  // (the stream is idle: the pair has completed)
            float xms = 0;
            if (hipEventElapsedTime(&xms, c->p2p_e0, c->p2p_e1) == hipSuccess) {
                c->stats.exchange_ms_sampled += xms;
                c->stats.exchanges_sampled += 1;
            } else {
                (void)hipGetLastError();
            }
            c->p2p_pending = false;
        }
        if (h->halt_req && !h->halt) h->halt = h->halt_req;  // raised by the last apply of the batch
        if (h->done || h->halt) c->pending = false;          // the last selection of the batch did not select
        if (c->multi && c->xrecv) {
            // what the exchanges of this batch really carried (the largest count of any rank, the same number on every rank):
            // the next batch transmits twice that plus a margin, never less than 2,048 records, never more than the buffers hold
            const uint32_t seen = h->xmax;
            c->exchange_max_records = std::max<uint64_t>(c->exchange_max_records, seen);
            c->xseen[3] = c->xseen[2]; c->xseen[2] = c->xseen[1]; c->xseen[1] = c->xseen[0]; c->xseen[0] = seen;
            if (optv(c, "delta_adapt", 1) && h->halt != HALT_DELTA_FULL && h->iter - rec_base >= 4 * check) {
                // (an overflow costs a recount of the whole stream: twice the largest count of the last four batches, and
                // nothing shrinks during the first batches of a job, whose merges differ most from one another)
                const uint64_t recent = std::max(std::max(c->xseen[0], c->xseen[1]), std::max(c->xseen[2], c->xseen[3]));
                const uint64_t gran = (uint64_t)std::max<int64_t>(1, optv(c, "delta_granule", 1024));
                const uint64_t raw = recent * (uint64_t)std::max<int64_t>(1, optv(c, "delta_headroom_pct", 200)) / 100ull + (uint64_t)std::max<int64_t>(0, optv(c, "delta_margin", 1024));
                const uint64_t want = ((raw + gran - 1) / gran) * gran;
                c->xeff = (uint32_t)std::min<uint64_t>(c->xcap, std::max<uint64_t>(want, std::min<uint64_t>((uint64_t)std::max<int64_t>(1, optv(c, "delta_floor", 2048)), c->xcap)));
            }
            if (optv(c, "trace_exchange", 0) && c->rank == 0) fprintf(stderr, "[yabpe] exchange: merges %u records<= %u next %u halt %u\n", h->iter - rec_base, seen, c->xeff, h->halt);
            HIPCHK(c, hipMemsetAsync(&c->st->xmax, 0, sizeof(uint32_t), c->stream));
            h->xmax = 0;
        }
        cand_n_all = h->cand_n;
        // does this rank's shard call for the sparse form?  (sites per merge never increase -- the best count is monotone: once
        // they are sparse relative to the number of tiles.  Pooled words: the count is weighted, what matters is how many
        // resident sites the last merge had.)
        unsigned long long want_sparse = (c->n_tiles && h->iter > rec_base &&
                 (c->weighted && h->sites ? h->sites : h->best_count) * 100 < (unsigned long long)c->n_tiles * (unsigned long long)optv(c, "split_pct", 100)) ? 1 : 0;
        if (c->multi) {  // replicas must be in lockstep: same merge count, same flags
            unsigned long long sig = ((unsigned long long)h->iter << 8) | (h->done ? 1u : 0u) | ((unsigned long long)(h->halt & 0x3f) << 1);
            // ... and agree on what must not differ between them: the batch limit follows the form (any rank sparse: all of
            // them), the candidate list is rebuilt for the longest replica's length
            unsigned long long v[3] = {sig, h->cand_n, want_sparse};
            YB_TRACE("comm_max3");
            TRY(comm_max3(c, v));
            YB_TRACE("comm_max3 back");
            if (v[0] != sig) return fail(c, YABPE_E_COMM, "ranks diverged (iter/flags %llx vs max %llx)", sig, v[0]);
            cand_n_all = (uint32_t)v[1];
            want_sparse = v[2];
        }
        if (want_sparse) sparse_global = true;
        if (h->halt == HALT_RESCAN) {  // the candidate set could not prove the maximum: redo that merge with the full scan
            h->halt = 0; h->halt_req = 0;
            TRY(state_push(c));
            c->cand_rescans++;
            skip_cand_once = true;
            while (ev_next > 0 && c->events[ev_next - 1].iter_rel >= i) --ev_next;  // those launches did nothing: timed again in the re-run
            continue;
        }
        if (h->halt) {
            if (h->halt == HALT_TABLE_FULL || h->halt == HALT_DELTA_FULL) {
                // the streams are consistent (the apply pass finished everywhere); only table updates were lost:
                // rebuild the table from the token streams, bigger
                const bool delta_full = h->halt == HALT_DELTA_FULL;
                h->halt = 0; h->halt_req = 0;
                TRY(state_push(c));
                if (delta_full) {  // a merge produced more records on some rank than an exchange transmits: 4x for everybody
                    if (c->xeff < c->xcap) {
                        c->xeff = (uint32_t)std::min<uint64_t>(c->xcap, 4ull * c->xeff);  // (the buffers hold more: no reallocation)
                    } else {
                        YB_TRACE("comm_buffers %u -> %u", c->xcap, c->xcap * 4);
                        TRY(comm_buffers(c, c->xcap * 4));
                    }
                    c->exchange_growths++;
                }
                YB_TRACE("table_rebuild (delta_full %d)", (int)delta_full);
                TRY(table_rebuild(c, delta_full ? c->table_cap : c->table_cap * 4));
                YB_TRACE("table_rebuild back");
                TRY(state_pull(c));
                i = h->iter - rec_base;  // resume after the last recorded merge
                while (ev_next > 0 && c->events[ev_next - 1].iter_rel >= i) --ev_next;
                continue;
            }
            if (h->halt == HALT_COMM) {
                // what this rank's receive area holds: the exchange it waited for last, and the flags of both halves per sender
                std::string seen;
                if (c->p2p && c->p2p_area) {
                    std::vector<unsigned long long> fl(2 * (size_t)c->n_ranks);
                    if (hipMemcpy(fl.data(), c->p2p_area + c->p2p_flags_off, fl.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
                        for (size_t q = 0; q < fl.size(); ++q) seen += (q ? " " : "") + std::to_string(fl[q]);
                    else
                        (void)hipGetLastError();
                }
                return fail(c, YABPE_E_COMM, "a peer's records did not arrive within the time limit (peer-to-peer exchange) after %u merges; rank %d launched exchange %llu, "
                            "flags in its receive area [half][sender]: %s", h->iter - rec_base, c->rank, (unsigned long long)c->p2p_seq, seen.c_str());
            }
            const char *why = h->halt == HALT_POOL_FULL ? "token byte pool exhausted (option pool_bytes)"
                              : h->halt == HALT_VOCAB_FULL ? "u16 token id space exhausted"
                                                           : "device halt";
            return fail(c, YABPE_E_CAPACITY, "%s after %u merges", why, h->iter - rec_base);
        }
        if (h->done || (i >= num_merges && !c->pending)) {
            finished = true;
            break;
        }
        // housekeeping between batches
        // kept at 12-30 % load: a key that is not at its home slot costs the flush of every merge a dependent trip (the
        // 1 GiB job: 1.60 s at 50-70 % load, 1.42 s at 12-30 %), and the table is scanned only now and then (candidate
        // rebuilds).  Same decision on every rank.
        if (h->table_entries * 100 > c->table_cap * (uint64_t)optv(c, "table_load_pct", 30))
        {
            YB_TRACE("table_grow");
            TRY(table_grow(c, std::max<uint64_t>(h->table_entries * (uint64_t)std::max<int64_t>(2, optv(c, "table_grow_x", 8)), 1ull << 16)));
            YB_TRACE("table_grow back");
        }
        if (c->n_tiles >= retile_min_tiles && !c->weighted &&
            (double)h->live_slots < retile_frac * (double)c->n_tiles * SPAN) {
            YB_TRACE("retile");
            TRY(retile_flat(c));
        }
    }
    HIPCHK(c, hipEventRecord(T.t1, c->stream));
    HIPCHK(c, hipEventSynchronize(T.t1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, T.t0, T.t1));
    if (split_marked) {
        float sms = 0;
        HIPCHK(c, hipEventElapsedTime(&sms, T.t_split, T.t1));
        c->stats.sparse_ms += sms;
        c->stats.sparse_merges += (h->iter - rec_base) > split_at ? (h->iter - rec_base) - split_at : 0;
        c->stats.sparse_launches += launches_sparse;
    }
    if (tail_marked) {
        float tms = 0;
        HIPCHK(c, hipEventElapsedTime(&tms, T.t_tail, T.t1));
        c->stats.tail_ms += tms;
        c->stats.tail_merges += (h->iter - rec_base) > tail_at ? (h->iter - rec_base) - tail_at : 0;
        c->stats.tail_launches += launches_sparse - launches_at_tail;
    }
    c->stats.train_ms += ms;

    // close the log of the last iteration and fold the remaining sites into T_i
    TRY(state_pull(c));
    const uint32_t n = h->iter - rec_base;
    c->rec_n = n;
    c->log_sites.assign(n, 0);
    c->log_live.assign(n, 0);
    if (n) {
        HIPCHK(c, hipMemcpy(c->log_sites.data(), c->rec_sites, (size_t)n * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(c->log_live.data(), c->rec_live, (size_t)n * 8, hipMemcpyDeviceToHost));
        c->log_sites[n - 1] += h->sites;  // (already closed by the selection when the last launch was a fused one: then h->sites is 0)
    }
    const uint64_t tokens_at_start = h->tokens_now - h->sites + [&] { uint64_t s = 0; for (uint32_t k = 0; k < n; ++k) s += c->log_sites[k]; return s; }();
    h->tokens_now -= h->sites;
    h->sites = 0;
    TRY(state_push(c));
    // algorithmic bytes: A_i = 2 * (T_i + W)
    {
        uint64_t T_ = tokens_at_start;
        std::vector<uint64_t> Ti(n);
        for (uint32_t k = 0; k < n; ++k) {
            Ti[k] = T_;
            c->stats.algo_bytes_total += 2 * (T_ + c->n_words_input);
            T_ -= c->log_sites[k];
        }
        // the streaming phase as a whole: its launches, the merges they applied, their algorithmic bytes; a launch applies a batch,
        // so the event-timed launches get the phase's bytes per launch (merge indices of sampled launches are lower bounds)
        if (!split_marked) {
            dense_applied = n - std::min<uint32_t>(n, (c->pending && !h->done) ? h->n_batch : 0u);
            launches_dense = h->n_select - std::min<uint32_t>(h->n_select, (c->pending && !h->done) ? 1u : 0u);
        }
        dense_applied = std::min(dense_applied, n);
        uint64_t dense_algo = 0, dense_actual = 0, dense_sampled_now = 0;
        for (uint32_t k = 0; k < dense_applied; ++k) {
            dense_algo += 2 * (Ti[k] + c->n_words_input);
            dense_actual += 2 * c->log_live[k] + 4ull * c->n_tiles;
        }
        // (actual: a pass reads the stream once for all the merges of its batch -- the first merge's live slots)
        c->stats.dense_launches += launches_dense;
        c->stats.dense_merges += dense_applied;
        for (size_t e = 0; e < ev_next; ++e) {
            uint32_t k = c->events[e].iter_rel;
            if (k >= n) continue;
            float ems = 0, sms = 0;
            if (hipEventElapsedTime(&ems, c->events[e].e0, c->events[e].e2) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (hipEventElapsedTime(&sms, c->events[e].e0, c->events[e].e1) != hipSuccess) { (void)hipGetLastError(); continue; }
            c->stats.apply_ms_sampled += ems;
            c->ev_iter.push_back(k);
            c->ev_us.push_back(ems * 1000.0f);
            c->ev_scan_us.push_back(sms * 1000.0f);
            c->stats.apply_launches_sampled += 1;
            // streaming form: one pass over the live stream + rewrite (+ selection when fused), a batch of merges.  (A launch queued
            // behind a stop flag -- the rest of a round after a rescan request -- returns in microseconds: not a pass.)
            if (!c->events[e].split && ems >= 0.02f) {
                c->stats.dense_ms_sampled += ems;
                dense_sampled_now += 1;
            }
            if (c->events[e].split) {  // (sparse form: the merge index of a sampled launch is a lower bound -- launches select batches)
                c->stats.scan_ms_sampled += sms;
                c->stats.scan_launches_sampled += 1;
                c->stats.scan_algo_bytes_sampled += 2 * (Ti[k] + c->n_words_input);
                c->stats.scan_actual_bytes_sampled += 2 * c->log_live[k] + 4ull * c->n_tiles;
            }
            c->stats.apply_algo_bytes_sampled += 2 * (Ti[k] + c->n_words_input);
            c->stats.apply_actual_bytes_sampled += 2 * c->log_live[k] + 4ull * c->n_tiles;
        }
        if (launches_dense && dense_sampled_now) {
            c->stats.dense_launches_sampled += dense_sampled_now;
            c->stats.dense_algo_bytes_sampled += dense_algo * dense_sampled_now / launches_dense;
            c->stats.dense_actual_bytes_sampled += dense_applied ? dense_actual / dense_applied * dense_sampled_now : 0;  // (one read of the live stream per launch)
        }
    }
    if (n) {
        if (out_left) HIPCHK(c, hipMemcpy(out_left, c->rec_left, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (out_right) HIPCHK(c, hipMemcpy(out_right, c->rec_right, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (out_merged) HIPCHK(c, hipMemcpy(out_merged, c->rec_merged, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (out_count) HIPCHK(c, hipMemcpy(out_count, c->rec_count, (size_t)n * 8, hipMemcpyDeviceToHost));
    }
    if (out_n_merges) *out_n_merges = n;
    if (optv(c, "verify", 0)) {
        uint64_t mm = 0;
        TRY(yabpe_verify_table(c, &mm));
        if (mm) return fail(c, YABPE_E_INTERNAL, "incremental pair table differs from a recount in %llu keys", (unsigned long long)mm);
    }
    return YABPE_OK;
}

int yabpe_n_tokens(yabpe_ctx *c, uint32_t *out) {
    if (!c || !out) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    TRY(state_pull(c));
    *out = c->st_host->n_tokens;
    return YABPE_OK;
}

int yabpe_token_bytes(yabpe_ctx *c, uint32_t id, uint8_t *out, uint32_t cap, uint32_t *out_len) {
    if (!c || !out_len) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    TRY(state_pull(c));
    if (id >= c->st_host->n_tokens) return fail(c, YABPE_E_INVALID, "token id %u out of range", id);
    uint32_t off = 0, len = 0;
    HIPCHK(c, hipMemcpy(&off, c->tt.off + id, 4, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&len, c->tt.len + id, 4, hipMemcpyDeviceToHost));
    *out_len = len;
    if (out && cap >= len && len) HIPCHK(c, hipMemcpy(out, c->tt.pool + off, len, hipMemcpyDeviceToHost));
    return YABPE_OK;
}

// ---------------------------------------------------------------------------------------------- stats / debug
int yabpe_stats(yabpe_ctx *c, yabpe_stats_t *out) {
    if (!c || !out) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    TRY(state_pull(c));
    c->stats.n_words = c->n_words;
    c->stats.n_words_input = c->n_words_input;
    c->stats.n_long_words = c->n_long;
    c->stats.tokens_initial = c->tokens_initial;
    c->stats.tokens_now = c->st_host->tokens_now;
    c->stats.merges_done = c->st_host->iter;
    c->stats.n_tiles = c->n_tiles;
    c->stats.live_slots = c->st_host->live_slots;
    c->stats.table_capacity = c->table_cap;
    c->stats.table_entries = c->st_host->table_entries;
    c->stats.cand_rebuilds = c->cand_rebuilds;
    c->stats.cand_rescans = c->cand_rescans;
    c->stats.fused_launches = c->fused_launches;
    c->stats.exchanges = c->exchanges;
    c->stats.exchange_bytes = (c->multi && c->exchanges) ? (16 + (c->xeff_sum / c->exchanges) * sizeof(DeltaRec)) * (uint64_t)c->n_ranks : 0;  // (mean over the exchanges)
    c->stats.exchange_cap_records = c->xcap;
    c->stats.exchange_p2p = c->p2p ? 1 : 0;
    c->stats.exchange_growths = c->exchange_growths;
    c->stats.exchange_max_records = c->exchange_max_records;
    c->stats.scan_skip_launches = c->scan_skip_launches;
    c->stats.scan_skip_tiles_read = 0;
    if (c->blk_read) {
        std::vector<unsigned long long> br(MAX_LISTS);
        HIPCHK(c, hipMemcpy(br.data(), c->blk_read, MAX_LISTS * 8, hipMemcpyDeviceToHost));
        for (auto v : br) c->stats.scan_skip_tiles_read += v;
    }
    *out = c->stats;
    return YABPE_OK;
}

int yabpe_latency_probe(yabpe_ctx *c, yabpe_latency_t *out) {
    if (!c || !out) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    // 2 GiB of u32 (eight times the Infinity Cache: a hop finds its line in no cache); 512 MiB when the device is nearly full
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    const uint32_t n = free_b > (8ull << 30) ? (1u << 29) : (1u << 27);
    uint32_t *next = nullptr, *sink = nullptr;
    unsigned long long *ticks = nullptr;
    struct Guard {  // (every way out frees the probe's buffers)
        uint32_t *&a, *&b;
        unsigned long long *&t;
        ~Guard() { dfree(a); dfree(b); dfree(t); }
    } guard{next, sink, ticks};
    TRY(dmalloc(c, &next, n));
    TRY(dmalloc(c, &sink, 1));
    TRY(dmalloc(c, &ticks, 1));
    double trip[3] = {0, 0, 0};
    const uint32_t hops = 1024;
    for (int mode = 0; mode < 3; ++mode) {
        // (the table is rewritten before every walk: whatever the last walk left in the caches is pushed out, and each walk is timed once, cold)
        hipLaunchKernelGGL(k_chain_init, dim3(n / 256), dim3(256), 0, c->stream, next, n, (uint32_t)mode);  // (one cycle over all entries, no fixed stride)
        hipLaunchKernelGGL(k_chain_walk, dim3(1), dim3(1), 0, c->stream, next, hops, mode, ticks, sink);
        HIPCHK(c, hipGetLastError());
        unsigned long long t = 0;
        HIPCHK(c, hipMemcpyAsync(&t, ticks, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        trip[mode] = (double)t / 100.0 / hops;
    }
    // dependent empty launches back to back
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    double gap = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        const int n_launch = 400;
        for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, c->stream, sink);
        HIPCHK(c, hipEventRecord(e0, c->stream));
        for (int k = 0; k < n_launch; ++k) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, c->stream, sink);
        HIPCHK(c, hipEventRecord(e1, c->stream));
        HIPCHK(c, hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        gap = std::min(gap, (double)ms * 1000.0 / n_launch);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    out->launch_gap_us = gap;
    out->load_trip_us = trip[0];
    out->coherent_trip_us = trip[1];
    out->atomic_trip_us = trip[2];
    return YABPE_OK;
}

int yabpe_iter_log(yabpe_ctx *c, uint64_t *out_sites, uint64_t *out_live_slots, uint32_t cap, uint32_t *out_n) {
    if (!c || !out_n) return YABPE_E_INVALID;
    uint32_t n = std::min<uint32_t>(cap, (uint32_t)c->log_sites.size());
    if (out_sites) memcpy(out_sites, c->log_sites.data(), (size_t)n * 8);
    if (out_live_slots) memcpy(out_live_slots, c->log_live.data(), (size_t)n * 8);
    *out_n = (uint32_t)c->log_sites.size();
    return YABPE_OK;
}

int yabpe_event_log(yabpe_ctx *c, uint32_t *out_iter, float *out_us, float *out_scan_us, uint32_t cap, uint32_t *out_n) {
    if (!c || !out_n) return YABPE_E_INVALID;
    uint32_t n = std::min<uint32_t>(cap, (uint32_t)c->ev_iter.size());
    if (out_iter) memcpy(out_iter, c->ev_iter.data(), (size_t)n * 4);
    if (out_us) memcpy(out_us, c->ev_us.data(), (size_t)n * 4);
    if (out_scan_us) memcpy(out_scan_us, c->ev_scan_us.data(), (size_t)n * 4);
    *out_n = (uint32_t)c->ev_iter.size();
    return YABPE_OK;
}

int yabpe_verify_table(yabpe_ctx *c, uint64_t *out_mismatches) {
    if (!c || !out_mismatches) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->have_words) return fail(c, YABPE_E_INVALID, "no corpus loaded");
    PairTable scratch{};
    HIPCHK(c, hipMemsetAsync(&c->scratch64[4], 0, 16, c->stream));  // [4] entries, [5] mismatches
    TRY(table_alloc(c, scratch, c->table_cap, &c->scratch64[4]));
    TRY(launch_count(c, scratch));
    uint32_t grid = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(1, c->table_cap / BLOCK));
    const bool dbg = getenv("YABPE_DEBUG_VERIFY") != nullptr;
    unsigned long long *dump = nullptr;
    if (dbg) {
        TRY(dmalloc(c, &dump, 48));
        HIPCHK(c, hipMemsetAsync(dump, 0, 48 * 8, c->stream));
    }
    CmpParams A{c->table, scratch, &c->scratch64[5], dump};
    hipLaunchKernelGGL(k_table_compare, dim3(grid), dim3(BLOCK), 0, c->stream, A);
    CmpParams B{scratch, c->table, &c->scratch64[5], dump};
    hipLaunchKernelGGL(k_table_compare, dim3(grid), dim3(BLOCK), 0, c->stream, B);
    HIPCHK(c, hipGetLastError());
    unsigned long long mm = 0;
    HIPCHK(c, hipMemcpyAsync(&mm, &c->scratch64[5], 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (dbg) {
        unsigned long long d[48];
        HIPCHK(c, hipMemcpy(d, dump, sizeof d, hipMemcpyDeviceToHost));
        for (unsigned i = 0; i < 16 && i < mm; ++i)
            fprintf(stderr, "[yabpe verify r%d] key (%llu,%llu) first %lld second %lld\n", c->rank, d[3 * i] >> 16, d[3 * i] & 0xffffull, (long long)d[3 * i + 1], (long long)d[3 * i + 2]);
        dfree(dump);
    }
    table_free(scratch);
    // the recount may have raised halt_req on the scratch table; it is not a training halt
    HIPCHK(c, hipMemsetAsync(&c->st->halt_req, 0, 4, c->stream));
    *out_mismatches = mm;
    return YABPE_OK;
}

int yabpe_stream_checksum(yabpe_ctx *c, uint64_t *out_sum, uint64_t *out_words, uint64_t *out_tokens) {
    if (!c) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->have_words) return fail(c, YABPE_E_INVALID, "no corpus loaded");
    HIPCHK(c, hipMemsetAsync(&c->scratch64[4], 0, 24, c->stream));
    if (c->n_tiles) {
        ChecksumParams P{c->tiles, c->tile_len, c->tile_wbase, c->wfreq, c->n_tiles, c->tt,
                         &c->scratch64[4], &c->scratch64[5], &c->scratch64[6]};
        hipLaunchKernelGGL(k_stream_checksum, dim3(cdiv64(c->n_tiles, BLOCK)), dim3(BLOCK), 0, c->stream, P);
        HIPCHK(c, hipGetLastError());
    }
    unsigned long long r[3];
    HIPCHK(c, hipMemcpyAsync(r, &c->scratch64[4], 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out_sum) *out_sum = r[0];
    if (out_words) *out_words = r[1];
    if (out_tokens) *out_tokens = r[2];
    return YABPE_OK;
}

int yabpe_memcpy_d2h(yabpe_ctx *c, void *dst_host, const void *src_dev, uint64_t n) {
    if (!c) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(dst_host, src_dev, n, hipMemcpyDeviceToHost));
    return YABPE_OK;
}

#ifdef YB_PROFILE_SCAN
int yabpe_debug_sel_profile(unsigned long long out[16]) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(yb::g_sel_prof), 128) == hipSuccess ? 0 : -1;
}
int yabpe_debug_sel_acc(unsigned long long out[72]) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(yb::g_sel_acc), 72 * 8) == hipSuccess ? 0 : -1;
}
int yabpe_debug_ss_profile(unsigned long long out[8]) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(yb::g_ss_prof), 64) == hipSuccess ? 0 : -1;
}
int yabpe_debug_launch_profile(unsigned long long *out, int reset) { // 65536 x 4 u64
    if (reset) {
        std::vector<unsigned long long> z(65536 * 4, 0ull);
        for (size_t i = 0; i < 65536; ++i) z[i * 4] = ~0ull;
        return hipMemcpyToSymbol(HIP_SYMBOL(yb::g_launch_prof), z.data(), z.size() * 8) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(yb::g_launch_prof), (size_t)65536 * 32) == hipSuccess ? 0 : -1;
}
int yabpe_debug_stop_hist(unsigned long long *out) { // 65536 x u64
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(yb::g_stop_hist), (size_t)65536 * 8) == hipSuccess ? 0 : -1;
}
int yabpe_debug_scan_profile(unsigned long long *out, uint32_t n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(yb::g_scan_prof), (size_t)std::min<uint32_t>(n_blocks, yb::MAX_LISTS_PROF) * 64) == hipSuccess ? 0 : -1;
}
int yabpe_debug_flush_profile(unsigned long long *out, uint32_t n_blocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(yb::g_flush_prof), (size_t)std::min<uint32_t>(n_blocks, yb::FLUSH_PROF_BLOCKS) * 32) == hipSuccess ? 0 : -1;
}
#endif
#ifdef YB_PROFILE_SLOW
int yabpe_debug_slow_profile(unsigned long long out[8]) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(yb::g_slow_prof), 64) == hipSuccess ? 0 : -1;
}
#endif

int yabpe_memcpy_h2d(yabpe_ctx *c, void *dst_dev, const void *src_host, uint64_t n) {
    if (!c) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(dst_dev, src_host, n, hipMemcpyHostToDevice));
    return YABPE_OK;
}

// ---------------------------------------------------------------------------------------------- synthetic corpus
int yabpe_synth_generate(yabpe_ctx *c, uint64_t target_bytes, uint32_t n_types, uint64_t seed, const uint8_t *alphabet,
                         uint32_t alphabet_len, int space_prefix, uint8_t **out_dev_bytes, uint64_t **out_dev_off,
                         uint64_t *out_n_words, uint64_t *out_n_bytes) {
    if (!c || !alphabet || !alphabet_len || !n_types || !out_dev_bytes || !out_dev_off || !out_n_words || !out_n_bytes)
        return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    SynthOut so{};
    int r = synth_generate(c->stream, target_bytes, n_types, seed, alphabet, alphabet_len, space_prefix, &so);
    if (r != 0) return fail(c, r == -2 ? YABPE_E_INTERNAL : YABPE_E_HIP, "synthetic generation failed (%d): %s", r, hipGetErrorString(hipGetLastError()));
    c->synth_bufs.push_back(so.bytes);
    c->synth_bufs.push_back(so.off);
    *out_dev_bytes = so.bytes;
    *out_dev_off = (uint64_t *)so.off;
    *out_n_words = so.n_words;
    *out_n_bytes = so.n_bytes;
    return YABPE_OK;
}

int yabpe_synth_generate_lex(yabpe_ctx *c, uint64_t target_bytes, uint32_t n_types, uint64_t seed, const uint8_t *lex_bytes,
                             const uint64_t *lex_off, uint8_t **out_dev_bytes, uint64_t **out_dev_off, uint64_t *out_n_pieces, uint64_t *out_n_bytes) {
    if (!c || !lex_bytes || !lex_off || !n_types || !out_dev_bytes || !out_dev_off || !out_n_pieces || !out_n_bytes) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    for (uint32_t j = 0; j < n_types; ++j)
        if (lex_off[j + 1] <= lex_off[j] || lex_off[j + 1] - lex_off[j] > 0xFFFFu) return fail(c, YABPE_E_INVALID, "lexicon entry %u is empty or longer than 65,535 bytes", j);
    // expected bytes per draw under the Zipf weights: how many draws the target needs (more if that falls short)
    long double num = 0, den = 0;
    for (uint32_t j = 0; j < n_types; ++j) {
        const long double w = (long double)((1ull << 40) / (j + 1ull));
        num += w * (long double)(lex_off[j + 1] - lex_off[j]);
        den += w;
    }
    const double mean_len = (double)(num / den);
    SynthOut so{};
    int r = -2;
    for (double slack = 1.02; r == -2 && slack < 3.0; slack *= 1.25)
        r = synth_generate_lex(c->stream, target_bytes, n_types, seed, lex_bytes, (const unsigned long long *)lex_off,
                               (unsigned long long)((double)target_bytes / mean_len * slack) + 65536, &so);
    if (r != 0) return fail(c, r == -2 ? YABPE_E_INTERNAL : YABPE_E_HIP, "synthetic text generation failed (%d): %s", r, hipGetErrorString(hipGetLastError()));
    c->synth_bufs.push_back(so.bytes);
    c->synth_bufs.push_back(so.off);
    *out_dev_bytes = so.bytes;
    *out_dev_off = (uint64_t *)so.off;
    *out_n_pieces = so.n_words;
    *out_n_bytes = so.n_bytes;
    return YABPE_OK;
}

int yabpe_synth_free(yabpe_ctx *c) {
    if (!c) return YABPE_E_INVALID;
    for (void *p : c->synth_bufs) dfree(p);
    c->synth_bufs.clear();
    return YABPE_OK;
}

// ---------------------------------------------------------------------------------------------- pre-tokeniser
int yabpe_pretokenize(yabpe_ctx *c, const uint8_t *text, uint64_t n_bytes, const uint64_t *chunk_off, uint32_t n_chunks,
                      const uint8_t *special_bytes, const uint32_t *special_off, uint32_t n_special,
                      const uint8_t **out_dev_text, uint64_t **out_dev_word_off, uint64_t *out_n_words, int64_t *out_bad_pos) {
    if (!c || !out_dev_text || !out_dev_word_off || !out_n_words || !out_bad_pos) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    *out_dev_text = nullptr; *out_dev_word_off = nullptr; *out_n_words = 0; *out_bad_pos = -1;
    if (n_bytes && !text) return fail(c, YABPE_E_INVALID, "text is NULL");
    if (n_special > 255) return fail(c, YABPE_E_CAPACITY, "at most 255 special tokens in the pre-tokeniser");
    if (n_special && (!special_bytes || !special_off)) return fail(c, YABPE_E_INVALID, "special token arrays are NULL");
    uint32_t max_len = 0;
    for (uint32_t s = 0; s < n_special; ++s) {
        if (special_off[s + 1] <= special_off[s])
            return fail(c, YABPE_E_INVALID, "special token %u is empty (it would match at every position)", s);
        max_len = std::max(max_len, special_off[s + 1] - special_off[s]);
    }
    std::vector<unsigned long long> chunks;
    if (chunk_off && n_chunks) {
        for (uint32_t k = 0; k < n_chunks; ++k) {
            if (chunk_off[k] > n_bytes || (k && chunk_off[k] < chunk_off[k - 1])) return fail(c, YABPE_E_INVALID, "chunk offsets must ascend inside the text");
            chunks.push_back(chunk_off[k]);
        }
        if (chunks[0] != 0) return fail(c, YABPE_E_INVALID, "the first chunk must start at 0");
    } else {
        chunks.push_back(0);
    }
    if (!c->pt_cls) {  // expand the generated runs into one byte per code point
        std::vector<uint8_t> cls(0x110000, PT_O);
        for (unsigned r = 0; r < YB_UNICODE_CLASS_NRUNS; ++r) {
            const unsigned lo = YB_UNICODE_CLASS_RUNS[r][0];
            const unsigned hi = r + 1 < YB_UNICODE_CLASS_NRUNS ? YB_UNICODE_CLASS_RUNS[r + 1][0] : 0x110000;
            memset(cls.data() + lo, (int)YB_UNICODE_CLASS_RUNS[r][1], hi - lo);
        }
        TRY(dmalloc(c, &c->pt_cls, cls.size()));
        HIPCHK(c, hipMemcpy(c->pt_cls, cls.data(), cls.size(), hipMemcpyHostToDevice));
    }
    // inputs on the device
    const uint8_t *d_text = text;
    void *own_text = nullptr;
    if (n_bytes && !is_device_ptr(text)) {
        HIPCHK(c, hipMalloc(&own_text, n_bytes));
        if (hipMemcpy(own_text, text, n_bytes, hipMemcpyHostToDevice) != hipSuccess) { dfree(own_text); return fail(c, YABPE_E_HIP, "staging the text failed"); }
        d_text = (const uint8_t *)own_text;
    }
    unsigned long long *d_chunks = nullptr;
    uint8_t *d_spb = nullptr;
    uint32_t *d_spo = nullptr;
    auto drop = [&]() { dfree(d_chunks); dfree(d_spb); dfree(d_spo); };
    if (hipMalloc((void **)&d_chunks, chunks.size() * 8) != hipSuccess ||
        hipMemcpy(d_chunks, chunks.data(), chunks.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { drop(); dfree(own_text); return fail(c, YABPE_E_HIP, "chunk table"); }
    PtSpecials sp{nullptr, nullptr, n_special, max_len};
    if (n_special) {
        const uint32_t tot = special_off[n_special];
        if (hipMalloc((void **)&d_spb, tot) != hipSuccess || hipMalloc((void **)&d_spo, (n_special + 1) * 4) != hipSuccess ||
            hipMemcpy(d_spb, special_bytes, tot, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_spo, special_off, (n_special + 1) * 4, hipMemcpyHostToDevice) != hipSuccess) { drop(); dfree(own_text); return fail(c, YABPE_E_HIP, "special token table"); }
        sp.bytes = d_spb;
        sp.off = d_spo;
    }
    PretokOut po{};
    const int r = pretokenize(c->stream, d_text, n_bytes, d_chunks, (uint32_t)chunks.size(), c->pt_cls, sp, &po);
    drop();
    if (r != 0) { dfree(own_text); return fail(c, YABPE_E_HIP, "pre-tokeniser failed: %s", hipGetErrorString(hipGetLastError())); }
    if (po.bad_pos >= 0) {
        dfree(own_text);
        *out_bad_pos = po.bad_pos;
        return fail(c, YABPE_E_UTF8, "invalid UTF-8 at byte %lld", po.bad_pos);
    }
    if (own_text) c->pretok_bufs.push_back(own_text);
    c->pretok_bufs.push_back(po.off);
    *out_dev_text = d_text;
    *out_dev_word_off = (uint64_t *)po.off;
    *out_n_words = po.n_words;
    return YABPE_OK;
}

int yabpe_pretokenize_free(yabpe_ctx *c) {
    if (!c) return YABPE_E_INVALID;
    for (void *p : c->pretok_bufs) dfree(p);
    c->pretok_bufs.clear();
    return YABPE_OK;
}

// ---------------------------------------------------------------------------------------------- multi-GPU (see yabpe_comm.h)
int yabpe_comm_unique_id(uint8_t out_id[128]) {
    if (!out_id) return YABPE_E_INVALID;
    Rccl *R = rccl();
    if (!R) return fail(nullptr, YABPE_E_COMM, "librccl.so could not be loaded");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = R->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, YABPE_E_COMM, "ncclGetUniqueId failed");
    memcpy(out_id, &id, 128);
    return YABPE_OK;
}

static int comm_attach(yabpe_ctx *c, int rank, int n_ranks) {
    if (c->have_words) return fail(c, YABPE_E_INVALID, "attach the communicator before yabpe_load_words");
    if (c->comm || c->ag_fn) return fail(c, YABPE_E_INVALID, "communicator already attached");
    c->rank = rank;
    c->n_ranks = n_ranks;
    return 0;
}
static int comm_finish(yabpe_ctx *c) {
    TRY(dmalloc(c, &c->xsmall, 4 * (uint64_t)c->n_ranks));
    TRY(comm_buffers(c, (uint32_t)optv(c, "delta_cap", 16384)));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return YABPE_OK;
}

int yabpe_comm_init(yabpe_ctx *c, int rank, int n_ranks, const uint8_t unique_id[128]) {
    if (!c || !unique_id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    TRY(comm_attach(c, rank, n_ranks));
    if (n_ranks == 1 && !optv(c, "force_comm", 0)) return YABPE_OK;
    c->multi = true;
    Rccl *R = rccl();
    if (!R) return fail(c, YABPE_E_COMM, "librccl.so could not be loaded");
    ncclUniqueId id;
    memcpy(&id, unique_id, 128);
    NCCLCHK(c, R->CommInitRank(&c->comm, n_ranks, id, rank));
    return comm_finish(c);
}

int yabpe_comm_enable_p2p(yabpe_ctx *c) {
    if (!c) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->multi) return YABPE_OK;  // one rank: nothing to exchange
    if (c->have_words) return fail(c, YABPE_E_INVALID, "enable the peer-to-peer exchange before yabpe_load_words");
    if (!c->p2p_e0) {
        HIPCHK(c, hipEventCreate(&c->p2p_e0));
        HIPCHK(c, hipEventCreate(&c->p2p_e1));
    }
    c->p2p = true;
    return p2p_setup(c);
}

int yabpe_comm_init_custom(yabpe_ctx *c, int rank, int n_ranks, yabpe_allgather_fn fn, void *user) {
    if (!c || !fn || n_ranks < 1 || rank < 0 || rank >= n_ranks) return YABPE_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    TRY(comm_attach(c, rank, n_ranks));
    if (n_ranks == 1 && !optv(c, "force_comm", 0)) return YABPE_OK;
    c->multi = true;
    c->ag_fn = fn;
    c->ag_user = user;
    return comm_finish(c);
}

}  // extern "C"
