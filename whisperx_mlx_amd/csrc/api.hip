// C ABI of libwxhip.so (include/wxhip.h): context, weight binding, workspace and the
// host-side orchestration of the gfx950 kernels.  No torch types, no CPU fallback.
#include "../../include/wxhip.h"
#include "../../include/wxhip_test.h"
#include "kernels.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
constexpr int N_FRAMES = 3000, N_SAMPLES = 480000, N_BIN = 201, T_PAD_ALIGN = 64;
inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

struct EncLayer {
    const h16 *ln1g, *ln1b, *qkw, *qkb, *vw, *vb, *ow, *ob, *ln2g, *ln2b, *fc1w, *fc1b, *fc2w, *fc2b;
};
struct DecLayer {
    const h16 *ln1g, *ln1b, *qkvw, *qkvb, *ow, *ob, *ln2g, *ln2b, *cqw, *cqb, *ckvw, *ckvb, *cow, *cob, *ln3g, *ln3b,
        *fc1w, *fc1b, *fc2w, *fc2b;
    // optional int8 copies of the six decode GEMV weights (bytes q + 128, one fp32 scale per output row)
    const unsigned char *qkvq = nullptr, *oq = nullptr, *cqq = nullptr, *coq = nullptr, *fc1q = nullptr, *fc2q = nullptr;
    const float *qkvs = nullptr, *os = nullptr, *cqs = nullptr, *cos = nullptr, *fc1s = nullptr, *fc2s = nullptr;
    // the six decode GEMV weights again in the tile-blocked layout of SkinnyArgs::w_blocked (fp16, or the int8 bytes when
    // those are bound): library-owned copies made by wx_finalize, what the decode step's GEMV launches stream
    void *qkv_blk = nullptr, *o_blk = nullptr, *cq_blk = nullptr, *co_blk = nullptr, *fc1_blk = nullptr, *fc2_blk = nullptr;
};
// captured decode steps by launch signature (buffers, batch rows, options): a scheduler that alternates full and ragged
// passes replays each shape's graph instead of re-capturing.  Bounded: when full, the entry replayed longest ago is
// evicted (a job's shapes -- up to 8 row counts x prompt/sample steps -- stay, stale ones of earlier jobs go), so a
// cache that fills in the middle of a job costs one capture, not a re-capture of everything the job uses.
struct GraphCache {
    struct Entry { hipGraphExec_t exec; unsigned long long last_use; };
    std::unordered_map<std::string, Entry> exec;
    static constexpr size_t kMax = 48;
    unsigned long long clock = 0;
    int generation = 0;           // bumped whenever captured shapes are dropped: hosts that remember which shapes are captured compare it (wx_graph_generation)
    void clear() {
        for (auto& kv : exec) hipGraphExecDestroy(kv.second.exec);
        exec.clear();
        ++generation;
    }
    void evict_oldest() {
        auto victim = exec.end();
        for (auto it = exec.begin(); it != exec.end(); ++it)
            if (victim == exec.end() || it->second.last_use < victim->second.last_use) victim = it;
        if (victim != exec.end()) {
            hipGraphExecDestroy(victim->second.exec);
            exec.erase(victim);
            ++generation;
        }
    }
};
}  // namespace

// The tile-blocked copies of the decode GEMV weights (wx_finalize) are shared by every context of the process that binds
// the same weight tensor: several contexts of one model (the backend's passes in flight) stream ONE copy (a
// copy per context was 1.5 GB fp16 each for large-v3, and concurrent passes could never meet in the L2 / Infinity Cache).
// Keyed by (device, source pointer, bytes); released with the last context that uses it.
namespace {
struct PackedKey {
    int device; const void* src; size_t bytes;
    bool operator==(const PackedKey& o) const { return device == o.device && src == o.src && bytes == o.bytes; }
};
struct PackedKeyHash {
    size_t operator()(const PackedKey& k) const { return std::hash<const void*>()(k.src) ^ (k.bytes * 1315423911u) ^ (size_t)k.device; }
};
struct PackedEntry { void* buf = nullptr; int refs = 0; };
std::mutex g_packed_mu;
std::unordered_map<PackedKey, PackedEntry, PackedKeyHash> g_packed;
}  // namespace

// The encoder's activations (conv stem, residual stream, Q|K, V^T, attention output, FC1 output: 6.0 GB at 128 rows of
// large-v3) exist ONCE per (device, model geometry), for all contexts of the process (round 5): the contexts of a backend
// never run two encoders at the same time by more than launch interleaving -- a pass encodes once and then decodes for a
// hundred times as long -- so a second copy per context bought nothing.  An encoder takes the buffers for the duration of
// its launches: wx_encode holds `mu` while it enqueues, makes its stream wait for `done` (the previous user's last kernel)
// and records `done` behind its own last kernel.  Grown when a context with more rows attaches (device drained first).
namespace {
struct EncWs {
    int device = 0, n_mels = 0, da = 0, T = 0;
    size_t rows = 0;
    int refs = 0;
    h16 *mel_pad = nullptr, *c1 = nullptr, *x = nullptr, *h = nullptr, *qk = nullptr, *vt = nullptr, *a = nullptr, *f = nullptr;
    hipEvent_t done = nullptr;
    bool used = false;
    std::mutex mu;
};
std::mutex g_encws_mu;
std::vector<EncWs*> g_encws;
}  // namespace

struct wx_ctx {
    int device = 0;
    std::string err;
    wx_model_dims d{};
    int maxB = 0;
    int n_cu = 256;
    bool finalized = false;
    std::unordered_map<std::string, std::pair<const void*, size_t>> w;
    const h16 *conv1w = nullptr, *conv1b = nullptr, *conv2w = nullptr, *conv2b = nullptr, *encpos = nullptr,
              *lnpostg = nullptr, *lnpostb = nullptr, *emb = nullptr, *decpos = nullptr, *declng = nullptr,
              *declnb = nullptr;
    std::vector<EncLayer> enc;
    std::vector<DecLayer> dec;
    std::vector<void*> allocs;
    // log-mel tables
    float *filters = nullptr, *twiddle = nullptr, *window = nullptr, *logspec = nullptr;
    int *filt_lo = nullptr, *filt_len = nullptr;
    unsigned* chunk_max = nullptr;
    // encoder workspace: aliases of the shared one (EncWs), refreshed under its lock by every wx_encode
    EncWs* ews = nullptr;
    int kv_ctx = 0;                // positions the self-attention KV cache holds per sequence (prompt <= 8 + n_text_ctx / 2 sampled)
    int Tpad = 0;
    h16 *mel_pad = nullptr, *c1 = nullptr, *x = nullptr, *h = nullptr, *qk = nullptr, *vt = nullptr, *a = nullptr,
        *f = nullptr;
    // decoder workspace
    h16 *ckv = nullptr, *kc = nullptr, *vc = nullptr, *xd = nullptr, *xn = nullptr, *qkv = nullptr, *att = nullptr,
        *cq = nullptr, *f1 = nullptr;
    float *partA = nullptr, *partQ = nullptr;
    float *logits = nullptr, *part = nullptr, *align_qk = nullptr;
    int vocab_ld = 0;
    int *d_pos = nullptr, *d_row = nullptr, *d_done = nullptr, *tok_tmp = nullptr;
    int* d_nact = nullptr;         // rows of the running decode that are chunks (the rest is padding, wx_decode_opts.n_active)
    unsigned* tickets = nullptr;   // [maxB][H] cross-attention split merge counters (self-resetting)
    unsigned* samp_ticket = nullptr;   // sampler tail: blocks finished this step (self-resetting)
    unsigned* samp_row_ticket = nullptr;   // [maxB] sampler row split: blocks of the row finished (self-resetting)
    float* samp_part = nullptr;        // [maxB][4][8] sampler row split records
    unsigned long long* gran = nullptr;   // [maxB][H][4][66] tagged {f32, tag} partial words of the cross-attention splits
    unsigned long long* gran_q = nullptr; // [RB][d/2] tagged {2 x fp16, tag}: cross-attention query, GEMV role -> attention role (declayer.hip)
    size_t gran_q_words = 0;
    unsigned* d_epoch = nullptr;   // device copy of `epoch` (part of the granule tag)
    int* d_err = nullptr;          // raised by a kernel that gave up waiting (checked by wx_device_status)
    hipEvent_t ahead_ev[2] = {nullptr, nullptr};    // wx_tuning.max_steps_ahead
    bool w_blocked = false;        // the decode step streams the library's tile-blocked copies of the GEMV weights (DecLayer::*_blk)
    struct PackedSlot { void* buf = nullptr; size_t bytes = 0; const void* src = nullptr; };
    std::unordered_map<std::string, PackedSlot> wpacked;   // the process-wide copies this context holds a reference to, by layer.weight
    int* d_selfq = nullptr;        // fused decode launch: attention blocks that computed their query themselves (wx_decode_stats)
    unsigned long long* prof = nullptr;   // launch timer of the fused decode launch: {start note, sum of durations (10 ns ticks), launches} (wx_launch_profile)
    unsigned epoch = 0;
    int merge_mode = 2;            // 2 tagged granules, 1 tickets, 0 separate combine kernel
    bool any_q8 = false;           // some decode GEMV weight is bound as int8
    std::atomic_flag busy = ATOMIC_FLAG_INIT;   // a context is single-threaded: entry points refuse concurrent entry
    int fused_combine = 1;
    int* cap_slot = nullptr;  // device [L][H]
    int n_cap = 0, cap_rows = 0;
    int heads_version = 0;   // bumped by wx_set_alignment_heads: captured steps bake the capture slots' buffer in
    // dtw workspace
    float *dtw_work = nullptr, *dtw_work2 = nullptr;
    unsigned char* dtw_trace = nullptr;
    int* dtw_rowmap = nullptr;
    // ctc scratch (grown on demand)
    void* ctc_scratch = nullptr;
    size_t ctc_scratch_bytes = 0;
    GraphCache graphs;
    int tn_small = 8, tn_cq = 8;   // output columns per block of the N = d decode GEMVs (tuned on MI355X)
    h16* hook_ln_buf = nullptr;    // test hooks (wx_skinny_f16 / _ex with a LayerNorm and more than 16 rows): SkinnyArgs::ln_scratch, grown on demand
    size_t hook_ln_elems = 0;
    int enc_cap = 0;               // wx_set_encoder_cap: the encoder's GEMM / attention launches take at most this many blocks (0: all CUs)
};

static int wx_fail(wx_ctx* ctx, hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    if (ctx) ctx->err = buf;
    return -1;
}
static int wx_err(wx_ctx* ctx, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return -2;
}

template <typename T>
static hipError_t ws_alloc(wx_ctx* ctx, T** p, size_t n_elems) {
    void* q = nullptr;
    const size_t bytes = n_elems * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    e = hipMemset(q, 0, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    ctx->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return hipSuccess;
}

// SkinnyArgs::ln_scratch for the test hooks: round_up(M, 16) x K halves, kept with the context
static hipError_t hook_ln_scratch(wx_ctx* ctx, SkinnyArgs& a, hipStream_t s) {
    if (!a.ln_g || a.M <= 16) return hipSuccess;
    const size_t need = (size_t)((a.M + 15) / 16 * 16) * a.K;
    if (ctx->hook_ln_elems < need) {
        (void)hipStreamSynchronize(s);
        if (ctx->hook_ln_buf) (void)hipFree(ctx->hook_ln_buf);
        ctx->hook_ln_buf = nullptr;
        ctx->hook_ln_elems = 0;
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, need * sizeof(h16));
        if (e != hipSuccess) return e;
        ctx->hook_ln_buf = reinterpret_cast<h16*>(q);
        ctx->hook_ln_elems = need;
    }
    a.ln_scratch = ctx->hook_ln_buf;
    return hipSuccess;
}

static void encws_free_buffers(EncWs* W) {
    for (h16** q : {&W->mel_pad, &W->c1, &W->x, &W->h, &W->qk, &W->vt, &W->a, &W->f}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    W->rows = 0;
}
static hipError_t encws_alloc_buffers(EncWs* W, size_t B, size_t Tpad) {
    const size_t T = W->T, da = W->da;
    struct { h16** p; size_t n; } want[8] = {
        {&W->mel_pad, B * (N_FRAMES + 2) * W->n_mels}, {&W->c1, B * (N_FRAMES + 2) * da}, {&W->x, B * T * da},
        {&W->h, B * T * da}, {&W->qk, B * T * 2 * da}, {&W->vt, B * da * Tpad}, {&W->a, B * T * da}, {&W->f, B * T * 4 * da}};
    for (auto& w : want) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, w.n * sizeof(h16));
        if (e == hipSuccess) e = hipMemset(q, 0, w.n * sizeof(h16));     // the zero rows either side of a chunk (mel_pad, c1) are never written again
        if (e != hipSuccess) {
            if (q) (void)hipFree(q);
            encws_free_buffers(W);
            return e;
        }
        *w.p = reinterpret_cast<h16*>(q);
    }
    W->rows = B;
    return hipSuccess;
}
static void encws_alias(wx_ctx* ctx) {
    EncWs* W = ctx->ews;
    ctx->mel_pad = W->mel_pad; ctx->c1 = W->c1; ctx->x = W->x; ctx->h = W->h;
    ctx->qk = W->qk; ctx->vt = W->vt; ctx->a = W->a; ctx->f = W->f;
}
// the shared encoder workspace of this context's device and model geometry, with room for its rows
static hipError_t encws_attach(wx_ctx* ctx) {
    const wx_model_dims& D = ctx->d;
    std::lock_guard<std::mutex> lock(g_encws_mu);
    EncWs* W = nullptr;
    for (EncWs* c : g_encws)
        if (c->device == ctx->device && c->n_mels == D.n_mels && c->da == D.n_audio_state && c->T == D.n_audio_ctx) W = c;
    if (!W) {
        W = new EncWs();
        W->device = ctx->device; W->n_mels = D.n_mels; W->da = D.n_audio_state; W->T = D.n_audio_ctx;
        hipError_t e = hipEventCreateWithFlags(&W->done, hipEventDisableTiming);
        if (e != hipSuccess) { delete W; return e; }
        g_encws.push_back(W);
    }
    std::lock_guard<std::mutex> wl(W->mu);
    if (W->rows < (size_t)ctx->maxB) {
        // grow: nobody may be using the old buffers (other contexts enqueue under W->mu, which is held; what they
        // enqueued earlier is drained here).  They pick the new addresses up at their next wx_encode.
        (void)hipDeviceSynchronize();
        encws_free_buffers(W);
        hipError_t e = encws_alloc_buffers(W, ctx->maxB, ctx->Tpad);
        if (e != hipSuccess) {
            if (W->refs == 0) {
                g_encws.erase(std::find(g_encws.begin(), g_encws.end(), W));
                (void)hipEventDestroy(W->done);
                // (W->mu is held by `wl`: leak the small struct rather than destroy a locked mutex)
            }
            return e;
        }
    }
    ++W->refs;
    ctx->ews = W;
    encws_alias(ctx);
    return hipSuccess;
}
static void encws_release(wx_ctx* ctx) {
    EncWs* W = ctx->ews;
    if (!W) return;
    ctx->ews = nullptr;
    std::lock_guard<std::mutex> lock(g_encws_mu);
    bool last;
    {
        std::lock_guard<std::mutex> wl(W->mu);
        last = --W->refs == 0;
        if (last) {
            encws_free_buffers(W);
            (void)hipEventDestroy(W->done);
        }
    }
    if (last) {
        g_encws.erase(std::find(g_encws.begin(), g_encws.end(), W));
        delete W;
    }
}

// One context = one launcher thread (its workspace, graphs and device-side state are not shareable).  The entry
// points that enqueue work hold this guard; a second thread entering meanwhile gets an error, not a corrupted workspace.
struct CtxGuard {
    wx_ctx* c;
    bool ok;
    explicit CtxGuard(wx_ctx* ctx) : c(ctx), ok(ctx && !ctx->busy.test_and_set(std::memory_order_acquire)) {}
    ~CtxGuard() {
        if (c && ok) c->busy.clear(std::memory_order_release);
    }
};
#define WX_ENTER(ctx)                                                                                              \
    CtxGuard _guard(ctx);                                                                                          \
    if ((ctx) && !_guard.ok) return wx_err(ctx, "context is in use by another thread (one context per launcher thread)")

extern "C" {

int wx_create(int device_id, const wx_model_dims* dims, int max_batch, wx_ctx** out) {
    if (!dims || !out || max_batch < 1 || max_batch > 128) return -2;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device_id) return -3;   // no GPU: fail loudly
    wx_ctx* ctx = new wx_ctx();
    ctx->device = device_id;
    ctx->d = *dims;
    ctx->maxB = max_batch;
    ctx->n_cu = 256;
    {
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cu > 0) ctx->n_cu = cu;
    }
    *out = ctx;
    if (hipSetDevice(device_id) != hipSuccess) return -3;
    if (dims->n_audio_state % 64 || dims->n_text_state % 64 || dims->n_audio_state / dims->n_audio_head != 64 ||
        dims->n_text_state / dims->n_text_head != 64 || dims->n_mels % 8 || dims->n_audio_ctx != 1500)
        return wx_err(ctx, "unsupported model dims (need d_head 64, n_mels % 8 == 0, n_audio_ctx 1500)");
    return 0;
}

void wx_destroy(wx_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    ctx->graphs.clear();
    for (hipEvent_t e : ctx->ahead_ev)
        if (e) (void)hipEventDestroy(e);
    for (void* p : ctx->allocs) hipFree(p);
    if (ctx->hook_ln_buf) (void)hipFree(ctx->hook_ln_buf);
    encws_release(ctx);
    {
        std::lock_guard<std::mutex> lock(g_packed_mu);
        for (auto& kv : ctx->wpacked) {
            auto it = g_packed.find(PackedKey{ctx->device, kv.second.src, kv.second.bytes});
            if (it != g_packed.end() && --it->second.refs == 0) { hipFree(it->second.buf); g_packed.erase(it); }
        }
        ctx->wpacked.clear();
    }
    if (ctx->ctc_scratch) hipFree(ctx->ctc_scratch);
    delete ctx;
    // a context that is destroyed because its workspace did not fit (wx_finalize: out of memory) must not leave the failed
    // hipMalloc behind as the thread's last error: the host's next, unrelated call (a torch allocation) would report it
    (void)hipGetLastError();
}

const char* wx_last_error(wx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int wx_bind_weight(wx_ctx* ctx, const char* name, const void* dptr, size_t nbytes) {
    if (!ctx || !name || !dptr) return -2;
    ctx->w[name] = {dptr, nbytes};
    return 0;
}

int wx_set_mel_filters(wx_ctx* ctx, const float* filters_host, int n_mels) {
    if (!ctx || n_mels != ctx->d.n_mels) return wx_err(ctx, "wx_set_mel_filters: n_mels mismatch");
    hipSetDevice(ctx->device);
    std::vector<int> lo(n_mels), len(n_mels);
    for (int m = 0; m < n_mels; ++m) {
        int a = 0, b = N_BIN;
        while (a < N_BIN && filters_host[m * N_BIN + a] == 0.f) ++a;
        while (b > a && filters_host[m * N_BIN + b - 1] == 0.f) --b;
        lo[m] = (a == N_BIN) ? 0 : a;
        len[m] = (a == N_BIN) ? 0 : b - a;
    }
    std::vector<float> tw(800), win(400);
    for (int i = 0; i < 400; ++i) {
        const double ang = 2.0 * M_PI * (double)i / 400.0;
        tw[2 * i] = (float)cos(ang);
        tw[2 * i + 1] = (float)sin(ang);
        win[i] = (float)(0.5 - 0.5 * cos(ang));   // torch.hann_window(400), periodic
    }
    if (!ctx->filters) {
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->filters, (size_t)n_mels * N_BIN));
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->twiddle, 800));
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->window, 400));
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->filt_lo, n_mels));
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->filt_len, n_mels));
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->logspec, (size_t)ctx->maxB * N_FRAMES * n_mels));
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->chunk_max, ctx->maxB));
    }
    WX_CHECK_HIP(hipMemcpy(ctx->filters, filters_host, sizeof(float) * n_mels * N_BIN, hipMemcpyHostToDevice));
    WX_CHECK_HIP(hipMemcpy(ctx->twiddle, tw.data(), sizeof(float) * 800, hipMemcpyHostToDevice));
    WX_CHECK_HIP(hipMemcpy(ctx->window, win.data(), sizeof(float) * 400, hipMemcpyHostToDevice));
    WX_CHECK_HIP(hipMemcpy(ctx->filt_lo, lo.data(), sizeof(int) * n_mels, hipMemcpyHostToDevice));
    WX_CHECK_HIP(hipMemcpy(ctx->filt_len, len.data(), sizeof(int) * n_mels, hipMemcpyHostToDevice));
    return 0;
}

static const h16* getw(wx_ctx* ctx, const std::string& name, size_t elems, bool& ok) {
    auto it = ctx->w.find(name);
    if (it == ctx->w.end()) {
        if (ok) ctx->err = "missing weight: " + name;
        ok = false;
        return nullptr;
    }
    if (it->second.second != elems * 2) {
        if (ok) ctx->err = "weight " + name + " has " + std::to_string(it->second.second) + " bytes, expected " +
                           std::to_string(elems * 2);
        ok = false;
        return nullptr;
    }
    return reinterpret_cast<const h16*>(it->second.first);
}

// A decode GEMV weight is either fp16 ("<name>.w") or int8 ("<name>.wq" bytes q + 128 and "<name>.ws" fp32 row scales);
// when the int8 pair is bound it is the one the decode step uses.
static const h16* getw_q8(wx_ctx* ctx, const std::string& base, size_t n_rows, size_t k, const unsigned char** q,
                          const float** sc, bool& ok) {
    auto iq = ctx->w.find(base + ".wq"), is = ctx->w.find(base + ".ws");
    if (iq == ctx->w.end() && is == ctx->w.end()) return getw(ctx, base + ".w", n_rows * k, ok);
    if (iq == ctx->w.end() || is == ctx->w.end() || iq->second.second != n_rows * k || is->second.second != n_rows * 4) {
        if (ok) ctx->err = "int8 weight " + base + ": need .wq (" + std::to_string(n_rows * k) + " bytes) and .ws (" +
                           std::to_string(n_rows * 4) + " bytes)";
        ok = false;
        return nullptr;
    }
    *q = reinterpret_cast<const unsigned char*>(iq->second.first);
    *sc = reinterpret_cast<const float*>(is->second.first);
    ctx->any_q8 = true;
    auto iw = ctx->w.find(base + ".w");
    return iw == ctx->w.end() ? nullptr : reinterpret_cast<const h16*>(iw->second.first);
}

int wx_finalize(wx_ctx* ctx) {
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    const wx_model_dims& D = ctx->d;
    const size_t da = D.n_audio_state, dt = D.n_text_state;
    bool ok = true;
    ctx->conv1w = getw(ctx, "enc.conv1.w", da * 3 * D.n_mels, ok);
    ctx->conv1b = getw(ctx, "enc.conv1.b", da, ok);
    ctx->conv2w = getw(ctx, "enc.conv2.w", da * 3 * da, ok);
    ctx->conv2b = getw(ctx, "enc.conv2.b", da, ok);
    ctx->encpos = getw(ctx, "enc.pos", (size_t)D.n_audio_ctx * da, ok);
    ctx->lnpostg = getw(ctx, "enc.lnpost.g", da, ok);
    ctx->lnpostb = getw(ctx, "enc.lnpost.b", da, ok);
    ctx->enc.resize(D.n_audio_layer);
    for (int i = 0; i < D.n_audio_layer; ++i) {
        const std::string p = "enc." + std::to_string(i) + ".";
        EncLayer& L = ctx->enc[i];
        L.ln1g = getw(ctx, p + "ln1.g", da, ok);   L.ln1b = getw(ctx, p + "ln1.b", da, ok);
        L.qkw = getw(ctx, p + "qk.w", 2 * da * da, ok);  L.qkb = getw(ctx, p + "qk.b", 2 * da, ok);
        L.vw = getw(ctx, p + "v.w", da * da, ok);  L.vb = getw(ctx, p + "v.b", da, ok);
        L.ow = getw(ctx, p + "o.w", da * da, ok);  L.ob = getw(ctx, p + "o.b", da, ok);
        L.ln2g = getw(ctx, p + "ln2.g", da, ok);   L.ln2b = getw(ctx, p + "ln2.b", da, ok);
        L.fc1w = getw(ctx, p + "fc1.w", 4 * da * da, ok);  L.fc1b = getw(ctx, p + "fc1.b", 4 * da, ok);
        L.fc2w = getw(ctx, p + "fc2.w", 4 * da * da, ok);  L.fc2b = getw(ctx, p + "fc2.b", da, ok);
    }
    ctx->emb = getw(ctx, "dec.emb", (size_t)D.n_vocab * dt, ok);
    ctx->decpos = getw(ctx, "dec.pos", (size_t)D.n_text_ctx * dt, ok);
    ctx->declng = getw(ctx, "dec.ln.g", dt, ok);
    ctx->declnb = getw(ctx, "dec.ln.b", dt, ok);
    ctx->dec.resize(D.n_text_layer);
    for (int i = 0; i < D.n_text_layer; ++i) {
        const std::string p = "dec." + std::to_string(i) + ".";
        DecLayer& L = ctx->dec[i];
        L.ln1g = getw(ctx, p + "ln1.g", dt, ok);   L.ln1b = getw(ctx, p + "ln1.b", dt, ok);
        L.qkvw = getw_q8(ctx, p + "qkv", 3 * dt, dt, &L.qkvq, &L.qkvs, ok);  L.qkvb = getw(ctx, p + "qkv.b", 3 * dt, ok);
        L.ow = getw_q8(ctx, p + "o", dt, dt, &L.oq, &L.os, ok);  L.ob = getw(ctx, p + "o.b", dt, ok);
        L.ln2g = getw(ctx, p + "ln2.g", dt, ok);   L.ln2b = getw(ctx, p + "ln2.b", dt, ok);
        L.cqw = getw_q8(ctx, p + "cq", dt, dt, &L.cqq, &L.cqs, ok);  L.cqb = getw(ctx, p + "cq.b", dt, ok);
        L.ckvw = getw(ctx, p + "ckv.w", 2 * dt * da, ok);  L.ckvb = getw(ctx, p + "ckv.b", 2 * dt, ok);
        L.cow = getw_q8(ctx, p + "co", dt, dt, &L.coq, &L.cos, ok);  L.cob = getw(ctx, p + "co.b", dt, ok);
        L.ln3g = getw(ctx, p + "ln3.g", dt, ok);   L.ln3b = getw(ctx, p + "ln3.b", dt, ok);
        L.fc1w = getw_q8(ctx, p + "fc1", 4 * dt, dt, &L.fc1q, &L.fc1s, ok);  L.fc1b = getw(ctx, p + "fc1.b", 4 * dt, ok);
        L.fc2w = getw_q8(ctx, p + "fc2", dt, 4 * dt, &L.fc2q, &L.fc2s, ok);  L.fc2b = getw(ctx, p + "fc2.b", dt, ok);
    }
    if (!ok) return -2;
    // (re)pack the decode GEMV weights (a later wx_finalize call follows a re-bind: same shapes, new values)
    ctx->w_blocked = (dt % 32 == 0);
#ifdef LAB_NO_WBLOCK          // lab builds only: the decode step streams the row-major weights as bound
    ctx->w_blocked = false;
#endif
    if (ctx->w_blocked) {
        bool drained = false;
        for (int i = 0; i < D.n_text_layer; ++i) {
            DecLayer& L = ctx->dec[i];
            struct { const h16* w; const unsigned char* q; void** dst; size_t n, k; } ws[6] = {
                {L.qkvw, L.qkvq, &L.qkv_blk, 3 * dt, dt}, {L.ow, L.oq, &L.o_blk, dt, dt}, {L.cqw, L.cqq, &L.cq_blk, dt, dt},
                {L.cow, L.coq, &L.co_blk, dt, dt}, {L.fc1w, L.fc1q, &L.fc1_blk, 4 * dt, dt}, {L.fc2w, L.fc2q, &L.fc2_blk, dt, 4 * dt}};
            for (auto& e : ws) {
                const int eb = e.q ? 1 : 2;
                const void* src = e.q ? (const void*)e.q : (const void*)e.w;
                auto& slot = ctx->wpacked[std::to_string(i) + "." + std::to_string(&e - ws)];
                const size_t want_bytes = e.n * e.k * eb;
                if (slot.buf && slot.bytes != want_bytes)
                    // a re-bind that changes the storage of a decode GEMV weight (int8 <-> fp16) does not fit the copy made
                    // at the first wx_finalize; a context is built for one storage (include/wxhip.h)
                    return wx_err(ctx, "wx_finalize: decode GEMV weight " + std::to_string(i) + "." + std::to_string(&e - ws) +
                                           " was packed as " + std::to_string(slot.bytes) + " bytes and is now bound as " +
                                           std::to_string(want_bytes) + " (int8 <-> fp16): create a new context");
                bool fill = ctx->finalized;        // an explicit re-finalize: the caller has changed the values, pack again
                bool shared = false;               // other contexts stream this copy: the device is drained before it is rewritten
                {
                    std::lock_guard<std::mutex> lock(g_packed_mu);
                    if (slot.buf && slot.src != src) {        // re-bound to another tensor: let go of the old copy
                        auto old = g_packed.find(PackedKey{ctx->device, slot.src, slot.bytes});
                        if (old != g_packed.end() && --old->second.refs == 0) { hipFree(old->second.buf); g_packed.erase(old); }
                        slot = wx_ctx::PackedSlot{};
                    }
                    if (!slot.buf) {
                        PackedEntry& pe = g_packed[PackedKey{ctx->device, src, want_bytes}];
                        if (!pe.buf) {
                            hipError_t me = hipMalloc(&pe.buf, want_bytes);
                            if (me != hipSuccess) { g_packed.erase(PackedKey{ctx->device, src, want_bytes}); WX_CHECK_HIP(me); }
                        }
                        // A context that attaches to an existing copy packs it AGAIN from the source: the copy
                        // is found by the source's address, and the values behind that address may have changed since the
                        // first context packed them (an in-place update, or the allocator handing the address to another
                        // tensor of the same size) -- the new context's other weights are read from the source as it is
                        // now, so its GEMV copy must be too.  Same values give the same bytes; 1.5 GB of packing per context.
                        fill = true;
                        ++pe.refs;
                        slot.buf = pe.buf; slot.bytes = want_bytes; slot.src = src;
                    }
                    auto cur = g_packed.find(PackedKey{ctx->device, slot.src, slot.bytes});
                    shared = cur != g_packed.end() && cur->second.refs > 1;
                }
                if (fill && shared && !drained) {
                    // other contexts' decodes (and their captured hipGraphs) read the copy that is about to be rewritten:
                    // nothing of theirs may be in flight while the pack kernel runs
                    WX_CHECK_HIP(hipDeviceSynchronize());
                    drained = true;
                }
                if (fill) WX_CHECK_HIP(launch_pack_gemv_weight(src, slot.buf, (int)e.n, (int)e.k, eb, nullptr));
                *e.dst = slot.buf;
            }
        }
        WX_CHECK_HIP(hipDeviceSynchronize());
    }
    if (ctx->finalized) return 0;
    const size_t B = ctx->maxB, T = D.n_audio_ctx;
    const size_t RB = round_up(ctx->maxB, 16);   // decode row buffers: whole MFMA row tiles
    ctx->Tpad = round_up(D.n_audio_ctx, T_PAD_ALIGN);
    WX_CHECK_HIP(encws_attach(ctx));      // the encoder's activations: one set per process and model geometry (EncWs)
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->ckv, (size_t)D.n_text_layer * B * T * 2 * dt));
    // self-attention KV cache: a decode holds at most the prompt (<= 8 tokens: no previous-text conditioning on this path,
    // whisperx/backends/mlx_whisper.py:79) plus n_text_ctx / 2 sampled positions, never all n_text_ctx of them
    ctx->kv_ctx = (int)std::min<size_t>(D.n_text_ctx, round_up(8 + D.n_text_ctx / 2, 8));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->kc, (size_t)D.n_text_layer * B * ctx->kv_ctx * dt));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->vc, (size_t)D.n_text_layer * B * ctx->kv_ctx * dt));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->xd, RB * dt));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->xn, RB * dt));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->partA, 8 * 16 * dt));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->partQ, 8 * 16 * dt));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->qkv, RB * 3 * dt));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->att, RB * dt));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->cq, RB * dt));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->f1, RB * 4 * dt));
    ctx->vocab_ld = round_up(D.n_vocab, 16);
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->logits, RB * ctx->vocab_ld));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->part, B * D.n_text_head * 16 * 66));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->tickets, B * D.n_text_head));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->samp_ticket, 4));
    WX_CHECK_HIP(hipMemset(ctx->samp_ticket, 0, sizeof(unsigned)));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->samp_row_ticket, B));
    WX_CHECK_HIP(hipMemset(ctx->samp_row_ticket, 0, sizeof(unsigned) * B));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->samp_part, B * 4 * 8));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->gran, B * D.n_text_head * 4 * 66));
    WX_CHECK_HIP(hipMemset(ctx->gran, 0, sizeof(unsigned long long) * B * D.n_text_head * 4 * 66));
    ctx->gran_q_words = RB * (dt / 2);
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->gran_q, ctx->gran_q_words));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->d_epoch, 4));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->d_err, 4));
    WX_CHECK_HIP(hipMemset(ctx->d_err, 0, sizeof(int)));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->d_selfq, 4));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->prof, 4 + 12 + 160));     // + lab records (LAB_DUMP_Q8)
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->d_pos, 4));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->d_row, 4));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->d_done, RB));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->d_nact, 4));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->tok_tmp, RB * D.n_text_ctx));
    WX_CHECK_HIP(ws_alloc(ctx, &ctx->cap_slot, (size_t)D.n_text_layer * D.n_text_head));
    {
        std::vector<int> neg((size_t)D.n_text_layer * D.n_text_head, -1);
        WX_CHECK_HIP(hipMemcpy(ctx->cap_slot, neg.data(), neg.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    ctx->finalized = true;
    return 0;
}

int wx_set_alignment_heads(wx_ctx* ctx, const int* layer_head, int n_heads) {
    if (!ctx || !ctx->finalized) return wx_err(ctx, "wx_set_alignment_heads: finalize first");
    hipSetDevice(ctx->device);
    const wx_model_dims& D = ctx->d;
    std::vector<int> slot((size_t)D.n_text_layer * D.n_text_head, -1);
    for (int i = 0; i < n_heads; ++i) {
        const int l = layer_head[2 * i], h = layer_head[2 * i + 1];
        if (l < 0 || l >= D.n_text_layer || h < 0 || h >= D.n_text_head) return wx_err(ctx, "alignment head out of range");
        slot[(size_t)l * D.n_text_head + h] = i;
    }
    WX_CHECK_HIP(hipDeviceSynchronize());     // nothing may still read the capture buffers that are replaced below
    WX_CHECK_HIP(hipMemcpy(ctx->cap_slot, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice));
    ++ctx->heads_version;
    if (n_heads != ctx->n_cap || !ctx->align_qk) {
        // a different head count: the old score / DTW buffers are released (graphs that wrote to them are keyed on
        // align_qk and heads_version and are never replayed again)
        void* old[] = {ctx->align_qk, ctx->dtw_work, ctx->dtw_work2, ctx->dtw_trace, ctx->dtw_rowmap};
        for (void* q : old) {
            if (!q) continue;
            for (auto it = ctx->allocs.begin(); it != ctx->allocs.end(); ++it)
                if (*it == q) { ctx->allocs.erase(it); break; }
            hipFree(q);
        }
        ctx->align_qk = nullptr; ctx->dtw_work = nullptr; ctx->dtw_work2 = nullptr; ctx->dtw_trace = nullptr; ctx->dtw_rowmap = nullptr;
        ctx->graphs.clear();
        ctx->n_cap = n_heads;
        ctx->cap_rows = D.n_text_ctx / 2;
        const size_t B = ctx->maxB, R = ctx->cap_rows, T = D.n_audio_ctx;
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->align_qk, B * n_heads * R * T));
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->dtw_work, B * (R + 1) * T));
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->dtw_work2, B * n_heads * (R + 1) * T));
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->dtw_trace, B * (R + 2) * (T + 1)));
        WX_CHECK_HIP(ws_alloc(ctx, &ctx->dtw_rowmap, B * (R + 1)));
    }
    return 0;
}

// ------------------------------------------------------------------------------- log-mel
int wx_logmel(wx_ctx* ctx, const float* pcm, long pcm_stride, const int32_t* n_valid, int B, void* mel_f16,
              float* mel_f32, void* stream) {
    if (!ctx || !ctx->filters) return wx_err(ctx, "wx_logmel: call wx_set_mel_filters first");
    if (B < 1 || B > ctx->maxB) return wx_err(ctx, "wx_logmel: bad batch");
    WX_ENTER(ctx);
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    LogmelArgs a{pcm, pcm_stride, n_valid, ctx->filters, ctx->filt_lo, ctx->filt_len, ctx->twiddle, ctx->window,
                 ctx->logspec, ctx->chunk_max, B, ctx->d.n_mels};
    WX_CHECK_HIP(launch_logmel(a, s));
    // fp16 output is the plain (B,3000,n_mels) tensor: "padded" view with 0 leading rows
    h16* oh = reinterpret_cast<h16*>(mel_f16);
    WX_CHECK_HIP(launch_logmel_finalize(ctx->logspec, ctx->chunk_max, mel_f32, oh ? oh - ctx->d.n_mels : nullptr,
                                        ctx->d.n_mels, N_FRAMES, B, ctx->d.n_mels, s));
    return 0;
}

// ------------------------------------------------------------------------------- encoder
static GemmArgs gemm_rowmajor(const h16* W, int N, int K, const h16* A, long lda, int M, const h16* bias,
                              const h16* R, long ldr, h16* out, long ldo) {
    GemmArgs g{};
    g.X = W; g.ldx = K; g.strideX = 0; g.RX = N;
    g.Y = A; g.ldy = lda; g.strideY = 0; g.RY = M;
    g.K = K;
    g.bias = bias; g.strideBias = 0; g.bias_on_y = 0;
    g.R = R; g.ldr = ldr; g.strideR = 0;
    g.out = out; g.ldo = ldo; g.strideOut = 0;
    return g;
}

int wx_encode(wx_ctx* ctx, const void* mel_f16, int B, void* enc_f16, void* stream) {
    if (!ctx || !ctx->finalized) return wx_err(ctx, "wx_encode: not finalized");
    if (B < 1 || B > ctx->maxB) return wx_err(ctx, "wx_encode: bad batch");
    WX_ENTER(ctx);
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    const wx_model_dims& D = ctx->d;
    const int d = D.n_audio_state, T = D.n_audio_ctx, H = D.n_audio_head, nm = D.n_mels;
    // the shared encoder workspace is this call's from here to its last launch: enqueue under its lock, behind the
    // previous user's last kernel (any context, any stream), and leave the event behind our own last kernel whatever happens
    EncWs* W = ctx->ews;
    std::lock_guard<std::mutex> ews_lock(W->mu);
    encws_alias(ctx);
    if (W->used) WX_CHECK_HIP(hipStreamWaitEvent(s, W->done, 0));
    struct Leave {
        EncWs* W; hipStream_t s;
        ~Leave() { (void)hipEventRecord(W->done, s); W->used = true; }
    } leave{W, s};
    // mel -> padded conv-stem input (one zero row either side of every chunk)
    WX_CHECK_HIP(hipMemcpy2DAsync(ctx->mel_pad + nm, (size_t)(N_FRAMES + 2) * nm * 2, mel_f16, (size_t)N_FRAMES * nm * 2,
                                  (size_t)N_FRAMES * nm * 2, B, hipMemcpyDeviceToDevice, s));
    {   // conv1 (k3, s1, p1) + GELU as a GEMM over 3 consecutive padded rows
        GemmArgs g{};
        g.X = ctx->conv1w; g.ldx = 3 * nm; g.RX = d;
        g.Y = ctx->mel_pad; g.ldy = nm; g.strideY = (long)(N_FRAMES + 2) * nm; g.RY = N_FRAMES;
        g.K = 3 * nm;
        g.bias = ctx->conv1b;
        g.out = ctx->c1 + d; g.ldo = d; g.strideOut = (long)(N_FRAMES + 2) * d;
        WX_CHECK_HIP(launch_gemm_f16(g, B, true, s));
    }
    {   // conv2 (k3, s2, p1) + GELU, + positional embedding
        GemmArgs g{};
        g.X = ctx->conv2w; g.ldx = 3 * d; g.RX = d;
        g.Y = ctx->c1; g.ldy = 2 * d; g.strideY = (long)(N_FRAMES + 2) * d; g.RY = T;
        g.K = 3 * d;
        g.bias = ctx->conv2b;
        g.R = ctx->encpos; g.ldr = d; g.strideR = 0;
        g.out = ctx->x; g.ldo = d; g.strideOut = (long)T * d;
        WX_CHECK_HIP(launch_gemm_f16(g, B, true, s));
    }
    const int M = B * T;
    const int cap = ctx->enc_cap;       // > 0: GEMM blocks own their CU, so this many CUs at most (attention: two blocks per CU)
    auto capped = [&](GemmArgs g) { g.max_blocks = cap; return g; };
    for (int i = 0; i < D.n_audio_layer; ++i) {
        const EncLayer& L = ctx->enc[i];
        WX_CHECK_HIP(launch_layernorm(ctx->x, d, L.ln1g, L.ln1b, ctx->h, d, M, d, s));
        {   // Q | K projection; the Q half leaves the GEMM scaled for the attention kernel's v_exp_f32 (attention.hip)
            GemmArgs g = capped(gemm_rowmajor(L.qkw, 2 * d, d, ctx->h, d, M, L.qkb, nullptr, 0, ctx->qk, 2 * d));
            g.xscale = ATTN_QSCALE; g.xscale_cols = d;
            WX_CHECK_HIP(launch_gemm_f16(g, 1, false, s));
        }
        {   // V^T[b][feature][t]
            GemmArgs g{};
            g.X = ctx->h; g.ldx = d; g.strideX = (long)T * d; g.RX = T;
            g.Y = L.vw; g.ldy = d; g.strideY = 0; g.RY = d;
            g.K = d;
            g.bias = L.vb; g.bias_on_y = 1;
            g.out = ctx->vt; g.ldo = ctx->Tpad; g.strideOut = (long)d * ctx->Tpad;
            WX_CHECK_HIP(launch_gemm_f16(g, B, false, s));
        }
        AttnArgs at{ctx->qk, 2L * d, (long)T * 2 * d, ctx->qk + d, 2L * d, (long)T * 2 * d,
                    ctx->vt, (long)ctx->Tpad, (long)d * ctx->Tpad, ctx->a, (long)d, (long)T * d, nullptr, T, H, B};
        at.max_blocks = 2 * cap;
        at.q_prescaled = 1;
        WX_CHECK_HIP(launch_attention(at, s));
        WX_CHECK_HIP(launch_gemm_f16(capped(gemm_rowmajor(L.ow, d, d, ctx->a, d, M, L.ob, ctx->x, d, ctx->x, d)), 1, false, s));
        WX_CHECK_HIP(launch_layernorm(ctx->x, d, L.ln2g, L.ln2b, ctx->h, d, M, d, s));
        WX_CHECK_HIP(launch_gemm_f16(capped(gemm_rowmajor(L.fc1w, 4 * d, d, ctx->h, d, M, L.fc1b, nullptr, 0, ctx->f, 4 * d)), 1, true, s));
        WX_CHECK_HIP(launch_gemm_f16(capped(gemm_rowmajor(L.fc2w, d, 4 * d, ctx->f, 4 * d, M, L.fc2b, ctx->x, d, ctx->x, d)), 1, false, s));
    }
    WX_CHECK_HIP(launch_layernorm(ctx->x, d, ctx->lnpostg, ctx->lnpostb, reinterpret_cast<h16*>(enc_f16), d, M, d, s));
    return 0;
}

// ------------------------------------------------------------------------------- decoder
__global__ void init_decode_kernel(int* tokens, int tok_ld, int n_ctx_fill, const int* prompt, int n_prompt, int eot,
                                   float* sum_logprob, float* no_speech, int* d_pos, int* d_row, int* d_done, int n_active, int* d_nact) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < n_ctx_fill; i += blockDim.x) tokens[(long)b * tok_ld + i] = (i < n_prompt) ? prompt[i] : eot;
    if (threadIdx.x == 0) {
        if (sum_logprob) sum_logprob[b] = 0.f;
        if (no_speech) no_speech[b] = 0.f;
        d_done[b] = b >= n_active;      // padding rows sit out of the attention kernels from the first sampled position on
        if (b == 0) {
            *d_nact = n_active;
            *d_pos = 0;
            *d_row = -(n_prompt - 1);
        }
    }
}

__global__ void done_kernel(const int* tokens, int tok_ld, const int* d_pos, int eot, int* d_done) {
    // after advance: *d_pos is the index of the newest token
    const int b = threadIdx.x;
    d_done[b] = tokens[(long)b * tok_ld + *d_pos] == eot;
}

// a new (decode call) epoch for the granule tags of the cross-attention split merge
static hipError_t bump_epoch(wx_ctx* ctx, hipStream_t s) {
    ctx->epoch = (ctx->epoch + 1) & 0xFFFFu;
    if (ctx->epoch == 0) {
        // the 16-bit epoch wrapped: granules written 65536 decode calls ago would carry valid-looking tags
        ctx->epoch = 1;
        hipError_t e = hipMemsetAsync(ctx->gran, 0, sizeof(unsigned long long) * (size_t)ctx->maxB * ctx->d.n_text_head * 4 * 66, s);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(ctx->gran_q, 0, sizeof(unsigned long long) * ctx->gran_q_words, s);
        if (e != hipSuccess) return e;
    }
    return hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->d_epoch), (int)ctx->epoch, 1, s);
}

static int cross_kv(wx_ctx* ctx, const h16* enc, int B, hipStream_t s) {
    const wx_model_dims& D = ctx->d;
    const int da = D.n_audio_state, dt = D.n_text_state, T = D.n_audio_ctx;
    for (int l = 0; l < D.n_text_layer; ++l) {
        const DecLayer& L = ctx->dec[l];
        // layer layout [K|V][maxB][H][T][64]: every (batch, head) panel is one contiguous 187.5 KiB run,
        // which is what the decode cross-attention streams per block
        h16* out = ctx->ckv + (size_t)l * ctx->maxB * T * 2 * dt;
        GemmArgs g = gemm_rowmajor(L.ckvw, 2 * dt, da, enc, da, B * T, L.ckvb, nullptr, 0, out, 2 * dt);
        g.hs_T = T; g.hs_H = D.n_text_head; g.hs_d = dt; g.hs_part_stride = (long)ctx->maxB * T * dt;
        if (ctx->enc_cap < 0) g.max_blocks = -1;      // wx_set_encoder_cap(-1): the one-tile-per-block kernel here too (tests)
        WX_CHECK_HIP(launch_gemm_f16(g, 1, false, s));
    }
    return 0;
}

#ifdef LAB_DUMP_Q8
// lab (step variant 6): compares the query the fused launch's GEMV role published as granules with the query a skinny_kernel
// launch wrote to memory for the same step; on the first mismatch the epilogue operands both kernels dumped are kept
__global__ void dbg_compare_q_kernel(const unsigned long long* __restrict__ gq, const h16* __restrict__ cq, int rows, int d,
                                     const int* __restrict__ d_pos, int layer, unsigned long long* __restrict__ dbg,
                                     const float* __restrict__ dump, float* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // one thread per granule (2 values)
    if (i == 0) atomicAdd(dbg + 7, 1ull);                      // launches of this kernel
    if (i >= rows * (d >> 1)) return;
    const int r = i / (d >> 1), c2 = i - r * (d >> 1);
    const unsigned g = (unsigned)gq[i];
    const unsigned m = *reinterpret_cast<const unsigned*>(cq + (long)r * d + 2 * c2);
    if (g != m) {
        if (atomicAdd(dbg, 1ull) == 0) {
            dbg[1] = (unsigned long long)*d_pos; dbg[2] = layer; dbg[3] = r; dbg[4] = 2 * c2; dbg[5] = g; dbg[6] = m;
            for (int k = 0; k < 8; ++k) { keep[k] = dump[(0 * 16 + r) * 8 + k]; keep[8 + k] = dump[(1 * 16 + r) * 8 + k]; }
        }
    }
}
#endif

struct StepCfg {
    const int* tokens; int tok_ld;
    int B; bool sample; bool logits;
    float* logits_out; long logits_ld;
    int cross_split; bool capture;
    SampleArgs sa;
    int sample_begin;
    bool profile;  // time the fused launches on the device (wx_tuning.profile_launches)
    int fc2_tn;    // 0/8 or 16 output columns per block of the K = 4d GEMV
    int variant;   // 1 = LayerNorm-fused GEMVs (10 kernels/layer), 2 = split-K GEMVs + resln (12 kernels/layer)
    // true (wx_decode_greedy, variants 1 / 3): the input embedding of position p is produced at the END of step p - 1 --
    // by the sampler's tail, which also advances the position (sampling steps), or by advance + embed (prompt steps);
    // false: embed first, advance last (teacher-forced wx_decode_logits, variant 2)
    bool embed_at_end;
};

static int pick_ksplit(int N, int K) {
    // enough blocks to pull on HBM from every CU, each wave still within its register budget
    const int tiles = (N + 15) / 16, nks = K / 32;
    int ks = 1;
    if (tiles < 192) ks = (512 + tiles - 1) / tiles;
    if (ks > 8) ks = 8;
    if (ks > nks) ks = nks;
    while ((nks + ks * 4 - 1) / (ks * 4) > 10) ++ks;
    return ks < 1 ? 1 : ks;
}

static int decode_step_v2(wx_ctx* ctx, const StepCfg& c, hipStream_t s) {
    const wx_model_dims& D = ctx->d;
    const int d = D.n_text_state, H = D.n_text_head, T = D.n_audio_ctx, B = c.B;
    const int ks_d = pick_ksplit(d, d), ks_f2 = pick_ksplit(d, 4 * d);
    auto gemv = [&](const h16* A, long lda, const h16* W, int N, int K, const h16* bias, int gelu, int ksplit,
                    h16* out_h, float* out_f, long ldo, float* part) -> hipError_t {
        Skinny2Args g{};
        g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.bias = bias; g.out_h = out_h; g.out_f = out_f; g.ldo = ldo;
        g.part = part; g.ldp = N; g.M = B; g.N = N; g.K = K; g.ksplit = ksplit; g.gelu = gelu;
        return launch_skinny2(g, s);
    };
    auto resln = [&](const float* part, int ksplit, const h16* bias, const h16* g, const h16* b, bool embed) -> hipError_t {
        ResLnArgs r{};
        r.x = ctx->xd; r.part = part; r.ldp = d; r.ksplit = ksplit; r.bias = bias;
        if (embed) { r.tokens = c.tokens; r.tok_ld = c.tok_ld; r.d_pos = ctx->d_pos; r.emb = ctx->emb; r.pos = ctx->decpos; }
        r.g = g; r.b = b; r.xn = ctx->xn; r.d = d;
        return launch_resln(r, B, s);
    };
    // x = token + positional embedding; xn = LN1(x) of layer 0
    WX_CHECK_HIP(resln(nullptr, 0, nullptr, ctx->dec[0].ln1g, ctx->dec[0].ln1b, true));
    for (int l = 0; l < D.n_text_layer; ++l) {
        const DecLayer& L = ctx->dec[l];
        WX_CHECK_HIP(gemv(ctx->xn, d, L.qkvw, 3 * d, d, L.qkvb, 0, 1, ctx->qkv, nullptr, 3 * d, nullptr));
        DecSelfAttnArgs sa{ctx->qkv, 3L * d,
                           ctx->kc + (size_t)l * ctx->maxB * ctx->kv_ctx * d,
                           ctx->vc + (size_t)l * ctx->maxB * ctx->kv_ctx * d,
                           (long)ctx->kv_ctx * d, ctx->att, (long)d, ctx->d_pos, B, H, d};
        WX_CHECK_HIP(launch_dec_self_attn(sa, ctx->qkv + d, ctx->qkv + 2 * d, 3L * d, s));
        WX_CHECK_HIP(gemv(ctx->att, d, L.ow, d, d, nullptr, 0, ks_d, nullptr, nullptr, 0, ctx->partA));
        WX_CHECK_HIP(resln(ctx->partA, ks_d, L.ob, L.ln2g, L.ln2b, false));
        const h16* kv = ctx->ckv + (size_t)l * ctx->maxB * T * 2 * d;
        DecCrossAttnArgs ca{};
        if (ks_d > 1) {
            WX_CHECK_HIP(gemv(ctx->xn, d, L.cqw, d, d, nullptr, 0, ks_d, nullptr, nullptr, 0, ctx->partQ));
            ca.q_part = ctx->partQ; ca.q_ldp = d; ca.q_ksplit = ks_d; ca.q_bias = L.cqb;
        } else {
            WX_CHECK_HIP(gemv(ctx->xn, d, L.cqw, d, d, L.cqb, 0, 1, ctx->cq, nullptr, d, nullptr));
            ca.q = ctx->cq; ca.ldq = d;
        }
        ca.K = kv; ca.ldk = 64; ca.strideK = (long)T * d;
        ca.V = kv + (size_t)ctx->maxB * T * d; ca.ldv = 64; ca.strideV = (long)T * d;
        ca.hstride = (long)T * 64;
        ca.tickets = (ctx->fused_combine && ctx->merge_mode != 0) ? ctx->tickets : nullptr;
        ca.gran = ctx->merge_mode == 2 ? ctx->gran : nullptr; ca.d_pos = ctx->d_pos; ca.d_epoch = ctx->d_epoch; ca.layer = l; ca.d_err = ctx->d_err;
        ca.out = ctx->att; ca.ldo = d;
        ca.qk_out = (c.capture && ctx->align_qk) ? ctx->align_qk : nullptr;
        ca.cap_slot = ctx->cap_slot + (size_t)l * H;
        ca.n_cap = ctx->n_cap; ca.cap_rows = ctx->cap_rows; ca.d_row = ctx->d_row;
        ca.B = B; ca.H = H; ca.T = T;
        WX_CHECK_HIP(launch_dec_cross_attn(ca, c.cross_split, ctx->part, s));
        WX_CHECK_HIP(gemv(ctx->att, d, L.cow, d, d, nullptr, 0, ks_d, nullptr, nullptr, 0, ctx->partA));
        WX_CHECK_HIP(resln(ctx->partA, ks_d, L.cob, L.ln3g, L.ln3b, false));
        WX_CHECK_HIP(gemv(ctx->xn, d, L.fc1w, 4 * d, d, L.fc1b, 1, 1, ctx->f1, nullptr, 4 * d, nullptr));
        WX_CHECK_HIP(gemv(ctx->f1, 4 * d, L.fc2w, d, 4 * d, nullptr, 0, ks_f2, nullptr, nullptr, 0, ctx->partA));
        const bool last = (l + 1 == D.n_text_layer);
        const h16* ng = last ? ctx->declng : ctx->dec[l + 1].ln1g;
        const h16* nb = last ? ctx->declnb : ctx->dec[l + 1].ln1b;
        WX_CHECK_HIP(resln(ctx->partA, ks_f2, L.fc2b, ng, nb, false));
    }
    if (c.logits || c.sample) {
        float* lo = c.logits_out ? c.logits_out : ctx->logits;
        const long ld = c.logits_out ? c.logits_ld : ctx->vocab_ld;
        WX_CHECK_HIP(gemv(ctx->xn, d, ctx->emb, D.n_vocab, d, nullptr, 0, 1, nullptr, lo, ld, nullptr));
    }
    if (c.sample) WX_CHECK_HIP(launch_sample(c.sa, s));
    WX_CHECK_HIP(launch_advance(ctx->d_pos, ctx->d_row, c.sample_begin, s));
    return 0;
}

static int decode_step_v1(wx_ctx* ctx, const StepCfg& c, hipStream_t s) {
    const wx_model_dims& D = ctx->d;
    const int d = D.n_text_state, H = D.n_text_head, T = D.n_audio_ctx, B = c.B;
    // variant 3: M-tiled GEMVs with ceil(N / #CU) columns per block.  More than 16 rows otherwise run the 16-row kernels
    // over groups of 16 rows (grid.y): same bits per row as a 16-row launch, weights re-read by the other groups from L2
    const bool bal = c.variant == 3;
    // Which GEMV kernels a launch of more than 16 rows takes (skinny.hip): the one-pass kernel (four row groups per block)
    // pays when OTHER passes are in flight -- the host says so with the fat FC2 tile, wx_tuning.fc2_tile_n = 16 -- and the
    // launch has at least four row groups: three passes in flight, 64 rows +2.1 %, 112 rows +3.2 %, but 48 rows -2.1 %,
    // 32 rows -8.8 %, and one pass alone -4 % (64 rows) / -1 % (112): profiles/r05_ab_wide_threshold.txt, r05_ab_wide_gemv_*.txt.
    // lab builds (common.h, WX_LAB_ENV): WX_NO_WIDE_GEMV=1 the row-group kernels at every width, WX_WIDE_GEMV_FROM_17=1 the one-pass kernel from 17 rows on
    static const int lab_wide = WX_LAB_GETENV_INT("WX_NO_WIDE_GEMV", 0) ? 1 : (WX_LAB_GETENV_INT("WX_WIDE_GEMV_FROM_17", 0) ? -1 : 0);
    const int no_wide = lab_wide ? lab_wide : ((c.fc2_tn == 16 && B >= 64) ? -1 : 1);
    auto gemv = [&](SkinnyArgs a) { a.no_wide = no_wide; return bal ? launch_skinny_mt(a, ctx->n_cu, s) : launch_skinny(a, s); };
    // the GEMV launches stream the tile-blocked copies of their weights (wx_finalize): one contiguous KiB per fragment load
    auto blocked = [&](SkinnyArgs& a, const void* blk) {
        if (!ctx->w_blocked || bal || !blk) return;
        if (a.Wq) a.Wq = (const unsigned char*)blk; else a.W = (const h16*)blk;
        a.w_blocked = 1;
    };
    if (!c.embed_at_end) WX_CHECK_HIP(launch_embed(c.tokens, c.tok_ld, ctx->d_pos, ctx->emb, ctx->decpos, ctx->xd, B, d, s));
    for (int l = 0; l < D.n_text_layer; ++l) {
        const DecLayer& L = ctx->dec[l];
        SkinnyArgs q{};
        q.A = ctx->xd; q.lda = d; q.W = L.qkvw; q.ldw = d; q.bias = L.qkvb; q.ln_g = L.ln1g; q.ln_b = L.ln1b;
        q.out_h = ctx->qkv; q.ldo = 3 * d; q.M = B; q.N = 3 * d; q.K = d; q.Wq = L.qkvq; q.wscale = L.qkvs;
        blocked(q, L.qkv_blk);
        q.ln_scratch = ctx->xn;           // wide launches: LayerNorm as a launch of its own, then ONE pass over the weights (skinny_wide_kernel)
        DecSelfAttnArgs sa{ctx->qkv, 3L * d,
                           ctx->kc + (size_t)l * ctx->maxB * ctx->kv_ctx * d,
                           ctx->vc + (size_t)l * ctx->maxB * ctx->kv_ctx * d,
                           (long)ctx->kv_ctx * d, ctx->att, (long)d, ctx->d_pos, B, H, d};
        const int att_blocked = bal ? 0 : 1;   // attention -> out-proj hand-off, k-blocked (<= 16 rows)
        sa.out_blocked = att_blocked;
        sa.done = c.sample ? ctx->d_done : nullptr;      // sampling steps of wx_decode_greedy: rows that emitted EOT sit out
        // (fusing these two the way the cross-attention is fused with its query GEMV below was measured: tokens identical,
        // 1 % slower single stream and no gain with passes in flight -- the cached keys are a few KB per head, there is
        // no stream to hide behind -- so they stay two launches)
        WX_CHECK_HIP(gemv(q));
        WX_CHECK_HIP(launch_dec_self_attn(sa, ctx->qkv + d, ctx->qkv + 2 * d, 3L * d, s));
        SkinnyArgs o{};
        o.A = ctx->att; o.lda = d; o.W = L.ow; o.ldw = d; o.bias = L.ob; o.R = ctx->xd; o.ldr = d;
        // several passes in flight (fc2_tn == 16): the N = d GEMVs as 80 blocks of 16 columns instead of 160 of 8 -- slower alone
        // (-3.5 % single stream), faster together (+0.9 %): what a kernel leaves free counts as much as how long it takes
        const int tn_d = c.fc2_tn == 16 ? 16 : ctx->tn_small;
        o.out_h = ctx->xd; o.ldo = d; o.M = B; o.N = d; o.K = d; o.tile_n = tn_d; o.Wq = L.oq; o.wscale = L.os; o.a_blocked = att_blocked;
        blocked(o, L.o_blk);
        SkinnyArgs cqa{};
        cqa.A = ctx->xd; cqa.lda = d; cqa.W = L.cqw; cqa.ldw = d; cqa.bias = L.cqb; cqa.ln_g = L.ln2g; cqa.ln_b = L.ln2b;
        cqa.out_h = ctx->cq; cqa.ldo = d; cqa.M = B; cqa.N = d; cqa.K = d; cqa.tile_n = ctx->tn_cq; cqa.Wq = L.cqq; cqa.wscale = L.cqs;
        blocked(cqa, L.cq_blk);
        cqa.ln_scratch = ctx->xn;
        const h16* kv = ctx->ckv + (size_t)l * ctx->maxB * T * 2 * d;
        DecCrossAttnArgs ca{};
        ca.q = ctx->cq; ca.ldq = d;
        ca.K = kv; ca.ldk = 64; ca.strideK = (long)T * d;
        ca.V = kv + (size_t)ctx->maxB * T * d; ca.ldv = 64; ca.strideV = (long)T * d;
        ca.hstride = (long)T * 64;
        ca.tickets = (ctx->fused_combine && ctx->merge_mode != 0) ? ctx->tickets : nullptr;
        ca.gran = ctx->merge_mode == 2 ? ctx->gran : nullptr; ca.d_pos = ctx->d_pos; ca.d_epoch = ctx->d_epoch; ca.layer = l; ca.d_err = ctx->d_err;
        ca.out = ctx->att; ca.ldo = d;
        ca.qk_out = (c.capture && ctx->align_qk) ? ctx->align_qk : nullptr;
        ca.cap_slot = ctx->cap_slot + (size_t)l * H;
        ca.n_cap = ctx->n_cap; ca.cap_rows = ctx->cap_rows; ca.d_row = ctx->d_row;
        ca.B = B; ca.H = H; ca.T = T; ca.out_blocked = att_blocked;
        ca.done = sa.done;
        WX_CHECK_HIP(gemv(o));
        if (c.variant == 4 && c.cross_split == 2 && dec_cq_xattn_supported(cqa, ca)) {
            // one launch for two dependent stages: the attention blocks have half of their keys in flight while the
            // GEMV blocks still compute the query (a per-head hand-off of 32 granules per attention block).  The
            // output projection in front of it stays a launch of its own: as a third role its all-to-all hand-off
            // (every LayerNorm block sweeps 10240 granules) cost 7 us per layer more than the kernel boundary.
            WX_CHECK_HIP(launch_dec_cq_xattn(cqa, ca, ctx->gran_q, s, nullptr, ctx->d_selfq, false, c.profile ? ctx->prof : nullptr));
#ifdef LAB_DUMP_Q8
        } else if (c.variant == 6 && c.cross_split == 2 && dec_cq_xattn_supported(cqa, ca)) {
            // lab: the GEMV launch AND the fused launch on the same input, then the two queries compared on the device
            float* dump = reinterpret_cast<float*>(ctx->prof + 16);            // [2 slots][16 rows][8]
            SkinnyArgs c0 = cqa, c1 = cqa;
            c0.lab_dump = dump; c0.lab_slot = 0; c1.lab_dump = dump; c1.lab_slot = 1;
            WX_CHECK_HIP(gemv(c0));
            WX_CHECK_HIP(launch_dec_cq_xattn(c1, ca, ctx->gran_q, s, nullptr, ctx->d_selfq));
            hipLaunchKernelGGL(dbg_compare_q_kernel, dim3((B * (d / 2) + 255) / 256), dim3(256), 0, s, ctx->gran_q, ctx->cq, B, d,
                               ctx->d_pos, l, ctx->prof + 8, dump, dump + 256);
#endif
        } else if (c.variant == 5 && c.cross_split == 2 && dec_cq_xattn_supported(cqa, ca)) {
            // lab: the GEMV as a launch of its own, then ONLY the attention role of the fused kernel (query from memory)
            WX_CHECK_HIP(gemv(cqa));
            WX_CHECK_HIP(launch_dec_cq_xattn(cqa, ca, ctx->gran_q, s, nullptr, ctx->d_selfq, true));
        } else {
            WX_CHECK_HIP(gemv(cqa));
            WX_CHECK_HIP(launch_dec_cross_attn(ca, c.cross_split, ctx->part, s));
        }
        SkinnyArgs co{};
        co.A = ctx->att; co.lda = d; co.W = L.cow; co.ldw = d; co.bias = L.cob; co.R = ctx->xd; co.ldr = d;
        co.out_h = ctx->xd; co.ldo = d; co.M = B; co.N = d; co.K = d; co.tile_n = tn_d; co.Wq = L.coq; co.wscale = L.cos; co.a_blocked = att_blocked;
        blocked(co, L.co_blk);
        co.prof = c.profile ? ctx->prof : nullptr;      // closes the launch timer of the fused launch in front of it
        WX_CHECK_HIP(gemv(co));
        int f2_blocked = 0;
        SkinnyArgs f1{};
        f1.A = ctx->xd; f1.lda = d; f1.W = L.fc1w; f1.ldw = d; f1.bias = L.fc1b; f1.ln_g = L.ln3g; f1.ln_b = L.ln3b;
        f1.out_h = ctx->f1; f1.ldo = 4 * d; f1.M = B; f1.N = 4 * d; f1.K = d; f1.gelu = 1; f1.Wq = L.fc1q; f1.wscale = L.fc1s;
        blocked(f1, L.fc1_blk);
        f1.ln_scratch = ctx->xn;
        if (!bal) { f1.out_blocked = 1; f2_blocked = 1; }     // FC1 -> FC2 hand-off in the k-blocked layout (<= 16 rows)
        WX_CHECK_HIP(gemv(f1));
        SkinnyArgs f2{};
        f2.A = ctx->f1; f2.lda = 4 * d; f2.W = L.fc2w; f2.ldw = 4 * d; f2.bias = L.fc2b; f2.R = ctx->xd; f2.ldr = d;
        f2.out_h = ctx->xd; f2.ldo = d; f2.M = B; f2.N = d; f2.K = 4 * d; f2.tile_n = ctx->tn_small; f2.wide_block = 1; f2.Wq = L.fc2q; f2.wscale = L.fc2s; f2.a_blocked = f2_blocked;
        if (c.fc2_tn == 16) f2.tile_n = 16;   // 80 blocks of 16 waves: slower alone, leaves 2/3 of the CUs to other passes in flight
        blocked(f2, L.fc2_blk);
        WX_CHECK_HIP(gemv(f2));
    }
    if (c.logits || c.sample) {
        // final LayerNorm + the 133 MB tied-embedding GEMV: fused (each tile-walking block normalises the rows once)
        // where that kernel applies, else one LayerNorm launch and the plain GEMV
        Skinny2Args lg{};
        if (skinny2_can_fuse_ln(B, D.n_vocab, d)) {
            lg.A = ctx->xd; lg.ln_g = ctx->declng; lg.ln_b = ctx->declnb;
        } else if (d <= 1280 && (d & 7) == 0) {
            // more than 16 rows: the same LayerNorm arithmetic as the fused prologue, as a launch of its own -- a row's
            // logits do not depend on the number of rows in its pass
            WX_CHECK_HIP(launch_ln_rows16(ctx->xd, d, ctx->declng, ctx->declnb, ctx->xn, d, B, d, s));
            lg.A = ctx->xn;
        } else {
            ResLnArgs r{};
            r.x = ctx->xd; r.g = ctx->declng; r.b = ctx->declnb; r.xn = ctx->xn; r.d = d;
            WX_CHECK_HIP(launch_resln(r, B, s));
            lg.A = ctx->xn;
        }
        lg.lda = d; lg.W = ctx->emb; lg.ldw = d;
        lg.out_f = c.logits_out ? c.logits_out : ctx->logits;
        lg.ldo = c.logits_out ? c.logits_ld : ctx->vocab_ld;
        lg.M = B; lg.N = D.n_vocab; lg.K = d; lg.ksplit = 1;
        WX_CHECK_HIP(launch_skinny2(lg, s));
    }
    if (c.sample) {
        SampleArgs sa = c.sa;
        sa.done = ctx->d_done;
        if (c.embed_at_end) {
            sa.emb = ctx->emb; sa.decpos = ctx->decpos; sa.x = ctx->xd; sa.d = d;
            sa.d_pos_w = ctx->d_pos; sa.d_row = ctx->d_row; sa.ticket = ctx->samp_ticket;
        }
        WX_CHECK_HIP(launch_sample(sa, s));
        if (c.embed_at_end) return 0;
    }
    WX_CHECK_HIP(launch_advance(ctx->d_pos, ctx->d_row, c.sample_begin, s));
    if (c.embed_at_end) WX_CHECK_HIP(launch_embed(c.tokens, c.tok_ld, ctx->d_pos, ctx->emb, ctx->decpos, ctx->xd, B, d, s));
    return 0;
}

static int decode_step(wx_ctx* ctx, const StepCfg& c, hipStream_t s) {
    if (c.variant == 2 && c.B > 16) return wx_err(ctx, "decode step variant 2 handles at most 16 rows");
    if ((c.variant == 4 || c.variant == 5 || c.variant == 6) && c.cross_split != 2) { StepCfg c1 = c; c1.variant = 1; return decode_step_v1(ctx, c1, s); }
    if (c.variant == 2 && ctx->any_q8) return wx_err(ctx, "decode step variant 2 has no int8 weight path");
    return c.variant == 2 ? decode_step_v2(ctx, c, s) : decode_step_v1(ctx, c, s);
}

static int run_step(wx_ctx* ctx, const StepCfg& c, const std::string& key, bool use_graph, hipStream_t s) {
    if (!use_graph) return decode_step(ctx, c, s);
    auto it = ctx->graphs.exec.find(key);
    if (it == ctx->graphs.exec.end()) {
        if (ctx->graphs.exec.size() >= GraphCache::kMax) {
            WX_CHECK_HIP(hipStreamSynchronize(s));   // a replay of the graph about to be destroyed may still be running
            ctx->graphs.evict_oldest();
        }
        // One capture at a time per process (contexts on other host threads keep replaying their graphs meanwhile).
        // Callers that run several contexts from several threads let each context enqueue its first pass of a launch
        // shape before going parallel (backend._decode_chunks does): a capture that races with another thread's
        // allocator / event traffic on the same device can be rejected by the runtime ("unjoined work").
        static std::mutex capture_mu;
        std::lock_guard<std::mutex> lock(capture_mu);
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        WX_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = decode_step(ctx, c, s);
        hipError_t e = hipStreamEndCapture(s, &graph);
        if (rc != 0) {
            if (graph) hipGraphDestroy(graph);
            return rc;
        }
        WX_CHECK_HIP(e);
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        WX_CHECK_HIP(e);
        it = ctx->graphs.exec.emplace(key, GraphCache::Entry{exec, 0}).first;
    }
    it->second.last_use = ++ctx->graphs.clock;
    WX_CHECK_HIP(hipGraphLaunch(it->second.exec, s));
    return 0;
}

int wx_decode_greedy(wx_ctx* ctx, const void* enc_f16, int B, const wx_decode_opts* o, int32_t* tokens_out,
                     float* sum_logprob, float* no_speech_prob, int* n_steps_out_host, void* stream) {
    if (!ctx || !ctx->finalized || !o) return wx_err(ctx, "wx_decode_greedy: not finalized");
    WX_ENTER(ctx);
    if (B < 1 || B > ctx->maxB) return wx_err(ctx, "wx_decode_greedy: bad batch");
    const wx_tuning tdef = WX_TUNING_DEFAULTS;
    const wx_tuning* t = o->tuning ? o->tuning : &tdef;      // NULL: the library's own configuration (include/wxhip_test.h)
    if (B > 16 && t->step_variant == 3 && (size_t)((B + 15) / 16) * 16 * (ctx->d.n_text_state + 8) * 2 > 150 * 1024)
        return wx_err(ctx, "wx_decode_greedy: at this model width one decode launch takes at most 48 rows");
    const wx_model_dims& D = ctx->d;
    if (o->n_prompt < 1 || o->n_prompt > 8) return wx_err(ctx, "wx_decode_greedy: bad prompt");
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    const int max_new = o->forced_len > 0 ? o->forced_len : o->sample_len;
    if (o->n_prompt + max_new > ctx->kv_ctx)
        return wx_err(ctx, "wx_decode_greedy: prompt + sample_len exceeds the " + std::to_string(ctx->kv_ctx) +
                               " positions of the self-attention cache (8 + n_text_ctx / 2)");
    if (o->capture_qk && (!ctx->align_qk || max_new > ctx->cap_rows))
        return wx_err(ctx, "wx_decode_greedy: capture_qk needs wx_set_alignment_heads and sample_len <= n_text_ctx/2");
    const int split = (t->cross_split == 1 || t->cross_split == 2 || t->cross_split == 4) ? t->cross_split : 2;

    const int n_active = (o->n_active > 0 && o->n_active < B) ? o->n_active : B;
    int rc = cross_kv(ctx, reinterpret_cast<const h16*>(enc_f16), n_active, s);     // padding rows keep whatever the cache holds
    if (rc) return rc;
    // prompt to the device (tok_tmp doubles as the staging buffer)
    WX_CHECK_HIP(launch_set_ints(ctx->tok_tmp, o->prompt, o->n_prompt, s));   // by value: `o` belongs to the caller
    hipLaunchKernelGGL(init_decode_kernel, dim3(B), dim3(64), 0, s, tokens_out, D.n_text_ctx, D.n_text_ctx, ctx->tok_tmp,
                       o->n_prompt, o->eot, sum_logprob, no_speech_prob, ctx->d_pos, ctx->d_row, ctx->d_done, n_active, ctx->d_nact);
    WX_CHECK_HIP(hipGetLastError());
    WX_CHECK_HIP(hipMemsetAsync(ctx->tickets, 0, sizeof(unsigned) * (size_t)ctx->maxB * D.n_text_head, s));
    WX_CHECK_HIP(hipMemsetAsync(ctx->samp_ticket, 0, sizeof(unsigned), s));   // self-resetting, but an aborted call may leave it
    WX_CHECK_HIP(hipMemsetAsync(ctx->samp_row_ticket, 0, sizeof(unsigned) * (size_t)ctx->maxB, s));
    WX_CHECK_HIP(bump_epoch(ctx, s));
    if (o->capture_qk)
        WX_CHECK_HIP(hipMemsetAsync(ctx->align_qk, 0, sizeof(float) * (size_t)B * ctx->n_cap * ctx->cap_rows * D.n_audio_ctx, s));

    StepCfg c{};
    c.tokens = tokens_out; c.tok_ld = D.n_text_ctx; c.B = B;
    c.cross_split = split; c.capture = o->capture_qk != 0; c.sample_begin = o->n_prompt;
    // 0 = default: fused launches where they apply (variant 4); 1 = one kernel per stage; 2 / 3 = older GEMV forms
    c.variant = (t->step_variant >= 1 && t->step_variant <= 6) ? t->step_variant : 4;     // 6: lab builds only (LAB_DUMP_Q8)
    c.fc2_tn = t->fc2_tile_n == 16 ? 16 : 0;
    c.profile = t->profile_launches != 0;
    c.embed_at_end = c.variant != 2;
    if (c.embed_at_end)   // position 0's input; every later position is embedded at the end of the step before it
        WX_CHECK_HIP(launch_embed(tokens_out, D.n_text_ctx, ctx->d_pos, ctx->emb, ctx->decpos, ctx->xd, B, D.n_text_state, s));
    c.sa = SampleArgs{ctx->logits, (long)ctx->vocab_ld, tokens_out, D.n_text_ctx, sum_logprob, no_speech_prob,
                      o->suppress_mask, ctx->d_pos, B, D.n_vocab, o->n_prompt, o->eot, o->no_speech,
                      o->timestamp_begin, o->blank0, o->blank1, o->rules, o->max_initial_ts, o->forced_len};
    c.sa.part = ctx->samp_part; c.sa.row_ticket = ctx->samp_row_ticket;
    c.sa.forced_lens = o->forced_len > 0 ? o->forced_lens : nullptr;
    c.sa.n_active = ctx->d_nact;
    // everything a captured step bakes into its kernel arguments
    char keybuf[384];
    snprintf(keybuf, sizeof keybuf, "%p|%p|%p|%p|%p|%p|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d", (void*)tokens_out,
             (void*)sum_logprob, (void*)no_speech_prob, (void*)o->suppress_mask, (void*)ctx->align_qk, (void*)c.sa.forced_lens, B, o->n_prompt, o->rules,
             o->max_initial_ts, o->forced_len, split, o->capture_qk, c.variant * 100 + c.fc2_tn, ctx->n_cap, ctx->cap_rows,
             o->eot, o->no_speech, o->timestamp_begin, o->blank0, o->blank1, ctx->heads_version);
    const std::string key = std::string(keybuf) + (c.profile ? "|t" : "");

    int sampled = 0;
    const int last_pos = o->n_prompt - 1 + max_new - 1;
    // wx_tuning.max_steps_ahead: two events leapfrog, each recorded every 2 * half steps and waited for before it is
    // recorded again -- the host is then between half and 2 * half steps ahead of the GPU
    const int half = t->max_steps_ahead > 0 ? (t->max_steps_ahead + 1) / 2 : 0;
    bool ev_used[2] = {false, false};
    for (int p = 0; p <= last_pos; ++p) {
        const bool samp = p >= o->n_prompt - 1;
        c.sample = samp;
        c.logits = samp;
        rc = run_step(ctx, c, key + (samp ? "|s" : "|p"), t->use_graph != 0, s);
        if (rc) return rc;
        if (half && (p + 1) % half == 0) {
            const int k = ((p + 1) / half) & 1;
            if (!ctx->ahead_ev[k]) WX_CHECK_HIP(hipEventCreateWithFlags(&ctx->ahead_ev[k], hipEventDisableTiming));
            if (ev_used[k]) WX_CHECK_HIP(hipEventSynchronize(ctx->ahead_ev[k]));
            WX_CHECK_HIP(hipEventRecord(ctx->ahead_ev[k], s));
            ev_used[k] = true;
        }
        if (samp) ++sampled;
        if (samp && o->forced_len <= 0 && t->check_every > 0 && (sampled % t->check_every) == 0 && p < last_pos) {
            int done[128];
            hipLaunchKernelGGL(done_kernel, dim3(1), dim3(B), 0, s, tokens_out, D.n_text_ctx, ctx->d_pos, o->eot, ctx->d_done);
            WX_CHECK_HIP(hipMemcpyAsync(done, ctx->d_done, sizeof(int) * B, hipMemcpyDeviceToHost, s));
            WX_CHECK_HIP(hipStreamSynchronize(s));
            bool all = true;
            for (int b = 0; b < B; ++b) all = all && done[b];
            if (all) break;
        }
    }
    if (n_steps_out_host) *n_steps_out_host = sampled;
    return 0;
}

int wx_decode_logits(wx_ctx* ctx, const void* enc_f16, int B, const int32_t* tokens, int n, float* logits_out, void* stream) {
    if (!ctx || !ctx->finalized) return wx_err(ctx, "wx_decode_logits: not finalized");
    if (B < 1 || B > ctx->maxB || n < 1 || n > ctx->kv_ctx) return wx_err(ctx, "wx_decode_logits: bad shape");
    WX_ENTER(ctx);
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    int rc = cross_kv(ctx, reinterpret_cast<const h16*>(enc_f16), B, s);
    if (rc) return rc;
    WX_CHECK_HIP(hipMemsetAsync(ctx->d_pos, 0, sizeof(int), s));
    WX_CHECK_HIP(bump_epoch(ctx, s));
    StepCfg c{};
    c.tokens = tokens; c.tok_ld = n; c.B = B; c.cross_split = 2; c.capture = false; c.sample_begin = n;
    c.variant = 4;      // the step kernels of wx_decode_greedy's default path
    for (int p = 0; p < n; ++p) {
        c.sample = false;
        c.logits = (p == n - 1);
        c.logits_out = logits_out;
        c.logits_ld = ctx->d.n_vocab;
        rc = decode_step(ctx, c, s);
        if (rc) return rc;
    }
    return 0;
}

int wx_device_status(wx_ctx* ctx, void* stream) {
    if (!ctx || !ctx->finalized) return -2;
    hipSetDevice(ctx->device);
    int err = 0;
    WX_CHECK_HIP(hipMemcpyAsync(&err, ctx->d_err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    WX_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (err) {
        // read and clear: the caller decides what to do about the poisoned rows (the backend decodes the batch again)
        WX_CHECK_HIP(hipMemsetAsync(ctx->d_err, 0, sizeof(int), (hipStream_t)stream));
        return wx_err(ctx, "a decode kernel gave up waiting for the other key splits of a cross-attention row (results poisoned)");
    }
    return 0;
}

// ---- do these streams run side by side? -------------------------------------------------------------------------
// The runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and the variable is read when
// the GPU is first touched -- possibly before this library's host could set it.  Streams that share a queue run one
// after the other, and a pass in flight on such a stream costs more than it brings, so the host asks before it decides
// how many passes to keep in flight: one block per stream spins for `usec` on the constant 100 MHz clock; the wall time
// of n spins launched together over the wall time of one is ~1 when every stream has a queue of its own, ~2 when two share.
namespace {
__global__ void spin_kernel(long ticks, int* sink) {
    const long t0 = (long)__builtin_amdgcn_s_memrealtime();
    long t = t0;
    for (int i = 0; i < (1 << 24) && t - t0 < ticks; ++i) t = (long)__builtin_amdgcn_s_memrealtime();   // bounded either way
    if (sink && t == 0) *sink = 1;
}
}  // namespace

int wx_streams_overlap(int device, void* const* streams, int n, int usec, float* factor) {
    if (!streams || !factor || n < 1 || n > 16 || usec < 10 || usec > 20000) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    const long ticks = 100L * usec;
    auto run = [&](int m) -> double {
        for (int i = 0; i < m; ++i)
            if (hipStreamSynchronize((hipStream_t)streams[i]) != hipSuccess) return -1.0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < m; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)streams[i], ticks, (int*)nullptr);
        for (int i = 0; i < m; ++i)
            if (hipStreamSynchronize((hipStream_t)streams[i]) != hipSuccess) return -1.0;
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    if (run(n) < 0) return -3;                 // warm: code object load, queue creation
    double one = 1e30, all = 1e30;
    for (int r = 0; r < 3; ++r) {
        const double a = run(1), b = run(n);
        if (a < 0 || b < 0) return -3;
        one = a < one ? a : one;
        all = b < all ? b : all;
    }
    *factor = (float)(all / one);
    return 0;
}

int wx_get_align_qk(wx_ctx* ctx, int B, float* qk_out, void* stream) {
    if (!ctx || !ctx->align_qk) return wx_err(ctx, "wx_get_align_qk: nothing captured");
    if (B < 1 || B > ctx->maxB) return wx_err(ctx, "wx_get_align_qk: bad batch");
    hipSetDevice(ctx->device);
    WX_CHECK_HIP(hipMemcpyAsync(qk_out, ctx->align_qk, sizeof(float) * (size_t)B * ctx->n_cap * ctx->cap_rows * ctx->d.n_audio_ctx,
                                hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int wx_dtw_path(wx_ctx* ctx, const int32_t* tokens, const int32_t* n_frames, int B, int n_prompt, int n_sampled, int eot, int mode, float qk_scale,
                int32_t* n_rows, int32_t* path_i, int32_t* path_j, int path_ld, int32_t* path_len, float* matrix_out,
                void* stream) {
    if (!ctx || !ctx->align_qk) return wx_err(ctx, "wx_dtw_path: no captured scores (wx_set_alignment_heads + capture_qk)");
    if (B < 1 || B > ctx->maxB) return wx_err(ctx, "wx_dtw_path: bad batch");
    const int T = ctx->d.n_audio_ctx, R = ctx->cap_rows;
    if (path_ld < T + R + 3) return wx_err(ctx, "wx_dtw_path: path_ld too small");
    WX_ENTER(ctx);
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    DtwArgs a{};
    a.qk = ctx->align_qk; a.tokens = tokens; a.n_frames = n_frames; a.tok_ld = ctx->d.n_text_ctx; a.sample_begin = n_prompt;
    a.work = ctx->dtw_work; a.work2 = ctx->dtw_work2; a.trace = ctx->dtw_trace; a.rowmap = ctx->dtw_rowmap;
    a.n_rows = n_rows; a.path_i = path_i; a.path_j = path_j; a.path_len = path_len;
    a.trace_stride = (long)(R + 2) * (T + 1); a.path_stride = path_ld;
    a.B = B; a.n_cap = ctx->n_cap; a.rows = R; a.T = T; a.eot = eot; a.mode = mode; a.qk_scale = qk_scale;
    a.n_sampled = n_sampled < R ? n_sampled : R;
    WX_CHECK_HIP(launch_dtw(a, s));
    if (matrix_out)
        WX_CHECK_HIP(hipMemcpyAsync(matrix_out, ctx->dtw_work, sizeof(float) * (size_t)B * (R + 1) * T, hipMemcpyDeviceToDevice, s));
    return 0;
}

int wx_ctc_align(wx_ctx* ctx, const float* logp, const int32_t* T, const int32_t* tokens, const int32_t* N, int S,
                 int Tmax, int Nmax, int V, int blank_id, int beam, int32_t* path_tok, float* path_score, int32_t* ok,
                 float* trellis_out, void* stream) {
    if (!ctx) return -2;
    if (S < 1 || Tmax < 1 || Nmax < 1 || V < 2) return wx_err(ctx, "wx_ctc_align: bad shape");
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t n_tr = trellis_out ? 0 : (size_t)S * Tmax * Nmax;
    const size_t n_w = (size_t)S * Tmax, n_bp = (size_t)S * (Tmax + 1) * 8;
    const size_t need = (n_tr + n_w) * sizeof(float) + n_bp * (2 * sizeof(int) + sizeof(float)) + 256;
    if (need > ctx->ctc_scratch_bytes) {
        WX_CHECK_HIP(hipStreamSynchronize(s));
        if (ctx->ctc_scratch) hipFree(ctx->ctc_scratch);
        ctx->ctc_scratch = nullptr;
        ctx->ctc_scratch_bytes = 0;
        WX_CHECK_HIP(hipMalloc(&ctx->ctc_scratch, need));
        ctx->ctc_scratch_bytes = need;
    }
    char* base = reinterpret_cast<char*>(ctx->ctc_scratch);
    CtcArgs a{};
    a.logp = logp; a.seg_stride = (long)Tmax * V; a.V = V; a.T = T; a.tokens = tokens; a.Nmax = Nmax; a.N = N;
    a.trellis = trellis_out ? trellis_out : reinterpret_cast<float*>(base);
    base += n_tr * sizeof(float);
    a.wild = reinterpret_cast<float*>(base); base += n_w * sizeof(float);
    a.bp_tok = reinterpret_cast<int*>(base); base += n_bp * sizeof(int);
    a.bp_par = reinterpret_cast<int*>(base); base += n_bp * sizeof(int);
    a.bp_prob = reinterpret_cast<float*>(base);
    a.path_tok = path_tok; a.path_score = path_score; a.ok = ok;
    a.S = S; a.Tmax = Tmax; a.blank = blank_id; a.beam = beam;
    WX_CHECK_HIP(launch_ctc(a, s));
    return 0;
}

int wx_sample_step(wx_ctx* ctx, const float* logits, long ldl, int32_t* tokens, int tok_ld, int n_tokens, int B,
                   const wx_decode_opts* o, float* sum_logprob, float* no_speech_prob, void* stream) {
    if (!ctx || !ctx->finalized || !o) return wx_err(ctx, "wx_sample_step: not finalized");
    if (n_tokens < o->n_prompt || n_tokens >= tok_ld) return wx_err(ctx, "wx_sample_step: bad n_tokens");
    if (B < 1 || B > ctx->maxB) return wx_err(ctx, "wx_sample_step: bad batch");
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    const int pos = n_tokens - 1;
    WX_CHECK_HIP(launch_set_ints(ctx->d_pos, &pos, 1, s));
    SampleArgs sa{logits, ldl, tokens, tok_ld, sum_logprob, no_speech_prob, o->suppress_mask, ctx->d_pos, B,
                  ctx->d.n_vocab, o->n_prompt, o->eot, o->no_speech, o->timestamp_begin, o->blank0, o->blank1,
                  o->rules, o->max_initial_ts, o->forced_len};
    sa.part = ctx->samp_part; sa.row_ticket = ctx->samp_row_ticket;
    WX_CHECK_HIP(hipMemsetAsync(ctx->samp_row_ticket, 0, sizeof(unsigned) * (size_t)ctx->maxB, s));
    WX_CHECK_HIP(launch_sample(sa, s));
    return 0;
}

int wx_probe(wx_ctx* ctx, int kind, int B, int iters, int arg, void* stream) {
    if (!ctx || !ctx->finalized) return wx_err(ctx, "wx_probe: not finalized");
    if (B < 1 || B > ctx->maxB || iters < 1) return wx_err(ctx, "wx_probe: bad args");
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    encws_alias(ctx);          // (the encoder kinds run on the shared encoder workspace as the last wx_encode of any context left it)
    const wx_model_dims& D = ctx->d;
    const int da = D.n_audio_state, dt = D.n_text_state, T = D.n_audio_ctx;
    for (int it = 0; it < iters; ++it) {
        switch (kind) {
        case 0: {   // decode cross attention, layer 0 (reads the resident cross-KV of the last decode)
            DecCrossAttnArgs ca{};
            const int l = it % D.n_text_layer;   // rotate over the layers: every launch streams bytes that are not cache resident
            ca.q = ctx->cq; ca.ldq = dt;
            ca.K = ctx->ckv; ca.ldk = 64; ca.strideK = (long)T * dt;
            ca.V = ctx->ckv + (size_t)ctx->maxB * T * dt; ca.ldv = 64; ca.strideV = (long)T * dt;
            ca.hstride = (long)T * 64;
        ca.tickets = (ctx->fused_combine && ctx->merge_mode != 0) ? ctx->tickets : nullptr;
        ca.gran = ctx->merge_mode == 2 ? ctx->gran : nullptr; ca.d_pos = ctx->d_pos; ca.d_epoch = ctx->d_epoch; ca.layer = l; ca.d_err = ctx->d_err;
            ca.out = ctx->att; ca.ldo = dt; ca.qk_out = nullptr; ca.cap_slot = ctx->cap_slot;
            ca.n_cap = ctx->n_cap; ca.cap_rows = ctx->cap_rows; ca.d_row = ctx->d_row;
            ca.B = B; ca.H = D.n_text_head; ca.T = T;
            ca.K += (size_t)l * ctx->maxB * T * 2 * dt;
            ca.V += (size_t)l * ctx->maxB * T * 2 * dt;
            // arg = nsplit + 16 * (threads / 64)
            WX_CHECK_HIP(launch_dec_cross_attn(ca, (arg & 15) > 0 ? (arg & 15) : 4, ctx->part, s, (arg >> 4) ? (arg >> 4) * 64 : 256));
            break;
        }
        case 13: {   // the fused launch of the default decode step: [LN + cross-Q GEMV] -> [cross attention], rotating over the layers
            const int l = it % D.n_text_layer;
            const DecLayer& L = ctx->dec[l];
            if (l == 0) WX_CHECK_HIP(bump_epoch(ctx, s));      // fresh tags per sweep over the layers: no earlier granule can match
            SkinnyArgs cqa{};
            cqa.A = ctx->xd; cqa.lda = dt; cqa.W = L.cqw; cqa.ldw = dt; cqa.bias = L.cqb; cqa.ln_g = L.ln2g; cqa.ln_b = L.ln2b;
            cqa.out_h = ctx->cq; cqa.ldo = dt; cqa.M = B; cqa.N = dt; cqa.K = dt; cqa.tile_n = ctx->tn_cq; cqa.Wq = L.cqq; cqa.wscale = L.cqs;
            DecCrossAttnArgs ca{};
            const h16* kv = ctx->ckv + (size_t)l * ctx->maxB * T * 2 * dt;
            ca.q = ctx->cq; ca.ldq = dt;
            ca.K = kv; ca.ldk = 64; ca.strideK = (long)T * dt;
            ca.V = kv + (size_t)ctx->maxB * T * dt; ca.ldv = 64; ca.strideV = (long)T * dt;
            ca.hstride = (long)T * 64;
            ca.gran = ctx->gran; ca.d_pos = ctx->d_pos; ca.d_epoch = ctx->d_epoch; ca.layer = l; ca.d_err = ctx->d_err;
            ca.out = ctx->att; ca.ldo = dt; ca.qk_out = nullptr; ca.cap_slot = ctx->cap_slot;
            ca.n_cap = ctx->n_cap; ca.cap_rows = ctx->cap_rows; ca.d_row = ctx->d_row;
            ca.B = B; ca.H = D.n_text_head; ca.T = T; ca.out_blocked = 1;
            if (!dec_cq_xattn_supported(cqa, ca)) return wx_err(ctx, "wx_probe 13: the fused launch does not apply to this model");
            WX_CHECK_HIP(launch_dec_cq_xattn(cqa, ca, ctx->gran_q, s, nullptr, ctx->d_selfq));
            break;
        }
        case 1: {   // encoder FC1 GEMM + GELU: [B*1500, d] x [4d, d]^T   (arg & 1: without the GELU, to price it)
            const EncLayer& L = ctx->enc[it % D.n_audio_layer];
            GemmArgs g = gemm_rowmajor(L.fc1w, 4 * da, da, ctx->h, da, B * T, L.fc1b, nullptr, 0, ctx->f, 4 * da);
            g.max_blocks = ctx->enc_cap;       // wx_set_encoder_cap applies to the probe as to wx_encode
            WX_CHECK_HIP(launch_gemm_f16(g, 1, !(arg & 1), s));
            break;
        }
        case 2: {   // encoder self attention
            AttnArgs at{ctx->qk, 2L * da, (long)T * 2 * da, ctx->qk + da, 2L * da, (long)T * 2 * da,
                        ctx->vt, (long)ctx->Tpad, (long)da * ctx->Tpad, ctx->a, (long)da, (long)T * da, nullptr, T,
                        D.n_audio_head, B};
            at.q_prescaled = 1;      // ctx->qk as the last wx_encode left it
            WX_CHECK_HIP(launch_attention(at, s));
            break;
        }
        case 3: {   // decode: LN + QKV skinny GEMM
            const DecLayer& L = ctx->dec[it % D.n_text_layer];
            Skinny2Args q{};
            q.A = ctx->xn; q.lda = dt; q.W = L.qkvw; q.ldw = dt; q.bias = L.qkvb;
            q.out_h = ctx->qkv; q.ldo = 3 * dt; q.M = B; q.N = 3 * dt; q.K = dt; q.ksplit = 1;
            WX_CHECK_HIP(launch_skinny2(q, s));
            break;
        }
        case 4: {   // decode: FC2 skinny GEMM (K = 4d), no residual so the probe does not drift
            const DecLayer& L = ctx->dec[it % D.n_text_layer];
            Skinny2Args f2{};
            f2.A = ctx->f1; f2.lda = 4 * dt; f2.W = L.fc2w; f2.ldw = 4 * dt;
            f2.part = ctx->partA; f2.ldp = dt; f2.M = B; f2.N = dt; f2.K = 4 * dt;
            f2.ksplit = arg > 0 ? arg : pick_ksplit(dt, 4 * dt);
            WX_CHECK_HIP(launch_skinny2(f2, s));
            break;
        }
        case 5: {   // decode: final LN + tied-embedding logits
            Skinny2Args lg{};
            lg.A = ctx->xn; lg.lda = dt; lg.W = ctx->emb; lg.ldw = dt;
            lg.out_f = ctx->logits; lg.ldo = ctx->vocab_ld; lg.M = B; lg.N = D.n_vocab; lg.K = dt; lg.ksplit = 1;
            WX_CHECK_HIP(launch_skinny2(lg, s));
            break;
        }
        case 6: {   // encoder FC2 GEMM (K = 4d) without residual
            const EncLayer& L = ctx->enc[it % D.n_audio_layer];
            GemmArgs g = gemm_rowmajor(L.fc2w, da, 4 * da, ctx->f, 4 * da, B * T, L.fc2b, nullptr, 0, ctx->a, da);
            g.max_blocks = ctx->enc_cap;
            WX_CHECK_HIP(launch_gemm_f16(g, 1, false, s));
            break;
        }
        case 7: case 8: case 9: case 10: case 12: {   // v1 GEMVs: 7 out-proj (K=d), 8 LN+fc1, 9 fc2 (K=4d), 10 LN+qkv
            const DecLayer& L = ctx->dec[it % D.n_text_layer];
            SkinnyArgs q{};
            q.M = B; q.lda = dt; q.ldw = dt; q.K = dt;
            if (kind == 7) { q.A = ctx->att; q.W = L.ow; q.bias = L.ob; q.out_h = ctx->cq; q.ldo = dt; q.N = dt; q.tile_n = arg; }
            if (kind == 8) { q.A = ctx->xd; q.W = L.fc1w; q.bias = L.fc1b; q.ln_g = L.ln3g; q.ln_b = L.ln3b; q.out_h = ctx->f1; q.ldo = 4 * dt; q.N = 4 * dt; q.gelu = 1; q.tile_n = (arg < 1000) ? arg : 0; }
            if (kind == 9) { q.A = ctx->f1; q.lda = 4 * dt; q.W = L.fc2w; q.ldw = 4 * dt; q.K = 4 * dt; q.bias = L.fc2b; q.out_h = ctx->cq; q.ldo = dt; q.N = dt; q.tile_n = arg & 31; q.wide_block = arg >> 5; }
            if (kind == 12) { q.A = ctx->xd; q.W = L.cqw; q.bias = L.cqb; q.ln_g = L.ln2g; q.ln_b = L.ln2b; q.out_h = ctx->cq; q.ldo = dt; q.N = dt; q.tile_n = arg; }
            if (kind == 10) { q.A = ctx->xd; q.W = L.qkvw; q.bias = L.qkvb; q.ln_g = L.ln1g; q.ln_b = L.ln1b; q.out_h = ctx->qkv; q.ldo = 3 * dt; q.N = 3 * dt; q.tile_n = (arg < 1000) ? arg : 0; }
            if (kind == 7) { q.Wq = L.oq; q.wscale = L.os; }
            if (kind == 8) { q.Wq = L.fc1q; q.wscale = L.fc1s; }
            if (kind == 9) { q.Wq = L.fc2q; q.wscale = L.fc2s; }
            if (kind == 12) { q.Wq = L.cqq; q.wscale = L.cqs; }
            if (kind == 10) { q.Wq = L.qkvq; q.wscale = L.qkvs; }
            if (arg >= 1000) {   // arg 1000: the M-tiled column-balanced kernel (decode step variant 3)
                q.tile_n = 0;
                q.wide_block = 0;
                WX_CHECK_HIP(launch_skinny_mt(q, ctx->n_cu, s));
            } else {
                WX_CHECK_HIP(launch_skinny(q, s));
            }
            break;
        }
        case 11: {   // decode self attention at position arg
            const int pos = arg;
            if (it == 0) WX_CHECK_HIP(launch_set_ints(ctx->d_pos, &pos, 1, s));
            DecSelfAttnArgs sa{ctx->qkv, 3L * dt, ctx->kc, ctx->vc, (long)ctx->kv_ctx * dt, ctx->att, (long)dt, ctx->d_pos, B,
                               D.n_text_head, dt};
            WX_CHECK_HIP(launch_dec_self_attn(sa, ctx->qkv + dt, ctx->qkv + 2 * dt, 3L * dt, s));
            break;
        }
        default:
            return wx_err(ctx, "wx_probe: unknown kind");
        }
    }
    return 0;
}

// Test hook (wxhip_test.h): the fused launch with its attention blocks polling a granule buffer that nobody publishes to
// (n_selfq_host != null), so that EVERY attention block takes the cold path of csrc/declayer.hip: after its bounded poll it
// computes the 64 query columns of its head itself (the GEMV role's own code for its row's 16-row group, same k order)
// and carries on.  Nothing is poisoned and no device flag is raised; `out_fused` must equal `out_ref` -- the two launches
// the fused one stands for -- bit for bit, and *n_selfq_host counts the blocks that took the path (B x heads).  With
// n_selfq_host == null the launch runs as the decode step issues it.  The product never calls this.
int wx_test_fused_selfq(wx_ctx* ctx, int B, void* out_fused, void* out_ref, int* n_selfq_host, void* stream) {
    if (!ctx || !ctx->finalized) return wx_err(ctx, "wx_test_fused_selfq: not finalized");
    if (B < 1 || B > ctx->maxB || !out_fused || !out_ref) return wx_err(ctx, "wx_test_fused_selfq: bad arguments");
    WX_ENTER(ctx);
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    const wx_model_dims& D = ctx->d;
    const int dt = D.n_text_state, T = D.n_audio_ctx;
    const DecLayer& L = ctx->dec[0];
    unsigned long long* silent = nullptr;
    WX_CHECK_HIP(hipMalloc(&silent, sizeof(unsigned long long) * ctx->gran_q_words));
    WX_CHECK_HIP(hipMemsetAsync(silent, 0, sizeof(unsigned long long) * ctx->gran_q_words, s));
    WX_CHECK_HIP(hipMemsetAsync(ctx->d_selfq, 0, sizeof(int), s));
    WX_CHECK_HIP(bump_epoch(ctx, s));
    SkinnyArgs cqa{};
    cqa.A = ctx->xd; cqa.lda = dt; cqa.W = L.cqw; cqa.ldw = dt; cqa.bias = L.cqb; cqa.ln_g = L.ln2g; cqa.ln_b = L.ln2b;
    cqa.out_h = ctx->cq; cqa.ldo = dt; cqa.M = B; cqa.N = dt; cqa.K = dt; cqa.tile_n = ctx->tn_cq; cqa.Wq = L.cqq; cqa.wscale = L.cqs;
    DecCrossAttnArgs ca{};
    ca.q = ctx->cq; ca.ldq = dt;
    ca.K = ctx->ckv; ca.ldk = 64; ca.strideK = (long)T * dt;
    ca.V = ctx->ckv + (size_t)ctx->maxB * T * dt; ca.ldv = 64; ca.strideV = (long)T * dt;
    ca.hstride = (long)T * 64;
    ca.tickets = ctx->tickets;
    ca.gran = ctx->gran; ca.d_pos = ctx->d_pos; ca.d_epoch = ctx->d_epoch; ca.layer = 0; ca.d_err = ctx->d_err;
    ca.ldo = dt; ca.cap_slot = ctx->cap_slot; ca.n_cap = ctx->n_cap; ca.cap_rows = ctx->cap_rows; ca.d_row = ctx->d_row;
    ca.B = B; ca.H = D.n_text_head; ca.T = T; ca.out_blocked = 0;
    int rc = 0;
    hipError_t e = hipSuccess;
    if (!dec_cq_xattn_supported(cqa, ca)) {
        rc = wx_err(ctx, "wx_test_fused_selfq: the fused launch does not apply to this model");
    } else {
        ca.out = reinterpret_cast<h16*>(out_fused);
        // n_selfq_host == null: the launch as the decode step issues it (the blocks poll the buffer that IS published to)
        e = launch_dec_cq_xattn(cqa, ca, ctx->gran_q, s, n_selfq_host ? silent : nullptr, ctx->d_selfq);
        // the two launches it stands for: LayerNorm + cross-Q GEMV, then the cross attention with two key splits
        ca.out = reinterpret_cast<h16*>(out_ref);
        if (e == hipSuccess) e = launch_skinny(cqa, s);
        if (e == hipSuccess) e = launch_dec_cross_attn(ca, 2, ctx->part, s);
        if (e == hipSuccess && n_selfq_host) e = hipMemcpyAsync(n_selfq_host, ctx->d_selfq, sizeof(int), hipMemcpyDeviceToHost, s);
    }
    hipError_t e2 = hipStreamSynchronize(s);
    hipFree(silent);
    if (rc) return rc;
    WX_CHECK_HIP(e);
    WX_CHECK_HIP(e2);
    WX_CHECK_HIP(hipMemsetAsync(ctx->d_selfq, 0, sizeof(int), s));
    return 0;
}

int wx_graph_generation(wx_ctx* ctx) { return ctx ? ctx->graphs.generation : -1; }

int wx_set_encoder_cap(wx_ctx* ctx, int max_blocks) {
    if (!ctx || max_blocks < -1 || (max_blocks > 0 && (max_blocks & 7)))
        return wx_err(ctx, "wx_set_encoder_cap: a multiple of 8 (0 = no cap, -1 = the one-tile-per-block GEMM kernel)");
    ctx->enc_cap = max_blocks;
    return 0;
}

int wx_test_raise_device_flag(wx_ctx* ctx, void* stream) {
    if (!ctx || !ctx->finalized) return -2;
    hipSetDevice(ctx->device);
    WX_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->d_err), 1, 1, (hipStream_t)stream));
    return 0;
}

#ifdef LAB_DUMP_Q8
extern "C" int wx_debug_read(wx_ctx* ctx, unsigned long long* out8, float* out16, void* stream) {      // lab: the record of step variant 6, read and cleared
    if (!ctx || !ctx->finalized || !out8 || !out16) return -2;
    hipSetDevice(ctx->device);
    WX_CHECK_HIP(hipMemcpyAsync(out8, ctx->prof + 8, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)stream));
    WX_CHECK_HIP(hipMemcpyAsync(out16, reinterpret_cast<float*>(ctx->prof + 16) + 256, 16 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    WX_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    WX_CHECK_HIP(hipMemsetAsync(ctx->prof + 8, 0, 8 * sizeof(unsigned long long), (hipStream_t)stream));
    return 0;
}
#endif

int wx_launch_profile(wx_ctx* ctx, double* avg_us, long long* n_launches, void* stream) {
    if (!ctx || !ctx->finalized || !avg_us || !n_launches) return -2;
    WX_ENTER(ctx);
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long rec[3] = {0, 0, 0};
    WX_CHECK_HIP(hipMemcpyAsync(rec, ctx->prof, sizeof rec, hipMemcpyDeviceToHost, s));
    WX_CHECK_HIP(hipStreamSynchronize(s));
    WX_CHECK_HIP(hipMemsetAsync(ctx->prof, 0, sizeof rec, s));      // read and clear
    *n_launches = (long long)rec[2];
    *avg_us = rec[2] ? (double)rec[1] * 0.01 / (double)rec[2] : 0.0;     // s_memrealtime ticks: 100 MHz
    return 0;
}

int wx_decode_stats(wx_ctx* ctx, int* selfq_out, void* stream) {
    if (!ctx || !ctx->finalized || !selfq_out) return -2;
    WX_ENTER(ctx);
    hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    WX_CHECK_HIP(hipMemcpyAsync(selfq_out, ctx->d_selfq, sizeof(int), hipMemcpyDeviceToHost, s));
    WX_CHECK_HIP(hipStreamSynchronize(s));
    if (*selfq_out) WX_CHECK_HIP(hipMemsetAsync(ctx->d_selfq, 0, sizeof(int), s));      // read and clear
    return 0;
}

// ------------------------------------------------------------------------------- test hooks
int wx_gemm_f16(wx_ctx* ctx, const void* X, long ldx, int RX, const void* Y, long ldy, int RY, int K, const void* bias,
                int bias_on_y, const void* R, long ldr, void* out, long ldo, int gelu, void* stream) {
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    GemmArgs g{};
    g.X = (const h16*)X; g.ldx = ldx; g.RX = RX; g.Y = (const h16*)Y; g.ldy = ldy; g.RY = RY; g.K = K;
    g.bias = (const h16*)bias; g.bias_on_y = bias_on_y; g.R = (const h16*)R; g.ldr = ldr; g.out = (h16*)out; g.ldo = ldo;
    g.max_blocks = ctx->enc_cap;       // wx_set_encoder_cap: > 0 a capped grid, -1 the one-tile-per-block kernel
    WX_CHECK_HIP(launch_gemm_f16(g, 1, gelu != 0, (hipStream_t)stream));
    return 0;
}

int wx_skinny_f16(wx_ctx* ctx, const void* A, long lda, int M, const void* W, long ldw, int N, int K, const void* bias,
                  const void* ln_g, const void* ln_b, const void* R, long ldr, void* out_h, float* out_f, long ldo,
                  int gelu, int tile_n, void* stream) {
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    SkinnyArgs a{};
    a.A = (const h16*)A; a.lda = lda; a.W = (const h16*)W; a.ldw = ldw; a.bias = (const h16*)bias;
    a.ln_g = (const h16*)ln_g; a.ln_b = (const h16*)ln_b; a.R = (const h16*)R; a.ldr = ldr;
    a.out_h = (h16*)out_h; a.out_f = out_f; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.gelu = gelu; a.tile_n = tile_n;
    a.no_wide = ctx->enc_cap < 0 ? 1 : -1;      // the one-pass kernel from 17 rows on (the tests' shapes); wx_set_encoder_cap(-1): row groups as blocks of their own
    WX_CHECK_HIP(hook_ln_scratch(ctx, a, (hipStream_t)stream));
    WX_CHECK_HIP(launch_skinny(a, (hipStream_t)stream));
    return 0;
}

int wx_pack_gemv_weight(wx_ctx* ctx, const void* w, int N, int K, int elem_bytes, void* out, void* stream) {
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    WX_CHECK_HIP(launch_pack_gemv_weight(w, out, N, K, elem_bytes, (hipStream_t)stream));
    return 0;
}

int wx_skinny_ex(wx_ctx* ctx, const void* A, long lda, int M, const void* W, const void* Wq, const float* wscale, long ldw, int N, int K,
                 const void* bias, const void* ln_g, const void* ln_b, const void* R, long ldr, void* out_h, float* out_f,
                 long ldo, int gelu, int tile_n, int wide_block, int w_blocked, void* stream) {
    if (!ctx || (!W == !Wq) || (Wq && !wscale)) return -2;
    hipSetDevice(ctx->device);
    SkinnyArgs a{};
    a.A = (const h16*)A; a.lda = lda; a.W = (const h16*)W; a.Wq = (const unsigned char*)Wq; a.wscale = wscale; a.ldw = ldw;
    a.w_blocked = w_blocked != 0; a.wide_block = wide_block != 0; a.bias = (const h16*)bias;
    a.ln_g = (const h16*)ln_g; a.ln_b = (const h16*)ln_b; a.R = (const h16*)R; a.ldr = ldr;
    a.out_h = (h16*)out_h; a.out_f = out_f; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.gelu = gelu; a.tile_n = tile_n;
    a.no_wide = ctx->enc_cap < 0 ? 1 : -1;
    WX_CHECK_HIP(hook_ln_scratch(ctx, a, (hipStream_t)stream));
    WX_CHECK_HIP(launch_skinny(a, (hipStream_t)stream));
    return 0;
}

int wx_skinny_mt_f16(wx_ctx* ctx, const void* A, long lda, int M, const void* W, long ldw, int N, int K, const void* bias,
                     const void* ln_g, const void* ln_b, const void* R, long ldr, void* out_h, float* out_f, long ldo,
                     int gelu, int n_cu, void* stream) {
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    SkinnyArgs a{};
    a.A = (const h16*)A; a.lda = lda; a.W = (const h16*)W; a.ldw = ldw; a.bias = (const h16*)bias;
    a.ln_g = (const h16*)ln_g; a.ln_b = (const h16*)ln_b; a.R = (const h16*)R; a.ldr = ldr;
    a.out_h = (h16*)out_h; a.out_f = out_f; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.gelu = gelu;
    WX_CHECK_HIP(launch_skinny_mt(a, n_cu > 0 ? n_cu : ctx->n_cu, (hipStream_t)stream));
    return 0;
}

int wx_skinny_q8(wx_ctx* ctx, const void* A, long lda, int M, const void* Wq, const float* wscale, long ldw, int N, int K,
                 const void* bias, const void* ln_g, const void* ln_b, const void* R, long ldr, void* out_h, float* out_f,
                 long ldo, int gelu, int balanced, void* stream) {
    if (!ctx || !Wq || !wscale) return -2;
    hipSetDevice(ctx->device);
    SkinnyArgs a{};
    a.A = (const h16*)A; a.lda = lda; a.Wq = (const unsigned char*)Wq; a.wscale = wscale; a.ldw = ldw; a.bias = (const h16*)bias;
    a.ln_g = (const h16*)ln_g; a.ln_b = (const h16*)ln_b; a.R = (const h16*)R; a.ldr = ldr;
    a.out_h = (h16*)out_h; a.out_f = out_f; a.ldo = ldo; a.M = M; a.N = N; a.K = K; a.gelu = gelu;
    if (balanced)
        WX_CHECK_HIP(launch_skinny_mt(a, ctx->n_cu, (hipStream_t)stream));
    else
        WX_CHECK_HIP(launch_skinny(a, (hipStream_t)stream));
    return 0;
}

int wx_skinny2_f16(wx_ctx* ctx, const void* A, long lda, int M, const void* W, long ldw, int N, int K, const void* bias,
                   int ksplit, int gelu, void* out_h, float* out_f, long ldo, float* part, void* stream) {
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    Skinny2Args g{};
    g.A = (const h16*)A; g.lda = lda; g.W = (const h16*)W; g.ldw = ldw; g.bias = (const h16*)bias;
    g.out_h = (h16*)out_h; g.out_f = out_f; g.ldo = ldo; g.part = part; g.ldp = N;
    g.M = M; g.N = N; g.K = K; g.ksplit = ksplit; g.gelu = gelu;
    WX_CHECK_HIP(launch_skinny2(g, (hipStream_t)stream));
    return 0;
}

int wx_skinny2_ln_f16(wx_ctx* ctx, const void* A, long lda, int M, const void* W, long ldw, int N, int K, const void* ln_g,
                      const void* ln_b, float* out_f, long ldo, void* stream) {
    if (!ctx) return -2;
    if (!ln_g || !ln_b || !skinny2_can_fuse_ln(M, N, K)) return wx_err(ctx, "wx_skinny2_ln_f16: shape outside the fused kernel (M <= 16, K <= 1280, N >= 32768)");
    hipSetDevice(ctx->device);
    Skinny2Args g{};
    g.A = (const h16*)A; g.lda = lda; g.W = (const h16*)W; g.ldw = ldw; g.out_f = out_f; g.ldo = ldo;
    g.M = M; g.N = N; g.K = K; g.ksplit = 1; g.ln_g = (const h16*)ln_g; g.ln_b = (const h16*)ln_b;
    WX_CHECK_HIP(launch_skinny2(g, (hipStream_t)stream));
    return 0;
}

int wx_resln_f16(wx_ctx* ctx, void* x, int M, int d, const float* part, int ksplit, const void* bias, const void* g,
                 const void* b, void* xn, void* stream) {
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    ResLnArgs r{};
    r.x = (h16*)x; r.part = part; r.ldp = d; r.ksplit = ksplit; r.bias = (const h16*)bias;
    r.g = (const h16*)g; r.b = (const h16*)b; r.xn = (h16*)xn; r.d = d;
    WX_CHECK_HIP(launch_resln(r, M, (hipStream_t)stream));
    return 0;
}

int wx_layernorm_f16(wx_ctx* ctx, const void* x, long ldx, const void* g, const void* b, void* y, long ldy, int rows,
                     int d, void* stream) {
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    WX_CHECK_HIP(launch_layernorm((const h16*)x, ldx, (const h16*)g, (const h16*)b, (h16*)y, ldy, rows, d, (hipStream_t)stream));
    return 0;
}

int wx_median7_rows(wx_ctx* ctx, const float* x, long ldx, int rows, int T, float* y, long ldy, void* stream) {
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    WX_CHECK_HIP(launch_median7_rows(x, ldx, rows, T, y, ldy, (hipStream_t)stream));
    return 0;
}

int wx_attention_f16(wx_ctx* ctx, const void* Q, long ldq, long strideQ, const void* K, long ldk, long strideK,
                     const void* VT, long ldvt, long strideVT, void* O, long ldo, long strideO, const int32_t* lens,
                     int T, int H, int B, void* stream) {
    if (!ctx) return -2;
    hipSetDevice(ctx->device);
    AttnArgs a{(const h16*)Q, ldq, strideQ, (const h16*)K, ldk, strideK, (const h16*)VT, ldvt, strideVT,
               (h16*)O, ldo, strideO, lens, T, H, B};
    WX_CHECK_HIP(launch_attention(a, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
