// C ABI for the wav2vec2 CTC forward (include/wxhip.h, wx_w2v_*): what
// whisperx/alignment.py:251-258 runs per segment, batched over padded segments with
// per-segment lengths so that every segment gets exactly the batch-of-one result.
#include "../../include/wxhip.h"
#include "kernels.h"

#include <cstdio>
#include <string>
#include <unordered_map>
#include <vector>

namespace {
struct W2vLayer {
    const h16 *qkw, *qkb, *vw, *vb, *ow, *ob, *ln1g, *ln1b, *fc1w, *fc1b, *fc2w, *fc2b, *ln2g, *ln2b;
};
}  // namespace

struct wx_w2v {
    int device = 0;
    std::string err;
    wx_w2v_dims d{};
    bool finalized = false;
    std::unordered_map<std::string, std::pair<const void*, size_t>> w;
    const float* conv0w = nullptr;
    const h16 *gng = nullptr, *gnb = nullptr, *fplng = nullptr, *fplnb = nullptr, *fpw = nullptr, *fpb = nullptr,
              *posw = nullptr, *posb = nullptr, *enclng = nullptr, *enclnb = nullptr, *lmw = nullptr, *lmb = nullptr;
    const h16* convw[8] = {};
    const h16 *convb[8] = {}, *convlng[8] = {}, *convlnb[8] = {};   // "layer" feature-encoder variant
    h16* hbuf = nullptr;                                             // stable-LN encoder: LN(x) scratch
    std::vector<W2vLayer> layers;
    // workspace, grown on demand
    std::vector<void*> bufs;
    size_t cap_S = 0, cap_n = 0;
    h16 *act[8] = {}, *feat = nullptr, *hp = nullptr, *x = nullptr, *qk = nullptr, *vt = nullptr, *a = nullptr, *f = nullptr;
    double* stats = nullptr;
    int *d_nf0 = nullptr, *d_lens = nullptr;
    std::vector<int> h_nf0, h_lens;   // kept alive across the async upload
    void* ctc_scratch = nullptr;
    size_t ctc_scratch_bytes = 0;
};

static int w2_fail(wx_w2v* ctx, hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    if (ctx) ctx->err = buf;
    return -1;
}
static int w2_err(wx_w2v* ctx, const std::string& m) {
    if (ctx) ctx->err = m;
    return -2;
}
#define W2_CHECK(expr)                                                             \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) return w2_fail(ctx, _e, #expr, __FILE__, __LINE__);  \
    } while (0)

static int frames_after(const wx_w2v_dims& d, long n, int upto) {
    for (int i = 0; i < upto; ++i) n = (n - d.conv_kernel[i]) / d.conv_stride[i] + 1;
    return (int)n;
}

extern "C" {

int wx_w2v_create(int device_id, const wx_w2v_dims* dims, wx_w2v** out) {
    if (!dims || !out) return -2;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device_id) return -3;
    wx_w2v* ctx = new wx_w2v();
    ctx->device = device_id;
    ctx->d = *dims;
    *out = ctx;
    const wx_w2v_dims& D = *dims;
    if (D.n_conv < 2 || D.n_conv > 8 || D.conv_kernel[0] != 10 || D.conv_stride[0] != 5 || D.hidden % 64 ||
        D.hidden / D.heads != 64 || D.conv_dim % 8 || D.vocab < 2 || D.vocab > 30000 || D.hidden % D.pos_groups ||
        (D.hidden / D.pos_groups) % 8 || D.norm_mode < 0 || D.norm_mode > 1 || D.conv_dim > 1024)
        return w2_err(ctx, "unsupported wav2vec2 config (need conv0 k10/s5, d_head 64, conv_dim <= 1024)");
    return 0;
}

void wx_w2v_destroy(wx_w2v* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (void* p : ctx->bufs) (void)hipFree(p);
    if (ctx->ctc_scratch) (void)hipFree(ctx->ctc_scratch);
    delete ctx;
}

const char* wx_w2v_last_error(wx_w2v* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int wx_w2v_bind_weight(wx_w2v* ctx, const char* name, const void* dptr, size_t nbytes) {
    if (!ctx || !name || !dptr) return -2;
    ctx->w[name] = {dptr, nbytes};
    return 0;
}

static const void* w2_get(wx_w2v* ctx, const std::string& name, size_t bytes, bool& ok) {
    auto it = ctx->w.find(name);
    if (it == ctx->w.end() || it->second.second != bytes) {
        if (ok) ctx->err = (it == ctx->w.end() ? "missing weight: " : "wrong size for weight: ") + name;
        ok = false;
        return nullptr;
    }
    return it->second.first;
}

int wx_w2v_finalize(wx_w2v* ctx) {
    if (!ctx) return -2;
    const wx_w2v_dims& D = ctx->d;
    const size_t C = D.conv_dim, d = D.hidden;
    bool ok = true;
    auto H = [&](const std::string& n, size_t elems) { return (const h16*)w2_get(ctx, n, elems * 2, ok); };
    ctx->conv0w = (const float*)w2_get(ctx, "fe.conv0.w", C * 10 * 4, ok);
    if (D.norm_mode == 0) {
        ctx->gng = H("fe.gn.g", C);
        ctx->gnb = H("fe.gn.b", C);
    } else {
        for (int i = 0; i < D.n_conv; ++i) {
            ctx->convb[i] = H("fe.conv" + std::to_string(i) + ".b", C);
            ctx->convlng[i] = H("fe.ln" + std::to_string(i) + ".g", C);
            ctx->convlnb[i] = H("fe.ln" + std::to_string(i) + ".b", C);
        }
    }
    for (int i = 1; i < D.n_conv; ++i) ctx->convw[i] = H("fe.conv" + std::to_string(i) + ".w", C * D.conv_kernel[i] * C);
    ctx->fplng = H("fp.ln.g", C);
    ctx->fplnb = H("fp.ln.b", C);
    ctx->fpw = H("fp.w", d * C);
    ctx->fpb = H("fp.b", d);
    ctx->posw = H("pos.w", d * (d / D.pos_groups) * D.pos_kernel);
    ctx->posb = H("pos.b", d);
    ctx->enclng = H("enc.ln.g", d);
    ctx->enclnb = H("enc.ln.b", d);
    ctx->layers.resize(D.layers);
    for (int i = 0; i < D.layers; ++i) {
        const std::string p = "l" + std::to_string(i) + ".";
        W2vLayer& L = ctx->layers[i];
        L.qkw = H(p + "qk.w", 2 * d * d);  L.qkb = H(p + "qk.b", 2 * d);
        L.vw = H(p + "v.w", d * d);        L.vb = H(p + "v.b", d);
        L.ow = H(p + "o.w", d * d);        L.ob = H(p + "o.b", d);
        L.ln1g = H(p + "ln1.g", d);        L.ln1b = H(p + "ln1.b", d);
        L.fc1w = H(p + "fc1.w", (size_t)D.ffn * d);  L.fc1b = H(p + "fc1.b", D.ffn);
        L.fc2w = H(p + "fc2.w", (size_t)D.ffn * d);  L.fc2b = H(p + "fc2.b", d);
        L.ln2g = H(p + "ln2.g", d);        L.ln2b = H(p + "ln2.b", d);
    }
    ctx->lmw = H("lm.w", (size_t)D.vocab * d);
    ctx->lmb = H("lm.b", D.vocab);
    if (!ok) return -2;
    ctx->finalized = true;
    return 0;
}

int wx_w2v_num_frames(const wx_w2v_dims* dims, long n_samples) {
    if (!dims) return -2;
    return frames_after(*dims, n_samples < 400 ? 400 : n_samples, dims->n_conv);
}

}  // extern "C"

template <typename T>
static hipError_t w2_alloc(wx_w2v* ctx, T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T) + 256);
    if (e != hipSuccess) return e;
    e = hipMemset(q, 0, n * sizeof(T) + 256);
    if (e != hipSuccess) return e;
    ctx->bufs.push_back(q);
    *p = (T*)q;
    return hipSuccess;
}

static int w2_reserve(wx_w2v* ctx, size_t S, size_t n_max) {
    if (S <= ctx->cap_S && n_max <= ctx->cap_n) return 0;
    (void)hipDeviceSynchronize();
    for (void* p : ctx->bufs) (void)hipFree(p);
    ctx->bufs.clear();
    S = S > ctx->cap_S ? S : ctx->cap_S;
    n_max = n_max > ctx->cap_n ? n_max : ctx->cap_n;
    const wx_w2v_dims& D = ctx->d;
    const size_t C = D.conv_dim, d = D.hidden;
    for (int i = 0; i < D.n_conv; ++i) {
        const size_t T = frames_after(D, (long)n_max, i + 1);
        W2_CHECK(w2_alloc(ctx, &ctx->act[i], S * T * C));
    }
    const size_t T = frames_after(D, (long)n_max, D.n_conv), Tpad = (T + 63) / 64 * 64, half = D.pos_kernel / 2;
    W2_CHECK(w2_alloc(ctx, &ctx->feat, S * T * C));
    W2_CHECK(w2_alloc(ctx, &ctx->hp, S * (T + 2 * half) * d));
    W2_CHECK(w2_alloc(ctx, &ctx->x, S * T * d));
    W2_CHECK(w2_alloc(ctx, &ctx->hbuf, S * T * d));
    W2_CHECK(w2_alloc(ctx, &ctx->qk, S * T * 2 * d));
    W2_CHECK(w2_alloc(ctx, &ctx->vt, S * d * Tpad));
    W2_CHECK(w2_alloc(ctx, &ctx->a, S * T * d));
    W2_CHECK(w2_alloc(ctx, &ctx->f, S * T * D.ffn));
    W2_CHECK(w2_alloc(ctx, &ctx->stats, S * (size_t)W2V_CONV0_STATS_PER_SEGMENT));      // [S][8 blocks][65 signal sums] (w2v.hip)
    W2_CHECK(w2_alloc(ctx, &ctx->d_nf0, S));
    W2_CHECK(w2_alloc(ctx, &ctx->d_lens, S));
    ctx->cap_S = S;
    ctx->cap_n = n_max;
    return 0;
}

static GemmArgs rowmajor(const h16* W, int N, int K, const h16* A, long lda, int M, const h16* bias, const h16* R,
                         long ldr, h16* out, long ldo) {
    GemmArgs g{};
    g.X = W; g.ldx = K; g.RX = N; g.Y = A; g.ldy = lda; g.RY = M; g.K = K;
    g.bias = bias; g.R = R; g.ldr = ldr; g.out = out; g.ldo = ldo;
    return g;
}

extern "C" {

int wx_w2v_emissions(wx_w2v* ctx, const float* pcm, long pcm_stride, const int32_t* n_samples_host, int S,
                     float* logp_out, int Tmax_out, int32_t* T_out_host, void* stream) {
    if (!ctx || !ctx->finalized) return w2_err(ctx, "wx_w2v_emissions: not finalized");
    if (S < 1 || pcm_stride < 400) return w2_err(ctx, "wx_w2v_emissions: need S >= 1 and pcm_stride >= 400 (pad short segments)");
    (void)hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    const wx_w2v_dims& D = ctx->d;
    const int C = D.conv_dim, d = D.hidden, H = D.heads;
    long n_max = 0;
    std::vector<int>& nf0 = ctx->h_nf0;
    std::vector<int>& lens = ctx->h_lens;
    nf0.resize(S);
    lens.resize(S);
    for (int i = 0; i < S; ++i) {
        long n = n_samples_host[i] < 400 ? 400 : n_samples_host[i];   // alignment.py:243-249
        if (n > pcm_stride) return w2_err(ctx, "wx_w2v_emissions: n_samples exceeds pcm_stride");
        n_max = n > n_max ? n : n_max;
        nf0[i] = frames_after(D, n, 1);
        lens[i] = frames_after(D, n, D.n_conv);
        if (T_out_host) T_out_host[i] = lens[i];
    }
    const int T = frames_after(D, n_max, D.n_conv);
    if (T > Tmax_out) return w2_err(ctx, "wx_w2v_emissions: Tmax_out too small");
    if (int rc = w2_reserve(ctx, S, n_max)) return rc;
    // strides follow the RESERVED capacity, not this call's n_max
    const long cap_n = (long)ctx->cap_n;
    W2_CHECK(launch_set_ints(ctx->d_nf0, nf0.data(), S, s));     // by value: the vectors die with this call
    W2_CHECK(launch_set_ints(ctx->d_lens, lens.data(), S, s));
    int Tl[9];
    Tl[0] = 0;
    for (int i = 0; i < D.n_conv; ++i) Tl[i + 1] = frames_after(D, n_max, i + 1);
    long Tcap[9];
    for (int i = 0; i < D.n_conv; ++i) Tcap[i + 1] = frames_after(D, cap_n, i + 1);
    {
        W2vConv0Args a{pcm, pcm_stride, ctx->d_nf0, ctx->conv0w, ctx->gng, ctx->gnb, ctx->stats, ctx->act[0], C,
                       (int)Tcap[1], D.conv_kernel[0], D.conv_stride[0]};
        // only the first Tl[1] frames are needed; Tmax doubles as the row stride of the buffer
        if (D.norm_mode == 0) {
            W2_CHECK(launch_w2v_conv0(a, S, s));
        } else {
            a.gamma = ctx->convlng[0];
            a.beta = ctx->convlnb[0];
            W2_CHECK(launch_w2v_conv0_ln(a, ctx->convb[0], S, s));
        }
    }
    for (int i = 1; i < D.n_conv; ++i) {
        GemmArgs g{};
        g.X = ctx->convw[i]; g.ldx = (long)D.conv_kernel[i] * C; g.RX = C;
        g.Y = ctx->act[i - 1]; g.ldy = (long)D.conv_stride[i] * C; g.strideY = Tcap[i] * C; g.RY = Tl[i + 1];
        g.K = D.conv_kernel[i] * C;
        g.out = ctx->act[i]; g.ldo = C; g.strideOut = Tcap[i + 1] * C;
        if (D.norm_mode == 0) {
            W2_CHECK(launch_gemm_f16(g, S, true, s));
        } else {   // conv + bias, then LayerNorm over channels + GELU in place (whole buffer: rows are independent)
            g.bias = ctx->convb[i];
            W2_CHECK(launch_gemm_f16(g, S, false, s));
            W2_CHECK(launch_layernorm(ctx->act[i], C, ctx->convlng[i], ctx->convlnb[i], ctx->act[i], C,
                                      (int)((S - 1) * Tcap[i + 1] + Tl[i + 1]), C, s, 1));
        }
    }
    const long Tc = Tcap[D.n_conv];
    const int half = D.pos_kernel / 2;
    // rows of padded segments are independent: when the batch fills its capacity one launch covers all segments
    auto ln_all = [&](const h16* in, const h16* g, const h16* b, h16* out, int width) -> hipError_t {
        if (T == Tc) return launch_layernorm(in, width, g, b, out, width, S * T, width, s);
        for (int sb = 0; sb < S; ++sb) {
            hipError_t e = launch_layernorm(in + sb * Tc * width, width, g, b, out + sb * Tc * width, width, T, width, s);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    const long hp_stride = (Tc + 2 * half) * d;
    // feature projection: LN(C) -> Linear(C -> d), written into the zero-padded positional-conv input
    W2_CHECK(ln_all(ctx->act[D.n_conv - 1], ctx->fplng, ctx->fplnb, ctx->feat, C));
    {
        GemmArgs g = rowmajor(ctx->fpw, d, C, ctx->feat, C, T, ctx->fpb, nullptr, 0, ctx->hp + (long)half * d, d);
        g.strideY = Tc * C;
        g.strideOut = hp_stride;
        W2_CHECK(launch_gemm_f16(g, S, false, s));
    }
    // zero everything from each segment's length up to the buffer CAPACITY: rows past this call's T can hold
    // a previous, longer batch
    W2_CHECK(launch_w2v_mask_rows(ctx->hp, hp_stride, half, (int)Tc, d, ctx->d_lens, S, s));
    // grouped positional conv (k = pos_kernel, zero padded) + bias + GELU, + residual
    const int cg = d / D.pos_groups;
    {
        // ONE launch for all groups (two-level batch: z1 = segment, z2 = group): 16 launches of 768 blocks each left a
        // quarter of their last round of blocks empty
        GemmArgs g{};
        g.X = ctx->posw; g.ldx = (long)D.pos_kernel * cg; g.RX = cg;
        g.Y = ctx->hp; g.ldy = d; g.strideY = hp_stride; g.RY = T;
        g.K = D.pos_kernel * cg;
        g.y_gather_group = cg / 8; g.y_gather_step = d;
        g.bias = ctx->posb;
        g.R = ctx->hp + (long)half * d; g.ldr = d; g.strideR = hp_stride;
        g.out = ctx->x; g.ldo = d; g.strideOut = Tc * d;
        g.zsplit = S;
        g.strideX2 = (long)cg * D.pos_kernel * cg; g.strideY2 = cg; g.strideBias2 = cg; g.strideR2 = cg; g.strideOut2 = cg;
        W2_CHECK(launch_gemm_f16(g, S * D.pos_groups, true, s));
    }
    if (!D.stable_ln) W2_CHECK(ln_all(ctx->x, ctx->enclng, ctx->enclnb, ctx->x, d));
    const long Tpad = (Tc + 63) / 64 * 64;
    for (int i = 0; i < D.layers; ++i) {
        const W2vLayer& L = ctx->layers[i];
        const h16* ain = ctx->x;
        if (D.stable_ln) {
            W2_CHECK(ln_all(ctx->x, L.ln1g, L.ln1b, ctx->hbuf, d));
            ain = ctx->hbuf;
        }
        GemmArgs q = rowmajor(L.qkw, 2 * d, d, ain, d, T, L.qkb, nullptr, 0, ctx->qk, 2 * d);
        q.strideY = Tc * d;
        q.strideOut = Tc * 2 * d;
        q.xscale = ATTN_QSCALE; q.xscale_cols = d;      // the Q half scaled for the attention kernel (AttnArgs::q_prescaled)
        W2_CHECK(launch_gemm_f16(q, S, false, s));
        GemmArgs v{};
        v.X = ain; v.ldx = d; v.strideX = Tc * d; v.RX = T;
        v.Y = L.vw; v.ldy = d; v.RY = d; v.K = d;
        v.bias = L.vb; v.bias_on_y = 1;
        v.out = ctx->vt; v.ldo = Tpad; v.strideOut = (long)d * Tpad;
        W2_CHECK(launch_gemm_f16(v, S, false, s));
        AttnArgs at{ctx->qk, 2L * d, Tc * 2 * d, ctx->qk + d, 2L * d, Tc * 2 * d, ctx->vt, Tpad, (long)d * Tpad,
                    ctx->a, (long)d, Tc * d, ctx->d_lens, T, H, S};
        at.q_prescaled = 1;
        W2_CHECK(launch_attention(at, s));
        GemmArgs o = rowmajor(L.ow, d, d, ctx->a, d, T, L.ob, ctx->x, d, ctx->x, d);
        o.strideY = Tc * d; o.strideR = Tc * d; o.strideOut = Tc * d;
        W2_CHECK(launch_gemm_f16(o, S, false, s));
        const h16* fin = ctx->x;
        if (D.stable_ln) {
            W2_CHECK(ln_all(ctx->x, L.ln2g, L.ln2b, ctx->hbuf, d));
            fin = ctx->hbuf;
        } else {
            W2_CHECK(ln_all(ctx->x, L.ln1g, L.ln1b, ctx->x, d));
        }
        GemmArgs f1 = rowmajor(L.fc1w, D.ffn, d, fin, d, T, L.fc1b, nullptr, 0, ctx->f, D.ffn);
        f1.strideY = Tc * d; f1.strideOut = Tc * D.ffn;
        W2_CHECK(launch_gemm_f16(f1, S, true, s));
        GemmArgs f2 = rowmajor(L.fc2w, d, D.ffn, ctx->f, D.ffn, T, L.fc2b, ctx->x, d, ctx->x, d);
        f2.strideY = Tc * D.ffn; f2.strideR = Tc * d; f2.strideOut = Tc * d;
        W2_CHECK(launch_gemm_f16(f2, S, false, s));
        if (!D.stable_ln) W2_CHECK(ln_all(ctx->x, L.ln2g, L.ln2b, ctx->x, d));
    }
    if (D.stable_ln) W2_CHECK(ln_all(ctx->x, ctx->enclng, ctx->enclnb, ctx->x, d));
    W2_CHECK(launch_w2v_lmhead(ctx->x, (long)Tc * d, ctx->lmw, ctx->lmb, logp_out, (long)Tmax_out * D.vocab, S, T, d, D.vocab, s));
    return 0;
}

int wx_w2v_ctc_align(wx_w2v* ctx, const float* logp, const int32_t* T, const int32_t* tokens, const int32_t* N, int S,
                     int Tmax, int Nmax, int V, int blank_id, int beam, int32_t* path_tok, float* path_score,
                     int32_t* ok, float* trellis_out, void* stream) {
    if (!ctx) return -2;
    if (S < 1 || Tmax < 1 || Nmax < 1 || V < 2) return w2_err(ctx, "wx_w2v_ctc_align: bad shape");
    (void)hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t n_tr = trellis_out ? 0 : (size_t)S * Tmax * Nmax;
    const size_t n_w = (size_t)S * Tmax, n_bp = (size_t)S * (Tmax + 1) * 8;
    const size_t need = (n_tr + n_w) * sizeof(float) + n_bp * (2 * sizeof(int) + sizeof(float)) + 256;
    if (need > ctx->ctc_scratch_bytes) {
        W2_CHECK(hipStreamSynchronize(s));
        if (ctx->ctc_scratch) (void)hipFree(ctx->ctc_scratch);
        ctx->ctc_scratch = nullptr;
        ctx->ctc_scratch_bytes = 0;
        W2_CHECK(hipMalloc(&ctx->ctc_scratch, need));
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
    W2_CHECK(launch_ctc(a, s));
    return 0;
}

}  // extern "C"
