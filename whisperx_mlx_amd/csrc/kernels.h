// Internal launch interfaces between the C-ABI layer (api.hip) and the kernels.
#pragma once
#include "common.h"

// ---- gemm.hip ---------------------------------------------------------------
struct GemmArgs {
    const h16* X; long ldx; long strideX; int RX;   // MFMA A role: out's contiguous dim
    const h16* Y; long ldy; long strideY; int RY;   // MFMA B role: out's row dim
    int K;                                          // multiple of 8
    const h16* bias; long strideBias; int bias_on_y;
    const h16* R; long ldr; long strideR;           // residual, indexed like out (may alias out)
    h16* out; long ldo; long strideOut;             // out[y*ldo + x]
    int hs_T, hs_H, hs_d; long hs_part_stride;     // hs_T > 0: head-split store (needs RX % 4 == 0, full tiles in x)
    int y_gather_group; long y_gather_step;         // >0: Y's K axis = taps of `group` 16-B chunks, `step` elements apart
    // two-level batch (128 x 128 kernel only): zsplit > 0 -> blockIdx.z = z2 * zsplit + z1; z1 takes the stride* fields above,
    // z2 the *2 fields (the 16 groups of wav2vec2's positional conv x the segments of the batch as ONE launch)
    int zsplit = 0; long strideX2 = 0, strideY2 = 0, strideBias2 = 0, strideR2 = 0, strideOut2 = 0;
    int max_blocks = 0;                             // > 0 (256 x 256 kernel only): at most this many blocks, each walking tiles --
                                                    // a launch confined to that many CUs (a block owns its CU); multiple of 8;
                                                    // < 0: the one-tile-per-block kernel instead of the tile-pipelined one
    // out columns x < xscale_cols (a multiple of 8) are multiplied by xscale after the bias, before the one rounding to fp16:
    // the attention scale d_head^-0.5 * log2(e) goes into the Q half of a Q|K projection this way (attention.hip)
    float xscale = 1.f; int xscale_cols = 0;
    // tile-pipelined kernel: blocks start up to this many 10 ns ticks apart (block b waits b / grid of it), so that the CUs'
    // epilogues -- 128 KiB of stores each -- do not all hit the fabric in the same microseconds; 0: all start together
    int stagger_ticks = 0;
    int nt_stores = 0;                              // tile-pipelined kernel: non-temporal output stores
    int general_epilogue = 0;                       // tile-pipelined kernel: 1 = the general epilogue on interior tiles too (lab A/B)
#ifdef WX_LAB_ENV
    unsigned long long* lab_stamps = nullptr;       // lab: s_memrealtime stamps of blocks 0 / 128, waves 0 / 4 / 7 along their tiles
#endif
};
hipError_t launch_gemm_f16(const GemmArgs& a, int batch, bool gelu, hipStream_t s);

// Skinny GEMM for decode (<=16 activation rows): out[m][n] = epi(sum_k A[m][k] W[n][k]).
struct SkinnyArgs {
    const h16* A; long lda;            // [16][K] activations (rows >= M are ignored)
    const h16* W; long ldw;            // [N][K]
    const unsigned char* Wq;           // non-null: int8 weights stored as q + 128, [N][K] bytes (ldw elements per row) ...
    const float* wscale;               // ... with one fp32 scale per output row: w = (byte - 128) * wscale[n]
    const h16* bias;                   // [N] or null
    const h16* ln_g; const h16* ln_b;  // if non-null: A := LayerNorm(A) over K (K == row length)
    const h16* R; long ldr;            // residual [16][N] or null (may alias out_h)
    h16* out_h; float* out_f; long ldo;  // exactly one of out_h/out_f
    int M, N, K;
    int gelu;
    int tile_n;                        // 0/16, 8 or 4 output columns per block (more blocks for small N)
    int wide_block;                    // K > 10*8*32: use 16 waves per block instead of 20 k-steps per wave
    // k-blocked activation layout [k / 32][16 rows][32] (skinny_kernel only, M <= 16): a wave's 16-row x 64-byte fragment
    // load is then ONE contiguous KiB instead of 16 half-used cache lines
    int out_blocked;                   // write out_h in that layout (the producer: LN+FC1)
    int a_blocked;                     // read A in that layout (the consumer: FC2)
    // W / Wq in the tile-blocked layout [N / 16][K / 32][16 rows][32] (launch_pack_gemv_weight; N % 16 == 0, ldw ignored):
    // the 16 rows x 64 bytes an MFMA weight fragment load takes are then ONE contiguous KiB (512 bytes for int8) instead
    // of 16 pieces 2 * K bytes apart.  skinny_kernel / skinny_vw2_kernel / the fused decode launch only.
    int w_blocked;
    // more than 16 rows: skinny_wide_kernel walks four row groups per block with the weights in registers (one pass over the
    // weights instead of one per group; same bits per row).  A fused LayerNorm then runs as a launch of its own into
    // `ln_scratch` (M rounded up to 16, x K halves, k-blocked); without it LN launches keep the row-group kernels.
    h16* ln_scratch;
    int no_wide;                       // 1: keep the row-group kernels whatever the rows; -1: the one-pass kernel from 17 rows on
                                       // (0: from 65 rows on, where it pays).  Tests hold the two against each other
#ifdef LAB_DUMP_Q8                    // lab builds only (tools/build_lab.py, tools/dbg_q8.py): the int8 epilogue's operands of one element
    float* lab_dump; int lab_slot;
#endif
    // optional launch timer of the launch IN FRONT of this one (the fused decode launch notes its start in prof[0]): the
    // first block adds "now - prof[0]" to prof[1], counts it in prof[2] and clears prof[0].  Null: off.
    unsigned long long* prof;
};
hipError_t launch_skinny(const SkinnyArgs& a, hipStream_t s);
// row-major [N][K] (elem_bytes 2: fp16, 1: int8 bytes) -> the tile-blocked layout of SkinnyArgs::w_blocked; N % 16 == 0, K % 32 == 0
hipError_t launch_pack_gemv_weight(const void* w, void* out, int N, int K, int elem_bytes, hipStream_t s);
// M <= 64 rows, tile_n chosen as ceil(N / n_cu): one balanced round of blocks (see skinny.hip)
hipError_t launch_skinny_mt(const SkinnyArgs& a, int n_cu, hipStream_t s);

// v2 decode GEMV: split-K over blocks, no LayerNorm prologue (see skinny.hip)
struct Skinny2Args {
    const h16* A; long lda;            // [16][K] activations
    const h16* W; long ldw;            // [N][K]
    const h16* bias;                   // ksplit == 1 only
    h16* out_h; float* out_f; long ldo;  // ksplit == 1: fp16 or fp32 output
    float* part; long ldp;             // ksplit > 1: fp32 partial tiles [ksplit][16][ldp]
    int M, N, K, ksplit, gelu;
    // optional (tile-walking kernel only: ksplit 1, M <= 16, K <= 1280): A := LayerNorm(A) over K, once per block
    const h16* ln_g; const h16* ln_b;
};
hipError_t launch_skinny2(const Skinny2Args& a, hipStream_t s);
bool skinny2_can_fuse_ln(int M, int N, int K);   // launch_skinny2 would take ln_g / ln_b for this shape
struct ResLnArgs {
    h16* x;                            // [M][d] residual stream (updated in place)
    const float* part; long ldp; int ksplit;   // partial tiles of the producing GEMV (may be 0)
    const h16* bias;                   // producing GEMV's bias (nullable)
    const int* tokens; int tok_ld; const int* d_pos; const h16* emb; const h16* pos;   // embedding mode
    const h16* g; const h16* b; h16* xn;       // LayerNorm -> xn (xn null: residual update only)
    int d;
};
hipError_t launch_resln(const ResLnArgs& a, int M, hipStream_t s);
// y = LayerNorm(x) for M rows with the arithmetic of the logits kernel's fused prologue (bit-identical rows), K <= 1280
hipError_t launch_ln_rows16(const h16* x, long ldx, const h16* g, const h16* b, h16* y, long ldy, int M, int K, hipStream_t s);

// ---- elementwise.hip ----------------------------------------------------------
hipError_t launch_layernorm(const h16* x, long ldx, const h16* g, const h16* b, h16* y, long ldy,
                            int rows, int d, hipStream_t s, int gelu = 0);
// Small host integer arrays reach the device as KERNEL ARGUMENTS (32 per launch), never as an asynchronous copy from
// caller-owned or stack memory: such a copy may read the host buffer after the call has returned.
hipError_t launch_set_ints(int* dst, const int* host_vals, int n, hipStream_t s);
hipError_t launch_embed(const int* tokens, int tok_ld, const int* d_pos, const h16* emb, const h16* pos,
                        h16* x, int B, int d, hipStream_t s);

// ---- logmel.hip -----------------------------------------------------------------
struct LogmelArgs {
    const float* pcm; long pcm_stride;  // [B][pcm_stride] f32, n_valid[b] samples valid, rest treated as 0
    const int* n_valid;
    const float* filters;               // [n_mels][201]
    const int* filt_lo; const int* filt_len;
    const float* twiddle;               // [400][2] cos,sin
    const float* window;                // [400]
    float* logspec;                     // [B][3000][n_mels] f32 scratch (log10, pre-clamp)
    unsigned* chunk_max;                // [B] ordered-uint max
    int B, n_mels;
};
hipError_t launch_logmel(const LogmelArgs& a, hipStream_t s);
hipError_t launch_logmel_finalize(const float* logspec, const unsigned* chunk_max, float* out_f32,
                                  h16* out_h, int out_h_ld, int out_h_rows, int B, int n_mels, hipStream_t s);

// ---- attention.hip ----------------------------------------------------------------
// Full (non-causal) self attention over T keys, d_head 64:  O = softmax(Q K^T / 8) V
struct AttnArgs {
    const h16* Q; long ldq; long strideQ;      // Q[b][t][h*64 + d]
    const h16* K; long ldk; long strideK;      // K[b][t][h*64 + d]
    const h16* VT; long ldvt; long strideVT;   // VT[b][h*64 + d][t], ldvt >= round_up(T,64), zero padded
    h16* O; long ldo; long strideO;            // O[b][t][h*64 + d]
    const int* lens;                           // optional per-batch valid length (null -> T)
    int T, H, B;
    int max_blocks = 0;                        // > 0: at most this many blocks, each walking (batch, head, query tile) units
    int q_prescaled = 0;                       // 1: Q already carries d_head^-0.5 * log2(e) (GemmArgs::xscale of the projection that wrote it)
#ifdef WX_LAB_ENV
    unsigned long long* lab_stamps = nullptr;  // lab: s_memtime stamps along the key tiles of a few blocks (tools/lab_attn_timeline.py)
#endif
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
constexpr float ATTN_QSCALE = 0.125f * 1.44269504088896340736f;   // d_head^-0.5 * log2(e): what AttnArgs::q_prescaled means

struct DecSelfAttnArgs {
    const h16* q; long ldq;          // [B][d] (this step's query, bias included)
    h16* kc; h16* vc;                // caches [B][n_ctx][d]; this step's k,v already appended at *d_pos
    long cache_stride;               // n_ctx * d
    h16* out; long ldo;              // [B][d]
    const int* d_pos;                // device scalar: position of the current token
    int B, H, d;
    int out_blocked;                 // write `out` k-blocked ([n/32][16][32]) for the following GEMV (B <= 16)
    const int* done;                 // optional [B]: rows that have emitted EOT take no part any more (their blocks return at once)
};
hipError_t launch_dec_self_attn(const DecSelfAttnArgs& a, const h16* knew, const h16* vnew, long ldnew, hipStream_t s);

struct DecCrossAttnArgs {
    const h16* q; long ldq;          // [B][d]  (or null when the query comes from partial tiles)
    const float* q_part; long q_ldp; int q_ksplit; const h16* q_bias;   // [ksplit][16][q_ldp] + bias[d]
    const h16* K; long ldk; long strideK;     // K[b][t][h*64+d] (hstride 64) or [b][h][t][64] (hstride T*64)
    const h16* V; long ldv; long strideV;
    long hstride;                             // element offset between heads
    unsigned* tickets;                        // [B][H] zeroed counters: merge the key splits in-launch (null: combine kernel)
    h16* out; long ldo;
    float* qk_out;                   // optional capture buffer [B][n_heads_cap][n_rows][T]
    const int* cap_slot;             // [H] -> capture slot or -1 (for this layer)
    int n_cap; int cap_rows;
    const int* d_row;                // device scalar: capture row for this step (<0: no capture)
    int B, H, T;
    int online;                      // 1: single pass (online softmax per wave) instead of the two-pass body
    // tagged-granule merge of the key splits (preferred over tickets): [B][H][nsplit][66] 8-byte {f32, tag} words
    unsigned long long* gran; const int* d_pos; const unsigned* d_epoch; int layer; int* d_err;
    int out_blocked;                 // as DecSelfAttnArgs::out_blocked
    const int* done;                 // as DecSelfAttnArgs::done: a finished row's K / V are not streamed any more
};
hipError_t launch_dec_cross_attn(const DecCrossAttnArgs& a, int nsplit, float* part, hipStream_t s, int threads = 256);

// ---- declayer.hip: dependent stages of a decoder layer in one launch (granule hand-off, see the file header) ----------
// [LayerNorm + cross-Q GEMV `g`] -> [cross attention `a`, both key splits of a (row, head) in one block]; gq: [rows][N/2]
// 8-byte granules.  Results are bit-identical to launch_skinny(g) + launch_dec_cross_attn(a, 2 splits, 256 threads).
bool dec_cq_xattn_supported(const SkinnyArgs& g, const DecCrossAttnArgs& a);
hipError_t launch_dec_cq_xattn(const SkinnyArgs& g, const DecCrossAttnArgs& a, unsigned long long* gq, hipStream_t s,
                               const unsigned long long* gq_poll = nullptr /* test hook: a buffer nobody publishes to */,
                               int* n_selfq = nullptr /* counter of attention blocks that computed their query themselves */,
                               bool q_in_memory = false /* attention role only: the query is in a.q (a GEMV launch ran in front) */,
                               unsigned long long* prof_slot = nullptr /* launch timer: block 0 notes the start here */);

// ---- sample.hip ---------------------------------------------------------------------
struct SampleArgs {
    const float* logits; long ldl;   // [B][ldl]
    int* tokens; int tok_ld;         // [B][tok_ld]
    float* sum_logprob; float* no_speech_prob;
    const unsigned char* suppress;   // [n_vocab] 1 = always suppressed (SuppressTokens + notimestamps)
    const int* d_pos;                // device scalar: index of the last written token
    int B, n_vocab;
    int sample_begin;
    int eot, no_speech, timestamp_begin, blank0, blank1;
    int rules, max_initial_ts, forced_len;
    // Optional fused tail (emb != null): the row's block writes the NEXT step's input x[b] = emb[next] + pos[n] and the
    // last block to finish advances the position counters -- two dependent launches less per decode position.
    const h16* emb; const h16* decpos; h16* x; int d;
    int* d_pos_w; int* d_row; unsigned* ticket;      // ticket: zeroed counter, self-resetting
    // Optional row split (part != null): 4 blocks per row hand 8-float records to the last one of the row to finish
    float* part; unsigned* row_ticket;               // [B][4][8] scratch, [B] zeroed counters (self-resetting)
    int* done;                                       // optional [B]: set to 1 when the row's newest token is EOT (read by the attention kernels)
    const int* forced_lens;                          // optional [B] (bench workload, with forced_len > 0): row b ends after this many tokens
    const int* n_active;                             // optional device scalar: rows >= *n_active are padding (they emit EOT at once and stay finished)
};
hipError_t launch_sample(const SampleArgs& a, hipStream_t s);
hipError_t launch_advance(int* d_pos, int* d_row, int sample_begin, hipStream_t s);

// ---- dtw.hip ---------------------------------------------------------------------------
struct DtwArgs {
    const float* qk;          // [B][n_cap][rows][T]
    const int* tokens; int tok_ld; int sample_begin;   // token history (decides which rows are text)
    const int* n_frames;      // optional [B]: encoder frames that carry audio (null -> T)
    float* work;              // [B][rows+1][T] alignment matrix scratch
    float* work2;             // [B][n_cap][rows+1][T] per-head scratch
    unsigned char* trace;     // [B][trace_stride]
    int* rowmap;              // [B][rows+1] scratch: decode step of every kept row
    int* n_rows;              // out [B] number of kept rows (text tokens + EOT row)
    int* path_i; int* path_j; // out [B][path_stride] DTW path, stored end -> start
    int* path_len;            // out [B]
    long trace_stride, path_stride;   // per-sequence strides of trace / path_i / path_j
    int B, n_cap, rows, T, eot, mode;
    int n_sampled;            // sampled positions of the last decode (<= rows)
    float qk_scale;
};
hipError_t launch_dtw(const DtwArgs& a, hipStream_t s);
hipError_t launch_median7_rows(const float* x, long ldx, int rows, int T, float* y, long ldy, hipStream_t s);

// ---- ctc.hip -----------------------------------------------------------------------------
struct CtcArgs {
    const float* logp; long seg_stride; int V;   // [S][Tmax][V] log-probs
    const int* T;                                // [S] frames
    const int* tokens; int Nmax; const int* N;   // [S][Nmax] (-1 = wildcard)
    float* trellis;                              // [S][Tmax][Nmax] scratch/out
    float* wild;                                 // [S][Tmax] scratch
    int* bp_tok; int* bp_par; float* bp_prob;    // [S][Tmax+1][8] scratch
    int* path_tok; float* path_score; int* ok;   // [S][Tmax], [S][Tmax], [S]
    int S, Tmax, blank, beam;
    int bp_in_lds;                               // set by launch_ctc: beam records fit in LDS
};
hipError_t launch_ctc(const CtcArgs& a, hipStream_t s);

// ---- w2v.hip -----------------------------------------------------------------------------
constexpr int W2V_CONV0_STATS_PER_SEGMENT = 8 * 65;      // 8 blocks x (10 sums + 55 products) doubles
struct W2vConv0Args {
    const float* pcm; long pcm_stride;     // [S][pcm_stride] f32, zero padded
    const int* n_frames;                   // [S] valid conv0 frames per segment
    const float* w;                        // [C][10] f32
    const h16* gamma; const h16* beta;     // GroupNorm affine [C]
    double* stats;                         // [S][W2V_CONV0_STATS_PER_SEGMENT]: partial sums of the signal statistics (w2v.hip)
    h16* out;                              // [S][Tmax][C]
    int C, Tmax, kernel, stride;
};
hipError_t launch_w2v_conv0(const W2vConv0Args& a, int S, hipStream_t s);
// "layer" feature-encoder variant: conv0 + bias -> LayerNorm over channels -> GELU (a.stats unused)
hipError_t launch_w2v_conv0_ln(const W2vConv0Args& a, const h16* bias, int S, hipStream_t s);
hipError_t launch_w2v_mask_rows(h16* x, long seg_stride, long row0, int Tmax, int d, const int* lens, int S, hipStream_t s);
hipError_t launch_w2v_lmhead(const h16* x, long x_seg, const h16* w, const h16* bias, float* logp, long logp_seg, int S, int rows,
                             int d, int V, hipStream_t s);
