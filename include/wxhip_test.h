/* wxhip_test.h -- test and measurement hooks of libwxhip.so.  NOT part of the drop-in boundary (include/wxhip.h):
 * nothing the reference calls is replaced by these.  They expose single kernels of the hot path -- the same code the
 * entry points of wxhip.h launch -- so that tests/ can compare each of them with the oracle through the C ABI, and so
 * that bench.py can time one kernel with HIP events on the context's own stream. */
#ifndef WXHIP_TEST_H
#define WXHIP_TEST_H
#include "wxhip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* wx_decode_opts.tuning: how the decode loop is launched and measured -- never WHAT it computes (every setting gives the
 * same tokens and log-probabilities; step_variant 2 / 3 are older GEMV forms kept for A/B runs and agree within rounding).
 * A zero-initialised struct is NOT the default: use WX_TUNING_DEFAULTS (what a NULL pointer means). */
typedef struct wx_tuning {
    int use_graph;              /* replay the decode step as a hipGraph                       */
    int check_every;            /* host polls the all-done flag every N steps (0 = never)     */
    int cross_split;            /* key split of the cross-attention kernel (1,2,4)            */
    int step_variant;           /* 0/4 = fused launches (dependent stages of a layer share a launch, csrc/declayer.hip) over the LayerNorm-fused GEMVs; 1 = one kernel per stage (more than 16 rows: over groups of 16 rows); 2 = split-K GEMVs + resln; 3 = M-tiled GEMVs; 5 = lab: the cross-Q GEMV as a launch of its own, then only the attention role of the fused kernel.  0, 1, 4 and 5 give identical tokens and log-probabilities, with fp16 and with int8 weights */
    int fc2_tile_n;             /* output columns per block of the N = d GEMVs (output projections, FC2): 0/8 = 160 blocks (fastest
                                   alone), 16 = 80 fat blocks that leave CUs to the other passes in flight; same tokens */
    int profile_launches;       /* != 0: the fused decode launches are timed on the device -- their first block notes its start on the
                                   constant 100 MHz clock, the first block of the launch behind it (the output projection) notes
                                   its own start, i.e. the fused launch's end plus the dispatch gap, and adds the difference up --
                                   and wx_launch_profile returns the average: how bench.py measures the dominant kernel LIVE,
                                   inside the timed region, whatever stream and hipGraph the launch is part of.  0: off */
    int max_steps_ahead;        /* > 0: the host thread inside wx_decode_greedy stays at most about this many decode steps ahead of the
                                   GPU (it waits on an event recorded that many steps back).  A free-running loop enqueues a whole
                                   pass -- ~35 000 kernel nodes -- before the first step has finished; launcher threads that have
                                   nothing else to do lose nothing by waiting, and the queues stay short (rocprofv3's kernel trace
                                   crashed on three 128-row passes enqueued that far ahead).  0: never wait */
} wx_tuning;
#define WX_TUNING_DEFAULTS {1, 8, 2, 0, 0, 0, 0}

/* one sampling step on caller-provided logits (f32 [B][ldl]) and token history
 * (int32 [B][tok_ld], n_tokens already written): the filter + greedy kernel of
 * wx_decode_greedy in isolation (BatchGreedyDecoder.update, batch_decoder.py:267-303). */
int wx_sample_step(wx_ctx* ctx, const float* logits, long ldl, int32_t* tokens, int tok_ld, int n_tokens,
                   int B, const wx_decode_opts* opts, float* sum_logprob, float* no_speech_prob, void* stream);

/* copy of the captured alignment-head scores: f32 [B][n_heads][sample_len][1500] */
int wx_get_align_qk(wx_ctx* ctx, int B, float* qk_out, void* stream);

/* measurement hook for bench.py: launches one hot kernel `iters` times with the
 * context's own resident operands (0 decode cross-attention, 1 encoder FC1 GEMM,
 * 2 encoder attention, 3 decode LN+QKV, 4 decode FC2, 5 logits, 6 encoder FC2 GEMM, 13 the fused
 * [LN + cross-Q GEMV -> cross attention] launch of the default decode step);
 * the caller brackets the call with HIP events on `stream`. */
int wx_probe(wx_ctx* ctx, int kind, int B, int iters, int arg, void* stream);

/* measurement hook for bench.py: synchronises `stream`, reads and clears the launch timer of the fused decode launches
 * (wx_tuning.profile_launches): *avg_us = average duration of a launch since the last call -- from its first block's start
 * to the start of the launch behind it, on the device's constant 100 MHz clock --, *n_launches = how many were timed. */
int wx_launch_profile(wx_ctx* ctx, double* avg_us, long long* n_launches, void* stream);

/* the forward-progress guarantee of the fused decode launch (csrc/declayer.hip) exercised on purpose: its attention
 * blocks poll a granule buffer nobody publishes to, so EVERY one of them computes its query itself after the short
 * poll.  out_fused [B][d] f16 = that launch's output on the context's resident operands (layer 0, the state the last
 * decode left); out_ref [B][d] f16 = the two launches it stands for (LayerNorm + cross-Q GEMV, cross attention with two
 * key splits) on the same operands: the test requires them bit-identical.  *n_selfq_host = blocks that took the path
 * (B * n_text_head).  n_selfq_host == NULL: the fused launch exactly as the decode step issues it (its blocks poll the
 * buffer the producers publish to), against the same two launches. */
int wx_test_fused_selfq(wx_ctx* ctx, int B, void* out_fused, void* out_ref, int* n_selfq_host, void* stream);

/* LAB (round 4): confines wx_encode's GEMM and attention launches to at most `max_blocks` compute units (a 256 x 256 GEMM
 * block owns its CU; persistent blocks walk the tiles) -- the encoder on a partition of the chip beside other contexts'
 * decode.  A multiple of 8 (the XCD-aware tile order is kept); 0 = no cap (the product).  -1: every GEMM on the
 * one-tile-per-block kernel instead of the tile-pipelined one (csrc/gemm.hip gemm_pipe_kernel; the tests hold the two
 * against each other).  Results are bit-identical in every setting.  Also applies to wx_gemm_f16. */
int wx_set_encoder_cap(wx_ctx* ctx, int max_blocks);

/* raises the context's device-side error flag on `stream`, as a decode kernel whose bounded wait for another key split
 * expired does (attention.hip: dec_cross_attn_kernel, step variant 1) -- so that the host's recovery (wx_device_status
 * reports it, WhisperHipBackend decodes the job again without key splits) can be exercised on purpose, in the middle of a
 * job, through the product path.  The product never calls it. */
int wx_test_raise_device_flag(wx_ctx* ctx, void* stream);

/* the width-7 reflect-padded running median of the DTW pre-processing (dtw.hip: `median7` / `reflect`, the device
 * functions dtw_median_mean_kernel and dtw_inrepo_row_kernel call) on a plain f32 matrix [rows][T], T >= 4:
 * median_filter_fixed, /root/reference/median_filter_fix.py:6-21 */
int wx_median7_rows(wx_ctx* ctx, const float* x, long ldx, int rows, int T, float* y, long ldy, void* stream);

/* ---- building blocks (the kernels the hot path launches) ------------------------------------------------ */
int wx_gemm_f16(wx_ctx* ctx, const void* X, long ldx, int RX, const void* Y, long ldy, int RY, int K,
                const void* bias, int bias_on_y, const void* R, long ldr, void* out, long ldo,
                int gelu, void* stream);
int wx_skinny_f16(wx_ctx* ctx, const void* A, long lda, int M, const void* W, long ldw, int N, int K,
                  const void* bias, const void* ln_g, const void* ln_b, const void* R, long ldr,
                  void* out_h, float* out_f, long ldo, int gelu, int tile_n /* 0 = 16; 1..16 columns per block */,
                  void* stream);
/* M-tiled (M <= 64), column-balanced decode GEMV: ceil(N / n_cu) columns per block (n_cu <= 0: the device's CU count) */
int wx_skinny_mt_f16(wx_ctx* ctx, const void* A, long lda, int M, const void* W, long ldw, int N, int K,
                     const void* bias, const void* ln_g, const void* ln_b, const void* R, long ldr,
                     void* out_h, float* out_f, long ldo, int gelu, int n_cu, void* stream);
/* the same GEMVs with int8 weights: Wq[n][k] = q + 128 (bytes), w = (Wq - 128) * wscale[n]; dequantised in
 * registers, fp16 activations, fp32 accumulation (reference spec: symmetric scale, dequantise then
 * float matmul, whisperx/backends/mlx_quantization.py:132-168).  balanced != 0: the M-tiled kernel; else more than 16 rows run over groups of 16 rows. */
int wx_skinny_q8(wx_ctx* ctx, const void* A, long lda, int M, const void* Wq, const float* wscale, long ldw, int N, int K,
                 const void* bias, const void* ln_g, const void* ln_b, const void* R, long ldr,
                 void* out_h, float* out_f, long ldo, int gelu, int balanced, void* stream);
/* the tile-blocked weight layout the decode step streams ([N / 16][K / 32][16][32]: one contiguous KiB -- 512 bytes for
 * int8 -- per MFMA fragment load; wx_finalize keeps such copies of the six decode GEMV weights of every layer):
 * wx_pack_gemv_weight writes it for a row-major [N][K] matrix (elem_bytes 2 = fp16, 1 = int8 bytes; N % 16 == 0,
 * K % 32 == 0); wx_skinny_ex is wx_skinny_f16 (W) or wx_skinny_q8 (Wq + wscale; exactly one of W / Wq) with every
 * option of the decode step's launches: w_blocked != 0 = the weights are in that layout (ldw ignored), wide_block != 0 =
 * the K = 4d forms (sixteen k-slices: 16 waves, or eight waves taking two slices each beyond 16 rows) */
int wx_pack_gemv_weight(wx_ctx* ctx, const void* w, int N, int K, int elem_bytes, void* out, void* stream);
int wx_skinny_ex(wx_ctx* ctx, const void* A, long lda, int M, const void* W, const void* Wq, const float* wscale, long ldw, int N, int K,
                 const void* bias, const void* ln_g, const void* ln_b, const void* R, long ldr,
                 void* out_h, float* out_f, long ldo, int gelu, int tile_n, int wide_block, int w_blocked, void* stream);
/* decode GEMV v2 (split-K over blocks; ksplit > 1 writes fp32 partials [ksplit][16][N]) and the
 * residual + LayerNorm kernel that consumes them */
int wx_skinny2_f16(wx_ctx* ctx, const void* A, long lda, int M, const void* W, long ldw, int N, int K,
                   const void* bias, int ksplit, int gelu, void* out_h, float* out_f, long ldo, float* part,
                   void* stream);
/* the logits form of the same GEMV with the final LayerNorm fused: out_f[m][n] = sum_k LN(A)[m][k] W[n][k], M <= 16,
 * K <= 1280, N >= 32768 (the blocks walk several 16-column tiles and normalise the rows once each); returns an error
 * for other shapes */
int wx_skinny2_ln_f16(wx_ctx* ctx, const void* A, long lda, int M, const void* W, long ldw, int N, int K,
                      const void* ln_g, const void* ln_b, float* out_f, long ldo, void* stream);
int wx_resln_f16(wx_ctx* ctx, void* x, int M, int d, const float* part, int ksplit, const void* bias,
                 const void* g, const void* b, void* xn, void* stream);
int wx_layernorm_f16(wx_ctx* ctx, const void* x, long ldx, const void* g, const void* b, void* y, long ldy,
                     int rows, int d, void* stream);
int wx_attention_f16(wx_ctx* ctx, const void* Q, long ldq, long strideQ, const void* K, long ldk, long strideK,
                     const void* VT, long ldvt, long strideVT, void* O, long ldo, long strideO,
                     const int32_t* lens, int T, int H, int B, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WXHIP_TEST_H */
