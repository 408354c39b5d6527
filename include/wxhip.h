/* wxhip.h -- C ABI of libwxhip.so, the MI355X (gfx950) hot path behind
 * whisperx.load_model(..., backend="hip").transcribe() and whisperx.alignment.align().
 *
 * The reference (sooth/whisperx-mlx) is pure Python and has no FFI of its own; each
 * entry point below replaces the Python-level call the reference makes into
 * third-party mlx-whisper / torch at the cited line, so that a maintainer can bind
 * it with ctypes from the backend class (INTEGRATION.md shows the stub).
 *
 * Conventions: every function returns 0 on success, <0 on error
 * (wx_last_error(ctx) gives the message, owned by ctx).  No exceptions cross the
 * boundary.  All buffers are caller-owned DEVICE pointers (PyTorch-ROCm tensors'
 * data_ptr()) unless marked "host"; `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  One ctx per (device, model); a ctx is not thread-safe,
 * independent ctxs are.  Nothing here falls back to the CPU.
 */
#ifndef WXHIP_H
#define WXHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wx_ctx wx_ctx;

/* Whisper ModelDimensions (mlx-whisper `model.dims`, read at
 * whisperx/backends/mlx_lightning.py:163 `self.model.dims.n_mels`). */
typedef struct {
    int n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} wx_model_dims;

/* logit-filter rule bits for wx_decode_opts.rules */
enum {
    WX_RULE_SUPPRESS_BLANK = 1,   /* SuppressBlank (first sampled token)                 */
    WX_RULE_SUPPRESS_TOKENS = 2,  /* SuppressTokens: folded into suppress_mask by the host */
    WX_RULE_TS_NOTIMESTAMPS = 4,  /* <|notimestamps|> never sampled: folded into mask      */
    WX_RULE_TS_PAIRS = 8,         /* timestamps come in pairs                              */
    WX_RULE_TS_MONOTONE = 16,     /* timestamps never decrease                             */
    WX_RULE_TS_INITIAL = 32,      /* first token is a timestamp <= max_initial_ts          */
    WX_RULE_TS_PROB = 64,         /* mlx_ultra_optimized_batch.py:52-69                    */
    WX_RULES_LIGHTNING = 127,     /* DecodingOptions defaults, mlx_lightning.py:187-193    */
    WX_RULES_OPTIMIZED_FINAL = 2 | 64 /* mlx_whisper_optimized_final.py:301-306 + patched apply */
};

struct wx_tuning;

typedef struct {
    int prompt[8];              /* sot, <|lang|>, <|task|> [, <|notimestamps|>]  (host)      */
    int n_prompt;               /* = sample_begin                                            */
    int sample_len;             /* max sampled tokens, 224 (n_text_ctx // 2)                 */
    int rules;                  /* WX_RULE_* bits                                            */
    int max_initial_ts;         /* max_initial_timestamp index (50 = 1.0 s), <0 = none       */
    int forced_len;             /* >0: bench workload, EOT suppressed, exactly this many tokens */
    int eot, no_speech, timestamp_begin, blank0, blank1;
    const uint8_t* suppress_mask; /* device [n_vocab], 1 = never sample this id              */
    int capture_qk;             /* keep alignment-head cross-attention scores for wx_dtw_path */
    const int32_t* forced_lens; /* bench workload only, with forced_len > 0: device [B], row b ends (EOT) after forced_lens[b] <= forced_len
                                   sampled tokens -- the length distribution of real speech instead of one length for all; NULL = forced_len */
    int n_active;               /* 0 / >= B: every row is a chunk.  0 < n_active < B: rows n_active..B-1 are PADDING -- a pass cut to the
                                   context's one launch shape (hipGraphs are captured per row count: a scheduler that always launches
                                   B = batch_size rows never captures a second one).  Padding rows need no encoder output (enc_f16
                                   still spans B rows; only the first n_active are read), count as finished from the first sampled
                                   position on -- so the attention kernels skip them like any row that has emitted EOT -- and their
                                   outputs are not meaningful.  Read from device memory by the kernels: not part of a captured launch */
    const struct wx_tuning* tuning; /* NULL: the library's own configuration of the decode loop (hipGraph replay, fused launches, two
                                   key splits, the all-done flag polled every 8 steps).  Launch-shape, scheduling and measurement
                                   knobs -- nothing that changes a token -- live in `struct wx_tuning`, include/wxhip_test.h: a host
                                   that keeps several passes in flight (whisperx_mlx_amd/backend.py) and the labs set them */
} wx_decode_opts;

/* ---- lifecycle -------------------------------------------------------------------- */
/* replaces mlx_whisper.load_models.load_model (whisperx/backends/mlx_lightning.py:9,74):
 * creates the context; weights are bound afterwards from caller-owned fp16 tensors. */
/* A context is single-threaded: one launcher thread per context (use several contexts for several passes in flight).
 * The entry points that enqueue work refuse concurrent entry with an error. */
int wx_create(int device_id, const wx_model_dims* dims, int max_batch, wx_ctx** out);
void wx_destroy(wx_ctx* ctx);
const char* wx_last_error(wx_ctx* ctx);
/* name = canonical packed name (see whisperx_mlx_amd/weights.py), dptr = device fp16.  The six decode GEMV weights of a
 * decoder layer (dec.<l>.{qkv,o,cq,co,fc1,fc2}) may instead be bound as int8: "<base>.wq" = bytes q + 128 [N][K] and
 * "<base>.ws" = fp32 scale per output row */
int wx_bind_weight(wx_ctx* ctx, const char* name, const void* dptr, size_t nbytes);
/* checks that every weight the dims require is bound and allocates the workspace.
 * The decode step does NOT read the six GEMV weights of a decoder layer where the caller bound them: wx_finalize makes
 * library-owned copies in the tile-blocked layout the decode GEMVs stream ([out/16][in/32][16][32], ~1.5 GB fp16 per
 * context for large-v3).  So (a) after changing a bound decode weight in place, or re-binding it, call wx_finalize again
 * (it re-packs; the workspace is kept) -- without it the decode keeps using the old values; the copies are shared by
 * every context of the process that binds the same tensor, so re-finalize while none of them has a decode in flight;
 * (b) a re-bind must keep the
 * storage (fp16 stays fp16, int8 stays int8): a context is built for one of them and wx_finalize fails otherwise. */
int wx_finalize(wx_ctx* ctx);
/* alignment heads (model.alignment_heads, mlx_whisper_optimized_final.py:146):
 * host int pairs (layer, head) */
int wx_set_alignment_heads(wx_ctx* ctx, const int* layer_head, int n_heads);
/* host tables for the log-mel kernel: filters [n_mels*201] f32 (audio.py:94-109) */
int wx_set_mel_filters(wx_ctx* ctx, const float* filters_host, int n_mels);

/* ---- hot path --------------------------------------------------------------------- */
/* replaces log_mel_spectrogram + pad_or_trim per 30 s chunk
 * (mlx_whisper_optimized_final.py:428-434; torch statement whisperx/audio.py:112-159).
 * pcm: f32 [B][pcm_stride], n_valid[b] valid samples (rest treated as zero padding).
 * mel_f16: [B][3000][n_mels] channels-last (nullable); mel_f32: same in f32 (nullable). */
int wx_logmel(wx_ctx* ctx, const float* pcm, long pcm_stride, const int32_t* n_valid, int B,
              void* mel_f16, float* mel_f32, void* stream);

/* replaces model.encoder(mel) (DecodingTask._get_audio_features,
 * mlx_whisper_batch_decoder.py:403).  mel_f16 [B][3000][n_mels] -> enc_f16 [B][1500][d]. */
int wx_encode(wx_ctx* ctx, const void* mel_f16, int B, void* enc_f16, void* stream);

/* replaces BatchDecodingTask._main_loop_batch + BatchGreedyDecoder.update + the logit
 * filters (mlx_whisper_batch_decoder.py:267-303,317-384).  tokens_out int32 [B][n_text_ctx]
 * receives prompt + sampled tokens; n_steps_out (host) the number of sampled positions. */
int wx_decode_greedy(wx_ctx* ctx, const void* enc_f16, int B, const wx_decode_opts* opts,
                     int32_t* tokens_out, float* sum_logprob, float* no_speech_prob,
                     int* n_steps_out_host, void* stream);

/* last-position logits (f32 [B][n_vocab]) of a teacher-forced token prefix tokens int32 [B][n] (device), through the
 * step kernels of wx_decode_greedy without sampling: the language-detection logits after <|sot|>
 * (mlx_lightning.py:371-390 detect_language), and the parity tests' view of the decoder. */
int wx_decode_logits(wx_ctx* ctx, const void* enc_f16, int B, const int32_t* tokens, int n,
                     float* logits_out, void* stream);

/* replaces extract_words_with_dtw's numeric part (mlx_whisper_optimized_final.py:128-211)
 * and mlx_whisper.timing.dtw (:201): softmax / z-norm / median-7 / DTW on the scores
 * captured by the last wx_decode_greedy (n_sampled = its n_steps_out).  mode 0 = published find_alignment, 1 = in-repo
 * variant.  n_frames (device int32 [B], nullable) = encoder frames that carry audio per chunk
 * (the published algorithm crops the matrix to num_frames // 2).  Outputs (device int32): n_rows[B]; path_i/path_j [B][path_ld] stored
 * end->start; path_len[B].  matrix_out (nullable) f32 [B][sample_len+1][1500]. */
int wx_dtw_path(wx_ctx* ctx, const int32_t* tokens, const int32_t* n_frames, int B, int n_prompt, int n_sampled, int eot, int mode,
                float qk_scale, int32_t* n_rows, int32_t* path_i, int32_t* path_j, int path_ld,
                int32_t* path_len, float* matrix_out, void* stream);

/* replaces get_trellis + backtrack_beam (whisperx/alignment.py:268-269; :387-404, :500-579).
 * logp f32 [S][Tmax][V], T[S], tokens int32 [S][Nmax] (-1 = wildcard), N[S].
 * Outputs: path_tok int32 [S][Tmax], path_score f32 [S][Tmax], ok int32 [S];
 * trellis_out (nullable) f32 [S][Tmax][Nmax].  ctx may be any live context. */
int wx_ctc_align(wx_ctx* ctx, const float* logp, const int32_t* T, const int32_t* tokens,
                 const int32_t* N, int S, int Tmax, int Nmax, int V, int blank_id, int beam,
                 int32_t* path_tok, float* path_score, int32_t* ok, float* trellis_out, void* stream);

/* ---- wav2vec2 CTC forward (alignment.py:251-258) -------------------------------------- */
typedef struct wx_w2v wx_w2v;
/* HF Wav2Vec2Config fields of the align model loaded at alignment.py:97-106 */
typedef struct {
    int n_conv, conv_dim;
    int conv_kernel[8], conv_stride[8];
    int hidden, heads, layers, ffn, vocab, pos_kernel, pos_groups;
    int norm_mode;   /* feat_extract_norm: 0 = "group" (wav2vec2-base), 1 = "layer" (wav2vec2-large / XLSR)      */
    int stable_ln;   /* do_stable_layer_norm: 0 = post-LN encoder (base), 1 = pre-LN encoder (large / XLSR)         */
} wx_w2v_dims;
int wx_w2v_create(int device_id, const wx_w2v_dims* dims, wx_w2v** out);
void wx_w2v_destroy(wx_w2v* ctx);
const char* wx_w2v_last_error(wx_w2v* ctx);
int wx_w2v_bind_weight(wx_w2v* ctx, const char* name, const void* dptr, size_t nbytes);
int wx_w2v_finalize(wx_w2v* ctx);
/* frames the model emits for n_samples (segments < 400 samples are padded to 400) */
int wx_w2v_num_frames(const wx_w2v_dims* dims, long n_samples);
/* replaces `emissions = model(waveform_segment).logits; log_softmax` (alignment.py:251-258),
 * batched: pcm f32 [S][pcm_stride] zero padded (device), n_samples_host[S] (host).
 * logp_out f32 [S][Tmax_out][vocab] (device); T_out_host[S] (host) frames per segment. */
int wx_w2v_emissions(wx_w2v* ctx, const float* pcm, long pcm_stride, const int32_t* n_samples_host, int S,
                     float* logp_out, int Tmax_out, int32_t* T_out_host, void* stream);

/* wx_ctc_align on an alignment-model context (alignment.py:268-269) */
int wx_w2v_ctc_align(wx_w2v* ctx, const float* logp, const int32_t* T, const int32_t* tokens, const int32_t* N,
                     int S, int Tmax, int Nmax, int V, int blank_id, int beam, int32_t* path_tok,
                     float* path_score, int32_t* ok, float* trellis_out, void* stream);

/* ---- multi-GPU ------------------------------------------------------------ */
/* The ONE collective of the path: an all-gather over RCCL/xGMI of the fixed-width per-chunk result records
 * ({chunk_id, n_tokens, tokens[224], sum_logprob, no_speech, n_words, word_tok_end[224], word_start_ms[224],
 * word_end_ms[224]} int32, whisperx_mlx_amd/parallel.py) -- every <= 30 s chunk is independent end to end
 * (whisperx/asr.py:70-87), so nothing else crosses GPUs.  nccl_comm: the caller's ncclComm_t (RCCL); local: this
 * rank's records, bytes_per_rank bytes, padded to the largest share; all_out: world * bytes_per_rank bytes (both
 * device).  RCCL is taken from the calling process (not linked into libwxhip.so); returns -4 when the process has
 * none.  Python hosts issue the same collective through torch.distributed (backend "nccl"), parallel.gather_records. */
int wx_gather_results(void* nccl_comm, const void* local, size_t bytes_per_rank, void* all_out, void* stream);

/* Synchronises `stream` and returns non-zero (wx_last_error says why) if any kernel since wx_finalize raised
 * the context's device-side error flag (a bounded in-kernel wait that gave up).  The reference has no such
 * hook: its ops are synchronous mlx/torch calls (whisperx/backends/mlx_whisper.py:340-420). */
int wx_device_status(wx_ctx* ctx, void* stream);

/* Synchronises `stream`, then reads and clears the context's decode counters.  *selfq_out = cross-attention blocks of
 * the fused decode launch that computed their query themselves because the producing blocks had not delivered within
 * the poll window (csrc/declayer.hip: same bits either way, the launch never waits on another block for long).  A single
 * pass alone on the GPU counts 0; with three or four passes in flight a launch's producers can sit behind another stream's
 * blocks on their XCD while its consumers already run on another (DESIGN.md section 1 has the measured counts). */
int wx_decode_stats(wx_ctx* ctx, int* selfq_out, void* stream);

/* Captured decode steps (hipGraphs) are cached per launch signature and the cache is dropped wholesale when it is full
 * or when the alignment heads change; the number of times that has happened.  A host that enqueues the first pass of
 * every launch shape from one thread before its launcher threads start (captures must not race with other threads'
 * launches) forgets which shapes it has seen whenever this number moves.  No GPU work, no synchronisation. */
int wx_graph_generation(wx_ctx* ctx);

/* Do kernels launched on these `n` (<= 16) streams at the same time run at the same time?  The HIP runtime maps a
 * process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the variable is set before the GPU is first
 * touched); streams that share a queue run one after the other.  One block per stream spins for `usec` microseconds:
 * *factor = wall time of n spins launched together / wall time of one (~1: every stream has its own queue; ~2: two
 * share).  The host keeps one stream per pass in flight (whisperx/asr.py:80-87 hands over whole batches; how many are
 * in flight is this backend's business) and asks before it settles on four.  Synchronises the streams. */
int wx_streams_overlap(int device, void* const* streams, int n, int usec, float* factor);

#ifdef __cplusplus
}
#endif
#endif /* WXHIP_H */
