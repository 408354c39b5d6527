"""WhisperHipEngine: thin Python owner of one libwxhip.so context.  PyTorch-ROCm is
used only for device buffers and the stream; every computation is a HIP kernel behind
the C ABI (include/wxhip.h)."""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import DecodeOpts, ModelDims, check, lib, ptr
from .audio import N_FRAMES, N_SAMPLES, mel_filters

RULE_SUPPRESS_BLANK, RULE_SUPPRESS_TOKENS, RULE_TS_NOTIMESTAMPS = 1, 2, 4
RULE_TS_PAIRS, RULE_TS_MONOTONE, RULE_TS_INITIAL, RULE_TS_PROB = 8, 16, 32, 64
RULES_LIGHTNING = 127
RULES_OPTIMIZED_FINAL = RULE_SUPPRESS_TOKENS | RULE_TS_PROB


@dataclass
class DecodeOutput:
    tokens: torch.Tensor          # int32 (B, n_text_ctx) on device: prompt + sampled, EOT filled
    sum_logprob: torch.Tensor     # f32 (B,)
    no_speech_prob: torch.Tensor  # f32 (B,)
    n_sampled: int
    n_prompt: int


class WhisperHipEngine:
    def __init__(self, dims, packed, max_batch=16, device_index=0, alignment_heads=None):
        if not torch.cuda.is_available():
            raise _lib.WxError("no ROCm GPU visible: the HIP backend has no CPU fallback")
        self.dims = dims
        self.device = torch.device("cuda", device_index)
        self.max_batch = max_batch
        self._L = lib()
        d = ModelDims(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head, dims.n_audio_layer,
                      dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head, dims.n_text_layer)
        h = C.c_void_p()
        rc = self._L.wx_create(device_index, C.byref(d), max_batch, C.byref(h))
        self.ctx = h
        check(self.ctx if h else None, rc, "wx_create")
        self.packed = packed            # keeps the tensors alive
        for name, t in packed.items():
            want = torch.uint8 if name.endswith(".wq") else torch.float32 if name.endswith(".ws") else torch.float16
            assert t.is_cuda and t.dtype == want and t.is_contiguous(), name
            check(self.ctx, self._L.wx_bind_weight(self.ctx, name.encode(), ptr(t), t.numel() * t.element_size()), "wx_bind_weight")
        try:
            check(self.ctx, self._L.wx_finalize(self.ctx), "wx_finalize")
        except Exception:
            self.close()            # a workspace that did not fit: give back what was allocated before the error is passed on
            raise
        filt = np.ascontiguousarray(mel_filters(dims.n_mels))
        check(self.ctx, self._L.wx_set_mel_filters(self.ctx, filt.ctypes.data_as(C.POINTER(C.c_float)), dims.n_mels),
              "wx_set_mel_filters")
        self.alignment_heads = None
        if alignment_heads:
            self.set_alignment_heads(alignment_heads)
        self.stream = torch.cuda.Stream(device=self.device)   # hipGraph capture needs a non-default stream
        B = max_batch
        self._tokens = torch.zeros(B, dims.n_text_ctx, dtype=torch.int32, device=self.device)
        self._sum_lp = torch.zeros(B, dtype=torch.float32, device=self.device)
        self._nsp = torch.zeros(B, dtype=torch.float32, device=self.device)
        self._masks = {}
        self._warm = set()     # launch shapes whose hipGraphs this context has captured (backend scheduler)
        self._warm_gen = 0
        self.pass_slots = None

    @property
    def warm(self):
        """launch shapes whose hipGraphs this context holds; emptied when the library has dropped its graph cache"""
        gen = self._L.wx_graph_generation(self.ctx)
        if gen != self._warm_gen:
            self._warm.clear()
            self._warm_gen = gen
        return self._warm

    def close(self):
        if getattr(self, "ctx", None):
            torch.cuda.synchronize(self.device)
            self._L.wx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _enter(self):
        """the engine's kernels run on its own stream (hipGraph capture needs a non-default one);
        order it after the caller's stream unless the caller already works on it"""
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)

    def _exit(self):
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            cur.wait_stream(self.stream)

    @property
    def _s(self):
        return C.c_void_p(self.stream.cuda_stream)

    def set_alignment_heads(self, heads):
        flat = (C.c_int * (2 * len(heads)))(*[v for lh in heads for v in lh])
        check(self.ctx, self._L.wx_set_alignment_heads(self.ctx, flat, len(heads)), "wx_set_alignment_heads")
        self.alignment_heads = list(heads)

    # ------------------------------------------------------------------ stages
    def logmel(self, pcm, n_valid=None, want_f32=False):
        """pcm: f32 (B, n<=480000) device tensor (rows zero-padded by the kernel's bounds).
        Returns fp16 (B, 3000, n_mels) [and f32 copy]."""
        assert pcm.is_cuda and pcm.dtype == torch.float32 and pcm.dim() == 2 and pcm.is_contiguous()
        B = pcm.shape[0]
        if n_valid is None:
            n_valid = torch.full((B,), min(pcm.shape[1], N_SAMPLES), dtype=torch.int32, device=self.device)
        n_valid = torch.clamp(n_valid.to(device=self.device, dtype=torch.int32), max=pcm.shape[1]).contiguous()
        mel = torch.empty(B, N_FRAMES, self.dims.n_mels, dtype=torch.float16, device=self.device)
        mel32 = torch.empty(B, N_FRAMES, self.dims.n_mels, dtype=torch.float32, device=self.device) if want_f32 else None
        self._enter()
        check(self.ctx, self._L.wx_logmel(self.ctx, ptr(pcm), pcm.stride(0), ptr(n_valid), B, ptr(mel), ptr(mel32), self._s),
              "wx_logmel")
        self._exit()
        return (mel, mel32) if want_f32 else mel

    def encode(self, mel):
        assert mel.is_cuda and mel.dtype == torch.float16 and mel.is_contiguous()
        B = mel.shape[0]
        enc = torch.empty(B, self.dims.n_audio_ctx, self.dims.n_audio_state, dtype=torch.float16, device=self.device)
        self._enter()
        check(self.ctx, self._L.wx_encode(self.ctx, ptr(mel), B, ptr(enc), self._s), "wx_encode")
        self._exit()
        return enc

    def set_encoder_cap(self, max_blocks):
        """lab: confine encode()'s GEMM / attention launches to at most `max_blocks` CUs (multiple of 8; 0 = no cap)"""
        check(self.ctx, self._L.wx_set_encoder_cap(self.ctx, int(max_blocks)), "wx_set_encoder_cap")

    def suppress_mask(self, ids, n_vocab=None):
        key = tuple(ids)
        if key not in self._masks:
            m = torch.zeros(n_vocab or self.dims.n_vocab, dtype=torch.uint8)
            if len(ids):
                m[torch.tensor(list(ids), dtype=torch.long)] = 1
            self._masks[key] = m.to(self.device)
        return self._masks[key]

    def decode(self, enc, tokenizer, prompt, rules=RULES_LIGHTNING, suppress_ids=(), sample_len=None,
               max_initial_ts=50, forced_len=0, capture_qk=False, use_graph=True, check_every=8, cross_split=2,
               step_variant=0, fc2_tile_n=0, forced_lens=None, rows=None, profile_launches=False, max_steps_ahead=0):
        """`rows` > enc.shape[0]: launch that many rows, the ones beyond the encoder output being padding (they count as
        finished at once and cost nothing in the attention kernels).  hipGraphs are captured per row count, so a
        scheduler that always launches its batch size never captures a second set (wx_decode_opts.n_active)."""
        n = enc.shape[0]
        B = max(n, min(int(rows or n), self.max_batch))
        o = DecodeOpts()
        o.n_active = n if B > n else 0
        for i, t in enumerate(prompt):
            o.prompt[i] = int(t)
        o.n_prompt = len(prompt)
        o.sample_len = sample_len or self.dims.n_text_ctx // 2
        o.rules = rules
        o.max_initial_ts = -1 if max_initial_ts is None else int(max_initial_ts)
        o.forced_len = int(forced_len)
        o.eot, o.no_speech, o.timestamp_begin = tokenizer.eot, tokenizer.no_speech, tokenizer.timestamp_begin
        bl = list(tokenizer.blank_tokens) + [-1, -1]
        o.blank0, o.blank1 = bl[0], bl[1]
        ids = set()
        if rules & RULE_SUPPRESS_TOKENS:
            ids.update(suppress_ids)
        if rules & RULE_TS_NOTIMESTAMPS:
            ids.add(tokenizer.no_timestamps)
        mask = self.suppress_mask(sorted(ids))
        o.suppress_mask = mask.data_ptr()
        o.capture_qk = int(bool(capture_qk))
        o.set_tuning(use_graph=bool(use_graph), check_every=check_every, cross_split=cross_split, step_variant=step_variant,
                     fc2_tile_n=fc2_tile_n, profile_launches=bool(profile_launches), max_steps_ahead=max_steps_ahead or 0)
        if forced_lens is not None:          # bench workload: per-row lengths (device int32 [B]), the caller keeps the tensor alive
            assert forced_len > 0 and forced_lens.is_cuda and forced_lens.dtype == torch.int32 and forced_lens.numel() >= B
            o.forced_lens = forced_lens.data_ptr()
        n_steps = C.c_int(0)
        self._enter()
        check(self.ctx, self._L.wx_decode_greedy(self.ctx, ptr(enc), B, C.byref(o), ptr(self._tokens), ptr(self._sum_lp),
                                                 ptr(self._nsp), C.byref(n_steps), self._s), "wx_decode_greedy")
        self._exit()
        return DecodeOutput(self._tokens[:n], self._sum_lp[:n], self._nsp[:n], n_steps.value, len(prompt))

    def decode_logits(self, enc, tokens):
        """teacher-forced last-position logits (test hook)."""
        B, n = tokens.shape
        tokens = tokens.to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty(B, self.dims.n_vocab, dtype=torch.float32, device=self.device)
        self._enter()
        check(self.ctx, self._L.wx_decode_logits(self.ctx, ptr(enc), B, ptr(tokens), n, ptr(out), self._s), "wx_decode_logits")
        self._exit()
        return out

    def align_qk(self, B):
        rows = self.dims.n_text_ctx // 2
        out = torch.empty(B, len(self.alignment_heads), rows, self.dims.n_audio_ctx, dtype=torch.float32, device=self.device)
        self._enter()
        check(self.ctx, self._L.wx_get_align_qk(self.ctx, B, ptr(out), self._s), "wx_get_align_qk")
        self._exit()
        return out

    def dtw_launch(self, dec: DecodeOutput, eot, mode=0, qk_scale=1.0, n_frames=None):
        """Launches the alignment-matrix + DTW kernels and returns the device result tensors
        (n_rows, path_i, path_j, path_len) without a host sync."""
        B = dec.tokens.shape[0]
        rows, T = self.dims.n_text_ctx // 2, self.dims.n_audio_ctx
        ld = T + rows + 4
        if getattr(self, "_dtw_out", None) is None or self._dtw_out[0].shape[0] < B:
            self._dtw_out = (torch.zeros(B, dtype=torch.int32, device=self.device),
                             torch.zeros(B, ld, dtype=torch.int32, device=self.device),
                             torch.zeros(B, ld, dtype=torch.int32, device=self.device),
                             torch.zeros(B, dtype=torch.int32, device=self.device))
        n_rows, pi, pj, plen = self._dtw_out
        if n_frames is not None:
            n_frames = n_frames.to(device=self.device, dtype=torch.int32).contiguous()
        self._enter()
        check(self.ctx, self._L.wx_dtw_path(self.ctx, ptr(dec.tokens), ptr(n_frames), B, dec.n_prompt, dec.n_sampled, eot, mode, C.c_float(qk_scale),
                                            ptr(n_rows), ptr(pi), ptr(pj), ld, ptr(plen), None, self._s), "wx_dtw_path")
        self._exit()
        return n_rows, pi, pj, plen

    def dtw_path(self, dec: DecodeOutput, eot, mode=0, qk_scale=1.0, want_matrix=False, n_frames=None):
        """Runs the alignment-matrix + DTW kernels on the scores captured by the last
        decode().  Returns per sequence (n_rows, path (2, L) int32 numpy in start->end order)."""
        B = dec.tokens.shape[0]
        rows, T = self.dims.n_text_ctx // 2, self.dims.n_audio_ctx
        ld = T + rows + 4
        n_rows = torch.zeros(B, dtype=torch.int32, device=self.device)
        pi = torch.zeros(B, ld, dtype=torch.int32, device=self.device)
        pj = torch.zeros(B, ld, dtype=torch.int32, device=self.device)
        plen = torch.zeros(B, dtype=torch.int32, device=self.device)
        mat = torch.zeros(B, rows + 1, T, dtype=torch.float32, device=self.device) if want_matrix else None
        if n_frames is not None:
            n_frames = n_frames.to(device=self.device, dtype=torch.int32).contiguous()
        self._enter()
        check(self.ctx, self._L.wx_dtw_path(self.ctx, ptr(dec.tokens), ptr(n_frames), B, dec.n_prompt, dec.n_sampled, eot, mode, C.c_float(qk_scale),
                                            ptr(n_rows), ptr(pi), ptr(pj), ld, ptr(plen), ptr(mat), self._s), "wx_dtw_path")
        self._exit()
        n_rows_h, pi_h, pj_h, plen_h = n_rows.cpu().numpy(), pi.cpu().numpy(), pj.cpu().numpy(), plen.cpu().numpy()
        out = []
        for b in range(B):
            L = int(plen_h[b])
            path = np.stack([pi_h[b, :L][::-1], pj_h[b, :L][::-1]]).astype(np.int32)
            out.append((int(n_rows_h[b]), path))
        return (out, mat) if want_matrix else out

    def check_status(self):
        """synchronises the engine's stream and raises WxError if a kernel raised the context's
        device-side error flag (a bounded in-kernel wait that gave up)"""
        check(self.ctx, self._L.wx_device_status(self.ctx, self._s), "wx_device_status")

    def launch_profile(self):
        """synchronises the engine's stream; (average us of a fused decode launch since the last call, launches timed) --
        decode(..., profile_launches=True) makes the launches time themselves on the device"""
        avg, n = C.c_double(0.0), C.c_longlong(0)
        check(self.ctx, self._L.wx_launch_profile(self.ctx, C.byref(avg), C.byref(n), self._s), "wx_launch_profile")
        return float(avg.value), int(n.value)

    def decode_stats(self):
        """synchronises the engine's stream; reads and clears the decode counters: {"selfq": attention blocks of the fused
        decode launch that computed their query themselves (declayer.hip)}"""
        n = C.c_int(0)
        check(self.ctx, self._L.wx_decode_stats(self.ctx, C.byref(n), self._s), "wx_decode_stats")
        return {"selfq": int(n.value)}

    def probe(self, kind, B, iters, arg=0):
        """bench hook: average duration (ms) of one hot kernel launched `iters` times back to
        back on the engine's stream, measured with HIP events on that stream."""
        check(self.ctx, self._L.wx_probe(self.ctx, kind, B, 2, arg, self._s), "wx_probe")   # warm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.stream):
            e0.record(self.stream)
            check(self.ctx, self._L.wx_probe(self.ctx, kind, B, iters, arg, self._s), "wx_probe")
            e1.record(self.stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / iters

    def ctc_align(self, logp, T, tokens, N, blank_id=0, beam=2, want_trellis=False):
        """logp f32 (S, Tmax, V); T int32 (S,); tokens int32 (S, Nmax) (-1 wildcard); N int32 (S,)."""
        S, Tmax, V = logp.shape
        Nmax = tokens.shape[1]
        logp = logp.to(self.device, torch.float32).contiguous()
        T = T.to(self.device, torch.int32).contiguous()
        tokens = tokens.to(self.device, torch.int32).contiguous()
        N = N.to(self.device, torch.int32).contiguous()
        path_tok = torch.full((S, Tmax), -1, dtype=torch.int32, device=self.device)
        path_score = torch.zeros(S, Tmax, dtype=torch.float32, device=self.device)
        ok = torch.zeros(S, dtype=torch.int32, device=self.device)
        trellis = torch.zeros(S, Tmax, Nmax, dtype=torch.float32, device=self.device) if want_trellis else None
        self._enter()
        check(self.ctx, self._L.wx_ctc_align(self.ctx, ptr(logp), ptr(T), ptr(tokens), ptr(N), S, Tmax, Nmax, V, blank_id,
                                             beam, ptr(path_tok), ptr(path_score), ptr(ok), ptr(trellis), self._s),
              "wx_ctc_align")
        self._exit()
        return path_tok, path_score, ok, trellis
