"""W2VHipModel: the wav2vec2 CTC align model on the GPU (opaque handle returned by
load_align_model, metadata["type"] == "hip").  Replaces `model(waveform).logits` +
log_softmax (/root/reference/whisperx/alignment.py:251-258) with a padded-batch forward in
HIP, and hosts the CTC trellis/backtrack kernels for align()."""
import ctypes as C
import json
import os
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

from . import _lib
from ._lib import W2vDims, lib, ptr


@dataclass
class W2VConfig:
    conv_dim: int = 512
    conv_kernel: List[int] = field(default_factory=lambda: [10, 3, 3, 3, 3, 2, 2])
    conv_stride: List[int] = field(default_factory=lambda: [5, 2, 2, 2, 2, 2, 2])
    hidden: int = 768
    heads: int = 12
    layers: int = 12
    ffn: int = 3072
    vocab: int = 32
    pos_kernel: int = 128
    pos_groups: int = 16
    norm_mode: int = 0      # 0 = feat_extract_norm "group"
    stable_ln: int = 0      # 0 = post-LN encoder

    def n_frames(self, n):
        n = max(int(n), 400)
        for k, s in zip(self.conv_kernel, self.conv_stride):
            n = (n - k) // s + 1
        return n


def pack_w2v(sd, cfg: W2VConfig, device):
    """HF Wav2Vec2ForCTC state_dict -> packed device tensors for libwxhip.so."""
    def h(x):
        return x.to(device=device, dtype=torch.float16).contiguous()

    pre = "wav2vec2."
    p = {}
    p["fe.conv0.w"] = sd[pre + "feature_extractor.conv_layers.0.conv.weight"].reshape(cfg.conv_dim, -1).to(
        device=device, dtype=torch.float32).contiguous()
    if cfg.norm_mode == 0:
        p["fe.gn.g"] = h(sd[pre + "feature_extractor.conv_layers.0.layer_norm.weight"])
        p["fe.gn.b"] = h(sd[pre + "feature_extractor.conv_layers.0.layer_norm.bias"])
    else:
        for i in range(len(cfg.conv_kernel)):
            p[f"fe.conv{i}.b"] = h(sd[pre + f"feature_extractor.conv_layers.{i}.conv.bias"])
            p[f"fe.ln{i}.g"] = h(sd[pre + f"feature_extractor.conv_layers.{i}.layer_norm.weight"])
            p[f"fe.ln{i}.b"] = h(sd[pre + f"feature_extractor.conv_layers.{i}.layer_norm.bias"])
    for i in range(1, len(cfg.conv_kernel)):
        w = sd[pre + f"feature_extractor.conv_layers.{i}.conv.weight"]          # (out, in, k)
        p[f"fe.conv{i}.w"] = h(w.permute(0, 2, 1).reshape(cfg.conv_dim, -1))   # [out][k][in]
    p["fp.ln.g"], p["fp.ln.b"] = h(sd[pre + "feature_projection.layer_norm.weight"]), h(sd[pre + "feature_projection.layer_norm.bias"])
    p["fp.w"], p["fp.b"] = h(sd[pre + "feature_projection.projection.weight"]), h(sd[pre + "feature_projection.projection.bias"])
    pc = pre + "encoder.pos_conv_embed.conv."
    if pc + "weight" in sd:
        pw = sd[pc + "weight"].float()
    else:   # weight-norm (dim=2): w = g * v / ||v||
        if pc + "parametrizations.weight.original0" in sd:
            g, v = sd[pc + "parametrizations.weight.original0"].float(), sd[pc + "parametrizations.weight.original1"].float()
        else:
            g, v = sd[pc + "weight_g"].float(), sd[pc + "weight_v"].float()
        pw = g * v / v.norm(dim=(0, 1), keepdim=True)
    # (out, in/groups, k) -> per group [out_local][k][in_local]
    p["pos.w"] = h(pw.permute(0, 2, 1).reshape(cfg.hidden, -1))
    p["pos.b"] = h(sd[pc + "bias"])
    p["enc.ln.g"], p["enc.ln.b"] = h(sd[pre + "encoder.layer_norm.weight"]), h(sd[pre + "encoder.layer_norm.bias"])
    for i in range(cfg.layers):
        s, q = f"{pre}encoder.layers.{i}.", f"l{i}."
        p[q + "qk.w"] = h(torch.cat([sd[s + "attention.q_proj.weight"], sd[s + "attention.k_proj.weight"]], 0))
        p[q + "qk.b"] = h(torch.cat([sd[s + "attention.q_proj.bias"], sd[s + "attention.k_proj.bias"]], 0))
        p[q + "v.w"], p[q + "v.b"] = h(sd[s + "attention.v_proj.weight"]), h(sd[s + "attention.v_proj.bias"])
        p[q + "o.w"], p[q + "o.b"] = h(sd[s + "attention.out_proj.weight"]), h(sd[s + "attention.out_proj.bias"])
        p[q + "ln1.g"], p[q + "ln1.b"] = h(sd[s + "layer_norm.weight"]), h(sd[s + "layer_norm.bias"])
        p[q + "fc1.w"], p[q + "fc1.b"] = h(sd[s + "feed_forward.intermediate_dense.weight"]), h(sd[s + "feed_forward.intermediate_dense.bias"])
        p[q + "fc2.w"], p[q + "fc2.b"] = h(sd[s + "feed_forward.output_dense.weight"]), h(sd[s + "feed_forward.output_dense.bias"])
        p[q + "ln2.g"], p[q + "ln2.b"] = h(sd[s + "final_layer_norm.weight"]), h(sd[s + "final_layer_norm.bias"])
    p["lm.w"], p["lm.b"] = h(sd["lm_head.weight"]), h(sd["lm_head.bias"])
    return p


def random_state_dict(cfg: W2VConfig, seed=0):
    """Seeded random fp16-rounded weights with the HF Wav2Vec2ForCTC names (throughput runs only:
    no align checkpoint ships with the reference)."""
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, s):
        return (torch.randn(*shape, generator=g) * s).half().float()

    w = {}
    c_in = 1
    for i, k in enumerate(cfg.conv_kernel):
        w[f"wav2vec2.feature_extractor.conv_layers.{i}.conv.weight"] = rnd(cfg.conv_dim, c_in, k, s=(2.0 / (c_in * k)) ** 0.5)
        c_in = cfg.conv_dim
    for p, d in (("wav2vec2.feature_extractor.conv_layers.0.layer_norm", cfg.conv_dim),
                 ("wav2vec2.feature_projection.layer_norm", cfg.conv_dim), ("wav2vec2.encoder.layer_norm", cfg.hidden)):
        w[p + ".weight"], w[p + ".bias"] = 1 + rnd(d, s=0.1), rnd(d, s=0.1)
    w["wav2vec2.feature_projection.projection.weight"] = rnd(cfg.hidden, cfg.conv_dim, s=0.05)
    w["wav2vec2.feature_projection.projection.bias"] = rnd(cfg.hidden, s=0.05)
    w["wav2vec2.encoder.pos_conv_embed.conv.weight"] = rnd(cfg.hidden, cfg.hidden // cfg.pos_groups, cfg.pos_kernel, s=0.02)
    w["wav2vec2.encoder.pos_conv_embed.conv.bias"] = rnd(cfg.hidden, s=0.05)
    for i in range(cfg.layers):
        p = f"wav2vec2.encoder.layers.{i}"
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[f"{p}.attention.{nm}.weight"], w[f"{p}.attention.{nm}.bias"] = rnd(cfg.hidden, cfg.hidden, s=0.04), rnd(cfg.hidden, s=0.05)
        for nm in ("layer_norm", "final_layer_norm"):
            w[f"{p}.{nm}.weight"], w[f"{p}.{nm}.bias"] = 1 + rnd(cfg.hidden, s=0.1), rnd(cfg.hidden, s=0.1)
        w[p + ".feed_forward.intermediate_dense.weight"], w[p + ".feed_forward.intermediate_dense.bias"] = rnd(cfg.ffn, cfg.hidden, s=0.04), rnd(cfg.ffn, s=0.05)
        w[p + ".feed_forward.output_dense.weight"], w[p + ".feed_forward.output_dense.bias"] = rnd(cfg.hidden, cfg.ffn, s=0.02), rnd(cfg.hidden, s=0.05)
    w["lm_head.weight"], w["lm_head.bias"] = rnd(cfg.vocab, cfg.hidden, s=0.1), rnd(cfg.vocab, s=0.1)
    return w


class W2VHipModel:
    def __init__(self, cfg: W2VConfig, packed, device_index=0):
        if not torch.cuda.is_available():
            raise _lib.WxError("no ROCm GPU visible: the HIP align model has no CPU fallback")
        self.cfg = cfg
        self.device = torch.device("cuda", device_index)
        self._L = lib()
        d = W2vDims()
        d.n_conv, d.conv_dim = len(cfg.conv_kernel), cfg.conv_dim
        for i, (k, s) in enumerate(zip(cfg.conv_kernel, cfg.conv_stride)):
            d.conv_kernel[i], d.conv_stride[i] = k, s
        d.hidden, d.heads, d.layers, d.ffn, d.vocab = cfg.hidden, cfg.heads, cfg.layers, cfg.ffn, cfg.vocab
        d.pos_kernel, d.pos_groups, d.norm_mode, d.stable_ln = cfg.pos_kernel, cfg.pos_groups, cfg.norm_mode, cfg.stable_ln
        self._dims = d
        h = C.c_void_p()
        rc = self._L.wx_w2v_create(device_index, C.byref(d), C.byref(h))
        self.ctx = h
        self._check(rc, "wx_w2v_create")
        self.packed = packed
        for name, t in packed.items():
            assert t.is_cuda and t.is_contiguous(), name
            self._check(self._L.wx_w2v_bind_weight(self.ctx, name.encode(), ptr(t), t.numel() * t.element_size()), "bind")
        self._check(self._L.wx_w2v_finalize(self.ctx), "wx_w2v_finalize")
        self.stream = torch.cuda.Stream(device=self.device)

    def _check(self, rc, what):
        if rc != 0:
            msg = self._L.wx_w2v_last_error(self.ctx).decode() if self.ctx else "no context"
            raise _lib.WxError(f"{what} failed (rc={rc}): {msg}")

    def close(self):
        if getattr(self, "ctx", None):
            torch.cuda.synchronize(self.device)
            self._L.wx_w2v_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def from_state_dict(cls, sd, cfg, device_index=0):
        return cls(cfg, pack_w2v(sd, cfg, torch.device("cuda", device_index)), device_index)

    @classmethod
    def from_hf_dir(cls, path, device_index=0):
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        cfg = W2VConfig(conv_dim=c["conv_dim"][0], conv_kernel=list(c["conv_kernel"]), conv_stride=list(c["conv_stride"]),
                        hidden=c["hidden_size"], heads=c["num_attention_heads"], layers=c["num_hidden_layers"],
                        ffn=c["intermediate_size"], vocab=c["vocab_size"], pos_kernel=c["num_conv_pos_embeddings"],
                        pos_groups=c["num_conv_pos_embedding_groups"],
                        norm_mode=0 if c.get("feat_extract_norm", "group") == "group" else 1,
                        stable_ln=int(bool(c.get("do_stable_layer_norm", False))))
        if os.path.exists(os.path.join(path, "model.safetensors")):
            from safetensors.torch import load_file
            sd = load_file(os.path.join(path, "model.safetensors"))
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        with open(os.path.join(path, "vocab.json")) as f:
            vocab = json.load(f)
        return cls.from_state_dict(sd, cfg, device_index), vocab

    def emissions(self, waveforms):
        """list of 1-D float32 arrays -> (log-probs f32 device tensor (S, Tmax, V), [T per segment])."""
        S = len(waveforms)
        n = [max(len(w), 400) for w in waveforms]
        n_max = max(n)
        # host arrays travel through ONE pinned staging buffer (grown on demand) and one asynchronous copy: a pageable
        # tensor built per call cost 30 ms for 16 x 30 s (zero fill + pageable H2D) against 14 ms of forward
        # (a FLAT buffer viewed at exactly (S, n_max): a copy out of a non-contiguous pinned view -- a corner of a larger 2-D
        # buffer -- is staged and synchronous in torch, round 5)
        flat = getattr(self, "_stage", None)
        if flat is None or flat.numel() < S * n_max:
            flat = torch.empty(max(S * n_max, flat.numel() if flat is not None else 0), dtype=torch.float32).pin_memory()
            self._stage = flat
        st = flat[: S * n_max].view(S, n_max)
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, "_stage_ev", None) is not None:
            self._stage_ev.synchronize()              # the previous call's copy out of the staging buffer has finished
        for i, w in enumerate(waveforms):
            a = np.asarray(w, dtype=np.float32).reshape(-1)
            row = st[i]
            row[: len(a)] = torch.from_numpy(a)
            row[len(a):] = 0.0
        pcm = torch.empty(S, n_max, dtype=torch.float32, device=self.device)
        pcm.copy_(st, non_blocking=True)
        self._stage_ev = torch.cuda.Event()
        self._stage_ev.record(cur)
        return self.emissions_device(pcm, n)

    def emissions_device(self, pcm, n):
        """pcm: f32 (S, n_max) device tensor, zero padded; n: samples per segment (>= 400 each)."""
        S, n_max = pcm.shape
        assert pcm.is_cuda and pcm.dtype == torch.float32 and pcm.is_contiguous() and n_max >= 400
        n = [max(int(v), 400) for v in n]
        Tmax = self.cfg.n_frames(max(n))
        logp = torch.zeros(S, Tmax, self.cfg.vocab, dtype=torch.float32, device=self.device)
        ns = (C.c_int * S)(*n)
        T = (C.c_int * S)()
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)
        self._check(self._L.wx_w2v_emissions(self.ctx, ptr(pcm), pcm.stride(0), ns, S, ptr(logp), Tmax, T,
                                             C.c_void_p(self.stream.cuda_stream)), "wx_w2v_emissions")
        if cur != self.stream:
            cur.wait_stream(self.stream)
        return logp, list(T)

    def ctc_align(self, logp, T, tokens, N, blank_id=0, beam=2, want_trellis=False):
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
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)
        self._check(self._L.wx_w2v_ctc_align(self.ctx, ptr(logp), ptr(T), ptr(tokens), ptr(N), S, Tmax, Nmax, V, blank_id,
                                             beam, ptr(path_tok), ptr(path_score), ptr(ok), ptr(trellis),
                                             C.c_void_p(self.stream.cuda_stream)), "wx_w2v_ctc_align")
        if cur != self.stream:
            cur.wait_stream(self.stream)
        return path_tok, path_score, ok, trellis
