"""Oracle: Whisper encoder / decoder forward.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED at the reference boundary: the reference calls third-party
``mlx_whisper`` for this arithmetic (model.encoder at
/root/reference/mlx_whisper_batch_decoder.py:403; model.decoder(tokens, xa,
kv_cache) -> (logits, kv_cache, cross_qk) at :70-72,84-86 and
/root/reference/mlx_whisper_optimized_final.py:74,87-89).  mlx-whisper (branch
``whisperx-optimizations``, unpinned commit, on mlx>=0.26.0) is not vendored and
cannot be installed here, so this restates the published OpenAI Whisper
architecture it implements, in torch-CPU fp32, and is cross-checked against
``transformers.WhisperForConditionalGeneration`` with seeded random weights
(tests/test_oracle_models.py).

Weight names follow OpenAI/mlx-whisper checkpoints:
  encoder.conv1.{weight(d,n_mels,3),bias}  encoder.conv2.{weight(d,d,3),bias}
  encoder.positional_embedding(1500,d)
  encoder.blocks.N.{attn_ln,mlp_ln}.{weight,bias}
  encoder.blocks.N.attn.{query.{weight,bias},key.weight,value.{weight,bias},out.{weight,bias}}
  encoder.blocks.N.mlp.{0,2}.{weight,bias}   encoder.ln_post.{weight,bias}
  decoder.token_embedding.weight(vocab,d)    decoder.positional_embedding(448,d)
  decoder.blocks.N.{attn_ln,cross_attn_ln,mlp_ln}, .attn.*, .cross_attn.*, .mlp.*
  decoder.ln.{weight,bias}
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class Dims:
    n_mels: int
    n_audio_ctx: int
    n_audio_state: int
    n_audio_head: int
    n_audio_layer: int
    n_vocab: int
    n_text_ctx: int
    n_text_state: int
    n_text_head: int
    n_text_layer: int


DIMS = {
    "tiny": Dims(80, 1500, 384, 6, 4, 51865, 448, 384, 6, 4),
    "large-v3": Dims(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 32),
    "large-v3-turbo": Dims(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 4),
    "distil-large-v3": Dims(128, 1500, 1280, 20, 32, 51866, 448, 1280, 20, 2),
}


def sinusoids(length, channels, max_timescale=10000):
    """Fixed encoder positional embedding (published Whisper `sinusoids`)."""
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def random_weights(dims: Dims, seed=0, std=0.02, dtype=torch.float16, emb_std=None):
    """Seeded N(0, std^2) weights in the exact checkpoint shapes, rounded through
    `dtype` (the GPU path stores fp16) and returned as fp32 tensors."""
    g = torch.Generator().manual_seed(seed)
    w = {}

    def rnd(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype).float()

    def ln(prefix, d):
        w[prefix + ".weight"] = (1.0 + torch.randn(d, generator=g) * 0.1).to(dtype).float()
        w[prefix + ".bias"] = rnd(d, s=0.1)

    def attn(prefix, d):
        w[prefix + ".query.weight"] = rnd(d, d)
        w[prefix + ".query.bias"] = rnd(d)
        w[prefix + ".key.weight"] = rnd(d, d)
        w[prefix + ".value.weight"] = rnd(d, d)
        w[prefix + ".value.bias"] = rnd(d)
        w[prefix + ".out.weight"] = rnd(d, d)
        w[prefix + ".out.bias"] = rnd(d)

    def mlp(prefix, d):
        w[prefix + ".0.weight"] = rnd(4 * d, d)
        w[prefix + ".0.bias"] = rnd(4 * d)
        w[prefix + ".2.weight"] = rnd(d, 4 * d)
        w[prefix + ".2.bias"] = rnd(d)

    d = dims.n_audio_state
    w["encoder.conv1.weight"] = rnd(d, dims.n_mels, 3)
    w["encoder.conv1.bias"] = rnd(d)
    w["encoder.conv2.weight"] = rnd(d, d, 3)
    w["encoder.conv2.bias"] = rnd(d)
    w["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, d).to(dtype).float()
    for i in range(dims.n_audio_layer):
        p = f"encoder.blocks.{i}"
        ln(p + ".attn_ln", d)
        attn(p + ".attn", d)
        ln(p + ".mlp_ln", d)
        mlp(p + ".mlp", d)
    ln("encoder.ln_post", d)
    d = dims.n_text_state
    w["decoder.token_embedding.weight"] = rnd(dims.n_vocab, d, s=emb_std or std)
    w["decoder.positional_embedding"] = rnd(dims.n_text_ctx, d)
    for i in range(dims.n_text_layer):
        p = f"decoder.blocks.{i}"
        ln(p + ".attn_ln", d)
        attn(p + ".attn", d)
        ln(p + ".cross_attn_ln", d)
        attn(p + ".cross_attn", d)
        ln(p + ".mlp_ln", d)
        mlp(p + ".mlp", d)
    ln("decoder.ln", d)
    return w


def _ln(x, w, prefix):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + ".weight"], w[prefix + ".bias"], 1e-5)


def _linear(x, w, prefix, bias=True):
    return F.linear(x, w[prefix + ".weight"], w.get(prefix + ".bias") if bias else None)


def _mha(q, k, v, n_head, mask=None):
    """Published Whisper qkv_attention: q,k scaled by d_head^-0.25 each, softmax in
    fp32.  Returns (out (B,Tq,d), qk (B,H,Tq,Tk) pre-softmax fp32)."""
    B, Tq, d = q.shape
    dh = d // n_head
    scale = dh ** -0.25
    qh = q.view(B, Tq, n_head, dh).permute(0, 2, 1, 3) * scale
    kh = k.view(B, k.shape[1], n_head, dh).permute(0, 2, 3, 1) * scale
    vh = v.view(B, v.shape[1], n_head, dh).permute(0, 2, 1, 3)
    qk = (qh @ kh).float()
    if mask is not None:
        qk = qk + mask
    p = torch.softmax(qk, dim=-1)
    out = (p @ vh).permute(0, 2, 1, 3).reshape(B, Tq, d)
    return out, qk


def conv_stem(w, mel):
    """mel (B, T, n_mels) channels-last (mlx layout, trace_mlx_whisper.py:83-88)."""
    x = mel.permute(0, 2, 1)
    x = F.gelu(F.conv1d(x, w["encoder.conv1.weight"], w["encoder.conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, w["encoder.conv2.weight"], w["encoder.conv2.bias"], stride=2, padding=1))
    return x.permute(0, 2, 1)


def encoder_block(w, p, x, n_head):
    h = _ln(x, w, p + ".attn_ln")
    a, _ = _mha(_linear(h, w, p + ".attn.query"), _linear(h, w, p + ".attn.key", bias=False),
                _linear(h, w, p + ".attn.value"), n_head)
    x = x + _linear(a, w, p + ".attn.out")
    h = _ln(x, w, p + ".mlp_ln")
    x = x + _linear(F.gelu(_linear(h, w, p + ".mlp.0")), w, p + ".mlp.2")
    return x


@torch.no_grad()
def encoder_forward(w, dims: Dims, mel, n_layers=None, return_stem=False):
    """(B, 3000, n_mels) -> (B, 1500, d).  AudioEncoder."""
    x = conv_stem(w, mel.float())
    x = x + w["encoder.positional_embedding"][: x.shape[1]]
    if return_stem:
        return x
    L = dims.n_audio_layer if n_layers is None else n_layers
    for i in range(L):
        x = encoder_block(w, f"encoder.blocks.{i}", x, dims.n_audio_head)
    if n_layers is not None:
        return x
    return _ln(x, w, "encoder.ln_post")


@torch.no_grad()
def cross_kv(w, dims: Dims, enc):
    """Per decoder layer K = xa Wk^T, V = xa Wv^T + b."""
    out = []
    for i in range(dims.n_text_layer):
        p = f"decoder.blocks.{i}.cross_attn"
        out.append((_linear(enc, w, p + ".key", bias=False), _linear(enc, w, p + ".value")))
    return out


@torch.no_grad()
def decoder_forward(w, dims: Dims, tokens, xkv, self_cache=None, offset=0):
    """tokens (B, n) int64; xkv from cross_kv(); self_cache list[(K,V)] or None.
    Returns logits (B, n, vocab) fp32, new self_cache, cross_qk list[(B,H,n,1500)]."""
    B, n = tokens.shape
    x = w["decoder.token_embedding.weight"][tokens] + w["decoder.positional_embedding"][offset: offset + n]
    total = offset + n
    mask = torch.full((n, total), float("-inf")).triu_(offset + 1) if n > 1 else None
    new_cache, cross_qk = [], []
    for i in range(dims.n_text_layer):
        p = f"decoder.blocks.{i}"
        h = _ln(x, w, p + ".attn_ln")
        k = _linear(h, w, p + ".attn.key", bias=False)
        v = _linear(h, w, p + ".attn.value")
        if self_cache is not None:
            k = torch.cat([self_cache[i][0], k], dim=1)
            v = torch.cat([self_cache[i][1], v], dim=1)
        new_cache.append((k, v))
        a, _ = _mha(_linear(h, w, p + ".attn.query"), k, v, dims.n_text_head, mask)
        x = x + _linear(a, w, p + ".attn.out")
        h = _ln(x, w, p + ".cross_attn_ln")
        a, qk = _mha(_linear(h, w, p + ".cross_attn.query"), xkv[i][0], xkv[i][1], dims.n_text_head)
        cross_qk.append(qk)
        x = x + _linear(a, w, p + ".cross_attn.out")
        h = _ln(x, w, p + ".mlp_ln")
        x = x + _linear(F.gelu(_linear(h, w, p + ".mlp.0")), w, p + ".mlp.2")
    x = _ln(x, w, "decoder.ln")
    logits = (x @ w["decoder.token_embedding.weight"].T).float()
    return logits, new_cache, cross_qk


def from_hf_state_dict(sd, dims: Dims):
    """Map a transformers WhisperForConditionalGeneration state_dict onto the
    OpenAI names (used only to cross-check this oracle against HF)."""
    w = {}

    def cp(dst, src):
        w[dst] = sd[src].float().clone()

    def attn(dst, src):
        cp(dst + ".query.weight", src + ".q_proj.weight")
        cp(dst + ".query.bias", src + ".q_proj.bias")
        cp(dst + ".key.weight", src + ".k_proj.weight")
        cp(dst + ".value.weight", src + ".v_proj.weight")
        cp(dst + ".value.bias", src + ".v_proj.bias")
        cp(dst + ".out.weight", src + ".out_proj.weight")
        cp(dst + ".out.bias", src + ".out_proj.bias")

    def ln(dst, src):
        cp(dst + ".weight", src + ".weight")
        cp(dst + ".bias", src + ".bias")

    for c in ("conv1", "conv2"):
        cp(f"encoder.{c}.weight", f"model.encoder.{c}.weight")
        cp(f"encoder.{c}.bias", f"model.encoder.{c}.bias")
    cp("encoder.positional_embedding", "model.encoder.embed_positions.weight")
    for i in range(dims.n_audio_layer):
        s, d = f"model.encoder.layers.{i}", f"encoder.blocks.{i}"
        ln(d + ".attn_ln", s + ".self_attn_layer_norm")
        attn(d + ".attn", s + ".self_attn")
        ln(d + ".mlp_ln", s + ".final_layer_norm")
        for a, b in (("0", "fc1"), ("2", "fc2")):
            cp(f"{d}.mlp.{a}.weight", f"{s}.{b}.weight")
            cp(f"{d}.mlp.{a}.bias", f"{s}.{b}.bias")
    ln("encoder.ln_post", "model.encoder.layer_norm")
    cp("decoder.token_embedding.weight", "model.decoder.embed_tokens.weight")
    cp("decoder.positional_embedding", "model.decoder.embed_positions.weight")
    for i in range(dims.n_text_layer):
        s, d = f"model.decoder.layers.{i}", f"decoder.blocks.{i}"
        ln(d + ".attn_ln", s + ".self_attn_layer_norm")
        attn(d + ".attn", s + ".self_attn")
        ln(d + ".cross_attn_ln", s + ".encoder_attn_layer_norm")
        attn(d + ".cross_attn", s + ".encoder_attn")
        ln(d + ".mlp_ln", s + ".final_layer_norm")
        for a, b in (("0", "fc1"), ("2", "fc2")):
            cp(f"{d}.mlp.{a}.weight", f"{s}.{b}.weight")
            cp(f"{d}.mlp.{a}.bias", f"{s}.{b}.bias")
    ln("decoder.ln", "model.decoder.layer_norm")
    return w


def to_numpy(w):
    return {k: v.numpy() for k, v in w.items()}
