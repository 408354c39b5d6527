"""Oracle: wav2vec2 CTC forward (emissions).  TEST INFRASTRUCTURE ONLY.

What /root/reference/whisperx/alignment.py:251-258 runs per segment:
``emissions = model(waveform_segment).logits`` (HF Wav2Vec2ForCTC, loaded at :97-106;
``en`` uses torchaudio WAV2VEC2_ASR_BASE_960H, the same architecture, :32,89-94) then
``torch.log_softmax(emissions, dim=-1)`` (:258).  The raw waveform is NOT normalised.
This is a plain torch-CPU fp32 restatement of the wav2vec2-base graph
(feat_extract_norm="group", post-LayerNorm encoder), cross-checked against
``transformers.Wav2Vec2ForCTC`` with seeded random weights in tests/test_oracle_models.py.
Weight names are the HF state_dict names (weight-norm of the positional conv already
folded: ``wav2vec2.encoder.pos_conv_embed.conv.weight``).
"""
from dataclasses import dataclass, field
from typing import List

import torch
import torch.nn.functional as F


@dataclass
class W2VDims:
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
    norm_mode: int = 0     # 0 = feat_extract_norm "group" (base), 1 = "layer" (large / XLSR, conv bias)
    stable_ln: int = 0     # 1 = do_stable_layer_norm (pre-LN encoder + final LN)


def n_frames(n_samples, dims: W2VDims):
    n = n_samples
    for k, s in zip(dims.conv_kernel, dims.conv_stride):
        n = (n - k) // s + 1
    return n


def fold_weight_norm(sd):
    """HF stores the positional conv with weight-norm (dim=2): w = g * v / ||v||_(0,1)."""
    p = "wav2vec2.encoder.pos_conv_embed.conv."
    if p + "weight" in sd:
        return sd
    sd = dict(sd)
    if p + "parametrizations.weight.original0" in sd:
        g, v = sd.pop(p + "parametrizations.weight.original0"), sd.pop(p + "parametrizations.weight.original1")
    else:
        g, v = sd.pop(p + "weight_g"), sd.pop(p + "weight_v")
    sd[p + "weight"] = g * v / v.norm(dim=(0, 1), keepdim=True)
    return sd


def random_weights(dims: W2VDims, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    w = {}

    def rnd(*shape, s):
        return (torch.randn(*shape, generator=g) * s).to(dtype).float()

    def ln(p, d):
        w[p + ".weight"] = (1 + torch.randn(d, generator=g) * 0.1).to(dtype).float()
        w[p + ".bias"] = rnd(d, s=0.1)

    c_in = 1
    for i, k in enumerate(dims.conv_kernel):
        w[f"wav2vec2.feature_extractor.conv_layers.{i}.conv.weight"] = rnd(dims.conv_dim, c_in, k, s=(2.0 / (c_in * k)) ** 0.5)
        if dims.norm_mode == 1:
            w[f"wav2vec2.feature_extractor.conv_layers.{i}.conv.bias"] = rnd(dims.conv_dim, s=0.05)
            ln(f"wav2vec2.feature_extractor.conv_layers.{i}.layer_norm", dims.conv_dim)
        c_in = dims.conv_dim
    if dims.norm_mode == 0:
        ln("wav2vec2.feature_extractor.conv_layers.0.layer_norm", dims.conv_dim)
    ln("wav2vec2.feature_projection.layer_norm", dims.conv_dim)
    w["wav2vec2.feature_projection.projection.weight"] = rnd(dims.hidden, dims.conv_dim, s=0.05)
    w["wav2vec2.feature_projection.projection.bias"] = rnd(dims.hidden, s=0.05)
    cg = dims.hidden // dims.pos_groups
    w["wav2vec2.encoder.pos_conv_embed.conv.weight"] = rnd(dims.hidden, cg, dims.pos_kernel, s=0.02)
    w["wav2vec2.encoder.pos_conv_embed.conv.bias"] = rnd(dims.hidden, s=0.05)
    ln("wav2vec2.encoder.layer_norm", dims.hidden)
    for i in range(dims.layers):
        p = f"wav2vec2.encoder.layers.{i}"
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[f"{p}.attention.{nm}.weight"] = rnd(dims.hidden, dims.hidden, s=0.04)
            w[f"{p}.attention.{nm}.bias"] = rnd(dims.hidden, s=0.05)
        ln(p + ".layer_norm", dims.hidden)
        w[p + ".feed_forward.intermediate_dense.weight"] = rnd(dims.ffn, dims.hidden, s=0.04)
        w[p + ".feed_forward.intermediate_dense.bias"] = rnd(dims.ffn, s=0.05)
        w[p + ".feed_forward.output_dense.weight"] = rnd(dims.hidden, dims.ffn, s=0.02)
        w[p + ".feed_forward.output_dense.bias"] = rnd(dims.hidden, s=0.05)
        ln(p + ".final_layer_norm", dims.hidden)
    w["lm_head.weight"] = rnd(dims.vocab, dims.hidden, s=0.1)
    w["lm_head.bias"] = rnd(dims.vocab, s=0.1)
    return w


def _ln(x, w, p):
    return F.layer_norm(x, (x.shape[-1],), w[p + ".weight"], w[p + ".bias"], 1e-5)


@torch.no_grad()
def feature_encoder(w, dims: W2VDims, wave):
    """wave (1, n) -> (1, T, 512)"""
    x = wave[:, None, :]
    for i, (k, s) in enumerate(zip(dims.conv_kernel, dims.conv_stride)):
        pre = f"wav2vec2.feature_extractor.conv_layers.{i}"
        x = F.conv1d(x, w[pre + ".conv.weight"], w.get(pre + ".conv.bias"), stride=s)
        if dims.norm_mode == 1:
            x = F.layer_norm(x.transpose(1, 2), (dims.conv_dim,), w[pre + ".layer_norm.weight"], w[pre + ".layer_norm.bias"],
                             1e-5).transpose(1, 2)
        elif i == 0:
            x = F.group_norm(x, dims.conv_dim, w["wav2vec2.feature_extractor.conv_layers.0.layer_norm.weight"],
                             w["wav2vec2.feature_extractor.conv_layers.0.layer_norm.bias"], 1e-5)
        x = F.gelu(x)
    return x.transpose(1, 2)


@torch.no_grad()
def forward_logits(w, dims: W2VDims, wave, upto=None):
    """wave (1, n) f32 -> logits (1, T, vocab).  `upto` returns an intermediate for tests."""
    x = feature_encoder(w, dims, wave.float())
    if upto == "features":
        return x
    x = _ln(x, w, "wav2vec2.feature_projection.layer_norm")
    x = F.linear(x, w["wav2vec2.feature_projection.projection.weight"], w["wav2vec2.feature_projection.projection.bias"])
    if upto == "projection":
        return x
    pc = F.conv1d(x.transpose(1, 2), w["wav2vec2.encoder.pos_conv_embed.conv.weight"],
                  w["wav2vec2.encoder.pos_conv_embed.conv.bias"], padding=dims.pos_kernel // 2, groups=dims.pos_groups)
    if dims.pos_kernel % 2 == 0:
        pc = pc[:, :, :-1]
    x = x + F.gelu(pc).transpose(1, 2)
    if not dims.stable_ln:
        x = _ln(x, w, "wav2vec2.encoder.layer_norm")
    if upto == "posconv":
        return x
    B, T, d = x.shape
    H, dh = dims.heads, d // dims.heads
    for i in range(dims.layers):
        p = f"wav2vec2.encoder.layers.{i}"
        hin = _ln(x, w, p + ".layer_norm") if dims.stable_ln else x
        q = F.linear(hin, w[p + ".attention.q_proj.weight"], w[p + ".attention.q_proj.bias"]) * dh ** -0.5
        k = F.linear(hin, w[p + ".attention.k_proj.weight"], w[p + ".attention.k_proj.bias"])
        v = F.linear(hin, w[p + ".attention.v_proj.weight"], w[p + ".attention.v_proj.bias"])
        qh = q.view(B, T, H, dh).transpose(1, 2)
        kh = k.view(B, T, H, dh).transpose(1, 2)
        vh = v.view(B, T, H, dh).transpose(1, 2)
        a = torch.softmax(qh @ kh.transpose(2, 3), dim=-1) @ vh
        a = a.transpose(1, 2).reshape(B, T, d)
        x = x + F.linear(a, w[p + ".attention.out_proj.weight"], w[p + ".attention.out_proj.bias"])
        if dims.stable_ln:
            fin = _ln(x, w, p + ".final_layer_norm")
        else:
            x = _ln(x, w, p + ".layer_norm")
            fin = x
        f = F.gelu(F.linear(fin, w[p + ".feed_forward.intermediate_dense.weight"], w[p + ".feed_forward.intermediate_dense.bias"]))
        x = x + F.linear(f, w[p + ".feed_forward.output_dense.weight"], w[p + ".feed_forward.output_dense.bias"])
        if not dims.stable_ln:
            x = _ln(x, w, p + ".final_layer_norm")
        if upto == f"layer{i}":
            return x
    if dims.stable_ln:
        x = _ln(x, w, "wav2vec2.encoder.layer_norm")
    return F.linear(x, w["lm_head.weight"], w["lm_head.bias"])


@torch.no_grad()
def emissions(w, dims: W2VDims, wave):
    """alignment.py:251-258: log_softmax of the CTC logits, (T, vocab) f32."""
    if wave.dim() == 1:
        wave = wave[None]
    if wave.shape[-1] < 400:      # alignment.py:243-249
        wave = F.pad(wave, (0, 400 - wave.shape[-1]))
    return torch.log_softmax(forward_logits(w, dims, wave), dim=-1)[0]
