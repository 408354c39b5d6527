"""Whisper weights: dims table, checkpoint-name -> packed-layout conversion, seeded
random weights for throughput runs (no checkpoints ship with the reference:
/root/reference/.gitignore:174-184), and loaders for local checkpoints.

Packed layout handed to libwxhip.so (all fp16, K-contiguous [out][in] like
nn.Linear, which is the MFMA operand layout):
  enc.conv1.w [d][3*n_mels]  (k-major: [out][k][in], the (out,k,in) layout mlx uses,
                              TRUE_BATCH_IMPLEMENTATION.md:138)      enc.conv1.b [d]
  enc.conv2.w [d][3*d]       enc.conv2.b   enc.pos [1500][d]   enc.lnpost.{g,b}
  enc.N.ln1.{g,b} enc.N.qk.w [2d][d] (query;key) enc.N.qk.b [2d] (query bias; 0)
  enc.N.v.{w,b} enc.N.o.{w,b} enc.N.ln2.{g,b} enc.N.fc1.{w,b} enc.N.fc2.{w,b}
  dec.emb [vocab][d] dec.pos [448][d] dec.ln.{g,b}
  dec.N.ln1 dec.N.qkv.w [3d][d] (q;k;v) dec.N.qkv.b (q bias;0;v bias) dec.N.o
  dec.N.ln2 dec.N.cq dec.N.ckv.w [2d][d] (cross key;value) dec.N.ckv.b (0;v bias)
  dec.N.co dec.N.ln3 dec.N.fc1 dec.N.fc2
"""
import json
import math
import os
from dataclasses import dataclass, asdict

import torch


@dataclass
class ModelDimensions:
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


def _dims(n_mels, d, heads, layers, vocab, dec_layers=None):
    return ModelDimensions(n_mels, 1500, d, heads, layers, vocab, 448, d, heads,
                           layers if dec_layers is None else dec_layers)


# name mapping follows whisperx/backends/mlx_lightning.py:46-73
MODEL_DIMS = {
    "tiny": _dims(80, 384, 6, 4, 51865), "tiny.en": _dims(80, 384, 6, 4, 51864),
    "base": _dims(80, 512, 8, 6, 51865), "base.en": _dims(80, 512, 8, 6, 51864),
    "small": _dims(80, 768, 12, 12, 51865), "small.en": _dims(80, 768, 12, 12, 51864),
    "medium": _dims(80, 1024, 16, 24, 51865), "medium.en": _dims(80, 1024, 16, 24, 51864),
    "large": _dims(80, 1280, 20, 32, 51865), "large-v1": _dims(80, 1280, 20, 32, 51865),
    "large-v2": _dims(80, 1280, 20, 32, 51865), "large-v3": _dims(128, 1280, 20, 32, 51866),
    "large-v3-turbo": _dims(128, 1280, 20, 32, 51866, 4), "turbo": _dims(128, 1280, 20, 32, 51866, 4),
    "distil-large-v3": _dims(128, 1280, 20, 32, 51866, 2), "distil-whisper-large-v3": _dims(128, 1280, 20, 32, 51866, 2),
}

# (layer, head) cross-attention heads correlated with word timing (openai-whisper
# _ALIGNMENT_HEADS / HF generation_config.alignment_heads).  The reference reads them as
# model.alignment_heads (mlx_whisper_optimized_final.py:146).  Listed from upstream
# knowledge; a checkpoint's own generation_config.json overrides them when present.
ALIGNMENT_HEADS = {
    "tiny": [(2, 2), (3, 0), (3, 2), (3, 3), (3, 4), (3, 5)],
    "tiny.en": [(1, 0), (2, 0), (2, 5), (3, 0), (3, 1), (3, 2), (3, 3), (3, 4)],
    "base": [(3, 1), (4, 2), (4, 3), (4, 7), (5, 1), (5, 2), (5, 4), (5, 6)],
    "base.en": [(3, 3), (4, 7), (5, 1), (5, 5), (5, 7)],
    "small": [(5, 3), (5, 9), (8, 0), (8, 4), (8, 7), (8, 8), (9, 0), (9, 7), (9, 9), (10, 5)],
    "medium": [(13, 15), (15, 4), (15, 15), (16, 1), (20, 0), (23, 4)],
    "large-v2": [(10, 12), (13, 17), (16, 11), (16, 12), (16, 13), (17, 15), (17, 16), (18, 4), (18, 11),
                 (18, 19), (19, 11), (21, 2), (21, 3), (22, 3), (22, 9), (22, 12), (23, 5), (23, 7), (23, 13),
                 (25, 5), (26, 1), (26, 12), (27, 15)],
    "large-v3": [(7, 0), (10, 17), (12, 18), (13, 12), (16, 1), (17, 14), (19, 11), (21, 4), (24, 1), (25, 6)],
    "large-v3-turbo": [(2, 4), (2, 11), (3, 3), (3, 6), (3, 11), (3, 14)],
    "turbo": [(2, 4), (2, 11), (3, 3), (3, 6), (3, 11), (3, 14)],
}


def default_alignment_heads(name, dims):
    """Heads for `name`; unknown models fall back to all heads of the last half of the
    decoder (what openai-whisper does before a mask is set)."""
    if name in ALIGNMENT_HEADS:
        return list(ALIGNMENT_HEADS[name])
    return [(l, h) for l in range(dims.n_text_layer // 2, dims.n_text_layer) for h in range(dims.n_text_head)]


def resolve_model_name(name):
    """Strips the hub prefixes the reference adds (mlx_lightning.py:49-69)."""
    n = name
    for pre in ("mlx-community/", "openai/"):
        if n.startswith(pre):
            n = n[len(pre):]
    if n.startswith("whisper-"):
        n = n[len("whisper-"):]
    if n.endswith("-mlx"):
        n = n[:-4]
    if n.startswith("distil-whisper-"):
        n = "distil-" + n[len("distil-whisper-"):]
    return n


def sinusoids(length, channels, max_timescale=10000):
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def random_checkpoint(dims, seed=0, std=0.02, device="cpu", emb_std=None):
    """Seeded N(0, std^2) fp16 weights with the exact checkpoint names and shapes
    (the throughput workload when no real checkpoint is on the box)."""
    g = torch.Generator(device=device).manual_seed(seed)
    w = {}

    def rnd(*shape, s=std):
        return (torch.randn(*shape, generator=g, device=device) * s).to(torch.float16)

    def ln(p, d):
        w[p + ".weight"] = (1.0 + torch.randn(d, generator=g, device=device) * 0.1).to(torch.float16)
        w[p + ".bias"] = rnd(d, s=0.1)

    def attn(p, d):
        for nm, bias in (("query", True), ("key", False), ("value", True), ("out", True)):
            w[f"{p}.{nm}.weight"] = rnd(d, d)
            if bias:
                w[f"{p}.{nm}.bias"] = rnd(d)

    def mlp(p, d):
        w[p + ".0.weight"] = rnd(4 * d, d)
        w[p + ".0.bias"] = rnd(4 * d)
        w[p + ".2.weight"] = rnd(d, 4 * d)
        w[p + ".2.bias"] = rnd(d)

    d = dims.n_audio_state
    w["encoder.conv1.weight"] = rnd(d, dims.n_mels, 3)
    w["encoder.conv1.bias"] = rnd(d)
    w["encoder.conv2.weight"] = rnd(d, d, 3)
    w["encoder.conv2.bias"] = rnd(d)
    w["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, d).to(device=device, dtype=torch.float16)
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


INT8_DECODE_WEIGHTS = ("qkv", "o", "cq", "co", "fc1", "fc2")


def quantize_rows_int8(w, granularity="row"):
    """Symmetric int8 weight quantisation, zero point 0 (the only in-tree spec of the reference's "int8 weights":
    whisperx/backends/mlx_quantization.py:86-91 scale = abs_max / 127, :143-146 q = clip(round(w / scale)), :148-150
    w' = q * scale, :161-162 dequantise then float matmul).  `granularity` "row": one scale per output row (SURVEY 8
    f4), "tensor": the reference's single scale.  Returns (bytes q + 128 as uint8 [N][K], scales fp32 [N])."""
    dev = w.device
    w = w.float().cpu()       # on the host: IEEE division, so the integers do not depend on the device's fp32 divide
    amax = w.abs().amax(dim=1) if granularity == "row" else w.abs().max().expand(w.shape[0])
    scale = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax)).to(torch.float32)
    q = torch.clamp(torch.round(w / scale[:, None]), -127, 127)
    return (q + 128).to(torch.uint8).contiguous().to(dev), scale.contiguous().to(dev)


def quantize_packed_decoder(p, dims, granularity="row", keep_last_fp16=True):
    """int8 copies of the decode GEMV weights of a packed checkpoint (in place: "<base>.w" is replaced by
    "<base>.wq" + "<base>.ws").  The reference's Whisper policy keeps the last decoder layer in fp16 and leaves the
    conv stem alone (mlx_quantization.py:321-328); the encoder (MFMA-bound) and the cross K/V projection (run once
    per batch) stay fp16 here as well."""
    last = dims.n_text_layer - 1
    for i in range(dims.n_text_layer):
        if keep_last_fp16 and i == last and dims.n_text_layer > 1:
            continue
        for nm in INT8_DECODE_WEIGHTS:
            base = f"dec.{i}.{nm}"
            q, sc = quantize_rows_int8(p.pop(base + ".w"), granularity)
            p[base + ".wq"], p[base + ".ws"] = q, sc
    return p


def pack(w, dims, device):
    """checkpoint names (OpenAI / mlx-whisper) -> packed fp16 device tensors."""
    def t(x):
        return x.to(device=device, dtype=torch.float16).contiguous()

    def z(n):
        return torch.zeros(n, dtype=torch.float16, device=device)

    p = {}
    p["enc.conv1.w"] = t(w["encoder.conv1.weight"].permute(0, 2, 1).reshape(dims.n_audio_state, -1))
    p["enc.conv1.b"] = t(w["encoder.conv1.bias"])
    p["enc.conv2.w"] = t(w["encoder.conv2.weight"].permute(0, 2, 1).reshape(dims.n_audio_state, -1))
    p["enc.conv2.b"] = t(w["encoder.conv2.bias"])
    p["enc.pos"] = t(w["encoder.positional_embedding"][: dims.n_audio_ctx])
    p["enc.lnpost.g"] = t(w["encoder.ln_post.weight"])
    p["enc.lnpost.b"] = t(w["encoder.ln_post.bias"])
    d = dims.n_audio_state
    for i in range(dims.n_audio_layer):
        s, q = f"encoder.blocks.{i}", f"enc.{i}"
        p[q + ".ln1.g"], p[q + ".ln1.b"] = t(w[s + ".attn_ln.weight"]), t(w[s + ".attn_ln.bias"])
        p[q + ".qk.w"] = t(torch.cat([w[s + ".attn.query.weight"], w[s + ".attn.key.weight"]], 0))
        p[q + ".qk.b"] = torch.cat([t(w[s + ".attn.query.bias"]), z(d)])
        p[q + ".v.w"], p[q + ".v.b"] = t(w[s + ".attn.value.weight"]), t(w[s + ".attn.value.bias"])
        p[q + ".o.w"], p[q + ".o.b"] = t(w[s + ".attn.out.weight"]), t(w[s + ".attn.out.bias"])
        p[q + ".ln2.g"], p[q + ".ln2.b"] = t(w[s + ".mlp_ln.weight"]), t(w[s + ".mlp_ln.bias"])
        p[q + ".fc1.w"], p[q + ".fc1.b"] = t(w[s + ".mlp.0.weight"]), t(w[s + ".mlp.0.bias"])
        p[q + ".fc2.w"], p[q + ".fc2.b"] = t(w[s + ".mlp.2.weight"]), t(w[s + ".mlp.2.bias"])
    p["dec.emb"] = t(w["decoder.token_embedding.weight"])
    p["dec.pos"] = t(w["decoder.positional_embedding"])
    p["dec.ln.g"], p["dec.ln.b"] = t(w["decoder.ln.weight"]), t(w["decoder.ln.bias"])
    d = dims.n_text_state
    for i in range(dims.n_text_layer):
        s, q = f"decoder.blocks.{i}", f"dec.{i}"
        p[q + ".ln1.g"], p[q + ".ln1.b"] = t(w[s + ".attn_ln.weight"]), t(w[s + ".attn_ln.bias"])
        p[q + ".qkv.w"] = t(torch.cat([w[s + ".attn.query.weight"], w[s + ".attn.key.weight"],
                                       w[s + ".attn.value.weight"]], 0))
        p[q + ".qkv.b"] = torch.cat([t(w[s + ".attn.query.bias"]), z(d), t(w[s + ".attn.value.bias"])])
        p[q + ".o.w"], p[q + ".o.b"] = t(w[s + ".attn.out.weight"]), t(w[s + ".attn.out.bias"])
        p[q + ".ln2.g"], p[q + ".ln2.b"] = t(w[s + ".cross_attn_ln.weight"]), t(w[s + ".cross_attn_ln.bias"])
        p[q + ".cq.w"], p[q + ".cq.b"] = t(w[s + ".cross_attn.query.weight"]), t(w[s + ".cross_attn.query.bias"])
        p[q + ".ckv.w"] = t(torch.cat([w[s + ".cross_attn.key.weight"], w[s + ".cross_attn.value.weight"]], 0))
        p[q + ".ckv.b"] = torch.cat([z(d), t(w[s + ".cross_attn.value.bias"])])
        p[q + ".co.w"], p[q + ".co.b"] = t(w[s + ".cross_attn.out.weight"]), t(w[s + ".cross_attn.out.bias"])
        p[q + ".ln3.g"], p[q + ".ln3.b"] = t(w[s + ".mlp_ln.weight"]), t(w[s + ".mlp_ln.bias"])
        p[q + ".fc1.w"], p[q + ".fc1.b"] = t(w[s + ".mlp.0.weight"]), t(w[s + ".mlp.0.bias"])
        p[q + ".fc2.w"], p[q + ".fc2.b"] = t(w[s + ".mlp.2.weight"]), t(w[s + ".mlp.2.bias"])
    return p


_HF_ATTN = {"q_proj": "query", "k_proj": "key", "v_proj": "value", "out_proj": "out"}


def hf_to_openai_names(sd):
    """transformers WhisperForConditionalGeneration state_dict -> OpenAI names."""
    out = {}
    for k, v in sd.items():
        k = k.replace("model.", "", 1) if k.startswith("model.") else k
        if k.startswith("proj_out"):
            continue
        k = k.replace("embed_tokens", "token_embedding").replace("embed_positions.weight", "positional_embedding")
        k = k.replace("layers.", "blocks.")
        k = k.replace("self_attn_layer_norm", "attn_ln").replace("encoder_attn_layer_norm", "cross_attn_ln")
        k = k.replace("final_layer_norm", "mlp_ln").replace("self_attn.", "attn.").replace("encoder_attn.", "cross_attn.")
        k = k.replace("fc1", "mlp.0").replace("fc2", "mlp.2")
        for a, b in _HF_ATTN.items():
            k = k.replace("." + a + ".", "." + b + ".")
        if k == "encoder.layer_norm.weight" or k == "encoder.layer_norm.bias":
            k = k.replace("layer_norm", "ln_post")
        if k == "decoder.layer_norm.weight" or k == "decoder.layer_norm.bias":
            k = k.replace("layer_norm", "ln")
        out[k] = v
    return out


def _load_weight_file(path):
    """(tensor dict, file name) of the first weight file a checkpoint directory holds: safetensors (mlx-community and
    transformers repos) or the weights.npz older mlx-community repos ship"""
    for f in ("weights.safetensors", "model.safetensors"):
        p = os.path.join(path, f)
        if os.path.exists(p):
            from safetensors.torch import load_file
            return load_file(p), f
    p = os.path.join(path, "weights.npz")
    if os.path.exists(p):
        import numpy as np
        with np.load(p) as z:
            return {k: torch.from_numpy(np.ascontiguousarray(z[k])) for k in z.files}, "weights.npz"
    raise FileNotFoundError(f"no weights.safetensors / model.safetensors / weights.npz under {path}")


def mlx_to_openai_names(sd, dims):
    """mlx-whisper checkpoint -> OpenAI names and torch layouts.  The mlx converter (the format the reference's backends
    download, whisperx/backends/mlx_lightning.py:46-74) renames the MLP linears `mlp.0` / `mlp.2` to `mlp1` / `mlp2`,
    stores conv weights as (out, k, in), drops `encoder.positional_embedding` (the model regenerates the sinusoids) and
    may carry an `alignment_heads` array next to the weights."""
    out, extra = {}, {}
    for k, v in sd.items():
        if k == "alignment_heads":
            extra["alignment_heads"] = [tuple(int(x) for x in row) for row in v.tolist()]
            continue
        k = k.replace(".mlp1.", ".mlp.0.").replace(".mlp2.", ".mlp.2.")
        out[k] = v
    for c in ("encoder.conv1.weight", "encoder.conv2.weight"):
        w = out[c]
        n_in = dims.n_mels if c.endswith("conv1.weight") else dims.n_audio_state
        if w.shape[1] == 3 and w.shape[2] == n_in:          # (out, k, in) -> torch (out, in, k)
            out[c] = w.permute(0, 2, 1).contiguous()
    if "encoder.positional_embedding" not in out:
        out["encoder.positional_embedding"] = sinusoids(dims.n_audio_ctx, dims.n_audio_state)
    return out, extra


def load_checkpoint_dir(path):
    """Loads a local checkpoint directory: either an mlx / OpenAI style one (config.json with n_mels... +
    weights.safetensors / model.safetensors / weights.npz) or a transformers one (config.json with d_model... +
    model.safetensors).  Returns (dims, weights dict with OpenAI names, extra dict)."""
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    sd, _fname = _load_weight_file(path)
    extra = {}
    if "n_mels" in cfg:
        dims = ModelDimensions(**{k: cfg[k] for k in asdict(MODEL_DIMS["tiny"])})
        sd, extra = mlx_to_openai_names(sd, dims)
    else:
        dims = ModelDimensions(cfg["num_mel_bins"], cfg["max_source_positions"], cfg["d_model"],
                               cfg["encoder_attention_heads"], cfg["encoder_layers"], cfg["vocab_size"],
                               cfg["max_target_positions"], cfg["d_model"], cfg["decoder_attention_heads"],
                               cfg["decoder_layers"])
        sd = hf_to_openai_names(sd)
    gc = os.path.join(path, "generation_config.json")
    if os.path.exists(gc):
        with open(gc) as f:
            g = json.load(f)
        if g.get("alignment_heads"):
            extra["alignment_heads"] = [tuple(x) for x in g["alignment_heads"]]
        if g.get("suppress_tokens"):
            extra["suppress_tokens"] = list(g["suppress_tokens"])
    return dims, sd, extra
