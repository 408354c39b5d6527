"""Checkpoint directories: the product's loader / name mapping (whisperx_mlx_amd.weights) on the two
on-disk formats a user has -- transformers (config.json + model.safetensors + generation_config.json) and mlx / OpenAI
(config.json with n_mels... + weights.safetensors, conv weights (out, k, in) as mlx stores them).  The mapped weights
must reproduce the transformers model's own logits through the oracle's forward."""
import json
import os
import warnings

import pytest
import torch

from oracle import whisper_ref as OW
from whisperx_mlx_amd import weights as WT

warnings.filterwarnings("ignore")


def _hf_model():
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    torch.manual_seed(1)
    cfg = WhisperConfig(vocab_size=600, num_mel_bins=16, encoder_layers=2, decoder_layers=3, encoder_attention_heads=2,
                        decoder_attention_heads=2, d_model=128, encoder_ffn_dim=512, decoder_ffn_dim=512,
                        max_source_positions=1500, max_target_positions=448, activation_function="gelu",
                        pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=3)
    return WhisperForConditionalGeneration(cfg).eval()


def test_transformers_directory(tmp_path):
    m = _hf_model()
    m.generation_config.alignment_heads = [[1, 0], [2, 1]]
    m.generation_config.suppress_tokens = [5, 7, 9]
    m.save_pretrained(tmp_path, safe_serialization=True)
    dims, sd, extra = WT.load_checkpoint_dir(str(tmp_path))
    assert (dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_layer, dims.n_vocab, dims.n_text_ctx,
            dims.n_text_state, dims.n_text_head, dims.n_text_layer) == (16, 1500, 128, 2, 600, 448, 128, 2, 3)
    assert extra["alignment_heads"] == [(1, 0), (2, 1)] and extra["suppress_tokens"] == [5, 7, 9]
    # same mapping as the oracle's own (independently written) one
    odims = OW.Dims(16, 1500, 128, 2, 2, 600, 448, 128, 2, 3)
    ref = OW.from_hf_state_dict(m.state_dict(), odims)
    for k, v in ref.items():
        assert k in sd, k
        assert torch.equal(sd[k].float(), v.float()), k
    # and the mapped weights give the transformers logits
    w = {k: v.float() for k, v in sd.items()}
    mel = torch.randn(2, 3000, 16)
    toks = torch.randint(0, 600, (2, 5))
    with torch.no_grad():
        out = m(input_features=mel.permute(0, 2, 1), decoder_input_ids=toks)
    enc = OW.encoder_forward(w, odims, mel)
    lg, _, _ = OW.decoder_forward(w, odims, toks, OW.cross_kv(w, odims, enc))
    assert (lg - out.logits).abs().max() < 1e-5
    # packing to the kernel layout keeps every tensor the C ABI asks for
    p = WT.pack(sd, dims, "cpu")
    assert p["dec.2.qkv.w"].shape == (384, 128) and p["enc.conv1.w"].shape == (128, 48) and p["dec.emb"].shape == (600, 128)


def test_mlx_style_directory(tmp_path):
    from safetensors.torch import save_file
    dims = WT.ModelDimensions(16, 1500, 128, 2, 2, 600, 448, 128, 2, 2)
    ck = WT.random_checkpoint(dims, seed=3)
    mlx = {k: v.clone() for k, v in ck.items()}
    for c in ("encoder.conv1.weight", "encoder.conv2.weight"):
        mlx[c] = mlx[c].permute(0, 2, 1).contiguous()          # mlx layout (out, k, in)
    save_file(mlx, str(tmp_path / "weights.safetensors"))
    from dataclasses import asdict
    json.dump(asdict(dims), open(tmp_path / "config.json", "w"))
    d2, sd, extra = WT.load_checkpoint_dir(str(tmp_path))
    assert d2 == dims and extra == {}
    for k, v in ck.items():
        assert torch.equal(sd[k], v), k
    with pytest.raises(FileNotFoundError):
        os.remove(tmp_path / "weights.safetensors")
        WT.load_checkpoint_dir(str(tmp_path))


def test_real_mlx_key_names_npz_and_alignment_heads(tmp_path):
    """the format the reference's backends download (mlx-community/whisper-*-mlx): MLP linears named mlp1 / mlp2, no
    encoder.positional_embedding (the model regenerates the sinusoids), conv weights (out, k, in), older repos as
    weights.npz with an `alignment_heads` array next to the weights"""
    import numpy as np
    from dataclasses import asdict
    dims = WT.ModelDimensions(16, 1500, 128, 2, 2, 600, 448, 128, 2, 2)
    ck = WT.random_checkpoint(dims, seed=5)
    mlx = {}
    for k, v in ck.items():
        if k == "encoder.positional_embedding":
            continue
        k2 = k.replace(".mlp.0.", ".mlp1.").replace(".mlp.2.", ".mlp2.")
        mlx[k2] = (v.permute(0, 2, 1).contiguous() if k.endswith(("conv1.weight", "conv2.weight")) else v).numpy()
    assert "encoder.blocks.0.mlp1.weight" in mlx and not any(".mlp.0." in k for k in mlx)
    mlx["alignment_heads"] = np.array([[1, 0], [1, 1]], dtype=np.int32)
    np.savez(tmp_path / "weights.npz", **mlx)
    json.dump(asdict(dims), open(tmp_path / "config.json", "w"))
    d2, sd, extra = WT.load_checkpoint_dir(str(tmp_path))
    assert d2 == dims and extra["alignment_heads"] == [(1, 0), (1, 1)] and "alignment_heads" not in sd
    for k, v in ck.items():
        if k == "encoder.positional_embedding":
            assert torch.allclose(sd[k].float(), WT.sinusoids(1500, 128), atol=0)      # regenerated, fp32
        else:
            assert torch.equal(sd[k], v), k
    p = WT.pack(sd, dims, "cpu")                # every tensor the C ABI asks for is there
    assert p["enc.pos"].shape == (1500, 128) and p["dec.1.fc1.w"].shape == (512, 128)
