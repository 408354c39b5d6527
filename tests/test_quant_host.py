"""int8 weight quantiser, host side: the product's torch quantiser against the oracle's numpy
restatement of whisperx/backends/mlx_quantization.py:86-91,143-150 -- integers and scales must be identical."""
import numpy as np
import pytest
import torch

from oracle import quant as OQ
from whisperx_mlx_amd import weights as WT


@pytest.mark.parametrize("gran", ["row", "tensor"])
def test_quantiser_equals_oracle(gran):
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(96, 160, generator=g) * 0.05).half()
    w[7] = 0
    w[9, 3] = 0.5                     # an outlier row
    qb, sc = WT.quantize_rows_int8(w, gran)
    q, s = OQ.quantize(w.float().numpy(), gran)
    assert qb.dtype == torch.uint8 and sc.dtype == torch.float32
    assert np.array_equal(qb.numpy().astype(np.int16) - 128, q.astype(np.int16))
    assert np.array_equal(sc.numpy(), s)
    assert q.min() >= -127 and q.max() <= 127 and s[7] == (1.0 if gran == "row" else s[0])
    # round trip error is at most half a step
    err = np.abs(OQ.dequantize(q, s) - w.float().numpy())
    assert (err <= s[:, None] * 0.5 + 1e-9).all()
    if gran == "row":
        assert np.abs(q).max(axis=1)[np.arange(96) != 7].min() == 127      # every non-zero row uses the full range


def test_packed_decoder_policy_and_oracle_checkpoint():
    dims = WT.MODEL_DIMS["tiny"]
    ck = WT.random_checkpoint(dims, seed=2)
    p = WT.quantize_packed_decoder(WT.pack(ck, dims, "cpu"), dims)
    last = dims.n_text_layer - 1
    for i in range(dims.n_text_layer):
        for nm in WT.INT8_DECODE_WEIGHTS:
            base = f"dec.{i}.{nm}"
            if i == last:
                assert base + ".w" in p and base + ".wq" not in p          # policy: last decoder layer stays fp16
            else:
                assert base + ".w" not in p and p[base + ".wq"].dtype == torch.uint8 and p[base + ".ws"].shape == (p[base + ".wq"].shape[0],)
    assert all(k.endswith((".w", ".b", ".g", ".pos", ".emb")) or k.endswith((".wq", ".ws")) for k in p)
    # the oracle's dequantised checkpoint = dequantised packed bytes (fused qkv rows = query | key | value rows)
    ckq = OQ.dequantized_checkpoint({k: v.float() for k, v in ck.items()}, dims.n_text_layer)
    d = dims.n_text_state
    deq = (p["dec.0.qkv.wq"].float() - 128) * p["dec.0.qkv.ws"][:, None]
    assert torch.equal(deq[:d], ckq["decoder.blocks.0.attn.query.weight"]) and torch.equal(deq[2 * d:], ckq["decoder.blocks.0.attn.value.weight"])
    assert torch.equal(ckq[f"decoder.blocks.{last}.mlp.0.weight"], ck[f"decoder.blocks.{last}.mlp.0.weight"].float())
    assert torch.equal(ckq["encoder.blocks.0.mlp.0.weight"], ck["encoder.blocks.0.mlp.0.weight"].float())
