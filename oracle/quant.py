"""Oracle: int8 weight quantisation of the decoder linears.  TEST INFRASTRUCTURE ONLY.

The reference's only statement of "int8 weights" (BASELINE config 5) is the unwired sketch
/root/reference/whisperx/backends/mlx_quantization.py:
  :86-91    symmetric: scale = abs_max / 127, zero_point = 0
  :143-146  q = clip(round(w / scale + zero_point), -128, 127)
  :148-150  w' = (q - zero_point) * scale          (float32)
  :161-162  y = x @ w'.T  (dequantise, then float matmul), + bias
  :321-328  Whisper policy: skip the conv stem, keep the last decoder layer in fp16
The sketch derives abs_max from activation statistics of a calibration run (:46-57), which cannot be what a
WEIGHT scale means; restated here with abs_max of the weight itself, per output row or per
tensor (the sketch's granularity).  PARITY UNPINNED at the reference boundary (the reference holds no vector
for it and the module is never called); pinned only against this restatement.
"""
import numpy as np
import torch

DECODER_LINEARS = ("attn.query", "attn.key", "attn.value", "attn.out", "cross_attn.query", "cross_attn.out", "mlp.0", "mlp.2")


def quantize(w: np.ndarray, granularity: str = "row"):
    """-> (q int8 [N][K], scale f32 [N])"""
    w = np.asarray(w, dtype=np.float32)
    amax = np.abs(w).max(axis=1) if granularity == "row" else np.full(w.shape[0], np.abs(w).max(), np.float32)
    scale = np.where(amax > 0, amax / np.float32(127.0), np.float32(1.0)).astype(np.float32)
    q = np.clip(np.rint(w / scale[:, None]), -127, 127).astype(np.int8)     # np.rint = round half to even, as mx.round / torch.round
    return q, scale


def dequantize(q: np.ndarray, scale: np.ndarray) -> np.ndarray:
    return q.astype(np.float32) * scale[:, None].astype(np.float32)


def dequantized_checkpoint(ck: dict, n_text_layer: int, granularity: str = "row", keep_last_fp16: bool = True) -> dict:
    """checkpoint (OpenAI names, fp32 tensors) whose decoder linear weights went through quantize -> dequantize:
    what the oracle's fp32 decoder runs on to mirror the HIP int8 path.  q/k/v are quantised as ONE [3d][d] matrix
    row by row (row scales make that identical to quantising them separately)."""
    out = dict(ck)
    for i in range(n_text_layer):
        if keep_last_fp16 and i == n_text_layer - 1 and n_text_layer > 1:
            continue
        for nm in DECODER_LINEARS:
            k = f"decoder.blocks.{i}.{nm}.weight"
            w = ck[k].float().numpy()
            if granularity == "tensor" and nm in ("attn.query", "attn.key", "attn.value"):
                # one scale for the fused QKV matrix, as the packed layout quantises it
                amax = max(np.abs(ck[f"decoder.blocks.{i}.attn.{n}.weight"].float().numpy()).max() for n in ("query", "key", "value"))
                scale = np.full(w.shape[0], amax / 127.0 if amax > 0 else 1.0, np.float32)
                q = np.clip(np.rint(w / scale[:, None]), -127, 127).astype(np.int8)
            else:
                q, scale = quantize(w, granularity)
            out[k] = torch.from_numpy(dequantize(q, scale))
    return out
