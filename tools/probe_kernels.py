#!/usr/bin/env python3
"""Times the hot kernels in isolation on the GPU box (HIP events on the engine stream)."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperx_mlx_amd import weights
from whisperx_mlx_amd.engine import WhisperHipEngine

dims = weights.MODEL_DIMS["large-v3"]
ck = weights.random_checkpoint(dims, seed=0, device="cuda")
PB = int(os.environ.get("PROBE_B", "16"))
eng = WhisperHipEngine(dims, weights.pack(ck, dims, "cuda"), max_batch=PB,
                       alignment_heads=weights.default_alignment_heads("large-v3", dims))
d = dims.n_text_state
MB = {"v1 out-proj tn16": 2 * d * d / 1e6, "v1 out-proj tn8": 2 * d * d / 1e6, "v1 out-proj tn4": 2 * d * d / 1e6,
      "v1 LN+cq tn16": 2 * d * d / 1e6, "v1 LN+cq tn8": 2 * d * d / 1e6, "v1 LN+cq tn4": 2 * d * d / 1e6,
      "v1 LN+fc1": 8 * d * d / 1e6, "v1 fc2 tn16": 8 * d * d / 1e6, "v1 fc2 tn8": 8 * d * d / 1e6, "v1 fc2 tn4": 8 * d * d / 1e6,
      "v1 LN+qkv": 6 * d * d / 1e6,
      "v2 fc2 splitK": 8 * d * d / 1e6, "v2 qkv": 6 * d * d / 1e6, "v2 logits": 2 * dims.n_vocab * d / 1e6,
      "fused cq+xattn": 122.88 + 2 * d * d / 1e6, "cross-attn split4": 122.88, "cross-attn split2": 122.88, "cross-attn s4 t128": 122.88, "cross-attn s8 t128": 122.88,
      "cross-attn s8 t256": 122.88, "cross-attn s5 t256": 122.88, "cross-attn s2 t512": 122.88, "cross-attn s4 t512": 122.88,
      "cross-attn s10 t128": 122.88, "cross-attn s4 t256": 122.88, "cross-attn s3 t256": 122.88, "cross-attn s6 t256": 122.88, "cross-attn s2 t256": 122.88, "cross-attn s3 t512": 122.88, "cross-attn s1 t512": 122.88, "cross-attn s6 t128": 122.88, "v1 fc2 tn8 w16": 8 * d * d / 1e6, "v1 fc2 tn16 w16": 8 * d * d / 1e6}
only = [a for a in sys.argv[1:] if not a.startswith('-')]
out = {}
for name, kind, arg in (("v1 out-proj tn16", 7, 16), ("v1 out-proj tn8", 7, 8), ("v1 out-proj tn4", 7, 4),
                        ("v1 LN+cq tn16", 12, 16), ("v1 LN+cq tn8", 12, 8), ("v1 LN+cq tn4", 12, 4),
                        ("v1 LN+fc1", 8, 0), ("bal o tn5", 7, 5), ("bal cq tn5", 12, 5), ("bal fc1 tn10", 8, 10), ("bal fc2 tn5 w16", 9, 5 + 32), ("bal fc2 tn5 w8", 9, 5),
                        ("bal qkv tn15", 10, 15), ("bal fc2 tn10 w16", 9, 10 + 32), ("v1 LN+fc1 tn8", 8, 8), ("v1 LN+qkv tn8", 10, 8), ("v1 fc2 tn16", 9, 16), ("v1 fc2 tn8", 9, 8), ("v1 fc2 tn4", 9, 4), ("v1 fc2 tn8 w16", 9, 8 + 32), ("v1 fc2 tn16 w16", 9, 16 + 32), ("v1 LN+qkv", 10, 0),
                        ("v2 fc2 splitK", 4, 0), ("v2 qkv", 3, 0), ("v2 logits", 5, 0), ("fused cq+xattn", 13, 0), ("cross-attn split4", 0, 4), ("cross-attn split2", 0, 2),
                        ("cross-attn s4 t128", 0, 4 + 16 * 2), ("cross-attn s8 t128", 0, 8 + 16 * 2), ("cross-attn s8 t256", 0, 8 + 16 * 4),
                        ("cross-attn s5 t256", 0, 5 + 16 * 4), ("cross-attn s2 t512", 0, 2 + 16 * 8), ("cross-attn s4 t512", 0, 4 + 16 * 8),
                        ("cross-attn s10 t128", 0, 10 + 16 * 2), ("cross-attn s4 t256", 0, 4 + 16 * 4), ("cross-attn s3 t256", 0, 3 + 16 * 4),
                        ("cross-attn s6 t256", 0, 6 + 16 * 4), ("cross-attn s2 t256", 0, 2 + 16 * 4), ("cross-attn s3 t512", 0, 3 + 16 * 8),
                        ("cross-attn s1 t512", 0, 1 + 16 * 8), ("cross-attn s6 t128", 0, 6 + 16 * 2),
                        ("v3 out-proj", 7, 1000), ("v3 LN+cq", 12, 1000), ("v3 LN+fc1", 8, 1000), ("v3 fc2", 9, 1000), ("v3 LN+qkv", 10, 1000), ("self-attn pos=75", 11, 75), ("self-attn pos=147", 11, 147),
                        ("enc fc1 gemm", 1, 0), ("enc fc2 gemm", 6, 0), ("enc attention", 2, 0)):
    if only and not any(o in name for o in only):
        continue
    ms = eng.probe(kind, PB, 64 if kind not in (1, 2, 6) else 8, arg)
    r = {"us": round(ms * 1e3, 2)}
    if name in MB:
        r["GB/s"] = round(MB[name] / ms, 1)
    out[name] = r
    print(f"{name:24s} {r}", flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "probe.json"), "w"), indent=1)
