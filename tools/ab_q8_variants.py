"""int8 decoder weights: the log-probability of sampled token 8 under the decode step variants (fused launch, one kernel per
stage, M-tiled GEMVs; graph and eager; 1 / 2 / 4 key splits) on a 1-layer large-v3-width model.  Round 3: the fused launch's
int8 instance differed from the others in rows 0, 1, 4 at this step (and nowhere else in 12 steps); int8 layers now take
the unfused pair (api.hip decode_step_v1).   python tools/ab_q8_variants.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from whisperx_mlx_amd import _lib, weights, engine as E
from whisperx_mlx_amd.engine import WhisperHipEngine
from whisperx_mlx_amd.tokenizer import get_tokenizer
wide = weights.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 1)
ckw = weights.random_checkpoint(wide, seed=5, std=0.03, emb_std=0.03)
packed = weights.quantize_packed_decoder(weights.pack(ckw, wide, "cuda"), wide)
eng = WhisperHipEngine(wide, packed, max_batch=16)
tok = get_tokenizer(wide.n_vocab)
B = 16
enc = eng.encode((torch.randn(B, 3000, wide.n_mels, generator=torch.Generator().manual_seed(3)) * 0.5).half().cuda())
lp = {}
for name, kw in (("v0 graph", dict(step_variant=0)), ("v1 graph", dict(step_variant=1)), ("v3 graph", dict(step_variant=3)),
                 ("v0 eager", dict(step_variant=0, use_graph=False)), ("v1 eager", dict(step_variant=1, use_graph=False)),
                 ("v5 graph (GEMV launch + fused kernel's attention role)", dict(step_variant=5)), ("v1 split1", dict(step_variant=1, cross_split=1)), ("v1 split4", dict(step_variant=1, cross_split=4))):
    for n in (7, 8):
        o = eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=n, **kw)
        lp[(name, n)] = o.sum_logprob.cpu().numpy().astype(np.float64).copy()
    print(name, "logprob of token 8, rows 0..5:", np.round((lp[(name, 8)] - lp[(name, 7)])[:6], 6).tolist(), flush=True)
