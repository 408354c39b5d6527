import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from whisperx_mlx_amd import _lib, weights, engine as E
from whisperx_mlx_amd.engine import WhisperHipEngine
from whisperx_mlx_amd.tokenizer import get_tokenizer
L = _lib.lib()
L.wx_debug_read.restype = C.c_int
L.wx_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_void_p]
wide = weights.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 1)
ckw = weights.random_checkpoint(wide, seed=5, std=0.03, emb_std=0.03)
for q8 in (True, False):
    packed = weights.pack(ckw, wide, "cuda")
    if q8:
        packed = weights.quantize_packed_decoder(packed, wide)
    eng = WhisperHipEngine(wide, packed, max_batch=16)
    tok = get_tokenizer(wide.n_vocab)
    enc = eng.encode((torch.randn(16, 3000, wide.n_mels, generator=torch.Generator().manual_seed(3)) * 0.5).half().cuda())
    for n in (7, 8, 12, 24):
        o = eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=n, step_variant=6, use_graph=False)
        rec = (C.c_ulonglong * 8)()
        assert L.wx_debug_read(eng.ctx, rec, eng._s) == 0
        r = list(rec)
        print("q8", q8, "n", n, "mismatching query granules:", r[0], "first: pos", r[1], "layer", r[2], "row", r[3], "col", r[4], "granule %08x" % r[5], "memory %08x" % r[6], flush=True)
    eng.close()
