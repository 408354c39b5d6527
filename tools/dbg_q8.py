"""The open question of DESIGN 5c on a lab build (python tools/build_lab.py dumpq8 LAB_DUMP_Q8): step variant 6 runs the
int8 cross-Q GEMV as skinny_kernel AND as the fused launch's GEMV role on the same input, compares the two queries on the
device and keeps, for the first mismatch, what both epilogues saw: t (sum of the eight partial tiles), row scale, bias, result.
    python tools/dbg_q8.py tools/_bin/libwxhip_dumpq8.so"""
import sys, os, ctypes as C, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperx_mlx_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from whisperx_mlx_amd import weights
from whisperx_mlx_amd.engine import WhisperHipEngine
from whisperx_mlx_amd.tokenizer import get_tokenizer
L = _lib.lib()
L.wx_debug_read.restype = C.c_int
L.wx_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_float), C.c_void_p]
bits = lambda f: "%08x" % struct.unpack("<I", struct.pack("<f", f))[0]
wide = weights.ModelDimensions(128, 1500, 1280, 20, 1, 51866, 448, 1280, 20, 1)
ckw = weights.random_checkpoint(wide, seed=5, std=0.03, emb_std=0.03)
for q8 in (True, False):
    packed = weights.pack(ckw, wide, "cuda")
    if q8:
        packed = weights.quantize_packed_decoder(packed, wide)
    eng = WhisperHipEngine(wide, packed, max_batch=16)
    tok = get_tokenizer(wide.n_vocab)
    enc = eng.encode((torch.randn(16, 3000, wide.n_mels, generator=torch.Generator().manual_seed(3)) * 0.5).half().cuda())
    for n in (7, 8, 24):
        eng.decode(enc, tok, tok.sot_sequence(), rules=0, forced_len=n, step_variant=6, use_graph=False)
        rec, fl = (C.c_ulonglong * 8)(), (C.c_float * 16)()
        assert L.wx_debug_read(eng.ctx, rec, fl, eng._s) == 0
        r = list(rec)
        print(f"q8 {q8} n {n}: {r[7]} comparisons, mismatching query granules {r[0]}; first: pos {r[1]} layer {r[2]} row {r[3]} col {r[4]} granule {r[5]:08x} memory {r[6]:08x}", flush=True)
        if r[0]:
            for name, o in (("skinny_kernel", 0), ("fused GEMV role", 8)):
                v = list(fl)[o: o + 8]
                print(f"   {name:16s} col 608: t {v[0]!r} ({bits(v[0])}) scale {v[1]!r} ({bits(v[1])}) bias {v[2]!r} -> {v[3]!r} ({bits(v[3])}) | col 609: t {v[4]!r} ({bits(v[4])}) scale {bits(v[5])} bias {v[6]!r} -> {v[7]!r} ({bits(v[7])})")
    eng.close()
