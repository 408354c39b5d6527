"""Helpers shared by the -m gpu tests: a tiny-dims engine and raw C-ABI callers."""
import ctypes as C
import functools

import numpy as np
import torch

from whisperx_mlx_amd import _lib, weights
from whisperx_mlx_amd.engine import WhisperHipEngine

TEST_DIMS = weights.ModelDimensions(80, 1500, 128, 2, 2, 51865, 448, 128, 2, 2)
TEST_HEADS = [(0, 1), (1, 0), (1, 1)]


@functools.lru_cache(maxsize=None)
def tiny_engine(seed=0, emb_std=0.1, std=0.2, max_batch=4):
    """2-layer, d=128 Whisper with the real vocabulary/special-token layout.  std/emb_std
    are chosen so that the random model emits varied text/timestamp/EOT tokens."""
    ck = weights.random_checkpoint(TEST_DIMS, seed=seed, std=std, emb_std=emb_std)
    packed = weights.pack(ck, TEST_DIMS, "cuda")
    eng = WhisperHipEngine(TEST_DIMS, packed, max_batch=max_batch, alignment_heads=TEST_HEADS)
    ck32 = {k: v.float() for k, v in ck.items()}
    return eng, ck32


def dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def gemm(eng, X, Y, K=None, bias=None, bias_on_y=False, R=None, gelu=False, out=None, ldo=None):
    """out[y][x] = epi(sum_k X[x][k] Y[y][k])"""
    L = _lib.lib()
    RX, RY = X.shape[0], Y.shape[0]
    K = K or X.shape[1]
    if out is None:
        out = torch.zeros(RY, RX, dtype=torch.float16, device="cuda")
    ldo = ldo or out.stride(0)
    torch.cuda.synchronize()
    rc = L.wx_gemm_f16(eng.ctx, _lib.ptr(X), X.stride(0), RX, _lib.ptr(Y), Y.stride(0), RY, K, _lib.ptr(bias),
                       int(bias_on_y), _lib.ptr(R), R.stride(0) if R is not None else 0, _lib.ptr(out), ldo,
                       int(gelu), None)
    _lib.check(eng.ctx, rc, "wx_gemm_f16")
    torch.cuda.synchronize()
    return out


def skinny(eng, A, W, bias=None, ln=None, R=None, gelu=False, f32=False, tile_n=0):
    L = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    out_h = None if f32 else torch.zeros(M, N, dtype=torch.float16, device="cuda")
    out_f = torch.zeros(M, N, dtype=torch.float32, device="cuda") if f32 else None
    g, b = (ln if ln is not None else (None, None))
    torch.cuda.synchronize()
    rc = L.wx_skinny_f16(eng.ctx, _lib.ptr(A), A.stride(0), M, _lib.ptr(W), W.stride(0), N, K, _lib.ptr(bias),
                         _lib.ptr(g), _lib.ptr(b), _lib.ptr(R), R.stride(0) if R is not None else 0,
                         _lib.ptr(out_h), _lib.ptr(out_f), N, int(gelu), int(tile_n), None)
    _lib.check(eng.ctx, rc, "wx_skinny_f16")
    torch.cuda.synchronize()
    return out_f if f32 else out_h


def skinny_mt(eng, A, W, bias=None, ln=None, R=None, gelu=False, f32=False, n_cu=0):
    L = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    out_h = None if f32 else torch.zeros(M, N, dtype=torch.float16, device="cuda")
    out_f = torch.zeros(M, N, dtype=torch.float32, device="cuda") if f32 else None
    g, b = (ln if ln is not None else (None, None))
    torch.cuda.synchronize()
    rc = L.wx_skinny_mt_f16(eng.ctx, _lib.ptr(A), A.stride(0), M, _lib.ptr(W), W.stride(0), N, K, _lib.ptr(bias),
                            _lib.ptr(g), _lib.ptr(b), _lib.ptr(R), R.stride(0) if R is not None else 0,
                            _lib.ptr(out_h), _lib.ptr(out_f), N, int(gelu), n_cu, None)
    _lib.check(eng.ctx, rc, "wx_skinny_mt_f16")
    torch.cuda.synchronize()
    return out_f if f32 else out_h


def skinny_q8(eng, A, Wq, scale, bias=None, ln=None, R=None, gelu=False, f32=False, balanced=False):
    L = _lib.lib()
    M, K = A.shape
    N = Wq.shape[0]
    out_h = None if f32 else torch.zeros(M, N, dtype=torch.float16, device="cuda")
    out_f = torch.zeros(M, N, dtype=torch.float32, device="cuda") if f32 else None
    g, b = (ln if ln is not None else (None, None))
    torch.cuda.synchronize()
    rc = L.wx_skinny_q8(eng.ctx, _lib.ptr(A), A.stride(0), M, _lib.ptr(Wq), _lib.ptr(scale), Wq.stride(0), N, K, _lib.ptr(bias),
                        _lib.ptr(g), _lib.ptr(b), _lib.ptr(R), R.stride(0) if R is not None else 0,
                        _lib.ptr(out_h), _lib.ptr(out_f), N, int(gelu), int(balanced), None)
    _lib.check(eng.ctx, rc, "wx_skinny_q8")
    torch.cuda.synchronize()
    return out_f if f32 else out_h


def pack_gemv_weight(eng, W):
    """row-major [N][K] fp16 or uint8 -> the tile-blocked layout the decode step streams"""
    L = _lib.lib()
    N, K = W.shape
    out = torch.zeros_like(W)
    torch.cuda.synchronize()
    _lib.check(eng.ctx, L.wx_pack_gemv_weight(eng.ctx, _lib.ptr(W), N, K, W.element_size(), _lib.ptr(out), None), "wx_pack_gemv_weight")
    torch.cuda.synchronize()
    return out


def skinny_ex(eng, A, W=None, Wq=None, scale=None, bias=None, ln=None, R=None, gelu=False, f32=False, tile_n=0, wide_block=False,
              w_blocked=False):
    """wx_skinny_f16 / wx_skinny_q8 with every option of the decode step's launches (tile-blocked weights, the K = 4d forms)"""
    L = _lib.lib()
    M, K = A.shape
    N = (W if W is not None else Wq).shape[0]
    out_h = None if f32 else torch.zeros(M, N, dtype=torch.float16, device="cuda")
    out_f = torch.zeros(M, N, dtype=torch.float32, device="cuda") if f32 else None
    g, b = (ln if ln is not None else (None, None))
    torch.cuda.synchronize()
    rc = L.wx_skinny_ex(eng.ctx, _lib.ptr(A), A.stride(0), M, _lib.ptr(W), _lib.ptr(Wq), _lib.ptr(scale), K, N, K, _lib.ptr(bias),
                        _lib.ptr(g), _lib.ptr(b), _lib.ptr(R), R.stride(0) if R is not None else 0,
                        _lib.ptr(out_h), _lib.ptr(out_f), N, int(gelu), int(tile_n), int(wide_block), int(w_blocked), None)
    _lib.check(eng.ctx, rc, "wx_skinny_ex")
    torch.cuda.synchronize()
    return out_f if f32 else out_h


def layernorm(eng, x, g, b):
    L = _lib.lib()
    y = torch.zeros_like(x)
    torch.cuda.synchronize()
    rc = L.wx_layernorm_f16(eng.ctx, _lib.ptr(x), x.stride(0), _lib.ptr(g), _lib.ptr(b), _lib.ptr(y), y.stride(0),
                            x.shape[0], x.shape[1], None)
    _lib.check(eng.ctx, rc, "wx_layernorm_f16")
    torch.cuda.synchronize()
    return y


def attention(eng, q, k, v, lens=None):
    """q,k,v: (B, T, H*64) fp16 device.  Builds the V^T operand like the V-projection GEMM does."""
    L = _lib.lib()
    B, T, d = q.shape
    H = d // 64
    Tpad = (T + 63) // 64 * 64
    vt = torch.zeros(B, d, Tpad, dtype=torch.float16, device="cuda")
    vt[:, :, :T] = v.transpose(1, 2)
    o = torch.zeros(B, T, d, dtype=torch.float16, device="cuda")
    lens_t = None if lens is None else dev(lens, torch.int32)
    torch.cuda.synchronize()
    rc = L.wx_attention_f16(eng.ctx, _lib.ptr(q), d, T * d, _lib.ptr(k), d, T * d, _lib.ptr(vt), Tpad, d * Tpad,
                            _lib.ptr(o), d, T * d, _lib.ptr(lens_t), T, H, B, None)
    _lib.check(eng.ctx, rc, "wx_attention_f16")
    torch.cuda.synchronize()
    return o


def rel_err(a, b):
    a = a.float().cpu()
    b = b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def skinny2(eng, A, W, bias=None, ksplit=1, gelu=False, f32=False):
    """returns out (ksplit == 1) or the fp32 partial tiles [ksplit][16][N]"""
    L = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    out_h = out_f = part = None
    if ksplit > 1:
        part = torch.zeros(ksplit, 16, N, dtype=torch.float32, device="cuda")
    elif f32:
        out_f = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    else:
        out_h = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()
    rc = L.wx_skinny2_f16(eng.ctx, _lib.ptr(A), A.stride(0), M, _lib.ptr(W), W.stride(0), N, K, _lib.ptr(bias), ksplit,
                          int(gelu), _lib.ptr(out_h), _lib.ptr(out_f), N, _lib.ptr(part), None)
    _lib.check(eng.ctx, rc, "wx_skinny2_f16")
    torch.cuda.synchronize()
    return part if ksplit > 1 else (out_f if f32 else out_h)


def resln(eng, x, part, bias, g, b):
    L = _lib.lib()
    M, d = x.shape
    x = x.clone()
    xn = torch.zeros_like(x)
    ks = 0 if part is None else part.shape[0]
    torch.cuda.synchronize()
    rc = L.wx_resln_f16(eng.ctx, _lib.ptr(x), M, d, _lib.ptr(part), ks, _lib.ptr(bias), _lib.ptr(g), _lib.ptr(b),
                        _lib.ptr(xn), None)
    _lib.check(eng.ctx, rc, "wx_resln_f16")
    torch.cuda.synchronize()
    return x, xn


def skinny2_ln(eng, A, W, g, b):
    """fp32 out[m][n] = LN(A)[m] . W[n] through the tile-walking logits kernel with the fused LayerNorm"""
    L = _lib.lib()
    M, K = A.shape
    N = W.shape[0]
    out = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rc = L.wx_skinny2_ln_f16(eng.ctx, _lib.ptr(A), A.stride(0), M, _lib.ptr(W), W.stride(0), N, K, _lib.ptr(g), _lib.ptr(b),
                             _lib.ptr(out), N, None)
    _lib.check(eng.ctx, rc, "wx_skinny2_ln_f16")
    torch.cuda.synchronize()
    return out
