"""-m gpu: numerics of the HIP building blocks (GEMM, skinny GEMM, LayerNorm, attention)
against a plain PyTorch fp32 statement of the same op, through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests import gpu_util as G   # noqa: E402


def _rand(shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().cuda()


@pytest.mark.parametrize("RX,RY,K", [(128, 128, 64), (256, 384, 128), (1280, 1500, 1280), (384, 3000, 240),
                                     (100, 77, 72), (32, 333, 768)])
def test_gemm_plain(RX, RY, K):
    eng, _ = G.tiny_engine()
    X, Y = _rand((RX, K), 0.5, 1), _rand((RY, K), 0.5, 2)
    out = G.gemm(eng, X, Y)
    ref = Y.float() @ X.float().T
    assert G.rel_err(out, ref) < 2e-3


def test_gemm_asymmetric_identity():
    """A = I check with an asymmetric operand: catches a transposed C write."""
    eng, _ = G.tiny_engine()
    K = 128
    X = torch.eye(K).half().cuda()
    Y = (torch.arange(256 * K).reshape(256, K) % 97).half().cuda()
    out = G.gemm(eng, X, Y)                    # out[y][x] = Y[y][x]
    assert torch.equal(out, Y)
    out2 = G.gemm(eng, Y, X)                   # out[y][x] = Y[x][y]
    assert torch.equal(out2, Y.T.contiguous())


def test_gemm_epilogues():
    eng, _ = G.tiny_engine()
    RX, RY, K = 256, 200, 192
    X, Y = _rand((RX, K), 0.3, 3), _rand((RY, K), 0.3, 4)
    bias_x, bias_y = _rand((RX,), 0.5, 5), _rand((RY,), 0.5, 6)
    R = _rand((RY, RX), 1.0, 7)
    base = Y.float() @ X.float().T
    out = G.gemm(eng, X, Y, bias=bias_x, gelu=True, R=R)
    ref = F.gelu(base + bias_x.float()[None, :]) + R.float()
    assert G.rel_err(out, ref) < 2e-3
    out = G.gemm(eng, X, Y, bias=bias_y, bias_on_y=True)
    assert G.rel_err(out, base + bias_y.float()[:, None]) < 2e-3
    # in-place residual (out aliases R), as the encoder's out-proj / fc2 do
    Rc = R.clone()
    out = G.gemm(eng, X, Y, bias=bias_x, R=Rc, out=Rc)
    assert G.rel_err(out, base + bias_x.float()[None, :] + R.float()) < 2e-3


@pytest.mark.parametrize("RX,RY,K", [(512, 512, 128), (768, 1499, 768), (1024, 700, 192), (1499, 768, 768), (520, 515, 3840)])
def test_gemm_256_tile_epilogues(RX, RY, K):
    """Shapes that take the 256 x 256 x 64 kernel (RX, RY >= 512, K % 64 == 0): even / odd K-tile counts, ragged last
    tiles in both directions (RX not a multiple of 8: scalar column tail), every epilogue form."""
    eng, _ = G.tiny_engine()
    X, Y = _rand((RX, K), 0.3, 30), _rand((RY, K), 0.3, 31)
    bias_x, bias_y = _rand((RX,), 0.5, 32), _rand((RY,), 0.5, 33)
    R = _rand((RY, RX), 1.0, 34)
    base = Y.float() @ X.float().T
    assert G.rel_err(G.gemm(eng, X, Y), base) < 2e-3
    out = G.gemm(eng, X, Y, bias=bias_x, gelu=True, R=R)
    assert G.rel_err(out, F.gelu(base + bias_x.float()[None, :]) + R.float()) < 2e-3
    out = G.gemm(eng, X, Y, bias=bias_y, bias_on_y=True)
    assert G.rel_err(out, base + bias_y.float()[:, None]) < 2e-3
    Rc = R.clone()
    out = G.gemm(eng, X, Y, bias=bias_x, R=Rc, out=Rc)
    assert G.rel_err(out, base + bias_x.float()[None, :] + R.float()) < 2e-3


@pytest.mark.parametrize("RX,RY,K", [(512, 512, 128), (768, 1499, 768), (1280, 6000, 1280), (5120, 4500, 1280), (1280, 3100, 5120),
                                     (1499, 768, 768), (520, 515, 3840), (3840, 70000, 384)])
def test_gemm_tile_pipelined_kernel_equals_one_tile_per_block(RX, RY, K):
    """gemm_pipe_kernel (round 5: one staging pipeline over all the tiles of a block -- the next tile's first k-tiles are
    fetched under the current tile's last MFMAs, the epilogue transposes through the 64 KiB of LDS the next tile's operands
    do not occupy) against gemm_8phase_kernel (one tile per block), which wx_set_encoder_cap(-1) selects: the same bits,
    for every epilogue form, ragged edges in both directions, blocks that walk 1 .. 16 tiles, and capped grids."""
    eng, _ = G.tiny_engine()
    X, Y = _rand((RX, K), 0.3, 40), _rand((RY, K), 0.3, 41)
    bias_x, bias_y = _rand((RX,), 0.5, 42), _rand((RY,), 0.5, 43)
    R = _rand((RY, RX), 1.0, 44)

    def forms():
        yield G.gemm(eng, X, Y)
        yield G.gemm(eng, X, Y, bias=bias_x, gelu=True)
        yield G.gemm(eng, X, Y, bias=bias_y, bias_on_y=True)
        Rc = R.clone()
        yield G.gemm(eng, X, Y, bias=bias_x, R=Rc, out=Rc)          # in place, as the encoder's out-proj / FC2

    try:
        eng.set_encoder_cap(-1)
        ref = [o.clone() for o in forms()]
        base = Y.float() @ X.float().T
        assert G.rel_err(ref[0], base) < 2e-3
        for cap in (0, 8, 64):
            eng.set_encoder_cap(cap)
            for k, o in enumerate(forms()):
                assert torch.equal(o, ref[k]), (cap, k)
    finally:
        eng.set_encoder_cap(0)
    eng.check_status()


def test_gemm_strided_rows_conv():
    """Implicit-GEMM convolution: a row of Y is k consecutive rows of a channels-last tensor."""
    eng, _ = G.tiny_engine()
    C_in, C_out, T = 64, 128, 500
    x = _rand((T + 2, C_in), 0.5, 8)
    x[0] = 0
    x[-1] = 0
    w = _rand((C_out, C_in, 3), 0.1, 9)
    wk = w.permute(0, 2, 1).reshape(C_out, 3 * C_in).contiguous()
    # stride 2: output t reads padded rows 2t .. 2t+2
    T_out = T // 2
    Y = torch.as_strided(x, (T_out, 3 * C_in), (2 * C_in, 1))
    out = G.gemm(eng, wk, Y, K=3 * C_in)
    ref = F.conv1d(x[1:-1].float().T[None], w.float(), stride=2, padding=1)[0].T
    assert G.rel_err(out, ref) < 2e-3


@pytest.mark.parametrize("M,N,K", [(16, 1280, 1280), (16, 5120, 1280), (16, 1280, 5120), (4, 384, 128), (3, 51865, 128),
                                   (16, 100, 96)])
def test_skinny(M, N, K):
    eng, _ = G.tiny_engine()
    A, W = _rand((M, K), 1.0, 10), _rand((N, K), 0.05, 11)
    bias, R = _rand((N,), 0.5, 12), _rand((M, N), 1.0, 13)
    base = A.float() @ W.float().T
    assert G.rel_err(G.skinny(eng, A, W), base) < 2e-3
    out = G.skinny(eng, A, W, bias=bias, gelu=True, R=R)
    assert G.rel_err(out, F.gelu(base + bias.float()) + R.float()) < 2e-3
    out = G.skinny(eng, A, W, f32=True)
    assert G.rel_err(out, base) < 1e-3


@pytest.mark.parametrize("M,N,K,n_cu", [(16, 1280, 1280, 0), (48, 1280, 1280, 0), (64, 5120, 960, 0), (48, 5120, 1280, 0), (48, 1280, 5120, 0),
                                        (33, 3840, 1280, 0), (5, 384, 384, 0), (20, 51865, 128, 2048), (64, 100, 96, 7)])
def test_skinny_mt(M, N, K, n_cu):
    """M-tiled column-balanced GEMV (decode step variant 3 / coalesced requests) vs fp32 torch."""
    eng, _ = G.tiny_engine()
    A, W = _rand((M, K), 1.0, 40), _rand((N, K), 0.05, 41)
    bias, R = _rand((N,), 0.5, 42), _rand((M, N), 1.0, 43)
    base = A.float() @ W.float().T
    assert G.rel_err(G.skinny_mt(eng, A, W, n_cu=n_cu), base) < 2e-3
    out = G.skinny_mt(eng, A, W, bias=bias, gelu=True, R=R, n_cu=n_cu)
    assert G.rel_err(out, F.gelu(base + bias.float()) + R.float()) < 2e-3
    assert G.rel_err(G.skinny_mt(eng, A, W, f32=True, n_cu=n_cu), base) < 1e-3
    if K <= 1280:
        g, b = _rand((K,), 0.2, 44) + 1, _rand((K,), 0.2, 45)
        a_ln = F.layer_norm(A.float(), (K,), g.float(), b.float(), 1e-5).half().float()
        assert G.rel_err(G.skinny_mt(eng, A, W, ln=(g, b), f32=True, n_cu=n_cu), a_ln @ W.float().T) < 2e-3


@pytest.mark.parametrize("M,N,K,balanced", [(16, 1280, 1280, False), (16, 5120, 1280, False), (16, 1280, 5120, False), (7, 384, 384, False),
                                            (16, 1280, 1280, True), (48, 1280, 5120, True), (40, 5120, 1280, True), (3, 100, 96, False)])
def test_skinny_int8_weights(M, N, K, balanced):
    """int8 decode GEMV: bytes q + 128 dequantised in registers; reference = fp32 matmul on the oracle's
    quantise -> dequantise weights.  The product's quantiser must produce the oracle's integers and scales exactly."""
    from oracle import quant as OQ
    from whisperx_mlx_amd import weights as WT
    eng, _ = G.tiny_engine()
    A, W = _rand((M, K), 1.0, 50), _rand((N, K), 0.05, 51)
    W[5] = 0            # an all-zero row: scale 1, q 0
    Wq, sc = WT.quantize_rows_int8(W)
    q_ref, s_ref = OQ.quantize(W.float().cpu().numpy())
    assert np.array_equal(Wq.cpu().numpy().astype(np.int16) - 128, q_ref.astype(np.int16)) and np.array_equal(sc.cpu().numpy(), s_ref)
    Wd = torch.from_numpy(OQ.dequantize(q_ref, s_ref)).cuda()
    bias, R = _rand((N,), 0.5, 52), _rand((M, N), 1.0, 53)
    base = A.float() @ Wd.T
    assert G.rel_err(G.skinny_q8(eng, A, Wq, sc, f32=True, balanced=balanced), base) < 1e-3
    out = G.skinny_q8(eng, A, Wq, sc, bias=bias, gelu=True, R=R, balanced=balanced)
    assert G.rel_err(out, F.gelu(base + bias.float()) + R.float()) < 2e-3
    if K <= 1280:
        g, b = _rand((K,), 0.2, 54) + 1, _rand((K,), 0.2, 55)
        a_ln = F.layer_norm(A.float(), (K,), g.float(), b.float(), 1e-5).half().float()
        assert G.rel_err(G.skinny_q8(eng, A, Wq, sc, ln=(g, b), f32=True, balanced=balanced), a_ln @ Wd.T) < 2e-3
    # per-tensor granularity (the reference sketch's) through the same kernel
    Wq1, sc1 = WT.quantize_rows_int8(W, "tensor")
    q1, s1 = OQ.quantize(W.float().cpu().numpy(), "tensor")
    assert np.array_equal(Wq1.cpu().numpy().astype(np.int16) - 128, q1.astype(np.int16)) and np.array_equal(sc1.cpu().numpy(), s1)
    assert G.rel_err(G.skinny_q8(eng, A, Wq1, sc1, f32=True, balanced=balanced), A.float() @ torch.from_numpy(OQ.dequantize(q1, s1)).cuda().T) < 1e-3


@pytest.mark.parametrize("M,N,K,ln,tile_n,wide", [(16, 1280, 1280, True, 0, False), (16, 3840, 1280, True, 0, False), (16, 1280, 5120, False, 8, True),
                                                  (16, 1280, 1280, False, 8, False), (64, 5120, 1280, True, 0, False), (40, 1280, 5120, False, 8, True),
                                                  (64, 1280, 5120, False, 16, True), (5, 384, 384, True, 16, False), (16, 512, 2048, False, 4, False)])
def test_skinny_tile_blocked_weights_same_bits(M, N, K, ln, tile_n, wide):
    """the weight layout the decode step streams ([N/16][K/32][16][32], one contiguous KiB per fragment load): packing
    is a permutation, and fp16 and int8 GEMVs on the packed copy give the bits of the same launch on the row-major
    matrix -- LayerNorm prologue, bias + GELU + residual, narrow tiles, more than 16 rows, the K = 4d forms (16 waves at
    <= 16 rows, the eight-wave kernel beyond)"""
    from whisperx_mlx_amd import weights as WT
    eng, _ = G.tiny_engine()
    A, W = _rand((M, K), 1.0, 60), _rand((N, K), 0.05, 61)
    bias, R = _rand((N,), 0.5, 62), _rand((M, N), 1.0, 63)
    Wb = G.pack_gemv_weight(eng, W)
    assert torch.equal(Wb, W.view(N // 16, 16, K // 32, 32).permute(0, 2, 1, 3).contiguous().view(N, K))
    lnp = (_rand((K,), 0.2, 64) + 1, _rand((K,), 0.2, 65)) if ln else None
    kw = dict(bias=bias, gelu=True, R=R, ln=lnp, tile_n=tile_n, wide_block=wide)
    row_major = G.skinny_ex(eng, A, W=W, **kw)
    assert torch.equal(G.skinny_ex(eng, A, W=Wb, w_blocked=True, **kw), row_major)
    if not wide:
        assert torch.equal(row_major, G.skinny(eng, A, W, bias=bias, gelu=True, R=R, ln=lnp, tile_n=tile_n))
    assert G.rel_err(row_major, F.gelu((F.layer_norm(A.float(), (K,), lnp[0].float(), lnp[1].float(), 1e-5).half().float() if ln else A.float())
                                       @ W.float().T + bias.float()) + R.float()) < 3e-3
    Wq, sc = WT.quantize_rows_int8(W)
    Wqb = G.pack_gemv_weight(eng, Wq)
    assert torch.equal(Wqb, Wq.view(N // 16, 16, K // 32, 32).permute(0, 2, 1, 3).contiguous().view(N, K))
    assert torch.equal(G.skinny_ex(eng, A, Wq=Wqb, scale=sc, w_blocked=True, **kw), G.skinny_ex(eng, A, Wq=Wq, scale=sc, **kw))


@pytest.mark.parametrize("tile_n", [1, 3, 5, 10, 15, 8])
def test_skinny_any_tile_width(tile_n):
    """tile_n columns per block for any 1..16 (N / #CU gives one balanced round of blocks: 5, 10, 15 on 256 CUs): same
    results as the 16-column tiles, with every epilogue form, ragged N and the LayerNorm prologue."""
    eng, _ = G.tiny_engine()
    for (M, N, K) in ((16, 1280, 1280), (16, 1280, 5120), (7, 101, 96)):
        A, W = _rand((M, K), 1.0, 60), _rand((N, K), 0.05, 61)
        bias, R = _rand((N,), 0.5, 62), _rand((M, N), 1.0, 63)
        base = A.float() @ W.float().T
        assert torch.equal(G.skinny(eng, A, W, f32=True, tile_n=tile_n), G.skinny(eng, A, W, f32=True))   # same summation order
        out = G.skinny(eng, A, W, bias=bias, gelu=True, R=R, tile_n=tile_n)
        assert G.rel_err(out, F.gelu(base + bias.float()) + R.float()) < 2e-3
        if K <= 1280:
            g, b = _rand((K,), 0.2, 64) + 1, _rand((K,), 0.2, 65)
            a_ln = F.layer_norm(A.float(), (K,), g.float(), b.float(), 1e-5).half().float()
            assert G.rel_err(G.skinny(eng, A, W, ln=(g, b), f32=True, tile_n=tile_n), a_ln @ W.float().T) < 2e-3


def test_skinny_layernorm_fused():
    eng, _ = G.tiny_engine()
    M, N, K = 16, 256, 1280
    A, W = _rand((M, K), 2.0, 14) + 0.5, _rand((N, K), 0.05, 15)
    g, b = _rand((K,), 0.2, 16) + 1, _rand((K,), 0.2, 17)
    a_ln = F.layer_norm(A.float(), (K,), g.float(), b.float(), 1e-5).half().float()
    out = G.skinny(eng, A, W, ln=(g, b), f32=True)
    assert G.rel_err(out, a_ln @ W.float().T) < 2e-3


@pytest.mark.parametrize("rows,d", [(7, 128), (1500, 384), (100, 768), (33, 1280), (10, 512)])
def test_layernorm(rows, d):
    eng, _ = G.tiny_engine()
    x = _rand((rows, d), 3.0, 18) + 1.0
    g, b = _rand((d,), 0.3, 19) + 1, _rand((d,), 0.3, 20)
    y = G.layernorm(eng, x, g, b)
    ref = F.layer_norm(x.float(), (d,), g.float(), b.float(), 1e-5)
    assert (y.float() - ref.cuda()).abs().max().item() < 1e-2


@pytest.mark.parametrize("B,T,H,lens", [(1, 1500, 2, None), (2, 200, 3, None), (3, 333, 1, [333, 64, 1]),
                                        (2, 1499, 2, [1499, 700])])
def test_attention(B, T, H, lens):
    eng, _ = G.tiny_engine()
    d = H * 64
    q, k, v = _rand((B, T, d), 1.0, 21), _rand((B, T, d), 1.0, 22), _rand((B, T, d), 1.0, 23)
    o = G.attention(eng, q, k, v, lens)
    for b in range(B):
        n = T if lens is None else lens[b]
        qh = q[b, :n].float().view(n, H, 64).transpose(0, 1)
        kh = k[b, :n].float().view(n, H, 64).transpose(0, 1)
        vh = v[b, :n].float().view(n, H, 64).transpose(0, 1)
        p = torch.softmax(qh @ kh.transpose(1, 2) * 0.125, dim=-1)
        ref = (p @ vh).transpose(0, 1).reshape(n, d)
        assert G.rel_err(o[b, :n], ref) < 4e-3, (b, n)


def test_attention_spiked_key():
    """Forces the online-softmax rescale: one key dominates late in the sequence."""
    eng, _ = G.tiny_engine()
    B, T, d = 1, 640, 64
    q, k, v = _rand((B, T, d), 1.0, 24), _rand((B, T, d), 1.0, 25), _rand((B, T, d), 1.0, 26)
    k[0, 600] = q[0, 5] * 4.0
    o = G.attention(eng, q, k, v)
    p = torch.softmax(q[0].float() @ k[0].float().T * 0.125, dim=-1)
    assert G.rel_err(o[0], p @ v[0].float()) < 4e-3


@pytest.mark.parametrize("M,N,K,ks", [(16, 1280, 1280, 1), (16, 1280, 1280, 7), (16, 1280, 5120, 7), (16, 3840, 1280, 1),
                                      (4, 128, 512, 8), (3, 51865, 128, 1), (16, 100, 96, 3)])
def test_skinny2_splitk_and_resln(M, N, K, ks):
    eng, _ = G.tiny_engine()
    A, W = _rand((M, K), 1.0, 30), _rand((N, K), 0.05, 31)
    bias = _rand((N,), 0.5, 32)
    base = A.float() @ W.float().T
    if ks == 1:
        assert G.rel_err(G.skinny2(eng, A, W), base) < 2e-3
        assert G.rel_err(G.skinny2(eng, A, W, bias=bias, gelu=True), F.gelu(base + bias.float())) < 2e-3
        assert G.rel_err(G.skinny2(eng, A, W, f32=True), base) < 1e-3
        return
    part = G.skinny2(eng, A, W, ksplit=ks)
    assert G.rel_err(part[:, :M].sum(0), base) < 1e-3
    # run-to-run bitwise reproducible (fixed summation order, no atomics)
    assert torch.equal(part, G.skinny2(eng, A, W, ksplit=ks))
    if N <= 2048:
        x = _rand((M, N), 1.0, 33)
        g, b = _rand((N,), 0.2, 34) + 1, _rand((N,), 0.2, 35)
        x_new, xn = G.resln(eng, x, part, bias, g, b)
        ref_x = (x.float() + bias.float() + part[:, :M].sum(0)).half()
        assert torch.equal(x_new, ref_x) or G.rel_err(x_new, ref_x) < 1e-3
        ref_n = F.layer_norm(ref_x.float(), (N,), g.float(), b.float(), 1e-5)
        assert (xn.float() - ref_n).abs().max().item() < 1e-2


@pytest.mark.parametrize("M,N,K", [(16, 51866, 384), (5, 40003, 1280), (24, 36000, 128), (32, 51866, 512), (64, 51866, 1280),
                                   (41, 40003, 384), (48, 51866, 1280), (128, 51866, 1280), (117, 51866, 1280), (65, 40003, 384)])
def test_skinny2_tile_walking_blocks_equal_one_tile_per_block(M, N, K, monkeypatch):
    """the logits GEMV that walks several 16-column tiles per block (N >= 32768 columns, <= 64 rows; beyond 32 rows a second
    group of waves takes rows 32.. of the same tiles): correct against
    torch, ragged last tile, fp16 / fp32 / bias + GELU outputs, and the same bits as a launch over a sub-range of the
    columns that is too small to take the walking kernel (same per-element summation order)"""
    eng, _ = G.tiny_engine()
    A, W = _rand((M, K), 1.0, 40), _rand((N, K), 0.05, 41)
    bias = _rand((N,), 0.5, 42)
    base = A.float() @ W.float().T
    out32 = G.skinny2(eng, A, W, f32=True)
    assert G.rel_err(out32, base) < 1e-3
    assert G.rel_err(G.skinny2(eng, A, W), base) < 2e-3
    assert G.rel_err(G.skinny2(eng, A, W, bias=bias, gelu=True), F.gelu(base + bias.float())) < 2e-3
    lo = 16 * 1000
    small = G.skinny2(eng, A, W[lo: lo + 8000], f32=True)     # 500 tiles: one tile per block
    assert torch.equal(small, out32[:, lo: lo + 8000])
    if M > 16:          # a row's logits do not depend on how many rows its launch holds
        assert torch.equal(G.skinny2(eng, A[:16].contiguous(), W, f32=True), out32[:16])
        assert torch.equal(G.skinny2(eng, A[M - 7:].contiguous(), W, f32=True), out32[M - 7:])
    if M > 64:          # two launches of <= 64 rows (a single launch with every block twice, paired on one XCD, was measured: no gain)
        assert torch.equal(G.skinny2(eng, A[64: min(M, 80)].contiguous(), W, f32=True), out32[64: min(M, 80)])
        assert torch.equal(G.skinny2(eng, A[:64].contiguous(), W, f32=True), out32[:64])
        h16 = G.skinny2(eng, A, W, bias=bias, gelu=True)
        assert torch.equal(G.skinny2(eng, A[64:].contiguous(), W, bias=bias, gelu=True), h16[64:])


@pytest.mark.parametrize("M,N,K", [(16, 51866, 1280), (3, 51865, 384), (9, 40003, 512)])
def test_logits_gemv_with_fused_final_layernorm(M, N, K):
    """the decode step's last two launches in one: rows normalised once per tile-walking block (the decode GEMVs' dot2 /
    fma_mix LayerNorm), then the tied-embedding GEMV -- against fp32 torch and against LayerNorm kernel + plain GEMV"""
    eng, _ = G.tiny_engine()
    A, W = _rand((M, K), 2.0, 50) + 0.3, _rand((N, K), 0.05, 51)
    g, b = _rand((K,), 0.2, 52) + 1, _rand((K,), 0.2, 53)
    out = G.skinny2_ln(eng, A, W, g, b)
    ref_n = F.layer_norm(A.float(), (K,), g.float(), b.float(), 1e-5)
    ref = ref_n.half().float() @ W.float().T
    assert G.rel_err(out, ref) < 2e-3
    two = G.skinny2(eng, G.layernorm(eng, A, g, b), W, f32=True)
    assert G.rel_err(out, two) < 2e-3
    with pytest.raises(Exception):
        G.skinny2_ln(eng, A, W[:4096], g, b)          # too few columns for the tile-walking kernel


@pytest.mark.parametrize("M,N,K,ln", [(40, 1280, 1280, True), (48, 1280, 5120, False), (33, 3840, 1280, True), (17, 384, 384, False)])
def test_skinny_row_groups_over_16_rows(M, N, K, ln):
    """coalesced requests: more than 16 rows run the 16-row GEMV kernel over groups of 16 rows (grid.y) -- every row gets
    the bits it gets in a 16-row launch, ragged last group included"""
    eng, _ = G.tiny_engine()
    A, W = _rand((M, K), 1.0, 60), _rand((N, K), 0.05, 61)
    bias, R = _rand((N,), 0.5, 62), _rand((M, N), 1.0, 63)
    lnp = (_rand((K,), 0.2, 64) + 1, _rand((K,), 0.2, 65)) if ln else None
    out = G.skinny(eng, A, W, bias=bias, ln=lnp, R=R, tile_n=8)
    x = F.layer_norm(A.float(), (K,), lnp[0].float(), lnp[1].float(), 1e-5).half().float() if ln else A.float()
    ref = x @ W.float().T + bias.float() + R.float()
    assert G.rel_err(out, ref) < 3e-3
    for g0 in range(0, M, 16):
        part = G.skinny(eng, A[g0: g0 + 16].contiguous(), W, bias=bias, ln=lnp, R=R[g0: g0 + 16].contiguous(), tile_n=8)
        assert torch.equal(part, out[g0: g0 + 16]), g0


def test_skinny_q8_row_groups_over_16_rows():
    """int8 decoder weights with coalesced requests: row groups of the 16-row int8 kernel give each row the bits of a
    16-row launch, and agree with the M-tiled int8 kernel to rounding"""
    from whisperx_mlx_amd import weights as WT
    eng, _ = G.tiny_engine()
    M, N, K = 37, 1280, 1280
    A, W = _rand((M, K), 1.0, 70), _rand((N, K), 0.05, 71)
    q, sc = WT.quantize_rows_int8(W.cpu())
    q, sc = q.cuda(), sc.cuda()
    bias = _rand((N,), 0.5, 72)
    out = G.skinny_q8(eng, A, q, sc, bias=bias)
    ref = A.float() @ ((q.float() - 128.0) * sc[:, None]).T + bias.float()
    assert G.rel_err(out, ref) < 3e-3
    for g0 in range(0, M, 16):
        part = G.skinny_q8(eng, A[g0: g0 + 16].contiguous(), q, sc, bias=bias)
        assert torch.equal(part, out[g0: g0 + 16]), g0
    assert G.rel_err(G.skinny_q8(eng, A, q, sc, bias=bias, balanced=True), out) < 2e-3
