"""Per-kernel parity through the C ABI (GPU).  References are plain torch fp32 ops on the SAME
bf16-rounded operands, so only the fp32 accumulation order differs (tight tolerances)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import masr_amd  # noqa: E402
from masr_amd import _cabi  # noqa: E402


def P(t):
    return C.c_void_p(t.data_ptr())


def bits(t):
    """bf16 tensor -> its storage viewed as int16 (what the C ABI takes)"""
    return t.view(torch.int16)


@pytest.fixture(scope="module")
def L():
    return _cabi.lib()


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 367, 512), (4000, 512, 2560), (37, 24, 8), (513, 2048, 512)])
def test_gemm_nt(L, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    B = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g)
    Cc = torch.full((M, N + 3), 7.0, device="cuda")
    _cabi.check(L.masr_test_gemm(P(A), K, P(B), K, M, N, K, 0, P(bias), 1, P(Cc), N + 3, S()))
    ref = torch.relu(A.float() @ B.float().t() + bias)
    torch.testing.assert_close(Cc[:, :N], ref, rtol=2e-4, atol=2e-3)
    assert torch.all(Cc[:, N:] == 7.0)          # no write outside the N columns


@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (128, 256, 1000), (367, 512, 500), (1536, 512, 4000), (8, 16, 5),
                                   (2048, 2048, 1000), (2000, 1800, 777), (2048, 1536, 64)])     # the last three take the LDS-DMA 128x128 kernel
def test_gemm_reduction_major(L, M, N, K):
    """wgrad form: C[m][n] = sum_k A[k][m] B[k][n] (transposing LDS reads)."""
    g = torch.Generator(device="cuda").manual_seed(M * 3 + N + K)
    lda, ldb = (M + 7) // 8 * 8 + 8, (N + 7) // 8 * 8
    A = torch.zeros(K, lda, device="cuda").bfloat16()
    A[:, :M] = torch.randn(K, M, device="cuda", generator=g).bfloat16()
    B = torch.zeros(K, ldb, device="cuda").bfloat16()
    B[:, :N] = torch.randn(K, N, device="cuda", generator=g).bfloat16()
    Cc = torch.zeros(M, N, device="cuda")
    _cabi.check(L.masr_test_gemm(P(A), lda, P(B), ldb, M, N, K, 1, None, 0, P(Cc), N, S()))
    ref = A[:, :M].float().t() @ B[:, :N].float()
    torch.testing.assert_close(Cc, ref, rtol=2e-4, atol=2e-3 * (K ** 0.5) / 8)


@pytest.mark.parametrize("rows,N,K", [(4000, 512, 512), (1000, 1536, 512), (777, 256, 2048), (63, 128, 128), (500, 200, 328), (592, 512, 2048),
                                      (4000, 2560, 512), (1000, 2048, 2304), (300, 2048, 2100), (1, 64, 64),
                                      # more than one round of 256 x 256 tiles (2 x 17 x 16 = 544 on 256 CUs)
                                      (200, 4200, 4096)])
def test_wgrad_grouped(L, rows, N, K):
    """mk_gemm_wgrad_grouped (the Linear weight gradients of a backward pass as one grid of 256 x 256 tiles on eight waves, LDS-DMA ring:
    lin_wgrad / flush_wgrads): dW = dY^T X and the fused bias gradient, two members per launch; reductions that are not multiples of the
    32-row k tile (4000, 777, 63, 1), output dims that are not multiples of the tile (200 x 328)."""
    g = torch.Generator(device="cuda").manual_seed(rows + N + K)
    lddy, ldx = (N + 7) // 8 * 8 + 8, (K + 7) // 8 * 8
    dy = torch.zeros(rows, lddy, device="cuda").bfloat16(); dy[:, :N] = torch.randn(rows, N, device="cuda", generator=g).bfloat16()
    x = torch.zeros(rows, ldx, device="cuda").bfloat16(); x[:, :K] = torch.randn(rows, K, device="cuda", generator=g).bfloat16()
    dW = torch.full((N + 1, K), 7.0, device="cuda"); db = torch.zeros(N, device="cuda")
    dW2 = torch.zeros(N, K, device="cuda"); db2 = torch.zeros(N, device="cuda")
    _cabi.check(L.masr_test_wgrad_grouped(P(dy), lddy, P(x), ldx, P(dW), P(db), P(dW2), P(db2), rows, N, K, S()))
    ref = dy[:, :N].float().t() @ x[:, :K].float()
    torch.testing.assert_close(dW[:N], ref, rtol=2e-4, atol=2e-3 * (rows ** 0.5) / 8)
    assert torch.all(dW[N] == 7.0) and torch.equal(dW[:N], dW2)
    torch.testing.assert_close(db, dy[:, :N].float().sum(0), rtol=1e-4, atol=1e-3 * rows ** 0.5)
    assert torch.equal(db, db2)
    # an empty reduction is refused (the tile loop clamps its rows to rows - 1)
    assert L.masr_test_wgrad_grouped(P(dy), lddy, P(x), ldx, P(dW), P(db), None, None, 0, N, K, S()) != 0
    assert b"rows, N and K must be >= 1" in L.masr_last_error()


@pytest.mark.parametrize("members,first,rows,rows_rest,N,K", [(7, 3, 1000, 148, 512, 512), (5, 2, 4000, 592, 1536, 512), (11, 4, 333, 37, 200, 328),
                                                              (9, 8, 700, 64, 256, 256), (6, 0, 500, 500, 300, 512), (13, 5, 250, 31, 512, 2048)])
def test_wgrad_grouped_two_segment_tile_list(L, members, first, rows, rows_rest, N, K):
    """the engine's merged launch: the encoder-row members (long reductions) are dispatched first, every XCD takes a contiguous run of
    each segment; mixed reduction lengths, tile totals that are no multiple of the 8 XCDs.  Every member must equal its own product."""
    g = torch.Generator(device="cuda").manual_seed(members * 100 + first)
    lddy, ldx = (N + 7) // 8 * 8, (K + 7) // 8 * 8
    dy = torch.zeros(rows, lddy, device="cuda").bfloat16(); dy[:, :N] = torch.randn(rows, N, device="cuda", generator=g).bfloat16()
    x = torch.zeros(rows, ldx, device="cuda").bfloat16(); x[:, :K] = torch.randn(rows, K, device="cuda", generator=g).bfloat16()
    dW = torch.full((members, N, K), 7.0, device="cuda")
    _cabi.check(L.masr_test_wgrad_grouped_n(P(dy), lddy, P(x), ldx, P(dW), N * K, members, first, rows, rows_rest, N, K, S()))
    ref_long = dy[:, :N].float().t() @ x[:, :K].float()
    ref_short = dy[:rows_rest, :N].float().t() @ x[:rows_rest, :K].float()
    for i in range(members):
        r, n = (ref_long, rows) if (first == 0 or i < first) else (ref_short, rows_rest)
        torch.testing.assert_close(dW[i], r, rtol=2e-4, atol=2e-3 * (n ** 0.5) / 8)
    # the same members as one plain list: identical bits (each element is reduced by one workgroup over its rows in order)
    if first:
        dW2 = torch.zeros(first, N, K, device="cuda")
        _cabi.check(L.masr_test_wgrad_grouped_n(P(dy), lddy, P(x), ldx, P(dW2), N * K, first, 0, rows, rows, N, K, S()))
        assert torch.equal(dW2, dW[:first])


def test_gemm_exact_integers(L):
    """A = I check with an asymmetric B (catches transposed fragment maps exactly)."""
    M = N = K = 64
    A = torch.eye(M, device="cuda").bfloat16()
    B = (torch.arange(N * K, device="cuda").reshape(N, K) % 61).float().bfloat16()
    Cc = torch.zeros(M, N, device="cuda")
    _cabi.check(L.masr_test_gemm(P(A), K, P(B), K, M, N, K, 0, None, 0, P(Cc), N, S()))
    assert torch.equal(Cc, B.float().t())
    Cc.zero_()
    _cabi.check(L.masr_test_gemm(P(A), M, P(B.t().contiguous()), N, M, N, K, 1, None, 0, P(Cc), N, S()))
    assert torch.equal(Cc, B.float().t())       # A^T = I, B given as [k][n]


# widths 9 / 21 / 20 / 5 take the 8-pixel-wide tiles, 48 / 32 / 16 / 80 the 16-wide ones
@pytest.mark.parametrize("B_,H,W,CIN,COUT", [(2, 10, 9, 64, 64), (1, 33, 21, 64, 128), (3, 16, 20, 128, 128), (2, 7, 5, 128, 64),
                                             (1, 12, 48, 128, 128), (1, 9, 32, 64, 128), (1, 20, 16, 128, 64), (1, 19, 80, 64, 64),
                                             (4, 500, 80, 64, 64), (5, 250, 40, 128, 128), (3, 301, 37, 64, 128)])   # more tiles than resident workgroups
def test_conv3x3(L, B_, H, W, CIN, COUT):
    g = torch.Generator(device="cuda").manual_seed(CIN + COUT + H)
    x = torch.randn(B_, H, W, CIN, device="cuda", generator=g).bfloat16()           # NHWC
    w = (torch.randn(COUT, CIN, 3, 3, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(COUT, device="cuda", generator=g)
    wk = w.permute(0, 2, 3, 1).reshape(COUT, 9 * CIN).contiguous()                   # [co][tap*CIN+ci]
    out = torch.zeros(B_, H, W, COUT, device="cuda").bfloat16()
    _cabi.check(L.masr_test_conv3x3(P(x), P(wk), P(bias), 1, P(out), B_, H, W, CIN, COUT, S()))
    ref = torch.relu(torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1)).permute(0, 2, 3, 1)
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=1e-2)              # bf16 output rounding


@pytest.mark.parametrize("B_,H,W,CIN,COUT", [(2, 38, 80, 64, 64), (3, 50, 40, 128, 128), (1, 33, 21, 128, 64), (2, 301, 80, 64, 64), (1, 17, 48, 128, 128),
                                             (2, 45, 83, 64, 64), (2, 27, 41, 64, 128), (1, 1000, 83, 64, 64)])   # idim 83: odd widths, 16-wide tiles with a ragged edge
def test_conv3x3_mask_and_pool(L, B_, H, W, CIN, COUT):
    """the two epilogue flavours of the engine: dgrad (no bias, output zeroed where the ReLU mask is <= 0) and
    forward with the fused 2x2 max-pool (floor mode, odd H / W drop the last row / column)"""
    g = torch.Generator(device="cuda").manual_seed(7 * CIN + COUT + H)
    x = torch.randn(B_, H, W, CIN, device="cuda", generator=g).bfloat16()
    w = (torch.randn(COUT, CIN, 3, 3, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(COUT, device="cuda", generator=g)
    wk = w.permute(0, 2, 3, 1).reshape(COUT, 9 * CIN).contiguous()
    conv = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, padding=1).permute(0, 2, 3, 1)
    # dgrad flavour
    mask = torch.relu(torch.randn(B_, H, W, COUT, device="cuda", generator=g)).bfloat16()
    out = torch.full((B_, H, W, COUT), 7.0, device="cuda").bfloat16()
    _cabi.check(L.masr_test_conv3x3_ex(P(x), P(wk), None, 0, P(mask), P(out), None, B_, H, W, CIN, COUT, S()))
    ref = torch.where(mask.float() > 0, conv, torch.zeros_like(conv))
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=1e-2)
    assert torch.equal(out.float() == 0, ref.bfloat16().float() == 0) or ((out.float() == 0) ^ (mask.float() <= 0)).sum() < 1e-3 * out.numel()
    # forward + pool flavour
    pool = torch.full((B_, H // 2, W // 2, COUT), -1.0, device="cuda").bfloat16()
    _cabi.check(L.masr_test_conv3x3_ex(P(x), P(wk), P(bias), 1, None, P(out), P(pool), B_, H, W, CIN, COUT, S()))
    ref = torch.relu(conv + bias)
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=1e-2)
    pref = torch.nn.functional.max_pool2d(out.float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.equal(pool.float(), pref)                       # pooling the stored map is exact


@pytest.mark.parametrize("B_,H,W,CIN,COUT", [(2, 50, 42, 128, 256), (1, 37, 21, 256, 256), (2, 33, 42, 256, 128), (8, 200, 42, 256, 256)])
def test_conv3x3_256_channels_as_passes_of_128(L, B_, H, W, CIN, COUT):
    """The BLSTM front-end's convs (src/modules/encoder.py:215-298: 128 -> 256 -> 256 behind the first pool) run as passes of 128 OUTPUT
    channels of the streaming kernel over the same patches (ConvArgs::out_cstride / out_coff): forward (bias + ReLU) and the dgrad flavour
    through a bf16 ReLU mask with 256 channels per pixel, against torch; a poisoned output shows that both passes wrote their half of every
    pixel and nothing else."""
    g = torch.Generator(device="cuda").manual_seed(5 * CIN + COUT + H)
    x = torch.randn(B_, H, W, CIN, device="cuda", generator=g).bfloat16()
    w = (torch.randn(COUT, CIN, 3, 3, device="cuda", generator=g) * 0.04).bfloat16()
    bias = torch.randn(COUT, device="cuda", generator=g)
    wk = w.permute(0, 2, 3, 1).reshape(COUT, 9 * CIN).contiguous()
    conv = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, padding=1).permute(0, 2, 3, 1)
    out = torch.full((B_ * H * W * COUT + 256,), float("nan"), device="cuda").bfloat16()          # (+ a guard zone behind the map)
    _cabi.check(L.masr_test_conv3x3(P(x), P(wk), P(bias), 1, P(out), B_, H, W, CIN, COUT, S()))
    assert bool(torch.isnan(out[-256:].float()).all())
    torch.testing.assert_close(out[:-256].view(B_, H, W, COUT).float(), torch.relu(conv + bias), rtol=1e-2, atol=2e-2)
    mask = torch.relu(torch.randn(B_, H, W, COUT, device="cuda", generator=g)).bfloat16()
    out2 = torch.full((B_, H, W, COUT), float("nan"), device="cuda").bfloat16()
    _cabi.check(L.masr_test_conv3x3_ex(P(x), P(wk), None, 0, P(mask), P(out2), None, B_, H, W, CIN, COUT, S()))
    torch.testing.assert_close(out2.float(), torch.where(mask.float() > 0, conv, torch.zeros_like(conv)), rtol=1e-2, atol=2e-2)


def _sign_words(m):
    """[B,H,W,128] map -> [B,H,W,4] dwords: dword q, byte h, bit j = (channel 32 h + 8 q + j > 0)"""
    B_, H, W, _ = m.shape
    pos = (m.float() > 0).reshape(B_, H, W, 4, 4, 8).permute(0, 1, 2, 4, 3, 5).to(torch.int64)        # [.., q, h, j]
    sh = (8 * torch.arange(4, device=m.device).view(4, 1) + torch.arange(8, device=m.device).view(1, 8)).view(1, 1, 1, 1, 4, 8)
    return (pos << sh).sum((-1, -2))


@pytest.mark.parametrize("B_,H,W", [(2, 37, 40), (1, 64, 21), (3, 250, 40), (2, 33, 41)])     # (41: not an 8-wide-tile width -> fallback paths)
def test_conv3x3_relu_mask_as_sign_bits(L, B_, H, W):
    """conv3's forward (64 -> 128) leaves the ReLU mask of its output as four dwords per pixel; conv4's masked dgrad (128 <- 128) reads
    those instead of the bf16 map and runs on 32-row tiles.  The words must equal the stored map's signs exactly, and the dgrad's output
    must be identical to the bf16-masked launch's."""
    g = torch.Generator(device="cuda").manual_seed(H + W)
    x = torch.randn(B_, H, W, 64, device="cuda", generator=g).bfloat16()
    w = (torch.randn(128, 64, 3, 3, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(128, device="cuda", generator=g) * 0.3
    wk = w.permute(0, 2, 3, 1).reshape(128, 9 * 64).contiguous()
    out = torch.zeros(B_, H, W, 128, device="cuda").bfloat16()
    bits = torch.full((B_, H, W, 4), -1, device="cuda", dtype=torch.int32)
    _cabi.check(L.masr_test_conv3x3_sign_bits(P(x), P(wk), P(bias), 1, None, None, P(out), P(bits), B_, H, W, 64, 128, S()))
    ref = torch.relu(torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1)).permute(0, 2, 3, 1)
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=1e-2)
    assert torch.equal(bits.to(torch.int64) & 0xFFFFFFFF, _sign_words(out))
    # the masked dgrad with that map as its mask: bf16 mask vs sign words
    dy = torch.randn(B_, H, W, 128, device="cuda", generator=g).bfloat16()
    wd = (torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.05).bfloat16()
    wdk = wd.permute(0, 2, 3, 1).reshape(128, 9 * 128).contiguous()
    o1 = torch.full((B_, H, W, 128), 7.0, device="cuda").bfloat16()
    o2 = torch.full((B_, H, W, 128), 9.0, device="cuda").bfloat16()
    _cabi.check(L.masr_test_conv3x3_sign_bits(P(dy), P(wdk), None, 0, P(out), None, P(o1), None, B_, H, W, 128, 128, S()))
    _cabi.check(L.masr_test_conv3x3_sign_bits(P(dy), P(wdk), None, 0, P(out), P(bits), P(o2), None, B_, H, W, 128, 128, S()))
    assert torch.equal(o1, o2)
    conv = torch.nn.functional.conv2d(dy.float().permute(0, 3, 1, 2), wd.float(), None, padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(o2.float(), torch.where(out.float() > 0, conv, torch.zeros_like(conv)), rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("B_,H,W,CIN,COUT", [(2, 38, 80, 64, 64), (3, 50, 40, 128, 128), (2, 301, 80, 64, 64), (1, 17, 48, 128, 128),
                                             (2, 45, 83, 64, 64), (2, 27, 41, 64, 128), (2, 64, 40, 64, 128), (1, 33, 20, 128, 128),
                                             (2, 33, 41, 128, 128)])      # (odd H and W: a launch that drops the map skips the cropped row / column's tiles)
def test_conv3x3_pool_codes_and_their_backward(L, B_, H, W, CIN, COUT):
    """the pooling convs of the engine emit one byte per pooled element (which window position won / nothing passed the ReLU) and may
    drop the full-resolution map; the pool + ReLU backward runs from those bytes.  Codes are checked against the stored map of a second
    launch (first maximum in row-major order, as torch's max_pool2d backward), the backward against torch autograd through
    relu -> max_pool2d on that map.  Shapes cover both streaming kernels (16- and 8-wide tiles) and the fallback (patch kernel)."""
    g = torch.Generator(device="cuda").manual_seed(3 * CIN + COUT + H)
    x = torch.randn(B_, H, W, CIN, device="cuda", generator=g).bfloat16()
    w = (torch.randn(COUT, CIN, 3, 3, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(COUT, device="cuda", generator=g) * 0.3
    wk = w.permute(0, 2, 3, 1).reshape(COUT, 9 * CIN).contiguous()
    H2, W2 = H // 2, W // 2
    out = torch.zeros(B_, H, W, COUT, device="cuda").bfloat16()
    pool = torch.zeros(B_, H2, W2, COUT, device="cuda").bfloat16()
    idx = torch.full((B_, H2, W2, COUT), 9, device="cuda", dtype=torch.uint8)
    _cabi.check(L.masr_test_conv3x3_pool_idx(P(x), P(wk), P(bias), P(out), P(pool), P(idx), 0, B_, H, W, CIN, COUT, S()))
    # reference codes from the stored map
    win = out.float()[:, :2 * H2, :2 * W2].reshape(B_, H2, 2, W2, 2, COUT).permute(0, 1, 3, 5, 2, 4).reshape(B_, H2, W2, COUT, 4)
    mx, arg = win.max(dim=-1)
    first = (win == mx.unsqueeze(-1)).float().argmax(dim=-1)            # first position holding the maximum
    code = torch.where(mx > 0, first, torch.full_like(first, 4)).to(torch.uint8)
    assert torch.equal(pool.float(), mx)
    assert torch.equal(idx, code)
    assert (code == 4).float().mean() < 0.5 and all((code == k).any() for k in range(4))      # the case is not degenerate
    # the same launch allowed to drop the map: identical pool and codes
    out2 = torch.full_like(out, 7.0)
    pool2 = torch.zeros_like(pool)
    idx2 = torch.full_like(idx, 9)
    _cabi.check(L.masr_test_conv3x3_pool_idx(P(x), P(wk), P(bias), P(out2), P(pool2), P(idx2), 1, B_, H, W, CIN, COUT, S()))
    assert torch.equal(pool2, pool) and torch.equal(idx2, idx)
    assert bool((out2.float() == 7.0).all()) or torch.equal(out2, out)  # dropped (streaming kernels) or stored (fallback), never half
    # the pool + ReLU backward those codes stand for (what the dgrad / weight-gradient kernels expand while staging: _unpool) against autograd
    gp = torch.randn(B_, H2, W2, COUT, device="cuda", generator=g).bfloat16()
    din = _unpool(idx, gp, H, W)
    pre = out.float().permute(0, 3, 1, 2).clone().requires_grad_(True)  # (the map is already ReLU'd: relu'(x) = [x > 0])
    y = torch.nn.functional.max_pool2d(torch.relu(pre), 2, 2)
    y.backward(gp.float().permute(0, 3, 1, 2))
    ref = pre.grad.permute(0, 2, 3, 1)
    if torch.equal(din.float(), ref):
        return
    # torch's CUDA max_pool2d backward breaks exact ties its own way: compare where the window has a unique maximum
    uniq = (win == mx.unsqueeze(-1)).sum(-1) == 1
    u = uniq.unsqueeze(-1).expand(-1, -1, -1, -1, 4).reshape(B_, H2, W2, COUT, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B_, 2 * H2, 2 * W2, COUT)
    assert torch.equal(din.float()[:, :2 * H2, :2 * W2][u], ref[:, :2 * H2, :2 * W2][u])


@pytest.mark.parametrize("B_,H,W", [(2, 37, 80), (1, 50, 83), (3, 16, 16), (1, 9, 5), (2, 64, 40), (16, 100, 80)])
def test_conv1_forward_on_the_fp32_mfma(L, B_, H, W):
    """The CIN = 1 conv of the VGG front-end (mono_transformer_torch.py:49-50) on v_mfma_f32_16x16x4_f32: fp32 arithmetic, one rounding to bf16.
    Against F.conv2d in fp32 + ReLU rounded to bf16: equal up to the one-ulp flips an fp32 summation order can cause at a rounding boundary (at
    most 0.1 % of the elements, each by one bf16 ulp); the ReLU sign word of every pixel is exactly (out > 0); widths that are not multiples of
    16 and maps smaller than a segment included; a poisoned output buffer shows that every element is written."""
    g = torch.Generator(device="cuda").manual_seed(H * 131 + W)
    x = torch.randn(B_, H, W, device="cuda", generator=g)
    w = torch.randn(64, 1, 3, 3, device="cuda", generator=g) * 0.3
    b = torch.randn(64, device="cuda", generator=g) * 0.2
    out = torch.full((B_, H, W, 64), float("nan"), device="cuda").bfloat16()
    bits = torch.full((B_, H, W), -1, device="cuda", dtype=torch.int64)
    _cabi.check(L.masr_test_conv1_fwd(P(x), P(w.reshape(64, 9).contiguous()), P(b), P(out), P(bits), B_, H, W, S()))
    ref = torch.relu(torch.nn.functional.conv2d(x.unsqueeze(1), w, b, padding=1)).permute(0, 2, 3, 1)
    refq = ref.bfloat16()
    assert bool(torch.isfinite(out.float()).all())
    diff = out != refq
    assert float(diff.float().mean()) <= 1e-3, float(diff.float().mean())
    torch.testing.assert_close(out.float(), ref, rtol=8e-3, atol=1e-6)                     # (bf16: 2^-8 relative)
    want = ((out.float() > 0).to(torch.int64) << torch.arange(64, device="cuda").view(1, 1, 1, 64)).sum(-1)
    assert torch.equal(bits, want)
    # without the sign words (evaluation / the BLSTM front-end): same map
    out2 = torch.zeros_like(out)
    _cabi.check(L.masr_test_conv1_fwd(P(x), P(w.reshape(64, 9).contiguous()), P(b), P(out2), None, B_, H, W, S()))
    assert torch.equal(out2, out)


def _unpool(codes, gp, H, W):
    """MaxPool2d(2, 2) + ReLU backward from the pool codes (window position 0..3 of the first maximum, 4 = nothing passed the ReLU): the map
    [B][H][W][C] the dgrad / weight-gradient kernels expand in their staging -- pure indexing, exact."""
    B_, H2, W2, C = gp.shape
    exp = torch.zeros(B_, H2, W2, C, 4, device=gp.device, dtype=gp.dtype)
    exp.scatter_(-1, codes.long().clamp(max=3).unsqueeze(-1), torch.where(codes < 4, gp, torch.zeros_like(gp)).unsqueeze(-1))
    full = torch.zeros(B_, H, W, C, device=gp.device, dtype=gp.dtype)
    full[:, :2 * H2, :2 * W2] = exp.reshape(B_, H2, W2, C, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B_, 2 * H2, 2 * W2, C)
    return full.contiguous()


@pytest.mark.parametrize("B_,H,W,CIN,COUT", [(2, 10, 9, 64, 64), (1, 33, 21, 64, 128), (2, 40, 20, 128, 128), (1, 21, 48, 128, 128), (1, 17, 80, 64, 64)])
def test_conv3x3_wgrad(L, B_, H, W, CIN, COUT):
    g = torch.Generator(device="cuda").manual_seed(CIN + COUT + H + 1)
    x = torch.randn(B_, H, W, CIN, device="cuda", generator=g).bfloat16()
    dy = torch.randn(B_, H, W, COUT, device="cuda", generator=g).bfloat16()
    n = int(L.masr_test_conv3x3_wgrad_slab_floats(B_, H, W, CIN, COUT))
    slab = torch.zeros(n, device="cuda")
    dw = torch.zeros(COUT, CIN, 3, 3, device="cuda")
    _cabi.check(L.masr_test_conv3x3_wgrad(P(x), P(dy), P(dw), P(slab), n, B_, H, W, CIN, COUT, S()))
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(False)
    wref = torch.zeros(COUT, CIN, 3, 3, device="cuda", requires_grad=True)
    y = torch.nn.functional.conv2d(xr, wref, None, padding=1)
    y.backward(dy.float().permute(0, 3, 1, 2))
    torch.testing.assert_close(dw, wref.grad, rtol=1e-3, atol=2e-2)


@pytest.mark.parametrize("B_,H,W,C", [(2, 10, 9, 64), (2, 40, 20, 128), (1, 21, 47, 128), (1, 33, 80, 64), (3, 16, 32, 64), (2, 33, 41, 128), (1, 50, 83, 64)])
def test_conv3x3_wgrad_from_pooled_gradient(L, B_, H, W, C):
    """ConvWgradArgs::dy_pooled: the weight-gradient kernel of the conv in FRONT of a MaxPool2d(2, 2) (mono_transformer_torch.py:49-60) expands
    pooled gradient + pool codes itself.  Even maps: same bits as staging the expanded map (_unpool; the LDS tile is identical).  Odd H / W: the
    cropped last row / column gets no gradient, so the pooled launch tiles the even part of the map only (41 columns -> five 8-wide tiles of 40
    instead of three 16-wide of 48) -- the same products summed over another partition: equal within fp32 rounding.  The bias gradient (summed in
    another order) within rounding."""
    g = torch.Generator(device="cuda").manual_seed(C + H + W)
    x = torch.randn(B_, H, W, C, device="cuda", generator=g).bfloat16()
    H2, W2 = H // 2, W // 2
    dyp = torch.randn(B_, H2, W2, C, device="cuda", generator=g).bfloat16()
    codes = torch.randint(0, 5, (B_, H2, W2, C), device="cuda", generator=g).to(torch.uint8)           # 4 = nothing passed the ReLU
    dy = _unpool(codes, dyp, H, W)
    n = int(L.masr_test_conv3x3_wgrad_slab_floats(B_, H, W, C, C))
    slab = torch.zeros(n, device="cuda")
    dw0 = torch.zeros(C, C, 3, 3, device="cuda"); dw1 = torch.zeros_like(dw0); db = torch.zeros(C, device="cuda")
    _cabi.check(L.masr_test_conv3x3_wgrad(P(x), P(dy), P(dw0), P(slab), n, B_, H, W, C, C, S()))
    _cabi.check(L.masr_test_conv3x3_wgrad_pooled(P(x), P(dyp), P(codes), P(dw1), P(db), P(slab), n, B_, H, W, C, C, S()))
    if H % 2 == 0 and W % 2 == 0:
        assert torch.equal(dw0, dw1)
    else:
        torch.testing.assert_close(dw1, dw0, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db, dy.float().sum((0, 1, 2)), rtol=1e-4, atol=1e-3)
    assert float(dw1.abs().max()) > 0


@pytest.mark.parametrize("B_,H,W", [(2, 40, 20), (1, 33, 41), (3, 250, 40), (2, 70, 24), (1, 21, 7), (2, 64, 8)])
def test_conv3x3_dgrad_from_pooled_gradient(L, B_, H, W):
    """ConvArgs::in_pooled: the masked 128 <- 128 dgrad behind the second MaxPool2d (mono_transformer_torch.py:57-58) builds its patches from
    the pooled gradient + pool codes (three producer waves expand the 2 x 2 windows while staging).  Same bits as the launch that reads the
    expanded map (_unpool); odd H / W: the cropped last row / column carries no gradient."""
    g = torch.Generator(device="cuda").manual_seed(7 * H + W)
    H2, W2 = H // 2, W // 2
    dyp = torch.randn(B_, H2, W2, 128, device="cuda", generator=g).bfloat16()
    codes = torch.randint(0, 5, (B_, H2, W2, 128), device="cuda", generator=g).to(torch.uint8)
    dy = _unpool(codes, dyp, H, W)
    mask = torch.randn(B_, H, W, 128, device="cuda", generator=g).bfloat16()
    words = _sign_words(mask).to(torch.int32)
    wd = (torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.05).bfloat16()
    wdk = wd.permute(0, 2, 3, 1).reshape(128, 9 * 128).contiguous()
    o1 = torch.full((B_, H, W, 128), 7.0, device="cuda").bfloat16()
    o2 = torch.full((B_, H, W, 128), 5.0, device="cuda").bfloat16()
    _cabi.check(L.masr_test_conv3x3_dgrad_pooled(P(dy), None, None, P(wdk), P(words), P(o1), B_, H, W, S()))
    _cabi.check(L.masr_test_conv3x3_dgrad_pooled(None, P(dyp), P(codes), P(wdk), P(words), P(o2), B_, H, W, S()))
    assert torch.equal(o1, o2)
    conv = torch.nn.functional.conv2d(dy.float().permute(0, 3, 1, 2), wd.float(), None, padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(o2.float(), torch.where(mask.float() > 0, conv, torch.zeros_like(conv)), rtol=1e-2, atol=2e-2)
    assert float(o2.float().abs().max()) > 0
    # a second launch on the same stream re-arms the tile counter correctly (the producers' first tile is static, the rest counted)
    o3 = torch.full_like(o2, 3.0)
    _cabi.check(L.masr_test_conv3x3_dgrad_pooled(None, P(dyp), P(codes), P(wdk), P(words), P(o3), B_, H, W, S()))
    assert torch.equal(o3, o2)


@pytest.mark.parametrize("B_,H,W", [(2, 38, 80), (1, 45, 83), (2, 301, 80), (1, 16, 16), (3, 17, 35), (16, 100, 80)])
def test_conv1_wgrad_fused_from_pooled_gradient(L, B_, H, W):
    """The 64 <- 64 dgrad behind the first MaxPool2d with conv1's weight gradient fused into its epilogue (mono_transformer_torch.py:49-52): fed
    the pooled gradient + codes (four producer waves) it returns the bits of the launch fed the expanded map; both agree with the plain
    formula dW1[c][tap] = sum_p mask(p, c) (dy * W2)(p, c) x(p + tap)."""
    g = torch.Generator(device="cuda").manual_seed(3 * H + W)
    H2, W2 = H // 2, W // 2
    dyp = torch.randn(B_, H2, W2, 64, device="cuda", generator=g).bfloat16()
    codes = torch.randint(0, 5, (B_, H2, W2, 64), device="cuda", generator=g).to(torch.uint8)
    dy = _unpool(codes, dyp, H, W)
    x1 = torch.randn(B_, H, W, device="cuda", generator=g)
    keep = torch.rand(B_, H, W, 64, device="cuda", generator=g) > 0.4
    words = (keep.to(torch.int64) << torch.arange(64, device="cuda").view(1, 1, 1, 64)).sum(-1)          # bit c = channel c passed conv1's ReLU
    wd = (torch.randn(64, 64, 3, 3, device="cuda", generator=g) * 0.05).bfloat16()
    wdk = wd.permute(0, 2, 3, 1).reshape(64, 9 * 64).contiguous()
    n = int(L.masr_test_conv1_wgrad_fused_slab_floats(B_, H, W))
    slab = torch.zeros(n, device="cuda")
    dwa = torch.zeros(64, 9, device="cuda"); dba = torch.zeros(64, device="cuda")
    dwb = torch.full((64, 9), 3.0, device="cuda"); dbb = torch.full((64,), 3.0, device="cuda")
    _cabi.check(L.masr_test_conv1_wgrad_fused(P(dy), None, None, P(wdk), P(words), P(x1), P(slab), n, P(dwa), P(dba), B_, H, W, S()))
    slab.zero_()
    _cabi.check(L.masr_test_conv1_wgrad_fused(None, P(dyp), P(codes), P(wdk), P(words), P(x1), P(slab), n, P(dwb), P(dbb), B_, H, W, S()))
    assert torch.equal(dwa, dwb) and torch.equal(dba, dbb)
    da1 = torch.nn.functional.conv2d(dy.float().permute(0, 3, 1, 2), wd.float(), None, padding=1) * keep.permute(0, 3, 1, 2)      # [B,64,H,W]
    da1 = da1.bfloat16().float()
    xp = torch.nn.functional.pad(x1.bfloat16().float(), (1, 1, 1, 1))
    ref = torch.stack([(da1 * xp[:, None, ky:ky + H, kx:kx + W]).sum((0, 2, 3)) for ky in range(3) for kx in range(3)], dim=1)
    scale = float(ref.abs().max())
    assert float((dwb - ref).abs().max()) < 2e-2 * scale + 1e-2
    torch.testing.assert_close(dbb, da1.sum((0, 2, 3)), rtol=2e-2, atol=2e-2 * float(da1.sum((0, 2, 3)).abs().max()) + 1e-2)


def _attn_ref(q, k, v, klens, causal, dout):
    B_, Tq, H, hd = q.shape
    Tk = k.shape[1]
    q, k, v = (t.float().permute(0, 2, 1, 3).requires_grad_(True) for t in (q, k, v))
    s = q @ k.transpose(-1, -2) / hd ** 0.5
    if causal:
        s = s + torch.triu(torch.full((Tq, Tk), float("-inf"), device=q.device), 1)
    if klens is not None:
        pad = torch.arange(Tk, device=q.device)[None, :] >= klens[:, None]
        s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    o = p @ v
    o.backward(dout.float().permute(0, 2, 1, 3))
    return o.permute(0, 2, 1, 3), q.grad.permute(0, 2, 1, 3), k.grad.permute(0, 2, 1, 3), v.grad.permute(0, 2, 1, 3)


@pytest.mark.parametrize("B_,H,Tq,Tk,hd,causal,masked", [
    (2, 2, 31, 31, 64, 1, False), (3, 4, 250, 250, 64, 0, True), (2, 8, 31, 250, 64, 0, True),
    (2, 4, 12, 16, 16, 0, True), (2, 4, 10, 10, 16, 1, False), (1, 2, 70, 130, 32, 0, True), (2, 2, 100, 100, 32, 1, False),
    # long sequences (3000-frame dev utterances: T' = 750): more than 8 query blocks, i.e. the dK/dV workgroups refresh their block of
    # log-sum-exp / rowsum(dO O) statistics inside the loop; ring depth << number of blocks; causal block skipping at a distance
    (1, 2, 750, 750, 64, 0, True), (1, 2, 700, 700, 64, 1, False), (2, 2, 40, 1100, 64, 0, True), (1, 2, 1100, 48, 64, 0, False), (1, 2, 600, 600, 32, 1, False)])
def test_attention(L, B_, H, Tq, Tk, hd, causal, masked):
    g = torch.Generator(device="cuda").manual_seed(Tq * 7 + Tk + hd)
    q = torch.randn(B_, Tq, H, hd, device="cuda", generator=g).bfloat16()
    k = torch.randn(B_, Tk, H, hd, device="cuda", generator=g).bfloat16()
    v = torch.randn(B_, Tk, H, hd, device="cuda", generator=g).bfloat16()
    dout = torch.randn(B_, Tq, H, hd, device="cuda", generator=g).bfloat16()
    klens = torch.randint(max(1, Tk // 2), Tk + 1, (B_,), device="cuda", generator=g).int() if masked else None
    o = torch.zeros_like(q); dq = torch.zeros_like(q); dk = torch.zeros_like(k); dv = torch.zeros_like(v)
    lse = torch.zeros(B_, H, Tq, device="cuda"); delta = torch.zeros(B_, H, Tq, device="cuda")
    _cabi.check(L.masr_test_attention(P(q), P(k), P(v), P(dout), P(o), P(dq), P(dk), P(dv), P(lse), P(delta),
                                      P(klens) if masked else None, B_, H, Tq, Tk, hd, causal, S()))
    ro, rdq, rdk, rdv = _attn_ref(q, k, v, klens.long() if masked else None, causal, dout)
    torch.testing.assert_close(o.float(), ro, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dq.float(), rdq, rtol=3e-2, atol=4e-2)
    torch.testing.assert_close(dk.float(), rdk, rtol=3e-2, atol=4e-2)
    torch.testing.assert_close(dv.float(), rdv, rtol=3e-2, atol=4e-2)
    if masked:                                   # masked keys receive exactly zero gradient
        for b in range(B_):
            assert torch.all(dk[b, int(klens[b]):] == 0) and torch.all(dv[b, int(klens[b]):] == 0)


@pytest.mark.parametrize("rows,E", [(592, 512), (4000, 512), (2051, 512), (37, 256), (70, 384), (5, 64), (2048, 1024)])
def test_layernorm_fwd_bwd(L, rows, E):
    """nn.LayerNorm (eps 1e-5) forward + backward on one [rows][E] matrix (mono_transformer_torch.py:74-98 via nn.Transformer*Layer):
    both row partitions of the backward (4 / 16 rows per workgroup), ragged last workgroups, the 16-byte (E % 256 == 0) and the
    4-byte lane layouts."""
    g = torch.Generator(device="cuda").manual_seed(rows + E)
    x = (torch.randn(rows, E, device="cuda", generator=g) * 1.7 + 0.3).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(E, device="cuda", generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(E, device="cuda", generator=g)).requires_grad_(True)
    dy = torch.randn(rows, E, device="cuda", generator=g)
    ref = torch.nn.functional.layer_norm(x, (E,), gamma, beta, 1e-5)
    ref.backward(dy)
    y = torch.zeros(rows, E, device="cuda"); y16 = torch.zeros(rows, E, device="cuda").bfloat16()
    mean = torch.zeros(rows, device="cuda"); rstd = torch.zeros(rows, device="cuda")
    dx = torch.full((rows + 1, E), 7.0, device="cuda"); dx16 = torch.zeros(rows, E, device="cuda").bfloat16()
    dgm = torch.zeros(E, device="cuda"); dbt = torch.zeros(E, device="cuda")
    slab = torch.zeros(int(L.masr_test_layernorm_slab_floats(rows, E)), device="cuda")
    _cabi.check(L.masr_test_layernorm(P(x.detach()), P(gamma.detach()), P(beta.detach()), P(dy), P(y), P(y16), P(mean), P(rstd), P(dx), P(dx16),
                                      P(dgm), P(dbt), P(slab), rows, E, 0.0, 0, 0, S()))
    torch.testing.assert_close(y, ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(y16.float(), ref.detach().bfloat16().float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(mean, x.detach().mean(1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dx[:rows], x.grad, rtol=1e-4, atol=2e-5)
    assert torch.all(dx[rows] == 7.0)                                    # nothing written behind the last row
    torch.testing.assert_close(dx16.float(), x.grad.bfloat16().float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(dgm, gamma.grad, rtol=1e-4, atol=1e-4 * rows ** 0.5)
    torch.testing.assert_close(dbt, beta.grad, rtol=1e-4, atol=1e-4 * rows ** 0.5)
    # the bf16 copy of dx carries the dropout of the sublayer output that feeds this LayerNorm's residual sum (keep-scale of element row * E + col)
    dxd = torch.full((rows + 1, E), 7.0, device="cuda"); dx16d = torch.zeros(rows, E, device="cuda").bfloat16()
    _cabi.check(L.masr_test_layernorm(P(x.detach()), P(gamma.detach()), P(beta.detach()), P(dy), P(y), P(y16), P(mean), P(rstd), P(dxd), P(dx16d),
                                      P(dgm), P(dbt), P(slab), rows, E, 0.25, 99, 4, S()))
    keep = torch.empty(rows * E, device="cuda")
    _cabi.check(L.masr_test_dropout_mask(99, 4, rows * E, 0.25, P(keep), S()))
    assert torch.equal(dxd[:rows], dx[:rows]) and torch.equal(dx16d, (dx[:rows] * keep.view(rows, E)).bfloat16())


@pytest.mark.parametrize("rows,E,K,split,drop", [(155, 512, 2048, 4, 0.0), (656, 512, 2048, 4, 0.1), (61, 512, 1536, 3, 0.0), (1024, 512, 1024, 2, 0.25),
                                                  (3, 256, 4096, 8, 0.1)])
def test_ksplit_gemm_summed_by_the_layernorm(L, rows, E, K, split, drop):
    """The decoder's few-row GEMMs with a long reduction run k-split, and the LayerNorm behind them sums the fp32 partial products
    (engine.hip ffn_fwd / ln_fwd, ffn_bwd / ln_bwd, attn_block_bwd; rowops.hip LnSumArgs).  Forward: sum + bias, dropout at the GEMM's element
    index, + residual, LayerNorm -- against mk_gemm's own epilogue followed by the plain LayerNorm (same numbers up to fp32 summation order)
    and against torch.  Backward: dy = sum + residual gradient."""
    g = torch.Generator(device="cuda").manual_seed(rows + K)
    A = (torch.randn(rows, K, device="cuda", generator=g) * 0.5).bfloat16()
    B = (torch.randn(E, K, device="cuda", generator=g) * 0.05).bfloat16()
    bias = torch.randn(E + 1, device="cuda", generator=g)[1:]                # (4-byte aligned only, like a bias inside the flat parameter buffer)
    res = torch.randn(rows, E, device="cuda", generator=g)
    gamma = 1 + 0.2 * torch.randn(E, device="cuda", generator=g); beta = 0.1 * torch.randn(E, device="cuda", generator=g)
    part = torch.zeros(split, rows, E, device="cuda")
    z = lambda *s: torch.zeros(*s, device="cuda")
    ssum, y, y16, mean, rstd = z(rows + 1, E), z(rows + 1, E), z(rows, E).bfloat16(), z(rows), z(rows)
    ssum[rows] = 7.0; y[rows] = 7.0
    _cabi.check(L.masr_test_ksplit_ln(P(A), P(B), rows, E, K, split, P(bias), P(res), drop, 1, 2, P(part), P(gamma), P(beta), P(ssum), P(y), P(y16),
                                      P(mean), P(rstd), None, None, None, None, S()))
    # the partial products are what they say
    ref_part = torch.stack([A[:, i * K // split:(i + 1) * K // split].float() @ B[:, i * K // split:(i + 1) * K // split].float().t() for i in range(split)])
    torch.testing.assert_close(part, ref_part, rtol=1e-4, atol=1e-4)
    # whole-reduction GEMM with the epilogue, then the plain LayerNorm
    C = z(rows, E)
    _cabi.check(L.masr_test_gemm_epi(P(A), K, P(B), K, rows, E, K, P(bias), 0, drop, P(res), None, P(C), None, S()))
    torch.testing.assert_close(ssum[:rows], C, rtol=2e-6, atol=2e-5)
    assert torch.all(ssum[rows] == 7.0) and torch.all(y[rows] == 7.0)
    keep = torch.ones(rows * E, device="cuda")
    if drop:
        _cabi.check(L.masr_test_dropout_mask(1, 2, rows * E, drop, P(keep), S()))
    ref_sum = (A.float() @ B.float().t() + bias) * keep.view(rows, E) + res
    torch.testing.assert_close(ssum[:rows], ref_sum, rtol=1e-5, atol=1e-4)
    ref_y = torch.nn.functional.layer_norm(ssum[:rows], (E,), gamma, beta, 1e-5)
    torch.testing.assert_close(y[:rows], ref_y, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(y16.float(), ref_y.bfloat16().float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(mean, ssum[:rows].mean(1), rtol=1e-5, atol=1e-5)
    # backward: dy = A.B^T + res (no bias, no dropout on the sum), x = an arbitrary matrix with its own statistics
    x = torch.randn(rows, E, device="cuda", generator=g) * 1.3 + 0.2
    y2, y2_16, mean2, rstd2 = z(rows, E), z(rows, E).bfloat16(), z(rows), z(rows)
    dgm, dbt, dx_ref, dx16_ref = z(E), z(E), z(rows, E), z(rows, E).bfloat16()
    slab0 = z(int(L.masr_test_layernorm_slab_floats(rows, E)))
    _cabi.check(L.masr_test_gemm_epi(P(A), K, P(B), K, rows, E, K, None, 0, 0.0, P(res), None, P(C), None, S()))
    _cabi.check(L.masr_test_layernorm(P(x), P(gamma), P(beta), P(C), P(y2), P(y2_16), P(mean2), P(rstd2), P(dx_ref), P(dx16_ref), P(dgm), P(dbt), P(slab0),
                                      rows, E, drop, 1, 2, S()))
    nb = (rows + 3) // 4
    dx, dx16, slab = z(rows + 1, E), z(rows, E).bfloat16(), z(nb, 2, E)
    dx[rows] = 7.0
    _cabi.check(L.masr_test_ksplit_ln(P(A), P(B), rows, E, K, split, None, P(res), drop, 1, 2, P(part), P(gamma), None, None, None, None,
                                      P(mean2), P(rstd2), P(x), P(dx), P(dx16), P(slab), S()))
    scale = float(dx_ref.abs().max())
    torch.testing.assert_close(dx[:rows], dx_ref, rtol=1e-5, atol=2e-6 * scale)
    assert torch.all(dx[rows] == 7.0)
    assert float((dx16.float() - dx16_ref.float()).abs().max()) <= 2 ** -7 * scale          # (one bf16 step where the fp32 value sits on a tie)
    torch.testing.assert_close(slab[:, 0].sum(0), dgm, rtol=1e-4, atol=1e-5 * scale * rows ** 0.5)
    torch.testing.assert_close(slab[:, 1].sum(0), dbt, rtol=1e-4, atol=1e-5 * scale * rows ** 0.5)


@pytest.mark.parametrize("B_,H,T,hd,mag", [(4, 4, 3, 16, 3e4), (2, 8, 250, 64, 3e3), (2, 4, 37, 64, 1e4), (1, 2, 130, 32, 1e3)])
def test_attention_huge_scores(L, B_, H, T, hd, mag):
    """A diverging run (train.py with an aggressive Noam schedule: |q|, |k| ~ 3e4, scores ~ 1e9) must still get torch's one-hot softmax
    and its vanishing gradients: the exponent is formed as a DIFFERENCE (s - m, s scale - lse) before any scaling -- with s sc2 - fl(m sc2)
    the rounding of the product alone is 2^16 at such scores (seen as gradients of 1e13 and a NaN run)."""
    g = torch.Generator(device="cuda").manual_seed(B_ + T)
    mk = lambda m: (torch.randn(B_, T, H, hd, device="cuda", generator=g) * m).bfloat16()
    q, k, v, dout = mk(mag), mk(mag), mk(mag), mk(1e-3)
    klens = torch.full((B_,), T, device="cuda", dtype=torch.int32)
    o = torch.zeros_like(q); dq = torch.zeros_like(q); dk = torch.zeros_like(k); dv = torch.zeros_like(v)
    lse = torch.zeros(B_, H, T, device="cuda"); delta = torch.zeros(B_, H, T, device="cuda")
    _cabi.check(L.masr_test_attention(P(q), P(k), P(v), P(dout), P(o), P(dq), P(dk), P(dv), P(lse), P(delta), P(klens), B_, H, T, T, hd, 0, S()))
    ro, rdq, rdk, rdv = _attn_ref(q, k, v, klens.long(), 0, dout)
    for t in (o, dq, dk, dv, lse):
        assert torch.isfinite(t.float()).all()
    torch.testing.assert_close(o.float(), ro, rtol=2e-2, atol=2e-2 * mag)
    # with a one-hot softmax dS vanishes: what is left in dq / dk is p (dP - delta) with delta taken from the bf16 output, i.e. bounded by
    # bf16 rounding x |dO| |v| |k|; the same bound holds for torch's own result
    bound = 2.0 ** -7 * float(dout.float().abs().max()) * mag * mag * hd ** 0.5
    assert float((dq.float() - rdq).abs().max()) <= bound and float((dk.float() - rdk).abs().max()) <= bound
    # dV = P^T dO: rows whose two best keys lie within a few units of each other are not one-hot, and at scores of 1e6..1e9 the fp32
    # accumulation order of q.k alone moves the exponent by an ulp of the score (0.06..64): those P differ between ANY two implementations.
    # Bounded by |dO| everywhere, equal where the softmax is one-hot (the bulk)
    dvd = (dv.float() - rdv).abs()
    assert float(dvd.max()) <= 1.01 * float(dout.float().abs().max()) * 2 and float((dvd > 3e-2 * float(dout.float().abs().max())).float().mean()) < 0.1


@pytest.mark.parametrize("B_,H,Tq,Tk,hd,causal,masked", [
    (2, 8, 250, 250, 64, 0, True), (3, 8, 37, 250, 64, 0, True), (2, 8, 37, 37, 64, 1, False), (2, 4, 150, 40, 64, 0, True), (2, 4, 130, 130, 64, 1, False),
    (2, 4, 70, 90, 16, 0, True), (2, 2, 33, 33, 32, 1, False), (1, 2, 600, 640, 64, 0, True)])
def test_attention_dropout_forward_backward(L, B_, H, Tq, Tk, hd, causal, masked):
    """Dropout on the attention probabilities (nn.MultiheadAttention(dropout=p), mono_transformer_torch.py:74-98), forward AND backward, in every
    kernel variant (head dim 64: 8-wave / 4-wave ring forward, 4 x 2 / 4 x 1 row-tile backward bodies on either side; head dim 16 / 32:
    register-staged): the backward regenerates the forward's masks from (seed, site, ((b H + h) Tq + i) Tk + j).  Reference: torch autograd
    through softmax -> (mask from masr_test_dropout_mask) -> P V on the same bf16 operands."""
    p, seed, site = 0.2, 4321, 7
    g = torch.Generator(device="cuda").manual_seed(Tq * 3 + Tk + hd)
    mk = lambda T: torch.randn(B_, T, H, hd, device="cuda", generator=g).bfloat16()
    q, k, v, dout = mk(Tq), mk(Tk), mk(Tk), mk(Tq)
    klens = torch.randint(max(1, Tk // 3), Tk + 1, (B_,), device="cuda", generator=g).int() if masked else None
    o = torch.zeros_like(q); dq = torch.zeros_like(q); dk = torch.zeros_like(k); dv = torch.zeros_like(v)
    lse = torch.zeros(B_, H, Tq, device="cuda")
    _cabi.check(L.masr_test_attention_dropout_bwd(P(q), P(k), P(v), P(dout), P(o), P(dq), P(dk), P(dv), P(lse), P(klens) if masked else None,
                                                  B_, H, Tq, Tk, hd, causal, p, seed, site, S()))
    keep = torch.empty(B_ * H * Tq * Tk, device="cuda")
    _cabi.check(L.masr_test_dropout_mask(seed, site, keep.numel(), p, P(keep), S()))
    keep = keep.view(B_, H, Tq, Tk)
    qf, kf, vf = [t.float().requires_grad_(True) for t in (q, k, v)]
    sc = torch.einsum("bqhd,bkhd->bhqk", qf, kf) / hd ** 0.5
    msk = torch.zeros(B_, 1, Tq, Tk, dtype=torch.bool, device="cuda")
    if masked:
        for b in range(B_):
            msk[b, :, :, int(klens[b]):] = True
    if causal:
        msk = msk | torch.triu(torch.ones(Tq, Tk, dtype=torch.bool, device="cuda"), 1)
    pr = torch.softmax(sc.masked_fill(msk, float("-inf")), -1) * keep
    ro = torch.einsum("bhqk,bkhd->bqhd", pr, vf)
    ro.backward(dout.float())
    rms = lambda t: float(t.pow(2).mean().sqrt())
    for name, a, r in (("o", o, ro.detach()), ("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        err = rms(a.float() - r) / rms(r)
        assert err < 0.012, f"{name}: rel rms error {err:.4f}"           # (bf16 rounding of P, dS, O and the outputs: 0.002-0.006; a wrong mask: > 0.3)


def test_dropout_keep_rate_and_scale_per_site():
    """nn.Dropout semantics at every site of the engine (PE dropout 1 / 100, attention probabilities, attention out-proj,
    FFN inner, FFN out: sites 1.. and 100.. in csrc/engine.hip): an element is kept with probability 1 - p and scaled by
    1/(1-p); masks of different sites, seeds and elements are independent.  The mask is a stateless hash of (seed, site,
    index): checked through the hash itself, through the GEMM epilogue and through the attention-probability path."""
    L = _cabi.lib()
    P = lambda t: C.c_void_p(t.data_ptr())
    S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = 1 << 20
    masks = {}
    for p in (0.1, 0.2):
        for site in (1, 2, 3, 4, 5, 100, 101, 106, 125):
            out = torch.empty(n, device="cuda")
            _cabi.check(L.masr_test_dropout_mask(1234, site, n, p, P(out), S()))
            vals = torch.unique(out)
            assert len(vals) == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1.0 / (1.0 - p)) < 1e-6
            keep = float((out > 0).float().mean())
            assert abs(keep - (1.0 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-4, (site, p, keep)          # 4 sigma
            assert abs(float(out.mean()) - 1.0) < 2e-3                                                  # mean-preserving
            masks[(p, site)] = out > 0
    a, b = masks[(0.2, 1)].float(), masks[(0.2, 2)].float()
    corr = float(((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std()))
    assert abs(corr) < 5e-3, "masks of two sites are correlated"
    # one hash word decides an element PAIR (low / high 16 bits, csrc/common.h): neighbours inside a pair and across pairs stay independent
    for lag in (1, 2):
        x, y = a[:-lag:2] if lag == 1 else a[:-lag], a[lag::2] if lag == 1 else a[lag:]
        x, y = x[:min(len(x), len(y))], y[:min(len(x), len(y))]
        assert abs(float(((x - x.mean()) * (y - y.mean())).mean() / (x.std() * y.std()))) < 5e-3, f"neighbouring elements (lag {lag}) are correlated"
    o2 = torch.empty(n, device="cuda")
    _cabi.check(L.masr_test_dropout_mask(1235, 1, n, 0.2, P(o2), S()))
    c = (o2 > 0).float()
    assert abs(float(((a - a.mean()) * (c - c.mean())).mean() / (a.std() * c.std()))) < 5e-3, "masks of two seeds are correlated"
    # GEMM epilogue site: C = 8 everywhere before dropout
    M, N, K, p = 512, 384, 64, 0.1
    A = torch.ones(M, K, device="cuda", dtype=torch.bfloat16)
    B = torch.full((N, K), 0.125, device="cuda", dtype=torch.bfloat16)
    Cm = torch.empty(M, N, device="cuda")
    _cabi.check(L.masr_test_gemm_dropout(P(A), K, P(B), K, M, N, K, p, 77, 3, P(Cm), N, S()))
    vals = torch.unique(Cm)
    assert len(vals) == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 8.0 / (1 - p)) < 1e-4
    assert abs(float((Cm > 0).float().mean()) - (1 - p)) < 4e-3
    ref = torch.empty(M * N, device="cuda")
    _cabi.check(L.masr_test_dropout_mask(77, 3, M * N, p, P(ref), S()))
    assert torch.equal(Cm.reshape(-1) > 0, ref > 0), "the epilogue indexes its mask by m*N+n (the backward regenerates it the same way)"
    # attention probabilities: q = k = 0 -> uniform probabilities, v = 1 -> o = (kept keys / Tk) / (1 - p): mean 1, binomial spread
    Bq, H, Tq, Tk, hd, p = 4, 4, 128, 256, 64, 0.2
    q = torch.zeros(Bq * Tq, H * hd, device="cuda", dtype=torch.bfloat16)
    k = torch.zeros(Bq * Tk, H * hd, device="cuda", dtype=torch.bfloat16)
    v = torch.ones(Bq * Tk, H * hd, device="cuda", dtype=torch.bfloat16)
    o = torch.empty(Bq * Tq, H * hd, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(Bq * H * Tq, device="cuda")
    _cabi.check(L.masr_test_attention_dropout(P(q), P(k), P(v), P(o), P(lse), Bq, H, Tq, Tk, hd, p, 5, 9, S()))
    of = o.float()
    assert abs(float(of.mean()) - 1.0) < 5e-3
    sd_expect = (p / (1 - p) / Tk) ** 0.5
    row = of.view(Bq * Tq, H, hd)[:, :, 0]
    assert abs(float(row.std()) - sd_expect) < 0.15 * sd_expect, (float(row.std()), sd_expect)


@pytest.mark.parametrize("M,N,K,flavour", [(4000, 2048, 512, "bias_relu"), (2100, 1152, 192, "bias"), (4000, 1024, 512, "plain"),
                                            (3000, 2048, 2048, "mask"), (4000, 4096, 512, "bias")])
def test_gemm_wide_outputs(L, M, N, K, flavour):
    """the encoder-row NT GEMMs at their full size through mk_gemm's epilogues: FFN first layer (bias + ReLU, bf16 out), K|V projection of the
    memory (bias), the plain and the ReLU-masked dgrads; ragged row counts (2100, 3000), a width that is not a multiple of 256 (1152)."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    B = (torch.randn(N, K, device="cuda", generator=g) * 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g) if flavour.startswith("bias") else None
    mask = (torch.randn(M, N, device="cuda", generator=g) > 0).bfloat16() if flavour == "mask" else None
    C16 = torch.full((M + 1, N), 3.0, device="cuda").bfloat16()
    _cabi.check(L.masr_test_gemm_epi(P(A), K, P(B), K, M, N, K, P(bias) if bias is not None else None, 1 if flavour == "bias_relu" else 0, 0.0,
                                     None, P(mask) if mask is not None else None, None, P(C16), S()))
    ref = A.float() @ B.float().t()
    if bias is not None:
        ref = ref + bias
    if flavour == "bias_relu":
        ref = ref.clamp_min(0)
    if mask is not None:
        ref = ref * mask.float()
    torch.testing.assert_close(C16[:M].float(), ref, rtol=1e-2, atol=2e-2 * (K ** 0.5) / 8)
    assert torch.all(C16[M].float() == 3.0)


@pytest.mark.parametrize("N,K,ldt,src", [(512, 512, 512, 4), (2048, 512, 2048, 5), (512, 2048, 512, 6), (367, 512, 384, 7), (1536, 512, 1536, 1001),
                                         (100, 200, 104, 9), (70, 66, 72, 11), (64, 64, 64, 8), (592, 130, 592, 13)])
def test_linear_operand_shadows(L, N, K, ldt, src):
    """masr_refresh's shadow pass on one Linear weight (the operands of `F.linear` and of its input gradient, mono_transformer_torch.py:74-98): k16
    = the weight rounded to bf16, t16 = its transpose with row pitch ldt; exact.  Every dword misalignment of the tensor inside the flat buffer,
    ragged row counts (367 = odim), widths that are no multiple of 64 or of 4 (the generic tiles), padded transposed rows whose pads stay
    untouched; what surrounds the tensor in the buffer is NaN (a tile that stored a neighbour's element would show)."""
    g = torch.Generator(device="cuda").manual_seed(N + K + src)
    P_ = torch.full((src + N * K + 64,), float("nan"), device="cuda")
    W = torch.randn(N, K, device="cuda", generator=g)
    P_[src:src + N * K] = W.reshape(-1)
    k16 = torch.full((N, K), 7.0, device="cuda").bfloat16()
    t16 = torch.full((K, ldt), 7.0, device="cuda").bfloat16()
    _cabi.check(L.masr_test_linear_shadows(P(P_), src, N, K, ldt, P(k16), P(t16), S()))
    want = W.bfloat16()
    assert torch.equal(k16, want)
    assert torch.equal(t16[:, :N], want.t())
    assert bool((t16[:, N:].float() == 7.0).all())
