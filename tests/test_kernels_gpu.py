"""Per-kernel parity: every C-ABI entry point (called through ctypes on a real MI355X) against
the CPU oracle / fp64 torch-CPU restatements of the same TF-1.3 semantics."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    assert torch.cuda.is_available(), 'GPU tests need a real MI355X (run through gpurun)'
    from demo2program_amd import build, kernels
    build.build_library()
    return kernels


def dev(t, dtype=torch.float32):
    return t.to(dtype).cuda().contiguous()


def close(a, b, atol=1e-4, rtol=1e-4):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    tol = atol + rtol * b.abs().max().item() if b.numel() else atol
    assert err <= tol, 'max abs err %.3e > tol %.3e' % (err, tol)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


# ------------------------------------------------------------------ batched batch norm
@pytest.mark.parametrize('R,C,G,inner', [(3200, 512, 1, 1), (640, 48, 10, 4), (70, 6, 2, 5)])
def test_bn_two_problems_in_one_launch_equal_two_calls(K, R, C, G, inner):
    """d2p_bn_group_{fwd,bwd}_batched: two independent BN problems (own inputs, parameters, moving statistics --
    the two relation networks) in one set of launches give bit for bit what two separate calls give."""
    nb, ps = 2, 3 * C + 8                     # parameters of problem b at + b*ps floats (as in a flat buffer)
    x = dev(rnd(nb, R, C, seed=41, scale=3.0))
    dy = dev(rnd(nb, R, C, seed=42))
    par = dev(rnd(nb, ps, seed=43))           # per problem: gamma | beta | bias-grad slot | padding
    gamma, beta = par[:, :C], par[:, C:2 * C]
    # separate calls
    ys, means, rstds, mms, mvs, dxs, dgs, dbs, dbias = [], [], [], [], [], [], [], [], []
    for b in range(nb):
        mm, mv = torch.zeros(C, device='cuda') + 0.25 * b, torch.ones(C, device='cuda')
        y, mean, rstd, _ = K.bn_fwd(x[b], gamma[b].contiguous(), beta[b].contiguous(), G, inner, moving=(mm, mv))
        dg, db, dbi = (torch.empty(C, device='cuda') for _ in range(3))
        dx = K.bn_bwd(x[b], dy[b], gamma[b].contiguous(), mean, rstd, G, inner, True, dg, db, dbias=dbi)
        for lst, v in zip((ys, means, rstds, mms, mvs, dxs, dgs, dbs, dbias), (y, mean, rstd, mm, mv, dx, dg, db, dbi)):
            lst.append(v.clone())
    # one batched call each way
    y2 = torch.empty(nb, R, C, device='cuda')
    mean2, rstd2 = torch.empty(nb, G, C, device='cuda'), torch.empty(nb, G, C, device='cuda')
    mm2 = torch.stack([torch.zeros(C, device='cuda') + 0.25 * b for b in range(nb)])
    mv2 = torch.ones(nb, C, device='cuda')
    K.bn_fwd_batched(x, par[0, :C], par[0, C:2 * C], ps, G, inner, y2, mean2, rstd2, moving=(mm2, mv2), mstride=C)
    grads = torch.zeros(nb, ps, device='cuda')
    dx2 = torch.empty(nb, R, C, device='cuda')
    K.bn_bwd_batched(x, dy, par[0, :C], ps, mean2, rstd2, G, inner, True, grads[0, :C], grads[0, C:2 * C], dx2,
                     dbias=grads[0, 2 * C:3 * C])
    for b in range(nb):
        assert torch.equal(y2[b], ys[b]) and torch.equal(mean2[b], means[b]) and torch.equal(rstd2[b], rstds[b])
        assert torch.equal(mm2[b], mms[b]) and torch.equal(mv2[b], mvs[b])
        assert torch.equal(dx2[b], dxs[b])
        assert torch.equal(grads[b, :C], dgs[b]) and torch.equal(grads[b, C:2 * C], dbs[b])
        assert torch.equal(grads[b, 2 * C:3 * C], dbias[b])
        assert grads[b, 3 * C:].abs().max().item() == 0


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('M,N,K_', [(64, 64, 16), (70, 50, 37), (320, 2048, 512), (6400, 6, 512),
                                    (33, 512, 5), (1, 1, 1), (130, 260, 1030), (144, 16, 20480), (36, 16, 40961),
                                    (288, 48, 12000), (432, 48, 8200),
                                    (1024, 1024, 64), (6400, 5, 512), (1600, 50, 512), (37, 64, 260), (6400, 512, 6),
                                    (333, 512, 16), (512, 6, 6400), (5, 512, 6400), (512, 50, 1600), (3, 700, 2048),
                                    # the small-problem path (32x32 tiles, K split across the waves)
                                    (320, 512, 512), (512, 512, 320), (70, 45, 300), (31, 33, 129), (100, 100, 1100),
                                    (1568, 50, 512), (32, 32, 128), (95, 200, 4099)])
def test_gemm_nn_nt_tn(K, M, N, K_):
    A = rnd(M, K_, seed=1)
    B = rnd(K_, N, seed=2)
    ref = A @ B
    tol = dict(atol=2e-6 * K_ + 1e-5, rtol=1e-5)
    close(K.matmul_nn(dev(A), dev(B)), ref, **tol)
    close(K.matmul_nt(dev(A), dev(B.t().contiguous())), ref, **tol)
    close(K.matmul_tn(dev(A.t().contiguous()), dev(B)), ref, **tol)


def test_gemm_epilogue_bias_lrelu_accumulate_and_strides(K):
    M, N, K_ = 100, 72, 48
    A, B, bias, C0 = rnd(M, K_, seed=3), rnd(K_, N, seed=4), rnd(N, seed=5), rnd(M, N, seed=6)
    out = K.matmul_nn(dev(A), dev(B), bias=dev(bias), act=1)
    close(out, oracle.lrelu(A @ B + bias), atol=1e-4)
    c = dev(C0)
    K.matmul_nn(dev(A), dev(B), out=c, accumulate=True)
    close(c, A @ B + C0, atol=1e-4)
    # strided C and A (LSTM layout: rows with stride T*4U, offset t*4U)
    big = torch.zeros(M, 3 * N, device='cuda')
    view = big[:, N:2 * N]
    K.gemm_raw('nn', M, N, K_, dev(A), K_, dev(B), N, view.data_ptr(), 3 * N)
    close(big[:, N:2 * N], A @ B, atol=1e-4)
    assert big[:, :N].abs().max().item() == 0 and big[:, 2 * N:].abs().max().item() == 0


def test_gemm_small_problem_path_epilogues_and_switch(K):
    """The 32x32 wave-split tile applies the same epilogue (bias, leaky relu, accumulate) and can be
    switched off (d2p_gemm_set_option bit 1): same values either way, within summation order."""
    from demo2program_amd.lib import call
    M, N, K_ = 320, 512, 512
    A, B, bias, C0 = rnd(M, K_, seed=13), rnd(K_, N, seed=14), rnd(N, seed=15), rnd(M, N, seed=16)
    outs = []
    for off in (0, 2):
        call.d2p_gemm_set_option(off)
        try:
            o1 = K.matmul_nn(dev(A), dev(B), bias=dev(bias), act=1)
            c = dev(C0)
            K.matmul_nt(dev(A), dev(B.t().contiguous()), out=c, accumulate=True)
            outs.append((o1.clone(), c.clone()))
        finally:
            call.d2p_gemm_set_option(0)
    close(outs[0][0], oracle.lrelu(A @ B + bias), atol=2e-4)
    close(outs[0][1], A @ B + C0, atol=2e-4)
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-4
    assert (outs[0][1] - outs[1][1]).abs().max().item() < 1e-4


@pytest.mark.parametrize('M,N,K_', [(320, 512, 512), (3200, 512, 512), (512, 512, 320), (70, 45, 300), (33, 64, 16)])
def test_gemm_strided_batch(K, M, N, K_):
    """d2p_gemm_f32_batched: 2 x 2 problems in one launch (shared A over the second index, as the
    relation network's P / Q projections), every kind, with bias + leaky relu and with accumulate."""
    nb1, nb0 = 2, 2
    A = rnd(nb1, M, K_, seed=21)
    B = rnd(nb1, nb0, K_, N, seed=22)
    bias = rnd(nb1, N, seed=23)
    C0 = rnd(nb1, nb0, M, N, seed=24)
    ref = torch.einsum('imk,ijkn->ijmn', A, B)
    tol = dict(atol=2e-6 * K_ + 1e-5, rtol=1e-5)
    dA, dB = dev(A), dev(B)
    out = torch.zeros(nb1, nb0, M, N, device='cuda')
    K.gemm_batched('nn', nb1, nb0, M, N, K_, dA, K_, (M * K_, 0), dB, N, (nb0 * K_ * N, K_ * N), out, N,
                   (nb0 * M * N, M * N), bias=dev(bias), sbias=(N, 0), act=1)
    close(out, oracle.lrelu(ref + bias[:, None, None, :]), **tol)
    Bt = dev(B.transpose(2, 3).contiguous())                       # [nb1, nb0, N, K]
    acc = dev(C0)
    K.gemm_batched('nt', nb1, nb0, M, N, K_, dA, K_, (M * K_, 0), Bt, K_, (nb0 * N * K_, N * K_), acc, N,
                   (nb0 * M * N, M * N), accumulate=True)
    close(acc, ref + C0, **tol)
    At = dev(A.transpose(1, 2).contiguous())                       # [nb1, K, M]
    out.zero_()
    K.gemm_batched('tn', nb1, nb0, M, N, K_, At, M, (K_ * M, 0), dB, N, (nb0 * K_ * N, K_ * N), out, N,
                   (nb0 * M * N, M * N))
    close(out, ref, **tol)
    # batch of one == the plain entry point
    one = torch.zeros(M, N, device='cuda')
    K.gemm_batched('nn', 1, 1, M, N, K_, dA, K_, (0, 0), dB, N, (0, 0), one, N, (0, 0))
    close(one, ref[0, 0], **tol)


@pytest.mark.parametrize('tile', [8, 9, 10, 11, 12])
@pytest.mark.parametrize('M,N,K_,splits', [(640, 256, 512, 1), (1000, 320, 256, 1), (512, 2048, 1600, 4),
                                            (130, 196, 96, 1), (6400, 512, 2048, 2)])
def test_gemm_lds_dma_pipeline_equals_staged_kernel(K, tile, M, N, K_, splits):
    """gemm_dma_kernel (forced through d2p_gemm_force_plan, tiles 8..12: persistent workgroups, operands DMA'd
    into an LDS ring by a producer wave) consumes K in the same order as the staged kernel: bit-identical
    results for the same split of K, all three operand layouts, bias + leaky relu and accumulate epilogues,
    ragged M / N edges."""
    from demo2program_amd.lib import call
    A, B, bias, C0 = rnd(M, K_, seed=31), rnd(K_, N, seed=32), rnd(N, seed=33), rnd(M, N, seed=34)
    dA, dB, dAt, dBt = dev(A), dev(B), dev(A.t().contiguous()), dev(B.t().contiguous())
    K.SCRATCH.reserve(8 * M * N * 4)

    def run():
        outs = [K.matmul_nn(dA, dB, bias=dev(bias), act=1).clone(), K.matmul_nt(dA, dBt).clone(),
                K.matmul_tn(dAt, dB).clone()]
        c = dev(C0)
        K.matmul_nn(dA, dB, out=c, accumulate=True)
        return outs + [c.clone()]
    try:
        call.d2p_gemm_force_plan(0 if tile in (8, 9) else (4 if tile == 10 else 1), splits)
        staged = run()
        call.d2p_gemm_force_plan(tile, splits)
        dma = run()
    finally:
        call.d2p_gemm_force_plan(-1, 0)
    for a, b in zip(staged, dma):
        assert torch.equal(a, b)
    close(dma[1], A @ B, atol=2e-6 * K_ + 1e-5, rtol=1e-5)
    close(dma[0], oracle.lrelu(A @ B + bias), atol=2e-6 * K_ + 1e-5, rtol=1e-5)


@pytest.mark.parametrize('R,N,K_,frac', [(6400, 2048, 512, 0.7), (6400, 512, 2048, 0.7), (100, 72, 48, 0.5), (640, 64, 37, 0.9)])
def test_gemm_over_a_list_of_rows(K, R, N, K_, frac):
    """d2p_gemm_f32_rows: the product for the listed rows only (rows of a padded batch past their sequence's length
    are skipped), result rows scattered to their places, the others untouched; same values as the dense product."""
    g = torch.Generator().manual_seed(3)
    A, B, bias = rnd(R, K_, seed=71), rnd(K_, N, seed=72), rnd(N, seed=73)
    keep = torch.rand(R, generator=g) < frac
    rows = torch.nonzero(keep).flatten().to(torch.int32)
    n = rows.numel()
    dA, dB, dBt, drows = dev(A), dev(B), dev(B.t().contiguous()), rows.cuda()
    tol = dict(atol=2e-6 * K_ + 1e-5, rtol=1e-5)
    for kind, Bm, ldb in (('nn', dB, N), ('nt', dBt, K_)):
        C = torch.full((R, N), 7.0, device='cuda')
        K.gemm_rows(kind, n, N, K_, dA, K_, Bm, ldb, C, N, drows, bias=dev(bias))
        ref = A @ B + bias
        close(C[keep.cuda()], ref[keep], **tol)
        assert (C[~keep.cuda()] == 7.0).all()


def test_gemm_is_transpose_detecting(K):
    # A = I with an asymmetric B catches a swapped C layout (cdna guide §3)
    n = 96
    B = torch.arange(n * n, dtype=torch.float64).reshape(n, n)
    close(K.matmul_nn(dev(torch.eye(n, dtype=torch.float64)), dev(B)), B, atol=0, rtol=0)


def test_colsum(K):
    X = rnd(777, 130, seed=7)
    close(K.colsum(dev(X)), X.sum(0), atol=1e-4)


# ------------------------------------------------------------------ conv
def _conv_ref(x, w, b):
    N, H, W, C = x.shape
    pt, pb = oracle.same_pad_s2k3(H)
    pl, pr = oracle.same_pad_s2k3(W)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), bias=b, stride=2)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(40, 8, 8, 16, 16), (40, 4, 4, 16, 32), (40, 2, 2, 32, 48),
                                            (6, 80, 80, 3, 16), (7, 5, 5, 48, 48), (9, 3, 3, 48, 48),
                                            (3, 10, 10, 48, 48), (5, 7, 9, 4, 8), (5, 20, 20, 32, 48), (33, 20, 20, 32, 48), (64, 10, 10, 48, 48), (2, 9, 6, 32, 48),
                                            (41, 4, 4, 16, 32), (37, 2, 2, 32, 48), (33, 8, 8, 16, 16),
                                            (3, 20, 20, 16, 32), (2, 21, 19, 16, 16), (5, 11, 14, 4, 16),
                                            (1, 1, 1, 16, 16), (700, 8, 8, 16, 16),
                                            (3, 80, 80, 4, 16), (2, 37, 80, 4, 16), (3, 40, 40, 16, 32),
                                            (2, 21, 40, 16, 32), (70, 6, 80, 4, 16), (5, 6, 40, 16, 32), (3, 3, 40, 16, 32),
                                            (3, 23, 21, 16, 32), (2, 40, 38, 16, 32)])
def test_conv_fwd_dgrad_wgrad(K, N, H, W, Cin, Cout):
    x = rnd(N, H, W, Cin, seed=1).requires_grad_(True)
    w = rnd(3, 3, Cin, Cout, seed=2, scale=0.3).requires_grad_(True)
    b = rnd(Cout, seed=3)
    y = _conv_ref(x, w, b)
    dy = rnd(*y.shape, seed=4)
    y.backward(dy)
    out = K.conv_fwd(dev(x), dev(w), dev(b), act=0)
    close(out, y, atol=1e-4)
    close(K.conv_fwd(dev(x), dev(w), dev(b), act=1), oracle.lrelu(y), atol=1e-4)
    dw = torch.empty(3, 3, Cin, Cout, device='cuda')
    K.conv_wgrad(dev(x), dev(dy), dw)
    close(dw, w.grad, atol=1e-5 * N * H * W + 1e-4, rtol=1e-5)
    if Cout % 4 == 0:
        close(K.conv_dgrad(dev(dy), dev(w), (N, H, W, Cin)), x.grad, atol=1e-4)


def test_conv_same_padding_is_asymmetric(K):
    # 4x4 ramp, all-ones kernel -> (0,1) padding, not (1,1)
    x = torch.arange(16, dtype=torch.float64).reshape(1, 4, 4, 1)
    w = torch.ones(3, 3, 1, 1, dtype=torch.float64)
    out = K.conv_fwd(dev(x), dev(w), dev(torch.zeros(1)), act=0).cpu().reshape(2, 2)
    # top-left window covers rows 0..2, cols 0..2 of the unpadded image
    assert out[0, 0].item() == float(x[0, 0:3, 0:3, 0].sum())
    assert out[1, 1].item() == float(x[0, 2:4, 2:4, 0].sum())


def test_conv_uint8_input(K):
    g = torch.Generator().manual_seed(0)
    xu = torch.randint(0, 256, (4, 80, 80, 3), generator=g, dtype=torch.uint8)
    w = rnd(3, 3, 3, 16, seed=2, scale=0.05)
    b = rnd(16, seed=3)
    ref = _conv_ref(xu.double(), w, b)
    out = K.conv_fwd(xu.cuda(), dev(w), dev(b), act=0)
    close(out, ref, atol=2e-3, rtol=1e-5)
    dy = rnd(*ref.shape, seed=5)
    dw = torch.empty(3, 3, 3, 16, device='cuda')
    K.conv_wgrad(xu.cuda(), dev(dy), dw)
    dwf = torch.empty(3, 3, 3, 16, device='cuda')
    K.conv_wgrad(dev(xu.double()), dev(dy), dwf)
    close(dw, dwf, atol=1e-2, rtol=1e-5)


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(41, 8, 8, 16, 16), (7, 12, 10, 4, 16), (3, 8, 8, 16, 32),
                                            (3, 80, 80, 4, 16), (5, 9, 80, 4, 16)])
def test_conv_uint8_frames_equal_float_frames(K, N, H, W, Cin, Cout):
    """uint8 frames (the dataset's own precision) are widened on load: same result as feeding the
    float copy, on every conv back end that takes them (whole-frame, direct, implicit GEMM)."""
    g = torch.Generator().manual_seed(3)
    xu = torch.randint(0, 256 if Cin == 4 else 2, (N, H, W, Cin), generator=g, dtype=torch.uint8)
    w = rnd(3, 3, Cin, Cout, seed=2, scale=0.05)
    b = rnd(Cout, seed=3)
    ref = _conv_ref(xu.double(), w, b)
    out = K.conv_fwd(xu.cuda(), dev(w), dev(b), act=0)
    close(out, ref, atol=2e-3, rtol=1e-5)
    close(out, K.conv_fwd(dev(xu.double()), dev(w), dev(b), act=0), atol=1e-4)
    dy = rnd(*ref.shape, seed=5)
    dw = torch.empty(3, 3, Cin, Cout, device='cuda')
    K.conv_wgrad(xu.cuda(), dev(dy), dw)
    dwf = torch.empty(3, 3, Cin, Cout, device='cuda')
    K.conv_wgrad(dev(xu.double()), dev(dy), dwf)
    close(dw, dwf, atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize('B,G,T,H,Cin', [(2, 3, 4, 20, 32), (3, 2, 5, 10, 48), (2, 5, 4, 5, 48), (9, 10, 20, 10, 48),
                                        (1, 2, 30, 20, 32)])      # (3 000 pixels per sequence: the weight gradient's offset table covers half a sequence)
def test_wide_layers_forward_with_statistics_and_input_affine(K, B, G, T, H, Cin):
    """Round 6, conv_wide.hip: the 48-channel layers' forward pass (filter in LDS; conv3 32 -> 48 on 20x20, conv4 / conv5
    48 -> 48 on 10x10 / 5x5; models/ops.py:27-33) with the batch-norm folding of ConvBnFold -- statistics of its own
    output per demonstration index out of the launch (sequences whose pixel count is no multiple of 16 end in a masked
    tile: 10x10 -> 5x5 frames), and the input read through the previous layer's batch-norm apply -- against an fp64
    reference and against the plain launch."""
    N = B * G * T
    Ho = (H + 1) // 2
    x = rnd(N, H, H, Cin, seed=21)
    w, b = rnd(3, 3, Cin, 48, seed=22, scale=0.1), rnd(48, seed=23)
    grp = (torch.arange(N) // T) % G
    S = K.conv_bn_slices((N, H, H, Cin), 48, G, T)
    assert S > 0
    ref = oracle.lrelu(_conv_ref(x, w, b))
    st = torch.zeros(G * S * 48 * 2, dtype=torch.float64, device='cuda')
    a = K.conv_fwd_bn(dev(x), dev(w), dev(b), G, T, S, st, act=1)
    close(a, ref, atol=1e-4)
    assert torch.equal(a, K.conv_fwd(dev(x), dev(w), dev(b), act=1))        # the same products in the same order
    mean, rstd, var = (torch.empty(G, 48, device='cuda') for _ in range(3))
    K.bn_stats_from_partials(st, B * T * Ho * Ho, 48, G, S, None, None, mean, rstd, var)
    ad = a.double().cpu()
    for gi in range(G):
        v = ad[grp == gi].reshape(-1, 48)
        close(mean[gi], v.mean(0), atol=1e-6 * float(v.abs().max()) + 1e-7)
        close(var[gi], v.var(0, unbiased=False), rtol=1e-5, atol=1e-7)
    # the input through an affine per (index, channel): x * sc + sh, zero padding of the NORMALISED tensor
    sc = (rnd(G, Cin, seed=24).abs() + 0.5)
    sh = rnd(G, Cin, seed=25)
    xn = x * sc[grp].view(N, 1, 1, Cin) + sh[grp].view(N, 1, 1, Cin)
    ref2 = oracle.lrelu(_conv_ref(xn, w, b))
    x_ext = torch.empty(N * H * H * Cin + G * Cin, device='cuda')
    xd, pad = x_ext[:N * H * H * Cin].view(N, H, H, Cin), x_ext[N * H * H * Cin:].view(G, Cin)
    xd.copy_(dev(x))
    pad.copy_(dev(-sh / sc))
    st2 = torch.zeros_like(st)
    a2 = K.conv_fwd_bn(xd, dev(w), dev(b), G, T, S, st2, act=1, in_affine=(dev(sc), dev(sh), pad))
    close(a2, ref2, atol=2e-5 * float(ref2.abs().max()) + 1e-4)
    K.bn_stats_from_partials(st2, B * T * Ho * Ho, 48, G, S, None, None, mean, rstd, var)
    a2d = a2.double().cpu()
    for gi in range(G):
        v = a2d[grp == gi].reshape(-1, 48)
        close(mean[gi], v.mean(0), atol=1e-6 * float(v.abs().max()) + 1e-7)
        close(rstd[gi], 1.0 / torch.sqrt(v.var(0, unbiased=False) + 1e-3), rtol=1e-5)
    # the weight gradient with the same affine on its input (conv_wide_wgrad_kernel<.., true>: scale on the finished rows,
    # shift / scale added to what is loaded, out-of-image taps on the pad pixel), and the plain one, against fp64
    dy = rnd(N, Ho, Ho, 48, seed=26)
    xr = xn.double().requires_grad_(True)
    wr_ = w.double().requires_grad_(True)
    _conv_ref(xr, wr_, b.double()).backward(dy.double())
    dw = torch.empty(3, 3, Cin, 48, device='cuda')
    K.conv_wgrad_bn(xd, dev(dy), dw, G, T, (dev(sc), dev(sh)))
    close(dw, wr_.grad, atol=2e-5 * float(wr_.grad.abs().max()) + 1e-4, rtol=1e-5)
    dwp = torch.empty(3, 3, Cin, 48, device='cuda')
    K.conv_wgrad(dev(xn), dev(dy), dwp)
    close(dwp, wr_.grad, atol=2e-5 * float(wr_.grad.abs().max()) + 1e-4, rtol=1e-5)
    K.conv_wgrad(dev(xn), dev(dy), dw)
    assert torch.equal(dw, dwp)                                  # deterministic combine
    # the input gradient (2x2 blocks of input pixels sharing a 2x2 neighbourhood of dY), plain and with the batch-norm
    # backward partial sums of the layer underneath per demonstration index: (sum dX, sum dX * xhat), xhat from that
    # layer's pre-norm activation `act` and its statistics
    xr2 = x.double().requires_grad_(True)
    _conv_ref(xr2, w.double(), b.double()).backward(dy.double())
    dxp = K.conv_dgrad(dev(dy), dev(w), (N, H, H, Cin))
    close(dxp, xr2.grad, atol=1e-4)
    act = rnd(N, H, H, Cin, seed=27)
    mu, rs = rnd(G, Cin, seed=28), rnd(G, Cin, seed=29).abs() + 0.5
    Sd = K.conv_dgrad_bn_slices((N, H, H, Cin), 48, G, T)
    assert Sd > 0
    std = torch.zeros(G * Sd * Cin * 2, dtype=torch.float64, device='cuda')
    dxb = K.conv_dgrad_bn(dev(dy), dev(w), (N, H, H, Cin), dev(act), dev(mu), dev(rs), G, T, std, Sd)
    assert torch.equal(dxb, dxp)
    got = std.view(G, Sd, Cin, 2).sum(1).cpu()
    gd = xr2.grad
    for gi in range(G):
        sel = grp == gi
        xhat = (act[sel].double() - mu[gi].double()) * rs[gi].double()
        want0 = gd[sel].reshape(-1, Cin).sum(0)
        want1 = (gd[sel] * xhat).reshape(-1, Cin).sum(0)
        scale = float(gd[sel].abs().sum(dim=(0, 1, 2)).max())
        close(got[gi, :, 0], want0, atol=2e-6 * scale + 1e-5, rtol=1e-5)
        close(got[gi, :, 1], want1, atol=1e-5 * scale + 1e-5, rtol=1e-5)


@pytest.mark.parametrize('B,G,T', [(3, 2, 2), (2, 5, 4)])
def test_batch_norm_folded_into_the_conv_launches(K, B, G, T):
    """Round 5, the ViZDoom-size layers (models/ops.py:14-33 behind models/model_full.py:216-231): the forward conv launch
    leaves the batch-norm partial sums of its output per demonstration index (d2p_conv2d_nhwc_s2_same_fwd_bn), the next
    layer's forward conv and weight gradient read the PRE-norm activation through the folded affine instead of a
    materialised normalised tensor.  Against the separate launches and an fp64 reference of the statistics."""
    N = B * G * T
    g = torch.Generator().manual_seed(11)
    xu = torch.randint(0, 256, (N, 80, 80, 4), generator=g, dtype=torch.uint8)
    xu[..., 3] = 0
    w1, b1 = rnd(3, 3, 4, 16, seed=2, scale=0.02), rnd(16, seed=3)
    grp = (torch.arange(N) // T) % G
    # ---- layer 1: frames in, statistics out
    S1 = K.conv_bn_slices((N, 80, 80, 4), 16, G, T)
    assert S1 > 0
    st1 = torch.zeros(G * S1 * 16 * 2, dtype=torch.float64, device='cuda')
    # (the activation the next layer reads through the affine carries its G pad pixels right behind its last element)
    a1_ext = torch.empty(N * 1600 * 16 + G * 16, device='cuda')
    a1, pad = a1_ext[:N * 1600 * 16].view(N, 40, 40, 16), a1_ext[N * 1600 * 16:].view(G, 16)
    K.conv_fwd_bn(xu.cuda(), dev(w1), dev(b1), G, T, S1, st1, act=1, out=a1)
    ref1 = K.conv_fwd(xu.cuda(), dev(w1), dev(b1), act=1)
    assert torch.equal(a1, ref1)                                   # the same products in the same order
    gam, bet = rnd(16, seed=5) + 1.0, rnd(16, seed=6)
    mean, rstd, var = (torch.empty(G, 16, device='cuda') for _ in range(3))
    sc, sh = torch.empty(G, 16, device='cuda'), torch.empty(G, 16, device='cuda')
    n1 = B * T * 40 * 40
    K.bn_stats_from_partials(st1, n1, 16, G, S1, dev(gam), dev(bet), mean, rstd, var, affine=(sc, sh, pad))
    a1d = a1.double().cpu()
    for gi in range(G):
        v = a1d[grp == gi].reshape(-1, 16)
        mu, vv = v.mean(0), v.var(0, unbiased=False)
        close(mean[gi], mu, atol=1e-6 * float(mu.abs().max()) + 1e-7)
        close(var[gi], vv, rtol=1e-5, atol=1e-7)
        close(rstd[gi], 1.0 / torch.sqrt(vv + 1e-3), rtol=1e-5)
        close(sc[gi], gam.double() / torch.sqrt(vv + 1e-3), rtol=1e-5)
        close(sh[gi], bet.double() - mu * gam.double() / torch.sqrt(vv + 1e-3), rtol=1e-4, atol=1e-5)
        close(pad[gi], -sh[gi].double().cpu() / sc[gi].double().cpu(), rtol=1e-6, atol=1e-7)
    # the separate launches' statistics
    _, m_ref, r_ref, _ = K.bn_fwd(a1.view(N * 1600, 16), dev(gam), dev(bet), G, T * 1600)
    close(mean, m_ref.double().cpu(), rtol=1e-6, atol=1e-6)
    close(rstd, r_ref.double().cpu(), rtol=1e-5)
    # ---- layer 2: the affine on the way in, statistics out; weight gradient with the same affine
    y1 = K.bn_apply_fwd(a1.view(N * 1600, 16), dev(gam), dev(bet), mean, rstd, G, T * 1600).view(N, 40, 40, 16)
    w2, b2 = rnd(3, 3, 16, 32, seed=7, scale=0.1), rnd(32, seed=8)
    S2 = K.conv_bn_slices((N, 40, 40, 16), 32, G, T)
    assert S2 > 0
    st2 = torch.zeros(G * S2 * 32 * 2, dtype=torch.float64, device='cuda')
    a2 = K.conv_fwd_bn(a1, dev(w2), dev(b2), G, T, S2, st2, act=1, in_affine=(sc, sh, pad))
    ref2 = K.conv_fwd(y1, dev(w2), dev(b2), act=1)
    close(a2, ref2.double().cpu(), atol=2e-5 * float(ref2.abs().max()))     # (one fp32 rounding of the affine apart)
    mean2, rstd2 = torch.empty(G, 32, device='cuda'), torch.empty(G, 32, device='cuda')
    K.bn_stats_from_partials(st2, B * T * 400, 32, G, S2, None, None, mean2, rstd2)
    a2d = a2.double().cpu()
    for gi in range(G):
        v = a2d[grp == gi].reshape(-1, 32)
        close(mean2[gi], v.mean(0), atol=1e-6 * float(v.abs().max()) + 1e-7)
        close(rstd2[gi], 1.0 / torch.sqrt(v.var(0, unbiased=False) + 1e-3), rtol=1e-5)
    # statistics without the affine input (layer 1 not folded)
    st2b = torch.zeros_like(st2)
    a2b = K.conv_fwd_bn(y1, dev(w2), dev(b2), G, T, S2, st2b, act=1)
    assert torch.equal(a2b, ref2)
    dy2 = dev(rnd(N, 20, 20, 32, seed=9))
    dw, dw_ref = torch.empty(3, 3, 16, 32, device='cuda'), torch.empty(3, 3, 16, 32, device='cuda')
    K.conv_wgrad_bn(a1, dy2, dw, G, T, (sc, sh))
    K.conv_wgrad(y1, dy2, dw_ref)
    close(dw, dw_ref.double().cpu(), atol=2e-5 * float(dw_ref.abs().max()))
    # ---- layer 1 backward: the batch-norm backward's apply pass folded into the weight gradient
    dy1 = dev(rnd(N, 40, 40, 16, seed=12))
    a1f = a1.reshape(N * 1600, 16)
    dg_ref, db_ref, dbias_ref = (torch.empty(16, device='cuda') for _ in range(3))
    da_ref = K.bn_bwd(a1f, dy1.view(N * 1600, 16), dev(gam), mean, rstd, G, T * 1600, True, dg_ref, db_ref, dbias=dbias_ref)
    dw1_ref = torch.empty(3, 3, 4, 16, device='cuda')
    K.conv_wgrad(xu.cuda(), da_ref.view(N, 40, 40, 16), dw1_ref)
    assert K.conv_bnbwd_ok((N, 80, 80, 4), 16)
    coef = torch.empty(G, 16, 4, device='cuda')
    dg, db, dbias, dw1 = tuple(torch.empty(16, device='cuda') for _ in range(3)) + (torch.empty(3, 3, 4, 16, device='cuda'),)
    K.bn_bwd_coef(a1f, dy1.view(N * 1600, 16), dev(gam), mean, rstd, G, T * 1600, coef, dg, db)
    K.conv_wgrad_bnbwd(xu.cuda(), a1, dy1, coef, G, T, dw1, dbias)
    assert torch.equal(dg, dg_ref) and torch.equal(db, db_ref)          # the same sums by the same kernels
    close(dw1, dw1_ref.double().cpu(), atol=3e-5 * float(dw1_ref.abs().max()))
    close(dbias, dbias_ref.double().cpu(), atol=3e-5 * float(dbias_ref.abs().max()) + 1e-4 * float(da_ref.abs().max()))
    # ... against an fp64 reference of the whole chain (batch-norm backward, then the weight gradient)
    a64, dy64 = a1.double().cpu().reshape(N, 1600, 16), dy1.double().cpu().reshape(N, 1600, 16)
    da64 = torch.zeros_like(a64)
    for gi in range(G):
        sel = grp == gi
        v, d = a64[sel].reshape(-1, 16), dy64[sel].reshape(-1, 16)
        mu, rs = v.mean(0), 1.0 / torch.sqrt(v.var(0, unbiased=False) + 1e-3)
        xh = (v - mu) * rs
        dxx = gam.double() * rs * (d - d.mean(0) - xh * (d * xh).mean(0))
        dxx = dxx * torch.where(v > 0, torch.ones_like(v), torch.full_like(v, 0.2))
        da64[sel] = dxx.reshape(-1, 1600, 16)
    close(dbias, da64.sum((0, 1)), atol=1e-4 * float(da64.abs().max()) * 40)
    xr = xu.double().requires_grad_(False)
    wref = torch.zeros(3, 3, 4, 16, dtype=torch.float64, requires_grad=True)
    _conv_ref(xr, wref, torch.zeros(16, dtype=torch.float64)).backward(da64.reshape(N, 40, 40, 16))
    close(dw1, wref.grad, atol=1e-4 * float(wref.grad.abs().max()))
    # ---- conv2's input gradient leaves layer 1's batch-norm-backward partial sums: the same coefficients without a pass
    # over (a1, dy1)
    Sd = K.conv_dgrad_bn_slices((N, 40, 40, 16), 32, G, T)
    assert Sd > 0
    std = torch.zeros(G * Sd * 16 * 2, dtype=torch.float64, device='cuda')
    dx_ref = K.conv_dgrad(dy2, dev(w2), (N, 40, 40, 16))
    dx_bn = K.conv_dgrad_bn(dy2, dev(w2), (N, 40, 40, 16), a1, mean, rstd, G, T, std, Sd)
    assert torch.equal(dx_bn, dx_ref)
    coef_a, coef_b = torch.empty(G, 16, 4, device='cuda'), torch.empty(G, 16, 4, device='cuda')
    dga, dba, dgb, dbb = (torch.empty(16, device='cuda') for _ in range(4))
    K.bn_bwd_coef(a1f, dx_ref.view(N * 1600, 16), dev(gam), mean, rstd, G, T * 1600, coef_a, dga, dba)
    K.bn_bwd_coef(a1f, dx_ref.view(N * 1600, 16), dev(gam), mean, rstd, G, T * 1600, coef_b, dgb, dbb, sums=(std, Sd))
    sc_ = float(coef_a.abs().max())
    close(coef_b, coef_a.double().cpu(), atol=2e-5 * sc_)
    close(dgb, dga.double().cpu(), atol=2e-5 * float(dga.abs().max()) + 1e-6)
    close(dbb, dba.double().cpu(), atol=2e-5 * float(dba.abs().max()) + 1e-6)
    # geometries without folding kernels are refused, not silently run
    assert K.conv_dgrad_bn_slices((N, 8, 8, 16), 32, G, T) == 0
    assert not K.conv_bnbwd_ok((N, 8, 8, 16), 16)
    assert K.conv_bn_slices((N, 8, 8, 16), 16, G, T) == 0
    with pytest.raises(Exception):
        K.conv_fwd_bn(dev(rnd(N, 8, 8, 16)), dev(rnd(3, 3, 16, 16)), dev(rnd(16)), G, T, 1, st1, act=1)


# ------------------------------------------------------------------ scheduled sampling
def test_sched_sample_statistics(K):
    """d2p_sched_sample: p = 0 keeps the ground truth, p = 1 always draws, the draw frequencies
    follow softmax(logits) (chi-square), the take rate follows p, and a new step counter gives
    new noise while the same counter reproduces it (graph replays read it from device memory)."""
    M, V = 20000, 6
    logits = torch.tensor([0.5, -1.0, 2.0, 0.0, 1.0, -3.0]).repeat(M, 1).cuda()
    gt = torch.full((M,), 4, dtype=torch.int32, device='cuda')
    rng = torch.tensor([1234, 7], dtype=torch.int64, device='cuda')
    out = torch.empty(M, dtype=torch.int32, device='cuda')
    flag = torch.empty(M, dtype=torch.int32, device='cuda')

    def draw(p, t=3):
        K.sched_sample(logits, gt, torch.tensor([p], device='cuda'), rng, t, out, flag)
        return out.cpu().numpy().copy(), flag.cpu().numpy().copy()

    ids, fl = draw(0.0)
    assert (ids == 4).all() and not fl.any()
    ids, fl = draw(1.0)
    assert fl.all()
    prob = torch.softmax(logits[0].double().cpu(), 0).numpy()
    cnt = np.bincount(ids, minlength=V)
    chi2 = ((cnt - M * prob) ** 2 / (M * prob)).sum()
    assert chi2 < 25.0, (chi2, cnt, M * prob)             # 5 dof: P(chi2 > 25) ~ 1e-4
    ids2, _ = draw(1.0)
    assert (ids2 == ids).all()                              # same (seed, counter, t): same noise
    rng[1] = 8
    ids3, _ = draw(1.0)
    assert (ids3 != ids).mean() > 0.3
    ids4, _ = draw(1.0, t=4)
    assert (ids4 != ids3).mean() > 0.3
    _, fl = draw(0.3)
    assert abs(fl.mean() - 0.3) < 0.015


# ------------------------------------------------------------------ batch norm
@pytest.mark.parametrize('B,k,inner,C,act', [(3, 4, 20, 16, True), (2, 10, 8, 48, True),
                                             (5, 1, 7, 512, True), (4, 3, 1, 512, False),
                                             (3, 2, 5, 6, True), (64, 10, 32, 16, True)])
def test_bn_group_fwd_bwd(K, B, k, inner, C, act):
    R = B * k * inner
    x = rnd(R, C, seed=1, scale=3.0)
    x[0, :] = 0.0                                    # exercises lrelu'(0) = 0.6
    pre = x.clone().requires_grad_(True)             # pre-activation
    gamma = (rnd(C, seed=2) + 1.5).requires_grad_(True)
    beta = rnd(C, seed=3).requires_grad_(True)
    a = oracle.lrelu(pre) if act else pre
    a4 = a.reshape(B, k, inner, C)
    ys, means, vars_ = [], [], []
    for i in range(k):                               # one BN call per demo index
        y, m, v = oracle.batch_norm_train(a4[:, i], beta, gamma)
        ys.append(y); means.append(m); vars_.append(v)
    yref = torch.stack(ys, dim=1).reshape(R, C)
    dy = rnd(R, C, seed=4)
    yref.backward(dy)
    a_dev = dev(a)
    var = torch.empty(k, C, device='cuda')
    mm_d, mv_d = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    y, mean, rstd, var = K.bn_fwd(a_dev, dev(gamma), dev(beta), k, inner, var=var, moving=(mm_d, mv_d))
    close(y, yref, atol=1e-4)
    mm_ref, mv_ref = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    for i in range(k):                               # one moving-average update per call
        mm_ref = 0.9 * mm_ref + 0.1 * means[i]
        mv_ref = 0.9 * mv_ref + 0.1 * vars_[i]
    close(mm_d, mm_ref, atol=1e-5)
    close(mv_d, mv_ref, atol=1e-5)
    close(mean, torch.stack(means), atol=1e-5)
    close(var, torch.stack(vars_), atol=1e-5)
    dgamma = torch.empty(C, device='cuda')
    dbeta = torch.empty(C, device='cuda')
    dx = K.bn_bwd(a_dev, dev(dy), dev(gamma), mean, rstd, k, inner, act, dgamma, dbeta)
    close(dx, pre.grad, atol=1e-4)
    # the same pass can also emit colsum(dx) = the gradient of a bias added before the activation
    dbias = torch.empty(C, device='cuda')
    dx2 = K.bn_bwd(a_dev, dev(dy), dev(gamma), mean, rstd, k, inner, act, dgamma, dbeta, dbias=dbias)
    close(dx2, dx, atol=1e-6)
    close(dbias, pre.grad.sum(0), atol=1e-3, rtol=1e-4)
    close(dgamma, gamma.grad, atol=1e-3, rtol=1e-4)
    close(dbeta, beta.grad, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize('bits', [1, 3])
def test_bn_folded_finalize_writes_every_gradient(K, bits):
    """d2p_bn_set_fold (the A/B switch of the ticket-folded finalize steps): whatever combination of folds a layer's
    size admits -- few partial sums (ticket fold), many (no ticket fold, folded column-sum finalize or not) -- dx,
    dgamma, dbeta and the bias gradient are all written and equal the default launches' (with many
    partial sums and one wavefront per (group, channel) nobody wrote dgamma / dbeta)."""
    shapes = [(3, 4, 20, 16), (64, 10, 32, 16), (40, 10, 64, 48), (16, 10, 20 * 16, 32), (5, 1, 7, 512),
              (32, 10, 20 * 64, 16)]
    try:
        for (B, k, inner, C) in shapes:
            R = B * k * inner
            a_dev, dy = dev(rnd(R, C, seed=11, scale=2.0)), dev(rnd(R, C, seed=12))
            gamma, beta = dev(rnd(C, seed=13) + 1.5), dev(rnd(C, seed=14))
            outs = []
            for fold in (0, bits):
                K.bn_set_fold(fold)
                _, mean, rstd, _ = K.bn_fwd(a_dev, gamma, beta, k, inner)
                dgamma, dbeta, dbias = (torch.full((C,), float('nan'), device='cuda') for _ in range(3))
                dx = K.bn_bwd(a_dev, dy, gamma, mean, rstd, k, inner, True, dgamma, dbeta, dbias=dbias)
                outs.append((dx.clone(), dgamma, dbeta, dbias))
            for name, ref, got in zip(('dx', 'dgamma', 'dbeta', 'dbias'), outs[0], outs[1]):
                assert torch.isfinite(got).all(), (name, B, k, inner, C)
                close(got, ref, atol=1e-4, rtol=1e-4)
    finally:
        K.bn_set_fold(0)


def test_bn_inference_mode(K):
    """is_training=False: normalise with the moving statistics (evaler.py:61)."""
    for R, C in ((37, 48), (20, 6), (640, 512)):
        x, gamma, beta = rnd(R, C, seed=1, scale=2.0), rnd(C, seed=2) + 1.5, rnd(C, seed=3)
        mm, mv = rnd(C, seed=4), rnd(C, seed=5).abs() + 0.1
        ref = oracle.batch_norm_infer(x, beta, gamma, mm, mv)
        close(K.bn_inference(dev(x), dev(gamma), dev(beta), dev(mm), dev(mv)), ref, atol=1e-5, rtol=1e-5)


def test_bn_moving_average_k_updates(K):
    mean, var = rnd(5, 8, seed=1), rnd(5, 8, seed=2).abs()
    mm, mv = torch.zeros(8, dtype=torch.float64), torch.ones(8, dtype=torch.float64)
    mmd, mvd = dev(mm), dev(mv)
    K.bn_update_moving(dev(mean), dev(var), mmd, mvd)
    for g in range(5):
        mm = 0.9 * mm + 0.1 * mean[g]
        mv = 0.9 * mv + 0.1 * var[g]
    close(mmd, mm, atol=1e-6)
    close(mvd, mv, atol=1e-6)


# ------------------------------------------------------------------ LSTM
def _lstm_case(K, M, T, I, U, masked, with_init, strided):
    x = rnd(M, T, I, seed=1).requires_grad_(True)
    kernel = rnd(I + U, 4 * U, seed=2, scale=0.3).requires_grad_(True)
    bias = rnd(4 * U, seed=3, scale=0.3).requires_grad_(True)
    h0 = rnd(M, U, seed=4).requires_grad_(True) if with_init else None
    c0 = rnd(M, U, seed=5).requires_grad_(True) if with_init else None
    g = torch.Generator().manual_seed(6)
    lens = torch.randint(0 if masked else 1, T + 1, (M,), generator=g)
    lens[0] = T
    if masked:
        outs, h, c = oracle.dynamic_rnn(x, lens, kernel, bias, c0=c0, h0=h0)
        n_steps = T
    else:                                           # decoder: no masking, n_steps = max(len)
        n_steps = T - 1
        cc = c0 if with_init else torch.zeros(M, U, dtype=torch.float64)
        hh = h0 if with_init else torch.zeros(M, U, dtype=torch.float64)
        outl = []
        for t in range(n_steps):
            cc, hh = oracle.basic_lstm_cell(x[:, t], cc, hh, kernel, bias)
            outl.append(hh)
        outs, h, c = torch.stack(outl, dim=1), hh, cc
    douts = rnd(*outs.shape, seed=7)
    dh, dc = rnd(M, U, seed=8), rnd(M, U, seed=9)
    (outs * douts).sum().backward(retain_graph=True)
    ((h * dh).sum() + (c * dc).sum()).backward()

    Wx, Wh = dev(kernel[:I]), dev(kernel[I:])
    # hoisted input projection in the chosen layout
    if strided:    # (m, t)-ordered rows: row stride T*4U, t stride 4U
        z = K.matmul_nn(dev(x.reshape(M * T, I)), Wx, bias=dev(bias))
        zrs, zts = T * 4 * U, 4 * U
    else:          # time-major
        z = K.matmul_nn(dev(x.transpose(0, 1).reshape(T * M, I)), Wx, bias=dev(bias))
        zrs, zts = 4 * U, M * 4 * U
    hout = torch.empty(n_steps, M, U, device='cuda')
    cs = torch.empty(n_steps, M, U, device='cuda')
    hf, cf = torch.empty(M, U, device='cuda'), torch.empty(M, U, device='cuda')
    lens_d = lens.to(torch.int32).cuda() if masked else None
    K.lstm_seq_fwd(z, zrs, zts, M, U, n_steps, Wh, dev(h0) if with_init else None,
                   dev(c0) if with_init else None, lens_d, hout, cs, hf, cf)
    close(hout.transpose(0, 1), outs, atol=2e-5)
    close(hf, h, atol=2e-5)
    close(cf, c, atol=2e-5)
    dz = torch.zeros_like(z)
    dh0, dc0 = torch.empty(M, U, device='cuda'), torch.empty(M, U, device='cuda')
    K.lstm_seq_bwd(z, zrs, zts, M, U, n_steps, Wh, dev(c0) if with_init else None, lens_d, cs,
                   dev(douts.transpose(0, 1)), dev(dh), dev(dc), dz, dh0, dc0)
    # parameter / input grads from dz
    if strided:
        dz2 = dz.reshape(M * T, 4 * U)
        x2 = dev(x.reshape(M * T, I))
        hprev = torch.cat([dev(h0).unsqueeze(0) if with_init else torch.zeros(1, M, U, device='cuda'),
                           hout[:-1]], 0)
        dzt = dz.reshape(M, T, 4 * U).transpose(0, 1).contiguous()[:n_steps]
        dx = K.matmul_nt(dz2, Wx).reshape(M, T, I)
    else:
        dz2 = dz[:n_steps * M]
        x2 = dev(x.transpose(0, 1).reshape(T * M, I))[:n_steps * M]
        hprev = torch.cat([dev(h0).unsqueeze(0) if with_init else torch.zeros(1, M, U, device='cuda'),
                           hout[:-1]], 0)
        dzt = dz2.reshape(n_steps, M, 4 * U)
        dx = torch.zeros(T, M, I, device='cuda')
        dx[:n_steps] = K.matmul_nt(dz2, Wx).reshape(n_steps, M, I)
        dx = dx.transpose(0, 1)
    dWx = K.matmul_tn(x2, dz2.reshape(x2.shape[0], 4 * U))
    dWh = K.matmul_tn(hprev.reshape(n_steps * M, U), dzt.reshape(n_steps * M, 4 * U))
    db = K.colsum(dz2.reshape(-1, 4 * U))
    tol = dict(atol=3e-4, rtol=1e-4)
    close(dx, x.grad, **tol)
    close(dWx, kernel.grad[:I], **tol)
    close(dWh, kernel.grad[I:], **tol)
    close(db, bias.grad, **tol)
    if with_init:
        close(dh0, h0.grad, **tol)
        close(dc0, c0.grad, **tol)


@pytest.mark.parametrize('backend', ['persistent', 'step', 'unfused'])
@pytest.mark.parametrize('masked,with_init,strided', [(True, False, True), (True, True, False),
                                                      (False, True, False), (False, False, False)])
def test_lstm_seq_fwd_bwd(K, masked, with_init, strided, backend):
    """The three back ends of d2p_lstm_seq_fwd/_bwd (one persistent launch per sequence, one fused
    launch per step, generic GEMM + gate kernel per step) against the oracle."""
    K.set_lstm_fused(backend != 'unfused')
    K.set_lstm_persistent(backend == 'persistent')
    try:
        _lstm_case(K, M=12, T=6, I=20, U=64, masked=masked, with_init=with_init, strided=strided)
        assert K.lstm_persist_error() == 0
    finally:
        K.set_lstm_fused(True)
        K.set_lstm_persistent(True)


def _persist_vs_step(K, M, U, T, masked, with_init, seed, reps=2):
    """Persistent kernels against the per-step kernels on the same inputs: the forward outputs must be
    bit-identical (same K split, same summation order, shared cell math), the backward within fp32
    round-off (the persistent kernel sums each wave's product in two accumulators).  A stale in-launch
    hand-off would show up as a forward mismatch."""
    g = torch.Generator().manual_seed(seed)
    z0 = ((torch.rand(T * M, 4 * U, generator=g) - 0.5) * 2).cuda()
    Wh = ((torch.rand(U, 4 * U, generator=g) - 0.5) * 0.2).cuda()
    h0 = (torch.rand(M, U, generator=g) - 0.5).cuda() if with_init else None
    c0 = (torch.rand(M, U, generator=g) - 0.5).cuda() if with_init else None
    lens = None
    if masked:
        lens = torch.randint(0, T + 1, (M,), generator=g)
        lens[0] = T
        lens = lens.to(torch.int32).cuda()
    dhout = (torch.rand(T, M, U, generator=g) - 0.5).cuda()
    dhf, dcf = (torch.rand(M, U, generator=g) - 0.5).cuda(), (torch.rand(M, U, generator=g) - 0.5).cuda()

    def run():
        z = z0.clone()
        o = dict(z=z, hout=torch.full((T, M, U), float('nan'), device='cuda'),
                 cs=torch.full((T, M, U), float('nan'), device='cuda'),
                 hf=torch.empty(M, U, device='cuda'), cf=torch.empty(M, U, device='cuda'),
                 dz=torch.full((T * M, 4 * U), float('nan'), device='cuda'),
                 dh0=torch.empty(M, U, device='cuda'), dc0=torch.empty(M, U, device='cuda'))
        K.lstm_seq_fwd(z, 4 * U, M * 4 * U, M, U, T, Wh, h0, c0, lens, o['hout'], o['cs'], o['hf'], o['cf'])
        K.lstm_seq_bwd(z, 4 * U, M * 4 * U, M, U, T, Wh, c0, lens, o['cs'], dhout, dhf, dcf, o['dz'], o['dh0'],
                       o['dc0'])
        torch.cuda.synchronize()
        return o
    K.set_lstm_persistent(False)
    try:
        ref = run()
    finally:
        K.set_lstm_persistent(True)
    for _ in range(reps):
        got = run()
        assert K.lstm_persist_error() == 0
        for n in ('z', 'hout', 'cs', 'hf', 'cf'):
            assert torch.equal(got[n], ref[n]), n
        for n in ('dz', 'dh0', 'dc0'):
            err = (got[n] - ref[n]).abs().max().item()
            assert err <= 2e-5 * ref[n].abs().max().item() + 1e-7, (n, err)


@pytest.mark.parametrize('M,U,T,masked,with_init', [
    (320, 512, 20, True, True),      # BASELINE config 2 / 4: 4 row domains x 5 phases, deferred epilogue
    (320, 512, 20, False, False),
    (32, 512, 40, False, True),      # program decoder: single-phase domains (no look-ahead)
    (400, 512, 12, True, True),      # k = 25, B = 16: 6-7 phases per domain
    (48, 512, 5, True, False),       # 3 single-phase domains, last rows padded
    (96, 512, 6, False, True),       # 2 phases per domain... look-ahead without deferral
    (176, 512, 5, True, True),       # 3 phases per domain
    (320, 64, 30, True, True), (35, 128, 4, True, True), (80, 256, 5, False, True), (512, 512, 4, False, True)])
def test_lstm_persistent_equals_per_step(K, M, U, T, masked, with_init):
    _persist_vs_step(K, M, U, T, masked, with_init, seed=M + U + T)


def test_lstm_persistent_long_sequence(K):
    """4000 hand-offs per workgroup: rare stale reads would surface here."""
    _persist_vs_step(K, 320, 512, 160, True, True, seed=3, reps=3)


@pytest.mark.parametrize('M,U', [(35, 128), (80, 256), (48, 512)])
def test_lstm_seq_fused_other_widths(K, M, U):
    _lstm_case(K, M=M, T=4, I=12, U=U, masked=True, with_init=True, strided=False)
    _lstm_case(K, M=M, T=4, I=12, U=U, masked=False, with_init=False, strided=False)


@pytest.mark.parametrize('M,masked', [(320, True), (320, False), (32, False), (400, True)])
def test_lstm_fused_equals_unfused_at_full_size(K, M, masked):
    """BASELINE config sizes (M = B*k = 320 rows, U = 512; M = 32 program rows; M = 400 for
    k=25): the fused recurrent-step path must reproduce the GEMM + gate-kernel path."""
    T, U = 7, 512
    g = torch.Generator().manual_seed(5)
    z0 = (torch.rand(T * M, 4 * U, generator=g) * 2 - 1).cuda()
    Wh = ((torch.rand(U, 4 * U, generator=g) * 2 - 1) * 0.05).cuda()
    h0 = (torch.rand(M, U, generator=g) * 2 - 1).cuda()
    c0 = (torch.rand(M, U, generator=g) * 2 - 1).cuda()
    lens = torch.randint(0, T + 1, (M,), generator=g).int().cuda() if masked else None
    dhout = (torch.rand(T, M, U, generator=g) * 2 - 1).cuda()
    dhf = (torch.rand(M, U, generator=g) * 2 - 1).cuda()
    dcf = (torch.rand(M, U, generator=g) * 2 - 1).cuda()
    res = []
    for fused in (True, False):
        K.set_lstm_fused(fused)
        z = z0.clone()
        hout, cs = torch.empty(T, M, U, device='cuda'), torch.empty(T, M, U, device='cuda')
        hf, cf = torch.empty(M, U, device='cuda'), torch.empty(M, U, device='cuda')
        K.lstm_seq_fwd(z, 4 * U, M * 4 * U, M, U, T, Wh, h0, c0, lens, hout, cs, hf, cf)
        dz = torch.zeros_like(z)
        dh0, dc0 = torch.empty(M, U, device='cuda'), torch.empty(M, U, device='cuda')
        K.lstm_seq_bwd(z, 4 * U, M * 4 * U, M, U, T, Wh, c0, lens, cs, dhout, dhf, dcf, dz, dh0, dc0)
        res.append((hout, cs, hf, cf, dz, dh0, dc0, z))
    K.set_lstm_fused(True)
    names = ('hout', 'cs', 'h_final', 'c_final', 'dz', 'dh0', 'dc0', 'z')
    for n, a, b in zip(names, res[0], res[1]):
        if masked and n == 'z':
            continue          # masked rows: the unfused path still accumulates h·Wh into unused z rows
        err = (a - b).abs().max().item()
        assert err <= 2e-5 * max(1.0, b.abs().max().item()), (n, err)


def test_lstm_multi_sequence_launch_equals_separate_calls(K):
    """Horizontally fused decoders: three independent LSTMs (different M, n_steps, weights)
    advanced by shared launches must give exactly what three separate calls give."""
    U = 512
    g = torch.Generator().manual_seed(9)
    specs = [(32, 11), (320, 6), (320, 6)]
    def mk(M, n, i):
        T = n + 1
        return dict(M=M, U=U, n_steps=n,
                    z0=(torch.rand(T * M, 4 * U, generator=g) * 2 - 1).cuda(),
                    Wh=((torch.rand(U, 4 * U, generator=g) * 2 - 1) * 0.05).cuda(),
                    h0=(torch.rand(M, U, generator=g) * 2 - 1).cuda(),
                    c0=(torch.rand(M, U, generator=g) * 2 - 1).cuda(),
                    dhout=(torch.rand(T, M, U, generator=g) * 2 - 1).cuda(), T=T)
    base = [mk(M, n, i) for i, (M, n) in enumerate(specs)]
    results = []
    for multi in (True, False):
        fw, bw, outs = [], [], []
        for b in base:
            M, T = b['M'], b['T']
            o = dict(z=b['z0'].clone(), hout=torch.zeros(T, M, U, device='cuda'),
                     cs=torch.zeros(T, M, U, device='cuda'), dz=torch.zeros(T * M, 4 * U, device='cuda'),
                     dh0=torch.zeros(M, U, device='cuda'), dc0=torch.zeros(M, U, device='cuda'))
            outs.append(o)
            fw.append(dict(M=M, U=U, n_steps=b['n_steps'], z=o['z'], Wh=b['Wh'], h0=b['h0'], c0=b['c0'],
                           hout=o['hout'], cs=o['cs']))
            bw.append(dict(M=M, U=U, n_steps=b['n_steps'], z=o['z'], Wh=b['Wh'], c0=b['c0'], cs=o['cs'],
                           dhout=b['dhout'], dz=o['dz'], dh0=o['dh0'], dc0=o['dc0']))
        K.set_lstm_persistent(False)     # the shared launches are per-step kernels: compare like with like
        if multi:
            K.lstm_seq_fwd_multi(fw)
            K.lstm_seq_bwd_multi(bw)
        else:
            for f, b_, base_b in zip(fw, bw, base):
                M = f['M']
                K.lstm_seq_fwd(f['z'], 4 * U, M * 4 * U, M, U, f['n_steps'], f['Wh'], f['h0'], f['c0'], None,
                               f['hout'], f['cs'], None, None)
                K.lstm_seq_bwd(b_['z'], 4 * U, M * 4 * U, M, U, b_['n_steps'], b_['Wh'], b_['c0'], None,
                               b_['cs'], b_['dhout'], None, None, b_['dz'], b_['dh0'], b_['dc0'])
        K.set_lstm_persistent(True)
        results.append(outs)
    for a, b in zip(*results):
        for n in ('z', 'hout', 'cs', 'dz', 'dh0', 'dc0'):
            assert torch.equal(a[n], b[n]), n


@pytest.mark.parametrize('specs,pairs', [([(320, 20), (32, 50)], 2), ([(32, 50), (320, 20)], 2),
                                         # 400 rows = 25 sub-tiles: 3 of the narrow forward kernel's 4 row domains could
                                         # not hold them (9 > 8 phases); 7 of the wide kernel's 8 do
                                         ([(400, 8), (16, 23)], 2)])
def test_lstm_two_sequences_in_one_persistent_launch(K, specs, pairs):
    """d2p_lstm_seq_{fwd,bwd}_multi with two sequences (the action and the program decoder: 320 rows x 20 steps
    beside 32 rows x 50 steps) puts both on disjoint workgroups of ONE persistent launch -- the small,
    latency-bound one hides beside the large one.  Only the row-domain split changes: results are bit-identical
    to two persistent launches, and the pair path must really have been taken."""
    from demo2program_amd.lib import load
    lib = load()                              # raw handle: this entry point returns a count, not a status
    U = 512
    g = torch.Generator().manual_seed(19)

    def mk(M, n):
        T = n
        return dict(M=M, n_steps=n, T=T,
                    z0=(torch.rand(T * M, 4 * U, generator=g) * 2 - 1).cuda(),
                    Wh=((torch.rand(U, 4 * U, generator=g) * 2 - 1) * 0.05).cuda(),
                    h0=(torch.rand(M, U, generator=g) * 2 - 1).cuda(),
                    c0=(torch.rand(M, U, generator=g) * 2 - 1).cuda(),
                    dhout=(torch.rand(T, M, U, generator=g) * 2 - 1).cuda())
    base = [mk(M, n) for (M, n) in specs]
    results = []
    # (round 4: the forward pair is taken by the wide-tile kernel, which keeps its own count)
    count = lambda: lib.d2p_lstm_persist_pair_launches() + lib.d2p_lstm_persist_wide_launches(2)   # noqa: E731
    before = count()
    K.lstm_persist_error(True)
    for multi in (True, False):
        fw, bw, outs = [], [], []
        for b in base:
            M, T = b['M'], b['T']
            o = dict(z=b['z0'].clone(), hout=torch.zeros(T, M, U, device='cuda'),
                     cs=torch.zeros(T, M, U, device='cuda'), dz=torch.zeros(T * M, 4 * U, device='cuda'),
                     dh0=torch.zeros(M, U, device='cuda'), dc0=torch.zeros(M, U, device='cuda'))
            outs.append(o)
            fw.append(dict(M=M, U=U, n_steps=b['n_steps'], z=o['z'], Wh=b['Wh'], h0=b['h0'], c0=b['c0'],
                           hout=o['hout'], cs=o['cs']))
            bw.append(dict(M=M, U=U, n_steps=b['n_steps'], z=o['z'], Wh=b['Wh'], c0=b['c0'], cs=o['cs'],
                           dhout=b['dhout'], dz=o['dz'], dh0=o['dh0'], dc0=o['dc0']))
        if multi:
            K.lstm_seq_fwd_multi(fw)
            K.lstm_seq_bwd_multi(bw)
        else:
            for f, b_ in zip(fw, bw):
                K.lstm_seq_fwd_multi([f])
                K.lstm_seq_bwd_multi([b_])
        results.append(outs)
    assert K.lstm_persist_error(True) == 0
    assert count() == before + pairs       # forward and backward pair launches
    for a, b in zip(*results):
        for n in ('z', 'hout', 'cs', 'dz', 'dh0', 'dc0'):
            assert torch.equal(a[n], b[n]), n


def test_lstm_three_backward_sequences_in_one_persistent_launch(K):
    """d2p_lstm_seq_bwd_multi with the three decoders (2 x 320 rows x 20 steps, 32 rows x 50 steps): one launch with
    3 + 3 + 2 row domains; bit-identical to three launches."""
    from demo2program_amd.lib import load
    lib = load()
    U = 512
    g = torch.Generator().manual_seed(23)
    specs = [(320, 20), (320, 20), (32, 50)]
    base = []
    for (M, n) in specs:
        base.append(dict(M=M, n=n, z0=(torch.rand(n * M, 4 * U, generator=g) * 2 - 1).cuda(),
                         Wh=((torch.rand(U, 4 * U, generator=g) * 2 - 1) * 0.05).cuda(),
                         h0=(torch.rand(M, U, generator=g) * 2 - 1).cuda(), c0=(torch.rand(M, U, generator=g) * 2 - 1).cuda(),
                         dhout=(torch.rand(n, M, U, generator=g) * 2 - 1).cuda()))
    results = []
    K.lstm_persist_error(True)
    before = lib.d2p_lstm_persist_pair_launches()
    for multi in (True, False):
        bw, outs = [], []
        for b in base:
            M, n = b['M'], b['n']
            o = dict(z=b['z0'].clone(), hout=torch.zeros(n, M, U, device='cuda'), cs=torch.zeros(n, M, U, device='cuda'),
                     dz=torch.zeros(n * M, 4 * U, device='cuda'), dh0=torch.zeros(M, U, device='cuda'),
                     dc0=torch.zeros(M, U, device='cuda'))
            K.lstm_seq_fwd_multi([dict(M=M, U=U, n_steps=n, z=o['z'], Wh=b['Wh'], h0=b['h0'], c0=b['c0'],
                                       hout=o['hout'], cs=o['cs'])])
            outs.append(o)
            bw.append(dict(M=M, U=U, n_steps=n, z=o['z'], Wh=b['Wh'], c0=b['c0'], cs=o['cs'], dhout=b['dhout'],
                           dz=o['dz'], dh0=o['dh0'], dc0=o['dc0']))
        if multi:
            K.lstm_seq_bwd_multi(bw)
        else:
            for b_ in bw:
                K.lstm_seq_bwd_multi([b_])
        results.append(outs)
    assert K.lstm_persist_error(True) == 0
    assert lib.d2p_lstm_persist_pair_launches() == before + 2
    for a, b in zip(*results):
        for n in ('dz', 'dh0', 'dc0'):
            assert torch.equal(a[n], b[n]), n


def test_gate_nonlinearities_are_fp32_accurate(K):
    """The gate math runs on the hardware exp2 / rcp units (common.h d2p_sigmoid / d2p_tanh): over the
    whole useful range, including saturation and tiny arguments, c' and h' of one cell step stay
    within 1e-6 absolute of an fp64 evaluation (fp32 rounding of the inputs alone is ~1e-7)."""
    M, U = 64, 512
    g = torch.Generator().manual_seed(5)
    z = (torch.rand(M, 4 * U, generator=g, dtype=torch.float64) * 2 - 1) * 12.0
    z[:8] *= 1e-3                                   # tiny pre-activations (tanh cancellation range)
    z[8:16] *= 8.0                                  # deep saturation, |z| up to ~100
    c_prev = (torch.rand(M, U, generator=g, dtype=torch.float64) * 2 - 1) * 2.0
    zf32, cf32 = z.float(), c_prev.float()
    zi, zj, zf, zo = [zf32.double()[:, i * U:(i + 1) * U] for i in range(4)]
    c_ref = cf32.double() * torch.sigmoid(zf + 1.0) + torch.sigmoid(zi) * torch.tanh(zj)
    h_ref = torch.tanh(c_ref) * torch.sigmoid(zo)
    c_out, h_out = torch.empty(M, U, device='cuda'), torch.empty(M, U, device='cuda')
    K.lstm_gate_fwd(zf32.cuda(), cf32.cuda(), None, None, 0, c_out, None, h_out)
    assert (c_out.double().cpu() - c_ref).abs().max().item() <= 1e-6 * max(1.0, c_ref.abs().max().item())
    assert (h_out.double().cpu() - h_ref).abs().max().item() <= 1e-6
    assert torch.isfinite(c_out).all() and torch.isfinite(h_out).all()


def test_lstm_known_answer_scalar_cell(K):
    # U=1-like check of gate order i,j,f,o and forget bias 1.0 (U padded to 4)
    U = 4
    z = torch.tensor([[0.5] * U + [-0.3] * U + [0.2] * U + [1.0] * U], dtype=torch.float64)
    c_prev = torch.full((1, U), 0.7, dtype=torch.float64)
    i, j, f, o = 0.5, -0.3, 0.2, 1.0
    sig = lambda v: 1 / (1 + math.exp(-v))
    c1 = 0.7 * sig(f + 1.0) + sig(i) * math.tanh(j)
    h1 = math.tanh(c1) * sig(o)
    c_out, h_out = torch.empty(1, U, device='cuda'), torch.empty(1, U, device='cuda')
    K.lstm_gate_fwd(dev(z), dev(c_prev), None, None, 0, c_out, None, h_out)
    assert abs(c_out[0, 0].item() - c1) < 1e-6 and abs(h_out[0, 0].item() - h1) < 1e-6


# ------------------------------------------------------------------ embedding
def test_embedding_gather_scatter_and_shift(K):
    V, E, R, T = 50, 64, 7, 5
    table = rnd(V + 1, E, seed=1)
    g = torch.Generator().manual_seed(2)
    tokens = torch.randint(0, V, (R, T), generator=g, dtype=torch.int32)
    ids = K.shift_tokens_tm(tokens.cuda(), V + 1)
    ref_ids = torch.cat([torch.full((R, 1), V + 1, dtype=torch.int32), tokens[:, :-1]], 1).t()
    assert torch.equal(ids.cpu(), ref_ids)
    out = K.embedding_gather(ids, dev(table))
    ref = oracle.model_full.embedding_lookup_oob0(table.float(), ref_ids.reshape(-1).long())
    close(out, ref.float(), atol=0, rtol=0)
    assert out[:R].abs().max().item() == 0          # <s> rows are zero vectors
    dout = rnd(T * R, E, seed=3)
    dtable = torch.empty(V + 1, E, device='cuda')
    K.embedding_scatter_add(ids, dev(dout), dtable)
    dref = torch.zeros(V + 1, E, dtype=torch.float64)
    flat = ref_ids.reshape(-1).long()
    ok = flat <= V
    dref.index_add_(0, flat[ok], dout[ok])
    close(dtable, dref, atol=1e-5)


# ------------------------------------------------------------------ losses
@pytest.mark.parametrize('mode,V,G', [('softmax', 50, 1), ('softmax', 6, 3), ('sigmoid', 5, 3)])
def test_xent_fwd_bwd(K, mode, V, G):
    B, T = 4, 9
    R = B * G
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(1, T + 1, (R,), generator=g)
    n_steps = int(lens.max())
    logits = rnd(T, R, V, seed=2, scale=3.0)
    logits[n_steps:] = 0
    logits = logits.requires_grad_(True)
    if mode == 'softmax':
        lab = F.one_hot(torch.randint(0, V, (R, T), generator=g), V).double()
    else:
        lab = torch.randint(0, 2, (R, T, V), generator=g).double()
    lab = lab * (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1)   # zero rows past len
    # oracle: one Sequence_Loss per group (demo index), mean over groups
    pred = logits.permute(1, 2, 0)                  # [R, V, T]
    gt = lab.permute(0, 2, 1)
    losses = []
    for gi in range(G):
        idx = torch.arange(gi, R, G)
        losses.append(oracle.sequence_loss(pred[idx], gt[idx], lens[idx], T, V,
                                           'program' if mode == 'softmax' else 'per'))
    loss = sum(losses) / G
    loss.backward()
    num, den = torch.empty(G, device='cuda'), torch.empty(G, device='cuda')
    K.xent_fwd(mode, dev(logits), dev(lab), 'rtv', lens.int().cuda(), T, R, V, G, n_steps, num, den)
    out = torch.empty(1, device='cuda')
    terms = torch.empty(1, device='cuda')
    K.loss_assemble([G], num, den, out, terms)
    close(out, loss.reshape(1), atol=1e-5)
    dl = torch.zeros(T, R, V, device='cuda')
    K.xent_bwd(mode, dev(logits), dev(lab), 'rtv', lens.int().cuda(), T, R, V, G, n_steps, den, 1.0, dl)
    close(dl, logits.grad, atol=1e-6)


def test_xent_program_label_layout_bvl(K):
    B, V, L = 3, 50, 8
    g = torch.Generator().manual_seed(3)
    lens = torch.tensor([8, 3, 5])
    toks = torch.randint(0, V, (B, L), generator=g)
    program = F.one_hot(toks, V).double().permute(0, 2, 1).contiguous()          # [B,V,L]
    program = program * (torch.arange(L) < lens.unsqueeze(1)).unsqueeze(1)
    logits = rnd(L, B, V, seed=4)
    ref = oracle.sequence_loss(logits.permute(1, 2, 0), program, lens, L, V, 'program')
    num, den = torch.empty(1, device='cuda'), torch.empty(1, device='cuda')
    K.xent_fwd('softmax', dev(logits), dev(program), 'bvl', lens.int().cuda(), L, B, V, 1, L, num, den)
    close(num / den, ref.reshape(1), atol=1e-5)


def test_zero_past_group_steps(K):
    T, B, G, V = 6, 3, 2, 4
    R = B * G
    lens = torch.tensor([2, 5, 3, 1, 1, 4])
    x = torch.ones(T, R, V, device='cuda')
    K.zero_past_group_steps(x, lens.int().cuda(), T, R, V, G)
    nst = [int(lens[0::2].max()), int(lens[1::2].max())]
    for t in range(T):
        for r in range(R):
            assert x[t, r, 0].item() == (1.0 if t < nst[r % G] else 0.0)


# ------------------------------------------------------------------ summarizer glue
def test_summarizer_glue(K):
    B, k, U = 3, 4, 8
    x = rnd(B, k, U, seed=1)
    out, bc = torch.empty(B, U, device='cuda'), torch.empty(B, k, U, device='cuda')
    K.group_mean(dev(x), B, k, U, out, bc)
    close(out, x.mean(1), atol=1e-6)
    close(bc, x.mean(1, keepdim=True).expand(B, k, U), atol=1e-6)
    dout, dbc = rnd(B, U, seed=2), rnd(B, k, U, seed=3)
    dx = dev(torch.ones(B, k, U, dtype=torch.float64))
    K.group_mean_bwd(dev(dout), dev(dbc), dx, B, k, U, True)
    close(dx, 1 + ((dout + dbc.sum(1)) / k).unsqueeze(1).expand(B, k, U), atol=1e-6)
    # rn pair: y[b,a,c] = lrelu(P[b,c] + Q[b,a] + bias)
    Pm, Qm, bias = rnd(B, k, U, seed=4), rnd(B, k, U, seed=5), rnd(U, seed=6)
    y = torch.empty(B, k, k, U, device='cuda')
    K.rn_pair_fwd(dev(Pm), dev(Qm), dev(bias), y, B, k, U)
    ref = oracle.lrelu(Pm.unsqueeze(1) + Qm.unsqueeze(2) + bias)
    close(y, ref, atol=1e-6)
    dy = rnd(B, k, k, U, seed=7)
    dP, dQ = torch.empty(B, k, U, device='cuda'), torch.empty(B, k, U, device='cuda')
    K.rn_pair_bwd(dev(dy), dP, dQ, B, k, U)
    close(dP, dy.sum(1), atol=1e-6)
    close(dQ, dy.sum(2), atol=1e-6)
    base = rnd(B, U, seed=8)
    o2 = torch.empty(B, U, device='cuda')
    K.pair_mean_fwd(dev(dy.reshape(B, k * k, U)), dev(base), o2, B, k * k, U)
    close(o2, dy.reshape(B, k * k, U).mean(1) + base, atol=1e-6)
    d2 = torch.empty(B, k * k, U, device='cuda')
    K.pair_mean_bwd(dev(dout), d2, B, k * k, U)
    close(d2, (dout / (k * k)).unsqueeze(1).expand(B, k * k, U), atol=1e-7)
    tr = K.transpose_rt(dev(x), B, k, U)
    close(tr, x.float().transpose(0, 1), atol=0, rtol=0)


def test_rn_factorisation_equals_concat_fc(K):
    # P + Q + b must equal fc1([feat_c || feat_a]) of models/model_full.py:335-343
    B, k, U = 2, 3, 16
    feat = rnd(B, k, U, seed=1)
    W1, b1 = rnd(2 * U, U, seed=2, scale=0.3), rnd(U, seed=3)
    tile1 = feat.unsqueeze(1).expand(B, k, k, U)
    tile2 = feat.unsqueeze(2).expand(B, k, k, U)
    ref = oracle.lrelu(torch.cat([tile1, tile2], 3).reshape(-1, 2 * U) @ W1 + b1)
    f2 = dev(feat.reshape(B * k, U))
    W1d = dev(W1)
    Pm = K.matmul_nn(f2, W1d[:U])
    Qm = K.matmul_nn(f2, W1d[U:])
    y = torch.empty(B * k * k, U, device='cuda')
    K.rn_pair_fwd(Pm, Qm, dev(b1), y, B, k, U)
    close(y, ref, atol=1e-5)


# ------------------------------------------------------------------ optimizer
@pytest.mark.parametrize('gscale', [0.01, 50.0])
def test_adam_clip_matches_oracle(K, gscale):
    n = 1003
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2, scale=gscale)
    params, grads = {'w': p0.clone()}, {'w': g0.clone()}
    m, v = {'w': torch.zeros(n, dtype=torch.float64)}, {'w': torch.zeros(n, dtype=torch.float64)}
    pd, md, vd = dev(p0), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    sumsq = torch.zeros(1, dtype=torch.float64, device='cuda')
    for step in (1, 2, 3):
        norm = oracle.adam_clip_step(params, grads, m, v, step, 1e-3)
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        K.l2norm_flat(dev(g0), 1.0, sumsq)
        assert abs(math.sqrt(sumsq.item()) - norm) < 1e-4 * max(1.0, norm)
        K.adam_clip_flat(pd, dev(g0), md, vd, sumsq, 1.0, 20.0, lr_t)
    close(pd, params['w'], atol=1e-6)
    close(md, m['w'], atol=1e-6, rtol=1e-5)


def test_adam_prescale_is_gradient_average(K):
    n = 256
    g = rnd(n, seed=3, scale=5.0)
    p1, p2 = dev(torch.zeros(n)), dev(torch.zeros(n))
    s = torch.zeros(1, dtype=torch.float64, device='cuda')
    for p, gg, pre in ((p1, g * 4, 0.25), (p2, g, 1.0)):
        m, v = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
        K.l2norm_flat(dev(gg), pre, s)
        K.adam_clip_flat(p, dev(gg), m, v, s, pre, 20.0, 1e-3)
    close(p1, p2, atol=1e-7)


def test_missing_gpu_tensor_fails_loudly(K):
    with pytest.raises(RuntimeError):
        K.matmul_nn(torch.zeros(4, 4), torch.zeros(4, 4))


@pytest.mark.gpu
@pytest.mark.parametrize('specs', [[(320, 20)], [(320, 20), (32, 50)], [(320, 6), (320, 6), (32, 9)], [(40, 5)],
                                   [(400, 4)], [(16, 3)]])
def test_lstm_bias_gradient_comes_out_of_the_backward_launch(K, specs):
    """d2p_lstm_bwd_desc.db: the cell's bias gradient (column sums of dz over every row and step) produced by the
    backward recurrence itself -- inside the persistent launch (per-workgroup sums folded by each column tile's last
    workgroup in domain order), or by a column-sum pass behind the per-step back ends.  Against the fp64 column
    sums of the dz the same call wrote; ragged row counts (40, 400 rows), sequence lengths, one / two / three
    sequences per launch; deterministic (two runs bit-identical)."""
    U = 512
    g = torch.Generator().manual_seed(29)
    K.lstm_persist_error(True)

    def run(persistent):
        K.set_lstm_persistent(persistent)
        fw, bw, outs = [], [], []
        gg = torch.Generator().manual_seed(31)
        for (M, n) in specs:
            lens = torch.randint(1, n + 1, (M,), generator=gg).int().cuda() if M != 32 else None
            o = dict(z=(torch.rand(n * M, 4 * U, generator=gg) * 2 - 1).cuda(), hout=torch.zeros(n, M, U, device='cuda'),
                     cs=torch.zeros(n, M, U, device='cuda'), dz=torch.zeros(n * M, 4 * U, device='cuda'),
                     dh0=torch.zeros(M, U, device='cuda'), dc0=torch.zeros(M, U, device='cuda'),
                     db=torch.full((4 * U,), 7.0, device='cuda'))
            Wh = ((torch.rand(U, 4 * U, generator=gg) * 2 - 1) * 0.05).cuda()
            h0 = (torch.rand(M, U, generator=gg) * 2 - 1).cuda()
            c0 = (torch.rand(M, U, generator=gg) * 2 - 1).cuda()
            dhout = (torch.rand(n, M, U, generator=gg) * 2 - 1).cuda()
            dhf = (torch.rand(M, U, generator=gg) * 2 - 1).cuda() if lens is not None else None
            outs.append(o)
            fw.append(dict(M=M, U=U, n_steps=n, z=o['z'], Wh=Wh, h0=h0, c0=c0, lens=lens, hout=o['hout'], cs=o['cs']))
            bw.append(dict(M=M, U=U, n_steps=n, z=o['z'], Wh=Wh, c0=c0, lens=lens, cs=o['cs'], dhout=dhout,
                           dh_final=dhf, dz=o['dz'], dh0=o['dh0'], dc0=o['dc0'], db=o['db']))
        K.lstm_seq_fwd_multi(fw)
        K.lstm_seq_bwd_multi(bw)
        torch.cuda.synchronize()
        return outs
    try:
        for persistent in (True, False):
            a = run(persistent)
            b = run(persistent)
            for o, o2 in zip(a, b):
                ref = o['dz'].double().sum(dim=0)
                scale = max(1.0, float(o['dz'].abs().double().sum(dim=0).max()))
                assert (o['db'].double() - ref).abs().max().item() <= 2e-6 * scale, persistent
                assert torch.equal(o['db'], o2['db']) and torch.equal(o['dz'], o2['dz'])
        assert K.lstm_persist_error(True) == 0
    finally:
        K.set_lstm_persistent(True)


@pytest.mark.gpu
@pytest.mark.parametrize('specs', [[(320, 20)], [(320, 20), (32, 50)], [(320, 6), (320, 6), (32, 9)], [(40, 5)]])
def test_lstm_packed_weight_images_kept_by_the_caller(K, specs):
    """d2p_lstm_pack_weights + d2p_lstm_*_desc.wpack: the persistent launches read the caller's packed weight images
    (one pack launch for all cells, off the critical path) instead of gathering fragments from the row-major Wh --
    bit-identical to the gathering form, forward and backward, one / two / three sequences per launch."""
    U = 512
    K.lstm_persist_error(True)

    def run(packed):
        fw, bw, outs, cells = [], [], [], []
        gg = torch.Generator().manual_seed(37)
        for (M, n) in specs:
            lens = torch.randint(1, n + 1, (M,), generator=gg).int().cuda() if M != 32 else None
            o = dict(z=(torch.rand(n * M, 4 * U, generator=gg) * 2 - 1).cuda(), hout=torch.zeros(n, M, U, device='cuda'),
                     cs=torch.zeros(n, M, U, device='cuda'), dz=torch.zeros(n * M, 4 * U, device='cuda'),
                     dh0=torch.zeros(M, U, device='cuda'), dc0=torch.zeros(M, U, device='cuda'),
                     db=torch.zeros(4 * U, device='cuda'))
            Wh = ((torch.rand(U, 4 * U, generator=gg) * 2 - 1) * 0.05).cuda()
            h0 = (torch.rand(M, U, generator=gg) * 2 - 1).cuda()
            c0 = (torch.rand(M, U, generator=gg) * 2 - 1).cuda()
            dhout = (torch.rand(n, M, U, generator=gg) * 2 - 1).cuda()
            wf = wb = None
            if packed:
                wf, wb = torch.zeros(4 * U * U, device='cuda'), torch.zeros(4 * U * U, device='cuda')
                cells.append((Wh, wf, wb))
            outs.append(o)
            fw.append(dict(M=M, U=U, n_steps=n, z=o['z'], Wh=Wh, h0=h0, c0=c0, lens=lens, hout=o['hout'], cs=o['cs'],
                           wpack=wf))
            bw.append(dict(M=M, U=U, n_steps=n, z=o['z'], Wh=Wh, c0=c0, lens=lens, cs=o['cs'], dhout=dhout,
                           dz=o['dz'], dh0=o['dh0'], dc0=o['dc0'], db=o['db'], wpack=wb))
        if cells:
            K.lstm_pack_weights(cells)
        K.lstm_seq_fwd_multi(fw)
        K.lstm_seq_bwd_multi(bw)
        torch.cuda.synchronize()
        return outs
    a, b = run(False), run(True)
    for o, o2 in zip(a, b):
        assert float(o['hout'].abs().max()) > 0 and float(o['dz'].abs().max()) > 0
        for name in ('hout', 'cs', 'dz', 'dh0', 'dc0', 'db'):
            assert torch.equal(o[name], o2[name]), name
    assert K.lstm_persist_error(True) == 0


@pytest.mark.gpu
@pytest.mark.parametrize('specs', [[(320, 20, 'lens')], [(320, 20, 'mask')], [(320, 6, 'mask'), (320, 6, 'mask'), (32, 9, None)],
                                   [(40, 5, 'lens')], [(400, 20, 'lens')], [(320, 20, 'lens'), (32, 50, None)],
                                   [(320, 20, 'lens0')], [(12, 7, 'lens')], [(320, 12, 'lens', 256)]])
def test_lstm_backward_over_rows_sorted_by_length(K, specs):
    """d2p_lstm_bwd_desc.rowmap / slab_steps: the backward recurrence groups rows of similar length into its row domains
    and runs each domain only for its longest row's steps.  'lens': a dynamic_rnn recurrence (lengths given, gradients of
    the final states); 'mask': a decoder (no lengths -- the incoming dhout is zero past each row's length, as behind a
    masked loss); 'lens0': lengths from 0, none reaching the step count; 12 rows: one ragged sub-tile; U = 256: more row
    domains than the sorted description takes (the order is then ignored).  dz, dh0, dc0 bit-identical to the launch that runs every step for every row (incl. the zeros of the
    skipped steps, over a dz buffer pre-filled with garbage); the bias gradient within summation-order round-off."""
    K.lstm_persist_error(True)

    def run(sort):
        fw, bw, outs = [], [], []
        gg = torch.Generator().manual_seed(41)
        for spec in specs:
            (M, n, mode), U = spec[:3], (spec[3] if len(spec) > 3 else 512)
            lens_h = torch.randint(max(1, n // 3), n + 1, (M,), generator=gg).int()
            if mode == 'lens0':      # rows of length 0 (nothing but their state passes through) and a batch whose
                mode = 'lens'        # longest row is shorter than the launch's step count
                lens_h = torch.clamp(lens_h - n // 3, max=n - 3)
            lens = lens_h.cuda() if mode == 'lens' else None
            o = dict(z=(torch.rand(n * M, 4 * U, generator=gg) * 2 - 1).cuda(), hout=torch.zeros(n, M, U, device='cuda'),
                     cs=torch.zeros(n, M, U, device='cuda'), dz=torch.full((n * M, 4 * U), 3.0, device='cuda'),
                     dh0=torch.zeros(M, U, device='cuda'), dc0=torch.zeros(M, U, device='cuda'),
                     db=torch.zeros(4 * U, device='cuda'))
            Wh = ((torch.rand(U, 4 * U, generator=gg) * 2 - 1) * 0.05).cuda()
            h0 = (torch.rand(M, U, generator=gg) * 2 - 1).cuda()
            c0 = (torch.rand(M, U, generator=gg) * 2 - 1).cuda()
            dhout = (torch.rand(n, M, U, generator=gg) * 2 - 1)
            if mode == 'mask':
                dhout = dhout * (torch.arange(n)[:, None] < lens_h[None, :]).float()[:, :, None]
            dhf = (torch.rand(M, U, generator=gg) * 2 - 1).cuda() if mode == 'lens' else None
            dcf = (torch.rand(M, U, generator=gg) * 2 - 1).cuda() if mode == 'lens' else None
            outs.append(o)
            fw.append(dict(M=M, U=U, n_steps=n, z=o['z'], Wh=Wh, h0=h0, c0=c0, lens=lens, hout=o['hout'], cs=o['cs']))
            bw.append(dict(M=M, U=U, n_steps=n, z=o['z'], Wh=Wh, c0=c0, lens=lens, cs=o['cs'], dhout=dhout.cuda(),
                           dh_final=dhf, dc_final=dcf, dz=o['dz'], dh0=o['dh0'], dc0=o['dc0'], db=o['db'],
                           row_order=K.lstm_row_order(lens_h.numpy()) if (sort and mode) else None))
        K.lstm_seq_fwd_multi(fw)
        K.lstm_seq_bwd_multi(bw)
        torch.cuda.synchronize()
        return outs
    a, b, c = run(False), run(True), run(True)
    for o, o2, o3 in zip(a, b, c):
        assert float(o['dz'].abs().max()) > 0
        for name in ('dz', 'dh0', 'dc0'):
            assert torch.equal(o[name], o2[name]), name
        scale = max(1.0, float(o['dz'].abs().double().sum(dim=0).max()))
        assert (o['db'].double() - o2['db'].double()).abs().max().item() <= 2e-6 * scale
        assert torch.equal(o2['db'], o3['db'])                       # deterministic
    assert K.lstm_persist_error(True) == 0


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,R,Kn', [(512, 2048, 6400, 4480), (48, 2048, 6400, 4512), (512, 2048, 1600, 864),
                                     (512, 512, 3296, 3200), (60, 256, 700, 333), (512, 512, 300, 64), (20, 36, 90, 50)])
def test_gemm_tn_over_lists_of_k_rows(K, M, N, R, Kn):
    """d2p_gemm_f32_tn_rows: C = A[rowsA]^T B[rowsB] (+ C) with both operands read through lists of K row indices
    (weight gradients over the rows inside their sequences; rowsA = rowsB - shift for dWh).  Against fp64, incl. a
    K that is not a multiple of the slab depth (the generic loaders), the accumulate form, and -- for lists padded
    with the index of a zero row -- equality with the dense product over all rows."""
    g = torch.Generator().manual_seed(M + N + Kn)
    shift = 7
    A = (torch.rand(R, M, generator=g) * 2 - 1).cuda()
    B = (torch.rand(R, N, generator=g) * 2 - 1).cuda()
    rowsB = (torch.randperm(R - shift, generator=g)[:Kn].sort().values + shift).int()
    rowsA = rowsB - shift
    C0 = (torch.rand(M, N, generator=g) * 2 - 1).cuda()
    ref = A.double().cpu()[rowsA.long()].t() @ B.double().cpu()[rowsB.long()]
    for acc in (False, True):
        C = C0.clone()
        K.gemm_tn_rows(M, N, Kn, A, M, rowsA.cuda(), B, N, rowsB.cuda(), C, N, accumulate=acc)
        want = ref + (C0.double().cpu() if acc else 0)
        assert (C.double().cpu() - want).abs().max().item() <= 2e-5 * max(1.0, float(Kn) ** 0.5)
    # zero rows of B left out of the list: same as the dense product (up to the summation order)
    Bz = B.clone()
    keep = torch.zeros(R, dtype=torch.bool)
    keep[rowsB.long()] = True
    Bz[~keep.cuda()] = 0
    Cl = torch.empty(M, N, device='cuda')
    K.gemm_tn_rows(M, N, Kn, A, M, rowsB.cuda(), Bz, N, rowsB.cuda(), Cl, N)
    full = K.matmul_tn(A, Bz)
    assert (Cl - full).abs().max().item() <= 2e-5 * max(1.0, float(Kn) ** 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize('M0,M1,N,R,Kn', [(512, 512, 2048, 6720, 4480),      # 256 tiles of 128 x 64: gemm_tn_direct128_kernel
                                           (128, 384, 4096, 3000, 1024),      # the split inside the first tile row block
                                           (128, 64, 256, 900, 384),          # any other geometry: two products
                                           (512, 512, 2048, 6720, 1000),      # K below the 128 x 64 form's
                                           (128, 0, 256, 900, 384)])          # no second operand
def test_gemm_tn_rows2_two_operands_one_product(K, M0, M1, N, R, Kn):
    """d2p_gemm_f32_tn_rows2: C[:M0] = A0[rows]^T B[rows'], C[M0:] = A1[rows]^T B[rows'] -- the input and the recurrent half of
    an LSTM's kernel gradient as one product on 128 x 64 tiles.  Bit-identical to the two d2p_gemm_f32_tn_rows products on
    64 x 64 tiles (same K partition between the waves, same tree), plain and accumulating; and against fp64."""
    from demo2program_amd.lib import load
    g = torch.Generator().manual_seed(M0 + N + Kn)
    lda0 = M0 + 64                                   # (a row stride that is not the width: x is a view of wider rows)
    A0 = (torch.rand(R, lda0, generator=g) * 2 - 1).cuda()
    A1 = (torch.rand(R, M1, generator=g) * 2 - 1).cuda()
    B = (torch.rand(R, N, generator=g) * 2 - 1).cuda()
    rowsB = (torch.randperm(R - 5, generator=g)[:Kn].sort().values + 5).int().cuda()
    rowsA = rowsB - 5
    C0 = (torch.rand(M0 + M1, N, generator=g) * 2 - 1).cuda()
    ref = torch.cat([A0[:, :M0], A1], 1).double().cpu()[rowsA.long().cpu()].t() @ B.double().cpu()[rowsB.long().cpu()]
    for acc in (False, True):
        got = C0.clone()
        K.gemm_tn_rows2(M0, M1, N, Kn, A0, lda0, A1, max(M1, 1), rowsA, B, N, rowsB, got, N, accumulate=acc)
        want = C0.clone()
        load().d2p_gemm_set_option(256)              # the separate products on 64 x 64 tiles
        try:
            K.gemm_tn_rows(M0, N, Kn, A0, lda0, rowsA, B, N, rowsB, want[:M0], N, accumulate=acc)
            if M1:
                K.gemm_tn_rows(M1, N, Kn, A1, M1, rowsA, B, N, rowsB, want[M0:], N, accumulate=acc)
        finally:
            load().d2p_gemm_set_option(0)
        assert torch.equal(got, want), (acc, float((got - want).abs().max()))
        full = ref + (C0.double().cpu() if acc else 0)
        assert (got.double().cpu() - full).abs().max().item() <= 2e-5 * max(1.0, float(Kn) ** 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize('M,N,R,Kn', [(512, 2048, 6720, 4480),       # 2 x 128 tiles of 128 x 64: one launch
                                       (128, 256, 900, 384)])         # any other geometry: one after the other
def test_gemm_tn_rows_x2_two_products_one_launch(K, M, N, R, Kn):
    """d2p_gemm_f32_tn_rows_x2: C0 = A0[rows]^T B0[rows'], C1 = A1[rows]^T B1[rows'] -- two independent products of one shape
    through one pair of row lists (the action and the perception decoder's recurrent kernel-gradient halves) in ONE launch of
    the 128 x 64-tile kernel.  Bit-identical to two d2p_gemm_f32_tn_rows products on 64 x 64 tiles, plain and accumulating."""
    from demo2program_amd.lib import load
    g = torch.Generator().manual_seed(M + N + Kn + 1)
    A = [(torch.rand(R, M, generator=g) * 2 - 1).cuda() for _ in range(2)]
    B = [(torch.rand(R, N, generator=g) * 2 - 1).cuda() for _ in range(2)]
    rowsB = (torch.randperm(R - 5, generator=g)[:Kn].sort().values + 5).int().cuda()
    rowsA = rowsB - 5
    C0 = [(torch.rand(M, N, generator=g) * 2 - 1).cuda() for _ in range(2)]
    for acc in (False, True):
        got = [c.clone() for c in C0]
        K.gemm_tn_rows_x2(M, N, Kn, A[0], M, B[0], N, got[0], A[1], M, B[1], N, got[1], N, rowsA, rowsB, accumulate=acc)
        want = [c.clone() for c in C0]
        load().d2p_gemm_set_option(256)
        try:
            for i in range(2):
                K.gemm_tn_rows(M, N, Kn, A[i], M, rowsA, B[i], N, rowsB, want[i], N, accumulate=acc)
        finally:
            load().d2p_gemm_set_option(0)
        for i in range(2):
            assert torch.equal(got[i], want[i]), (acc, i, float((got[i] - want[i]).abs().max()))
            ref = A[i].double().cpu()[rowsA.long().cpu()].t() @ B[i].double().cpu()[rowsB.long().cpu()]
            full = ref + (C0[i].double().cpu() if acc else 0)
            assert (got[i].double().cpu() - full).abs().max().item() <= 2e-5 * max(1.0, float(Kn) ** 0.5)


@pytest.mark.gpu
def test_loss_backward_of_three_decoders_and_their_projection_in_one_launch(K):
    """d2p_xent_bwd_dhout_multi: dlogits of a softmax ('bvl' labels), a grouped softmax and a grouped sigmoid loss
    AND dhout = dlogits . proj^T of each, in one launch -- dlogits equal to the per-loss kernels' (1e-7), dhout
    against the fp64 product; ragged row counts, n_steps below T."""
    g = torch.Generator().manual_seed(41)
    U = 96
    probs, refs = [], []
    for (mode, kind, T, R, V, G) in (('softmax', 'bvl', 9, 5, 50, 1), ('softmax', 'rtv', 7, 12, 6, 3),
                                     ('sigmoid', 'rtv', 7, 12, 5, 3)):
        lens = torch.randint(1, T, (R,), generator=g)
        n_steps = int(lens.max())
        logits = (torch.rand(T, R, V, generator=g) * 6 - 3).cuda()
        if mode == 'softmax':
            lab = F.one_hot(torch.randint(0, V, (R, T), generator=g), V).float()
        else:
            lab = torch.randint(0, 2, (R, T, V), generator=g).float()
        lab = lab * (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1)
        labels = (lab.permute(0, 2, 1) if kind == 'bvl' else lab).contiguous().cuda()       # [B,V,L] or [R,T,V]
        den = (torch.rand(G, generator=g) * 5 + 1).cuda()
        proj = (torch.rand(U, V, generator=g) * 2 - 1).cuda()
        dl = torch.full((T * R, V), 9.0, device='cuda')
        dh = torch.full((T * R, U), 9.0, device='cuda')
        probs.append(dict(mode=mode, logits=logits, labels=labels, lab_kind=kind, lens=lens.int().cuda(), T=T, R=R, V=V,
                          G=G, n_steps=n_steps, den=den, scale=0.7, dlogits=dl, proj=proj, dhout=dh, U=U))
        ref = torch.zeros(T * R, V, device='cuda')
        K.xent_bwd(mode, logits, labels, kind, lens.int().cuda(), T, R, V, G, n_steps, den, 0.7, ref)
        refs.append((ref, n_steps * R))
    K.xent_bwd_dhout_multi(probs)
    for q, (ref, rows) in zip(probs, refs):
        assert (q['dlogits'][:rows] - ref[:rows]).abs().max().item() <= 1e-7, q['mode']
        want = ref[:rows].double().cpu() @ q['proj'].double().cpu().t()
        assert (q['dhout'][:rows].double().cpu() - want).abs().max().item() <= 1e-5
        if rows < q['dhout'].shape[0]:                              # rows past n_steps: untouched
            assert float(q['dhout'][rows:].min()) == 9.0 and float(q['dlogits'][rows:].min()) == 9.0


@pytest.mark.parametrize('U', [128, 512])
def test_logits_computed_inside_the_loss_backward_launch(K, U):
    """d2p_xent_bwd_desc.hout / .logits_out (round 4): the launch first computes logits = hout . proj (V = 50, 6, 5, 16
    columns: every lane layout of the in-launch product -- one column group, K split across lane groups) and then
    differentiates them.  logits against the fp64 product, dlogits / dhout against the same launch fed with those
    logits (bit-identical), rows past n_steps untouched."""
    g = torch.Generator().manual_seed(43 + U)
    probs, fed = [], []
    for (mode, kind, T, R, V, G) in (('softmax', 'bvl', 9, 5, 50, 1), ('softmax', 'rtv', 7, 12, 6, 3),
                                     ('sigmoid', 'rtv', 7, 12, 5, 3)):
        lens = torch.randint(1, T, (R,), generator=g)
        n_steps = int(lens.max())
        hout = (torch.rand(T, R, U, generator=g) * 2 - 1).cuda()
        if mode == 'softmax':
            lab = F.one_hot(torch.randint(0, V, (R, T), generator=g), V).float()
        else:
            lab = torch.randint(0, 2, (R, T, V), generator=g).float()
        lab = lab * (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1)
        labels = (lab.permute(0, 2, 1) if kind == 'bvl' else lab).contiguous().cuda()
        den = (torch.rand(G, generator=g) * 5 + 1).cuda()
        proj = ((torch.rand(U, V, generator=g) * 2 - 1) * 0.3).cuda()
        base = dict(mode=mode, labels=labels, lab_kind=kind, lens=lens.int().cuda(), T=T, R=R, V=V, G=G, n_steps=n_steps,
                    den=den, scale=0.7, proj=proj, U=U)
        probs.append(dict(base, logits=torch.full((T, R, V), 9.0, device='cuda'), hout=hout,
                          dlogits=torch.full((T * R, V), 9.0, device='cuda'), dhout=torch.full((T * R, U), 9.0, device='cuda')))
        fed.append(dict(base, dlogits=torch.full((T * R, V), 9.0, device='cuda'),
                        dhout=torch.full((T * R, U), 9.0, device='cuda')))
    K.xent_bwd_dhout_multi(probs)
    for q, f in zip(probs, fed):
        rows = q['n_steps'] * q['R']
        lg = q['logits'].view(-1, q['V'])
        want = q['hout'].view(-1, U)[:rows].double().cpu() @ q['proj'].double().cpu()
        assert (lg[:rows].double().cpu() - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item()), q['V']
        if rows < lg.shape[0]:
            assert float(lg[rows:].min()) == 9.0
        f['logits'] = q['logits']
    K.xent_bwd_dhout_multi(fed)
    for q, f in zip(probs, fed):
        assert torch.equal(q['dlogits'], f['dlogits']) and torch.equal(q['dhout'], f['dhout'])


# ------------------------------------------------------------------ wide-tile persistent forward kernel (round 4)
def _wide_seq(M, T, U, masked, init, seed, lo=0):
    g = torch.Generator().manual_seed(seed)
    q = dict(M=M, U=U, n_steps=T, z0=((torch.rand(T * M, 4 * U, generator=g) - 0.5) * 2).cuda(),
             Wh=((torch.rand(U, 4 * U, generator=g) - 0.5) * 0.2).cuda(),
             hout=torch.empty(T, M, U, device='cuda'), cs=torch.empty(T, M, U, device='cuda'),
             h_final=torch.empty(M, U, device='cuda'), c_final=torch.empty(M, U, device='cuda'))
    q['z'] = q['z0'].clone()
    if init:
        q['h0'], q['c0'] = (torch.rand(M, U, generator=g) - 0.5).cuda(), (torch.rand(M, U, generator=g) - 0.5).cuda()
    if masked:
        lens = torch.randint(lo, T + 1, (M,), generator=g)
        lens[0] = T
        q['lens_host'], q['lens'] = lens.numpy().astype(np.int64), lens.to(torch.int32).cuda()
    return q


def _wide_run(K, seqs, sort):
    for q in seqs:
        q['z'].copy_(q['z0'])
        for n in ('hout', 'cs', 'h_final', 'c_final'):
            q[n].fill_(float('nan'))
        q.pop('row_order', None)
        if sort and q.get('lens') is not None:
            q['row_order'] = K.lstm_row_order(q['lens_host'])
    K.lstm_seq_fwd_multi(seqs)
    torch.cuda.synchronize()
    return [[q[n].clone() for n in ('z', 'hout', 'cs', 'h_final', 'c_final')] for q in seqs]


def _wide_check(K, seqs, sort=False, must_take=True, reps=2):
    from demo2program_amd.lib import load
    lib = load()
    K.set_lstm_persistent(False)
    try:
        refs = [_wide_run(K, [q], False)[0] for q in seqs]
    finally:
        K.set_lstm_persistent(True)
    K.lstm_persist_error(True)
    before = [lib.d2p_lstm_persist_wide_launches(i) for i in range(4)]
    for _ in range(reps):
        got = _wide_run(K, seqs, sort)
        assert K.lstm_persist_error() == 0
        for g_, r_ in zip(got, refs):
            for name, a, b in zip(('z', 'hout', 'cs', 'h_final', 'c_final'), g_, r_):
                assert torch.equal(a, b), name
    after = [lib.d2p_lstm_persist_wide_launches(i) for i in range(4)]
    if must_take:
        assert after[len(seqs)] - before[len(seqs)] == reps
        if sort:
            assert after[0] - before[0] == reps


@pytest.mark.parametrize('xcd_local', [1, 0])
@pytest.mark.parametrize('M,U,T,masked,init', [(12, 64, 6, 1, 1), (35, 128, 4, 1, 1), (80, 256, 5, 0, 1), (48, 512, 4, 1, 0),
                                               (320, 512, 20, 1, 1), (320, 512, 20, 0, 0), (32, 512, 40, 0, 1),
                                               (333, 512, 9, 1, 1), (1000, 512, 5, 1, 1)])
def test_lstm_wide_forward_equals_per_step(K, M, U, T, masked, init, xcd_local):
    """The wide-tile persistent forward kernel (16 units per column tile; d2p_lstm_seq_fwd_multi with a flag buffer)
    against the one-launch-per-step kernels: z, hout, cs and the final states BIT-IDENTICAL -- same K split, same
    summation order -- with write-through hand-offs and with the L2-local ones of domains found on one XCD, unsorted
    and (with lengths) sorted by length: the steps a sorted domain skips are filled in as the masked steps would have
    written them (zeros in hout, the carried cell state in cs)."""
    from demo2program_amd.lib import call
    call.d2p_lstm_persist_set_fwd_wide(1, 0, 0, xcd_local)
    try:
        _wide_check(K, [_wide_seq(M, T, U, masked, init, seed=M + T)])
        if masked:
            _wide_check(K, [_wide_seq(M, T, U, masked, init, seed=M + T)], sort=True)
    finally:
        call.d2p_lstm_persist_set_fwd_wide(1, 0, 0, 1)


def test_lstm_wide_forward_sorted_with_zero_length_rows_and_training_lengths(K):
    q = _wide_seq(320, 20, 512, True, True, seed=5, lo=8)              # lengths 8..20, as the training batches
    _wide_check(K, [q], sort=True)
    q = _wide_seq(320, 20, 512, True, False, seed=6)
    q['lens_host'][5:40] = 0
    q['lens'] = torch.from_numpy(q['lens_host'].astype(np.int32)).cuda()
    _wide_check(K, [q], sort=True)
    q = _wide_seq(320, 20, 512, True, True, seed=7)                    # the longest row shorter than the step count
    q['lens_host'] = np.minimum(q['lens_host'], 13)
    q['lens'] = torch.from_numpy(q['lens_host'].astype(np.int32)).cuda()
    _wide_check(K, [q], sort=True)


@pytest.mark.parametrize('la_from,defer_from', [(2, 3), (3, 5), (4, 8)])
def test_lstm_three_forward_sequences_in_one_wide_launch(K, la_from, defer_from):
    """d2p_lstm_seq_fwd_multi with the three decoders (2 x 320 rows x 20 steps, 32 rows x 50 steps): one launch of the
    wide-tile kernel with 3 + 3 + 2 row domains; bit-identical to the per-step kernels in every hand-off form
    (own-row polling, look-ahead from the middle of the chain, deferred gate math)."""
    from demo2program_amd.lib import call
    call.d2p_lstm_persist_set_fwd_wide(1, la_from, defer_from, 1)
    try:
        _wide_check(K, [_wide_seq(320, 20, 512, False, True, 11), _wide_seq(320, 20, 512, False, True, 12),
                        _wide_seq(32, 50, 512, False, True, 13)])
        _wide_check(K, [_wide_seq(320, 20, 512, False, True, 11), _wide_seq(32, 50, 512, False, True, 13)])
        _wide_check(K, [_wide_seq(320, 20, 512, True, True, 14, lo=8), _wide_seq(100, 7, 512, False, True, 15),
                        _wide_seq(32, 50, 512, False, True, 13)], sort=True)
        # 400 rows = 25 sub-tiles: no split of the 8 row domains holds two of them and a third sequence -- falls back
        _wide_check(K, [_wide_seq(400, 20, 512, False, True, 16), _wide_seq(400, 20, 512, False, True, 17),
                        _wide_seq(16, 32, 512, False, True, 18)], must_take=False)
    finally:
        call.d2p_lstm_persist_set_fwd_wide(1, 3, 5, 1)


def test_lstm_wide_forward_long_sequence(K):
    """thousands of hand-offs per workgroup, sorted and not: rare stale reads would surface here"""
    _wide_check(K, [_wide_seq(320, 160, 512, True, True, 21)], reps=2)
    _wide_check(K, [_wide_seq(320, 160, 512, True, True, 21)], sort=True, reps=2)


# ---------------------------------------------------------------- the Karel State_Encoder forward in one launch
def _encoder_chain(K, x, B, G, T, prm):
    """the separate launches: conv -> bn(train) three times, then the time-major transpose"""
    NF = B * G * T
    out = dict(a=[], y=[], mean=[], rstd=[], var=[])
    cur = x
    for l, (cout, hw) in enumerate(((16, 4), (32, 2), (48, 1))):
        a = K.conv_fwd(cur, prm['w'][l], prm['b'][l], act=1)
        var = torch.empty(G, cout, device='cuda')
        y, mean, rstd, _ = K.bn_fwd(a.view(NF * hw * hw, cout), prm['gamma'][l], prm['beta'][l], G, T * hw * hw, var=var)
        out['a'].append(a); out['y'].append(y); out['mean'].append(mean); out['rstd'].append(rstd); out['var'].append(var)
        cur = y.view(NF, hw, hw, cout)
    out['feats_tm'] = out['y'][2].view(B * G, T, 48).transpose(0, 1).contiguous()
    return out


@pytest.mark.parametrize('B,G,T,u8', [(32, 10, 20, True), (32, 10, 20, False), (5, 3, 8, True), (1, 1, 4, False),
                                     (40, 10, 20, True), (7, 32, 12, True), (64, 10, 8, False)])
def test_karel_encoder_one_launch_matches_the_separate_launches(K, B, G, T, u8):
    """d2p_karel_encoder_fwd against conv -> batch norm x 3 (models/model_full.py:362-381 of the reference): the
    same values to fp32 summation order, and the statistics / normalised outputs consistent with the activations
    the launch itself wrote (checked in fp64 on the host)."""
    if not K.karel_encoder_ok(B, G, T):
        pytest.skip('geometry not taken by the one-launch kernel on this device')
    g = torch.Generator().manual_seed(B * 1000 + G * 10 + T)
    NF = B * G * T
    if u8:
        x = (torch.rand(NF, 8, 8, 16, generator=g) < 0.15).to(torch.uint8).cuda()
    else:
        x = torch.randn(NF, 8, 8, 16, generator=g).cuda()
    prm = dict(w=[], b=[], gamma=[], beta=[])
    for cin, cout in ((16, 16), (16, 32), (32, 48)):
        prm['w'].append((torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda())
        prm['b'].append((torch.randn(cout, generator=g) * 0.1).cuda())
        prm['gamma'].append((1 + 0.2 * torch.randn(cout, generator=g)).cuda())
        prm['beta'].append((0.1 * torch.randn(cout, generator=g)).cuda())
    ref = _encoder_chain(K, x, B, G, T, prm)
    nan = float('nan')
    a = [torch.full((NF, 4, 4, 16), nan, device='cuda'), torch.full((NF, 2, 2, 32), nan, device='cuda'),
         torch.full((NF, 1, 1, 48), nan, device='cuda')]
    y = [torch.full((NF * 16, 16), nan, device='cuda'), torch.full((NF * 4, 32), nan, device='cuda')]
    stat = lambda: [torch.full((G, c), nan, device='cuda') for c in (16, 32, 48)]       # noqa: E731
    mean, rstd, var = stat(), stat(), stat()
    feats_tm = torch.full((T, B * G, 48), nan, device='cuda')
    ws = torch.empty(K._load_lib().d2p_karel_encoder_ws_bytes(B, G, T), dtype=torch.uint8, device='cuda')
    for rep in range(3):                                      # (the arrival counters of three slots)
        K.karel_encoder_fwd(x, B, G, T, prm['w'], prm['b'], prm['gamma'], prm['beta'], a, y, feats_tm, mean, rstd,
                            var, ws)
        torch.cuda.synchronize()
        assert K.lstm_persist_error() == 0
        for l in range(3):
            # (the separate launches pick their conv kernel by shape -- not the same summation order everywhere)
            torch.testing.assert_close(a[l], ref['a'][l], rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(mean[l], ref['mean'][l], rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(var[l], ref['var'][l], rtol=1e-4, atol=1e-6)
            torch.testing.assert_close(rstd[l], ref['rstd'][l], rtol=1e-4, atol=0)
        for l in range(2):
            torch.testing.assert_close(y[l], ref['y'][l], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(feats_tm, ref['feats_tm'], rtol=1e-4, atol=1e-4)
        # the statistics are those of the activations this launch wrote (fp64 on the host)
        for l, hw in enumerate((4, 2, 1)):
            c = a[l].shape[-1]
            v = a[l].double().view(B, G, T * hw * hw, c).permute(1, 0, 2, 3).reshape(G, -1, c)
            torch.testing.assert_close(mean[l].double(), v.mean(1), rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(var[l].double(), v.var(1, unbiased=False), rtol=1e-5, atol=1e-8)
            yl = (prm['gamma'][l].double() * (v - mean[l].double()[:, None]) * rstd[l].double()[:, None]
                  + prm['beta'][l].double()).view(G, B, T * hw * hw, c).permute(1, 0, 2, 3)
            got = y[l].view(B, G, T * hw * hw, c) if l < 2 else feats_tm.transpose(0, 1).reshape(B, G, T, c)
            torch.testing.assert_close(got.double(), yl, rtol=1e-5, atol=1e-5)
        feats_tm.fill_(nan)


@pytest.mark.parametrize('B,G,T,u8', [(32, 10, 20, True), (32, 10, 20, False), (5, 3, 8, True), (1, 1, 4, False),
                                     (40, 10, 20, True), (7, 32, 12, True), (64, 10, 8, False)])
def test_karel_encoder_backward_one_launch_matches_the_separate_launches(K, B, G, T, u8):
    """d2p_karel_encoder_bwd against the chain it replaces (transpose, then per layer d2p_bn_group_bwd, the conv's weight
    gradient and input gradient: models/model_full.py:216-231 / models/ops.py:14-33 under tf.gradients), on the
    activations and statistics the one-launch forward wrote; and against torch autograd in fp64 on the host for the
    full-size case.  Two runs bit-identical (every sum has a fixed order)."""
    if not (K.karel_encoder_ok(B, G, T) and K.karel_encoder_bwd_ok(B, G, T)):
        pytest.skip('geometry not taken by the one-launch kernels on this device')
    g = torch.Generator().manual_seed(B * 1000 + G * 10 + T + 7)
    NF = B * G * T
    if u8:
        x = (torch.rand(NF, 8, 8, 16, generator=g) < 0.15).to(torch.uint8).cuda()
    else:
        x = torch.randn(NF, 8, 8, 16, generator=g).cuda()
    prm = dict(w=[], b=[], gamma=[], beta=[])
    for cin, cout in ((16, 16), (16, 32), (32, 48)):
        prm['w'].append((torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5).cuda())
        prm['b'].append((torch.randn(cout, generator=g) * 0.1).cuda())
        prm['gamma'].append((1 + 0.2 * torch.randn(cout, generator=g)).cuda())
        prm['beta'].append((0.1 * torch.randn(cout, generator=g)).cuda())
    a = [torch.empty(NF, 4, 4, 16, device='cuda'), torch.empty(NF, 2, 2, 32, device='cuda'), torch.empty(NF, 1, 1, 48, device='cuda')]
    y = [torch.empty(NF * 16, 16, device='cuda'), torch.empty(NF * 4, 32, device='cuda')]
    stat = lambda: [torch.empty(G, c, device='cuda') for c in (16, 32, 48)]       # noqa: E731
    mean, rstd, var = stat(), stat(), stat()
    feats_tm = torch.empty(T, B * G, 48, device='cuda')
    ws = torch.empty(K._load_lib().d2p_karel_encoder_ws_bytes(B, G, T), dtype=torch.uint8, device='cuda')
    K.karel_encoder_fwd(x, B, G, T, prm['w'], prm['b'], prm['gamma'], prm['beta'], a, y, feats_tm, mean, rstd, var, ws)
    dfeat_tm = torch.randn(T, B * G, 48, generator=g).cuda()
    dfeat_tm[T - 1, ::3] = 0                                  # (rows past a demonstration's length carry no gradient)

    # the separate launches
    ref = dict(dw=[None] * 3, db=[None] * 3, dgamma=[None] * 3, dbeta=[None] * 3)
    dy = dfeat_tm.transpose(0, 1).contiguous().view(NF, 48)
    xin = [x, y[0].view(NF, 4, 4, 16), y[1].view(NF, 2, 2, 32)]
    for l, (cin, cout, hw, hin) in reversed(list(enumerate(((16, 16, 4, 8), (16, 32, 2, 4), (32, 48, 1, 2))))):
        ref['dgamma'][l], ref['dbeta'][l], ref['db'][l] = (torch.empty(cout, device='cuda') for _ in range(3))
        da = K.bn_bwd(a[l].view(NF * hw * hw, cout), dy.view(NF * hw * hw, cout), prm['gamma'][l], mean[l], rstd[l], G,
                      T * hw * hw, True, ref['dgamma'][l], ref['dbeta'][l], dbias=ref['db'][l])
        ref['dw'][l] = K.conv_wgrad(xin[l], da.view(NF, hw, hw, cout), torch.empty(3, 3, cin, cout, device='cuda'))
        if l > 0:
            dy = K.conv_dgrad(da.view(NF, hw, hw, cout), prm['w'][l], (NF, hin, hin, cin))

    nan = float('nan')
    out = {k: [torch.full_like(t, nan) for t in ref[k]] for k in ref}
    wsb = torch.empty(K._load_lib().d2p_karel_encoder_bwd_ws_bytes(B, G, T), dtype=torch.uint8, device='cuda')
    first = None
    for rep in range(3):
        K.karel_encoder_bwd(x, dfeat_tm, B, G, T, prm['w'], prm['gamma'], prm['beta'], a, mean, rstd, out['dw'], out['db'],
                            out['dgamma'], out['dbeta'], wsb)
        torch.cuda.synchronize()
        assert K.lstm_persist_error() == 0
        for k in ref:
            for l in range(3):
                scale = float(ref[k][l].abs().max()) + 1e-6
                err = float((out[k][l] - ref[k][l]).abs().max())
                assert err <= 2e-5 * scale + 1e-6, (k, l, err, scale)
        snap = [t.clone() for k in sorted(out) for t in out[k]]
        if first is None:
            first = snap
        else:
            assert all(torch.equal(u, v) for u, v in zip(first, snap))
        for k in out:
            for t in out[k]:
                t.fill_(nan)

    if (B, G, T) == (32, 10, 20):
        # second opinion: torch autograd in fp64 on the host, through conv -> +bias -> lrelu -> batch norm per index
        import torch.nn.functional as F
        P = {k: [t.double().cpu().requires_grad_(True) for t in prm[k]] for k in prm}
        cur = x.double().cpu().permute(0, 3, 1, 2)
        for l, hin in enumerate((8, 4, 2)):
            wt = P['w'][l].permute(3, 2, 0, 1)
            cur = F.conv2d(F.pad(cur, (0, 1, 0, 1)), wt, P['b'][l], stride=2)
            cur = 0.6 * cur + 0.4 * cur.abs()
            c, hw = cur.shape[1], cur.shape[2]
            v = cur.view(B, G, T, c, hw, hw)
            mu = v.mean(dim=(0, 2, 4, 5), keepdim=True)
            va = v.var(dim=(0, 2, 4, 5), unbiased=False, keepdim=True)
            v = (v - mu) / torch.sqrt(va + 1e-3) * P['gamma'][l].view(1, 1, 1, c, 1, 1) + P['beta'][l].view(1, 1, 1, c, 1, 1)
            cur = v.view(NF, c, hw, hw)
        feats = cur.view(B * G, T, 48).transpose(0, 1)
        feats.backward(dfeat_tm.double().cpu())
        K.karel_encoder_bwd(x, dfeat_tm, B, G, T, prm['w'], prm['gamma'], prm['beta'], a, mean, rstd, out['dw'], out['db'],
                            out['dgamma'], out['dbeta'], wsb)
        torch.cuda.synchronize()
        for k, pk in (('dw', 'w'), ('db', 'b'), ('dgamma', 'gamma'), ('dbeta', 'beta')):
            for l in range(3):
                r = P[pk][l].grad
                err = float((out[k][l].double().cpu() - r).abs().max())
                assert err <= 2e-4 * float(r.abs().max()) + 1e-6, (k, l, err)


@pytest.mark.parametrize('n,rows,E', [(6400, 9, 2048), (1568, 53, 2048), (6400, 32, 512), (1000, 31, 68), (257, 3, 64)])
def test_rows_summed_by_key_equal_index_add(K, n, rows, E):
    """d2p_embedding_scatter_add_oob0 on rows_by_key_kernel (n >= 256): out[v] = sum of the rows whose id is v, ids
    outside the table dropped (the gather reads zeros there: models/model_full.py:294) -- against
    index_add_ in fp64; two runs bit-identical (every sum has a fixed order)."""
    g = torch.Generator().manual_seed(n + rows + E)
    ids = torch.randint(0, rows + 2, (n,), generator=g, dtype=torch.int32)      # rows, rows + 1: outside the table
    ids[::7] = -1
    x = torch.randn(n, E, generator=g)
    out = torch.full((rows, E), float('nan'), device='cuda')
    K.embedding_scatter_add(ids.cuda(), x.cuda(), out)
    ok = (ids >= 0) & (ids < rows)
    ref = torch.zeros(rows, E, dtype=torch.float64).index_add_(0, ids[ok].long(), x[ok].double())
    torch.testing.assert_close(out.double().cpu(), ref, rtol=1e-5, atol=2e-4)
    out2 = torch.full((rows, E), float('nan'), device='cuda')
    K.embedding_scatter_add(ids.cuda(), x.cuda(), out2)
    assert torch.equal(out, out2)


def test_per_fc_bn_statistics_from_the_gram_matrix(K):
    """d2p_per_fc_bn_stats: mean / rstd / var per demonstration index of u = per . W + b (the Per_Encoder's fc in front of
    its batch norm, models/model_full.py:383-398) from gram = A^T A alone, against the statistics of the rows themselves
    in fp64."""
    G, P, U, T, B = 10, 5, 512, 20, 32
    g = torch.Generator().manual_seed(11)
    per = torch.rand(T, B, G, P, generator=g)                       # rows t*M + b*k + i
    W, b = torch.randn(P, U, generator=g) * 0.5, torch.randn(U, generator=g) * 0.1
    NCp = 64
    A = torch.zeros(T * B * G, NCp)
    A4 = A.view(T, B, G, NCp)
    for i in range(G):
        A4[:, :, i, i * (P + 1):i * (P + 1) + P] = per[:, :, i]
        A4[:, :, i, i * (P + 1) + P] = 1.0
    gram = (A.double().t() @ A.double()).float().cuda()
    mean, rstd, var = (torch.empty(G, U, device='cuda') for _ in range(3))
    K.per_fc_bn_stats(G, P, T * B, W.cuda(), b.cuda(), gram, mean, rstd, var)
    u = per.double() @ W.double() + b.double()                      # [T, B, G, U]
    rows = u.permute(2, 0, 1, 3).reshape(G, T * B, U)
    torch.testing.assert_close(mean.double().cpu(), rows.mean(1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(var.double().cpu(), rows.var(1, unbiased=False), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rstd.double().cpu(), 1.0 / torch.sqrt(rows.var(1, unbiased=False) + 1e-3), rtol=1e-4, atol=0)


@pytest.mark.parametrize('G,P,T,B,E', [(10, 5, 20, 32, 2048), (3, 2, 7, 4, 256), (15, 8, 4, 2, 64)])
def test_per_rows_transposed_product_from_the_structure_of_the_rows(K, G, P, T, B, E):
    """d2p_per_rows_tn: S = A^T dz with A built as derive_per_rows builds it (per values in the columns of the row's
    demonstration index, a 1 behind them) against the fp64 product; pad rows of S zero; two runs bit-identical."""
    g = torch.Generator().manual_seed(G * 100 + P)
    M = B * G
    per = torch.rand(T, M, P, generator=g)
    dz = torch.randn(T * M, E, generator=g)
    NCp = (G * (P + 1) + 3) // 4 * 4 + 4
    A = torch.zeros(T * M, NCp, dtype=torch.float64)
    for i in range(G):
        rows = torch.arange(T * M)[torch.arange(T * M) % G == i]
        A[rows, i * (P + 1):i * (P + 1) + P] = per.view(T * M, P)[rows].double()
        A[rows, i * (P + 1) + P] = 1.0
    n = (T - 1) * M if T > 4 else T * M                      # (the decoded steps only)
    ref = A[:n].t() @ dz[:n].double()
    S = torch.full((NCp, E), float('nan'), device='cuda')
    K.per_rows_tn(G, per.view(T * M, P).cuda(), dz.cuda(), S, n)
    torch.testing.assert_close(S.double().cpu(), ref, rtol=1e-5, atol=1e-4)
    assert float(S[G * (P + 1):].abs().max()) == 0.0
    S2 = torch.full((NCp, E), float('nan'), device='cuda')
    K.per_rows_tn(G, per.view(T * M, P).cuda(), dz.cuda(), S2, n)
    assert torch.equal(S, S2)


@pytest.mark.parametrize('G,P,T,B,E', [(10, 5, 20, 32, 2048), (3, 2, 7, 4, 260), (25, 8, 4, 2, 64)])
def test_per_rows_product_from_the_structure_of_the_rows(K, G, P, T, B, E):
    """d2p_per_rows_nn: z = A . HWx + bias with A as derive_per_rows builds it, against the fp64 product; rows past the
    decoded steps untouched."""
    g = torch.Generator().manual_seed(G * 10 + P)
    M = B * G
    per = torch.rand(T, M, P, generator=g)
    NCp = (G * (P + 1) + 3) // 4 * 4 + 4
    HWx, bias = torch.randn(NCp, E, generator=g), torch.randn(E, generator=g)
    A = torch.zeros(T * M, NCp, dtype=torch.float64)
    for i in range(G):
        rows = torch.arange(T * M)[torch.arange(T * M) % G == i]
        A[rows, i * (P + 1):i * (P + 1) + P] = per.view(T * M, P)[rows].double()
        A[rows, i * (P + 1) + P] = 1.0
    n = (T - 1) * M
    z = torch.full((T * M, E), 7.0, device='cuda')
    K.per_rows_nn(G, per.view(T * M, P).cuda(), HWx.cuda(), bias.cuda(), z, n)
    ref = A[:n] @ HWx.double() + bias.double()
    torch.testing.assert_close(z[:n].double().cpu(), ref, rtol=1e-5, atol=1e-5)
    assert bool((z[n:] == 7.0).all())


# ------------------------------------------------------------------ the decoders' small gradient products (round 6)
@pytest.mark.parametrize('U', [128, 512])
def test_small_pair_products_of_several_decoders_in_one_launch(K, U):
    """d2p_small_pair_products: G1 = A^T S (the input half of the LSTM kernel's gradient from the dz rows summed by input
    token / perception column) and G2 = S Wx^T (the embedding gradient, models/model_full.py:282-296; the perception
    encoder's Q) for R = 7 / 51 / 60 / 176 rows -- against fp64; rows of S past R are never read into the result."""
    probs, refs = [], []
    for i, R in enumerate([7, 51, 60, 176]):
        S, A, Wx = rnd(R + 3, 4 * U, seed=120 + i), rnd(R, U, seed=130 + i), rnd(U, 4 * U, seed=140 + i, scale=0.1)
        S[R:] = float('nan')                                # (buffers carry spare rows: tok + 2, padded perception columns)
        G1, G2 = torch.full((U, 4 * U), 3.0, device='cuda'), torch.full((R + 2, U), 5.0, device='cuda')
        probs.append((R, U, dev(S), dev(A), dev(Wx), G1, G2))
        refs.append((A.double().t() @ S[:R].double(), S[:R].double() @ Wx.double().t()))
    K.small_pair_products(probs)
    for (R, _, _, _, _, G1, G2), (r1, r2) in zip(probs, refs):
        close(G1, r1, atol=2e-5 * float(r1.abs().max()) + 1e-5, rtol=1e-5)
        close(G2[:R], r2, atol=2e-5 * float(r2.abs().max()) + 1e-5, rtol=1e-5)
        assert (G2[R:] == 5.0).all()
