"""Tensor-level wrappers over the C ABI (include/d2p.h).

torch is used here only for device memory (``torch.empty``), the current stream and
views; every arithmetic operation is a call into libd2p_hip.so.  All tensors must be
CUDA (ROCm) fp32 / int32 and contiguous unless a stride is passed explicitly.
"""
import numpy as np
import torch

from .lib import call, current_stream, load as _load_lib, ptr


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('demo2program_amd kernels need CUDA/ROCm tensors; got a %s tensor. '
                               'There is no CPU fallback.' % t.device)


class Scratch(object):
    """One growing device scratch buffer.  Every kernel is launched on the same stream, so
    successive ops may reuse it (stream order serialises their accesses).

    A buffer that has been outgrown is RETIRED, not freed: a captured hipGraph keeps replaying
    with the workspace pointers it recorded, so that memory must stay allocated (and must not be
    handed to another tensor) for as long as the process lives.  Growth is geometric, so the
    retired buffers add up to less than the live one."""

    def __init__(self):
        self.buf = None
        self.retired = []

    def get(self, nbytes):
        nbytes = int(nbytes)
        if nbytes == 0:
            return None, 0
        if self.buf is None or self.buf.numel() < nbytes:
            if self.buf is not None:
                self.retired.append(self.buf)
                nbytes = max(nbytes, 2 * self.buf.numel())
            self.buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device='cuda')
        return self.buf.data_ptr(), self.buf.numel()

    def reserve(self, nbytes):
        self.get(nbytes)


class _ScratchByStream(object):
    """A Scratch per stream: ops issued on different streams may run concurrently and must
    not share scratch memory (ops on one stream are serialised by stream order)."""

    def __init__(self):
        self.by_stream = {}

    def _cur(self):
        key = torch.cuda.current_stream().cuda_stream
        s = self.by_stream.get(key)
        if s is None:
            s = self.by_stream[key] = Scratch()
        return s

    def get(self, nbytes):
        return self._cur().get(nbytes)

    def reserve(self, nbytes):
        self._cur().reserve(nbytes)


SCRATCH = _ScratchByStream()


# ---------------------------------------------------------------- GEMM
def gemm_raw(kind, M, N, K, A, lda, B, ldb, C, ldc, bias=None, act=0, accumulate=False,
             allow_split=True):
    """kind in {'nn','nt','tn'}; A/B/C are data pointers (ints) or tensors."""
    ws, wsb = (None, 0)
    if allow_split:
        ws, wsb = SCRATCH.get(call.d2p_gemm_ws_bytes(M, N, K))
    fn = getattr(call, 'd2p_gemm_f32_' + kind)
    fn(M, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc, _p(bias), act, 1 if accumulate else 0,
       ws, wsb, current_stream())


def gemm_rows(kind, n_rows, N, K, A, lda, B, ldb, C, ldc, rows, bias=None):
    """C[rows[x]] = A[rows[x]] . B (kind 'nn', B [K, N]) or A[rows[x]] . B^T ('nt', B [N, K]) for x < n_rows."""
    ws, wsb = SCRATCH.get(call.d2p_gemm_ws_bytes(n_rows, N, K))
    call.d2p_gemm_f32_rows({'nn': 0, 'nt': 1}[kind], n_rows, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc, _p(bias),
                           _p(rows), ws, wsb, current_stream())


def gemm_tn_rows(M, N, K, A, lda, rowsA, B, ldb, rowsB, C, ldc, accumulate=False):
    """C[M, N] (+)= sum_{x < K} A[rowsA[x], :M]^T B[rowsB[x], :N] (both operands through lists of K row indices)."""
    ws, wsb = SCRATCH.get(call.d2p_gemm_ws_bytes(M, N, K))
    call.d2p_gemm_f32_tn_rows(M, N, K, _p(A), lda, _p(rowsA), _p(B), ldb, _p(rowsB), _p(C), ldc,
                              1 if accumulate else 0, ws, wsb, current_stream())


def gemm_tn_rows2(M0, M1, N, K, A0, lda0, A1, lda1, rowsA, B, ldb, rowsB, C, ldc, accumulate=False):
    """C[:M0] (+)= A0[rowsA]^T B[rowsB], C[M0:M0+M1] (+)= A1[rowsA]^T B[rowsB]: the two halves of an LSTM's kernel gradient
    as one product (d2p_gemm_f32_tn_rows2)."""
    ws, wsb = SCRATCH.get(max(call.d2p_gemm_ws_bytes(M0 + M1, N, K), call.d2p_gemm_ws_bytes(max(M0, M1), N, K)))
    call.d2p_gemm_f32_tn_rows2(M0, M1, N, K, _p(A0), lda0, _p(A1), lda1, _p(rowsA), _p(B), ldb, _p(rowsB), _p(C), ldc,
                               1 if accumulate else 0, ws, wsb, current_stream())


def gemm_tn_rows_x2(M, N, K, A0, lda0, B0, ldb0, C0, A1, lda1, B1, ldb1, C1, ldc, rowsA, rowsB, accumulate=False):
    """C0 (+)= A0[rowsA]^T B0[rowsB] and C1 (+)= A1[rowsA]^T B1[rowsB]: two products of one shape in one launch
    (d2p_gemm_f32_tn_rows_x2)."""
    ws, wsb = SCRATCH.get(call.d2p_gemm_ws_bytes(M, N, K))
    call.d2p_gemm_f32_tn_rows_x2(M, N, K, _p(A0), lda0, _p(B0), ldb0, _p(C0), _p(A1), lda1, _p(B1), ldb1, _p(C1), ldc,
                                 _p(rowsA), _p(rowsB), 1 if accumulate else 0, ws, wsb, current_stream())


def gemm_batched(kind, nb1, nb0, M, N, K, A, lda, sA, B, ldb, sB, C, ldc, sC, bias=None, sbias=(0, 0), act=0,
                 accumulate=False):
    """nb1 x nb0 problems of one shape in one launch; sA / sB / sC / sbias = (stride over the first batch
    index, stride over the second), in floats.  A / B / C: tensors or data pointers of problem (0, 0)."""
    call.d2p_gemm_f32_batched({'nn': 0, 'nt': 1, 'tn': 2}[kind], nb1, nb0, M, N, K, _p(A), lda, sA[0], sA[1],
                              _p(B), ldb, sB[0], sB[1], _p(C), ldc, sC[0], sC[1], _p(bias), sbias[0], sbias[1],
                              act, 1 if accumulate else 0, current_stream())


def _p(x):
    if x is None or isinstance(x, int):
        return x
    return x.data_ptr()


def matmul_nn(A, B, out=None, bias=None, act=0, accumulate=False):
    """out[M,N] = A[M,K] @ B[K,N] (+bias, act)."""
    _require_gpu(A, B)
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    gemm_raw('nn', M, N, K, A, A.stride(0), B, B.stride(0), out, out.stride(0), bias, act, accumulate)
    return out


def matmul_nt(A, B, out=None, accumulate=False):
    """out[M,N] = A[M,K] @ B[N,K]^T."""
    _require_gpu(A, B)
    M, K = A.shape
    N, K2 = B.shape
    assert K == K2
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    gemm_raw('nt', M, N, K, A, A.stride(0), B, B.stride(0), out, out.stride(0), None, 0, accumulate)
    return out


def matmul_tn(A, B, out=None, accumulate=False):
    """out[M,N] = A[K,M]^T @ B[K,N]."""
    _require_gpu(A, B)
    K, M = A.shape
    K2, N = B.shape
    assert K == K2
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    gemm_raw('tn', M, N, K, A, A.stride(0), B, B.stride(0), out, out.stride(0), None, 0, accumulate)
    return out


def colsum(X, out=None, rows=None):
    _require_gpu(X)
    R, C = X.shape
    if rows is not None:
        R = rows
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=X.device)
    ws, wsb = SCRATCH.get(call.d2p_colsum_ws_bytes(R, C))
    call.d2p_colsum_f32(R, C, ptr(X), X.stride(0), ptr(out), ws, wsb, current_stream())
    return out


# ---------------------------------------------------------------- conv
def conv_out_hw(H, W):
    return (H + 1) // 2, (W + 1) // 2


def conv_fwd(x, w, bias, act=1, out=None):
    """x [N,H,W,Cin] fp32 or uint8; w [3,3,Cin,Cout]; returns act(conv+bias) [N,Ho,Wo,Cout]."""
    _require_gpu(x, w)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    Ho, Wo = conv_out_hw(H, W)
    if out is None:
        out = torch.empty(N, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    call.d2p_conv2d_nhwc_s2_same_fwd(N, H, W, Cin, Cout, ptr(x), 1 if x.dtype == torch.uint8 else 0,
                                     ptr(w), ptr(bias), act, ptr(out), current_stream())
    return out


def conv_bn_slices(x_shape, Cout, G, seq):
    """slices per demonstration index the batch-norm-folding forward conv of this geometry writes statistics for (0: no
    such kernel -- the separate conv / batch-norm launches run)"""
    N, H, W, Cin = x_shape
    return int(_load_lib().d2p_conv_bn_slices(N, H, W, Cin, Cout, G, seq))


def conv_bn_affine_ok(x_shape, Cout, G, seq):
    """this layer's folding forward / weight-gradient launches read their input through the previous layer's batch-norm apply"""
    N, H, W, Cin = x_shape
    return bool(_load_lib().d2p_conv_bn_affine_ok(N, H, W, Cin, Cout, G, seq))


def conv_fwd_bn(x, w, bias, G, seq, S, stats, act=1, out=None, in_affine=None):
    """conv_fwd that also leaves the batch-norm partial sums of its output in stats [G, S, Cout, 2] (fp64) and, with
    in_affine = (scale, shift) [G, Cin], reads its input as x * scale[g] + shift[g] (d2p_conv2d_nhwc_s2_same_fwd_bn)."""
    _require_gpu(x, w, stats)
    N, H, W, Cin = x.shape
    Cout = w.shape[3]
    Ho, Wo = conv_out_hw(H, W)
    if out is None:
        out = torch.empty(N, Ho, Wo, Cout, dtype=torch.float32, device=x.device)
    assert stats.dtype == torch.float64 and stats.numel() >= G * S * Cout * 2
    sc, sh = in_affine[:2] if in_affine is not None else (None, None)
    if in_affine is not None:
        # the pad pixels (-shift / scale per index) right behind x's last element
        assert len(in_affine) == 3 and in_affine[2].data_ptr() == x.data_ptr() + x.numel() * 4, 'pad pixels must follow x'
    call.d2p_conv2d_nhwc_s2_same_fwd_bn(N, H, W, Cin, Cout, ptr(x), 1 if x.dtype == torch.uint8 else 0, ptr(w), ptr(bias), act,
                                        ptr(out), G, seq, ptr(sc), ptr(sh), ptr(stats), S, current_stream())
    return out


def conv_wgrad_bn(x, dy, dw, G, seq, in_affine):
    """conv_wgrad whose input is x * scale[g] + shift[g] (the previous layer's batch-norm apply, never materialised)"""
    _require_gpu(x, dy, dw)
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    ws, wsb = SCRATCH.get(call.d2p_conv_ws_bytes(N, H, W, Cin, Cout))
    call.d2p_conv2d_nhwc_s2_same_wgrad_bn(N, H, W, Cin, Cout, ptr(x), 1 if x.dtype == torch.uint8 else 0, ptr(dy), ptr(dw),
                                          G, seq, ptr(in_affine[0]), ptr(in_affine[1]), ws, wsb, current_stream())
    return dw


def bn_stats_from_partials(stats, n_per_group, C, G, S, gamma, beta, mean, rstd, var=None, affine=None):
    """mean / rstd / var [G, C] from a folding conv launch's partial sums; affine = (scale, shift[, pad]) [G, C] buffers
    to receive gamma * rstd, beta - mean * gamma * rstd and the pad pixels -shift / scale (which must sit right behind
    the activation the affine belongs to when a folding forward conv reads it: conv_fwd_bn)"""
    sc, sh, pad = (tuple(affine) + (None,))[:3] if affine is not None else (None, None, None)
    call.d2p_bn_stats_from_partials(n_per_group, C, G, S, ptr(stats), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(var),
                                    ptr(sc), ptr(sh), ptr(pad), current_stream())


def conv_bnbwd_ok(x_shape, Cout):
    N, H, W, Cin = x_shape
    return bool(_load_lib().d2p_conv_bnbwd_ok(N, H, W, Cin, Cout))


def bn_bwd_coef(x2d, dy, gamma, mean, rstd, G, inner, coef, dgamma, dbeta, sums=None):
    """the sums of bn_bwd, leaving coef [G, C, 4] of dx = (k1 dy + k2 x + k3) lrelu'(x) instead of dx (d2p_bn_group_bwd_coef);
    sums = (stats, S): the partial sums the producer of dy left (conv_dgrad_bn) -- no pass over x and dy at all"""
    _require_gpu(x2d, dy, coef)
    R, C = x2d.shape
    ws, wsb = SCRATCH.get(call.d2p_bn_ws_bytes(R, C, G))
    st, S = sums if sums is not None else (None, 0)
    call.d2p_bn_group_bwd_coef(R, C, G, inner, ptr(x2d), ptr(dy), ptr(gamma), ptr(mean), ptr(rstd), ptr(coef), ptr(dgamma),
                               ptr(dbeta), ptr(st), S, ws, wsb, current_stream())
    return coef


def conv_dgrad_bn_slices(x_shape, Cout, G, seq):
    N, H, W, Cin = x_shape
    return int(_load_lib().d2p_conv_dgrad_bn_slices(N, H, W, Cin, Cout, G, seq))


def conv_dgrad_bn(dy, w, x_shape, act, mean, rstd, G, seq, stats, S, dx=None):
    """conv_dgrad that also leaves the batch-norm-backward partial sums of the layer whose output gradient it writes
    (act: that layer's pre-norm activation, mean / rstd [G, Cin]) in stats [G, S, Cin, 2] fp64"""
    _require_gpu(dy, w, act, stats)
    N, H, W, Cin = x_shape
    Cout = w.shape[3]
    if dx is None:
        dx = torch.empty(N, H, W, Cin, dtype=torch.float32, device=dy.device)
    assert stats.dtype == torch.float64 and stats.numel() >= G * S * Cin * 2
    call.d2p_conv2d_nhwc_s2_same_dgrad_bn(N, H, W, Cin, Cout, ptr(dy), ptr(w), ptr(dx), ptr(act), ptr(mean), ptr(rstd), G, seq,
                                          ptr(stats), S, current_stream())
    return dx


def conv_wgrad_bnbwd(x, act, dy, coef, G, seq, dw, dbias):
    """weight (and bias) gradient of a conv layer from the gradient w.r.t. its batch-norm OUTPUT: the batch-norm backward's
    apply pass is formed on load from act and dy (d2p_conv2d_nhwc_s2_same_wgrad_bnbwd)"""
    _require_gpu(x, act, dy, dw)
    N, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    ws, wsb = SCRATCH.get(call.d2p_conv_ws_bytes(N, H, W, Cin, Cout))
    call.d2p_conv2d_nhwc_s2_same_wgrad_bnbwd(N, H, W, Cin, Cout, ptr(x), 1 if x.dtype == torch.uint8 else 0, ptr(act), ptr(dy),
                                             ptr(coef), G, seq, ptr(dw), ptr(dbias), ws, wsb, current_stream())
    return dw


def bn_apply_fwd(x2d, gamma, beta, mean, rstd, G, inner, y=None):
    _require_gpu(x2d)
    R, C = x2d.shape
    if y is None:
        y = torch.empty_like(x2d)
    call.d2p_bn_apply_fwd(R, C, G, inner, ptr(x2d), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(y), current_stream())
    return y


def karel_encoder_ok(B, G, T):
    """whether the one-launch State_Encoder forward (d2p_karel_encoder_fwd) takes this batch geometry"""
    return _load_lib().d2p_karel_encoder_ws_bytes(B, G, T) > 0


def karel_encoder_fwd(x, B, G, T, w, bias, gamma, beta, a, y, feats_tm, mean, rstd, var, ws):
    """x [B*G*T, 8, 8, 16] fp32 / uint8; w, bias, gamma, beta, a, mean, rstd, var: three tensors each (layers 1-3),
    y: two (layers 1-2); feats_tm [T, B*G, 48]; ws: a uint8 buffer of d2p_karel_encoder_ws_bytes(B, G, T)."""
    _require_gpu(x, feats_tm, ws)
    import ctypes
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[ptr(t) for t in ts])        # noqa: E731
    call.d2p_karel_encoder_fwd(B, G, T, ptr(x), 1 if x.dtype == torch.uint8 else 0, arr(w), arr(bias), arr(gamma),
                               arr(beta), arr(a), arr(y), ptr(feats_tm), arr(mean), arr(rstd), arr(var), ptr(ws),
                               ws.numel() * ws.element_size(), current_stream())


def rn_ok(B, k, U):
    """whether the four-launch form of the relation networks' pointwise chains (d2p_rn_*) takes this geometry"""
    return _load_lib().d2p_rn_ws_bytes(B, k, U) > 0


def rn_fc1_fwd(P, Q, bias, gamma, beta, pstride, B, k, U, y1a, y1, mean, rstd, var, moving, decay, ws):
    """y1a = lrelu(P[b,c] + Q[b,a] + bias), y1 = its batch norm, both summaries (d2p_rn_fc1_fwd); moving = (mm, mv) [2, U] or None"""
    _require_gpu(P, y1a, ws)
    mm, mv = moving if moving is not None else (None, None)
    call.d2p_rn_fc1_fwd(B, k, U, ptr(P), ptr(Q), ptr(bias), ptr(gamma), ptr(beta), pstride, ptr(y1a), ptr(y1), ptr(mean),
                        ptr(rstd), ptr(var), ptr(mm), ptr(mv), decay, ptr(ws), ws.numel() * ws.element_size(), current_stream())


def rn_fc2_fwd(y2a, gamma, beta, pstride, feat, B, k, U, out, psum, mean, rstd, var, moving, decay, ws):
    """out = mean over a program's pairs of batch norm(y2a) (+ mean over k of feat), from one read of y2a (d2p_rn_fc2_fwd)"""
    _require_gpu(y2a, out, ws)
    mm, mv = moving if moving is not None else (None, None)
    call.d2p_rn_fc2_fwd(B, k, U, ptr(y2a), ptr(gamma), ptr(beta), pstride, ptr(feat), ptr(out), ptr(psum), ptr(mean), ptr(rstd),
                        ptr(var), ptr(mm), ptr(mv), decay, ptr(ws), ws.numel() * ws.element_size(), current_stream())


def rn_fc2_bwd(y2a, dout, psum, gamma, pstride, mean, rstd, B, k, U, dpre, dgamma, dbeta, ws):
    _require_gpu(y2a, dout, dpre, ws)
    assert dout.is_contiguous() and psum.is_contiguous()
    call.d2p_rn_fc2_bwd(B, k, U, ptr(y2a), ptr(dout), ptr(psum), ptr(gamma), pstride, ptr(mean), ptr(rstd), ptr(dpre),
                        ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel() * ws.element_size(), current_stream())


def rn_fc1_bwd(y1a, dy1, gamma, pstride, mean, rstd, B, k, U, dP, dQ, dgamma, dbeta, dbias, dbias2, ws):
    _require_gpu(y1a, dy1, dP, ws)
    call.d2p_rn_fc1_bwd(B, k, U, ptr(y1a), ptr(dy1), ptr(gamma), pstride, ptr(mean), ptr(rstd), ptr(dP), ptr(dQ), ptr(dgamma),
                        ptr(dbeta), ptr(dbias), ptr(dbias2), ptr(ws), ws.numel() * ws.element_size(), current_stream())


def karel_encoder_bwd_ok(B, G, T):
    """whether the one-launch State_Encoder backward (d2p_karel_encoder_bwd) takes this batch geometry"""
    return _load_lib().d2p_karel_encoder_bwd_ws_bytes(B, G, T) > 0


def karel_encoder_bwd(x, dfeat_tm, B, G, T, w, gamma, beta, a, mean, rstd, dw, db, dgamma, dbeta, ws):
    """the backward of karel_encoder_fwd from dfeat_tm [T, B*G, 48]: dw, db, dgamma, dbeta (three tensors each) are
    written; ws: a uint8 buffer of d2p_karel_encoder_bwd_ws_bytes(B, G, T)."""
    _require_gpu(x, dfeat_tm, ws)
    import ctypes
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[ptr(t) for t in ts])        # noqa: E731
    assert dfeat_tm.is_contiguous() and all(t.is_contiguous() for t in list(dw) + list(db) + list(dgamma) + list(dbeta))
    call.d2p_karel_encoder_bwd(B, G, T, ptr(x), 1 if x.dtype == torch.uint8 else 0, ptr(dfeat_tm), arr(w), arr(gamma),
                               arr(beta), arr(a), arr(mean), arr(rstd), arr(dw), arr(db), arr(dgamma), arr(dbeta), ptr(ws),
                               ws.numel() * ws.element_size(), current_stream())


def conv_wgrad(x, dy, dw):
    _require_gpu(x, dy, dw)
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    ws, wsb = SCRATCH.get(call.d2p_conv_ws_bytes(N, H, W, Cin, Cout))
    call.d2p_conv2d_nhwc_s2_same_wgrad(N, H, W, Cin, Cout, ptr(x), 1 if x.dtype == torch.uint8 else 0,
                                       ptr(dy), ptr(dw), ws, wsb, current_stream())
    return dw


def conv_dgrad(dy, w, x_shape, dx=None):
    _require_gpu(dy, w)
    N, H, W, Cin = x_shape
    Cout = w.shape[3]
    if dx is None:
        dx = torch.empty(N, H, W, Cin, dtype=torch.float32, device=dy.device)
    call.d2p_conv2d_nhwc_s2_same_dgrad(N, H, W, Cin, Cout, ptr(dy), ptr(w), ptr(dx), current_stream())
    return dx


# ---------------------------------------------------------------- batch norm
def bn_fwd(x2d, gamma, beta, G, inner, y=None, mean=None, rstd=None, var=None, moving=None, decay=0.9):
    """x2d [R,C]; returns (y, mean[G,C], rstd[G,C], var) -- var only if a buffer is passed.
    moving = (moving_mean, moving_var) [C]: also applies this call's G moving-average updates."""
    _require_gpu(x2d)
    R, C = x2d.shape
    if y is None:
        y = torch.empty_like(x2d)
    if mean is None:
        mean = torch.empty(G, C, dtype=torch.float32, device=x2d.device)
    if rstd is None:
        rstd = torch.empty(G, C, dtype=torch.float32, device=x2d.device)
    ws, wsb = SCRATCH.get(call.d2p_bn_ws_bytes(R, C, G))
    mm, mv = moving if moving is not None else (None, None)
    call.d2p_bn_group_fwd(R, C, G, inner, ptr(x2d), ptr(gamma), ptr(beta), ptr(y), ptr(mean),
                          ptr(rstd), ptr(var), ptr(mm), ptr(mv), decay, ws, wsb, current_stream())
    return y, mean, rstd, var


def bn_bwd(x2d, dy, gamma, mean, rstd, G, inner, act_bwd, dgamma, dbeta, dx=None, dbias=None, sums=None):
    """Training-mode BN backward (+ lrelu' of the layer underneath when act_bwd).  dbias, if
    given, receives colsum(dx): the gradient of the bias added before the activation.
    sums = (stats, S): the partial sums the producer of dy left (conv_dgrad_bn) -- no partial-sum pass over x and dy."""
    _require_gpu(x2d, dy)
    R, C = x2d.shape
    if dx is None:
        dx = torch.empty_like(x2d)
    ws, wsb = SCRATCH.get(call.d2p_bn_ws_bytes(R, C, G))
    if sums is not None:
        call.d2p_bn_group_bwd_sums(R, C, G, inner, ptr(x2d), ptr(dy), ptr(gamma), ptr(mean), ptr(rstd),
                                   1 if act_bwd else 0, ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dbias), ptr(sums[0]), sums[1],
                                   ws, wsb, current_stream())
        return dx
    call.d2p_bn_group_bwd(R, C, G, inner, ptr(x2d), ptr(dy), ptr(gamma), ptr(mean), ptr(rstd),
                          1 if act_bwd else 0, ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dbias), ws, wsb,
                          current_stream())
    return dx


def bn_fwd_batched(x3d, gamma, beta, pstride, G, inner, y, mean, rstd, moving=None, mstride=0, decay=0.9):
    """nb = x3d.shape[0] independent BN problems [R, C] in one set of launches: problem b uses gamma + b*pstride
    (beta likewise), moving statistics + b*mstride, mean / rstd [nb, G, C], y [nb, R, C]."""
    _require_gpu(x3d)
    nb, R, C = x3d.shape
    ws, wsb = SCRATCH.get(call.d2p_bn_batched_ws_bytes(nb, R, C, G))
    mm, mv = moving if moving is not None else (None, None)
    call.d2p_bn_group_fwd_batched(nb, x3d.stride(0), y.stride(0), pstride, mstride, R, C, G, inner, ptr(x3d), ptr(gamma),
                                  ptr(beta), ptr(y), ptr(mean), ptr(rstd), None, ptr(mm), ptr(mv), decay, ws, wsb,
                                  current_stream())
    return y, mean, rstd


def bn_bwd_batched(x3d, dy3d, gamma, pstride, mean, rstd, G, inner, act_bwd, dgamma, dbeta, dx, dbias=None):
    """Backward of bn_fwd_batched: dgamma / dbeta / dbias of problem b land at + b*pstride."""
    _require_gpu(x3d, dy3d)
    nb, R, C = x3d.shape
    assert dy3d.stride(0) == x3d.stride(0)
    ws, wsb = SCRATCH.get(call.d2p_bn_batched_ws_bytes(nb, R, C, G))
    call.d2p_bn_group_bwd_batched(nb, x3d.stride(0), dx.stride(0), pstride, R, C, G, inner, ptr(x3d), ptr(dy3d), ptr(gamma),
                                  ptr(mean), ptr(rstd), 1 if act_bwd else 0, ptr(dx), ptr(dgamma), ptr(dbeta),
                                  ptr(dbias), ws, wsb, current_stream())
    return dx


def per_affine_rows(G, P, W, b, gamma, beta, mean, rstd, H):
    """H [NCp, U]: the rows pe = A . H is built from (d2p.h: d2p_per_affine_rows)."""
    NCp, U = H.shape
    call.d2p_per_affine_rows(G, P, U, NCp, ptr(W), ptr(b), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(H),
                             current_stream())
    return H


def per_rows_nn(G, per_tm2d, HWx, bias, z, rows):
    """z[:rows] = A . HWx + bias from per_tm2d [.., P] (d2p_per_rows_nn: the structure of A, no GEMM)."""
    call.d2p_per_rows_nn(rows, G, per_tm2d.shape[1], HWx.shape[0], HWx.shape[1], ptr(per_tm2d), ptr(HWx), ptr(bias), ptr(z),
                         current_stream())
    return z


def per_rows_tn_ok(rows, G, P, E):
    return 0 < G <= 15 and 0 < P <= 8 and rows > 0 and rows % G == 0 and E >= 64 and E % 4 == 0


def per_rows_tn(G, per_tm2d, dz, S, rows):
    """S [NCp, E] = A^T dz[:rows] from per_tm2d [rows.., P] (d2p_per_rows_tn: the structure of A, no GEMM)."""
    P_, (NCp, E) = per_tm2d.shape[1], S.shape
    ws, wsb = SCRATCH.get(call.d2p_per_rows_tn_ws_bytes(rows, G, NCp, E))
    call.d2p_per_rows_tn(rows, G, P_, NCp, E, ptr(per_tm2d), ptr(dz), ptr(S), ws, wsb, current_stream())
    return S


def per_fc_bn_stats(G, P, rows_per_group, W, b, gram, mean, rstd, var=None):
    """mean / rstd / var [G, U] of per . W + b per demonstration index, from gram alone (d2p_per_fc_bn_stats)."""
    U = W.shape[1]
    call.d2p_per_fc_bn_stats(G, P, U, gram.shape[0], rows_per_group, ptr(W), ptr(b), ptr(gram), ptr(mean), ptr(rstd),
                             ptr(var), current_stream())


def per_fc_bn_bwd(G, P, rows_per_group, W, b, gamma, mean, rstd, Q, gram, dW, db, dgamma, dbeta):
    NCp, U = Q.shape
    call.d2p_per_fc_bn_bwd(G, P, U, NCp, rows_per_group, ptr(W), ptr(b), ptr(gamma), ptr(mean), ptr(rstd), ptr(Q),
                           ptr(gram), ptr(dW), ptr(db), ptr(dgamma), ptr(dbeta), current_stream())


def bn_inference(x2d, gamma, beta, moving_mean, moving_var, y=None):
    """is_training=False batch norm: normalise [R, C] rows with the moving statistics."""
    _require_gpu(x2d)
    R, C = x2d.shape
    if y is None:
        y = torch.empty_like(x2d)
    call.d2p_bn_inference_fwd(R, C, ptr(x2d), ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var),
                              ptr(y), current_stream())
    return y


def bn_update_moving(mean, var, moving_mean, moving_var, decay=0.9):
    G, C = mean.shape
    call.d2p_bn_update_moving(C, G, decay, ptr(mean), ptr(var), ptr(moving_mean), ptr(moving_var),
                              current_stream())


def bn_set_fold(bits):
    """Process-global A/B switch (d2p_bn_set_fold): bit 0 = batch-norm finalize steps folded into the partial-sum /
    apply launches by tickets (off by default: no gain), bit 1 = the round-2 finalize kernels (one wavefront per
    channel walking the groups) instead of one wavefront per (group, channel)."""
    call.d2p_bn_set_fold(int(bits))


# ---------------------------------------------------------------- LSTM
def set_lstm_fused(on):
    """Process-global knob: fused recurrent-step kernels (default) vs GEMM + gate per step."""
    call.d2p_lstm_set_fused(1 if on else 0)


_LSTM_PERSISTENT = [True]


def set_lstm_persistent(on):
    """Process-global knob: one persistent launch per sequence (default) vs one fused launch per step."""
    call.d2p_lstm_set_persistent(1 if on else 0)
    _LSTM_PERSISTENT[0] = bool(on)


lstm_set_persistent = set_lstm_persistent


def lstm_is_persistent():
    return _LSTM_PERSISTENT[0]


def lstm_persist_inject_error():
    """Test hook: sets the persistent kernels' status word as a timed-out hand-off would."""
    call.d2p_lstm_persist_inject_error()


def lstm_persist_error(reset=True):
    """Synchronising: non-zero when a persistent LSTM launch gave up on a hand-off (results invalid)."""
    return _load_lib().d2p_lstm_persist_error(1 if reset else 0)


def lstm_seq_fwd(z, z_row_stride, z_t_stride, M, U, n_steps, Wh, h0, c0, lens, hout, cs,
                 h_final, c_final):
    ws, wsb = SCRATCH.get(call.d2p_lstm_ws_bytes(M, U))
    call.d2p_lstm_seq_fwd(M, U, n_steps, ptr(z), z_row_stride, z_t_stride, ptr(Wh), ptr(h0), ptr(c0),
                          ptr(lens), ptr(hout), ptr(cs), ptr(h_final), ptr(c_final), ws, wsb,
                          current_stream())


def lstm_seq_bwd(z, z_row_stride, z_t_stride, M, U, n_steps, Wh, c0, lens, cs, dhout, dh_final,
                 dc_final, dz, dh0, dc0):
    ws, wsb = SCRATCH.get(call.d2p_lstm_ws_bytes(M, U))
    call.d2p_lstm_seq_bwd(M, U, n_steps, ptr(z), z_row_stride, z_t_stride, ptr(Wh), ptr(c0),
                          ptr(lens), ptr(cs), ptr(dhout), ptr(dh_final), ptr(dc_final), ptr(dz),
                          ptr(dh0), ptr(dc0), ws, wsb, current_stream())


class _MultiWs(object):
    """One workspace per concurrently advancing sequence (they must not share scratch)."""

    def __init__(self):
        self.bufs = {}

    def get(self, slot, nbytes):
        b = self.bufs.get(slot)
        if b is None or b.numel() < nbytes:
            b = self.bufs[slot] = torch.empty(int(nbytes), dtype=torch.uint8, device='cuda')
        return b


_MULTI_WS = _MultiWs()


class _LstmFlags(object):
    """Flag buffers of the "direct" persistent launches (d2p_lstm_*_desc.flags / .epoch): one per (direction, sequence
    slot, stream), zeroed once; the epoch of a buffer rises by n_steps + 2 with every launch that got it.  Not under
    hipGraph capture (the epoch would be baked into the graph): those launches keep their preparation launch."""

    def __init__(self):
        self.bufs = {}

    def take(self, key, n_steps):
        if torch.cuda.is_current_stream_capturing():
            return None, 0
        ent = self.bufs.get(key)
        if ent is None:
            ent = self.bufs[key] = [torch.zeros(int(call.d2p_lstm_flag_words()), dtype=torch.int32, device='cuda'), 0]
        if ent[1] + n_steps + 2 >= (1 << 31):
            ent[0].zero_()                       # stream-ordered: behind every launch that used the buffer
            ent[1] = 0
        epoch = ent[1]
        ent[1] += n_steps + 2
        return ent[0].data_ptr(), epoch


_LSTM_FLAGS = _LstmFlags()


def lstm_row_order(lens, pad_to=16):
    """(rowmap, slab_steps) of d2p_lstm_bwd_desc for host lengths `lens` (numpy, one per row): the rows by decreasing
    length (stable) as a device int32 tensor, and the longest length of every 16 consecutive rows of that order as a
    host int32 array."""
    lens = np.asarray(lens).reshape(-1).astype(np.int64)
    order = np.argsort(-lens, kind='stable').astype(np.int32)
    steps = np.ascontiguousarray(lens[order][::pad_to].astype(np.int32))
    return torch.from_numpy(order).cuda(), steps


def lstm_set_sorted(on):
    call.d2p_lstm_persist_set_sorted(1 if on else 0)


def lstm_set_cu_budget(cus):
    """CUs the persistent recurrences are planned for (0: all); fewer leaves whole CUs to the other queue."""
    call.d2p_lstm_persist_set_cu_budget(int(cus))


def lstm_pack_weights(cells):
    """cells: list (<= 8) of (Wh [U, 4U], Wf image or None, Wb image or None): the packed weight images of the persistent
    kernels in one launch."""
    import ctypes
    n = len(cells)
    U = cells[0][0].shape[0]
    arrs = [(ctypes.c_void_p * n)(*[ptr(c[j]) for c in cells]) for j in range(3)]
    call.d2p_lstm_pack_weights(n, U, ctypes.cast(arrs[0], ctypes.c_void_p), ctypes.cast(arrs[1], ctypes.c_void_p),
                               ctypes.cast(arrs[2], ctypes.c_void_p), current_stream())


def lstm_seq_fwd_multi(seqs):
    """seqs: list (<= 3) of dicts with the d2p_lstm_seq_fwd arguments (time-major z)."""
    import ctypes
    from .lib import LstmFwdDesc
    arr = (LstmFwdDesc * len(seqs))()
    keep = []
    for i, q in enumerate(seqs):
        M, U = q['M'], q['U']
        nb = call.d2p_lstm_ws_bytes(M, U)
        ws = _MULTI_WS.get(('f', i, torch.cuda.current_stream().cuda_stream), nb)   # per stream: launches on two streams run concurrently
        d = arr[i]
        d.M, d.U, d.n_steps = M, U, q['n_steps']
        d.z, d.z_row_stride, d.z_t_stride = ptr(q['z']), 4 * U, M * 4 * U
        d.Wh, d.h0, d.c0, d.lens = ptr(q['Wh']), ptr(q.get('h0')), ptr(q.get('c0')), ptr(q.get('lens'))
        d.hout, d.cs = ptr(q['hout']), ptr(q['cs'])
        d.h_final, d.c_final = ptr(q.get('h_final')), ptr(q.get('c_final'))
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
        d.flags, d.epoch = _LSTM_FLAGS.take(('f', i, torch.cuda.current_stream().cuda_stream), q['n_steps'])
        d.wpack = ptr(q.get('wpack'))
        order = q.get('row_order')          # (rowmap: device int32 [M], slab_steps: host numpy int32 [ceil(M/16)])
        if order is not None and q.get('lens') is not None:
            rowmap, steps = order
            assert rowmap.dtype == torch.int32 and rowmap.numel() == q['M'] and steps.dtype == np.int32
            assert steps.size == (q['M'] + 15) // 16 and steps.flags['C_CONTIGUOUS']
            keep.append(steps)
            d.rowmap, d.slab_steps = ptr(rowmap), steps.ctypes.data
    call.d2p_lstm_seq_fwd_multi(len(seqs), ctypes.cast(arr, ctypes.c_void_p), current_stream())


def lstm_seq_bwd_multi(seqs):
    import ctypes
    from .lib import LstmBwdDesc
    arr = (LstmBwdDesc * len(seqs))()
    keep = []
    for i, q in enumerate(seqs):
        M, U = q['M'], q['U']
        nb = call.d2p_lstm_ws_bytes(M, U)
        if q.get('db') is not None:      # the per-step back ends take the bias gradient by a column-sum pass over dz
            nb = max(nb, call.d2p_colsum_ws_bytes(q['n_steps'] * M, 4 * U))
        ws = _MULTI_WS.get(('b', i, torch.cuda.current_stream().cuda_stream), nb)
        d = arr[i]
        d.M, d.U, d.n_steps = M, U, q['n_steps']
        d.z, d.z_row_stride, d.z_t_stride = ptr(q['z']), 4 * U, M * 4 * U
        d.Wh, d.c0, d.lens, d.cs = ptr(q['Wh']), ptr(q.get('c0')), ptr(q.get('lens')), ptr(q['cs'])
        d.dhout, d.dh_final, d.dc_final = ptr(q.get('dhout')), ptr(q.get('dh_final')), ptr(q.get('dc_final'))
        d.dz, d.dh0, d.dc0 = ptr(q['dz']), ptr(q.get('dh0')), ptr(q.get('dc0'))
        d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
        d.db = ptr(q.get('db'))            # optional: the bias gradient (column sums of dz), produced with the launch
        d.flags, d.epoch = _LSTM_FLAGS.take(('b', i, torch.cuda.current_stream().cuda_stream), q['n_steps'] + 1)
        d.wpack = ptr(q.get('wpack'))
        order = q.get('row_order')          # (rowmap: device int32 [M], slab_steps: host numpy int32 [ceil(M/16)])
        if order is not None:
            rowmap, steps = order
            assert rowmap.dtype == torch.int32 and rowmap.numel() == q['M'] and steps.dtype == np.int32
            assert steps.size == (q['M'] + 15) // 16 and steps.flags['C_CONTIGUOUS']
            keep.append(steps)
            d.rowmap, d.slab_steps = ptr(rowmap), steps.ctypes.data
    call.d2p_lstm_seq_bwd_multi(len(seqs), ctypes.cast(arr, ctypes.c_void_p), current_stream())


def lstm_gate_fwd(z, c_prev, h_prev, lens, t, c_out, hs_out, h_out):
    M = c_out.shape[0]
    U = c_out.shape[1]
    call.d2p_lstm_gate_fwd(M, U, ptr(z), z.stride(0), ptr(c_prev), ptr(h_prev), ptr(lens), t,
                           ptr(c_out), ptr(hs_out), ptr(h_out), current_stream())


def lstm_gate_bwd(z, c_prev, c, dh_in, dh_out_grad, lens, t, dc, dz, dh_pass):
    M, U = c.shape
    call.d2p_lstm_gate_bwd(M, U, ptr(z), z.stride(0), ptr(c_prev), ptr(c), ptr(dh_in),
                           ptr(dh_out_grad), ptr(lens), t, ptr(dc), ptr(dz), dz.stride(0),
                           ptr(dh_pass), current_stream())


# ---------------------------------------------------------------- embedding
def shift_tokens_tm(tokens, start_id, out=None):
    """tokens [R,T] int32 -> ids [T,R] int32 with <s> first."""
    R, T = tokens.shape
    if out is None:
        out = torch.empty(T, R, dtype=torch.int32, device=tokens.device)
    call.d2p_shift_tokens_tm(R, T, ptr(tokens), start_id, ptr(out), current_stream())
    return out


def embedding_gather(ids, table, out=None, n=None):
    rows, E = table.shape
    if n is None:
        n = ids.numel()
    if out is None:
        out = torch.empty(ids.numel(), E, dtype=torch.float32, device=table.device)
    call.d2p_embedding_gather_oob0(n, rows, E, ptr(ids), ptr(table), ptr(out), current_stream())
    return out


def embedding_scatter_add(ids, dout, dtable, n=None):
    rows, E = dtable.shape
    if n is None:
        n = ids.numel()
    ws, wsb = SCRATCH.get(call.d2p_embedding_scatter_ws_bytes(n, rows, E))
    call.d2p_embedding_scatter_add_oob0(n, rows, E, ptr(ids), ptr(dout), ptr(dtable), ws, wsb,
                                        current_stream())
    return dtable


def greedy_decode(table_proj, Wh, proj, h0, c0, start_id, end_id, L, logits, ids, lengths):
    """table_proj [V+1,4U]; logits [L,M,V], ids [L,M] int32, lengths [M] int32 (outputs)."""
    M, U = h0.shape
    V = proj.shape[1]
    ws, wsb = SCRATCH.get(call.d2p_greedy_ws_bytes(M, U, V))
    call.d2p_greedy_decode(M, U, V, L, ptr(table_proj), ptr(Wh), ptr(proj), ptr(h0), ptr(c0),
                           start_id, end_id, ptr(logits), ptr(ids), ptr(lengths), ws, wsb,
                           current_stream())


def argmax_rows(x2d, out=None):
    rows, V = x2d.shape
    if out is None:
        out = torch.empty(rows, dtype=torch.int32, device=x2d.device)
    call.d2p_argmax_rows(rows, V, ptr(x2d), x2d.stride(0), ptr(out), current_stream())
    return out


# ---------------------------------------------------------------- losses
def _lab_strides(kind, V, T):
    # program labels [B,V,L]; action/per labels [B,k,T,V] flattened to rows r=(b,i)
    if kind == 'bvl':
        return V * T, 1, T
    return T * V, V, 1


def xent_fwd(mode, logits, labels, lab_kind, lens, T, R, V, G, n_steps, num, den):
    ws, wsb = SCRATCH.get(call.d2p_xent_ws_bytes(G))
    rs, ts, vs = _lab_strides(lab_kind, V, T)
    fn = call.d2p_softmax_xent_masked_fwd if mode == 'softmax' else call.d2p_sigmoid_xent_masked_fwd
    fn(T, R, V, G, n_steps, ptr(logits), ptr(labels), rs, ts, vs, ptr(lens), ptr(num), ptr(den),
       ws, wsb, current_stream())


def xent_bwd(mode, logits, labels, lab_kind, lens, T, R, V, G, n_steps, den, scale, dlogits):
    rs, ts, vs = _lab_strides(lab_kind, V, T)
    fn = call.d2p_softmax_xent_masked_bwd if mode == 'softmax' else call.d2p_sigmoid_xent_masked_bwd
    fn(T, R, V, G, n_steps, ptr(logits), ptr(labels), rs, ts, vs, ptr(lens), ptr(den), scale,
       ptr(dlogits), current_stream())


def xent_bwd_dhout_multi(probs):
    """One launch: dlogits and dhout = dlogits . proj^T of up to three decoders.  probs: dicts with mode
    ('softmax' | 'sigmoid'), logits, labels, lab_kind, lens, T, R, V, G, n_steps, den, scale, dlogits, proj, dhout, U."""
    import ctypes
    from .lib import XentBwdDesc
    arr = (XentBwdDesc * len(probs))()
    for d, q in zip(arr, probs):
        rs, ts, vs = _lab_strides(q['lab_kind'], q['V'], q['T'])
        d.sigmoid = 1 if q['mode'] == 'sigmoid' else 0
        d.R, d.V, d.G, d.n_steps, d.U = q['R'], q['V'], q['G'], q['n_steps'], q['U']
        d.logits, d.labels, d.label_rs, d.label_ts, d.label_vs = ptr(q['logits']), ptr(q['labels']), rs, ts, vs
        d.lens, d.den, d.scale = ptr(q['lens']), ptr(q['den']), q['scale']
        d.dlogits, d.proj, d.dhout = ptr(q['dlogits']), ptr(q['proj']), ptr(q['dhout'])
        if q.get('hout') is not None:       # the launch computes the logits itself: logits <- hout . proj, then as above
            d.hout, d.logits_out = ptr(q['hout']), ptr(q['logits'])
        if q.get('loss_part') is not None:  # per-workgroup sums of the rows' loss values (loss_from_partials)
            assert q['loss_part'].numel() >= xent_blocks(q['n_steps'], q['R']) * q['G']
            d.loss_part = ptr(q['loss_part'])
    call.d2p_xent_bwd_dhout_multi(len(probs), ctypes.cast(arr, ctypes.c_void_p), current_stream())


def pair_products_ok(R, U):
    return 1 <= R <= 256 and U >= 128 and U % 128 == 0


def small_pair_products(probs):
    """G1 = A^T S and G2 = S Wx^T for up to four decoders in one launch (d2p_small_pair_products).
    probs: (R, U, S [>= R, 4U], A [R, U], Wx [U, 4U], G1 [U, 4U], G2 [R, U]) tuples."""
    import ctypes
    from .lib import PairProductsDesc
    arr = (PairProductsDesc * len(probs))()
    for d, (R, U, S_, A, Wx, G1, G2) in zip(arr, probs):
        assert S_.stride(0) == 4 * U and A.stride(0) == U and Wx.stride(0) == 4 * U and G1.stride(0) == 4 * U and G2.stride(0) == U
        d.R, d.U, d.N4 = R, U, 4 * U
        d.S, d.A, d.Wx, d.G1, d.G2 = ptr(S_), ptr(A), ptr(Wx), ptr(G1), ptr(G2)
    call.d2p_small_pair_products(len(probs), ctypes.cast(arr, ctypes.c_void_p), current_stream())


def xent_blocks(n_steps, R):
    """workgroups (of 16 rows) xent_bwd_dhout_multi gives a problem = rows of its loss_part"""
    return (n_steps * R + 15) // 16


def loss_from_partials(groups, nblocks, parts, dens, nums, loss, term_losses):
    """The loss value from the loss_part arrays of xent_bwd_dhout_multi (d2p_loss_from_partials)."""
    import ctypes
    n = len(groups)
    ga, na = (ctypes.c_int * n)(*groups), (ctypes.c_int * n)(*nblocks)
    pa = (ctypes.c_void_p * n)(*[ptr(t) for t in parts])
    call.d2p_loss_from_partials(n, ctypes.cast(ga, ctypes.c_void_p), ctypes.cast(na, ctypes.c_void_p),
                                ctypes.cast(pa, ctypes.c_void_p), ptr(dens), ptr(nums), ptr(loss), ptr(term_losses),
                                current_stream())


def loss_assemble(groups, nums, dens, loss, term_losses):
    import ctypes
    arr = (ctypes.c_int * len(groups))(*groups)
    call.d2p_loss_assemble(len(groups), ctypes.cast(arr, ctypes.c_void_p), ptr(nums), ptr(dens),
                           ptr(loss), ptr(term_losses), current_stream())


def zero_past_group_steps(logits, lens, T, R, V, G):
    call.d2p_zero_past_group_steps(T, R, V, G, ptr(lens), ptr(logits), current_stream())


# ---------------------------------------------------------------- summarizer glue
def group_mean(x, B, k, U, out, bcast=None):
    call.d2p_group_mean(B, k, U, ptr(x), ptr(out), ptr(bcast), current_stream())


def group_mean_bwd(dout, dbcast, dx, B, k, U, accumulate):
    call.d2p_group_mean_bwd(B, k, U, ptr(dout), ptr(dbcast), ptr(dx), 1 if accumulate else 0,
                            current_stream())


def group_max(x, B, k, U, out, arg):
    call.d2p_group_max(B, k, U, ptr(x), ptr(out), ptr(arg), current_stream())


def group_max_bwd(dout, arg, dx, B, k, U, accumulate):
    call.d2p_group_max_bwd(B, k, U, ptr(dout), ptr(arg), ptr(dx), 1 if accumulate else 0, current_stream())


def rn_pair_fwd(P, Q, bias, y, B, k, U, scopes=1, bias_stride=0):
    call.d2p_rn_pair_fwd(B, k, U, ptr(P), ptr(Q), ptr(bias), scopes, bias_stride, ptr(y), current_stream())


def rn_pair_bwd(dy, dP, dQ, B, k, U):
    call.d2p_rn_pair_bwd(B, k, U, ptr(dy), ptr(dP), ptr(dQ), current_stream())


def pair_mean_fwd(y, base, out, B, kk, U):
    call.d2p_pair_mean_fwd(B, kk, U, ptr(y), ptr(base), ptr(out), current_stream())


def pair_mean_bwd(dout, dy, B, kk, U):
    call.d2p_pair_mean_bwd(B, kk, U, ptr(dout), ptr(dy), current_stream())


def axpy(a, x, y, accumulate=True, n=None):
    if n is None:
        n = x.numel()
    call.d2p_axpy(n, a, ptr(x), ptr(y), 1 if accumulate else 0, current_stream())


def transpose_rt(x, R, T, C, out=None):
    if out is None:
        out = torch.empty(T, R, C, dtype=torch.float32, device=x.device)
    call.d2p_transpose_rt(R, T, C, ptr(x), ptr(out), current_stream())
    return out


def pad_axis(x, outer, C, Cp, inner, out, unpad=False):
    """out[o,c,i] = x[o,c,i] (c < C) else 0 over [outer,Cp,inner]; unpad=True slices back."""
    call.d2p_pad_axis(outer, C, Cp, inner, ptr(x), ptr(out), 1 if x.dtype == torch.uint8 else 0,
                      1 if unpad else 0, current_stream())
    return out


def sched_sample(logits2d, gt_next, p_sample_dev, rng_dev, t, next_ids, sampled_flag=None):
    """One scheduled-sampling decision per row: next_ids = Categorical(logits) with probability
    p_sample_dev[0] else gt_next (seq2seq.ScheduledEmbeddingTrainingHelper)."""
    M, V = logits2d.shape
    call.d2p_sched_sample(M, V, ptr(logits2d), ptr(gt_next), ptr(p_sample_dev), ptr(rng_dev), int(t),
                          ptr(next_ids), ptr(sampled_flag), current_stream())
    return next_ids


# ---------------------------------------------------------------- optimizer
def l2norm_flat(g, prescale, sumsq):
    ws, wsb = SCRATCH.get(call.d2p_l2norm_ws_bytes(g.numel()))
    call.d2p_l2norm_flat(g.numel(), ptr(g), prescale, ptr(sumsq), ws, wsb, current_stream())


def adam_clip_flat(p, g, m, v, sumsq, prescale, clip, lr_t, b1=0.9, b2=0.999, eps=1e-8, lr_t_dev=None,
                   counters=None, fail_slot=None, mirror=None):
    """counters (int64[2], device): the guarded form -- the update is skipped while the persistent recurrent
    kernels' status word is set (or fail_slot[0] != 0); counters[0] / [1] count applied / skipped steps; mirror
    (int64[2], PINNED host memory): receives both after this step."""
    if counters is None:
        call.d2p_adam_clip_flat(p.numel(), ptr(p), ptr(g), ptr(m), ptr(v), ptr(sumsq), prescale, clip,
                                lr_t, ptr(lr_t_dev), b1, b2, eps, current_stream())
    else:
        assert counters.dtype == torch.int64 and counters.numel() >= 2
        call.d2p_adam_clip_flat_guarded(p.numel(), ptr(p), ptr(g), ptr(m), ptr(v), ptr(sumsq), prescale, clip,
                                        lr_t, ptr(lr_t_dev), b1, b2, eps, ptr(fail_slot), ptr(counters),
                                        ptr(mirror), current_stream())


def step_status_publish(slot):
    """slot[0] = 1.0 if the persistent recurrent kernels' status word is set, else 0.0 (stream-async)."""
    call.d2p_step_status_publish(ptr(slot), current_stream())
