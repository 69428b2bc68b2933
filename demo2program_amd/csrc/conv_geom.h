// Geometry shared by the two convolution back ends (conv.hip: implicit-im2col GEMM;
// conv_direct.hip: register-resident-weight direct kernels).
// TF SAME padding for k=3,s=2: out = ceil(n/2), pad_total = max((out-1)*2+3-n, 0),
// pad_before = pad_total/2  => even n: (0,1); odd n: (1,1).   
#pragma once
#include "common.h"

struct ConvGeom {
    int N, H, W, Cin, Cout, Ho, Wo, pt, pl;
};

static inline void same_pad(int n, int* out, int* before) {
    *out = (n + 1) / 2;
    int total = (*out - 1) * 2 + 3 - n;
    if (total < 0) total = 0;
    *before = total / 2;
}

static inline ConvGeom make_geom(int N, int H, int W, int Cin, int Cout) {
    ConvGeom g;
    g.N = N; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
    same_pad(H, &g.Ho, &g.pt);
    same_pad(W, &g.Wo, &g.pl);
    return g;
}

// Batch-norm pieces folded into a conv launch (round 5; the ViZDoom-size layers, models/ops.py:14-33: conv -> lrelu ->
// batch norm with statistics per demonstration index g = (frame / seq) % G):
//   in_scale / in_shift [G, Cin] (or null): the launch's input is x * scale[g] + shift[g] -- x is the PREVIOUS layer's
//       pre-norm activation and the affine its batch-norm apply (gamma * rstd, beta - mean * gamma * rstd), so the
//       normalised tensor is never written; zero padding applies to the normalised values;
//   stats (or null): the launch leaves [G][S][Cout][2] fp64 partial sums (sum, sum of squares) of ITS outputs behind --
//       the layout d2p_bn_stats_from_partials reads; S slices per index, chosen by d2p_conv_bn_slices.
struct ConvBnFold {
    int G, seq, S;
    const float* in_scale; const float* in_shift;
    double* stats;
};

// The input-gradient launch of layer l+1 leaving the batch-norm BACKWARD partial sums of layer l: dx is the gradient w.r.t.
// layer l's batch-norm output, act its pre-norm activation, mean / rstd [G, Cin] its statistics; stats [G][S][Cin][2]
// fp64 = (sum dx, sum dx * xhat) per (index, slice) -- the layout bn_finalize_bwd reads.
struct ConvDgradBn {
    int G, seq, S;
    const float* act; const float* mean; const float* rstd;
    double* stats;
};

// Direct back end (conv_direct.hip).  Each returns 1 when it handled the call, 0 when the
// geometry is not one it is instantiated for (caller falls through to the GEMM back end), or a
// negative D2P_E* / hipError code.
int d2p_conv_direct_fwd(const ConvGeom& g, const void* x, int x_is_u8, const float* w,
                        const float* bias, int act, float* y, hipStream_t st, const ConvBnFold* bn = nullptr);
int d2p_conv_direct_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st,
                          const ConvDgradBn* bn = nullptr);
int d2p_conv_direct_wgrad(const ConvGeom& g, const void* x, int x_is_u8, const float* dy, float* dw,
                          void* ws, size_t ws_bytes, hipStream_t st, const ConvBnFold* bn = nullptr);
size_t d2p_conv_direct_wgrad_ws(const ConvGeom& g);
void d2p_conv_direct_enable(int fwd, int dgrad, int wgrad);

// Whole-frame back end (conv_frames.hip): same return convention; tried first when the
// direction's selector is 2 (the default).
int d2p_conv_frames_fwd(const ConvGeom& g, const void* x, int x_is_u8, const float* w, const float* bias,
                        int act, float* y, hipStream_t st);
int d2p_conv_frames_wgrad(const ConvGeom& g, const void* x, int x_is_u8, const float* dy, float* dw,
                          void* ws, size_t ws_bytes, hipStream_t st);
size_t d2p_conv_frames_wgrad_ws(const ConvGeom& g);
void d2p_conv_frames_tune(int tiles_per_wave);

// Row-strip back end (conv_rows.hip): weight gradients of the narrow layers of large frames.
int d2p_conv_rows_wgrad(const ConvGeom& g, const void* x, int x_is_u8, const float* dy, float* dw, void* ws,
                        size_t ws_bytes, hipStream_t st, const ConvBnFold* bn = nullptr);
size_t d2p_conv_rows_wgrad_ws(const ConvGeom& g);
int d2p_conv_rows_wgrad_bnbwd(const ConvGeom& g, const void* x, int x_is_u8, const float* act, const float* dy,
                              const float* coef, int G, int seq, float* dw, float* dbias, void* ws, size_t ws_bytes,
                              hipStream_t st);
int d2p_conv_rows_fwd(const ConvGeom& g, const void* x, int x_is_u8, const float* w, const float* bias, int act,
                      float* y, hipStream_t st, const ConvBnFold* bn = nullptr);
void d2p_conv_rows_fwd_tune(int workgroups);
int d2p_conv_rows_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st,
                        const ConvDgradBn* bn = nullptr);
int d2p_conv_rows_dgrad_slices(const ConvGeom& g, int G, int seq);
void d2p_conv_rows_dgrad_tune(int workgroups);
void d2p_conv_rows_tune(int wgrad_workgroups);
void d2p_conv_frames_wgrad_cap(int cap);

// Wide back end (conv_wide.hip, round 6): the 48-output-channel layers (32 -> 48, 48 -> 48), filter in LDS; same return
// convention.  bn (optional): statistics out and / or the input read through the previous layer's batch-norm apply.
int d2p_conv_wide_fwd(const ConvGeom& g, const void* x, int x_is_u8, const float* w, const float* bias, int act, float* y,
                      hipStream_t st, const ConvBnFold* bn = nullptr);
int d2p_conv_wide_bn_slices(const ConvGeom& g, int G, int seq);
int d2p_conv_wide_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st,
                        const ConvDgradBn* bn = nullptr);
int d2p_conv_wide_dgrad_bn_slices(const ConvGeom& g, int G, int seq);
void d2p_conv_wide_set_dgrad_1632(int on);
int d2p_conv_wide_fwd2(const ConvGeom& g, const void* x, int x_is_u8, const float* w, const float* bias, int act, float* y,
                       hipStream_t st, const ConvBnFold* bn = nullptr);
int d2p_conv_wide_fwd2_bn_slices(const ConvGeom& g, int G, int seq);
void d2p_conv_wide_set_fwd2(int on);
int d2p_conv_wide_wgrad(const ConvGeom& g, const void* x, int x_is_u8, const float* dy, float* dw, void* ws, size_t ws_bytes,
                        hipStream_t st, const ConvBnFold* bn = nullptr);
size_t d2p_conv_wide_wgrad_ws(const ConvGeom& g);

// floor(n / d) for 0 <= n < 2^31 and d >= 1 by multiply-high + shift (exact: m = ceil(2^(31+s) / d),
// s = ceil(log2 d)).  An integer division costs ~25 VALU instructions on gfx950; the implicit-GEMM
// loaders decompose a row / column index several times per 16-byte load (VALU : MFMA was 10-38 : 1).
struct D2pDiv {
    unsigned int m;
    int s;
};
static inline D2pDiv d2p_make_div(int d) {
    D2pDiv f;
    f.s = 0;
    while ((1L << f.s) < d) ++f.s;
    const unsigned long long num = 1ULL << (31 + f.s);
    f.m = (unsigned int)((num + (unsigned long long)d - 1) / (unsigned long long)d);
    return f;
}
#if defined(__HIPCC__)
__device__ __forceinline__ int d2p_div(int n, D2pDiv f) {
    return (int)((unsigned int)(((unsigned long long)(unsigned int)n * f.m) >> 31) >> f.s);
}
#endif
