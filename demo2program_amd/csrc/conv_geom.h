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

// Direct back end (conv_direct.hip).  Each returns 1 when it handled the call, 0 when the
// geometry is not one it is instantiated for (caller falls through to the GEMM back end), or a
// negative D2P_E* / hipError code.
int d2p_conv_direct_fwd(const ConvGeom& g, const void* x, int x_is_u8, const float* w,
                        const float* bias, int act, float* y, hipStream_t st);
int d2p_conv_direct_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st);
int d2p_conv_direct_wgrad(const ConvGeom& g, const void* x, int x_is_u8, const float* dy, float* dw,
                          void* ws, size_t ws_bytes, hipStream_t st);
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
                        size_t ws_bytes, hipStream_t st);
size_t d2p_conv_rows_wgrad_ws(const ConvGeom& g);
int d2p_conv_rows_fwd(const ConvGeom& g, const void* x, int x_is_u8, const float* w, const float* bias, int act,
                      float* y, hipStream_t st);
void d2p_conv_rows_fwd_tune(int workgroups);
int d2p_conv_rows_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st);
void d2p_conv_rows_dgrad_tune(int workgroups);
void d2p_conv_rows_tune(int wgrad_workgroups);
void d2p_conv_frames_wgrad_cap(int cap);
