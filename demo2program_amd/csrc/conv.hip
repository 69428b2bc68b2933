// K1: NHWC 3x3 stride-2 TF-"SAME" convolution as implicit-im2col fp32 MFMA GEMM
// (include/d2p.h).  Replaces slim.conv2d at models/ops.py:30.
//
// GEMM views (rows m = (n, oy, ox) over N*Ho*Wo output pixels, kk = (ky, kx, c)):
//   fwd   : Y[m, co]   = sum_kk  col(x)[m, kk] * W[kk, co]            (+bias, lrelu)
//   wgrad : dW[kk, co] = sum_m   col(x)[m, kk] * dY[m, co]            (split-K over m)
//   dgrad : dX[p, c]   = sum_(tap,co) colT(dY)[p, (tap,co)] * W[tap, c, co]
// The im2col matrix is never materialised: the A-operand loader gathers NHWC pixels
// (4 consecutive channels per 16-byte load when Cin % 4 == 0) straight into the LDS tile.
// TF SAME padding for k=3,s=2: out = ceil(n/2), pad_total = max((out-1)*2+3-n, 0),
// pad_before = pad_total/2  => even n: (0,1); odd n: (1,1).   
#include "gemm_core.h"

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

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const uint8_t* p) { return (float)*p; }
__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ld4(const uint8_t* p, float (&v)[4]) {
    uint32_t t = *reinterpret_cast<const uint32_t*>(p);
    v[0] = (float)(t & 255u); v[1] = (float)((t >> 8) & 255u);
    v[2] = (float)((t >> 16) & 255u); v[3] = (float)(t >> 24);
}

// col(x)[m, kk]; usable in the KCONTIG role (fwd: x=m, k=kk) ...
template <typename T>
struct Im2colElem {
    const T* x;
    ConvGeom g;
    int vec;   // Cin % 4 == 0 and base aligned
    // branch-free variant (vec path only): clamped address + select
    __device__ __forceinline__ bool gather4_fast(int m, int kk, bool ok, float (&v)[4]) const {
        const int howo = g.Ho * g.Wo;
        const int n = m / howo;
        const int rem = m - n * howo;
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        const int tap = kk / g.Cin, c = kk - tap * g.Cin;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = oy * 2 - g.pt + ky, ix = ox * 2 - g.pl + kx;
        ok = ok & (iy >= 0) & (iy < g.H) & (ix >= 0) & (ix < g.W);
        ld4(x + (ok ? (((long)n * g.H + iy) * g.W + ix) * g.Cin + c : 0L), v);
        return ok;
    }
    __device__ __forceinline__ void gather4(int m, int kk, int kklim, float (&v)[4]) const {
        // 4 consecutive kk for one m
        const int howo = g.Ho * g.Wo;
        const int n = m / howo;
        const int rem = m - n * howo;
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        const int iy0 = oy * 2 - g.pt, ix0 = ox * 2 - g.pl;
        if (vec && kk + 3 < kklim) {
            const int tap = kk / g.Cin, c = kk - tap * g.Cin;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int iy = iy0 + ky, ix = ix0 + kx;
            if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                ld4(x + (((long)n * g.H + iy) * g.W + ix) * g.Cin + c, v);
            else
                v[0] = v[1] = v[2] = v[3] = 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k1 = kk + j;
                float val = 0.f;
                if (k1 < kklim) {
                    const int tap = k1 / g.Cin, c = k1 - tap * g.Cin;
                    const int ky = tap / 3, kx = tap - ky * 3;
                    const int iy = iy0 + ky, ix = ix0 + kx;
                    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                        val = ld1(x + (((long)n * g.H + iy) * g.W + ix) * g.Cin + c);
                }
                v[j] = val;
            }
        }
    }
};

template <typename T>
struct Im2colKC {   // fwd A operand: x = m (rows), k = kk
    static constexpr bool KCONTIG = true;
    Im2colElem<T> e;
    int Mrows;
    bool fast_ok(int K) const { return e.vec && Mrows > 0 && K >= 4; }
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        if (FAST) return e.gather4_fast(x, k, (x < Mrows) & (k < klim), v);
        if (x >= Mrows) { v[0] = v[1] = v[2] = v[3] = 0.f; return true; }
        e.gather4(x, k, klim, v);
        return true;
    }
};

template <typename T>
struct Im2colXC {   // wgrad A operand (A^T·B form): x = kk (output rows), k = m (reduction)
    static constexpr bool KCONTIG = false;
    Im2colElem<T> e;
    int KK;         // 9*Cin
    bool fast_ok(int K) const { return e.vec && K > 0; }
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        if (FAST) return e.gather4_fast(k, x, (x < KK) & (k < klim), v);
        if (k >= klim) { v[0] = v[1] = v[2] = v[3] = 0.f; return true; }
        e.gather4(k, x, KK, v);
        return true;
    }
};

// dgrad A operand: rows p = (n, iy, ix) input pixels; k = (tap, co).
// value = dY[n, oy, ox, co] where oy*2 + ky - pt == iy (needs parity match and range).
struct ColTKC {
    static constexpr bool KCONTIG = true;
    const float* dy;
    ConvGeom g;
    int Prows;
    bool fast_ok(int K) const { return false; }   // parity test keeps this one on the guarded path
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        v[0] = v[1] = v[2] = v[3] = 0.f;
        if (x >= Prows || k >= klim) return true;
        const int hw = g.H * g.W;
        const int n = x / hw;
        const int rem = x - n * hw;
        const int iy = rem / g.W, ix = rem - iy * g.W;
        // Cout % 4 == 0 is required by the entry point, so 4 consecutive k share a tap
        const int tap = k / g.Cout, co = k - tap * g.Cout;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int ty = iy + g.pt - ky, tx = ix + g.pl - kx;
        if (ty < 0 || tx < 0 || (ty & 1) || (tx & 1)) return true;
        const int oy = ty >> 1, ox = tx >> 1;
        if (oy >= g.Ho || ox >= g.Wo) return true;
        ld4(dy + (((long)n * g.Ho + oy) * g.Wo + ox) * g.Cout + co, v);
        return true;
    }
};

// dgrad B operand: columns x = c (input channel), k = (tap, co): W[tap, c, co]
struct WDgradKC {
    static constexpr bool KCONTIG = true;
    const float* w;
    int Cin, Cout;
    bool fast_ok(int K) const { return false; }
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        v[0] = v[1] = v[2] = v[3] = 0.f;
        if (x >= Cin || k >= klim) return true;
        const int tap = k / Cout, co = k - tap * Cout;
        ld4(w + ((long)tap * Cin + x) * Cout + co, v);
        return true;
    }
};

static int check_conv(int N, int H, int W, int Cin, int Cout) {
    D2P_REQUIRE(N >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, D2P_EINVAL,
                "conv: bad geometry N=%d H=%d W=%d Cin=%d Cout=%d", N, H, W, Cin, Cout);
    D2P_REQUIRE((long)N * H * W * (Cin > Cout ? Cin : Cout) < (1L << 31), D2P_EINVAL,
                "conv: tensor too large for 32-bit row indexing");
    return D2P_OK;
}

extern "C" size_t d2p_conv_ws_bytes(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    return d2p_plan_ws_bytes(9 * Cin, Cout, N * g.Ho * g.Wo);
}

extern "C" int d2p_conv2d_nhwc_s2_same_fwd(int N, int H, int W, int Cin, int Cout, const void* x,
                                           int x_is_u8, const float* w, const float* bias, int act,
                                           float* y, d2p_stream_t stream) {
    int rc = check_conv(N, H, W, Cin, Cout);
    if (rc) return rc;
    if (N == 0) return D2P_OK;
    D2P_REQUIRE(x && w && y, D2P_EINVAL, "conv fwd: null pointer");
    D2P_REQUIRE(act == 0 || act == 1, D2P_EINVAL, "conv fwd: unknown act %d", act);
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    const int M = N * g.Ho * g.Wo, K = 9 * Cin;
    const int vecx = (Cin % 4 == 0) && (((uintptr_t)x & 15) == 0);
    DenseXC bl{w, Cout, Cout, (Cout % 4 == 0) && (((uintptr_t)w & 15) == 0)};
    EpiDense ep{y, Cout, bias, act, 0};
    if (x_is_u8) {
        Im2colKC<uint8_t> al{{(const uint8_t*)x, g, (Cin % 4 == 0) && (((uintptr_t)x & 3) == 0)}, M};
        return d2p_launch_gemm(al, bl, ep, M, Cout, K, nullptr, 0, as_stream(stream), "conv_fwd_u8", D2P_PROF_CONV);
    }
    Im2colKC<float> al{{(const float*)x, g, vecx}, M};
    return d2p_launch_gemm(al, bl, ep, M, Cout, K, nullptr, 0, as_stream(stream), "conv_fwd", D2P_PROF_CONV);
}

extern "C" int d2p_conv2d_nhwc_s2_same_wgrad(int N, int H, int W, int Cin, int Cout, const void* x,
                                             int x_is_u8, const float* dy, float* dw, void* ws,
                                             size_t ws_bytes, d2p_stream_t stream) {
    int rc = check_conv(N, H, W, Cin, Cout);
    if (rc) return rc;
    D2P_REQUIRE(dw && (N == 0 || (x && dy)), D2P_EINVAL, "conv wgrad: null pointer");
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    const int Mred = N * g.Ho * g.Wo, KK = 9 * Cin;
    const int vecx = (Cin % 4 == 0) && (((uintptr_t)x & 15) == 0);
    DenseXC bl{dy, Cout, Cout, (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0)};
    EpiDense ep{dw, Cout, nullptr, 0, 0};
    if (x_is_u8) {
        Im2colXC<uint8_t> al{{(const uint8_t*)x, g, (Cin % 4 == 0) && (((uintptr_t)x & 3) == 0)}, KK};
        return d2p_launch_gemm(al, bl, ep, KK, Cout, Mred, ws, ws_bytes, as_stream(stream), "conv_wgrad_u8", D2P_PROF_CONV);
    }
    Im2colXC<float> al{{(const float*)x, g, vecx}, KK};
    return d2p_launch_gemm(al, bl, ep, KK, Cout, Mred, ws, ws_bytes, as_stream(stream), "conv_wgrad", D2P_PROF_CONV);
}

extern "C" int d2p_conv2d_nhwc_s2_same_dgrad(int N, int H, int W, int Cin, int Cout,
                                             const float* dy, const float* w, float* dx,
                                             d2p_stream_t stream) {
    int rc = check_conv(N, H, W, Cin, Cout);
    if (rc) return rc;
    if (N == 0) return D2P_OK;
    D2P_REQUIRE(dy && w && dx, D2P_EINVAL, "conv dgrad: null pointer");
    D2P_REQUIRE(Cout % 4 == 0 && (((uintptr_t)dy & 15) == 0) && (((uintptr_t)w & 15) == 0), D2P_EALIGN,
                "conv dgrad: needs Cout %% 4 == 0 and 16-byte aligned dy/w (Cout=%d)", Cout);
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    const int P = N * H * W, K = 9 * Cout;
    ColTKC al{dy, g, P};
    WDgradKC bl{w, Cin, Cout};
    EpiDense ep{dx, Cin, nullptr, 0, 0};
    return d2p_launch_gemm(al, bl, ep, P, Cin, K, nullptr, 0, as_stream(stream), "conv_dgrad", D2P_PROF_CONV);
}
