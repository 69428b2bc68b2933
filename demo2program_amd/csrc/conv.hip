// K1: NHWC 3x3 stride-2 TF-"SAME" convolution as implicit-im2col fp32 MFMA GEMM
// (include/d2p.h).  Replaces slim.conv2d at models/ops.py:30.
//
// GEMM views (rows m = (n, oy, ox) over N*Ho*Wo output pixels, kk = (ky, kx, c)):
//   fwd   : Y[m, co]   = sum_kk  col(x)[m, kk] * W[kk, co]            (+bias, lrelu)
//   wgrad : dW[kk, co] = sum_m   col(x)[m, kk] * dY[m, co]            (split-K over m)
//   dgrad : dX[p, c]   = sum_(tap,co) colT(dY)[p, (tap,co)] * W[tap, c, co]
// The im2col matrix is never materialised: the A-operand loader gathers NHWC pixels
// (4 consecutive channels per 16-byte load when Cin % 4 == 0) straight into the LDS tile.
// Geometry and SAME padding: conv_geom.h.  Small-channel layers (Cout <= 32, and the Karel
// 2x2 -> 1x1 layer) are served by the direct kernels in conv_direct.hip; everything else, and
// every call when those are disabled, runs here.
#include "gemm_core.h"
#include "conv_geom.h"

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
    D2pDiv d_howo, d_wo, d_cin;
    static Im2colElem make(const T* x, const ConvGeom& g, int vec) {
        return Im2colElem{x, g, vec, d2p_make_div(g.Ho * g.Wo), d2p_make_div(g.Wo), d2p_make_div(g.Cin)};
    }
    // branch-free variant (vec path only): clamped address + select
    __device__ __forceinline__ bool gather4_fast(int m, int kk, bool ok, float (&v)[4]) const {
        const int howo = g.Ho * g.Wo;
        const int n = d2p_div(m, d_howo);
        const int rem = m - n * howo;
        const int oy = d2p_div(rem, d_wo), ox = rem - oy * g.Wo;
        const int tap = d2p_div(kk, d_cin), c = kk - tap * g.Cin;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = oy * 2 - g.pt + ky, ix = ox * 2 - g.pl + kx;
        ok = ok & (iy >= 0) & (iy < g.H) & (ix >= 0) & (ix < g.W);
        // 32-bit offsets: check_conv bounds every tensor below 2^31 elements
        ld4(x + (ok ? ((n * g.H + iy) * g.W + ix) * g.Cin + c : 0), v);
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

// ---- dgrad by output-parity classes -----------------------------------------------------
// With stride 2 an input pixel (iy, ix) only receives taps whose parity matches:
// ky == (iy + pt) mod 2, so rows with (iy+pt) even see ky in {0, 2}, odd rows ky = 1 (same in
// x).  The 4 classes (qy, qx) have 4 / 2 / 2 / 1 valid taps: 2.25 per pixel instead of 9.
// Each class is its own exact GEMM: rows m = (n, j, i) with iy = iy0 + 2j, ix = ix0 + 2i;
// k = (t, co), t = ty*ntx + tx, ky = qy + 2*ty, kx = qx + 2*tx; oy = j + (iy0+pt-ky)/2.
struct DgradClass {
    int qy, qx, nty, ntx, iy0, ix0, Hc, Wc;
};

struct DgradAKC {   // A operand: colT(dY) restricted to one parity class
    static constexpr bool KCONTIG = true;
    const float* dy;
    ConvGeom g;
    DgradClass c;
    int Mrows;
    int vec;
    D2pDiv d_hw, d_wc, d_cout, d_ntx;
    bool fast_ok(int K) const { return vec && Mrows > 0 && K >= 4; }
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        bool ok = (x < Mrows) & (k < klim);
        const int hw = c.Hc * c.Wc;
        const int n = d2p_div(x, d_hw);
        const int rem = x - n * hw;
        const int j = d2p_div(rem, d_wc), i = rem - j * c.Wc;
        const int t = d2p_div(k, d_cout), co = k - t * g.Cout;
        const int ty = d2p_div(t, d_ntx), tx = t - ty * c.ntx;
        const int oy = j + ((c.iy0 + g.pt - (c.qy + 2 * ty)) >> 1);
        const int ox = i + ((c.ix0 + g.pl - (c.qx + 2 * tx)) >> 1);
        ok = ok & (oy >= 0) & (oy < g.Ho) & (ox >= 0) & (ox < g.Wo);
        ld4(dy + (ok ? ((n * g.Ho + oy) * g.Wo + ox) * g.Cout + co : 0), v);
        if (!FAST && !ok) v[0] = v[1] = v[2] = v[3] = 0.f;
        return FAST ? ok : true;
    }
};

struct DgradBKC {   // B operand: columns x = c (input channel), k = (t, co): W[ky, kx, c, co]
    static constexpr bool KCONTIG = true;
    const float* w;
    int Cin, Cout;
    DgradClass c;
    int vec;
    D2pDiv d_cout, d_ntx;
    bool fast_ok(int K) const { return vec && Cin > 0 && K >= 4; }
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        const bool ok = (x < Cin) & (k < klim);
        const int t = d2p_div(k, d_cout), co = k - t * Cout;
        const int ty = d2p_div(t, d_ntx), tx = t - ty * c.ntx;
        const int tap = (c.qy + 2 * ty) * 3 + (c.qx + 2 * tx);
        ld4(w + (ok ? (tap * Cin + x) * Cout + co : 0), v);
        if (!FAST && !ok) v[0] = v[1] = v[2] = v[3] = 0.f;
        return FAST ? ok : true;
    }
};

struct EpiDgrad {   // scatter the class rows back to their stride-2 pixel positions
    float* dx;
    int H, W, Cin;
    DgradClass c;
    D2pDiv d_hw, d_wc;
    __device__ __forceinline__ float col_value(int) const { return 0.f; }
    __device__ __forceinline__ bool has_c() const { return false; }
    __device__ __forceinline__ float c_value(int, int) const { return 0.f; }
    __device__ __forceinline__ void store(int row, int col, float v) const {
        const int hw = c.Hc * c.Wc;
        const int n = d2p_div(row, d_hw);
        const int rem = row - n * hw;
        const int j = d2p_div(rem, d_wc), i = rem - j * c.Wc;
        dx[((n * H + c.iy0 + 2 * j) * W + c.ix0 + 2 * i) * Cin + col] = v;
    }
    __device__ __forceinline__ void operator()(int row, int col, float v) const { store(row, col, v); }
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
    const size_t a = d2p_plan_ws_bytes(9 * Cin, Cout, N * g.Ho * g.Wo), b = d2p_conv_direct_wgrad_ws(g);
    return a > b ? a : b;
}

extern "C" int d2p_conv_set_direct(int fwd, int dgrad, int wgrad) {
    d2p_conv_wide_set_dgrad_1632(dgrad != 3);            // (dgrad = 3: as 2, with the 16 -> 32 layer on the row-strip kernel)
    if (dgrad == 3) dgrad = 2;
    d2p_conv_wide_set_fwd2(fwd != 3);                    // (fwd = 3: as 2, with the 16 -> 32 layer on the gather kernel)
    if (fwd == 3) fwd = 2;
    d2p_conv_direct_enable(fwd, dgrad, wgrad);
    return D2P_OK;
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
    rc = d2p_conv_direct_fwd(g, x, x_is_u8, w, bias, act, y, as_stream(stream));
    if (rc != 0) return rc < 0 ? rc : D2P_OK;
    const int M = N * g.Ho * g.Wo, K = 9 * Cin;
    const int vecx = (Cin % 4 == 0) && (((uintptr_t)x & 15) == 0);
    DenseXC bl{w, Cout, Cout, (Cout % 4 == 0) && (((uintptr_t)w & 15) == 0)};
    EpiDense ep{y, Cout, bias, act, 0};
    if (x_is_u8) {
        Im2colKC<uint8_t> al{Im2colElem<uint8_t>::make((const uint8_t*)x, g, (Cin % 4 == 0) && (((uintptr_t)x & 3) == 0)), M};
        return d2p_launch_gemm(al, bl, ep, M, Cout, K, nullptr, 0, as_stream(stream), "conv_fwd_u8", D2P_PROF_CONV);
    }
    Im2colKC<float> al{Im2colElem<float>::make((const float*)x, g, vecx), M};
    return d2p_launch_gemm(al, bl, ep, M, Cout, K, nullptr, 0, as_stream(stream), "conv_fwd", D2P_PROF_CONV);
}

// ---- batch norm folded into the conv launches (round 5; ConvBnFold in conv_geom.h) ------------------------------------
// slices per demonstration index the folding forward launch of this geometry writes statistics for; 0: not taken
extern "C" int d2p_conv_bn_slices(int N, int H, int W, int Cin, int Cout, int G, int seq) {
    if (N <= 0 || G <= 0 || seq <= 0 || N % (G * seq) != 0) return 0;
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    if (Cout == 48) return d2p_conv_wide_bn_slices(g, G, seq);      // (the 48-channel layers: conv_wide.hip)
    {
        const int S2 = d2p_conv_wide_fwd2_bn_slices(g, G, seq);     // (the large 16 -> 32 layer in block form)
        if (S2 > 0) return S2;
    }
    long units;                                  // work units of one index: strips (first layer) or 16-pixel tiles
    if (Cin == 4 && Cout == 16 && W == 80) units = (long)N / G * g.Ho;
    else if (Cin == 16 && Cout == 32 && H * W >= 400 && (seq * g.Ho * g.Wo) % 16 == 0) units = (long)N / G * g.Ho * g.Wo / 16;
    else return 0;
    // what the folding launches themselves require of the geometry (conv_direct.hip direct_bn_ok: whole 16-pixel tiles per
    // sequence, four at least; 32-bit element offsets incl. the pad pixels behind the tensor) -- answered HERE, so that a
    // caller that asks first never meets D2P_EINVAL from the launch
    if ((seq * g.Ho * g.Wo) % 16 != 0 || seq * g.Ho * g.Wo / 16 < 4) return 0;
    if ((size_t)N * H * W * Cin + (size_t)G * Cin >= (1ull << 32)) return 0;
    // ~2048 workgroups (8 per CU) of >= 16 units each, as the plain launches of these kernels
    long S = 2048 / G;
    if (S * 16 > units) S = units / 16;
    return (int)(S < 1 ? 1 : S);
}
// 1 when this layer's forward AND weight-gradient launches take their input through the previous layer's batch-norm apply
// (ConvBnFold::in_scale / in_shift: x = that layer's pre-norm activation with its G pad pixels behind it), so that the
// normalised tensor need not be written: the 16 -> 32 layer of the large frames (row-strip / gather kernels) and the
// 48-channel layers (conv_wide.hip)
extern "C" int d2p_conv_bn_affine_ok(int N, int H, int W, int Cin, int Cout, int G, int seq) {
    if (d2p_conv_bn_slices(N, H, W, Cin, Cout, G, seq) <= 0) return 0;
    if (Cin == 16 && Cout == 32) return 1;
    if (Cout == 48) {
        ConvGeom g = make_geom(N, H, W, Cin, Cout);
        return d2p_conv_wide_wgrad_ws(g) > 0 ? 1 : 0;
    }
    return 0;
}
extern "C" int d2p_conv2d_nhwc_s2_same_fwd_bn(int N, int H, int W, int Cin, int Cout, const void* x, int x_is_u8,
                                              const float* w, const float* bias, int act, float* y, int G, int seq,
                                              const float* in_scale, const float* in_shift, double* stats, int S,
                                              d2p_stream_t stream) {
    int rc = check_conv(N, H, W, Cin, Cout);
    if (rc) return rc;
    D2P_REQUIRE(x && w && y && stats, D2P_EINVAL, "conv fwd (bn): null pointer");
    D2P_REQUIRE(act == 0 || act == 1, D2P_EINVAL, "conv fwd (bn): unknown act %d", act);
    D2P_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), D2P_EINVAL, "conv fwd (bn): in_scale and in_shift go together");
    D2P_REQUIRE(S >= 1 && S == d2p_conv_bn_slices(N, H, W, Cin, Cout, G, seq), D2P_EINVAL,
                "conv fwd (bn): geometry not taken, or S = %d is not d2p_conv_bn_slices()", S);
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    ConvBnFold bn{G, seq, S, in_scale, in_shift, stats};
    rc = d2p_conv_direct_fwd(g, x, x_is_u8, w, bias, act, y, as_stream(stream), &bn);
    if (rc < 0) return rc;
    D2P_REQUIRE(rc == 1, D2P_EINVAL, "conv fwd (bn): this combination (Cin=%d, affine input %d, u8 %d) has no folding kernel",
                Cin, in_scale != nullptr, x_is_u8);
    return D2P_OK;
}
extern "C" int d2p_conv2d_nhwc_s2_same_wgrad_bn(int N, int H, int W, int Cin, int Cout, const void* x, int x_is_u8,
                                                const float* dy, float* dw, int G, int seq, const float* in_scale,
                                                const float* in_shift, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    int rc = check_conv(N, H, W, Cin, Cout);
    if (rc) return rc;
    D2P_REQUIRE(dw && x && dy && in_scale && in_shift, D2P_EINVAL, "conv wgrad (bn): null pointer");
    D2P_REQUIRE(G >= 1 && seq >= 1 && N % (G * seq) == 0, D2P_EINVAL, "conv wgrad (bn): N=%d is not a multiple of G*seq", N);
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    ConvBnFold bn{G, seq, 0, in_scale, in_shift, nullptr};
    rc = d2p_conv_direct_wgrad(g, x, x_is_u8, dy, dw, ws, ws_bytes, as_stream(stream), &bn);
    if (rc < 0) return rc;
    D2P_REQUIRE(rc == 1, D2P_EINVAL, "conv wgrad (bn): no folding kernel for Cin=%d Cout=%d W=%d", Cin, Cout, W);
    return D2P_OK;
}

// the input gradient of layer l+1 that also leaves layer l's batch-norm-backward partial sums (ConvDgradBn)
extern "C" int d2p_conv_dgrad_bn_slices(int N, int H, int W, int Cin, int Cout, int G, int seq) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    const int S = d2p_conv_wide_dgrad_bn_slices(g, G, seq);      // (the same order as d2p_conv_direct_dgrad tries the kernels)
    return S > 0 ? S : d2p_conv_rows_dgrad_slices(g, G, seq);
}
extern "C" int d2p_conv2d_nhwc_s2_same_dgrad_bn(int N, int H, int W, int Cin, int Cout, const float* dy, const float* w,
                                                float* dx, const float* act, const float* mean, const float* rstd, int G,
                                                int seq, double* stats, int S, d2p_stream_t stream) {
    int rc = check_conv(N, H, W, Cin, Cout);
    if (rc) return rc;
    D2P_REQUIRE(dy && w && dx && act && mean && rstd && stats, D2P_EINVAL, "conv dgrad (bn): null pointer");
    D2P_REQUIRE(S >= 1 && S == d2p_conv_dgrad_bn_slices(N, H, W, Cin, Cout, G, seq), D2P_EINVAL,
                "conv dgrad (bn): geometry not taken, or S = %d is not d2p_conv_dgrad_bn_slices()", S);
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    ConvDgradBn bn{G, seq, S, act, mean, rstd, stats};
    rc = d2p_conv_direct_dgrad(g, dy, w, dx, as_stream(stream), &bn);
    if (rc < 0) return rc;
    D2P_REQUIRE(rc == 1, D2P_EINVAL, "conv dgrad (bn): no folding kernel (alignment?) for Cin=%d Cout=%d W=%d", Cin, Cout, W);
    return D2P_OK;
}

// the weight gradient of a layer whose batch-norm backward is folded in: dy = gradient w.r.t. the layer's batch-norm
// output, act = its pre-norm activation, coef from d2p_bn_group_bwd_coef; also the bias gradient.  D2P_EINVAL when the
// geometry has no such kernel (d2p_conv_bnbwd_ok).
extern "C" int d2p_conv_bnbwd_ok(int N, int H, int W, int Cin, int Cout) {
    return N > 0 && Cin == 4 && Cout == 16 && W == 80 && H > 0;
}
extern "C" int d2p_conv2d_nhwc_s2_same_wgrad_bnbwd(int N, int H, int W, int Cin, int Cout, const void* x, int x_is_u8,
                                                   const float* act, const float* dy, const float* coef, int G, int seq,
                                                   float* dw, float* dbias, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    int rc = check_conv(N, H, W, Cin, Cout);
    if (rc) return rc;
    D2P_REQUIRE(x && act && dy && coef && dw, D2P_EINVAL, "conv wgrad (bn backward): null pointer");
    D2P_REQUIRE(G >= 1 && seq >= 1 && N % (G * seq) == 0, D2P_EINVAL, "conv wgrad (bn backward): N=%d is not a multiple of G*seq", N);
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    rc = d2p_conv_rows_wgrad_bnbwd(g, x, x_is_u8, act, dy, coef, G, seq, dw, dbias, ws, ws_bytes, as_stream(stream));
    if (rc < 0) return rc;
    D2P_REQUIRE(rc == 1, D2P_EINVAL, "conv wgrad (bn backward): no folding kernel for Cin=%d Cout=%d W=%d", Cin, Cout, W);
    return D2P_OK;
}

extern "C" int d2p_conv2d_nhwc_s2_same_wgrad(int N, int H, int W, int Cin, int Cout, const void* x,
                                             int x_is_u8, const float* dy, float* dw, void* ws,
                                             size_t ws_bytes, d2p_stream_t stream) {
    int rc = check_conv(N, H, W, Cin, Cout);
    if (rc) return rc;
    D2P_REQUIRE(dw && (N == 0 || (x && dy)), D2P_EINVAL, "conv wgrad: null pointer");
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    rc = d2p_conv_direct_wgrad(g, x, x_is_u8, dy, dw, ws, ws_bytes, as_stream(stream));
    if (rc != 0) return rc < 0 ? rc : D2P_OK;
    const int Mred = N * g.Ho * g.Wo, KK = 9 * Cin;
    const int vecx = (Cin % 4 == 0) && (((uintptr_t)x & 15) == 0);
    DenseXC bl{dy, Cout, Cout, (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0)};
    EpiDense ep{dw, Cout, nullptr, 0, 0};
    if (x_is_u8) {
        Im2colXC<uint8_t> al{Im2colElem<uint8_t>::make((const uint8_t*)x, g, (Cin % 4 == 0) && (((uintptr_t)x & 3) == 0)), KK};
        return d2p_launch_gemm(al, bl, ep, KK, Cout, Mred, ws, ws_bytes, as_stream(stream), "conv_wgrad_u8", D2P_PROF_CONV);
    }
    Im2colXC<float> al{Im2colElem<float>::make((const float*)x, g, vecx), KK};
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
    // (Running the four classes chunk-by-chunk over frames, to keep dY in the Infinity Cache, was
    // measured slower: the class GEMMs are MFMA-bound on the half-empty N=16 tile, not HBM-bound.)
    ConvGeom g = make_geom(N, H, W, Cin, Cout);
    rc = d2p_conv_direct_dgrad(g, dy, w, dx, as_stream(stream));
    if (rc != 0) return rc < 0 ? rc : D2P_OK;
    for (int qy = 0; qy < 2; ++qy)
        for (int qx = 0; qx < 2; ++qx) {
            DgradClass c;
            c.qy = qy; c.qx = qx;
            c.nty = qy == 0 ? 2 : 1;
            c.ntx = qx == 0 ? 2 : 1;
            c.iy0 = qy ^ (g.pt & 1);             // (iy + pt) & 1 == qy
            c.ix0 = qx ^ (g.pl & 1);
            c.Hc = (H - c.iy0 + 1) / 2;
            c.Wc = (W - c.ix0 + 1) / 2;
            if (c.Hc <= 0 || c.Wc <= 0) continue;
            const int Mc = N * c.Hc * c.Wc, K = c.nty * c.ntx * Cout;
            const D2pDiv d_hw = d2p_make_div(c.Hc * c.Wc), d_wc = d2p_make_div(c.Wc),
                         d_cout = d2p_make_div(Cout), d_ntx = d2p_make_div(c.ntx);
            DgradAKC al{dy, g, c, Mc, 1, d_hw, d_wc, d_cout, d_ntx};
            DgradBKC bl{w, Cin, Cout, c, 1, d_cout, d_ntx};
            EpiDgrad ep{dx, H, W, Cin, c, d_hw, d_wc};
            rc = d2p_launch_gemm(al, bl, ep, Mc, Cin, K, nullptr, 0, as_stream(stream), "conv_dgrad",
                                 D2P_PROF_CONV);
            if (rc) return rc;
        }
    return D2P_OK;
}
