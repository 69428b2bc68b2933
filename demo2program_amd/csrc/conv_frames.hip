// K1 (whole-frame back end): the Karel demonstration encoder's three layers
// (8x8x16 -> 4x4x16 -> 2x2x32 -> 1x1x48; models/model_full.py:216-231) with LDS-staged input.
//
// The gather-from-global direct kernels (conv_direct.hip) turned out to be bound by the
// texture-address path, not by HBM, MFMA or VALU: a fragment-shaped 16-byte gather touches 16
// different cache lines per 16-lane group (~64-90 TA cycles per wave-instruction, 9 of them per
// 16-pixel tile), which caps conv1 at ~0.35 us per tile per CU whatever else is removed.
// Here a tile is a run of WHOLE frames (16 / (Ho*Wo) of them = 16 output pixels), which is one
// contiguous 4-8 KB piece of the NHWC tensor:
//   * HBM/L2 -> VGPR as full-line, lane-linear 16-byte loads (4-8 per lane per tile),
//   * VGPR -> a wave-private LDS image with a 16-byte pad per pixel (bank spreading),
//   * MFMA B fragments by ds_read_b128 at per-lane offsets computed ONCE per kernel: every tile
//     has the same geometry, so all tap/bounds arithmetic leaves the loop; out-of-image taps
//     point at a zero slot,
//   * filter in registers, 16x16x4 fp32 MFMA, [channel][pixel] result -> 16-byte NHWC stores,
//     exactly as in conv_direct.hip.
// LDS images are private to a wave (no workgroup barrier anywhere); two images in ping-pong so
// the next tile's global loads are in flight during the current tile's MFMAs.
#include "conv_geom.h"
#include "gemm_core.h"
#include "prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define D2P_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int pad_out(int n) { return (n + 1) / 2; }
constexpr int pad_before(int n) {
    int total = (pad_out(n) - 1) * 2 + 3 - n;
    return total < 0 ? 0 : total / 2;
}
constexpr int popcount9(int m) { int c = 0; for (int i = 0; i < 9; ++i) c += (m >> i) & 1; return c; }
constexpr int nth_tap(int m, int n) {
    for (int i = 0; i < 9; ++i) if ((m >> i) & 1) { if (n == 0) return i; --n; }
    return 0;
}
constexpr bool axis_touches(int n, int k) {   // does tap offset k ever land inside an axis of size n?
    for (int o = 0; o < pad_out(n); ++o) { int i = 2 * o - pad_before(n) + k; if (i >= 0 && i < n) return true; }
    return false;
}
constexpr int tap_mask(int H, int W) {
    int m = 0;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx)
            if (axis_touches(H, ky) && axis_touches(W, kx)) m |= 1 << (3 * ky + kx);
    return m;
}

template <int CIN, int COUT, int H, int W>
struct FrameShape {
    static constexpr int Ho = pad_out(H), Wo = pad_out(W), PT = pad_before(H), PL = pad_before(W);
    static constexpr int HW = Ho * Wo;            // output pixels per frame (1, 2, 4, 8 or 16)
    static constexpr int F = 16 / HW;             // frames per tile
    static constexpr int NPIX = F * H * W;        // staged input pixels per tile
    static constexpr int PSF = CIN + 4;           // floats per staged pixel (16-byte pad)
    static constexpr int BUF = NPIX * PSF + 4;    // + zero slot
    static constexpr int CIN_ = CIN;
    static constexpr int CHUNK = NPIX * CIN;      // contiguous input elements per tile
    static constexpr int MASK = tap_mask(H, W);
    static constexpr int CB = CIN / 16, NT = popcount9(MASK), NCH = NT * CB, NB = COUT / 16;
    static constexpr int Q4 = CIN / 4;            // 16-byte pieces per pixel
    static constexpr int NL = NPIX * Q4 / 64;     // pieces per lane per tile
    static_assert(16 % HW == 0 && HW <= 16, "frame must hold a power-of-two number of output pixels");
    static_assert(NPIX * Q4 % 64 == 0, "tile must be a whole number of wave-wide 16-byte loads");
    static constexpr size_t lds_bytes = (size_t)4 * 2 * BUF * sizeof(float);
};

// Register staging of one tile's contiguous input chunk and its scatter into the padded LDS
// image.  float frames: NL 16-byte loads per lane, one ds_write_b128 each.  uint8 frames (the
// dataset's own precision for the first layer: Karel states are booleans): one 16-byte load
// carries 16 channels = 4 float4 stores after widening -- a quarter of the HBM bytes.
template <typename T, class S>
struct Stager;

template <class S>
struct Stager<float, S> {
    static constexpr int N = S::NL;
    f32x4 r[N];
    int soff[N];
    __device__ __forceinline__ void init(int lane) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = i * 64 + lane;
            soff[i] = (j / S::Q4) * S::PSF + (j % S::Q4) * 4;
        }
    }
    __device__ __forceinline__ void load(const float* __restrict__ x, int tile, long total, int lane) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            long e = (long)tile * S::CHUNK + (i * 64 + lane) * 4;
            e = e < total - 4 ? e : total - 4;           // last tile may run past the tensor
            r[i] = *reinterpret_cast<const f32x4*>(x + e);
        }
    }
    __device__ __forceinline__ void store(float* img) const {
#pragma unroll
        for (int i = 0; i < N; ++i) *reinterpret_cast<f32x4*>(img + soff[i]) = r[i];
    }
};

template <class S>
struct Stager<uint8_t, S> {
    static_assert(S::CHUNK % 1024 == 0 && S::CIN_ % 16 == 0, "uint8 staging moves whole 16-channel groups");
    static constexpr int N = S::CHUNK / 1024;            // 16-byte loads per lane
    uint4 r[N];
    int soff[N];
    __device__ __forceinline__ void init(int lane) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int b = (i * 64 + lane) * 16;          // first channel-element of this piece
            soff[i] = (b / S::CIN_) * S::PSF + (b % S::CIN_);
        }
    }
    __device__ __forceinline__ void load(const uint8_t* __restrict__ x, int tile, long total, int lane) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            long e = (long)tile * S::CHUNK + (i * 64 + lane) * 16;
            e = e < total - 16 ? e : total - 16;
            r[i] = *reinterpret_cast<const uint4*>(x + e);
        }
    }
    __device__ __forceinline__ void store(float* img) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t w4[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 v;
                v.x = (float)(w4[k] & 255u); v.y = (float)((w4[k] >> 8) & 255u);
                v.z = (float)((w4[k] >> 16) & 255u); v.w = (float)(w4[k] >> 24);
                *reinterpret_cast<f32x4*>(img + soff[i] + 4 * k) = v;
            }
        }
    }
};

template <int CIN, int COUT, int H, int W, typename T>
__global__ void __launch_bounds__(256)
conv_frames_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                       int act, float* __restrict__ y, int nframes, int ntiles) {
    using S = FrameShape<CIN, COUT, H, W>;
    constexpr int NCH = S::NCH, NB = S::NB, CB = S::CB, PSF = S::PSF, BUF = S::BUF;
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, p = lane & 15, q = lane >> 4, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wid, NW = gridDim.x * 4;
    float* const img0 = lds + (size_t)wid * 2 * BUF;
    float* const img1 = img0 + BUF;
    if (lane == 0) {
        *reinterpret_cast<f32x4*>(img0 + S::NPIX * PSF) = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(img1 + S::NPIX * PSF) = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // per-lane fragment offsets into an image (floats): pixel p of the tile, k-slice q
    int toff[NCH];
    {
        const int f = p / S::HW, r = p % S::HW, oy = r / S::Wo, ox = r % S::Wo;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int tap = nth_tap(S::MASK, ch / CB), ky = tap / 3, kx = tap % 3;
            const int iy = 2 * oy - S::PT + ky, ix = 2 * ox - S::PL + kx;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            toff[ch] = ok ? ((f * H + iy) * W + ix) * PSF + (ch % CB) * 16 + 4 * q : S::NPIX * PSF;
        }
    }
    // staging: the tile's contiguous chunk -> padded image
    Stager<T, S> st;
    st.init(lane);

    // filter -> registers (A operand): wr[ch][j][b] = W[tap, 16 cb + 4q + j][16 b + p]
    float wr[NCH][4][NB];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                wr[ch][j][b] = w[(nth_tap(S::MASK, ch / CB) * CIN + (ch % CB) * 16 + 4 * q + j) * COUT + b * 16 + p];
    f32x4 bv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) bv[b] = bias ? *reinterpret_cast<const f32x4*>(bias + b * 16 + 4 * q)
                                               : f32x4{0.f, 0.f, 0.f, 0.f};

    const long total = (long)nframes * H * W * CIN;   // elements in x (a multiple of 16)
    const int P = nframes * S::HW;
    auto step = [&](int tile, const float* rimg, float* wimg) {
        const int nt = tile + NW;
        st.load(x, nt < ntiles ? nt : ntiles - 1, total, lane);
        f32x4 acc[2][NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[0][b] = acc[1][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(rimg + toff[ch]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[j & 1][b] = D2P_MFMA16(wr[ch][j][b], bb[j], acc[j & 1][b]);
        }
        const int pix = tile * 16 + p;
        if (pix < P) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                f32x4 o = (acc[0][b] + acc[1][b]) + bv[b];
                if (act) { o.x = d2p_lrelu(o.x); o.y = d2p_lrelu(o.y); o.z = d2p_lrelu(o.z); o.w = d2p_lrelu(o.w); }
                *reinterpret_cast<f32x4*>(y + (long)pix * COUT + b * 16 + 4 * q) = o;
            }
        }
        st.store(wimg);
    };

    int tile = wave;
    if (tile < ntiles) {
        st.load(x, tile, total, lane);
        st.store(img0);
    }
    while (tile < ntiles) {
        step(tile, img0, img1);
        tile += NW;
        if (tile >= ntiles) break;
        step(tile, img1, img0);
        tile += NW;
    }
}

int g_frames_tpw = 0;   // 0: automatic

template <int CIN, int COUT, int H, int W, typename T>
int launch_frames_fwd(const ConvGeom& g, const T* x, const float* w, const float* bias, int act, float* y,
                      hipStream_t st) {
    using S = FrameShape<CIN, COUT, H, W>;
    static bool attr_set = false;
    if (!attr_set) {
        D2P_HIP(hipFuncSetAttribute((const void*)conv_frames_fwd_kernel<CIN, COUT, H, W, T>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::lds_bytes));
        attr_set = true;
    }
    const int ntiles = ceil_div(g.N, S::F);
    // measured on MI355X: one tile per wave until 2 waves per SIMD are busy, then grid-stride
    long waves = g_frames_tpw > 0 ? ((long)ntiles + g_frames_tpw - 1) / g_frames_tpw : ntiles;
    if (waves > 2048) waves = 2048;
    const int blocks = ceil_div((int)waves, 4);
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * g.N * S::HW * 9 * CIN * COUT);
    hipLaunchKernelGGL((conv_frames_fwd_kernel<CIN, COUT, H, W, T>), dim3(blocks), dim3(256), S::lds_bytes, st, x, w,
                       bias, act, y, g.N, ntiles);
    D2P_LAUNCH_CHECK("conv_frames_fwd");
    return 1;
}

// ------------------------------------------------------------------------------------------
// wgrad: dW[tap, ci, co] = sum over pixels x[pix@tap, ci] * dY[pix, co].  The reduction index
// of the MFMA is the pixel (4 per instruction), so each wave keeps the WHOLE dW (36-96
// accumulator VGPRs) and walks tiles of 16 pixels.  x is staged through the wave's LDS image
// (one image: the next tile waits in registers while this one is consumed); the A operand is
// ds_read_b32 at per-lane offsets fixed for the kernel; dY (64-byte rows) comes straight from
// global memory.  WAVES waves per workgroup are summed through LDS in a fixed tree, so one
// slab per workgroup reaches the deterministic combine pass.
template <int CIN, int COUT, int H, int W, int WAVES, typename T>
__global__ void __launch_bounds__(WAVES * 64)
conv_frames_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slabs,
                         int nframes, int ntiles) {
    using S = FrameShape<CIN, COUT, H, W>;
    constexpr int CB = S::CB, NT = S::NT, AB = NT * CB, NBO = COUT / 16, PSF = S::PSF;
    constexpr int IMG = S::NPIX * PSF + 4;             // floats per wave image (+ zero slot)
    constexpr int ACC = AB * NBO * 4;                  // accumulator floats per lane
    constexpr int KK = 9 * CIN;
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, c = lane & 15, kq = lane >> 4, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * WAVES + wid, NW = gridDim.x * WAVES;
    float* const img = lds + (size_t)wid * IMG;
    if (lane == 0) *reinterpret_cast<f32x4*>(img + S::NPIX * PSF) = f32x4{0.f, 0.f, 0.f, 0.f};

    // A-operand offsets: MFMA m of a tile reduces over pixels 4m + kq (kq = lane >> 4)
    int toff[AB][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int pi = 4 * m + kq;
        const int f = pi / S::HW, r = pi % S::HW, oy = r / S::Wo, ox = r % S::Wo;
#pragma unroll
        for (int a = 0; a < AB; ++a) {
            const int tap = nth_tap(S::MASK, a / CB), ky = tap / 3, kx = tap % 3;
            const int iy = 2 * oy - S::PT + ky, ix = 2 * ox - S::PL + kx;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            toff[a][m] = ok ? ((f * H + iy) * W + ix) * PSF + (a % CB) * 16 + c : S::NPIX * PSF;
        }
    }
    Stager<T, S> st;
    st.init(lane);

    f32x4 acc[AB][NBO];
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < NBO; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long total = (long)nframes * H * W * CIN;
    const int P = nframes * S::HW;
    auto load_tile = [&](int tile, float (&bv)[4][NBO]) {
        st.load(x, tile, total, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int pix = tile * 16 + 4 * m + kq;
            const int pc = pix < P ? pix : P - 1;
#pragma unroll
            for (int b = 0; b < NBO; ++b) bv[m][b] = dy[(long)pc * COUT + b * 16 + c];
        }
    };

    float bv[4][NBO];
    int tile = wave;
    if (tile < ntiles) load_tile(tile, bv);
    while (tile < ntiles) {
        // stage this tile, then immediately put the next one in flight
        st.store(img);
        float bc[4][NBO];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const bool valid = tile * 16 + 4 * m + kq < P;
#pragma unroll
            for (int b = 0; b < NBO; ++b) bc[m][b] = valid ? bv[m][b] : 0.f;
        }
        const int nt = tile + NW;
        load_tile(nt < ntiles ? nt : ntiles - 1, bv);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int a = 0; a < AB; ++a) {
                const float av = img[toff[a][m]];
#pragma unroll
                for (int b = 0; b < NBO; ++b) acc[a][b] = D2P_MFMA16(av, bc[m][b], acc[a][b]);
            }
        tile = nt;
    }

    // fixed-order tree over the workgroup's waves (scratch reuses the image area)
    __syncthreads();
    float* const red = lds;
#pragma unroll
    for (int step = 1; step < WAVES; step *= 2) {
        if ((wid & (2 * step - 1)) == step) {
            float* dst = red + (size_t)(wid / (2 * step)) * ACC * 64;
#pragma unroll
            for (int a = 0; a < AB; ++a)
#pragma unroll
                for (int b = 0; b < NBO; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[((a * NBO + b) * 4 + r) * 64 + lane] = acc[a][b][r];
        }
        __syncthreads();
        if ((wid & (2 * step - 1)) == 0) {
            const float* src = red + (size_t)(wid / (2 * step)) * ACC * 64;
#pragma unroll
            for (int a = 0; a < AB; ++a)
#pragma unroll
                for (int b = 0; b < NBO; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][b][r] += src[((a * NBO + b) * 4 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (wid == 0) {
        float* slab = slabs + (long)blockIdx.x * KK * COUT;
#pragma unroll
        for (int a = 0; a < AB; ++a)
#pragma unroll
            for (int b = 0; b < NBO; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    slab[(nth_tap(S::MASK, a / CB) * CIN + (a % CB) * 16 + 4 * kq + r) * COUT + b * 16 + c] = acc[a][b][r];
    } else if (wid == 1) {
        // taps that never touch the image contribute exact zeros
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            if (!((S::MASK >> tap) & 1))
                for (int i = lane; i < CIN * COUT / 4; i += 64)
                    *reinterpret_cast<f32x4*>(slabs + (long)blockIdx.x * KK * COUT + tap * CIN * COUT + i * 4) =
                        f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

int g_frames_wgrad_cap = 0;   // 0: per-layer default

template <int CIN, int COUT, int H, int W, int WAVES, int CAP, typename T>
struct WgradLaunch {
    using S = FrameShape<CIN, COUT, H, W>;
    static constexpr int ACC = S::NT * S::CB * (COUT / 16) * 4;
    static constexpr size_t img_bytes = (size_t)WAVES * (S::NPIX * S::PSF + 4) * sizeof(float);
    static constexpr size_t red_bytes = (size_t)(WAVES / 2) * ACC * 64 * sizeof(float);
    static constexpr size_t lds_bytes = img_bytes > red_bytes ? img_bytes : red_bytes;
    static int blocks(int N) {
        const int ntiles = ceil_div(N, S::F);
        int b = ceil_div(ntiles, WAVES);
        const int cap = g_frames_wgrad_cap > 0 ? g_frames_wgrad_cap : CAP;   // slabs written = workgroups
        if (b > cap) b = cap;
        return b < 1 ? 1 : b;
    }
    static int run(const ConvGeom& g, const T* x, const float* dy, float* dw, void* ws, size_t ws_bytes,
                   hipStream_t st) {
        static bool attr_set = false;
        if (!attr_set) {
            D2P_HIP(hipFuncSetAttribute((const void*)conv_frames_wgrad_kernel<CIN, COUT, H, W, WAVES, T>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            attr_set = true;
        }
        const int KK = 9 * CIN;
        const int nb = blocks(g.N);
        D2P_REQUIRE(ws && ws_bytes >= (size_t)nb * KK * COUT * sizeof(float), D2P_EWS,
                    "conv wgrad: workspace too small (%zu bytes)", ws_bytes);
        float* slabs = (float*)ws;
        D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * g.N * S::HW * KK * COUT);
        hipLaunchKernelGGL((conv_frames_wgrad_kernel<CIN, COUT, H, W, WAVES, T>), dim3(nb), dim3(WAVES * 64),
                           lds_bytes, st, x, dy, slabs, g.N, ceil_div(g.N, S::F));
        D2P_LAUNCH_CHECK("conv_frames_wgrad");
        EpiDense ep{dw, COUT, nullptr, 0, 0};
        const long total = (long)KK * COUT;
        if (nb <= 16) {
            hipLaunchKernelGGL((gemm_splitk_reduce_flat_kernel<EpiDense>), dim3((int)((total + 255) / 256)), dim3(256),
                               0, st, ep, slabs, KK, COUT, nb);
        } else {
            hipLaunchKernelGGL((gemm_splitk_reduce_kernel<EpiDense>), dim3((int)((total * 16 + 255) / 256)), dim3(256),
                               0, st, ep, slabs, KK, COUT, nb);
        }
        D2P_LAUNCH_CHECK("conv_frames_wgrad_combine");
        return 1;
    }
};

}   // namespace

void d2p_conv_frames_tune(int tpw) { g_frames_tpw = tpw > 0 ? tpw : 0; }
void d2p_conv_frames_wgrad_cap(int cap) { g_frames_wgrad_cap = cap; }

int d2p_conv_frames_fwd(const ConvGeom& g, const void* x, int x_is_u8, const float* w, const float* bias,
                        int act, float* y, hipStream_t st) {
    if (g.N < 1) return 0;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || (bias && ((uintptr_t)bias & 15))) return 0;
    if (x_is_u8) {      // frames as stored (booleans / bytes): first layer only
        if (g.Cin == 16 && g.Cout == 16 && g.H == 8 && g.W == 8)
            return launch_frames_fwd<16, 16, 8, 8, uint8_t>(g, (const uint8_t*)x, w, bias, act, y, st);
        return 0;
    }
    const float* xf = (const float*)x;
    if (g.Cin == 16 && g.Cout == 16 && g.H == 8 && g.W == 8) return launch_frames_fwd<16, 16, 8, 8, float>(g, xf, w, bias, act, y, st);
    if (g.Cin == 16 && g.Cout == 32 && g.H == 4 && g.W == 4) return launch_frames_fwd<16, 32, 4, 4, float>(g, xf, w, bias, act, y, st);
    if (g.Cin == 32 && g.Cout == 48 && g.H == 2 && g.W == 2) return launch_frames_fwd<32, 48, 2, 2, float>(g, xf, w, bias, act, y, st);
    return 0;
}

size_t d2p_conv_frames_wgrad_ws(const ConvGeom& g) {
    if (g.Cin == 16 && g.Cout == 16 && g.H == 8 && g.W == 8) return (size_t)WgradLaunch<16, 16, 8, 8, 4, 256, float>::blocks(g.N) * 9 * 16 * 16 * 4;
    if (g.Cin == 16 && g.Cout == 32 && g.H == 4 && g.W == 4) return (size_t)WgradLaunch<16, 32, 4, 4, 4, 128, float>::blocks(g.N) * 9 * 16 * 32 * 4;
    if (g.Cin == 32 && g.Cout == 48 && g.H == 2 && g.W == 2) return (size_t)WgradLaunch<32, 48, 2, 2, 4, 64, float>::blocks(g.N) * 9 * 32 * 48 * 4;
    return 0;
}

int d2p_conv_frames_wgrad(const ConvGeom& g, const void* x, int x_is_u8, const float* dy, float* dw, void* ws,
                          size_t ws_bytes, hipStream_t st) {
    if (g.N < 1) return 0;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 3)) return 0;
    if (x_is_u8) {
        if (g.Cin == 16 && g.Cout == 16 && g.H == 8 && g.W == 8)
            return WgradLaunch<16, 16, 8, 8, 4, 256, uint8_t>::run(g, (const uint8_t*)x, dy, dw, ws, ws_bytes, st);
        return 0;
    }
    const float* xf = (const float*)x;
    if (g.Cin == 16 && g.Cout == 16 && g.H == 8 && g.W == 8) return WgradLaunch<16, 16, 8, 8, 4, 256, float>::run(g, xf, dy, dw, ws, ws_bytes, st);
    if (g.Cin == 16 && g.Cout == 32 && g.H == 4 && g.W == 4) return WgradLaunch<16, 32, 4, 4, 4, 128, float>::run(g, xf, dy, dw, ws, ws_bytes, st);
    if (g.Cin == 32 && g.Cout == 48 && g.H == 2 && g.W == 2) return WgradLaunch<32, 48, 2, 2, 4, 64, float>::run(g, xf, dy, dw, ws, ws_bytes, st);
    return 0;
}
