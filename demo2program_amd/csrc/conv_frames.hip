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

unsigned* d2p_persist_err_ptr();       // lstm_persist.hip: the status word of the guarded optimizer step

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
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


// ------------------------------------------------------------------------------------------
// The whole Karel State_Encoder forward in ONE launch (round 4): three layers of
// conv3x3 s2 SAME + bias + lrelu + batch norm (training mode, statistics per demonstration index: models/ops.py:14-33,
// models/model_full.py:216-231,373-379), 8x8x16 -> 4x4x16 -> 2x2x32 -> 1x1x48, and the transpose to the time-major
// feature rows the first LSTM reads.  The chain it replaces is 13 launches (3 conv + 3 x (partial sums, finalize,
// apply) + transpose) of 5-10 us each: launch-bound (0.12 of the fp32 MFMA peak over the conv launches alone).
//
// A workgroup owns the frames of ONE demonstration index g for a few programs b (frames (b*G + g)*T .. +T-1): the
// batch-norm statistics of a layer mix all frames of an index, so the S workgroups of an index meet at a barrier per
// layer -- partial sums as write-through stores, an arrival counter per (slice count S, layer, index), every workgroup
// adds the S partial sums in slice order (fp64: the same statistics as the bn_partial / finalize launches up to the
// order of an fp64 sum).  A layer's activations never leave the CU on the way to the next layer: the epilogue writes
// them into the next layer's LDS image (the padded pixel-major form conv_frames_fwd_kernel stages from memory), the
// barrier's tail normalises that image in place.  What backward needs goes to memory on the way: a_l (pre-norm), y_l
// (post-norm, the next layer's input), mean / rstd / var per (index, channel).
// Needs the G*S workgroups co-resident (grid <= CUs, checked on the host); every spin is bounded and reports through the
// persistent kernels' status word (the guarded optimizer step then skips the step and the trainer re-runs it on the
// separate launches).
// The arrival counters are MONOTONIC 64-bit tickets, never reset and not chosen by the host: a workgroup's ticket t =
// fetch_add(counter, 1) belongs to generation t / S and waits for the counter to reach (t / S + 1) * S.  Launches that
// share a counter are ordered (one stream, or graph replays of one stream's capture) and every workgroup of a launch
// counts itself in exactly once per layer even when it gives up, so a launch always leaves a multiple of S behind --
// which makes the barrier safe under hipGraph replay (round 4 picked a counter slot on the HOST per launch: baked into
// a captured node, every replay after the first fell through).  One counter set per slice count S,
// so that launches of different geometry never share one.
#define ENC_MAXG 32
#define ENC_MAXS 256
__device__ unsigned long long g_enc_counters[ENC_MAXS + 1][3][ENC_MAXG];

struct EncArgs {
    const void* x;
    const float *w[3], *bias[3], *gamma[3], *beta[3];
    float *a[3], *y[2];              // a_l [NF, Ho, Wo, C_l]; y_1, y_2 (y_3 only as feats_tm)
    float* feats_tm;                 // [T, B*G, 48]
    float *mean[3], *rstd[3], *var[3];   // [G, C_l]
    double* part;                    // [3][G][S][48][2]
    unsigned long long* counters;    // g_enc_counters[S]
    unsigned* err;
    int B, G, T, S, nb;
    unsigned long long* trace;       // diagnostic (d2p_karel_encoder_set_trace): [workgroup][10] wall-clock stamps, or null
};

template <class SH>
__device__ __forceinline__ void enc_filter(const float* __restrict__ w, int p, int q, float (&wr)[SH::NCH][4][SH::NB]) {
#pragma unroll
    for (int ch = 0; ch < SH::NCH; ++ch)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < SH::NB; ++b)
                wr[ch][j][b] = w[(nth_tap(SH::MASK, ch / SH::CB) * SH::CIN_ + (ch % SH::CB) * 16 + 4 * q + j) * (SH::NB * 16) + b * 16 + p];
}
template <class SH>
__device__ __forceinline__ void enc_toff(int p, int q, int H, int W, int (&toff)[SH::NCH]) {
    const int f = p / SH::HW, r = p % SH::HW, oy = r / SH::Wo, ox = r % SH::Wo;
#pragma unroll
    for (int ch = 0; ch < SH::NCH; ++ch) {
        const int tap = nth_tap(SH::MASK, ch / SH::CB), ky = tap / 3, kx = tap % 3;
        const int iy = 2 * oy - SH::PT + ky, ix = 2 * ox - SH::PL + kx;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        toff[ch] = ok ? ((f * H + iy) * W + ix) * SH::PSF + (ch % SH::CB) * 16 + 4 * q : -1;
    }
}

// the barrier of layer `layer` among the S workgroups of index g: publish this workgroup's channel sums, wait for all
// S, add them in slice order -> mean / rstd of (g, c) in LDS (and, from slice 0, in memory for backward)
#define ENC_AUX_SC1 16          // buffer-instruction cache policy: sc1 (agent scope: the partial sums cross XCDs)
#define ENC_OOB 0x7fffff00      // an offset past every buffer: the load returns zeros

// Statistics of layer `layer`, index g, from the S workgroups of the index: each writes its fp64 partial sums (write-through),
// counts itself in and waits for the others; then the S x C pairs are read by the whole workgroup at once (R = 256 / C
// threads per channel, each summing the slices r, r + R, ... -- independent loads in flight -- then the R sums in order:
// the same order in every workgroup, so all of them normalise with the same bits).
template <int C>
__device__ __forceinline__ void enc_stats(const EncArgs& a, int layer, int g, int s, int n_per_group, double* wsum,
                                          float* mean_l, float* rstd_l, int* flag) {
    constexpr int R = 256 / C;
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(
        a.part + ((long)layer * a.G + g) * a.S * 96, 0, a.S * 96 * (int)sizeof(double), 0x00020000);
    if (tid < C) {
        // wsum: [4 waves][C][2] in LDS, added in wave order
        double s0 = 0.0, s1 = 0.0;
        for (int w = 0; w < 4; ++w) { s0 += wsum[(w * C + tid) * 2]; s1 += wsum[(w * C + tid) * 2 + 1]; }
        const double pr[2] = {s0, s1};
        i32x4 pv;
        __builtin_memcpy(&pv, pr, 16);
        __builtin_amdgcn_raw_buffer_store_b128(pv, res, (s * 48 + tid) * 16, 0, ENC_AUX_SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned long long* cnt = a.counters + (long)layer * ENC_MAXG + g;
        const unsigned long long ticket = __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = (ticket / (unsigned)a.S + 1ull) * (unsigned)a.S;
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 800000u || ((spins & 1023u) == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(a.err, (0x7eu << 24) | 0x800000u | (blockIdx.x & 0xffffu), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *flag = ok;
    }
    __syncthreads();
    {
        const int r = tid / C, c = tid - r * C;
        if (r < R) {
            double p0 = 0.0, p1 = 0.0;
            for (int s2 = r; s2 < a.S; s2 += 4 * R) {
                i32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int sx = s2 + u * R;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(res, sx < a.S ? (sx * 48 + c) * 16 : ENC_OOB, 0, ENC_AUX_SC1);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double pr[2];
                    __builtin_memcpy(pr, &v[u], 16);
                    p0 += pr[0];
                    p1 += pr[1];
                }
            }
            wsum[(r * C + c) * 2] = p0;
            wsum[(r * C + c) * 2 + 1] = p1;
        }
    }
    __syncthreads();
    if (tid < C) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) { s0 += wsum[(r * C + tid) * 2]; s1 += wsum[(r * C + tid) * 2 + 1]; }
        const double mu = s0 / n_per_group;
        double var = s1 / n_per_group - mu * mu;
        if (var < 0.0) var = 0.0;
        const float m = (float)mu, rs = (float)(1.0 / sqrt(var + 1e-3));
        mean_l[tid] = m;
        rstd_l[tid] = rs;
        if (s == 0) {
            a.mean[layer][g * C + tid] = m;
            a.rstd[layer][g * C + tid] = rs;
            a.var[layer][g * C + tid] = (float)var;
        }
    }
    __syncthreads();
}

// per-lane channel sums of a wave's epilogue values -> wsum[wave][C][2]: lanes (p, q) hold channels 16b + 4q + r of
// pixel p; the 16 pixel lanes of a quad are added by xor-shuffles (fixed order), lane p == 0 writes
template <int NB>
__device__ __forceinline__ void enc_fold_sums(const double (&sa)[NB][4], const double (&sb)[NB][4], int C, int wid, int p, int q,
                                              double* wsum) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double u = sa[b][r], v = sb[b][r];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                u += __shfl_xor(u, off, 64);
                v += __shfl_xor(v, off, 64);
            }
            if (p == 0) {
                wsum[(wid * C + b * 16 + 4 * q + r) * 2] = u;
                wsum[(wid * C + b * 16 + 4 * q + r) * 2 + 1] = v;
            }
        }
}

template <typename XT>
__global__ void __launch_bounds__(256)
karel_encoder_fwd_kernel(EncArgs a) {
    using S1 = FrameShape<16, 16, 8, 8>;
    using S2 = FrameShape<16, 32, 4, 4>;
    using S3 = FrameShape<32, 48, 2, 2>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, q = lane >> 4, wid = tid >> 6;
    const int g = blockIdx.x / a.S, s = blockIdx.x % a.S;
    const int b0 = s * a.nb, nbl = min(a.nb, a.B - b0);
    const int T = a.T, nfr = nbl > 0 ? nbl * T : 0;              // frames of this workgroup (local frame lf = bl*T + t)
    const int nfr16 = (a.nb * T + 15) / 16 * 16;
    // LDS: [conv1 staging: 4 waves x 2 images][IMG2: y1 of nfr frames][IMG3: y2][A3: a3][wsum][mean, rstd][flag]
    float* stage1 = lds;
    float* img2 = stage1 + 4 * 2 * S1::BUF;                       // nfr16 x 16 px x PSF2 (+ zero slot)
    float* img3 = img2 + (size_t)nfr16 * 16 * S2::PSF + 4;        // nfr16 x 4 px x PSF3 (+ zero slot)
    float* a3s = img3 + (size_t)nfr16 * 4 * S3::PSF + 4;          // nfr16 x 48
    double* wsum = reinterpret_cast<double*>(a3s + (size_t)nfr16 * 48);      // [4][48][2], then [R][C][2] (<= 512)
    float* mean_l = reinterpret_cast<float*>(wsum + 512);         // [48]
    float* rstd_l = mean_l + 48;                                  // [48]
    int* flag = reinterpret_cast<int*>(rstd_l + 48);
    const int n1 = a.B * T * 16, n2 = a.B * T * 4, n3 = a.B * T;  // values per (index, channel) of each layer
    auto frame_of = [&](int lf) { return ((long)(b0 + lf / T) * a.G + g) * T + (lf % T); };

    if (a.trace && tid == 0) a.trace[blockIdx.x * 10 + 0] = wall_clock64();
    // ================= layer 1: 8x8x16 -> 4x4x16, one frame per tile =================
    {
        float wr[S1::NCH][4][S1::NB];
        enc_filter<S1>(a.w[0], p, q, wr);
        int toff[S1::NCH];
        enc_toff<S1>(p, q, 8, 8, toff);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias[0] + 4 * q);
        float* const im0 = stage1 + (size_t)wid * 2 * S1::BUF;
        float* const im1 = im0 + S1::BUF;
        if (lane == 0) {
            *reinterpret_cast<f32x4*>(im0 + S1::NPIX * S1::PSF) = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(im1 + S1::NPIX * S1::PSF) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ch = 0; ch < S1::NCH; ++ch) if (toff[ch] < 0) toff[ch] = S1::NPIX * S1::PSF;
        // frame i of this wave: lf = wid + 4 i.  D frames ahead in registers (the HBM latency of a 1-KB frame is several
        // frames' products), the next frame's image written to the other LDS buffer before this frame's products
        constexpr int D = sizeof(XT) == 1 ? 4 : 2;
        Stager<XT, S1> st[D];
#pragma unroll
        for (int j = 0; j < D; ++j) st[j].init(lane);
        const long total = (long)a.B * a.G * T * S1::CHUNK;
        double sa[1][4] = {{0.0, 0.0, 0.0, 0.0}}, sb[1][4] = {{0.0, 0.0, 0.0, 0.0}};
        const int nmine = nfr > wid ? (nfr - wid + 3) / 4 : 0;
        const XT* xin = reinterpret_cast<const XT*>(a.x);
        auto frame_i = [&](int i) { return (int)frame_of(wid + 4 * (i < nmine ? i : nmine - 1)); };
        if (nmine > 0) {
#pragma unroll
            for (int j = 0; j < D; ++j) st[j].load(xin, frame_i(j), total, lane);
            st[0].store(im0);
            st[0].load(xin, frame_i(D), total, lane);
        }
        for (int base = 0; base < nmine; base += D) {
#pragma unroll
            for (int jj = 0; jj < D; ++jj) {
                const int i = base + jj;                   // slot (i + 1) % D == (jj + 1) % D holds frame i + 1
                if (i < nmine) {
                    const int par = i & 1;
                    const float* rimg = par ? im1 : im0;
                    float* wimg = par ? im0 : im1;
                    st[(jj + 1) % D].store(wimg);
                    st[(jj + 1) % D].load(xin, frame_i(i + 1 + D), total, lane);
                    const int lf = wid + 4 * i;
                    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int ch = 0; ch < S1::NCH; ++ch) {
                        const f32x4 bb = *reinterpret_cast<const f32x4*>(rimg + toff[ch]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j & 1] = D2P_MFMA16(wr[ch][j][0], bb[j], acc[j & 1]);
                    }
                    f32x4 o = (acc[0] + acc[1]) + bv;
                    o.x = d2p_lrelu(o.x); o.y = d2p_lrelu(o.y); o.z = d2p_lrelu(o.z); o.w = d2p_lrelu(o.w);
                    *reinterpret_cast<f32x4*>(a.a[0] + (frame_of(lf) * 16 + p) * 16 + 4 * q) = o;
                    *reinterpret_cast<f32x4*>(img2 + ((size_t)lf * 16 + p) * S2::PSF + 4 * q) = o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sa[0][r] += (double)o[r]; sb[0][r] += (double)o[r] * (double)o[r]; }
                }
            }
        }
        enc_fold_sums<1>(sa, sb, 16, wid, p, q, wsum);
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[blockIdx.x * 10 + 1] = wall_clock64();
    float wr2[S2::NCH][4][S2::NB];                                // (the next layer's filter loads fly during the exchange)
    enc_filter<S2>(a.w[1], p, q, wr2);
    enc_stats<16>(a, 0, g, s, n1, wsum, mean_l, rstd_l, flag);
    if (a.trace && tid == 0) a.trace[blockIdx.x * 10 + 2] = wall_clock64();
    // normalise IMG2 in place (-> y1, also to memory)
    for (int i = tid; i < nfr * 16 * 4; i += 256) {
        const int px = i >> 2, c4 = (i & 3) * 4, lf = px >> 4;
        float* v = img2 + (size_t)px * S2::PSF + c4;
        f32x4 x4 = *reinterpret_cast<f32x4*>(v), o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = a.gamma[0][c4 + r] * (x4[r] - mean_l[c4 + r]) * rstd_l[c4 + r] + a.beta[0][c4 + r];
        *reinterpret_cast<f32x4*>(v) = o;
        *reinterpret_cast<f32x4*>(a.y[0] + (frame_of(lf) * 16 + (px & 15)) * 16 + c4) = o;
    }
    if (tid == 0) *reinterpret_cast<f32x4*>(img2 + (size_t)nfr16 * 16 * S2::PSF) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    if (a.trace && tid == 0) a.trace[blockIdx.x * 10 + 3] = wall_clock64();
    // ================= layer 2: 4x4x16 -> 2x2x32, four frames per tile =================
    {
        auto& wr = wr2;
        int toff[S2::NCH];
        enc_toff<S2>(p, q, 4, 4, toff);
        f32x4 bv[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) bv[b] = *reinterpret_cast<const f32x4*>(a.bias[1] + b * 16 + 4 * q);
        const int zero2 = nfr16 * 16 * S2::PSF;                      // the zero slot, relative to img2
        double sa[2][4], sb[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) sa[b][r] = sb[b][r] = 0.0;
        const int ntile = (nfr + 3) / 4;
        for (int tile = wid; tile < ntile; tile += 4) {
            f32x4 acc[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[0][b] = acc[1][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < S2::NCH; ++ch) {
                const int o_ = toff[ch] >= 0 ? tile * 4 * 16 * S2::PSF + toff[ch] : zero2;      // (a select of the offset, not of the load)
                const f32x4 bb = *reinterpret_cast<const f32x4*>(img2 + o_);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[j & 1][b] = D2P_MFMA16(wr[ch][j][b], bb[j], acc[j & 1][b]);
            }
            const int lf = tile * 4 + p / 4, opx = p % 4;                 // this lane's frame and output pixel
            if (lf < nfr) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 o = (acc[0][b] + acc[1][b]) + bv[b];
                    o.x = d2p_lrelu(o.x); o.y = d2p_lrelu(o.y); o.z = d2p_lrelu(o.z); o.w = d2p_lrelu(o.w);
                    *reinterpret_cast<f32x4*>(a.a[1] + (frame_of(lf) * 4 + opx) * 32 + b * 16 + 4 * q) = o;
                    *reinterpret_cast<f32x4*>(img3 + ((size_t)lf * 4 + opx) * S3::PSF + b * 16 + 4 * q) = o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sa[b][r] += (double)o[r]; sb[b][r] += (double)o[r] * (double)o[r]; }
                }
            }
        }
        enc_fold_sums<2>(sa, sb, 32, wid, p, q, wsum);
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[blockIdx.x * 10 + 4] = wall_clock64();
    float wr3[S3::NCH][4][S3::NB];
    enc_filter<S3>(a.w[2], p, q, wr3);
    enc_stats<32>(a, 1, g, s, n2, wsum, mean_l, rstd_l, flag);
    if (a.trace && tid == 0) a.trace[blockIdx.x * 10 + 5] = wall_clock64();
    for (int i = tid; i < nfr * 4 * 8; i += 256) {
        const int px = i >> 3, c4 = (i & 7) * 4, lf = px >> 2;
        float* v = img3 + (size_t)px * S3::PSF + c4;
        f32x4 x4 = *reinterpret_cast<f32x4*>(v), o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = a.gamma[1][c4 + r] * (x4[r] - mean_l[c4 + r]) * rstd_l[c4 + r] + a.beta[1][c4 + r];
        *reinterpret_cast<f32x4*>(v) = o;
        *reinterpret_cast<f32x4*>(a.y[1] + (frame_of(lf) * 4 + (px & 3)) * 32 + c4) = o;
    }
    if (tid == 0) *reinterpret_cast<f32x4*>(img3 + (size_t)nfr16 * 4 * S3::PSF) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    if (a.trace && tid == 0) a.trace[blockIdx.x * 10 + 6] = wall_clock64();
    // ================= layer 3: 2x2x32 -> 1x1x48, sixteen frames per tile =================
    {
        auto& wr = wr3;
        int toff[S3::NCH];
        enc_toff<S3>(p, q, 2, 2, toff);
        f32x4 bv[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) bv[b] = *reinterpret_cast<const f32x4*>(a.bias[2] + b * 16 + 4 * q);
        const int zero3 = nfr16 * 4 * S3::PSF;
        double sa[3][4], sb[3][4];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) sa[b][r] = sb[b][r] = 0.0;
        const int ntile = (nfr + 15) / 16;
        for (int tile = wid; tile < ntile; tile += 4) {
            f32x4 acc[2][3];
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[0][b] = acc[1][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < S3::NCH; ++ch) {
                const int o_ = toff[ch] >= 0 ? tile * 16 * 4 * S3::PSF + toff[ch] : zero3;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(img3 + o_);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int b = 0; b < 3; ++b) acc[j & 1][b] = D2P_MFMA16(wr[ch][j][b], bb[j], acc[j & 1][b]);
            }
            const int lf = tile * 16 + p;
            if (lf < nfr) {
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    f32x4 o = (acc[0][b] + acc[1][b]) + bv[b];
                    o.x = d2p_lrelu(o.x); o.y = d2p_lrelu(o.y); o.z = d2p_lrelu(o.z); o.w = d2p_lrelu(o.w);
                    *reinterpret_cast<f32x4*>(a.a[2] + frame_of(lf) * 48 + b * 16 + 4 * q) = o;
                    *reinterpret_cast<f32x4*>(a3s + (size_t)lf * 48 + b * 16 + 4 * q) = o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sa[b][r] += (double)o[r]; sb[b][r] += (double)o[r] * (double)o[r]; }
                }
            }
        }
        enc_fold_sums<3>(sa, sb, 48, wid, p, q, wsum);
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[blockIdx.x * 10 + 7] = wall_clock64();
    enc_stats<48>(a, 2, g, s, n3, wsum, mean_l, rstd_l, flag);
    if (a.trace && tid == 0) a.trace[blockIdx.x * 10 + 8] = wall_clock64();
    // y3, time-major: feats_tm[t][m = b*G + g][48]
    const long M = (long)a.B * a.G;
    for (int i = tid; i < nfr * 12; i += 256) {
        const int lf = i / 12, c4 = (i - lf * 12) * 4;
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(a3s + (size_t)lf * 48 + c4);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = a.gamma[2][c4 + r] * (x4[r] - mean_l[c4 + r]) * rstd_l[c4 + r] + a.beta[2][c4 + r];
        const long m = (long)(b0 + lf / T) * a.G + g;
        *reinterpret_cast<f32x4*>(a.feats_tm + ((long)(lf % T) * M + m) * 48 + c4) = o;
    }
    if (a.trace && tid == 0) a.trace[blockIdx.x * 10 + 9] = wall_clock64();
}

static size_t enc_lds_bytes(int nb, int T) {
    using S1 = FrameShape<16, 16, 8, 8>;
    using S2 = FrameShape<16, 32, 4, 4>;
    using S3 = FrameShape<32, 48, 2, 2>;
    const size_t nfr16 = (size_t)(nb * T + 15) / 16 * 16;
    return (4 * 2 * S1::BUF + nfr16 * 16 * S2::PSF + 4 + nfr16 * 4 * S3::PSF + 4 + nfr16 * 48 + 512 * 2 + 96 + 4) * sizeof(float);
}
static bool enc_plan(int B, int G, int T, int& S, int& nb) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    if (B < 1 || G < 1 || G > ENC_MAXG || T < 4 || T % 4 != 0 || G > cus || cus / G > ENC_MAXS) return false;
    S = cus / G < B ? cus / G : B;
    nb = (B + S - 1) / S;
    S = (B + nb - 1) / nb;                       // no empty slices
    while (enc_lds_bytes(nb, T) > 150 * 1024) return false;
    return true;
}

unsigned long long* g_enc_trace = nullptr;

}   // namespace

// diagnostic (tools/karel_encoder_time.py): 10 wall-clock stamps (100 MHz) per workgroup of the following launches
extern "C" int d2p_karel_encoder_set_trace(void* buf) {
    g_enc_trace = (unsigned long long*)buf;
    return D2P_OK;
}

extern "C" size_t d2p_karel_encoder_ws_bytes(int B, int G, int T) {
    int S = 0, nb = 0;
    if (!enc_plan(B, G, T, S, nb)) return 0;
    return (size_t)3 * G * S * 48 * 2 * sizeof(double);
}

// (declared in include/d2p.h)
extern "C" int d2p_karel_encoder_fwd(int B, int G, int T, const void* x, int x_is_u8, const float* const* w,
                                     const float* const* bias, const float* const* gamma, const float* const* beta,
                                     float* const* a, float* const* y, float* feats_tm, float* const* mean,
                                     float* const* rstd, float* const* var, void* ws, size_t ws_bytes,
                                     d2p_stream_t stream) {
    int S = 0, nb = 0;
    D2P_REQUIRE(enc_plan(B, G, T, S, nb), D2P_EINVAL,
                "karel encoder: B=%d G=%d T=%d does not fit one launch (T %% 4 == 0, G <= 32, G*S workgroups co-resident)", B, G, T);
    D2P_REQUIRE(x && w && bias && gamma && beta && a && y && feats_tm && mean && rstd && var, D2P_EINVAL,
                "karel encoder: null pointer");
    D2P_REQUIRE(ws && ws_bytes >= d2p_karel_encoder_ws_bytes(B, G, T), D2P_EWS, "karel encoder: workspace too small");
    EncArgs e;
    e.x = x;
    for (int l = 0; l < 3; ++l) {
        D2P_REQUIRE(w[l] && bias[l] && gamma[l] && beta[l] && a[l] && mean[l] && rstd[l] && var[l] && (l == 2 || y[l]),
                    D2P_EINVAL, "karel encoder: null pointer (layer %d)", l + 1);
        e.w[l] = w[l]; e.bias[l] = bias[l]; e.gamma[l] = gamma[l]; e.beta[l] = beta[l];
        e.a[l] = a[l]; e.mean[l] = mean[l]; e.rstd[l] = rstd[l]; e.var[l] = var[l];
        if (l < 2) e.y[l] = y[l];
        D2P_REQUIRE((((uintptr_t)a[l] | (uintptr_t)bias[l]) & 15) == 0 && (l == 2 || ((uintptr_t)y[l] & 15) == 0), D2P_EALIGN,
                    "karel encoder: 16-byte alignment (layer %d)", l + 1);
    }
    D2P_REQUIRE((((uintptr_t)x | (uintptr_t)feats_tm) & 15) == 0, D2P_EALIGN, "karel encoder: 16-byte alignment");
    e.feats_tm = feats_tm;
    e.part = (double*)ws;
    static unsigned long long* counters = nullptr;
    if (!counters) D2P_HIP(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_enc_counters)));
    e.counters = counters + (size_t)S * 3 * ENC_MAXG;
    e.err = d2p_persist_err_ptr();
    e.B = B; e.G = G; e.T = T; e.S = S; e.nb = nb;
    e.trace = g_enc_trace;
    const size_t lds = enc_lds_bytes(nb, T);
    hipStream_t st = as_stream(stream);
    // conv flops of the three layers (as the separate launches count them)
    const double nf = (double)B * G * T;
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * nf * (16 * 144 * 16 + 4 * 144 * 32 + 1 * 288 * 48));
    static bool attr = false;
    if (!attr) {
        D2P_HIP(hipFuncSetAttribute((const void*)karel_encoder_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        D2P_HIP(hipFuncSetAttribute((const void*)karel_encoder_fwd_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    if (x_is_u8) hipLaunchKernelGGL(karel_encoder_fwd_kernel<uint8_t>, dim3(G * S), dim3(256), lds, st, e);
    else hipLaunchKernelGGL(karel_encoder_fwd_kernel<float>, dim3(G * S), dim3(256), lds, st, e);
    D2P_LAUNCH_CHECK("karel_encoder_fwd");
    return D2P_OK;
}

namespace {

// ------------------------------------------------------------------------------------------
// The whole Karel State_Encoder BACKWARD in one launch + one combine launch (round 5).
// The chain it replaces is 21 launches of 5-15 us each at the END of a step, where nothing runs beside them: the
// transpose of the feature gradient, and per layer batch-norm backward (partial sums, finalize, apply, column sums), the
// weight gradient + its combine, the input gradient (models/model_full.py:216-231, models/ops.py:14-33 in reverse).
//
// Same decomposition as the forward launch: workgroup (index g, slice s) owns the frames of a few programs of ONE
// demonstration index through all three layers; what the layers hand to each other stays in LDS:
//   D3 [frame][48]   gradient of layer 3's batch-norm output (read time-major: no transpose pass), turned in place into
//                    the gradient of its conv output (batch-norm backward + lrelu'),
//   D2 [frame][2x2][32], D1 [frame][4x4][16]   the same for layers 2 and 1; the input gradients of layers 3 and 2 are
//                    written straight into them (layer 3's taps map 1:1 onto layer 2's four pixels; layer 2's nine taps
//                    are scattered from 16-row output tiles, a frame's tile belonging to one wave, in a fixed order),
//   A3, A2, A1       the saved pre-norm activations a_l (batch-norm backward needs x-hat and the sign; the layer
//                    above needs y_l = batch norm(a_l) as its conv input: formed on load, one multiply-add).
// The batch-norm backward of a layer needs sum(dy) and sum(dy * x-hat) over ALL frames of the index: the S workgroups
// of an index meet once per layer exactly as in the forward launch (fp64 partial sums, monotonic arrival tickets).
// Products on the matrix pipe (16x16x4 fp32), operands from LDS:
//   input gradient   C[ci][pixel] = sum_co W[tap][ci][co] dpre[pixel][co]   (A = filter in registers, B = 16-byte LDS reads),
//   weight gradient  C[ci][co] = sum_pixel y[pixel@tap][ci] dpre[pixel][co]  (A, B = 4-byte LDS reads; layers 3 / 2: the
//                    taps dealt out to the waves, no cross-wave sum; layer 1: frames dealt out, x staged per wave as in
//                    the forward launch, the four waves' sums added in a fixed tree).
// Every parameter gradient (filters, biases, gamma, beta) is a sum over the workgroups: each writes ONE slab of
// ENC_SLAB floats, karel_encoder_bwd_fold_kernel adds the slabs in workgroup order (deterministic) into the gradient
// tensors -- the second launch.
#define ENC_SLAB_W1 0
#define ENC_SLAB_W2 2304                     // 9 * 16 * 16
#define ENC_SLAB_W3 6912                     // + 9 * 16 * 32; layer 3: the 4 taps that touch a 2x2 frame, [4][32][48]
#define ENC_SLAB_DB 13056                    // + 4 * 32 * 48; bias gradients 16 | 32 | 48
#define ENC_SLAB_GB 13152                    // (dgamma, dbeta) of layer 1 (16 | 16), layer 2 (32 | 32), layer 3 (48 | 48)
#define ENC_SLAB 13376                      // 13344 used, padded to whole 64-element blocks
__device__ unsigned long long g_enc_bwd_counters[ENC_MAXS + 1][3][ENC_MAXG];

struct EncBwdArgs {
    const void* x;                   // frames [B, G, T, 8, 8, 16]
    const float* dfeat_tm;           // [T, B*G, 48]: gradient of the time-major features
    const float *w[3], *gamma[3], *beta[3];
    const float* a[3];               // saved pre-norm activations
    const float *mean[3], *rstd[3];  // [G, C_l]
    double* part;                    // [3][G][S][48][2]
    float* slabs;                    // [G*S][ENC_SLAB]
    unsigned long long* counters;    // g_enc_bwd_counters[S]
    unsigned* err;
    int B, G, T, S, nb;
    unsigned long long* trace;       // diagnostic (d2p_karel_encoder_set_trace): [workgroup][16] wall-clock stamps, or null
};

// this workgroup's sums over its npx pixels: wsum[r][c] = (sum dy, sum dy * x-hat) of thread row r (R = 256 / C rows)
template <int C>
__device__ __forceinline__ void enc_bwd_sums(const float* A, int psa, const float* D, int psd, int npx, const float* mean,
                                             const float* rstd, double* wsum) {
    constexpr int R = 256 / C;
    const int r = threadIdx.x / C, c = threadIdx.x - r * C;
    if (r < R) {
        const float mu = mean[c], rs = rstd[c];
        double s1 = 0.0, s2 = 0.0;
        for (int px = r; px < npx; px += R) {
            const float dy = D[px * psd + c], xh = (A[px * psa + c] - mu) * rs;
            s1 += (double)dy;
            s2 += (double)dy * (double)xh;
        }
        wsum[(r * C + c) * 2] = s1;
        wsum[(r * C + c) * 2 + 1] = s2;
    }
    __syncthreads();
}

// the exchange of layer `layer` among the S workgroups of index g (as enc_stats), in two halves so that work that does
// not depend on it runs while the other workgroups arrive.  publish: own sums -> slab (dbeta, dgamma are sums over every
// workgroup) and -> part, then this workgroup's arrival ticket (its target count stays in LDS).
template <int C>
__device__ __forceinline__ void enc_bwd_publish(const EncBwdArgs& a, int layer, int g, int s, double* wsum, int* flag,
                                                float* slab_gb) {
    constexpr int R = 256 / C;
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(
        a.part + ((long)layer * a.G + g) * a.S * 96, 0, a.S * 96 * (int)sizeof(double), 0x00020000);
    if (tid < C) {
        double s0 = 0.0, s1 = 0.0;
        for (int w = 0; w < R; ++w) { s0 += wsum[(w * C + tid) * 2]; s1 += wsum[(w * C + tid) * 2 + 1]; }
        const double pr[2] = {s0, s1};
        i32x4 pv;
        __builtin_memcpy(&pv, pr, 16);
        __builtin_amdgcn_raw_buffer_store_b128(pv, res, (s * 48 + tid) * 16, 0, ENC_AUX_SC1);
        slab_gb[tid] = (float)s1;            // dgamma
        slab_gb[C + tid] = (float)s0;        // dbeta
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned long long* cnt = a.counters + (long)layer * ENC_MAXG + g;
        const unsigned long long ticket = __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = (ticket / (unsigned)a.S + 1ull) * (unsigned)a.S;
        __builtin_memcpy(flag + 2, &target, 8);
    }
}
// collect: wait for the S arrivals, add the S partial sums in slice order -> cst[5][48] = gamma*rstd, m1, m2, mean, rstd of (g, c)
template <int C>
__device__ __forceinline__ void enc_bwd_collect(const EncBwdArgs& a, int layer, int g, int n_per_group, double* wsum,
                                                float* cst, int* flag) {
    constexpr int R = 256 / C;
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(
        a.part + ((long)layer * a.G + g) * a.S * 96, 0, a.S * 96 * (int)sizeof(double), 0x00020000);
    if (tid == 0) {
        unsigned long long* cnt = a.counters + (long)layer * ENC_MAXG + g;
        unsigned long long target;
        __builtin_memcpy(&target, flag + 2, 8);
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 800000u || ((spins & 1023u) == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(a.err, (0x7du << 24) | 0x800000u | (blockIdx.x & 0xffffu), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *flag = ok;
    }
    __syncthreads();
    {
        const int r = tid / C, c = tid - r * C;
        if (r < R) {
            double p0 = 0.0, p1 = 0.0;
            for (int s2 = r; s2 < a.S; s2 += 4 * R) {
                i32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int sx = s2 + u * R;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(res, sx < a.S ? (sx * 48 + c) * 16 : ENC_OOB, 0, ENC_AUX_SC1);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    double pr[2];
                    __builtin_memcpy(pr, &v[u], 16);
                    p0 += pr[0];
                    p1 += pr[1];
                }
            }
            wsum[(r * C + c) * 2] = p0;
            wsum[(r * C + c) * 2 + 1] = p1;
        }
    }
    __syncthreads();
    if (tid < C) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) { s0 += wsum[(r * C + tid) * 2]; s1 += wsum[(r * C + tid) * 2 + 1]; }
        const float rs = a.rstd[layer][g * C + tid];
        cst[tid] = a.gamma[layer][tid] * rs;
        cst[48 + tid] = (float)(s0 / n_per_group);
        cst[96 + tid] = (float)(s1 / n_per_group);
        cst[144 + tid] = a.mean[layer][g * C + tid];
        cst[192 + tid] = rs;
    }
    __syncthreads();
}

// D <- gamma rstd (dy - m1 - x-hat m2) lrelu'(a) in place (the formula of bn_apply_bwd_kernel); the column sums of the
// result (this workgroup's share of the bias gradient) -> slab_db
template <int C>
__device__ __forceinline__ void enc_bwd_apply(const float* A, int psa, float* D, int psd, int npx, const float* cst,
                                              double* wsum, float* slab_db) {
    constexpr int R = 256 / C;
    const int r = threadIdx.x / C, c = threadIdx.x - r * C;
    if (r < R) {
        const float k1 = cst[c], m1 = cst[48 + c], m2 = cst[96 + c], mu = cst[144 + c], rs = cst[192 + c];
        float sum = 0.f;
        for (int px = r; px < npx; px += R) {
            const float av = A[px * psa + c];
            const float xh = (av - mu) * rs;
            float d = k1 * (D[px * psd + c] - m1 - xh * m2);
            d *= d2p_lrelu_grad_from_out(av);
            D[px * psd + c] = d;
            sum += d;
        }
        wsum[r * C + c] = (double)sum;
    }
    __syncthreads();
    if (threadIdx.x < C) {
        double t = 0.0;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) t += wsum[rr * C + threadIdx.x];
        slab_db[threadIdx.x] = (float)t;
    }
    __syncthreads();
}

struct EncBwdLds {      // offsets in floats
    int r0, r1, r2, wsum, cst, flag, zero, total;
};
__host__ __device__ inline EncBwdLds enc_bwd_lds(int nfr) {
    const int nfr16 = (nfr + 15) / 16 * 16;
    EncBwdLds L;
    const int s0 = 288 * nfr16 > 10272 ? 288 * nfr16 : 10272;       // A2 | D2 (36 floats per pixel); later conv1's staging images
    const int s1 = 256 * nfr > 4608 ? 256 * nfr : 4608;              // A1 (16 per pixel); later the weight-gradient tree
    const int s2 = 320 * nfr > 104 * nfr16 ? 320 * nfr : 104 * nfr16;   // D1 (20 per pixel); before that A3 | D3 (52 per frame)
    L.r0 = 0; L.r1 = s0; L.r2 = s0 + s1;
    L.wsum = s0 + s1 + s2;                      // 512 doubles
    L.cst = L.wsum + 1024;
    L.flag = L.cst + 240;                      // ok flag, -, the arrival target (64 bits)
    L.zero = L.flag + 8;                        // 36 zeros: the pixel outside a 2x2 gradient frame
    L.total = L.zero + 36;
    return L;
}

template <typename XT>
__global__ void __launch_bounds__(256)
karel_encoder_bwd_kernel(EncBwdArgs a) {
    using S1 = FrameShape<16, 16, 8, 8>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, q = lane >> 4, wid = tid >> 6;
    const int g = blockIdx.x / a.S, s = blockIdx.x % a.S;
    const int b0 = s * a.nb, nbl = min(a.nb, a.B - b0);
    const int T = a.T, nfr = nbl > 0 ? nbl * T : 0;              // (no empty slices: enc_plan)
    const int nfr16 = (nfr + 15) / 16 * 16;
    const EncBwdLds L = enc_bwd_lds(a.nb * T);
    float* const A2 = lds + L.r0;                                 // [nfr16 * 4][36]
    float* const D2 = A2 + (size_t)((a.nb * T + 15) / 16 * 16) * 144;
    float* const A1 = lds + L.r1;                                 // [nfr * 16][16]
    float* const D1 = lds + L.r2;                                 // [nfr * 16][20]
    float* const A3 = D1;                                         // [nfr16][52]  (dead before layer 2's input gradient lands in D1)
    float* const D3 = A3 + (size_t)((a.nb * T + 15) / 16 * 16) * 52;
    double* const wsum = reinterpret_cast<double*>(lds + L.wsum);
    float* const cst = lds + L.cst;
    int* const flag = reinterpret_cast<int*>(lds + L.flag);
    float* const slab = a.slabs + (size_t)blockIdx.x * ENC_SLAB;
    const long M = (long)a.B * a.G;
    const int n1 = a.B * T * 16, n2 = a.B * T * 4, n3 = a.B * T;
    auto frame_of = [&](int lf) { return ((long)(b0 + lf / T) * a.G + g) * T + (lf % T); };
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 0] = wall_clock64();

    // ---------------- layer 3's activations and the incoming gradient -> LDS ----------------
    for (int i = tid; i < nfr16 * 12; i += 256) {
        const int lf = i / 12, c4 = (i - lf * 12) * 4;
        f32x4 av = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
        if (lf < nfr) {
            av = *reinterpret_cast<const f32x4*>(a.a[2] + frame_of(lf) * 48 + c4);
            const long m = (long)(b0 + lf / T) * a.G + g;
            dv = *reinterpret_cast<const f32x4*>(a.dfeat_tm + ((long)(lf % T) * M + m) * 48 + c4);
        }
        *reinterpret_cast<f32x4*>(A3 + lf * 52 + c4) = av;
        *reinterpret_cast<f32x4*>(D3 + lf * 52 + c4) = dv;
    }
    if (tid < 9) *reinterpret_cast<f32x4*>(lds + L.zero + tid * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 1] = wall_clock64();

    // ================= layer 3: 2x2x32 -> 1x1x48 =================
    enc_bwd_sums<48>(A3, 52, D3, 52, nfr, a.mean[2] + g * 48, a.rstd[2] + g * 48, wsum);
    enc_bwd_publish<48>(a, 2, g, s, wsum, flag, slab + ENC_SLAB_GB + 96);
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 2] = wall_clock64();
    // while the index's other workgroups arrive: the activations of layers 2 and 1 -> LDS
    for (int i = tid; i < nfr16 * 32; i += 256) {
        const int lf = i >> 5, rem = i & 31;
        f32x4 av = {0.f, 0.f, 0.f, 0.f};
        if (lf < nfr) av = *reinterpret_cast<const f32x4*>(a.a[1] + frame_of(lf) * 128 + rem * 4);
        *reinterpret_cast<f32x4*>(A2 + (lf * 4 + (rem >> 3)) * 36 + (rem & 7) * 4) = av;
    }
    for (int i = tid; i < nfr * 64; i += 256) {
        const int lf = i >> 6, rem = i & 63;
        *reinterpret_cast<f32x4*>(A1 + (size_t)i * 4) = *reinterpret_cast<const f32x4*>(a.a[0] + frame_of(lf) * 256 + rem * 4);
    }
    enc_bwd_collect<48>(a, 2, g, n3, wsum, cst, flag);
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 3] = wall_clock64();
    enc_bwd_apply<48>(A3, 52, D3, 52, nfr, cst, wsum, slab + ENC_SLAB_DB + 48);
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 4] = wall_clock64();
    {
        // input gradient: D2[frame][pixel w][ci] = sum_co W3[tap][ci][co] dpre3[frame][co]; wave w: tap (ky, kx) =
        // (w / 2, w % 2), the tap that reads pixel w of the 2x2 frame (pad_before(2) = 0)
        const int tapid = (wid >> 1) * 3 + (wid & 1);
        float wr[2][3][4];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int ob = 0; ob < 3; ++ob)
#pragma unroll
                for (int j = 0; j < 4; ++j) wr[cb][ob][j] = a.w[2][((tapid * 32 + cb * 16 + p) * 48) + ob * 16 + 4 * q + j];
        for (int ft = 0; ft < nfr16 / 16; ++ft) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ob = 0; ob < 3; ++ob) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(D3 + (ft * 16 + p) * 52 + ob * 16 + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) acc[cb] = D2P_MFMA16(wr[cb][ob][j], bb[j], acc[cb]);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                *reinterpret_cast<f32x4*>(D2 + ((ft * 16 + p) * 4 + wid) * 36 + cb * 16 + 4 * q) = acc[cb];
        }
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 5] = wall_clock64();

    // ================= layer 2: 4x4x16 -> 2x2x32 =================
    enc_bwd_sums<32>(A2, 36, D2, 36, nfr * 4, a.mean[1] + g * 32, a.rstd[1] + g * 32, wsum);
    enc_bwd_publish<32>(a, 1, g, s, wsum, flag, slab + ENC_SLAB_GB + 32);
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 6] = wall_clock64();
    {
        // while the others arrive -- layer 3's weight gradient (nothing downstream reads it):
        // dW3[tap][ci][co] = sum_frame y2[frame][pixel w][ci] dpre3[frame][co], wave w: tap w
        float sc[2], sh[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int ci = cb * 16 + p;
            sc[cb] = a.gamma[1][ci] * a.rstd[1][g * 32 + ci];
            sh[cb] = a.beta[1][ci] - a.mean[1][g * 32 + ci] * sc[cb];
        }
        f32x4 acc[2][3];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int ob = 0; ob < 3; ++ob) acc[cb][ob] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < nfr16 / 4; ++ks) {
            const int f = 4 * ks + q;
            float av[2], bv[3];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) av[cb] = A2[(f * 4 + wid) * 36 + cb * 16 + p] * sc[cb] + sh[cb];
#pragma unroll
            for (int ob = 0; ob < 3; ++ob) bv[ob] = D3[f * 52 + ob * 16 + p];      // (rows past nfr: zeros)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ob = 0; ob < 3; ++ob) acc[cb][ob] = D2P_MFMA16(av[cb], bv[ob], acc[cb][ob]);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int ob = 0; ob < 3; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    slab[ENC_SLAB_W3 + (wid * 32 + cb * 16 + 4 * q + r) * 48 + ob * 16 + p] = acc[cb][ob][r];
    }
    enc_bwd_collect<32>(a, 1, g, n2, wsum, cst, flag);           // (its first barrier: A3 / D3 are dead from here)
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 7] = wall_clock64();
    enc_bwd_apply<32>(A2, 36, D2, 36, nfr * 4, cst, wsum, slab + ENC_SLAB_DB + 16);
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 8] = wall_clock64();
    {
        // input gradient, gathered by the parity class (ey, ex) of the input pixel: pixel (2 cy + ey, 2 cx + ex) of a 4x4
        // frame receives the taps ky in {0, 2} (ey = 0) or {1} (ey = 1) from output row oy = cy - (ky == 2) (pad_before(4) =
        // 0), likewise in x -- 4 + 2 + 2 + 1 tap instances, each C[ci][pixel] (K = 32 co) on a tile of 4 frames x the
        // class's 2x2 pixels; a row outside the 2x2 gradient frame reads the zero pixel.  Every D1 element is written once.
        float wr[9][2][4];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int j = 0; j < 4; ++j) wr[tap][ob][j] = a.w[1][(tap * 16 + p) * 32 + ob * 16 + 4 * q + j];
        const int fl = p >> 2, cy = (p >> 1) & 1, cx = p & 1;
        const int zoff = (int)(lds + L.zero - D2);
        for (int tile = wid; tile < nfr / 4; tile += 4) {
            const int fbase = (tile * 4 + fl) * 4;
#pragma unroll
            for (int ey = 0; ey < 2; ++ey)
#pragma unroll
                for (int ex = 0; ex < 2; ++ex) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ky = ey; ky < 3; ky += 2)
#pragma unroll
                        for (int kx = ex; kx < 3; kx += 2) {
                            const int oy = cy - (ky == 2 ? 1 : 0), ox = cx - (kx == 2 ? 1 : 0);
                            const int off = (oy >= 0 && ox >= 0) ? (fbase + oy * 2 + ox) * 36 : zoff;
#pragma unroll
                            for (int ob = 0; ob < 2; ++ob) {
                                const f32x4 bb = *reinterpret_cast<const f32x4*>(D2 + off + ob * 16 + 4 * q);
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    if (j & 1) acc1 = D2P_MFMA16(wr[ky * 3 + kx][ob][j], bb[j], acc1);
                                    else acc0 = D2P_MFMA16(wr[ky * 3 + kx][ob][j], bb[j], acc0);
                                }
                            }
                        }
                    *reinterpret_cast<f32x4*>(D1 + ((tile * 4 + fl) * 16 + (2 * cy + ey) * 4 + 2 * cx + ex) * 20 + 4 * q) = acc0 + acc1;
                }
        }
    }
    __syncthreads();
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 9] = wall_clock64();

    // ================= layer 1: 8x8x16 -> 4x4x16 (no input gradient) =================
    enc_bwd_sums<16>(A1, 16, D1, 20, nfr * 16, a.mean[0] + g * 16, a.rstd[0] + g * 16, wsum);
    enc_bwd_publish<16>(a, 0, g, s, wsum, flag, slab + ENC_SLAB_GB);
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 10] = wall_clock64();
    {
        // while the others arrive -- layer 2's weight gradient: (tap, 16-column block) pairs dealt out to the waves; the
        // reduction index of a matrix instruction is (frame, output pixel q)
        const float sc = a.gamma[0][p] * a.rstd[0][g * 16 + p], sh = a.beta[0][p] - a.mean[0][g * 16 + p] * sc;
#pragma unroll 1
        for (int pr = wid; pr < 18; pr += 4) {
            const int tap = pr >> 1, ob = pr & 1, ky = tap / 3, kx = tap - 3 * ky;
            const int iy = 2 * (q >> 1) + ky, ix = 2 * (q & 1) + kx;
            const bool ok = iy < 4 && ix < 4;
            const int aoff = ok ? (iy * 4 + ix) * 16 + p : 0;
            const float scv = ok ? sc : 0.f, shv = ok ? sh : 0.f;      // (outside the frame: the zero padding, not beta)
            const int boff = q * 36 + ob * 16 + p;
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int f = 0; f < nfr; f += 4) {                        // (nfr % 4 == 0)
                float av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { av[u] = A1[(f + u) * 256 + aoff]; bv[u] = D2[(f + u) * 144 + boff]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = D2P_MFMA16(av[u] * scv + shv, bv[u], acc[u]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[ENC_SLAB_W2 + (tap * 16 + 4 * q + r) * 32 + ob * 16 + p] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
        }
    }
    enc_bwd_collect<16>(a, 0, g, n1, wsum, cst, flag);           // (its first barrier: A2 / D2 are dead from here)
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 11] = wall_clock64();
    enc_bwd_apply<16>(A1, 16, D1, 20, nfr * 16, cst, wsum, slab + ENC_SLAB_DB);
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 12] = wall_clock64();
    {
        // weight gradient: frames dealt out to the waves (lf = wid + 4 i), x staged through wave-private images in the
        // dead A2 | D2 region as in the forward launch
        float* const im0 = lds + L.r0 + (size_t)wid * 2 * S1::BUF;
        float* const im1 = im0 + S1::BUF;
        if (lane == 0) {
            *reinterpret_cast<f32x4*>(im0 + S1::NPIX * S1::PSF) = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(im1 + S1::NPIX * S1::PSF) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        int toff[9][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int opx = 4 * m + q, oy = opx >> 2, ox = opx & 3;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = 2 * oy + tap / 3, ix = 2 * ox + tap % 3;
                toff[tap][m] = (iy < 8 && ix < 8) ? (iy * 8 + ix) * S1::PSF + p : S1::NPIX * S1::PSF;
            }
        }
        f32x4 acc[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc[tap] = f32x4{0.f, 0.f, 0.f, 0.f};
        Stager<XT, S1> st;
        st.init(lane);
        const long total = (long)a.B * a.G * T * S1::CHUNK;
        const XT* xin = reinterpret_cast<const XT*>(a.x);
        const int nmine = nfr > wid ? (nfr - wid + 3) / 4 : 0;
        if (nmine > 0) st.load(xin, (int)frame_of(wid), total, lane);
        for (int i = 0; i < nmine; ++i) {
            float* const img = (i & 1) ? im1 : im0;
            st.store(img);
            const int nx = i + 1 < nmine ? i + 1 : i;
            st.load(xin, (int)frame_of(wid + 4 * nx), total, lane);
            const int lf = wid + 4 * i;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float bv = D1[(lf * 16 + 4 * m + q) * 20 + p];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) acc[tap] = D2P_MFMA16(img[toff[tap][m]], bv, acc[tap]);
            }
        }
        // the four waves' sums in a fixed tree through the dead A1 region
        __syncthreads();
        if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 13] = wall_clock64();
        float* const red = lds + L.r1;
#pragma unroll
        for (int step = 1; step < 4; step *= 2) {
            if ((wid & (2 * step - 1)) == step) {
                float* dst = red + (size_t)(wid / (2 * step)) * 36 * 64;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(tap * 4 + r) * 64 + lane] = acc[tap][r];
            }
            __syncthreads();
            if ((wid & (2 * step - 1)) == 0) {
                const float* src = red + (size_t)(wid / (2 * step)) * 36 * 64;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[tap][r] += src[(tap * 4 + r) * 64 + lane];
            }
            __syncthreads();
        }
        if (wid == 0) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[ENC_SLAB_W1 + (tap * 16 + 4 * q + r) * 16 + p] = acc[tap][r];
        }
    }
    if (a.trace && tid == 0) a.trace[blockIdx.x * 16 + 14] = wall_clock64();
}

// the slabs of all workgroups added in workgroup order -> the twelve gradient tensors (layer 3's five taps that never
// touch a 2x2 frame get their exact zeros)
struct EncFoldArgs {
    const float* slabs;
    int nslab;
    float *dw[3], *db[3], *dgamma[3], *dbeta[3];
};
// 256 threads = 4 waves x 64 consecutive elements (coalesced 256-byte rows of a slab); wave w adds the slabs
// [w * nslab / 4, (w + 1) * nslab / 4) in order, eight loads in flight, then the four partial sums meet in LDS in wave order
__global__ void __launch_bounds__(256) karel_encoder_bwd_fold_kernel(EncFoldArgs a) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= ENC_SLAB) {                           // (ENC_SLAB is a multiple of 64: whole blocks of zeros)
        const int i = e - ENC_SLAB;
        if (wv == 0 && i < 5 * 1536) {
            const int t5 = i / 1536;
            a.dw[2][(t5 == 0 ? 2 : t5 + 4) * 1536 + (i - t5 * 1536)] = 0.f;      // taps 2, 5, 6, 7, 8
        }
        return;
    }
    const int n0 = (int)((long)a.nslab * wv / 4), n1 = (int)((long)a.nslab * (wv + 1) / 4);
    const float* sp = a.slabs + e;
    float sum = 0.f;
    int n = n0;
    for (; n + 8 <= n1; n += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sp[(size_t)(n + u) * ENC_SLAB];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
    }
    for (; n < n1; ++n) sum += sp[(size_t)n * ENC_SLAB];
    part[wv][lane] = sum;
    __syncthreads();
    if (wv != 0) return;
    sum = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    if (e < ENC_SLAB_W2) a.dw[0][e] = sum;
    else if (e < ENC_SLAB_W3) a.dw[1][e - ENC_SLAB_W2] = sum;
    else if (e < ENC_SLAB_DB) {
        const int i = e - ENC_SLAB_W3, t4 = i / 1536;
        a.dw[2][((t4 >> 1) * 3 + (t4 & 1)) * 1536 + (i - t4 * 1536)] = sum;
    } else if (e < ENC_SLAB_GB) {
        const int i = e - ENC_SLAB_DB;
        if (i < 16) a.db[0][i] = sum;
        else if (i < 48) a.db[1][i - 16] = sum;
        else a.db[2][i - 48] = sum;
    } else {
        const int i = e - ENC_SLAB_GB;
        if (i < 16) a.dgamma[0][i] = sum;
        else if (i < 32) a.dbeta[0][i - 16] = sum;
        else if (i < 64) a.dgamma[1][i - 32] = sum;
        else if (i < 96) a.dbeta[1][i - 64] = sum;
        else if (i < 144) a.dgamma[2][i - 96] = sum;
        else if (i < 192) a.dbeta[2][i - 144] = sum;
    }
}

static bool enc_bwd_plan(int B, int G, int T, int& S, int& nb) {
    if (!enc_plan(B, G, T, S, nb)) return false;
    return (size_t)enc_bwd_lds(nb * T).total * sizeof(float) <= 160 * 1024;
}

}   // namespace

extern "C" size_t d2p_karel_encoder_bwd_ws_bytes(int B, int G, int T) {
    int S = 0, nb = 0;
    if (!enc_bwd_plan(B, G, T, S, nb)) return 0;
    return (size_t)3 * G * S * 48 * 2 * sizeof(double) + (size_t)G * S * ENC_SLAB * sizeof(float);
}

// (declared in include/d2p.h)
extern "C" int d2p_karel_encoder_bwd(int B, int G, int T, const void* x, int x_is_u8, const float* dfeat_tm,
                                     const float* const* w, const float* const* gamma, const float* const* beta,
                                     const float* const* a, const float* const* mean, const float* const* rstd,
                                     float* const* dw, float* const* db, float* const* dgamma, float* const* dbeta, void* ws,
                                     size_t ws_bytes, d2p_stream_t stream) {
    int S = 0, nb = 0;
    D2P_REQUIRE(enc_bwd_plan(B, G, T, S, nb), D2P_EINVAL,
                "karel encoder backward: B=%d G=%d T=%d does not fit one launch", B, G, T);
    D2P_REQUIRE(x && dfeat_tm && w && gamma && beta && a && mean && rstd && dw && db && dgamma && dbeta, D2P_EINVAL,
                "karel encoder backward: null pointer");
    D2P_REQUIRE(ws && ws_bytes >= d2p_karel_encoder_bwd_ws_bytes(B, G, T), D2P_EWS, "karel encoder backward: workspace too small");
    EncBwdArgs e;
    EncFoldArgs f;
    e.x = x; e.dfeat_tm = dfeat_tm;
    for (int l = 0; l < 3; ++l) {
        D2P_REQUIRE(w[l] && gamma[l] && beta[l] && a[l] && mean[l] && rstd[l] && dw[l] && db[l] && dgamma[l] && dbeta[l],
                    D2P_EINVAL, "karel encoder backward: null pointer (layer %d)", l + 1);
        D2P_REQUIRE(((uintptr_t)a[l] & 15) == 0, D2P_EALIGN, "karel encoder backward: 16-byte alignment (layer %d)", l + 1);
        e.w[l] = w[l]; e.gamma[l] = gamma[l]; e.beta[l] = beta[l]; e.a[l] = a[l]; e.mean[l] = mean[l]; e.rstd[l] = rstd[l];
        f.dw[l] = dw[l]; f.db[l] = db[l]; f.dgamma[l] = dgamma[l]; f.dbeta[l] = dbeta[l];
    }
    D2P_REQUIRE((((uintptr_t)x | (uintptr_t)dfeat_tm | (uintptr_t)ws) & 15) == 0, D2P_EALIGN,
                "karel encoder backward: 16-byte alignment");
    e.part = (double*)ws;
    e.slabs = (float*)((char*)ws + (size_t)3 * G * S * 48 * 2 * sizeof(double));
    static unsigned long long* counters = nullptr;
    if (!counters) D2P_HIP(hipGetSymbolAddress((void**)&counters, HIP_SYMBOL(g_enc_bwd_counters)));
    e.counters = counters + (size_t)S * 3 * ENC_MAXG;
    e.err = d2p_persist_err_ptr();
    e.B = B; e.G = G; e.T = T; e.S = S; e.nb = nb;
    e.trace = g_enc_trace;
    f.slabs = e.slabs; f.nslab = G * S;
    const size_t lds = (size_t)enc_bwd_lds(nb * T).total * sizeof(float);
    hipStream_t st = as_stream(stream);
    // conv flops as the separate launches count them: three weight gradients, two input gradients
    const double nf = (double)B * G * T;
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * nf * (16 * 144 * 16 + 2 * 4 * 144 * 32 + 2 * 1 * 288 * 48));
    static bool attr = false;
    if (!attr) {
        D2P_HIP(hipFuncSetAttribute((const void*)karel_encoder_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        D2P_HIP(hipFuncSetAttribute((const void*)karel_encoder_bwd_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    if (x_is_u8) hipLaunchKernelGGL(karel_encoder_bwd_kernel<uint8_t>, dim3(G * S), dim3(256), lds, st, e);
    else hipLaunchKernelGGL(karel_encoder_bwd_kernel<float>, dim3(G * S), dim3(256), lds, st, e);
    D2P_LAUNCH_CHECK("karel_encoder_bwd");
    static_assert(ENC_SLAB % 64 == 0, "the combine launch works on whole 64-element blocks");
    hipLaunchKernelGGL(karel_encoder_bwd_fold_kernel, dim3((ENC_SLAB + 5 * 1536) / 64), dim3(256), 0, st, f);
    D2P_LAUNCH_CHECK("karel_encoder_bwd_fold");
    return D2P_OK;
}

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
