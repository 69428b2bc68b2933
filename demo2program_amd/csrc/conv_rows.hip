// K1 (row-strip back end): weight gradient of the two narrow ViZDoom layers
// (80x80x{3->4} -> 40x40x16 and 40x40x16 -> 20x20x32; models/model_full.py:216-231), the largest
// single kernels of that configuration.  Same idea as conv_frames.hip at a size where a frame no
// longer fits a tile: a wave's unit of work is ONE OUTPUT ROW of one frame.  Its three input rows
// are contiguous in NHWC, so they are fetched with lane-linear 16-byte loads (uint8 frames: 16
// bytes = 4 pixels, widened on the way) into a wave-private LDS strip with a zero pixel of halo on
// either side; the A operand (x, reduction index = pixel, 4 pixels per MFMA) is ds_read_b32 at
// per-lane offsets that are fixed for the kernel plus a constant per k-step; dY rows (contiguous)
// go straight to registers in the MFMA B layout.  No bounds arithmetic in the loop at all: column
// overhang lands in the halo, out-of-image rows are staged as zeros.  The whole dW lives in
// accumulators; waves are summed in a fixed tree; one slab per workgroup -> deterministic combine.
#include "conv_geom.h"
#include "gemm_core.h"
#include "prof.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define D2P_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define D2P_OPAQUE_I(v) asm volatile("" : "+v"(v))

namespace {

constexpr int r_out(int n) { return (n + 1) / 2; }
constexpr int r_before(int n) {
    int total = (r_out(n) - 1) * 2 + 3 - n;
    return total < 0 ? 0 : total / 2;
}

template <int CIN, int COUT, int W>
struct RowShape {
    static constexpr int Wo = r_out(W), PL = r_before(W);
    static constexpr int KS = (Wo + 3) / 4;                  // MFMA k-steps (4 output pixels each) per row
    static constexpr int PSF = CIN == 4 ? 4 : CIN + 8;       // floats per staged pixel (bank spreading)
    static constexpr int ROWPIX = 8 * KS + 4;                // staged pixels per row incl. halo / overhang
    static constexpr int ROWF = ROWPIX * PSF;
    static constexpr int IMG = 3 * ROWF;                     // floats per wave
    static constexpr int AB = CIN == 4 ? 3 : 9 * (CIN / 16); // 16-row blocks of dW
    static constexpr int NBO = COUT / 16;
    static constexpr int RP4 = W * CIN / 4;                  // 16-byte float pieces per input row
    static constexpr int NLF = (3 * RP4 + 63) / 64;          // float pieces per lane per strip
    static constexpr int RP16 = W * CIN / 16;                // 16-byte uint8 pieces per input row
    static constexpr int NLU = (3 * RP16 + 63) / 64;
    static_assert(W * CIN % 16 == 0, "rows must be whole 16-byte pieces");
    static_assert(2 * (4 * KS - 1) + 2 + 1 < ROWPIX, "overhang must stay inside the staged row");
};

// -- staging of the strip's three input rows (float / uint8) ----------------------------------
template <typename T, class S, int CIN, int W>
struct RowStager;

template <class S, int CIN, int W>
struct RowStager<float, S, CIN, W> {
    static constexpr int N = S::NLF;
    f32x4 r[N];
    __device__ __forceinline__ void load(const float* __restrict__ x, long frame_off, int iy0, int H, int lane) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = i * 64 + lane;
            const int row = j / S::RP4, wi = j - row * S::RP4;
            const int iy = iy0 + row;
            const bool ok = (row < 3) & (iy >= 0) & (iy < H);
            const int iyc = min(max(iy, 0), H - 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + frame_off + ((long)iyc * W) * CIN + (row < 3 ? wi : 0) * 4);
            r[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // the previous layer's batch-norm apply on the way in (ConvBnFold::in_scale / in_shift): this lane's pieces all
    // hold channel quad lane % (CIN / 4) (64 and the row's piece count are multiples of CIN / 4); rows outside the
    // image stay zero -- the zero padding is of the NORMALISED tensor
    __device__ __forceinline__ void load_affine(const float* __restrict__ x, long frame_off, int iy0, int H, int lane,
                                                f32x4 sc, f32x4 sh) {
        static_assert(S::RP4 % (CIN / 4) == 0 && 64 % (CIN / 4) == 0, "a lane keeps one channel quad");
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = i * 64 + lane;
            const int row = j / S::RP4, wi = j - row * S::RP4;
            const int iy = iy0 + row;
            const bool ok = (row < 3) & (iy >= 0) & (iy < H);
            const int iyc = min(max(iy, 0), H - 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + frame_off + ((long)iyc * W) * CIN + (row < 3 ? wi : 0) * 4);
            // (a 0/1 multiply, not a select: hipcc turns `ok ? v * sc + sh : 0` into a branch around the load and waits
            //  vmcnt(0) at every join -- measured +100 us on the 16 -> 32 layer)
            const float okf = ok ? 1.f : 0.f;
            r[i] = (v * sc + sh) * okf;
        }
    }
    __device__ __forceinline__ void store(float* img, int lane) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = i * 64 + lane;
            const int row = j / S::RP4, wi = j - row * S::RP4;
            if (row < 3)
                *reinterpret_cast<f32x4*>(img + row * S::ROWF + (wi / (CIN / 4) + 1) * S::PSF + (wi % (CIN / 4)) * 4) = r[i];
        }
    }
};

template <class S, int CIN, int W>
struct RowStager<uint8_t, S, CIN, W> {
    static constexpr int N = S::NLU;
    uint4 r[N];
    __device__ __forceinline__ void load(const uint8_t* __restrict__ x, long frame_off, int iy0, int H, int lane) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = i * 64 + lane;
            const int row = j / S::RP16, wi = j - row * S::RP16;
            const int iy = iy0 + row;
            const bool ok = (row < 3) & (iy >= 0) & (iy < H);
            const int iyc = min(max(iy, 0), H - 1);
            const uint4 v = *reinterpret_cast<const uint4*>(x + frame_off + ((long)iyc * W) * CIN + (row < 3 ? wi : 0) * 16);
            r[i] = ok ? v : uint4{0u, 0u, 0u, 0u};
        }
    }
    __device__ __forceinline__ void store(float* img, int lane) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int j = i * 64 + lane;
            const int row = j / S::RP16, wi = j - row * S::RP16;
            if (row >= 3) continue;
            const uint32_t w4[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = wi * 16 + 4 * k;                    // first channel-element of this float4
                f32x4 v;
                v.x = (float)(w4[k] & 255u); v.y = (float)((w4[k] >> 8) & 255u);
                v.z = (float)((w4[k] >> 16) & 255u); v.w = (float)(w4[k] >> 24);
                *reinterpret_cast<f32x4*>(img + row * S::ROWF + (e / CIN + 1) * S::PSF + e % CIN) = v;
            }
        }
    }
};

// BNBWD (round 5): `dy` is the gradient w.r.t. the batch-norm OUTPUT of this layer and `act` the layer's pre-norm
// activation; the conv's own output gradient da = (k1 * dy + k2 * act + k3) * lrelu'(act) -- coef [G, COUT, 4] from
// d2p_bn_group_bwd_coef -- is formed as the dY fragments are loaded (the batch-norm apply pass that used to write it,
// a read of act and dy and a write of da, is gone), and the bias gradient, its column sums, comes out with the slab:
// bsum [workgroups][COUT].
struct RowsBnBwd {
    const float* act; const float* coef; float* bsum;
    int G; unsigned seq_m; int seq_s; unsigned g_m; int g_s;      // (n / seq) % G by multiply-high (d2p_make_div)
};
template <int CIN, int COUT, int W, typename T, bool AFFINE = false, bool BNBWD = false>
__global__ void __launch_bounds__(256)
conv_rows_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slabs, int nframes,
                       int H, int Ho, int pt, int bnG, int bnseq, const float* __restrict__ in_scale,
                       const float* __restrict__ in_shift, RowsBnBwd bw) {
    using S = RowShape<CIN, COUT, W>;
    constexpr int AB = S::AB, NBO = S::NBO, KS = S::KS, PSF = S::PSF, Wo = S::Wo;
    constexpr int ACC = AB * NBO * 4, KK = 9 * CIN;
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, c = lane & 15, kq = lane >> 4, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wid, NW = gridDim.x * 4;
    float* const img = lds + (size_t)wid * S::IMG;
    for (int i = lane; i < S::IMG; i += 64) img[i] = 0.f;        // halo / overhang stay zero for good

    // A-operand offset of k-step 0 for each 16-row block of dW; a k-step advances 8 input pixels
    int aoff[AB];
    float amask[AB];
#pragma unroll
    for (int a = 0; a < AB; ++a) {
        int tap, ci;
        if (CIN == 4) { tap = 4 * a + (c >> 2); ci = c & 3; }
        else { tap = a / (CIN / 16); ci = (a % (CIN / 16)) * 16 + c; }
        const bool ok = tap < 9;
        const int ky = ok ? tap / 3 : 0, kx = ok ? tap % 3 : 0;
        aoff[a] = ky * S::ROWF + (2 * kq - S::PL + kx + 1) * PSF + ci;
        amask[a] = ok ? 1.f : 0.f;
    }

    f32x4 acc[AB][NBO];
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < NBO; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nstrips = nframes * Ho;
    const long frame_elems = (long)H * W * CIN;
    // No software pipeline: a strip is "load rows + dY -> LDS -> 30-90 MFMAs", and latency is hidden by
    // occupancy instead (<= 128 VGPRs and 4-13 KB of LDS per wave: up to 8 waves per SIMD).  Both
    // register-prefetch variants were slower -- a "cur = next" copy let the compiler rotate the
    // loop so each strip waited for its own loads (9 us per strip); explicit ping-pong sets kept
    // the distance but cost 128-400 VGPRs, i.e. the occupancy that was hiding the latency.
    RowStager<T, S, CIN, W> st;
    float colmask[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) colmask[s] = (4 * s + kq < Wo) ? 1.f : 0.f;
    float bsum[NBO];            // BNBWD: this lane's sum of da (channel 16b + c, pixels of quarter kq)
#pragma unroll
    for (int b = 0; b < NBO; ++b) bsum[b] = 0.f;
    for (int strip = wave; strip < nstrips; strip += NW) {
        const int n = strip / Ho, oy = strip - n * Ho;
        if constexpr (AFFINE) {
            // (x is the previous layer's PRE-norm activation: its batch-norm apply, per demonstration index, on the way in)
            const int ao = ((n / bnseq) % bnG) * CIN + 4 * (lane & (CIN / 4 - 1));
            st.load_affine(x, (long)n * frame_elems, 2 * oy - pt, H, lane, *reinterpret_cast<const f32x4*>(in_scale + ao),
                           *reinterpret_cast<const f32x4*>(in_shift + ao));
        } else {
            st.load(x, (long)n * frame_elems, 2 * oy - pt, H, lane);
        }
        float bv[KS][NBO];
        if constexpr (BNBWD) {
            float av[KS][NBO];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ox = 4 * s + kq;
                const int oxc = ox < Wo ? ox : Wo - 1;
#pragma unroll
                for (int b = 0; b < NBO; ++b) {
                    bv[s][b] = dy[((long)strip * Wo + oxc) * COUT + b * 16 + c];
                    av[s][b] = bw.act[((long)strip * Wo + oxc) * COUT + b * 16 + c];
                }
            }
            const int sq_ = d2p_div(n, D2pDiv{bw.seq_m, bw.seq_s});
            const int g_ = sq_ - d2p_div(sq_, D2pDiv{bw.g_m, bw.g_s}) * bw.G;
#pragma unroll
            for (int b = 0; b < NBO; ++b) {
                const f32x4 k = *reinterpret_cast<const f32x4*>(bw.coef + ((long)g_ * COUT + b * 16 + c) * 4);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const float a_ = av[s][b];
                    float d = (k.x * bv[s][b] + k.y * a_ + k.z) * d2p_lrelu_grad_from_out(a_);
                    if (!(4 * (KS - 1) + 3 < Wo)) d *= colmask[s];
                    bv[s][b] = d;
                    bsum[b] += d;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int ox = 4 * s + kq;
                const int oxc = ox < Wo ? ox : Wo - 1;
#pragma unroll
                for (int b = 0; b < NBO; ++b) bv[s][b] = dy[((long)strip * Wo + oxc) * COUT + b * 16 + c];
            }
        }
        st.store(img, lane);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float bc[NBO];
#pragma unroll
            for (int b = 0; b < NBO; ++b) bc[b] = (BNBWD || 4 * (KS - 1) + 3 < Wo) ? bv[s][b] : bv[s][b] * colmask[s];
#pragma unroll
            for (int a = 0; a < AB; ++a) {
                float av = img[aoff[a] + s * 8 * PSF];
                if (CIN == 4 && a == AB - 1) av *= amask[a];
#pragma unroll
                for (int b = 0; b < NBO; ++b) acc[a][b] = D2P_MFMA16(av, bc[b], acc[a][b]);
            }
        }
    }

    // fixed-order tree over the 4 waves (scratch reuses the strip area)
    __syncthreads();
    if constexpr (BNBWD) {
        // the bias gradient's share of this workgroup: the four pixel quarters of a wave by two shuffles, the waves
        // through LDS in wave order
        float* const bred = lds;
#pragma unroll
        for (int b = 0; b < NBO; ++b) {
            float v = bsum[b];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16) bred[wid * COUT + b * 16 + lane] = v;
        }
        __syncthreads();
        if (threadIdx.x < COUT)
            bw.bsum[(long)blockIdx.x * COUT + threadIdx.x] = ((bred[threadIdx.x] + bred[COUT + threadIdx.x]) +
                                                              bred[2 * COUT + threadIdx.x]) + bred[3 * COUT + threadIdx.x];
        __syncthreads();
    }
    float* const red = lds;
#pragma unroll
    for (int step = 1; step < 4; step *= 2) {
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
                for (int r = 0; r < 4; ++r) {
                    int row;                 // flattened (tap, ci) = D row 4*kq + r of block a
                    if (CIN == 4) row = 16 * a + 4 * kq + r;
                    else row = (a / (CIN / 16)) * CIN + (a % (CIN / 16)) * 16 + 4 * kq + r;
                    if (row < KK) slab[row * COUT + b * 16 + c] = acc[a][b][r];
                }
    }
}

// per-lane fp64 channel sums of a wave's epilogue values -> out[c*2 + {0,1}] for the workgroup: lanes (p, q) hold
// channels 16b + 4q + r of pixel lane p; the 16 pixel lanes by xor-shuffles, the 4 waves through LDS in wave order
template <int NB>
__device__ __forceinline__ void rows_fold_stats(const double (&sa)[NB][4], const double (&sb)[NB][4], int C, int wid, int p,
                                                int q, double* wsum, double* out) {
    __syncthreads();                      // (the staging area is free: every wave is behind its last strip)
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
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 2 * C) {
        const int c = tid >> 1, k = tid & 1;
        out[c * 2 + k] = ((wsum[(0 * C + c) * 2 + k] + wsum[(1 * C + c) * 2 + k]) + wsum[(2 * C + c) * 2 + k]) +
                         wsum[(3 * C + c) * 2 + k];
    }
}

// ------------------------------------------------------------------------------------------
// forward for the 3(4)-channel first layer: same staging, one output row per strip, 16-pixel MFMA
// tiles along the row (the last tile of a 40-pixel row is half empty: 17 % padding, the price of
// keeping every tile inside one row so that all offsets are per-lane constants).
template <int CIN, int COUT, int W>
struct FwdRowShape {
    static constexpr int Wo = r_out(W), PL = r_before(W);
    static constexpr int NTILE = (Wo + 15) / 16;
    static constexpr int PSF = CIN == 4 ? 4 : CIN + 4;
    static constexpr int ROWPIX = 32 * NTILE + 4;
    static constexpr int ROWF = ROWPIX * PSF;
    static constexpr int IMG = 3 * ROWF;
    static constexpr int NCH = CIN == 4 ? 3 : 9 * (CIN / 16);
    static constexpr int NB = COUT / 16;
    static constexpr int RP4 = W * CIN / 4, NLF = (3 * RP4 + 63) / 64;
    static constexpr int RP16 = W * CIN / 16, NLU = (3 * RP16 + 63) / 64;
    static_assert(W + 2 <= ROWPIX, "staged row must hold the image row and its halo");
};

// STATS (round 5, the ViZDoom-size layers): the launch also leaves the batch-norm statistics' partial sums behind
// (models/ops.py:14-33: conv -> lrelu -> batch norm; the separate partial-sum pass re-read the 655 MB of conv1's
// output).  Statistics are per demonstration index g = (frame / seq) % G, so the work is dealt out by index: workgroup
// (g, s) = blockIdx.x takes slice s of the strips of index g's frames -- strips j of the index, frame-major, j ->
// global strip ((j / (seq*Ho)) * G + g) * seq*Ho + j % (seq*Ho) -- and writes stats[((g*S + s)*COUT + c)*2 + {0,1}] =
// (sum, sum of squares) of its outputs in fp64: lanes, then waves in a fixed order (the layout bn_finalize reads).
struct RowsBn {
    int G, seq, S, per_slice;          // per_slice: strips of a slice
    double* stats;
};
template <int CIN, int COUT, int W, typename T, bool STATS = false>
__global__ void __launch_bounds__(256)
conv_rows_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int act,
                     float* __restrict__ y, int nframes, int H, int Ho, int pt, RowsBn bn) {
    using S = FwdRowShape<CIN, COUT, W>;
    constexpr int NCH = S::NCH, NB = S::NB, PSF = S::PSF, Wo = S::Wo, KK = 9 * CIN;
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, p = lane & 15, q = lane >> 4, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wid, NW = gridDim.x * 4;
    float* const img = lds + (size_t)wid * S::IMG;
    for (int i = lane; i < S::IMG; i += 64) img[i] = 0.f;

    // B-fragment offset of tile 0 per 16-deep chunk (a tile advances 32 input pixels); filter -> registers
    int boff[NCH];
    float bmask[NCH];
    float wr[NCH][4][NB];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        int tap, c0;
        if (CIN == 4) { tap = 4 * ch + q; c0 = 0; }
        else { tap = ch / (CIN / 16); c0 = (ch % (CIN / 16)) * 16 + 4 * q; }
        const bool ok = tap < 9;
        const int ky = ok ? tap / 3 : 0, kx = ok ? tap % 3 : 0;
        boff[ch] = ky * S::ROWF + (2 * p - S::PL + kx + 1) * PSF + c0;
        bmask[ch] = ok ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kg = CIN == 4 ? tap * 4 + j : tap * CIN + c0 + j;
#pragma unroll
            for (int b = 0; b < NB; ++b) wr[ch][j][b] = (ok && kg < KK) ? w[kg * COUT + b * 16 + p] : 0.f;
        }
    }
    f32x4 bv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) bv[b] = bias ? *reinterpret_cast<const f32x4*>(bias + b * 16 + 4 * q)
                                               : f32x4{0.f, 0.f, 0.f, 0.f};

    const int nstrips = nframes * Ho;
    const long frame_elems = (long)H * W * CIN;
    RowStager<T, S, CIN, W> st;
    // statistics: per-lane fp32 sums over the wave's strips (packed adds; a lane sees <= NTILE pixels per strip, ~100
    // values per sum), moved into fp64 ONCE behind the loop -- fp64 adds per value cost this MFMA-light kernel 13 % (a
    // VALU instruction beside an fp32 MFMA chain is paid in full), a flush branch in the loop as much
    double sa[NB][4], sb[NB][4];
    f32x4 fs[NB], fq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        fs[b] = fq[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) sa[b][r] = sb[b][r] = 0.0;
    }
    auto flush = [&]() {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { sa[b][r] += (double)fs[b][r]; sb[b][r] += (double)fq[b][r]; }
            fs[b] = fq[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // plain launch: strips interleaved over all waves; STATS: the strips of this workgroup's (index, slice), the
    // tensor's strip advanced without divisions: (sequence sq, strip sr inside it) of the index
    const int sg = STATS ? (int)blockIdx.x / bn.S : 0, ss = STATS ? (int)blockIdx.x - sg * bn.S : 0;
    const int seqs = bn.seq * Ho;                                   // strips of one (program, index) sequence
    const int per_idx = STATS ? (nframes / (bn.G * bn.seq)) * seqs : 0;
    const int j0 = STATS ? ss * bn.per_slice : wave, j1 = STATS ? min(j0 + bn.per_slice, per_idx) : nstrips;
    const int swid = __builtin_amdgcn_readfirstlane(wid);          // (scalar index arithmetic)
    int sq = STATS ? (j0 + swid) / seqs : 0, sr = STATS ? (j0 + swid) - sq * seqs : 0;
    for (int jj = STATS ? j0 + swid : j0; jj < j1; jj += STATS ? 4 : NW) {
        int strip = jj;
        if (STATS) {
            strip = (sq * bn.G + sg) * seqs + sr;
            sr += 4;
            if (sr >= seqs) { sr -= seqs; ++sq; }
        }
        const int n = strip / Ho, oy = strip - n * Ho;
        st.load(x, (long)n * frame_elems, 2 * oy - pt, H, lane);
        st.store(img, lane);
#pragma unroll
        for (int t = 0; t < S::NTILE; ++t) {
            f32x4 acc[2][NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[0][b] = acc[1][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                f32x4 bb = *reinterpret_cast<const f32x4*>(img + boff[ch] + t * 32 * PSF);
                if (CIN == 4 && ch == NCH - 1) bb *= bmask[ch];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[j & 1][b] = D2P_MFMA16(wr[ch][j][b], bb[j], acc[j & 1][b]);
            }
            const int ox = 16 * t + p;
            if (ox < Wo) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    f32x4 o = (acc[0][b] + acc[1][b]) + bv[b];
                    if (act) { o.x = d2p_lrelu(o.x); o.y = d2p_lrelu(o.y); o.z = d2p_lrelu(o.z); o.w = d2p_lrelu(o.w); }
                    *reinterpret_cast<f32x4*>(y + ((long)strip * Wo + ox) * COUT + b * 16 + 4 * q) = o;
                    if (STATS) {
                        fs[b] += o;
                        fq[b] += o * o;
                    }
                }
            }
        }
    }
    if (STATS) flush();
    if (STATS) rows_fold_stats<NB>(sa, sb, COUT, wid, p, q, reinterpret_cast<double*>(lds), bn.stats + (long)blockIdx.x * COUT * 2);
}

int g_rows_fwd_wgs = 2048;

template <int CIN, int COUT, int W, typename T>
int launch_rows_fwd(const ConvGeom& g, const T* x, const float* w, const float* bias, int act, float* y, hipStream_t st,
                    const ConvBnFold* bn = nullptr) {
    using S = FwdRowShape<CIN, COUT, W>;
    constexpr size_t lds_bytes = (size_t)4 * S::IMG * sizeof(float);
    static_assert(lds_bytes >= (size_t)4 * COUT * 2 * sizeof(double), "the statistics' fold reuses the staging area");
    static bool attr_set = false;
    if (!attr_set) {
        D2P_HIP(hipFuncSetAttribute((const void*)conv_rows_fwd_kernel<CIN, COUT, W, T, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        D2P_HIP(hipFuncSetAttribute((const void*)conv_rows_fwd_kernel<CIN, COUT, W, T, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * g.N * g.Ho * g.Wo * 9 * CIN * COUT);
    if (bn && bn->stats) {
        const int per_idx = g.N / bn->G * g.Ho;           // strips of one demonstration index
        RowsBn rb{bn->G, bn->seq, bn->S, ceil_div(per_idx, bn->S), bn->stats};
        hipLaunchKernelGGL((conv_rows_fwd_kernel<CIN, COUT, W, T, true>), dim3(bn->G * bn->S), dim3(256), lds_bytes, st, x, w,
                           bias, act, y, g.N, g.H, g.Ho, g.pt, rb);
        D2P_LAUNCH_CHECK("conv_rows_fwd_stats");
        return 1;
    }
    int nb = ceil_div(g.N * g.Ho, 4 * 4);                 // >= 4 strips per wave
    if (nb > g_rows_fwd_wgs) nb = g_rows_fwd_wgs;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL((conv_rows_fwd_kernel<CIN, COUT, W, T, false>), dim3(nb), dim3(256), lds_bytes, st, x, w, bias, act, y,
                       g.N, g.H, g.Ho, g.pt, RowsBn{});
    D2P_LAUNCH_CHECK("conv_rows_fwd");
    return 1;
}

// ------------------------------------------------------------------------------------------
// dgrad for 40x40x16 <- 20x20x32: dX[iy, ix, ci] = sum over the taps whose parity matches
// (ky = (iy + pt) mod 2 (+2), kx likewise) of dY[(iy + pt - ky)/2, (ix + pl - kx)/2, :] . W[ky, kx, ci, :].
// A strip is TWO input rows of equal row parity (so 2 x 20 = 40 pixels per column-parity class =
// 3 MFMA tiles, 17 % padding instead of 37 % for one row); the 2-3 dY rows they need are staged
// with a zero halo pixel on both sides.  Per column-parity class only its 1, 2 or 4 taps are
// multiplied (2.25 per pixel on average, as in the parity-class GEMMs of conv.hip).
template <int CIN, int COUT, int W>
struct DgradRowShape {
    static constexpr int Wo = r_out(W), PL = r_before(W);
    static constexpr int PSF = COUT + 4;                     // floats per staged dY pixel
    static constexpr int ROWPIX = Wo + 2;
    static constexpr int ROWF = ROWPIX * PSF;
    static constexpr int NROW = 3;
    static constexpr int IMG = NROW * ROWF + 4;
    static constexpr int CC = COUT / 16;
    static constexpr int RP4 = Wo * COUT / 4;                // float4 pieces per dY row
    static constexpr int NL = (NROW * RP4 + 63) / 64;
    // pixels of one column-parity class in a 2-row strip, and its 16-pixel tiles
    static constexpr int X0 = (PL & 1) ? 1 : 0;              // first ix with (ix + pl) even
    static constexpr int NC0 = (W - X0 + 1) / 2, NC1 = (W - (1 - X0) + 1) / 2;
    static constexpr int NT0 = (2 * NC0 + 15) / 16, NT1 = (2 * NC1 + 15) / 16;
    static_assert(CIN == 16, "one 16-channel block of dX per MFMA row tile");
};

// STATS (round 5): dx is the gradient w.r.t. the PREVIOUS layer's batch-norm output; the launch also leaves that batch
// norm's backward partial sums -- (sum dx, sum dx * xhat) per (demonstration index, channel), xhat = (act - mean) *
// rstd with `act` the previous layer's pre-norm activation, read at the positions this launch writes -- so the separate
// partial-sum pass (a read of act and dx, 1.3 GB at 80x80 frames) is gone.  Work dealt out by index as in the folding
// forward kernels: workgroup (g, s) = blockIdx.x takes slice s of the strips of index g's frames and writes
// stats[((g*S + s)*CIN + c)*2 + {0,1}] (fp64; lanes, then waves in a fixed order) -- what bn_finalize_bwd reads.
struct RowsDgradBn {
    int G, seq, S, per_slice;
    const float* act; const float* mean; const float* rstd;
    double* stats;
};
template <int CIN, int COUT, int W, bool STATS = false>
__global__ void __launch_bounds__(256)
conv_rows_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int nframes,
                       int H, int Ho, int pt, RowsDgradBn bn) {
    using S = DgradRowShape<CIN, COUT, W>;
    constexpr int CC = S::CC, PSF = S::PSF, Wo = S::Wo, PL = S::PL, NL = S::NL;
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, p = lane & 15, q = lane >> 4, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wid, NW = gridDim.x * 4;
    float* const img = lds + (size_t)wid * S::IMG;
    for (int i = lane; i < S::IMG; i += 64) img[i] = 0.f;

    // W^T -> registers: wr[tap][cc][j] = W[tap][ci = p][co = 16 cc + 4q + j]
    float wr[9][CC][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int j = 0; j < 4; ++j) wr[tap][cc][j] = w[(tap * CIN + p) * COUT + cc * 16 + 4 * q + j];

    // per-lane pixel of tile t in column-parity class ex: idx = 16 t + p -> (row j, column i of the class)
    constexpr int NTMAX = S::NT0 > S::NT1 ? S::NT0 : S::NT1;
    int poff[2][NTMAX], xoff[2][NTMAX];
#pragma unroll
    for (int ex = 0; ex < 2; ++ex) {
        const int nc = ex == 0 ? S::NC0 : S::NC1;
        const int ix0 = ex == 0 ? S::X0 : 1 - S::X0;
        const int oxb = (ix0 + PL - ex) / 2;                  // ox of column i = 0 for the tap with kx = ex
#pragma unroll
        for (int t = 0; t < NTMAX; ++t) {
            const int idx = 16 * t + p;
            const bool ok = idx < 2 * nc;
            const int idc = ok ? idx : 2 * nc - 1;            // padding lanes re-read the last pixel, store nothing
            const int j = idc / nc, i = idc - j * nc;
            poff[ex][t] = (j * S::ROWPIX + i + oxb + 1) * PSF + 4 * q;
            xoff[ex][t] = ok ? (2 * j * W + ix0 + 2 * i) * CIN + 4 * q : -1;
        }
    }

    // strips: per frame, rows of parity class ey = (iy + pt) & 1, two at a time
    const int e0 = pt & 1;                                   // first iy with (iy + pt) even
    const int n0 = (H - e0 + 1) / 2, n1 = (H - (1 - e0) + 1) / 2;
    const int s0 = (n0 + 1) / 2, s1 = (n1 + 1) / 2, per_frame = s0 + s1;
    const long nstrips = (long)nframes * per_frame;
    // STATS: the strips of this workgroup's (index, slice); the tensor's strip advanced without divisions
    const int sg = STATS ? (int)blockIdx.x / bn.S : 0, ss = STATS ? (int)blockIdx.x - sg * bn.S : 0;
    const int sps = bn.seq * per_frame;                                 // strips of one (program, index) sequence
    const int per_idx = STATS ? (nframes / (bn.G * bn.seq)) * sps : 0;
    const int swid = __builtin_amdgcn_readfirstlane(wid);
    const long j0 = STATS ? (long)ss * bn.per_slice + swid : wave;
    const long j1 = STATS ? min((long)(ss + 1) * bn.per_slice, (long)per_idx) : nstrips;
    int sq = STATS ? (int)(j0 / sps) : 0, sr = STATS ? (int)(j0 - (long)sq * sps) : 0;
    f32x4 mu4 = {0.f, 0.f, 0.f, 0.f}, rs4 = mu4, fs = mu4, fq = mu4;
    if (STATS) {
        mu4 = *reinterpret_cast<const f32x4*>(bn.mean + sg * CIN + 4 * q);
        rs4 = *reinterpret_cast<const f32x4*>(bn.rstd + sg * CIN + 4 * q);
    }
    for (long jj = j0; jj < j1; jj += STATS ? 4 : NW) {
        long strip = jj;
        if (STATS) {
            strip = (long)(sq * bn.G + sg) * sps + sr;
            sr += 4;
            if (sr >= sps) { sr -= sps; ++sq; }
        }
        const int n = (int)(strip / per_frame);
        int sidx = (int)(strip - (long)n * per_frame);
        const int ey = sidx >= s0 ? 1 : 0;
        if (ey) sidx -= s0;
        const int iy0 = (ey ? 1 - e0 : e0) + 4 * sidx;       // rows iy0 and iy0 + 2
        // staged dY rows r = 0..2 <-> oy = oy_first + r ; ey = 0: taps ky = 0 (r = j + 1) and ky = 2 (r = j);
        // ey = 1: tap ky = 1 (r = j)
        const int oy_first = ey ? (iy0 + pt - 1) / 2 : (iy0 + pt) / 2 - 1;
        f32x4 st[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int jj = i * 64 + lane;
            const int r = jj / S::RP4, wi = jj - r * S::RP4;
            const int oy = oy_first + r;
            const bool ok = (r < S::NROW) & (oy >= 0) & (oy < Ho);
            const int oyc = min(max(oy, 0), Ho - 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(dy + (((long)n * Ho + oyc) * Wo) * COUT + (r < S::NROW ? wi : 0) * 4);
            st[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int jj = i * 64 + lane;
            const int r = jj / S::RP4, wi = jj - r * S::RP4;
            if (r < S::NROW)
                *reinterpret_cast<f32x4*>(img + (r * S::ROWPIX + wi / (COUT / 4) + 1) * PSF + (wi % (COUT / 4)) * 4) = st[i];
        }
        float* const out = dx + ((long)n * H + iy0) * W * CIN;
        const bool row1 = iy0 + 2 < H;
        // STATS: the activation at the positions this strip writes, requested NOW (a load issued in the epilogue, where
        // its value is needed, exposed a full memory latency per tile: + 240 us); lanes that store nothing re-read 0
        f32x4 apre[2][NTMAX];
        if (STATS) {
            const float* const abase = bn.act + (out - dx);
#pragma unroll
            for (int ex = 0; ex < 2; ++ex)
#pragma unroll
                for (int t = 0; t < NTMAX; ++t) {
                    const int xo = xoff[ex][t];
                    int ao = (xo >= 0 && (row1 || xo < W * CIN)) ? xo : 0;
                    D2P_OPAQUE_I(ao);
                    apre[ex][t] = *reinterpret_cast<const f32x4*>(abase + ao);
                }
        }
        auto run = [&](auto EYc) {
            constexpr int EY = decltype(EYc)::value;
#pragma unroll
            for (int ex = 0; ex < 2; ++ex) {
                const int nt = ex == 0 ? S::NT0 : S::NT1;
#pragma unroll
                for (int t = 0; t < NTMAX; ++t) {
                    if (t >= nt) continue;
                    f32x4 acc[2];
                    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ty = 0; ty < (EY ? 1 : 2); ++ty)
#pragma unroll
                        for (int tx = 0; tx < (ex ? 1 : 2); ++tx) {
                            const int ky = EY ? 1 : 2 * ty, kx = ex ? 1 : 2 * tx;
                            // row: ky = 0 -> r = j + 1, ky = 2 -> r = j, ky = 1 -> r = j;  column: kx = 2 -> ox - 1
                            const int toff = ((EY == 0 && ky == 0) ? S::ROWF : 0) - ((ex == 0 && kx == 2) ? PSF : 0);
#pragma unroll
                            for (int cc = 0; cc < CC; ++cc) {
                                const f32x4 bb = *reinterpret_cast<const f32x4*>(img + poff[ex][t] + toff + cc * 16);
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[j & 1] = D2P_MFMA16(wr[ky * 3 + kx][cc][j], bb[j], acc[j & 1]);
                            }
                        }
                    const int xo = xoff[ex][t];
                    // second row of the strip may be past the image (odd row count)
                    if (xo >= 0 && (row1 || xo < W * CIN)) {
                        const f32x4 o = acc[0] + acc[1];
                        *reinterpret_cast<f32x4*>(out + xo) = o;
                        if (STATS) {
                            fs += o;
                            fq += o * ((apre[ex][t] - mu4) * rs4);
                        }
                    }
                }
            }
        };
        if (ey) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 0>{});
    }
    if (STATS) {
        // lanes (p, q) hold channels 4q + r of their pixels: fp32 sums of a lane's few hundred values -> fp64, the 16
        // pixel lanes by xor-shuffles, the 4 waves through LDS in wave order
        __syncthreads();
        double* wsum = reinterpret_cast<double*>(lds);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double u = (double)fs[r], v = (double)fq[r];
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                u += __shfl_xor(u, off, 64);
                v += __shfl_xor(v, off, 64);
            }
            if (p == 0) {
                wsum[(wid * CIN + 4 * q + r) * 2] = u;
                wsum[(wid * CIN + 4 * q + r) * 2 + 1] = v;
            }
        }
        __syncthreads();
        const int tid = threadIdx.x;
        if (tid < 2 * CIN) {
            const int c = tid >> 1, k = tid & 1;
            bn.stats[((long)blockIdx.x * CIN + c) * 2 + k] =
                ((wsum[(0 * CIN + c) * 2 + k] + wsum[(1 * CIN + c) * 2 + k]) + wsum[(2 * CIN + c) * 2 + k]) +
                wsum[(3 * CIN + c) * 2 + k];
        }
    }
}

int g_rows_dgrad_wgs = 512;     // measured best (occupancy-limited: 160 VGPRs, 38 KB LDS per workgroup)

// strips of one frame in conv_rows_dgrad_kernel's enumeration (two rows of equal parity per strip)
static int rows_dgrad_per_frame(const ConvGeom& g) {
    const int e0 = g.pt & 1;
    const int n0 = (g.H - e0 + 1) / 2, n1 = (g.H - (1 - e0) + 1) / 2;
    return (n0 + 1) / 2 + (n1 + 1) / 2;
}
template <int CIN, int COUT, int W>
int launch_rows_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st,
                      const ConvDgradBn* bn = nullptr) {
    using S = DgradRowShape<CIN, COUT, W>;
    constexpr size_t lds_bytes = (size_t)4 * S::IMG * sizeof(float);
    static_assert(lds_bytes >= (size_t)4 * CIN * 2 * sizeof(double), "the statistics' fold reuses the staging area");
    static bool attr_set = false;
    if (!attr_set) {
        D2P_HIP(hipFuncSetAttribute((const void*)conv_rows_dgrad_kernel<CIN, COUT, W, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        D2P_HIP(hipFuncSetAttribute((const void*)conv_rows_dgrad_kernel<CIN, COUT, W, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * g.N * g.Ho * g.Wo * 9 * CIN * COUT);
    if (bn) {
        const int per_idx = g.N / bn->G * rows_dgrad_per_frame(g);
        RowsDgradBn rb{bn->G, bn->seq, bn->S, ceil_div(per_idx, bn->S), bn->act, bn->mean, bn->rstd, bn->stats};
        hipLaunchKernelGGL((conv_rows_dgrad_kernel<CIN, COUT, W, true>), dim3(bn->G * bn->S), dim3(256), lds_bytes, st, dy, w,
                           dx, g.N, g.H, g.Ho, g.pt, rb);
        D2P_LAUNCH_CHECK("conv_rows_dgrad_stats");
        return 1;
    }
    long strips = (long)g.N * ((g.H + 3) / 4 * 2 + 2);
    int nb = (int)((strips + 15) / 16);
    if (nb > g_rows_dgrad_wgs) nb = g_rows_dgrad_wgs;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL((conv_rows_dgrad_kernel<CIN, COUT, W, false>), dim3(nb), dim3(256), lds_bytes, st, dy, w, dx, g.N, g.H,
                       g.Ho, g.pt, RowsDgradBn{});
    D2P_LAUNCH_CHECK("conv_rows_dgrad");
    return 1;
}

int g_rows_wgrad_wgs = 0;      // 0: per-layer default (template CAP)

// out[c] = sum over workgroups of bsum[workgroup][c] (fp64, lanes stride the workgroups: deterministic)
__global__ void __launch_bounds__(256)
rows_bias_reduce_kernel(int C, int nblocks, const float* __restrict__ bsum, float* __restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double acc = 0.0;
    for (int b = lane; b < nblocks; b += 64) acc += (double)bsum[(long)b * C + c];
    acc = wave_reduce_sum(acc);
    if (lane == 0) out[c] = (float)acc;
}

template <int CIN, int COUT, int W, typename T, int CAP>
struct RowsWgrad {
    using S = RowShape<CIN, COUT, W>;
    static constexpr int ACC = S::AB * S::NBO * 4;
    static constexpr size_t img_bytes = (size_t)4 * S::IMG * sizeof(float);
    static constexpr size_t red_bytes = (size_t)2 * ACC * 64 * sizeof(float);
    static constexpr size_t lds_bytes = img_bytes > red_bytes ? img_bytes : red_bytes;
    static int blocks(const ConvGeom& g) {
        int b = ceil_div(g.N * g.Ho, 4 * 8);             // >= 8 strips per wave
        const int cap = g_rows_wgrad_wgs > 0 ? g_rows_wgrad_wgs : CAP;   // measured: kernel keeps scaling to 2048
        if (b > cap) b = cap;                                              // workgroups, the combine pass does not
        return b < 1 ? 1 : b;
    }
    // the weight gradient with the layer's batch-norm backward folded in (conv_rows_wgrad_kernel, BNBWD): ws holds the
    // slabs and, behind them, the workgroups' bias-gradient shares
    static int run_bnbwd(const ConvGeom& g, const T* x, const float* act, const float* dy, const float* coef, int G, int seq,
                         float* dw, float* dbias, void* ws, size_t ws_bytes, hipStream_t st) {
        static bool attr_set = false;
        if (!attr_set) {
            D2P_HIP(hipFuncSetAttribute((const void*)conv_rows_wgrad_kernel<CIN, COUT, W, T, false, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            attr_set = true;
        }
        const int KK = 9 * CIN, nb = blocks(g);
        const size_t slab_bytes = (size_t)nb * KK * COUT * sizeof(float);
        D2P_REQUIRE(ws && ws_bytes >= slab_bytes + (size_t)nb * COUT * sizeof(float), D2P_EWS,
                    "conv wgrad (bn backward): workspace too small (%zu bytes)", ws_bytes);
        float* slabs = (float*)ws;
        float* bsum = (float*)((char*)ws + slab_bytes);
        const D2pDiv ds = d2p_make_div(seq), dg = d2p_make_div(G);
        RowsBnBwd bw{act, coef, bsum, G, ds.m, ds.s, dg.m, dg.s};
        D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * g.N * g.Ho * g.Wo * KK * COUT);
        hipLaunchKernelGGL((conv_rows_wgrad_kernel<CIN, COUT, W, T, false, true>), dim3(nb), dim3(256), lds_bytes, st, x, dy,
                           slabs, g.N, g.H, g.Ho, g.pt, 1, 1, (const float*)nullptr, (const float*)nullptr, bw);
        D2P_LAUNCH_CHECK("conv_rows_wgrad_bnbwd");
        EpiDense ep{dw, COUT, nullptr, 0, 0};
        const long total = (long)KK * COUT;
        hipLaunchKernelGGL((gemm_splitk_reduce_kernel<EpiDense>), dim3((int)((total * 16 + 255) / 256)), dim3(256), 0,
                           st, ep, slabs, KK, COUT, nb);
        D2P_LAUNCH_CHECK("conv_rows_wgrad_combine");
        if (dbias) {
            hipLaunchKernelGGL(rows_bias_reduce_kernel, dim3(ceil_div(COUT, 4)), dim3(256), 0, st, COUT, nb, bsum, dbias);
            D2P_LAUNCH_CHECK("rows_bias_reduce");
        }
        return 1;
    }
    static int run(const ConvGeom& g, const T* x, const float* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st,
                   const ConvBnFold* bn = nullptr) {
        constexpr bool CAN_AFFINE = std::is_same<T, float>::value && CIN >= 16;
        static bool attr_set = false;
        if (!attr_set) {
            D2P_HIP(hipFuncSetAttribute((const void*)conv_rows_wgrad_kernel<CIN, COUT, W, T, false>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            if constexpr (CAN_AFFINE)
                D2P_HIP(hipFuncSetAttribute((const void*)conv_rows_wgrad_kernel<CIN, COUT, W, T, true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            attr_set = true;
        }
        const bool affine = bn && bn->in_scale;
        if (affine && !CAN_AFFINE) return 0;
        const int KK = 9 * CIN, nb = blocks(g);
        D2P_REQUIRE(ws && ws_bytes >= (size_t)nb * KK * COUT * sizeof(float), D2P_EWS,
                    "conv wgrad: workspace too small (%zu bytes)", ws_bytes);
        float* slabs = (float*)ws;
        D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * g.N * g.Ho * g.Wo * KK * COUT);
        if constexpr (CAN_AFFINE) {
            if (affine)
                hipLaunchKernelGGL((conv_rows_wgrad_kernel<CIN, COUT, W, T, true>), dim3(nb), dim3(256), lds_bytes, st, x, dy,
                                   slabs, g.N, g.H, g.Ho, g.pt, bn->G, bn->seq, bn->in_scale, bn->in_shift, RowsBnBwd{});
        }
        if (!affine)
            hipLaunchKernelGGL((conv_rows_wgrad_kernel<CIN, COUT, W, T, false>), dim3(nb), dim3(256), lds_bytes, st, x, dy,
                               slabs, g.N, g.H, g.Ho, g.pt, 1, 1, (const float*)nullptr, (const float*)nullptr, RowsBnBwd{});
        D2P_LAUNCH_CHECK("conv_rows_wgrad");
        EpiDense ep{dw, COUT, nullptr, 0, 0};
        const long total = (long)KK * COUT;
        hipLaunchKernelGGL((gemm_splitk_reduce_kernel<EpiDense>), dim3((int)((total * 16 + 255) / 256)), dim3(256), 0,
                           st, ep, slabs, KK, COUT, nb);
        D2P_LAUNCH_CHECK("conv_rows_wgrad_combine");
        return 1;
    }
};

}   // namespace

void d2p_conv_rows_tune(int wgrad_wgs) { g_rows_wgrad_wgs = wgrad_wgs > 0 ? wgrad_wgs : 0; }
void d2p_conv_rows_fwd_tune(int wgs) { if (wgs > 0) g_rows_fwd_wgs = wgs; }

static int rows_key(const ConvGeom& g) {
    if (g.Cin == 4 && g.Cout == 16 && g.W == 80) return 1;
    if (g.Cin == 16 && g.Cout == 32 && g.W == 40) return 2;
    return 0;
}

int d2p_conv_rows_fwd(const ConvGeom& g, const void* x, int x_is_u8, const float* w, const float* bias, int act,
                      float* y, hipStream_t st, const ConvBnFold* bn) {
    if (!(g.Cin == 4 && g.Cout == 16 && g.W == 80) || g.N < 1) return 0;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || (bias && ((uintptr_t)bias & 15))) return 0;
    if (bn && (bn->in_scale || !bn->stats)) return 0;       // (the first layer's input is the frames: no affine to fold)
    if (x_is_u8) return launch_rows_fwd<4, 16, 80, uint8_t>(g, (const uint8_t*)x, w, bias, act, y, st, bn);
    return launch_rows_fwd<4, 16, 80, float>(g, (const float*)x, w, bias, act, y, st, bn);
}

int d2p_conv_rows_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st,
                        const ConvDgradBn* bn) {
    if (!(g.Cin == 16 && g.Cout == 32 && g.W == 40) || g.N < 1) return 0;
    if (((uintptr_t)dy & 15) || ((uintptr_t)dx & 15)) return 0;
    if (bn && (bn->G < 1 || bn->seq < 1 || bn->S < 1 || g.N % (bn->G * bn->seq) != 0 || !bn->act || !bn->stats ||
               (((uintptr_t)bn->act | (uintptr_t)bn->mean | (uintptr_t)bn->rstd) & 15)))
        return 0;
    return launch_rows_dgrad<16, 32, 40>(g, dy, w, dx, st, bn);
}
// slices per demonstration index of the statistics-folding input-gradient launch (0: no such kernel)
int d2p_conv_rows_dgrad_slices(const ConvGeom& g, int G, int seq) {
    if (!(g.Cin == 16 && g.Cout == 32 && g.W == 40) || g.N < 1 || G < 1 || seq < 1 || g.N % (G * seq) != 0) return 0;
    const long units = (long)g.N / G * rows_dgrad_per_frame(g);
    long S = g_rows_dgrad_wgs / G;           // (as many workgroups as the plain launch: occupancy-limited, 512 measured best)
    if (S < 1) S = 1;
    if (S * 16 > units) S = units / 16;
    return (int)(S < 1 ? 1 : S);
}
void d2p_conv_rows_dgrad_tune(int wgs) { if (wgs > 0) g_rows_dgrad_wgs = wgs; }

size_t d2p_conv_rows_wgrad_ws(const ConvGeom& g) {
    if (!rows_key(g)) return 0;
    // (slabs, and the bias-gradient shares of the batch-norm-backward form behind them)
    return (size_t)(g_rows_wgrad_wgs > 2048 ? g_rows_wgrad_wgs : 2048) * (9 * g.Cin + 1) * g.Cout * sizeof(float);
}

// the first layer's weight gradient with its batch-norm backward folded in (RowsBnBwd): 1 = taken, 0 = no such kernel
int d2p_conv_rows_wgrad_bnbwd(const ConvGeom& g, const void* x, int x_is_u8, const float* act, const float* dy,
                              const float* coef, int G, int seq, float* dw, float* dbias, void* ws, size_t ws_bytes,
                              hipStream_t st) {
    if (rows_key(g) != 1 || g.N < 1 || G < 1 || seq < 1) return 0;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 3) || ((uintptr_t)act & 3) || ((uintptr_t)coef & 15)) return 0;
    if (x_is_u8) return RowsWgrad<4, 16, 80, uint8_t, 1024>::run_bnbwd(g, (const uint8_t*)x, act, dy, coef, G, seq, dw, dbias, ws, ws_bytes, st);
    return RowsWgrad<4, 16, 80, float, 1024>::run_bnbwd(g, (const float*)x, act, dy, coef, G, seq, dw, dbias, ws, ws_bytes, st);
}

int d2p_conv_rows_wgrad(const ConvGeom& g, const void* x, int x_is_u8, const float* dy, float* dw, void* ws,
                        size_t ws_bytes, hipStream_t st, const ConvBnFold* bn) {
    const int key = rows_key(g);
    if (!key || g.N < 1) return 0;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 3)) return 0;
    if (bn && bn->in_scale) {
        if (key != 2 || x_is_u8 || (((uintptr_t)bn->in_scale | (uintptr_t)bn->in_shift) & 15)) return 0;
        return RowsWgrad<16, 32, 40, float, 512>::run(g, (const float*)x, dy, dw, ws, ws_bytes, st, bn);
    }
    if (key == 1) {
        if (x_is_u8) return RowsWgrad<4, 16, 80, uint8_t, 1024>::run(g, (const uint8_t*)x, dy, dw, ws, ws_bytes, st);
        return RowsWgrad<4, 16, 80, float, 1024>::run(g, (const float*)x, dy, dw, ws, ws_bytes, st);
    }
    if (x_is_u8) return 0;
    return RowsWgrad<16, 32, 40, float, 512>::run(g, (const float*)x, dy, dw, ws, ws_bytes, st);
}
