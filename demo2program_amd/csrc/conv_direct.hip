// K1 (direct back end): NHWC 3x3 stride-2 TF-"SAME" convolution for the narrow layers of the
// demonstration encoder (models/ops.py:27-33 called from models/model_full.py:216-231), written
// around v_mfma_f32_16x16x4_f32 with the filter held in registers and no LDS staging.
//
// Why a second back end: the Karel layers are 16/32/48 channels wide over 8x8 .. 1x1 frames.
// On the 32-wide tiles of the implicit-im2col GEMM (conv.hip) half of every MFMA is padding,
// and each workgroup spends more time on prologue/epilogue than on its 9 K-iterations.  Here
//   * one wave owns a tile of 16 output pixels x all output channels,
//   * the whole filter (<= 108 VGPRs per lane) is loaded once per wave and reused for every
//     tile the wave walks through,
//   * activations go HBM/L2 -> VGPR as one 16-byte load per (pixel, tap, 4 channels) and feed
//     the MFMA B operand directly: a lane's float4 holds 4 consecutive k, MFMA j of a 16-deep
//     chunk consumes component j (the reduction index is permuted identically on both
//     operands, which a dot product does not notice),
//   * the filter is the A operand, so D comes out as [channel][pixel]: a lane ends up with 4
//     consecutive channels of one pixel = one 16-byte NHWC store.
// Fragment layout of v_mfma_f32_16x16x4_f32 (lane l): A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15], D[i = 4*(l>>4) + r][j = l&15], r = 0..3.
//
// fwd   : Y[pix, co]  = sum_(tap,c) x[pix@tap, c] * W[tap, c, co]           (+bias, lrelu)
// dgrad : dX[pix, c]  = sum_(tap,co) dY[(pix+pad-tap)/2, co] * W[tap, c, co]   per parity class
// wgrad : dW[tap,c,co] = sum_pix x[pix@tap, c] * dY[pix, co]    (reduction index = pixel)
#include "conv_geom.h"
#include "gemm_core.h"
#include "prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

static int g_direct_fwd = 2, g_direct_dgrad = 2, g_direct_wgrad = 2;   // 0 GEMM, 1 gather kernels, 2 + whole-frame kernels
void d2p_conv_direct_enable(int fwd, int dgrad, int wgrad) {
    g_direct_fwd = fwd; g_direct_dgrad = dgrad; g_direct_wgrad = wgrad;
}

// ---- exact division of n < 2^31 by a run-time constant (Granlund-Montgomery) ---------------
struct FastDiv {
    uint32_t m;
    int s;
};
static FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.s = 0;
    while ((1L << f.s) < d) ++f.s;
    const uint64_t num = 1ULL << (31 + f.s);
    f.m = (uint32_t)((num + d - 1) / d);
    return f;
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) {
    return (int)((uint32_t)(((uint64_t)(uint32_t)n * f.m) >> 31) >> f.s);
}

struct DirectGeom {
    int N, H, W, Ho, Wo, pt, pl;
    int P;                  // output pixels N*Ho*Wo
    FastDiv d_howo, d_wo;
};
static DirectGeom make_direct(const ConvGeom& g) {
    DirectGeom d;
    d.N = g.N; d.H = g.H; d.W = g.W; d.Ho = g.Ho; d.Wo = g.Wo; d.pt = g.pt; d.pl = g.pl;
    d.P = g.N * g.Ho * g.Wo;
    d.d_howo = make_fastdiv(g.Ho * g.Wo);
    d.d_wo = make_fastdiv(g.Wo);
    return d;
}

// ---- compile-time tap sets --------------------------------------------------------------
// MASK bit (3*ky + kx) is set when that tap can touch the image at all for the geometry;
// the 2x2 -> 1x1 Karel layer only ever sees ky, kx in {0, 1}  (mask 0x1B).
constexpr int popcount9(int m) { int c = 0; for (int i = 0; i < 9; ++i) c += (m >> i) & 1; return c; }
constexpr int nth_tap(int m, int n) {
    for (int i = 0; i < 9; ++i) if ((m >> i) & 1) { if (n == 0) return i; --n; }
    return 0;
}
static int tap_mask_for(const ConvGeom& g) {
    int m = 0;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            bool any_y = false, any_x = false;
            for (int oy = 0; oy < g.Ho; ++oy) { int iy = 2 * oy - g.pt + ky; any_y |= iy >= 0 && iy < g.H; }
            for (int ox = 0; ox < g.Wo; ++ox) { int ix = 2 * ox - g.pl + kx; any_x |= ix >= 0 && ix < g.W; }
            if (any_y && any_x) m |= 1 << (3 * ky + kx);
        }
    return m;
}

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ldg4(const uint8_t* p) {
    const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
    f32x4 v;
    v.x = (float)(t & 255u); v.y = (float)((t >> 8) & 255u);
    v.z = (float)((t >> 16) & 255u); v.w = (float)(t >> 24);
    return v;
}
__device__ __forceinline__ float ldg1(const float* p) { return *p; }
__device__ __forceinline__ float ldg1(const uint8_t* p) { return (float)*p; }

// Out-of-image taps load from the clamped coordinate (always a legal address) and are zeroed
// when consumed: "ok ? addr : 0" selects invite the compiler to branch around the address
// arithmetic or the load itself, and every such join costs a conservative s_waitcnt.
__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }
#define D2P_OPAQUE(v) asm volatile("" : "+v"(v))

#define D2P_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ==========================================================================================
// forward
// ==========================================================================================
template <int CIN, int MASK>
struct FwdShape {
    static constexpr int CB = CIN >= 16 ? CIN / 16 : 1;                 // 16-channel blocks per tap
    static constexpr int NCH = CIN >= 16 ? popcount9(MASK) * CB : 3;    // 16-deep K chunks
};

// B-operand gather for one 16-pixel tile: v[ch] = 4 consecutive k of chunk ch for this lane's
// pixel; ok bit ch says whether the tap was inside the image (select deferred to use).
// PADROW (the batch-norm-folding form with an input affine): an out-of-image tap reads the "pad pixel" of the
// workgroup's demonstration index instead -- padoff = its offset in x -- whose value the affine maps to zero, so
// nothing is masked afterwards (okf is not written).
template <int CIN, int MASK, typename T, bool PADROW = false>
__device__ __forceinline__ void fwd_gather(const DirectGeom& g, const T* __restrict__ x, int pix, int q,
                                           f32x4 (&v)[FwdShape<CIN, MASK>::NCH],
                                           float (&okf)[FwdShape<CIN, MASK>::NCH], unsigned padoff = 0u) {
    constexpr int CB = FwdShape<CIN, MASK>::CB, NCH = FwdShape<CIN, MASK>::NCH;
    const bool valid = pix < g.P;
    const int pc = valid ? pix : 0;
    const int n = fdiv(pc, g.d_howo);
    const int rem = pc - n * g.Ho * g.Wo;
    const int oy = fdiv(rem, g.d_wo), ox = rem - oy * g.Wo;
    const int iy0 = 2 * oy - g.pt, ix0 = 2 * ox - g.pl;
    const int fbase = n * g.H;
    if (CIN >= 16) {
        // three row terms and three column terms per tile; a tap is one add of each
        int rowoff[3], coloff[3];
        bool rok[3], cok[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int iy = iy0 + k, ix = ix0 + k;
            rok[k] = valid & ((unsigned)iy < (unsigned)g.H);
            cok[k] = (unsigned)ix < (unsigned)g.W;
            rowoff[k] = (fbase + clampi(iy, g.H - 1)) * g.W * CIN;
            coloff[k] = clampi(ix, g.W - 1) * CIN + 4 * q;
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int tap = nth_tap(MASK, ch / CB);
            const int ky = tap / 3, kx = tap % 3;
            unsigned off = (unsigned)(rowoff[ky] + coloff[kx] + (ch % CB) * 16);
            if (PADROW) off = (rok[ky] & cok[kx]) ? off : padoff + (unsigned)((ch % CB) * 16 + 4 * q);   // (a select of the offset, not of the load)
            D2P_OPAQUE(off);
            v[ch] = ldg4(x + off);
            if (!PADROW) okf[ch] = (rok[ky] & cok[kx]) ? 1.f : 0.f;
        }
    } else {   // CIN == 4: a chunk is 4 taps x 4 channels, this lane's tap is 4*ch + q
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int tap = 4 * ch + q;
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            const int iy = iy0 + ky, ix = ix0 + kx;
            const bool ok = valid & (tap < 9) & ((unsigned)iy < (unsigned)g.H) & ((unsigned)ix < (unsigned)g.W);
            int off = ((fbase + clampi(iy, g.H - 1)) * g.W + clampi(ix, g.W - 1)) * CIN;
            D2P_OPAQUE(off);
            v[ch] = ldg4(x + off);
            okf[ch] = ok ? 1.f : 0.f;
        }
    }
}

// BN (round 5, the ViZDoom-size layers; ConvBnFold in conv_geom.h): the work is dealt out by demonstration index --
// workgroup (g, s) = blockIdx.x takes slice s of the 16-pixel tiles of index g's frames (tile j of the index ->
// global tile ((j / tps) * G + g) * tps + j % tps, tps = tiles of one (program, index) sequence) -- so that
//   STATS:  the launch leaves stats[((g*S + s)*COUT + c)*2 + {0,1}] = (sum, sum of squares) of its outputs (fp64; lanes,
//           then waves in a fixed order) -- no separate partial-sum pass over the output;
//   AFFINE: its input is x * in_scale[g] + in_shift[g], the previous layer's batch-norm apply (x = that layer's
//           pre-norm activation; the normalised tensor is never written).  Costs the loop NOTHING over the plain kernel
//           (an fp32 VALU instruction beside an fp32 MFMA chain is paid in full, profiles/r05_mfma_gate_corun.txt): the
//           workgroup's index is fixed, so the scale goes into its filter registers once (W' = W * scale[c]), the
//           loop adds shift / scale to what it loads, and an out-of-image tap loads the index's PAD PIXEL --
//           -shift / scale, G*CIN floats behind x's last element, written with the statistics
//           (d2p_bn_stats_from_partials) -- so that tap contributes W' * 0: no mask multiply either.
struct DirectBn {
    int G, S, tps, per_slice, per_idx;      // per_idx: tiles of one index, per_slice: tiles of a slice
    const float* in_scale; const float* in_nshift;    // (in_nshift: shift / scale, what the loop adds)
    double* stats;
    unsigned pad0;                          // AFFINE: float offset of index 0's pad pixel in x
};
template <int CIN, int COUT, int MASK, typename T, bool STATS = false, bool AFFINE = false>
__global__ void __launch_bounds__(256)
conv_direct_fwd_kernel(DirectGeom g, const T* __restrict__ x, const float* __restrict__ w,
                       const float* __restrict__ bias, int act, float* __restrict__ y, int ntiles,
                       int keep, DirectBn bn) {
    constexpr int CB = FwdShape<CIN, MASK>::CB, NCH = FwdShape<CIN, MASK>::NCH, NB = COUT / 16;
    constexpr bool BN = STATS || AFFINE;
    const int lane = threadIdx.x & 63, p = lane & 15, q = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), NW = BN ? 4 : gridDim.x * 4;
    const int sg = BN ? (int)blockIdx.x / bn.S : 0, ss = BN ? (int)blockIdx.x - sg * bn.S : 0;
    // BN: `tile` below counts the tiles of this workgroup's index; gtile() maps it to the tile of the tensor
    const int tlo = BN ? ss * bn.per_slice : 0, thi = BN ? min(tlo + bn.per_slice, bn.per_idx) : ntiles;
    auto gtile = [&](int j) {
        if (!BN) return j;
        const int b_ = j / bn.tps;
        return (b_ * bn.G + sg) * bn.tps + (j - b_ * bn.tps);
    };
    f32x4 asc[CB], ash[CB];
    const unsigned padoff = AFFINE ? bn.pad0 + (unsigned)(sg * CIN) : 0u;
    if (AFFINE) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            asc[cb] = ldg4(bn.in_scale + sg * CIN + cb * 16 + 4 * q);
            // the pad pixel holds -shift / scale: what the loop adds is its negative
            ash[cb] = -ldg4(x + padoff + cb * 16 + 4 * q);
        }
    }
    // statistics: per-lane fp32 sums over the wave's tiles (packed adds: 8 instructions per tile; a lane sees one pixel
    // per tile, a few dozen values per sum), moved into fp64 ONCE behind the loop.  fp64 adds per tile cost 25 % of this
    // kernel (a VALU instruction beside an fp32 MFMA chain is paid in full), a flush every few tiles nearly as much --
    // a branch in the loop makes hipcc wait for the prefetch at the join.
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

    // filter -> registers (A operand): wr[ch][j][blk] = W[k = chunk ch, 4q + j][co = 16 blk + p]
    float wr[NCH][4][NB];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int kg;
            bool ok = true;
            if (CIN >= 16) kg = nth_tap(MASK, ch / CB) * CIN + (ch % CB) * 16 + 4 * q + j;
            else { const int tap = 4 * ch + q; ok = tap < 9; kg = tap * 4 + j; }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float t = w[(ok ? kg : 0) * COUT + b * 16 + p];
                wr[ch][j][b] = ok ? t : 0.f;
            }
        }
    if (AFFINE) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < NB; ++b) wr[ch][j][b] *= asc[ch % CB][j];
    }
    f32x4 bv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (bias) bv[b] = ldg4(bias + b * 16 + 4 * q);
        else bv[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // Two register sets in ping-pong (no copies: a "cur = nxt" rotation makes the compiler wait
    // for the prefetch at the bottom of every iteration).
    // (gt: the tile of the tensor; gtn: the wave's next one, < 0 past its last)
    auto step = [&](int gt, int gtn, const f32x4 (&cur)[NCH], const float (&okc)[NCH], f32x4 (&nxt)[NCH],
                    float (&okn)[NCH]) {
        // prefetch the wave's next tile (unconditional: past the end it re-reads a clamped pixel)
        fwd_gather<CIN, MASK, T, AFFINE>(g, x, gtn >= 0 ? gtn * 16 + p : g.P, q, nxt, okn, padoff);
        __builtin_amdgcn_sched_barrier(0);
        // two accumulator sets (even / odd chunks) so consecutive MFMAs are independent
        f32x4 acc[2][NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[0][b] = acc[1][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            f32x4 bb;                             // out-of-image taps -> 0 (clamped loads are finite)
            if (AFFINE) bb = cur[ch] + ash[ch % CB];        // (x + shift / scale; a pad pixel gives exactly 0)
            else bb = cur[ch] * okc[ch];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[j & 1][b] = D2P_MFMA16(wr[ch][j][b], bb[j], acc[j & 1][b]);
        }
        const int pix = gt * 16 + p;
        if (pix < g.P) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                f32x4 o = (acc[0][b] + acc[1][b]) + bv[b];
                if (act) { o.x = d2p_lrelu(o.x); o.y = d2p_lrelu(o.y); o.z = d2p_lrelu(o.z); o.w = d2p_lrelu(o.w); }
                *reinterpret_cast<f32x4*>(y + (long)pix * COUT + b * 16 + 4 * q) = o;
                if (STATS) {
                    fs[b] += o;
                    fq[b] += o * o;
                }
            }
        }

        __builtin_amdgcn_sched_barrier(0);
    };
    f32x4 r0[NCH], r1[NCH];
    float ok0[NCH], ok1[NCH];
    int tile = BN ? tlo + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : wave;      // (BN: scalar index arithmetic)
    // the tile of the tensor, advanced without divisions: (sequence tb, tile tr inside it) of this workgroup's index
    int tb = BN ? tile / bn.tps : 0, tr = BN ? tile - tb * bn.tps : 0;
    auto advance = [&]() {           // -> the tensor's tile for `tile` after tile += NW (< 0 past the end)
        tile += NW;
        if (BN) {
            tr += NW;
            if (tr >= bn.tps) { tr -= bn.tps; ++tb; }
        }
        return tile < thi ? (BN ? (tb * bn.G + sg) * bn.tps + tr : tile) : -1;
    };
    int gt = tile < thi ? gtile(tile) : -1;
    fwd_gather<CIN, MASK, T, AFFINE>(g, x, gt >= 0 ? gt * 16 + p : g.P, q, r0, ok0, padoff);
    while (gt >= 0) {
        int gtn = advance();
        step(gt, gtn, r0, ok0, r1, ok1);
        gt = gtn;
        if (gt < 0) break;
        gtn = advance();
        step(gt, gtn, r1, ok1, r0, ok0);
        gt = gtn;
    }
    if (STATS) {
        flush();
        // lanes (p, q) hold channels 16b + 4q + r of pixel lane p: the 16 pixel lanes by xor-shuffles, the 4 waves
        // through LDS in wave order
        __shared__ double wsum[4 * COUT * 2];
        const int wid = threadIdx.x >> 6;
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
                    wsum[(wid * COUT + b * 16 + 4 * q + r) * 2] = u;
                    wsum[(wid * COUT + b * 16 + 4 * q + r) * 2 + 1] = v;
                }
            }
        __syncthreads();
        const int tid = threadIdx.x;
        if (tid < 2 * COUT) {
            const int c = tid >> 1, k = tid & 1;
            bn.stats[((long)blockIdx.x * COUT + c) * 2 + k] =
                ((wsum[(0 * COUT + c) * 2 + k] + wsum[(1 * COUT + c) * 2 + k]) + wsum[(2 * COUT + c) * 2 + k]) +
                wsum[(3 * COUT + c) * 2 + k];
        }
    }
    // `keep` is always 0.  The prefetched sets stay live on the exit path, so the compiler cannot
    // sink a step's prefetch below the loop-exit test into the next step (which would serialise
    // load latency and MFMA work again).
    if (keep) {
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) t += r0[ch] + r1[ch];
        *reinterpret_cast<f32x4*>(y) = t;
    }
}

// ==========================================================================================
// dgrad: one parity class (EY, EX) per workgroup
// ==========================================================================================
struct DgradCls {
    int iy0, ix0, Hc, Wc, P, blk0, nblk, ntiles;
    FastDiv d_hw, d_w;
};
struct DgradArgs {
    DgradCls c[4];   // indexed by e = 2*EY + EX, EY = (iy + pt) & 1
};

template <int E>   // E = 0: taps {0, 2};  E = 1: tap {1}
struct ClsTaps {
    static constexpr int n = E ? 1 : 2;
    static constexpr int k(int i) { return E ? 1 : 2 * i; }
};

template <int CIN, int COUT, int MASK, int EY, int EX>
__device__ __forceinline__ void dgrad_class(const DirectGeom& g, const DgradCls& c,
                                            const float* __restrict__ dy, const float* __restrict__ w,
                                            float* __restrict__ dx, int keep) {
    constexpr int CC = COUT / 16, NBI = CIN / 16;
    constexpr int NTY = ClsTaps<EY>::n, NTX = ClsTaps<EX>::n, NT = NTY * NTX;
    const int lane = threadIdx.x & 63, p = lane & 15, q = lane >> 4;
    const int wave = (blockIdx.x - c.blk0) * 4 + (threadIdx.x >> 6), NW = c.nblk * 4;

    // W^T -> registers (A operand): wr[t][cc][j][b] = W[tap t][ci = 16 b + p][co = 16 cc + 4q + j]
    float wr[NT][CC][4][NBI];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tap = ClsTaps<EY>::k(t / NTX) * 3 + ClsTaps<EX>::k(t % NTX);
        const bool used = (MASK >> tap) & 1;
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < NBI; ++b)
                    wr[t][cc][j][b] = used ? w[(tap * CIN + b * 16 + p) * COUT + cc * 16 + 4 * q + j] : 0.f;
    }

    auto gather = [&](int j, f32x4 (&v)[NT][CC], unsigned& okmask, int& xoff) {
        const bool valid = j < c.P;
        const int jc = valid ? j : 0;
        const int n = fdiv(jc, c.d_hw);
        const int rem = jc - n * c.Hc * c.Wc;
        const int a = fdiv(rem, c.d_w), bcol = rem - a * c.Wc;
        const int iy = c.iy0 + 2 * a, ix = c.ix0 + 2 * bcol;
        xoff = valid ? ((n * g.H + iy) * g.W + ix) * CIN : -1;
        okmask = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ky = ClsTaps<EY>::k(t / NTX), kx = ClsTaps<EX>::k(t % NTX);
            const int oy = (iy + g.pt - ky) >> 1, ox = (ix + g.pl - kx) >> 1;
            const bool ok = valid & (oy >= 0) & (oy < g.Ho) & (ox >= 0) & (ox < g.Wo);
            int off = ((n * g.Ho + clampi(oy, g.Ho - 1)) * g.Wo + clampi(ox, g.Wo - 1)) * COUT + 4 * q;
            D2P_OPAQUE(off);
            if ((MASK >> (ky * 3 + kx)) & 1) {
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) v[t][cc] = ldg4(dy + off + cc * 16);
                okmask |= (unsigned)ok << t;
            }
        }
    };

    auto step = [&](int tile, const f32x4 (&cur)[NT][CC], unsigned okc, int xoc, f32x4 (&nxt)[NT][CC],
                    unsigned& okn, int& xon) {
        const int nt = tile + NW;
        gather(nt < c.ntiles ? nt * 16 + p : c.P, nxt, okn, xon);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NBI];
#pragma unroll
        for (int b = 0; b < NBI; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tap = ClsTaps<EY>::k(t / NTX) * 3 + ClsTaps<EX>::k(t % NTX);
            if (!((MASK >> tap) & 1)) continue;
            const bool ok = (okc >> t) & 1u;
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                f32x4 bb = cur[t][cc];
                bb.x = ok ? bb.x : 0.f; bb.y = ok ? bb.y : 0.f; bb.z = ok ? bb.z : 0.f; bb.w = ok ? bb.w : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int b = 0; b < NBI; ++b) acc[b] = D2P_MFMA16(wr[t][cc][j][b], bb[j], acc[b]);
            }
        }
        if (xoc >= 0) {
#pragma unroll
            for (int b = 0; b < NBI; ++b) *reinterpret_cast<f32x4*>(dx + xoc + b * 16 + 4 * q) = acc[b];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x4 r0[NT][CC], r1[NT][CC];
    unsigned ok0, ok1;
    int xo0, xo1;
    int tile = wave;
    gather(tile < c.ntiles ? tile * 16 + p : c.P, r0, ok0, xo0);
    while (tile < c.ntiles) {
        step(tile, r0, ok0, xo0, r1, ok1, xo1);
        tile += NW;
        if (tile >= c.ntiles) break;
        step(tile, r1, ok1, xo1, r0, ok0, xo0);
        tile += NW;
    }
    if (keep) {   // always 0: see conv_direct_fwd_kernel
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) t += r0[i][cc] + r1[i][cc];
        *reinterpret_cast<f32x4*>(dx) = t;
    }
}

template <int CIN, int COUT, int MASK>
__global__ void __launch_bounds__(256)
conv_direct_dgrad_kernel(DirectGeom g, DgradArgs a, const float* __restrict__ dy,
                         const float* __restrict__ w, float* __restrict__ dx, int keep) {
    const int b = blockIdx.x;
    if (b < a.c[1].blk0) dgrad_class<CIN, COUT, MASK, 0, 0>(g, a.c[0], dy, w, dx, keep);
    else if (b < a.c[2].blk0) dgrad_class<CIN, COUT, MASK, 0, 1>(g, a.c[1], dy, w, dx, keep);
    else if (b < a.c[3].blk0) dgrad_class<CIN, COUT, MASK, 1, 0>(g, a.c[2], dy, w, dx, keep);
    else dgrad_class<CIN, COUT, MASK, 1, 1>(g, a.c[3], dy, w, dx, keep);
}

// ==========================================================================================
// wgrad: reduction over pixels, 4 pixels per MFMA, the whole dW tile set in accumulators
// ==========================================================================================
template <int CIN, int MASK>
struct WgShape {
    static constexpr int CB = CIN >= 16 ? CIN / 16 : 1;
    static constexpr int AB = CIN >= 16 ? popcount9(MASK) * CB : 3;   // 16-row blocks of dW
};

template <int CIN, int COUT, int MASK, typename T>
__global__ void __launch_bounds__(256)
conv_direct_wgrad_kernel(DirectGeom g, const T* __restrict__ x, const float* __restrict__ dy,
                         float* __restrict__ slabs, int ngroups, int keep) {
    constexpr int CB = WgShape<CIN, MASK>::CB, AB = WgShape<CIN, MASK>::AB, NBO = COUT / 16;
    constexpr int KK = 9 * CIN;
    __shared__ float red[AB * NBO * 4][64];
    const int lane = threadIdx.x & 63, c = lane & 15, kq = lane >> 4, wid = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wid, NW = gridDim.x * 4;

    f32x4 acc[AB][NBO];
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < NBO; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane tap of the CIN == 4 layout (row i of block a <-> tap 4a + (i >> 2), channel i & 3)
    auto load_group = [&](int grp, float (&av)[4][AB], float (&bv)[4][NBO], unsigned (&okm)[4]) {
        int pix = grp * 16 + 4 * kq;     // this lane's 4 consecutive pixels
        const int pc = pix < g.P ? pix : 0;
        int n = fdiv(pc, g.d_howo);
        const int rem = pc - n * g.Ho * g.Wo;
        int oy = fdiv(rem, g.d_wo), ox = rem - oy * g.Wo;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const bool valid = (pix + m) < g.P && grp < ngroups;
            const int iy0 = 2 * oy - g.pt, ix0 = 2 * ox - g.pl;
            okm[m] = 0;
            int boff = min(pix + m, g.P - 1) * COUT + c;
            D2P_OPAQUE(boff);
#pragma unroll
            for (int b = 0; b < NBO; ++b) bv[m][b] = dy[boff + b * 16];
            okm[m] |= (unsigned)valid << 31;
#pragma unroll
            for (int a = 0; a < AB; ++a) {
                int ky, kx, coff;
                bool tap_ok = true;
                if (CIN >= 16) {
                    const int tap = nth_tap(MASK, a / CB);
                    ky = tap / 3; kx = tap % 3; coff = (a % CB) * 16 + c;
                } else {
                    const int tap = 4 * a + (c >> 2);
                    tap_ok = tap < 9;
                    ky = (tap * 11) >> 5; kx = tap - 3 * ky; coff = c & 3;
                }
                const int iy = iy0 + ky, ix = ix0 + kx;
                const bool ok = valid & tap_ok & (iy >= 0) & (iy < g.H) & (ix >= 0) & (ix < g.W);
                int aoff = ((min(n, g.N - 1) * g.H + clampi(iy, g.H - 1)) * g.W + clampi(ix, g.W - 1)) * CIN + coff;
                D2P_OPAQUE(aoff);
                av[m][a] = ldg1(x + aoff);
                okm[m] |= (unsigned)ok << a;
            }
            // advance to the next pixel in (n, oy, ox) order
            ++ox;
            if (ox == g.Wo) { ox = 0; ++oy; if (oy == g.Ho) { oy = 0; ++n; } }
        }
    };

    auto step = [&](int grp, const float (&ac)[4][AB], const float (&bc)[4][NBO], const unsigned (&okc)[4],
                    float (&an)[4][AB], float (&bn)[4][NBO], unsigned (&okn)[4]) {
        load_group(grp + NW, an, bn, okn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const bool valid = okc[m] >> 31;
            float bvv[NBO];
#pragma unroll
            for (int b = 0; b < NBO; ++b) bvv[b] = valid ? bc[m][b] : 0.f;
#pragma unroll
            for (int a = 0; a < AB; ++a) {
                const float aa = ((okc[m] >> a) & 1u) ? ac[m][a] : 0.f;
#pragma unroll
                for (int b = 0; b < NBO; ++b) acc[a][b] = D2P_MFMA16(aa, bvv[b], acc[a][b]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    float a0[4][AB], b0[4][NBO], a1[4][AB], b1[4][NBO];
    unsigned ok0[4], ok1[4];
    int grp = wave;
    load_group(grp, a0, b0, ok0);
    while (grp < ngroups) {
        step(grp, a0, b0, ok0, a1, b1, ok1);
        grp += NW;
        if (grp >= ngroups) break;
        step(grp, a1, b1, ok1, a0, b0, ok0);
        grp += NW;
    }
    if (keep) {   // always 0: keeps both prefetch sets live on the exit path (see fwd kernel)
        float t = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int a = 0; a < AB; ++a) t += a0[m][a] + a1[m][a];
#pragma unroll
            for (int b = 0; b < NBO; ++b) t += b0[m][b] + b1[m][b];
        }
        acc[0][0][0] += t;
    }

    // workgroup reduction in fixed order (wave 1, then 2, then 3 onto wave 0) through one LDS
    // buffer, then one slab per workgroup
    for (int src = 1; src < 4; ++src) {
        if (wid == src) {
#pragma unroll
            for (int a = 0; a < AB; ++a)
#pragma unroll
                for (int b = 0; b < NBO; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(a * NBO + b) * 4 + r][lane] = acc[a][b][r];
        }
        __syncthreads();
        if (wid == 0) {
#pragma unroll
            for (int a = 0; a < AB; ++a)
#pragma unroll
                for (int b = 0; b < NBO; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][b][r] += red[(a * NBO + b) * 4 + r][lane];
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
                    const float s = acc[a][b][r];
                    int row;   // flattened (tap, ci)
                    if (CIN >= 16) row = nth_tap(MASK, a / CB) * CIN + (a % CB) * 16 + 4 * kq + r;
                    else row = 16 * a + 4 * kq + r;
                    if (row < KK) slab[row * COUT + b * 16 + c] = s;
                }
        // taps that never touch the image contribute exact zeros
        if (CIN >= 16) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                if (!((MASK >> tap) & 1))
                    for (int i = lane; i < CIN * COUT; i += 64) slab[tap * CIN * COUT + i] = 0.f;
        }
    }
}

// ==========================================================================================
// host dispatch
// ==========================================================================================
static int direct_waves(int units, int per_wave) {
    // waves wanted for `units` tiles: a few tiles per wave so the filter load amortises, capped
    // at 4 waves per SIMD over the 256 CUs
    long w = ((long)units + per_wave - 1) / per_wave;
    if (w > 4096) w = 4096;
    if (w < 1) w = 1;
    return (int)w;
}

static int g_direct_fwd_tpw = 2, g_direct_dgrad_tpw = 2, g_direct_wgrad_wgs = 256;
extern "C" int d2p_conv_direct_tune(int fwd_tpw, int dgrad_tpw, int wgrad_wgs) {
    if (fwd_tpw > 0) g_direct_fwd_tpw = fwd_tpw;
    if (dgrad_tpw > 0) g_direct_dgrad_tpw = dgrad_tpw;
    if (dgrad_tpw > 0) d2p_conv_rows_dgrad_tune(dgrad_tpw * 256);
    if (wgrad_wgs > 0) g_direct_wgrad_wgs = wgrad_wgs;
    if (wgrad_wgs != 0) d2p_conv_frames_wgrad_cap(wgrad_wgs > 0 ? wgrad_wgs : 0);
    if (wgrad_wgs != 0) d2p_conv_rows_tune(wgrad_wgs);
    if (fwd_tpw != 0) d2p_conv_frames_tune(fwd_tpw > 0 ? fwd_tpw : 0);
    if (fwd_tpw > 0) d2p_conv_rows_fwd_tune(fwd_tpw * 256);
    return D2P_OK;
}

template <int CIN, int COUT, int MASK, typename T>
static int launch_fwd(const ConvGeom& g, const T* x, const float* w, const float* bias, int act, float* y,
                      hipStream_t st) {
    DirectGeom d = make_direct(g);
    const int ntiles = ceil_div(d.P, 16);
    const int blocks = ceil_div(direct_waves(ntiles, g_direct_fwd_tpw), 4);
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * d.P * 9 * g.Cin * COUT);
    hipLaunchKernelGGL((conv_direct_fwd_kernel<CIN, COUT, MASK, T>), dim3(blocks), dim3(256), 0, st, d, x, w,
                       bias, act, y, ntiles, 0, DirectBn{});
    D2P_LAUNCH_CHECK("conv_direct_fwd");
    return 1;
}

// the batch-norm-folding forms (ConvBnFold): tiles dealt out by demonstration index; needs whole tiles per sequence
static bool direct_bn_ok(const ConvGeom& g, const ConvBnFold& bn) {
    return bn.G >= 1 && bn.seq >= 1 && bn.S >= 1 && g.N % (bn.G * bn.seq) == 0 && (bn.seq * g.Ho * g.Wo) % 16 == 0 &&
           bn.seq * g.Ho * g.Wo / 16 >= 4 && (size_t)g.N * g.H * g.W * g.Cin + (size_t)bn.G * g.Cin < (1ull << 32) &&
           (!bn.in_scale || ((((uintptr_t)bn.in_scale | (uintptr_t)bn.in_shift) & 15) == 0 && bn.in_shift));
}
template <int CIN, int COUT, int MASK>
static int launch_fwd_bn(const ConvGeom& g, const float* x, const float* w, const float* bias, int act, float* y,
                         hipStream_t st, const ConvBnFold& bn) {
    if (!direct_bn_ok(g, bn) || !bn.stats) return 0;
    DirectGeom d = make_direct(g);
    DirectBn b;
    b.G = bn.G; b.S = bn.S;
    b.tps = bn.seq * g.Ho * g.Wo / 16;
    b.per_idx = g.N / (bn.G * bn.seq) * b.tps;
    b.per_slice = ceil_div(b.per_idx, bn.S);
    b.in_scale = bn.in_scale; b.in_nshift = nullptr; b.stats = bn.stats;
    b.pad0 = (unsigned)((size_t)g.N * g.H * g.W * g.Cin);
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * d.P * 9 * g.Cin * COUT);
    if (bn.in_scale)
        hipLaunchKernelGGL((conv_direct_fwd_kernel<CIN, COUT, MASK, float, true, true>), dim3(bn.G * bn.S), dim3(256), 0, st, d,
                           x, w, bias, act, y, 0, 0, b);
    else
        hipLaunchKernelGGL((conv_direct_fwd_kernel<CIN, COUT, MASK, float, true, false>), dim3(bn.G * bn.S), dim3(256), 0, st, d,
                           x, w, bias, act, y, 0, 0, b);
    D2P_LAUNCH_CHECK("conv_direct_fwd_bn");
    return 1;
}

int d2p_conv_direct_fwd(const ConvGeom& g, const void* x, int x_is_u8, const float* w, const float* bias,
                        int act, float* y, hipStream_t st, const ConvBnFold* bn) {
    if (bn) {
        // the batch-norm-folding forms: first layer (frames in, statistics out) on the row-strip kernel, the 16 -> 32
        // layer (affine in, statistics out) on the gather kernel; anything else: 0 = not taken (the caller runs the
        // separate launches)
        int rc = d2p_conv_rows_fwd(g, x, x_is_u8, w, bias, act, y, st, bn);
        if (rc != 0) return rc;
        rc = d2p_conv_wide_fwd(g, x, x_is_u8, w, bias, act, y, st, bn);        // (the 48-channel layers, round 6)
        if (rc != 0) return rc;
        rc = d2p_conv_wide_fwd2(g, x, x_is_u8, w, bias, act, y, st, bn);       // (the large 16 -> 32 layer in block form)
        if (rc != 0) return rc;
        if (x_is_u8 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || (bias && ((uintptr_t)bias & 15))) return 0;
        if (g.Cin == 16 && g.Cout == 32 && g.H * g.W >= 400) return launch_fwd_bn<16, 32, 0x1FF>(g, (const float*)x, w, bias, act, y, st, *bn);
        return 0;
    }
    if (!g_direct_fwd) return 0;
    if (g_direct_fwd >= 2) {
        int rc = d2p_conv_frames_fwd(g, x, x_is_u8, w, bias, act, y, st);
        if (rc != 0) return rc;
        rc = d2p_conv_rows_fwd(g, x, x_is_u8, w, bias, act, y, st);
        if (rc != 0) return rc;
        rc = d2p_conv_wide_fwd(g, x, x_is_u8, w, bias, act, y, st);
        if (rc != 0) return rc;
        rc = d2p_conv_wide_fwd2(g, x, x_is_u8, w, bias, act, y, st);
        if (rc != 0) return rc;
    }
    if (((uintptr_t)x & (x_is_u8 ? 3 : 15)) || ((uintptr_t)y & 15) || (bias && ((uintptr_t)bias & 15))) return 0;
    const int mask = tap_mask_for(g);
    const int key = g.Cin * 100 + g.Cout;
#define D2P_FWD_CASE(CI, CO, MK)                                                                   \
    do {                                                                                           \
        if (x_is_u8) return launch_fwd<CI, CO, MK, uint8_t>(g, (const uint8_t*)x, w, bias, act, y, st); \
        return launch_fwd<CI, CO, MK, float>(g, (const float*)x, w, bias, act, y, st);             \
    } while (0)
    if (key == 416) D2P_FWD_CASE(4, 16, 0x1FF);
    if (key == 1616) D2P_FWD_CASE(16, 16, 0x1FF);
    if (key == 1632) D2P_FWD_CASE(16, 32, 0x1FF);
    if (key == 3248 && mask == 0x1B) D2P_FWD_CASE(32, 48, 0x1B);
#undef D2P_FWD_CASE
    return 0;
}

template <int CIN, int COUT, int MASK>
static int launch_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st) {
    DirectGeom d = make_direct(g);
    DgradArgs a;
    int blk = 0;
    for (int e = 0; e < 4; ++e) {
        const int ey = e >> 1, ex = e & 1;
        DgradCls& c = a.c[e];
        c.iy0 = ey ^ (g.pt & 1);            // (iy + pt) & 1 == ey
        c.ix0 = ex ^ (g.pl & 1);
        c.Hc = c.iy0 < g.H ? (g.H - c.iy0 + 1) / 2 : 0;
        c.Wc = c.ix0 < g.W ? (g.W - c.ix0 + 1) / 2 : 0;
        c.P = g.N * c.Hc * c.Wc;
        c.ntiles = ceil_div(c.P, 16);
        c.nblk = c.P > 0 ? ceil_div(direct_waves(c.ntiles, g_direct_dgrad_tpw), 4) : 0;
        c.blk0 = blk;
        blk += c.nblk;
        c.d_hw = make_fastdiv(c.Hc * c.Wc > 0 ? c.Hc * c.Wc : 1);
        c.d_w = make_fastdiv(c.Wc > 0 ? c.Wc : 1);
    }
    if (blk == 0) return 1;
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * d.P * 9 * g.Cin * COUT);
    hipLaunchKernelGGL((conv_direct_dgrad_kernel<CIN, COUT, MASK>), dim3(blk), dim3(256), 0, st, d, a, dy, w, dx, 0);
    D2P_LAUNCH_CHECK("conv_direct_dgrad");
    return 1;
}

int d2p_conv_direct_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st,
                          const ConvDgradBn* bn) {
    if (bn) {       // (statistics out: the block-form kernel of conv_wide.hip, else the row-strip kernel of the 16 -> 32 layer)
        const int rc = d2p_conv_wide_dgrad(g, dy, w, dx, st, bn);
        return rc != 0 ? rc : d2p_conv_rows_dgrad(g, dy, w, dx, st, bn);
    }
    if (!g_direct_dgrad) return 0;
    if (g_direct_dgrad >= 2) {
        int rc = d2p_conv_wide_dgrad(g, dy, w, dx, st);            // (the 48-channel layers and the large 16 -> 32 layer, round 6)
        if (rc != 0) return rc;
        rc = d2p_conv_rows_dgrad(g, dy, w, dx, st);
        if (rc != 0) return rc;
    }
    if (((uintptr_t)dy & 15) || ((uintptr_t)dx & 15)) return 0;
    const int mask = tap_mask_for(g);
    const int key = g.Cin * 100 + g.Cout;
    if (key == 1632) return launch_dgrad<16, 32, 0x1FF>(g, dy, w, dx, st);
    if (key == 1616) return launch_dgrad<16, 16, 0x1FF>(g, dy, w, dx, st);
    if (key == 3248 && mask == 0x1B) return launch_dgrad<32, 48, 0x1B>(g, dy, w, dx, st);
    return 0;
}

static bool wgrad_supported(const ConvGeom& g) {
    const int key = g.Cin * 100 + g.Cout;
    return key == 416 || key == 1616 || key == 1632 || (key == 3248 && tap_mask_for(g) == 0x1B);
}

static int wgrad_blocks(const ConvGeom& g) {
    const int ngroups = ceil_div(g.N * g.Ho * g.Wo, 16);
    int blocks = ceil_div(ngroups, 4);
    if (blocks > g_direct_wgrad_wgs) blocks = g_direct_wgrad_wgs;
    return blocks < 1 ? 1 : blocks;
}

size_t d2p_conv_direct_wgrad_ws(const ConvGeom& g) {
    size_t fr = d2p_conv_frames_wgrad_ws(g);
    const size_t rw = d2p_conv_rows_wgrad_ws(g);
    if (rw > fr) fr = rw;
    const size_t ww = d2p_conv_wide_wgrad_ws(g);
    if (ww > fr) fr = ww;
    if (!wgrad_supported(g)) return fr;
    const size_t di = (size_t)wgrad_blocks(g) * 9 * g.Cin * g.Cout * sizeof(float);
    return fr > di ? fr : di;
}

template <int CIN, int COUT, int MASK, typename T>
static int launch_wgrad(const ConvGeom& g, const T* x, const float* dy, float* dw, void* ws, size_t ws_bytes,
                        hipStream_t st) {
    DirectGeom d = make_direct(g);
    const int ngroups = ceil_div(d.P, 16);
    const int blocks = wgrad_blocks(g);
    const int KK = 9 * CIN;
    D2P_REQUIRE(ws && ws_bytes >= (size_t)blocks * KK * COUT * sizeof(float), D2P_EWS,
                "conv wgrad: workspace too small (%zu bytes)", ws_bytes);
    float* slabs = (float*)ws;
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * d.P * KK * COUT);
    hipLaunchKernelGGL((conv_direct_wgrad_kernel<CIN, COUT, MASK, T>), dim3(blocks), dim3(256), 0, st, d, x, dy,
                       slabs, ngroups, 0);
    D2P_LAUNCH_CHECK("conv_direct_wgrad");
    EpiDense ep{dw, COUT, nullptr, 0, 0};
    const long total = (long)KK * COUT;
    int rb = (int)((total * 16 + 255) / 256);
    hipLaunchKernelGGL((gemm_splitk_reduce_kernel<EpiDense>), dim3(rb), dim3(256), 0, st, ep, slabs, KK, COUT,
                       blocks);
    D2P_LAUNCH_CHECK("conv_direct_wgrad_combine");
    return 1;
}

int d2p_conv_direct_wgrad(const ConvGeom& g, const void* x, int x_is_u8, const float* dy, float* dw, void* ws,
                          size_t ws_bytes, hipStream_t st, const ConvBnFold* bn) {
    if (bn) {       // (input affine: the row-strip kernel of the 16 -> 32 layer, the wide kernel of the 48-channel layers)
        const int rc = d2p_conv_rows_wgrad(g, x, x_is_u8, dy, dw, ws, ws_bytes, st, bn);
        return rc != 0 ? rc : d2p_conv_wide_wgrad(g, x, x_is_u8, dy, dw, ws, ws_bytes, st, bn);
    }
    if (!g_direct_wgrad) return 0;
    if (g_direct_wgrad >= 2) {
        int rc = d2p_conv_frames_wgrad(g, x, x_is_u8, dy, dw, ws, ws_bytes, st);
        if (rc != 0) return rc;
        rc = d2p_conv_rows_wgrad(g, x, x_is_u8, dy, dw, ws, ws_bytes, st);
        if (rc != 0) return rc;
        rc = d2p_conv_wide_wgrad(g, x, x_is_u8, dy, dw, ws, ws_bytes, st);          // (the 48-channel layers, round 6)
        if (rc != 0) return rc;
    }
    if (!wgrad_supported(g)) return 0;
    if (((uintptr_t)x & (x_is_u8 ? 0 : 3)) || ((uintptr_t)dy & 3)) return 0;
    if (g.N == 0) return 0;
    const int key = g.Cin * 100 + g.Cout;
#define D2P_WG_CASE(CI, CO, MK)                                                                        \
    do {                                                                                               \
        if (x_is_u8) return launch_wgrad<CI, CO, MK, uint8_t>(g, (const uint8_t*)x, dy, dw, ws, ws_bytes, st); \
        return launch_wgrad<CI, CO, MK, float>(g, (const float*)x, dy, dw, ws, ws_bytes, st);          \
    } while (0)
    if (key == 416) D2P_WG_CASE(4, 16, 0x1FF);
    if (key == 1616) D2P_WG_CASE(16, 16, 0x1FF);
    if (key == 1632) D2P_WG_CASE(16, 32, 0x1FF);
    if (key == 3248) D2P_WG_CASE(32, 48, 0x1B);
#undef D2P_WG_CASE
    return 0;
}
