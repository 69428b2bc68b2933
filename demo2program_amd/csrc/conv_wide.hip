// K1 (wide back end, round 6): the 48-channel layers of the ViZDoom State_Encoder -- conv3 32 -> 48 on 20x20,
// conv4 / conv5 48 -> 48 on 10x10 / 5x5 frames (models/ops.py:27-33 called from models/model_full.py:216-231),
// 3x3 stride 2 TF-"SAME".  640 k x 288 x 48, 160 k x 432 x 48 and 57.6 k x 432 x 48 as GEMMs: on the generic
// implicit-GEMM kernel a quarter of every 64-wide tile multiplies zeros (N = 48) and its im2col loaders decompose an
// index per 16-byte load -- 0.34-0.38 of the fp32 MFMA peak at conv3 (profiles/r05h_kernel_stats_vizdoom.md).  The
// register-resident-filter kernels of conv_direct.hip stop at 16 -> 32: a 32 -> 48 filter is 216 registers per lane.
//
// Here the FILTER LIVES IN LDS, in MFMA fragment order, shared by the workgroup's eight waves:
//   * v_mfma_f32_16x16x4_f32 with the filter as the A operand and 16 output pixels as B, N = 48 as three 16-channel
//     blocks: D = [channel][pixel], a lane ends with 4 consecutive channels of one pixel = one 16-byte NHWC store;
//   * a wave owns whole 16-pixel tiles (all K, all 48 channels): no partial sums, no barrier in the loop; its B operand
//     is gathered HBM/L2 -> VGPR with one 16-byte load per (pixel, tap, 4 channels) as in conv_direct.hip, the A
//     fragments are lane-linear ds_read_b128 (one per 4 MFMAs, conflict-free);
//   * two to four waves per SIMD: one wave's gather latency and epilogue sit under the others' MFMA chains (a wave
//     keeps ONE tile's operands in registers: 72 / 108 of them);
//   * the batch-norm folding of round 5 (ConvBnFold, conv_geom.h) carries over: tiles are dealt out by demonstration
//     index, STATS leaves the fp64 (sum, sum of squares) partials of the launch's own outputs, AFFINE reads the input
//     through the previous layer's batch-norm apply at no cost in the loop (scale folded into the LDS filter copy,
//     shift / scale added to what is loaded, out-of-image taps load the index's pad pixel).  Sequences whose pixel count
//     is not a multiple of 16 (10x10 -> 5x5: 500 pixels per sequence of 20 frames) end in a ragged, masked tile.
#include "conv_geom.h"
#include "gemm_core.h"
#include "prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define D2P_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define D2P_OPAQUE_U(v) asm volatile("" : "+v"(v))

namespace {

constexpr int WCO = 48;          // output channels of every layer here
constexpr int WNB = 3;           // 16-channel blocks
// waves per workgroup: three per SIMD at 32 input channels (153 registers), two at 48 (211)
constexpr int wide_waves(int cin) { return cin == 32 ? 12 : 8; }

struct WDiv {
    uint32_t m;
    int s;
};
WDiv make_wdiv(int d) {
    WDiv f;
    f.s = 0;
    while ((1L << f.s) < d) ++f.s;
    const uint64_t num = 1ULL << (31 + f.s);
    f.m = (uint32_t)((num + d - 1) / d);
    return f;
}
__device__ __forceinline__ int wdiv(int n, WDiv f) { return (int)((uint32_t)(((uint64_t)(uint32_t)n * f.m) >> 31) >> f.s); }

struct WideGeom {
    int N, H, W, Ho, Wo, pt, pl, P;
    WDiv d_howo, d_wo;
};
WideGeom make_wide(const ConvGeom& g) {
    WideGeom d;
    d.N = g.N; d.H = g.H; d.W = g.W; d.Ho = g.Ho; d.Wo = g.Wo; d.pt = g.pt; d.pl = g.pl;
    d.P = g.N * g.Ho * g.Wo;
    d.d_howo = make_wdiv(g.Ho * g.Wo);
    d.d_wo = make_wdiv(g.Wo);
    return d;
}

// how the launch's 16-pixel tiles are dealt out: workgroup (g, s) = blockIdx.x takes slice s of the tiles of
// demonstration index g; tile j of an index = tile j % tps of its sequence j / tps (a sequence = the seq frames of one
// (program, index) pair, frames ordered (program, index, t)); a plain launch is G = 1 with the whole batch as one sequence
struct WideDeal {
    int G, S;
    int seqpix;              // output pixels of one sequence
    int tps;                 // tiles of one sequence (the last one ragged when seqpix % 16 != 0)
    int per_idx, per_slice;  // tiles of one index / of one slice
    const float* in_scale;   // [G, CIN] or null (AFFINE)
    double* stats;           // [G][S][48][2] or null (STATS)
    unsigned pad0;           // AFFINE: float offset of index 0's pad pixel in x
};

__device__ __forceinline__ f32x4 wldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ int wclamp(int v, int hi) { return min(max(v, 0), hi); }

template <int CIN, bool STATS, bool AFFINE>
__global__ void __launch_bounds__(wide_waves(CIN) * 64)
conv_wide_fwd_kernel(WideGeom g, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                     int act, float* __restrict__ y, WideDeal dl) {
    constexpr int CB = CIN / 16, NCH = 9 * CB, WWAVES = wide_waves(CIN);
    extern __shared__ __attribute__((aligned(16))) float wide_lds[];
    f32x4* wl = reinterpret_cast<f32x4*>(wide_lds);                 // [NCH][3][64]: A fragments of (chunk, block)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;
    const int sg = (int)blockIdx.x / dl.S, ss = (int)blockIdx.x - sg * dl.S;

    // ---- filter -> LDS in fragment order: wl[(ch*3 + b)*64 + l][j] = W[k = chunk ch, 4q + j][co = 16 b + p]
    //      (AFFINE: times the input scale of k's channel -- the index is fixed for the workgroup)
    for (int i = tid; i < NCH * WNB * 64; i += WWAVES * 64) {
        const int l = i & 63, cb3 = i >> 6;
        const int ch = cb3 / WNB, b = cb3 - ch * WNB;
        const int pp = l & 15, qq = l >> 4;
        const int cin0 = (ch % CB) * 16 + 4 * qq;
        const int kbase = (ch / CB) * CIN + cin0;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = w[(kbase + j) * WCO + b * 16 + pp];
        if (AFFINE) v *= wldg4(dl.in_scale + sg * CIN + cin0);
        wl[i] = v;
    }
    const unsigned padoff = AFFINE ? dl.pad0 + (unsigned)(sg * CIN) : 0u;
    f32x4 ash[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        // the pad pixel holds -shift / scale: what the loop adds is its negative
        if (AFFINE) ash[cb] = -wldg4(x + padoff + cb * 16 + 4 * q);
        else ash[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 bv[WNB];
#pragma unroll
    for (int b = 0; b < WNB; ++b) {
        if (bias) bv[b] = wldg4(bias + b * 16 + 4 * q);
        else bv[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 fs[WNB], fq[WNB];
#pragma unroll
    for (int b = 0; b < WNB; ++b) fs[b] = fq[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int tlo = ss * dl.per_slice, thi = min(tlo + dl.per_slice, dl.per_idx);
    int tile = tlo + __builtin_amdgcn_readfirstlane(wave);
    int tb = tile / dl.tps, tr = tile - tb * dl.tps;           // (sequence, tile inside it) of this index
    const int HoWo = g.Ho * g.Wo;
    for (; tile < thi; tile += WWAVES) {
        const int local = tr * 16 + p;
        const bool valid = local < dl.seqpix;
        const int pix = (tb * dl.G + sg) * dl.seqpix + (valid ? local : 0);
        // ---- gather: v[ch] = 4 consecutive k of chunk ch (tap ch / CB, channels 16 (ch % CB) + 4q ..) of this lane's pixel
        const int n = wdiv(pix, g.d_howo);
        const int rem = pix - n * HoWo;
        const int oy = wdiv(rem, g.d_wo), ox = rem - oy * g.Wo;
        const int iy0 = 2 * oy - g.pt, ix0 = 2 * ox - g.pl;
        int rowoff[3], coloff[3];
        unsigned okbits = 0u;
        bool rok[3], cok[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int iy = iy0 + k, ix = ix0 + k;
            rok[k] = valid & ((unsigned)iy < (unsigned)g.H);
            cok[k] = (unsigned)ix < (unsigned)g.W;
            rowoff[k] = (n * g.H + wclamp(iy, g.H - 1)) * g.W * CIN;
            coloff[k] = wclamp(ix, g.W - 1) * CIN + 4 * q;
        }
        f32x4 v[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int tap = ch / CB, ky = tap / 3, kx = tap % 3;
            unsigned off = (unsigned)(rowoff[ky] + coloff[kx] + (ch % CB) * 16);
            const bool ok = rok[ky] & cok[kx];
            if (AFFINE) off = ok ? off : padoff + (unsigned)((ch % CB) * 16 + 4 * q);     // (a select of the offset, not of the load)
            else if (ch % CB == 0) okbits |= ok ? (1u << tap) : 0u;
            D2P_OPAQUE_U(off);
            v[ch] = wldg4(x + off);
        }
        // ---- 4 * 3 MFMAs per chunk, A fragments from LDS
        f32x4 acc[2][WNB];
#pragma unroll
        for (int b = 0; b < WNB; ++b) acc[0][b] = acc[1][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            f32x4 bb;
            if (AFFINE) bb = v[ch] + ash[ch % CB];                 // (x + shift / scale; a pad pixel gives exactly 0)
            else bb = v[ch] * (((okbits >> (ch / CB)) & 1u) ? 1.f : 0.f);
#pragma unroll
            for (int b = 0; b < WNB; ++b) {
                const f32x4 a4 = wl[(ch * WNB + b) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j & 1][b] = D2P_MFMA16(a4[j], bb[j], acc[j & 1][b]);
            }
        }
        // ---- epilogue: bias, leaky ReLU, store, statistics
        const float vf = valid ? 1.f : 0.f;
#pragma unroll
        for (int b = 0; b < WNB; ++b) {
            f32x4 o = (acc[0][b] + acc[1][b]) + bv[b];
            if (act) { o.x = d2p_lrelu(o.x); o.y = d2p_lrelu(o.y); o.z = d2p_lrelu(o.z); o.w = d2p_lrelu(o.w); }
            if (valid) *reinterpret_cast<f32x4*>(y + (long)pix * WCO + b * 16 + 4 * q) = o;
            if (STATS) {
                o *= vf;
                fs[b] += o;
                fq[b] += o * o;
            }
        }
        tr += WWAVES;
        while (tr >= dl.tps) { tr -= dl.tps; ++tb; }
    }
    if (STATS) {
        // lanes (p, q) hold channels 16b + 4q + r of pixel lane p: the 16 pixel lanes by xor-shuffles (fp64 from here
        // on), the eight waves through LDS in wave order
        __shared__ double wsum[WWAVES * WCO * 2];
#pragma unroll
        for (int b = 0; b < WNB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double u = (double)fs[b][r], s2 = (double)fq[b][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    u += __shfl_xor(u, off, 64);
                    s2 += __shfl_xor(s2, off, 64);
                }
                if (p == 0) {
                    wsum[(wave * WCO + b * 16 + 4 * q + r) * 2] = u;
                    wsum[(wave * WCO + b * 16 + 4 * q + r) * 2 + 1] = s2;
                }
            }
        __syncthreads();
        if (tid < 2 * WCO) {
            const int c = tid >> 1, k = tid & 1;
            double t = 0.0;
#pragma unroll
            for (int wv = 0; wv < WWAVES; ++wv) t += wsum[(wv * WCO + c) * 2 + k];
            dl.stats[((long)blockIdx.x * WCO + c) * 2 + k] = t;
        }
    }
}

bool wide_geom_ok(const ConvGeom& g) {
    if (g.Cout != WCO || (g.Cin != 32 && g.Cin != 48)) return false;
    if (g.H < 3 || g.W < 3) return false;                  // every tap touches the image somewhere
    return (size_t)g.N * g.H * g.W * g.Cin + (size_t)4096 * g.Cin < (1ull << 32) && (size_t)g.N * g.Ho * g.Wo < (1ull << 31) / 16;
}

// slices per index: ONE round of workgroups over the chip (a workgroup fills its CU: 12 waves of 153 registers / 8 waves of
// 211 and 54 / 81 KB of LDS) -- G * S <= CUs, a tile per wave at least.  (Two workgroups per CU's worth of slices ran as a
// second, partly filled round: 410 workgroups at conv3 = 0.8 of the chip on average; 60 at conv5 left 196 CUs idle.)
int wide_cus() {
    static int n = 0;
    if (n <= 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
            n = v;
        else
            (void)hipGetLastError();
    }
    return n > 0 ? n : 256;
}
int wide_slices(long per_idx, int G, int cin) {
    long S = wide_cus() / G;
    const long cap = per_idx / wide_waves(cin);
    if (S > cap) S = cap;
    return (int)(S < 1 ? 1 : S);
}

template <int CIN>
int launch_wide_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, int act, float* y, hipStream_t st,
                    const ConvBnFold* bn) {
    WideGeom d = make_wide(g);
    WideDeal dl;
    const int G = bn ? bn->G : 1, seq = bn ? bn->seq : g.N;
    dl.G = G;
    dl.seqpix = seq * g.Ho * g.Wo;
    dl.tps = ceil_div(dl.seqpix, 16);
    dl.per_idx = g.N / (G * seq) * dl.tps;
    dl.S = bn && bn->S > 0 ? bn->S : wide_slices(dl.per_idx, G, CIN);
    dl.per_slice = ceil_div(dl.per_idx, dl.S);
    dl.in_scale = bn ? bn->in_scale : nullptr;
    dl.stats = bn ? bn->stats : nullptr;
    dl.pad0 = (unsigned)((size_t)g.N * g.H * g.W * g.Cin);
    constexpr size_t lds = (size_t)9 * (CIN / 16) * WNB * 64 * sizeof(f32x4);
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * d.P * 9 * g.Cin * WCO);
    const dim3 grid(G * dl.S), block(wide_waves(CIN) * 64);
#define D2P_WIDE_LAUNCH(ST, AF)                                                                                     \
    do {                                                                                                            \
        auto kern = conv_wide_fwd_kernel<CIN, ST, AF>;                                                              \
        static bool attr = false;                                                                                   \
        if (!attr) {                                                                                                \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            D2P_REQUIRE(e == hipSuccess, (int)e, "conv wide fwd: %s", hipGetErrorString(e));                        \
            attr = true;                                                                                            \
        }                                                                                                           \
        hipLaunchKernelGGL(kern, grid, block, lds, st, d, x, w, bias, act, y, dl);                                  \
    } while (0)
    if (dl.stats && dl.in_scale) D2P_WIDE_LAUNCH(true, true);
    else if (dl.stats) D2P_WIDE_LAUNCH(true, false);
    else if (dl.in_scale) D2P_WIDE_LAUNCH(false, true);
    else D2P_WIDE_LAUNCH(false, false);
#undef D2P_WIDE_LAUNCH
    D2P_LAUNCH_CHECK("conv_wide_fwd");
    return 1;
}


// ==========================================================================================
// forward, block form (the 16 -> 32 layer of the large frames, 40x40 -> 20x20: 23.6 GFLOP per pass at config 4): a lane's
// unit is a 2x2 BLOCK of output pixels, whose four 3x3 windows are one 5x5 window of the input -- 25 gathers and one
// index decomposition for four output pixels instead of 36 and four, the input affine added once per loaded value, and
// an A fragment read from LDS feeds 16 MFMAs.  (The register-filter gather kernel of conv_direct.hip spends as many
// clocks on index arithmetic, masks and epilogue as on its 72 MFMAs per 16 pixels: 0.51 of the fp32 MFMA peak.)  Same
// dealing by demonstration index, STATS and AFFINE as conv_wide_fwd_kernel; tiles are 16 blocks of one sequence.
// ==========================================================================================
constexpr int WF2_WAVES = 8;
struct WF2Blk {
    int Hb, Wb;               // blocks per frame: ceil(Ho / 2), ceil(Wo / 2)
    WDiv d_hw, d_w;
};

template <int CIN, int COUT, bool STATS, bool AFFINE>
__global__ void __launch_bounds__(WF2_WAVES * 64)
conv_wide_fwd2_kernel(WideGeom g, WF2Blk c, const float* __restrict__ x, const float* __restrict__ w,
                      const float* __restrict__ bias, int act, float* __restrict__ y, WideDeal dl) {
    constexpr int CB = CIN / 16, NB = COUT / 16;
    static_assert(CB == 1, "one 16-channel block of input per tap");
    extern __shared__ __attribute__((aligned(16))) float wide_lds[];
    f32x4* wl = reinterpret_cast<f32x4*>(wide_lds);                 // [9][NB][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;
    const int sg = (int)blockIdx.x / dl.S, ss = (int)blockIdx.x - sg * dl.S;
    for (int i = tid; i < 9 * NB * 64; i += WF2_WAVES * 64) {
        const int l = i & 63, tb_ = i >> 6;
        const int tap = tb_ / NB, b = tb_ - tap * NB;
        const int pp = l & 15, qq = l >> 4;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = w[(tap * CIN + 4 * qq + j) * COUT + b * 16 + pp];
        if (AFFINE) v *= wldg4(dl.in_scale + sg * CIN + 4 * qq);
        wl[i] = v;
    }
    const unsigned padoff = AFFINE ? dl.pad0 + (unsigned)(sg * CIN) : 0u;
    const f32x4 ash = AFFINE ? -wldg4(x + padoff + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bv[NB], fs[NB], fq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        bv[b] = bias ? wldg4(bias + b * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        fs[b] = fq[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int tlo = ss * dl.per_slice, thi = min(tlo + dl.per_slice, dl.per_idx);
    int tile = tlo + __builtin_amdgcn_readfirstlane(wave);
    int tb = tile / dl.tps, tr = tile - tb * dl.tps;
    const int HbWb = c.Hb * c.Wb;
    for (; tile < thi; tile += WF2_WAVES) {
        const int local = tr * 16 + p;
        const bool valid = local < dl.seqpix;
        const int lc = valid ? local : 0;
        const int t = wdiv(lc, c.d_hw);
        const int rem = lc - t * HbWb;
        const int a = wdiv(rem, c.d_w), bcol = rem - a * c.Wb;
        const int n = (tb * dl.G + sg) * (dl.seqpix / HbWb) + t;
        // ---- the 5x5 window: rows 4a - pt + r, columns 4 bcol - pl + r
        const int iy0 = 4 * a - g.pt, ix0 = 4 * bcol - g.pl;
        int rowoff[5], coloff[5];
        bool rok[5], cok[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            rok[r] = valid & ((unsigned)(iy0 + r) < (unsigned)g.H);
            cok[r] = (unsigned)(ix0 + r) < (unsigned)g.W;
            rowoff[r] = (n * g.H + wclamp(iy0 + r, g.H - 1)) * g.W * CIN;
            coloff[r] = wclamp(ix0 + r, g.W - 1) * CIN + 4 * q;
        }
        f32x4 v[5][5];
#pragma unroll
        for (int ry = 0; ry < 5; ++ry)
#pragma unroll
            for (int rx = 0; rx < 5; ++rx) {
                unsigned off = (unsigned)(rowoff[ry] + coloff[rx]);
                const bool ok = rok[ry] & cok[rx];
                if (AFFINE) off = ok ? off : padoff + (unsigned)(4 * q);
                D2P_OPAQUE_U(off);
                v[ry][rx] = wldg4(x + off);
                if (AFFINE) v[ry][rx] += ash;                          // (x + shift / scale; a pad pixel gives exactly 0)
                else v[ry][rx] *= ok ? 1.f : 0.f;
            }
        f32x4 acc[4][NB];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[o][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const f32x4 a4 = wl[(tap * NB + b) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        acc[o][b] = D2P_MFMA16(a4[j], v[2 * (o >> 1) + ky][2 * (o & 1) + kx][j], acc[o][b]);
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int oy = 2 * a + (o >> 1), ox = 2 * bcol + (o & 1);
            const bool in = valid & (oy < g.Ho) & (ox < g.Wo);
            const long yoff = ((long)(n * g.Ho + (in ? oy : 0)) * g.Wo + (in ? ox : 0)) * COUT;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                f32x4 r_ = acc[o][b] + bv[b];
                if (act) { r_.x = d2p_lrelu(r_.x); r_.y = d2p_lrelu(r_.y); r_.z = d2p_lrelu(r_.z); r_.w = d2p_lrelu(r_.w); }
                if (in) *reinterpret_cast<f32x4*>(y + yoff + b * 16 + 4 * q) = r_;
                if (STATS) {
                    r_ *= in ? 1.f : 0.f;
                    fs[b] += r_;
                    fq[b] += r_ * r_;
                }
            }
        }
        tr += WF2_WAVES;
        while (tr >= dl.tps) { tr -= dl.tps; ++tb; }
    }
    if (STATS) {
        __shared__ double wsum[WF2_WAVES * COUT * 2];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double u = (double)fs[b][r], s2 = (double)fq[b][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    u += __shfl_xor(u, off, 64);
                    s2 += __shfl_xor(s2, off, 64);
                }
                if (p == 0) {
                    wsum[(wave * COUT + b * 16 + 4 * q + r) * 2] = u;
                    wsum[(wave * COUT + b * 16 + 4 * q + r) * 2 + 1] = s2;
                }
            }
        __syncthreads();
        if (tid < 2 * COUT) {
            const int ch = tid >> 1, k = tid & 1;
            double tsum = 0.0;
#pragma unroll
            for (int wv = 0; wv < WF2_WAVES; ++wv) tsum += wsum[(wv * COUT + ch) * 2 + k];
            dl.stats[((long)blockIdx.x * COUT + ch) * 2 + k] = tsum;
        }
    }
}

bool wide_fwd2_geom_ok(const ConvGeom& g) {
    return g.Cin == 16 && g.Cout == 32 && g.H >= 3 && g.W >= 3 && g.H * g.W >= 400 &&
           (size_t)g.N * g.H * g.W * g.Cin + (size_t)4096 * g.Cin < (1ull << 32) && (size_t)g.N * g.Ho * g.Wo < (1ull << 31) / 32;
}
int wide_fwd2_slices(const ConvGeom& g, int G, int seq) {
    const int Hb = (g.Ho + 1) / 2, Wb = (g.Wo + 1) / 2;
    const long tps = ((long)seq * Hb * Wb + 15) / 16;
    const long per_idx = (long)g.N / (G * seq) * tps;
    long S = wide_cus() / G;                               // (one workgroup per CU: ~190 registers, two waves per SIMD)
    const long cap = per_idx / WF2_WAVES;
    if (S > cap) S = cap;
    return (int)(S < 1 ? 1 : S);
}

template <int CIN, int COUT>
int launch_wide_fwd2(const ConvGeom& g, const float* x, const float* w, const float* bias, int act, float* y, hipStream_t st,
                     const ConvBnFold* bn) {
    WideGeom d = make_wide(g);
    WF2Blk c{};
    c.Hb = (g.Ho + 1) / 2; c.Wb = (g.Wo + 1) / 2;
    c.d_hw = make_wdiv(c.Hb * c.Wb);
    c.d_w = make_wdiv(c.Wb);
    WideDeal dl{};
    const int G = bn ? bn->G : 1, seq = bn ? bn->seq : g.N;
    dl.G = G;
    dl.seqpix = seq * c.Hb * c.Wb;
    dl.tps = ceil_div(dl.seqpix, 16);
    dl.per_idx = g.N / (G * seq) * dl.tps;
    dl.S = bn && bn->S > 0 ? bn->S : wide_fwd2_slices(g, G, seq);
    dl.per_slice = ceil_div(dl.per_idx, dl.S);
    dl.in_scale = bn ? bn->in_scale : nullptr;
    dl.stats = bn ? bn->stats : nullptr;
    dl.pad0 = (unsigned)((size_t)g.N * g.H * g.W * g.Cin);
    constexpr size_t lds = (size_t)9 * (COUT / 16) * 64 * sizeof(f32x4);
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * d.P * 9 * g.Cin * COUT);
    const dim3 grid(G * dl.S), block(WF2_WAVES * 64);
    if (dl.stats && dl.in_scale)
        hipLaunchKernelGGL((conv_wide_fwd2_kernel<CIN, COUT, true, true>), grid, block, lds, st, d, c, x, w, bias, act, y, dl);
    else if (dl.stats)
        hipLaunchKernelGGL((conv_wide_fwd2_kernel<CIN, COUT, true, false>), grid, block, lds, st, d, c, x, w, bias, act, y, dl);
    else if (dl.in_scale)
        hipLaunchKernelGGL((conv_wide_fwd2_kernel<CIN, COUT, false, true>), grid, block, lds, st, d, c, x, w, bias, act, y, dl);
    else
        hipLaunchKernelGGL((conv_wide_fwd2_kernel<CIN, COUT, false, false>), grid, block, lds, st, d, c, x, w, bias, act, y, dl);
    D2P_LAUNCH_CHECK("conv_wide_fwd2");
    return 1;
}

// ==========================================================================================
// input gradient: dX[pix, ci] = sum_(tap, co) dY[(pix + pad - tap) / 2, co] * W[tap, ci, co].  An input pixel sees the taps
// of its parity class only (even coordinate: taps {0, 2}, odd: tap {1} -- 4 / 2 / 2 / 1 taps), and the four pixels of a
// 2x2 BLOCK of the input (one of each class) read the same 2x2 neighbourhood of dY.  So the unit of work is the block:
// a lane's 16-pixel tile is 16 blocks, the neighbourhood is gathered once (12 16-byte loads serve all nine taps), one
// index decomposition per four output pixels, and a wave runs 9 x 3 x 4 x CIN/16 MFMAs per tile like the forward kernel
// -- W^T fragments of all nine taps in LDS (A operand [ci][co]: 16-byte loads, co is contiguous in W).  (The first form,
// one parity class per workgroup, gathered per class: 0.42 of the fp32 MFMA peak at conv3 -- the single-tap class spent
// more on its index arithmetic than on its 24 MFMAs per tile.)
// BNSTATS (ConvDgradBn): dX is the gradient w.r.t. the previous layer's batch-norm OUTPUT; the launch also leaves that
// batch norm's backward partial sums per demonstration index -- (sum dX, sum dX * xhat) from the previous layer's
// pre-norm activation read at the pixels it writes -- so no pass over (activation, dX) is needed for them.
// ==========================================================================================
constexpr int WDG_WAVES = 8;
struct WDgBlk {
    int Hb, Wb;               // blocks per frame
    int iy0[2], ix0[2];       // first row / column of parity class e: (iy + pt) & 1 == e
    int Hc[2], Wc[2];         // rows / columns of the class
    WDiv d_hw, d_w;           // / (Hb * Wb), / Wb
    const float* act; const float* mean; const float* rstd;     // BNSTATS: [pixels, CIN], [G, CIN], [G, CIN]
};

template <int CIN, int COUT, int PT, int PL, bool BNSTATS>
__global__ void __launch_bounds__(WDG_WAVES * 64)
conv_wide_dgrad_kernel(WideGeom g, WDgBlk c, const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                       WideDeal dl) {
    constexpr int CC = COUT / 16, NBI = CIN / 16;
    extern __shared__ __attribute__((aligned(16))) float wide_lds[];
    f32x4* wl = reinterpret_cast<f32x4*>(wide_lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;
    const int sg = (int)blockIdx.x / dl.S, ss = (int)blockIdx.x - sg * dl.S;
    // ---- W^T -> LDS: wl[((tap * CC + cc) * NBI + b) * 64 + l][j] = W[tap][ci = 16 b + p][co = 16 cc + 4q + j]
    for (int i = tid; i < 9 * CC * NBI * 64; i += WDG_WAVES * 64) {
        const int l = i & 63;
        int r = i >> 6;
        const int b = r % NBI;
        r /= NBI;
        const int cc = r % CC, tap = r / CC;
        wl[i] = wldg4(w + (tap * CIN + b * 16 + (l & 15)) * COUT + cc * 16 + 4 * (l >> 4));
    }
    f32x4 fs[NBI], fq[NBI], mu4[NBI];       // BNSTATS: per-lane sums of dX and of dX * (activation - mean)
#pragma unroll
    for (int b = 0; b < NBI; ++b) {
        fs[b] = fq[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        mu4[b] = BNSTATS ? wldg4(c.mean + sg * CIN + b * 16 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int tlo = ss * dl.per_slice, thi = min(tlo + dl.per_slice, dl.per_idx);
    int tile = tlo + __builtin_amdgcn_readfirstlane(wave);
    int tb = tile / dl.tps, tr = tile - tb * dl.tps;
    const int HbWb = c.Hb * c.Wb;
    for (; tile < thi; tile += WDG_WAVES) {
        const int local = tr * 16 + p;                        // block of this lane inside its sequence
        const bool valid = local < dl.seqpix;
        const int lc = valid ? local : 0;
        const int t = wdiv(lc, c.d_hw);
        const int rem = lc - t * HbWb;
        const int a = wdiv(rem, c.d_w), bcol = rem - a * c.Wb;
        const int n = (tb * dl.G + sg) * (dl.seqpix / HbWb) + t;      // frame
        // ---- the 2x2 neighbourhood of dY: rows a + PT - 1, a + PT; columns likewise
        const int r_lo = a + PT - 1, c_lo = bcol + PL - 1;
        bool rok[2], cok[2];
        int roff[2], coff[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            rok[k] = valid & ((unsigned)(r_lo + k) < (unsigned)g.Ho);
            cok[k] = (unsigned)(c_lo + k) < (unsigned)g.Wo;
            roff[k] = (n * g.Ho + wclamp(r_lo + k, g.Ho - 1)) * g.Wo;
            coff[k] = wclamp(c_lo + k, g.Wo - 1);
        }
        f32x4 v[2][2][CC];
        float okf[2][2];
#pragma unroll
        for (int ry = 0; ry < 2; ++ry)
#pragma unroll
            for (int rx = 0; rx < 2; ++rx) {
                unsigned off = (unsigned)((roff[ry] + coff[rx]) * COUT + 4 * q);
                D2P_OPAQUE_U(off);
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) v[ry][rx][cc] = wldg4(dy + off + cc * 16);
                okf[ry][rx] = (rok[ry] & cok[rx]) ? 1.f : 0.f;
            }
        // ---- nine taps: tap (ky, kx) belongs to class (ky odd, kx odd) and reads neighbour (row(ky), col(kx)):
        //      ky = 0 -> the high row, ky = 2 -> the low row, ky = 1 -> row a (= high for PT = 0, low for PT = 1)
        f32x4 acc[4][2][NBI];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < NBI; ++b) acc[e][0][b] = acc[e][1][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            constexpr int dummy = 0;
            (void)dummy;
            const int ky = tap / 3, kx = tap % 3;
            const int ey = ky & 1, ex = kx & 1;
            const int ry = ky == 0 ? 1 : (ky == 2 ? 0 : (PT ? 0 : 1));
            const int rx = kx == 0 ? 1 : (kx == 2 ? 0 : (PL ? 0 : 1));
            const int e = 2 * ey + ex;
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                const f32x4 bb = v[ry][rx][cc] * okf[ry][rx];
#pragma unroll
                for (int b = 0; b < NBI; ++b) {
                    const f32x4 a4 = wl[((tap * CC + cc) * NBI + b) * 64 + lane];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[e][jj & 1][b] = D2P_MFMA16(a4[jj], bb[jj], acc[e][jj & 1][b]);
                }
            }
        }
        // ---- the block's four input pixels
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ey = e >> 1, ex = e & 1;
            const bool in = valid & (a < c.Hc[ey]) & (bcol < c.Wc[ex]);
            const int iy = c.iy0[ey] + 2 * a, ix = c.ix0[ex] + 2 * bcol;
            const long xoff = ((long)(n * g.H + (in ? iy : 0)) * g.W + (in ? ix : 0)) * CIN;
#pragma unroll
            for (int b = 0; b < NBI; ++b) {
                const f32x4 o = acc[e][0][b] + acc[e][1][b];
                if (in) *reinterpret_cast<f32x4*>(dx + xoff + b * 16 + 4 * q) = o;
                if (BNSTATS) {
                    const f32x4 av = wldg4(c.act + xoff + b * 16 + 4 * q);
                    const f32x4 om = o * (in ? 1.f : 0.f);
                    fs[b] += om;
                    fq[b] += om * (av - mu4[b]);       // (centred before the product: no cancellation at the end)
                }
            }
        }
        tr += WDG_WAVES;
        while (tr >= dl.tps) { tr -= dl.tps; ++tb; }
    }
    if (BNSTATS) {
        // (sum dX, sum dX * xhat) with xhat = (act - mean) rstd: rstd * sum dX (act - mean), in fp64 from here on;
        // the 16 pixel lanes by xor-shuffles, the waves through LDS in wave order
        __shared__ double wsum[WDG_WAVES * 48 * 2];
#pragma unroll
        for (int b = 0; b < NBI; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double u = (double)fs[b][r], s2 = (double)fq[b][r];
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    u += __shfl_xor(u, off, 64);
                    s2 += __shfl_xor(s2, off, 64);
                }
                if (p == 0) {
                    wsum[(wave * CIN + b * 16 + 4 * q + r) * 2] = u;
                    wsum[(wave * CIN + b * 16 + 4 * q + r) * 2 + 1] = s2;
                }
            }
        __syncthreads();
        if (tid < CIN) {
            double sd = 0.0, sda = 0.0;
#pragma unroll
            for (int wv = 0; wv < WDG_WAVES; ++wv) {
                sd += wsum[(wv * CIN + tid) * 2];
                sda += wsum[(wv * CIN + tid) * 2 + 1];
            }
            dl.stats[((long)blockIdx.x * CIN + tid) * 2] = sd;
            dl.stats[((long)blockIdx.x * CIN + tid) * 2 + 1] = (double)c.rstd[sg * CIN + tid] * sda;
        }
    }
}

// slices per index of the input-gradient launch (one round of workgroups; a tile per wave at least)
void wide_dgrad_blocks(const ConvGeom& g, WDgBlk& c) {
    for (int e = 0; e < 2; ++e) {
        c.iy0[e] = e ^ (g.pt & 1);            // (iy + pt) & 1 == e
        c.ix0[e] = e ^ (g.pl & 1);
        c.Hc[e] = c.iy0[e] < g.H ? (g.H - c.iy0[e] + 1) / 2 : 0;
        c.Wc[e] = c.ix0[e] < g.W ? (g.W - c.ix0[e] + 1) / 2 : 0;
    }
    c.Hb = c.Hc[0] > c.Hc[1] ? c.Hc[0] : c.Hc[1];
    c.Wb = c.Wc[0] > c.Wc[1] ? c.Wc[0] : c.Wc[1];
    c.d_hw = make_wdiv(c.Hb * c.Wb);
    c.d_w = make_wdiv(c.Wb);
}
int wide_dgrad_slices(const ConvGeom& g, int G, int seq) {
    WDgBlk cb{};
    wide_dgrad_blocks(g, cb);
    const int Hb = cb.Hb, Wb = cb.Wb;
    const long tps = ((long)seq * Hb * Wb + 15) / 16;
    const long per_idx = (long)g.N / (G * seq) * tps;
    // (the 16 -> 32 layer: ~100 registers and 18 KB of LDS -- two workgroups per CU, four waves per SIMD, for an HBM-bound launch)
    long S = (g.Cout == 32 ? 2 : 1) * wide_cus() / G;
    const long cap = per_idx / WDG_WAVES;
    if (S > cap) S = cap;
    return (int)(S < 1 ? 1 : S);
}

template <int CIN, int COUT>
int launch_wide_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st, const ConvDgradBn* bn) {
    WideGeom d = make_wide(g);
    WDgBlk c{};
    wide_dgrad_blocks(g, c);
    c.act = bn ? bn->act : nullptr; c.mean = bn ? bn->mean : nullptr; c.rstd = bn ? bn->rstd : nullptr;
    WideDeal dl{};
    const int G = bn ? bn->G : 1, seq = bn ? bn->seq : g.N;
    dl.G = G;
    dl.seqpix = seq * c.Hb * c.Wb;                 // (blocks of one sequence)
    dl.tps = ceil_div(dl.seqpix, 16);
    dl.per_idx = g.N / (G * seq) * dl.tps;
    dl.S = bn ? bn->S : wide_dgrad_slices(g, 1, g.N);
    dl.per_slice = ceil_div(dl.per_idx, dl.S);
    dl.stats = bn ? bn->stats : nullptr;
    constexpr size_t lds = (size_t)9 * (COUT / 16) * (CIN / 16) * 64 * sizeof(f32x4);
    D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * d.P * 9 * g.Cin * COUT);
    const dim3 grid(G * dl.S), block(WDG_WAVES * 64);
#define D2P_WIDE_DG(PT_, PL_, BN_)                                                                                  \
    do {                                                                                                            \
        auto kern = conv_wide_dgrad_kernel<CIN, COUT, PT_, PL_, BN_>;                                                     \
        static bool attr = false;                                                                                   \
        if (!attr) {                                                                                                \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            D2P_REQUIRE(e == hipSuccess, (int)e, "conv wide dgrad: %s", hipGetErrorString(e));                      \
            attr = true;                                                                                            \
        }                                                                                                           \
        hipLaunchKernelGGL(kern, grid, block, lds, st, d, c, dy, w, dx, dl);                                        \
    } while (0)
    const int pp = (g.pt ? 2 : 0) | (g.pl ? 1 : 0);
    if (bn) {
        if (pp == 0) D2P_WIDE_DG(0, 0, true);
        else if (pp == 1) D2P_WIDE_DG(0, 1, true);
        else if (pp == 2) D2P_WIDE_DG(1, 0, true);
        else D2P_WIDE_DG(1, 1, true);
    } else {
        if (pp == 0) D2P_WIDE_DG(0, 0, false);
        else if (pp == 1) D2P_WIDE_DG(0, 1, false);
        else if (pp == 2) D2P_WIDE_DG(1, 0, false);
        else D2P_WIDE_DG(1, 1, false);
    }
#undef D2P_WIDE_DG
    D2P_LAUNCH_CHECK("conv_wide_dgrad");
    return 1;
}

// ==========================================================================================
// weight gradient: dW[tap, ci, co] = sum_pix x[pix @ tap, ci] * dY[pix, co] -- the reduction index is the PIXEL (4 per
// MFMA), A = x [ci][pixel], B = dY [pixel][co], the whole dW in accumulators: 18 / 27 row blocks x 3 column blocks, split
// over the three waves of a pixel stream by the tap's row ky (6 / 9 row blocks = 72 / 108 accumulator registers each).
// A workgroup = 4 pixel streams x 3 waves; nothing goes through LDS in the loop: both operands are 4-byte buffer loads
// (16 consecutive channels of a pixel per quarter wave) whose per-lane offset comes from a TABLE in LDS, built once per
// workgroup over the pixels of one block of frames (a sequence of one demonstration index, or 16 frames of a plain
// launch): entry [ky][pixel] = the byte offsets of x at (ky, kx = 0..2) and of the dY pixel, an out-of-image tap or a
// pixel past the block = an offset past the buffer (a buffer load returns 0 there) -- no index arithmetic and no masks
// in the loop, one ds_read_b128 per four pixels' worth of loads.
// AFFINE (the input read through the previous layer's batch-norm apply, x' = sc x + sh inside the image, 0 outside):
// the index is fixed per workgroup, so the scale multiplies the finished rows of dW, the loop adds sh / sc to what it
// loads, and an out-of-image tap loads the index's pad pixel (-sh / sc) -- one add per load.
// Each workgroup leaves one [9 CIN, 48] slab (streams added in a fixed order through LDS); gemm_splitk_reduce_kernel
// adds the slabs in order.
// ==========================================================================================
constexpr int WWG_STREAMS = 4, WWG_WAVES = 3 * WWG_STREAMS;
constexpr unsigned WWG_SENT = 0x7fff0000u;         // a byte offset past every buffer (sizes are checked < 2^31 - 2^17)
constexpr int WWG_MAX_TPS = 170;                   // tiles of one block of frames: 3 x 16 x 170 table entries = 127.5 KB

struct WideWg {
    int G, S, seq, fb, nsub;     // frames per sequence, per block, blocks per sequence
    int blkpix, tps;             // output pixels / tiles of one block
    int per_idx, per_slice;      // tiles of one index / of one slice
    const float* in_scale;       // [G, CIN] or null
    unsigned pad0;               // AFFINE: float offset of index 0's pad pixel in x
    unsigned x_bytes, dy_bytes;  // buffer sizes (x: incl. the pad pixels when AFFINE)
};

template <int CIN, bool AFFINE, bool TAIL>
__global__ void __launch_bounds__(WWG_WAVES * 64)
conv_wide_wgrad_kernel(WideGeom g, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slabs,
                       WideWg dl) {
    constexpr int CB = CIN / 16, NA = 3 * CB;           // row blocks of one wave: (kx, cb)
    extern __shared__ __attribute__((aligned(16))) float wide_lds[];
    uint4* tab = reinterpret_cast<uint4*>(wide_lds);                // [3][tps * 16]
    f32x4* red = reinterpret_cast<f32x4*>(wide_lds);                // (after the loop) [3][NA][3][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), p = lane & 15, q = lane >> 4;
    const int ks = wave % 3, ps = wave / 3;
    const int sg = (int)blockIdx.x / dl.S, ss = (int)blockIdx.x - sg * dl.S;
    const int HoWo = g.Ho * g.Wo, npix = dl.tps * 16;
    // ---- the offset table of one block of frames
    for (int i = tid; i < 3 * npix; i += WWG_WAVES * 64) {
        const int ky = i / npix, local = i - ky * npix;
        uint4 e{WWG_SENT, WWG_SENT, WWG_SENT, WWG_SENT};
        if (local < dl.blkpix) {
            const int t = wdiv(local, g.d_howo);
            const int rem = local - t * HoWo;
            const int oy = wdiv(rem, g.d_wo), ox = rem - oy * g.Wo;
            const int iy = 2 * oy - g.pt + ky, ix0 = 2 * ox - g.pl;
            if ((unsigned)iy < (unsigned)g.H) {
                const unsigned row = (unsigned)((t * g.H + iy) * g.W);
                if ((unsigned)ix0 < (unsigned)g.W) e.x = (row + ix0) * CIN * 4u;
                if ((unsigned)(ix0 + 1) < (unsigned)g.W) e.y = (row + ix0 + 1) * CIN * 4u;
                if ((unsigned)(ix0 + 2) < (unsigned)g.W) e.z = (row + ix0 + 2) * CIN * 4u;
            }
            e.w = (unsigned)local * (WCO * 4u);
        }
        tab[i] = e;
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)dl.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, (int)dl.dy_bytes, 0x00020000);
    float tsh[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) tsh[cb] = AFFINE ? -x[dl.pad0 + sg * CIN + cb * 16 + p] : 0.f;
    const unsigned pad_abs = AFFINE ? (dl.pad0 + (unsigned)(sg * CIN)) * 4u : 0u;
    f32x4 acc[NA][WNB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < WNB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int tlo = ss * dl.per_slice, thi = min(tlo + dl.per_slice, dl.per_idx);
    int tile = tlo + ps;
    int tb = tile / dl.tps, tr = tile - tb * dl.tps;            // (block of this index, tile inside it)
    const unsigned frame_bytes = (unsigned)(g.H * g.W * CIN) * 4u;
    for (; tile < thi; tile += WWG_STREAMS) {
        const int sb = tb / dl.nsub, sub = tb - sb * dl.nsub;
        const int frame0 = (sb * dl.G + sg) * dl.seq + sub * dl.fb;
        const unsigned xs = (unsigned)frame0 * frame_bytes;                     // (wave-uniform: scalar offsets)
        const unsigned ys = (unsigned)frame0 * (unsigned)(HoWo * WCO * 4);
        const unsigned padrel = pad_abs - xs;
        // a plain launch's LAST block may hold fewer than fb frames: its pixels past the batch must read zeros on BOTH
        // operands -- by offsets past the buffer in the VGPR offset (the scalar offset need not be part of the range check)
        // (TAIL: a compile-time choice -- the test inside the loop cost the affine form 12 % when it was a run-time one)
        const int lim = TAIL ? min(dl.fb, g.N - frame0) * HoWo : 0;
        float A[4][NA], B[4][WNB];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 e = tab[ks * npix + tr * 16 + 4 * q + kk];
            if (TAIL && tr * 16 + 4 * q + kk >= lim) e = uint4{WWG_SENT, WWG_SENT, WWG_SENT, WWG_SENT};
            unsigned off[3] = {e.x, e.y, e.z};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if (AFFINE) off[kx] = off[kx] == WWG_SENT ? padrel : off[kx];
                off[kx] += (unsigned)(p * 4);
            }
#pragma unroll
            for (int a = 0; a < NA; ++a)
                A[kk][a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(off[a / CB] + (a % CB) * 64), (int)xs, 0));
            const unsigned ob = e.w + (unsigned)(p * 4);
#pragma unroll
            for (int b = 0; b < WNB; ++b)
                B[kk][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, (int)(ob + b * 64), (int)ys, 0));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const float av = AFFINE ? A[kk][a] + tsh[a % CB] : A[kk][a];
#pragma unroll
                for (int b = 0; b < WNB; ++b) acc[a][b] = D2P_MFMA16(av, B[kk][b], acc[a][b]);
            }
        tr += WWG_STREAMS;
        while (tr >= dl.tps) { tr -= dl.tps; ++tb; }
    }
    if (AFFINE) {
        // rows ci = 16 cb + 4q + r of every block times the input scale of ci
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const f32x4 sc = wldg4(dl.in_scale + sg * CIN + cb * 16 + 4 * q);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int b = 0; b < WNB; ++b) acc[kx * CB + cb][b] *= sc;
        }
    }
    // ---- the four pixel streams added in a fixed order (3 + 2, then + 1, then + 0) through LDS
    __syncthreads();                                  // (every wave is done with the table)
#pragma unroll 1
    for (int s_ = WWG_STREAMS - 1; s_ >= 1; --s_) {
        if (ps == s_) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < WNB; ++b) red[((ks * NA + a) * WNB + b) * 64 + lane] = acc[a][b];
        }
        __syncthreads();
        if (ps == s_ - 1) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < WNB; ++b) acc[a][b] += red[((ks * NA + a) * WNB + b) * 64 + lane];
        }
        __syncthreads();
    }
    if (ps == 0) {
        float* slab = slabs + (long)blockIdx.x * (9 * CIN * WCO);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int kx = a / CB, cb = a % CB;
            const int row0 = (ks * 3 + kx) * CIN + cb * 16 + 4 * q;
#pragma unroll
            for (int b = 0; b < WNB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(row0 + r) * WCO + b * 16 + p] = acc[a][b][r];
        }
    }
}

// frames per block of the offset table: a whole sequence when its tiles fit the table, else its largest divisor that does
int wide_wg_fb(const ConvGeom& g, int seq) {
    const int howo = g.Ho * g.Wo;
    for (int fb = seq; fb >= 1; --fb)
        if (seq % fb == 0 && ceil_div(fb * howo, 16) <= WWG_MAX_TPS) return fb;
    return 0;
}
bool wide_wgrad_ok(const ConvGeom& g) {
    return wide_geom_ok(g) && (size_t)g.N * g.H * g.W * g.Cin * 4 + (size_t)4096 * g.Cin * 4 < (1ull << 31) - (1u << 17) &&
           (size_t)g.N * g.Ho * g.Wo * WCO * 4 < (1ull << 31) - (1u << 17) && g.Ho * g.Wo <= 16 * WWG_MAX_TPS;
}
int wide_wgrad_blocks(const ConvGeom& g, int G, int seq, WideWg* out) {
    WideWg dl{};
    const bool plain = G <= 0;
    dl.G = plain ? 1 : G;
    dl.seq = plain ? wide_wg_fb(g, 16) : seq;
    dl.fb = plain ? dl.seq : wide_wg_fb(g, seq);
    if (dl.fb < 1) return 0;
    dl.nsub = dl.seq / dl.fb;
    dl.blkpix = dl.fb * g.Ho * g.Wo;
    dl.tps = ceil_div(dl.blkpix, 16);
    const long nseq = plain ? ceil_div(g.N, dl.seq) : g.N / (dl.G * dl.seq);
    dl.per_idx = (int)(nseq * dl.nsub * dl.tps);
    long S = wide_cus() / dl.G;
    const long cap = dl.per_idx / WWG_STREAMS;
    if (S > cap) S = cap;
    dl.S = (int)(S < 1 ? 1 : S);
    dl.per_slice = ceil_div(dl.per_idx, dl.S);
    if (out) *out = dl;
    return dl.G * dl.S;
}

template <int CIN>
int launch_wide_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st,
                      const ConvBnFold* bn) {
    WideGeom d = make_wide(g);
    WideWg dl;
    const int blocks = wide_wgrad_blocks(g, bn ? bn->G : 0, bn ? bn->seq : 0, &dl);
    if (blocks < 1) return 0;
    const int KK = 9 * CIN;
    D2P_REQUIRE(ws && ws_bytes >= (size_t)blocks * KK * WCO * sizeof(float), D2P_EWS, "conv wide wgrad: workspace too small (%zu bytes)",
                ws_bytes);
    dl.in_scale = bn ? bn->in_scale : nullptr;
    dl.pad0 = (unsigned)((size_t)g.N * g.H * g.W * g.Cin);
    dl.x_bytes = (unsigned)(((size_t)g.N * g.H * g.W * g.Cin + (bn ? (size_t)dl.G * g.Cin : 0)) * 4);
    dl.dy_bytes = (unsigned)((size_t)g.N * g.Ho * g.Wo * WCO * 4);
    const size_t tab_bytes = (size_t)3 * dl.tps * 16 * sizeof(uint4), red_bytes = (size_t)3 * 3 * (CIN / 16) * WNB * 64 * sizeof(f32x4);
    const size_t lds = tab_bytes > red_bytes ? tab_bytes : red_bytes;
    float* slabs = (float*)ws;
    {
        D2pProfScope prof(st, D2P_PROF_CONV, 2.0 * d.P * KK * WCO);
#define D2P_WIDE_WG(AF, TL)                                                                                              \
    do {                                                                                                                 \
        auto kern = conv_wide_wgrad_kernel<CIN, AF, TL>;                                                                 \
        static size_t have = 0;                                                                                          \
        if (lds > have) {                                                                                                \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            D2P_REQUIRE(e == hipSuccess, (int)e, "conv wide wgrad: %s", hipGetErrorString(e));                           \
            have = lds;                                                                                                  \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(WWG_WAVES * 64), lds, st, d, x, dy, slabs, dl);                      \
    } while (0)
        if (dl.in_scale) D2P_WIDE_WG(true, false);            // (sequences are whole blocks: no tail)
        else if (!bn && g.N % dl.fb != 0) D2P_WIDE_WG(false, true);
        else D2P_WIDE_WG(false, false);
#undef D2P_WIDE_WG
        D2P_LAUNCH_CHECK("conv_wide_wgrad");
        EpiDense ep{dw, WCO, nullptr, 0, 0};
        const long total = (long)KK * WCO;
        const int rb = (int)((total * 16 + 255) / 256);
        hipLaunchKernelGGL((gemm_splitk_reduce_kernel<EpiDense>), dim3(rb), dim3(256), 0, st, ep, slabs, KK, WCO, blocks);
        D2P_LAUNCH_CHECK("conv_wide_wgrad_combine");
    }
    return 1;
}

}   // namespace

// slices per demonstration index the folding forward launch writes statistics for; 0: geometry not taken
int d2p_conv_wide_bn_slices(const ConvGeom& g, int G, int seq) {
    if (!wide_geom_ok(g) || G < 1 || seq < 1 || g.N % (G * seq) != 0 || G > 4096) return 0;
    const long tps = ((long)seq * g.Ho * g.Wo + 15) / 16;
    return wide_slices((long)g.N / (G * seq) * tps, G, g.Cin);
}

// 1: handled, 0: not a geometry of this back end, < 0: error (the convention of conv_geom.h)
int d2p_conv_wide_fwd(const ConvGeom& g, const void* x, int x_is_u8, const float* w, const float* bias, int act, float* y,
                      hipStream_t st, const ConvBnFold* bn) {
    if (x_is_u8 || !wide_geom_ok(g)) return 0;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || (bias && ((uintptr_t)bias & 15))) return 0;
    if (bn) {
        if (bn->G < 1 || bn->seq < 1 || g.N % (bn->G * bn->seq) != 0 || bn->G > 4096) return 0;
        if (bn->stats && bn->S != d2p_conv_wide_bn_slices(g, bn->G, bn->seq)) return 0;
        if (bn->in_scale && (((uintptr_t)bn->in_scale) & 15)) return 0;
        if (!bn->stats && !bn->in_scale) bn = nullptr;
    }
    if (g.N == 0) return 1;
    if (g.Cin == 32) return launch_wide_fwd<32>(g, (const float*)x, w, bias, act, y, st, bn);
    return launch_wide_fwd<48>(g, (const float*)x, w, bias, act, y, st, bn);
}

// the input-gradient kernel also takes the 16 -> 32 layer of the large frames (40x40 -> 20x20): one 16-channel block out,
// two in -- HBM-bound there (it writes 655 MB at config 4), which the block form serves with a quarter of the index
// arithmetic and 8 instead of 18 gathers per four output pixels
static int g_wide_dgrad_1632 = 1;            // (0: that layer stays on the row-strip kernel; d2p_conv_wide_set_dgrad_1632)
void d2p_conv_wide_set_dgrad_1632(int on) { g_wide_dgrad_1632 = on ? 1 : 0; }
static bool wide_dgrad_geom_ok(const ConvGeom& g) {
    if (wide_geom_ok(g)) return true;
    return g_wide_dgrad_1632 && g.Cin == 16 && g.Cout == 32 && g.H >= 3 && g.W >= 3 && g.H * g.W >= 400 &&
           (size_t)g.N * g.H * g.W * g.Cin < (1ull << 31) && (size_t)g.N * g.Ho * g.Wo * g.Cout < (1ull << 32) / 2;
}

// slices per index of the input-gradient launch that also leaves the previous layer's batch-norm-backward sums; 0: not taken
int d2p_conv_wide_dgrad_bn_slices(const ConvGeom& g, int G, int seq) {
    if (!wide_dgrad_geom_ok(g) || G < 1 || seq < 1 || g.N % (G * seq) != 0 || G > 4096) return 0;
    return wide_dgrad_slices(g, G, seq);
}

int d2p_conv_wide_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, hipStream_t st, const ConvDgradBn* bn) {
    if (!wide_dgrad_geom_ok(g)) return 0;
    if (((uintptr_t)dy & 15) || ((uintptr_t)dx & 15) || ((uintptr_t)w & 15)) return 0;
    if ((size_t)g.N * g.Ho * g.Wo * g.Cout >= (1ull << 32)) return 0;
    if (bn) {
        if (!bn->act || !bn->mean || !bn->rstd || !bn->stats || ((uintptr_t)bn->act & 15)) return 0;
        if (bn->S < 1 || bn->S != d2p_conv_wide_dgrad_bn_slices(g, bn->G, bn->seq)) return 0;
    }
    if (g.N == 0) return 1;
    if (g.Cout == 32) return launch_wide_dgrad<16, 32>(g, dy, w, dx, st, bn);
    if (g.Cin == 32) return launch_wide_dgrad<32, 48>(g, dy, w, dx, st, bn);
    return launch_wide_dgrad<48, 48>(g, dy, w, dx, st, bn);
}

size_t d2p_conv_wide_wgrad_ws(const ConvGeom& g) {
    if (!wide_wgrad_ok(g)) return 0;
    // (the most workgroups any dealing of this geometry uses: one per CU)
    return (size_t)wide_cus() * 9 * g.Cin * WCO * sizeof(float);
}

// bn (optional): in_scale / in_shift = the input read through the previous layer's batch-norm apply (the pad pixels behind x)
int d2p_conv_wide_wgrad(const ConvGeom& g, const void* x, int x_is_u8, const float* dy, float* dw, void* ws, size_t ws_bytes,
                        hipStream_t st, const ConvBnFold* bn) {
    if (x_is_u8 || !wide_wgrad_ok(g)) return 0;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15)) return 0;
    if (bn) {
        if (bn->G < 1 || bn->seq < 1 || g.N % (bn->G * bn->seq) != 0 || bn->G > 4096 || !bn->in_scale) return 0;
        if (((uintptr_t)bn->in_scale) & 15) return 0;
    }
    if (g.N == 0) return 0;
    if (g.Cin == 32) return launch_wide_wgrad<32>(g, (const float*)x, dy, dw, ws, ws_bytes, st, bn);
    return launch_wide_wgrad<48>(g, (const float*)x, dy, dw, ws, ws_bytes, st, bn);
}

// ---- the 16 -> 32 layer of the large frames in block form (conv_wide_fwd2_kernel)
static int g_wide_fwd2 = 1;                  // (0: that layer stays on the gather kernel of conv_direct.hip)
void d2p_conv_wide_set_fwd2(int on) { g_wide_fwd2 = on ? 1 : 0; }
int d2p_conv_wide_fwd2_bn_slices(const ConvGeom& g, int G, int seq) {
    if (!g_wide_fwd2 || !wide_fwd2_geom_ok(g) || G < 1 || seq < 1 || g.N % (G * seq) != 0 || G > 4096) return 0;
    return wide_fwd2_slices(g, G, seq);
}
int d2p_conv_wide_fwd2(const ConvGeom& g, const void* x, int x_is_u8, const float* w, const float* bias, int act, float* y,
                       hipStream_t st, const ConvBnFold* bn) {
    if (!g_wide_fwd2 || x_is_u8 || !wide_fwd2_geom_ok(g)) return 0;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || (bias && ((uintptr_t)bias & 15))) return 0;
    if (bn) {
        if (bn->G < 1 || bn->seq < 1 || g.N % (bn->G * bn->seq) != 0 || bn->G > 4096) return 0;
        if (bn->stats && bn->S != d2p_conv_wide_fwd2_bn_slices(g, bn->G, bn->seq)) return 0;
        if (bn->in_scale && (((uintptr_t)bn->in_scale) & 15)) return 0;
        if (!bn->stats && !bn->in_scale) bn = nullptr;
    }
    if (g.N == 0) return 1;
    return launch_wide_fwd2<16, 32>(g, (const float*)x, w, bias, act, y, st, bn);
}
