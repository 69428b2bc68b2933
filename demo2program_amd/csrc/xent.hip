// K7: masked, mask-count-normalised sequence cross-entropies (include/d2p.h).
// Replaces Sequence_Loss at models/model_full.py:620-657 and the loss sum at :918-932,
// :1014-1038, :1061-1079.
//
// One wavefront per (t, r) logits row: V (50 / 42 program tokens, 6 / 12 actions, 5 / 6
// perception bits) fits the 64 lanes, so max / sum-exp / dot products are pure cross-lane
// reductions with no LDS.  Per-group numerators / denominators are reduced in two
// deterministic stages (fixed partial order), never with float atomics.
#include "common.h"

#define XENT_S 64   // row splits per group in the forward reduction

extern "C" size_t d2p_xent_ws_bytes(int G) {
    return G > 0 ? (size_t)G * XENT_S * 2 * sizeof(float) : 0;
}

struct LabelView {
    const float* p;
    long rs, ts, vs;
    __device__ __forceinline__ float at(int r, int t, int v) const {
        return p[(long)r * rs + (long)t * ts + (long)v * vs];
    }
};

// MODE 0: softmax CE = lse*sum(lab) - sum(lab*x);  MODE 1: mean_v sigmoid CE.
template <int MODE>
__device__ __forceinline__ float row_loss(const float* x, bool have_logits, const LabelView& lab,
                                          int r, int t, int V, int lane) {
    if (MODE == 0) {
        float mx = -INFINITY;
        for (int v = lane; v < V; v += 64) mx = fmaxf(mx, have_logits ? x[v] : 0.f);
        mx = wave_reduce_max(mx);
        float se = 0.f, sl = 0.f, slx = 0.f;
        for (int v = lane; v < V; v += 64) {
            const float xv = have_logits ? x[v] : 0.f;
            const float lv = lab.at(r, t, v);
            se += expf(xv - mx);
            sl += lv;
            slx += lv * xv;
        }
        se = wave_reduce_sum(se);
        sl = wave_reduce_sum(sl);
        slx = wave_reduce_sum(slx);
        return (mx + logf(se)) * sl - slx;
    } else {
        float s = 0.f;
        for (int v = lane; v < V; v += 64) {
            const float xv = have_logits ? x[v] : 0.f;
            const float lv = lab.at(r, t, v);
            // [TF-1.3] max(x,0) - x*z + log(1 + exp(-|x|))
            s += fmaxf(xv, 0.f) - xv * lv + log1pf(expf(-fabsf(xv)));
        }
        return wave_reduce_sum(s) / (float)V;
    }
}

template <int MODE>
__global__ void __launch_bounds__(256)
xent_fwd_partial_kernel(int T, int R, int V, int G, int n_steps, const float* logits,
                        LabelView lab, const int* lens, float* partial) {
    __shared__ float red[2][4];
    const int g = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = R / G;               // rows per group per time step
    const int n = T * rg;
    float num = 0.f, den = 0.f;
    for (int j = s * 4 + wave; j < n; j += S * 4) {
        const int t = j / rg, r = (j - t * rg) * G + g;
        if (t < lens[r]) {              // tf.sequence_mask(len, maxlen=T)
            num += row_loss<MODE>(logits + ((long)t * R + r) * V, t < n_steps, lab, r, t, V, lane);
            den += 1.f;
        }
    }
    if (lane == 0) { red[0][wave] = num; red[1][wave] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((long)g * S + s) * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[((long)g * S + s) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ void xent_fwd_final_kernel(int G, int S, const float* partial, float* num, float* den) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    float a = 0.f, b = 0.f;
    for (int s = 0; s < S; ++s) {
        a += partial[((long)g * S + s) * 2 + 0];
        b += partial[((long)g * S + s) * 2 + 1];
    }
    num[g] = a;
    den[g] = b;
}

template <int MODE>
__global__ void __launch_bounds__(256)
xent_bwd_kernel(int R, int V, int G, int n_steps, const float* logits, LabelView lab,
                const int* lens, const float* den, float scale, float* dlogits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nrows = (long)n_steps * R;
    for (long row = blockIdx.x * 4L + wave; row < nrows; row += (long)gridDim.x * 4L) {
        const int t = (int)(row / R), r = (int)(row - (long)t * R);
        const float* x = logits + row * V;
        float* dx = dlogits + row * V;
        if (t >= lens[r]) {
            for (int v = lane; v < V; v += 64) dx[v] = 0.f;
            continue;
        }
        const float w = scale / ((float)G * den[r % G]);
        if (MODE == 0) {
            float mx = -INFINITY;
            for (int v = lane; v < V; v += 64) mx = fmaxf(mx, x[v]);
            mx = wave_reduce_max(mx);
            float se = 0.f;
            for (int v = lane; v < V; v += 64) se += expf(x[v] - mx);
            se = wave_reduce_sum(se);
            const float inv = 1.f / se;
            // [TF-1.3] SoftmaxCrossEntropyWithLogits backprop = softmax - labels
            for (int v = lane; v < V; v += 64) dx[v] = w * (expf(x[v] - mx) * inv - lab.at(r, t, v));
        } else {
            const float wv = w / (float)V;
            for (int v = lane; v < V; v += 64) dx[v] = wv * (d2p_sigmoid(x[v]) - lab.at(r, t, v));
        }
    }
}

// ---- loss backward of several decoders in ONE launch, through the output projection -------------------------------
// dlogits = d loss / d logits (as xent_bwd_kernel) AND dhout = dlogits . proj^T (the Dense(use_bias=False) projection
// of models/model_full.py:463-464, proj [U, V]) for up to three decoders: round 2 ran three xent_bwd launches and
// three K = V (5 / 6 / 50) GEMMs in front of the backward recurrences.  A workgroup takes 16 logits rows: each wave
// turns 4 of them into dlogits (one lane per token, cross-lane reductions), the rows meet in LDS, then every thread
// owns 2 units x 16 rows of dhout (V multiply-adds each; proj rows stay in registers).
#define XB_ROWS 16
#define XB_MAXV 64
struct XbProb {
    int mode, R, V, G, n_steps, U;
    const float* logits; LabelView lab; const int* lens; const float* den; float scale;
    float* dlogits; const float* proj; float* dhout;
    int blk0;             // first workgroup of this problem
    const float* hout; float* logits_out;     // optional (round 4): the logits themselves, hout . proj, in front
    float* loss_part;     // optional (round 4): [workgroup of this problem][G] sums of the rows' loss values
};
struct XbArgs { int n; XbProb p[3]; };

// logits = hout . proj of a workgroup's 16 rows (round 4: the Dense(use_bias=False) projection of
// models/model_full.py:463-464 in FRONT of the loss backward, in the same launch).  The 5- / 6- / 50-column products
// were three skinny GEMM launches of 17 us each on the critical path between the decoders' forward and backward
// recurrences -- 50 workgroups each walking K = 512 alone; here ~900 workgroups do 16 rows each.
// Round 5: on the matrix pipe.  [16 rows x U] . [U x V] is a 16 x 16 x 4 MFMA shape: lane (row m, k quarter kq) loads
// 16 bytes of its hout row straight from memory (four consecutive k: the A operands of four MFMAs, no LDS staging of
// the 32 KB of rows), the B operands are proj[k][n] for the tile's 16 columns (zeros past V); the four waves take a
// quarter of U each and meet in LDS in wave order.  (Round 4's form staged the rows in LDS and multiplied on the vector
// ALUs: 20 us of the launch's 42.)
typedef float xb_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void xb_logits16(const XbProb& q, long row0, long nrows, float* part, float (*lg)[XB_MAXV]) {
    const int V = q.V, U = q.U, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, m = lane & 15, kq = lane >> 4;
    const int NT = (V + 15) >> 4;
    const long row = row0 + m < nrows ? row0 + m : nrows - 1;            // (rows past the end: repeated, never stored)
    const int kw = U >> 2;                                               // this wave's share of K
    const float* hp = q.hout + row * U + w * kw + 4 * kq;
    xb_f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = xb_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < kw; c += 16) {
        const float4 a4 = *reinterpret_cast<const float4*>(hp + c);
        const float* pk = q.proj + (long)(w * kw + c + 4 * kq) * V + m;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (nt < NT) {
                const bool ok = nt * 16 + m < V;
                const float* pp = pk + nt * 16;
                const float b0 = ok ? pp[0] : 0.f, b1 = ok ? pp[V] : 0.f, b2 = ok ? pp[2 * V] : 0.f, b3 = ok ? pp[3 * V] : 0.f;
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b2, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b3, acc[nt], 0, 0, 0);
            }
        }
    }
    // part[wave][tile][r][lane]: lane (n = lane & 15, q = lane >> 4) holds rows 4 q + r of column 16 tile + n
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
        if (nt < NT)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[((w * 4 + nt) * 4 + r) * 64 + lane] = acc[nt][r];
    __syncthreads();
    for (int i = tid; i < XB_ROWS * NT * 16; i += 256) {
        const int r16 = i / (NT * 16), vv = i - r16 * (NT * 16);
        const int nt = vv >> 4, n = vv & 15, l2 = (r16 >> 2) * 16 + n, r = r16 & 3;
        float sum = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) sum += part[((w2 * 4 + nt) * 4 + r) * 64 + l2];
        if (vv < V) {
            lg[r16][vv] = sum;
            if (row0 + r16 < nrows) q.logits_out[(row0 + r16) * V + vv] = sum;
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256)
xent_bwd_dhout_kernel(XbArgs a) {
    __shared__ float dl[XB_ROWS][XB_MAXV];
    int pi = 0;
    if (a.n > 1 && (int)blockIdx.x >= a.p[1].blk0) pi = 1;
    if (a.n > 2 && (int)blockIdx.x >= a.p[2].blk0) pi = 2;
    const XbProb& q = a.p[pi];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nrows = (long)q.n_steps * q.R;
    const long row0 = ((long)blockIdx.x - q.blk0) * XB_ROWS;
    const int V = q.V;
    const bool own_logits = q.hout != nullptr;
    if (own_logits) {
        extern __shared__ __attribute__((aligned(16))) float xb_dyn[];      // [4 waves][4 tiles][4][64] partial logits tiles
        xb_logits16(q, row0, nrows, xb_dyn, dl);     // dl holds the 16 rows' logits
    }
    __shared__ float rowloss[XB_ROWS];
    __shared__ int rowgrp[XB_ROWS];
    const bool want_loss = q.loss_part != nullptr;
    for (int i = 0; i < 4; ++i) {
        const int lr = wave * 4 + i;
        const long row = row0 + lr;
        float out = 0.f, rl = 0.f;
        int grp = -1;
        if (row < nrows) {
            const int t = (int)(row / q.R), r = (int)(row - (long)t * q.R);
            float xl = 0.f;                                        // this lane's logit of the row
            if (lane < V) xl = own_logits ? dl[lr][lane] : q.logits[row * V + lane];
            if (t < q.lens[r]) {
                grp = r % q.G;
                const float w = q.scale / ((float)q.G * q.den[grp]);
                const float lb = lane < V ? q.lab.at(r, t, lane) : 0.f;
                if (q.mode == 0) {
                    float mx = lane < V ? xl : -INFINITY;
                    mx = wave_reduce_max(mx);
                    const float ex = lane < V ? expf(xl - mx) : 0.f;
                    const float se = wave_reduce_sum(ex);
                    // [TF-1.3] SoftmaxCrossEntropyWithLogits backprop = softmax - labels
                    if (lane < V) out = w * (ex * (1.f / se) - lb);
                    if (want_loss) {                               // the row's loss value, as row_loss<0> computes it
                        const float sl = wave_reduce_sum(lb), slx = wave_reduce_sum(lb * xl);
                        rl = (mx + logf(se)) * sl - slx;
                    }
                } else {
                    if (lane < V) out = (w / (float)V) * (d2p_sigmoid(xl) - lb);
                    if (want_loss) {
                        // [TF-1.3] max(x,0) - x*z + log(1 + exp(-|x|))
                        const float e = lane < V ? fmaxf(xl, 0.f) - xl * lb + log1pf(expf(-fabsf(xl))) : 0.f;
                        rl = wave_reduce_sum(e) / (float)V;
                    }
                }
            }
            if (lane < V) q.dlogits[row * V + lane] = out;
        }
        dl[lr][lane] = lane < V ? out : 0.f;
        if (want_loss && lane == 0) { rowloss[lr] = rl; rowgrp[lr] = grp; }
    }
    __syncthreads();
    if (want_loss && (int)threadIdx.x < q.G) {
        // this workgroup's 16 rows by loss group (row order): the sums of all workgroups are added by
        // d2p_loss_from_partials in workgroup order
        float sum = 0.f;
        for (int lr = 0; lr < XB_ROWS; ++lr)
            if (rowgrp[lr] == (int)threadIdx.x) sum += rowloss[lr];
        q.loss_part[((long)blockIdx.x - q.blk0) * q.G + threadIdx.x] = sum;
    }
    const int U = q.U;
    for (int u = threadIdx.x; u < U; u += 256) {
        float acc[XB_ROWS];
#pragma unroll
        for (int r = 0; r < XB_ROWS; ++r) acc[r] = 0.f;
        const float* pr = q.proj + (long)u * V;
        for (int v = 0; v < V; ++v) {
            const float w = pr[v];
#pragma unroll
            for (int r = 0; r < XB_ROWS; ++r) acc[r] += dl[r][v] * w;
        }
#pragma unroll
        for (int r = 0; r < XB_ROWS; ++r)
            if (row0 + r < nrows) q.dhout[(row0 + r) * U + u] = acc[r];
    }
}

extern "C" int d2p_xent_bwd_dhout_multi(int nprob, const d2p_xent_bwd_desc* d, d2p_stream_t stream) {
    D2P_REQUIRE(nprob >= 1 && nprob <= 3 && d, D2P_EINVAL, "xent_bwd_dhout: 1..3 problems");
    XbArgs a;
    a.n = 0;
    int blocks = 0;
    for (int i = 0; i < nprob; ++i) {
        const d2p_xent_bwd_desc& q = d[i];
        D2P_REQUIRE(q.R >= 0 && q.V > 0 && q.V <= XB_MAXV && q.G > 0 && q.n_steps >= 0 && q.U > 0 && q.R % q.G == 0,
                    D2P_EINVAL, "xent_bwd_dhout: bad sizes R=%d V=%d G=%d n_steps=%d U=%d", q.R, q.V, q.G, q.n_steps, q.U);
        if (q.n_steps == 0 || q.R == 0) continue;
        D2P_REQUIRE((q.logits || q.hout) && q.labels && q.lens && q.den && q.dlogits && q.proj && q.dhout, D2P_EINVAL,
                    "xent_bwd_dhout: null pointer");
        D2P_REQUIRE(!q.hout || (q.logits_out && q.U % 128 == 0 && q.U <= 512 && (((uintptr_t)q.hout & 15) == 0)), D2P_EINVAL,
                    "xent_bwd_dhout: hout needs logits_out, U %% 128 == 0, U <= 512 and 16-byte alignment");
        XbProb& o = a.p[a.n++];
        o.mode = q.sigmoid ? 1 : 0; o.R = q.R; o.V = q.V; o.G = q.G; o.n_steps = q.n_steps; o.U = q.U;
        o.logits = q.logits; o.lab = LabelView{q.labels, q.label_rs, q.label_ts, q.label_vs}; o.lens = q.lens;
        o.den = q.den; o.scale = q.scale; o.dlogits = q.dlogits; o.proj = q.proj; o.dhout = q.dhout;
        o.hout = q.hout; o.logits_out = q.logits_out; o.loss_part = q.loss_part;
        o.blk0 = blocks;
        blocks += (int)(((long)q.n_steps * q.R + XB_ROWS - 1) / XB_ROWS);
    }
    if (a.n == 0) return D2P_OK;
    for (int i = a.n; i < 3; ++i) a.p[i] = a.p[0];
    size_t dyn = 0;
    for (int i = 0; i < a.n; ++i)
        if (a.p[i].hout) {
            const size_t need = (size_t)4 * 4 * 4 * 64 * sizeof(float);
            dyn = need > dyn ? need : dyn;
        }
    hipLaunchKernelGGL(xent_bwd_dhout_kernel, dim3(blocks), dim3(256), dyn, as_stream(stream), a);
    D2P_LAUNCH_CHECK("xent_bwd_dhout");
    return D2P_OK;
}

static int xent_check(int T, int R, int V, int G, int n_steps) {
    D2P_REQUIRE(T >= 0 && R >= 0 && V > 0 && G > 0 && n_steps >= 0 && n_steps <= T, D2P_EINVAL,
                "xent: bad sizes T=%d R=%d V=%d G=%d n_steps=%d", T, R, V, G, n_steps);
    D2P_REQUIRE(R % G == 0, D2P_EINVAL, "xent: R=%d not a multiple of G=%d", R, G);
    return D2P_OK;
}

template <int MODE>
static int xent_fwd(int T, int R, int V, int G, int n_steps, const float* logits,
                    const float* labels, long lrs, long lts, long lvs, const int* lens,
                    float* num, float* den, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    int rc = xent_check(T, R, V, G, n_steps);
    if (rc) return rc;
    D2P_REQUIRE(num && den && lens && labels && (n_steps == 0 || logits), D2P_EINVAL,
                "xent fwd: null pointer");
    D2P_REQUIRE(ws && ws_bytes >= d2p_xent_ws_bytes(G), D2P_EWS, "xent fwd: workspace too small");
    hipStream_t st = as_stream(stream);
    LabelView lab{labels, lrs, lts, lvs};
    float* partial = (float*)ws;
    hipLaunchKernelGGL((xent_fwd_partial_kernel<MODE>), dim3(G, XENT_S), dim3(256), 0, st, T, R, V, G,
                       n_steps, logits, lab, lens, partial);
    D2P_LAUNCH_CHECK("xent_fwd_partial");
    hipLaunchKernelGGL(xent_fwd_final_kernel, dim3(ceil_div(G, 64)), dim3(64), 0, st, G, XENT_S,
                       partial, num, den);
    D2P_LAUNCH_CHECK("xent_fwd_final");
    return D2P_OK;
}

template <int MODE>
static int xent_bwd(int T, int R, int V, int G, int n_steps, const float* logits,
                    const float* labels, long lrs, long lts, long lvs, const int* lens,
                    const float* den, float scale, float* dlogits, d2p_stream_t stream) {
    int rc = xent_check(T, R, V, G, n_steps);
    if (rc) return rc;
    if (n_steps == 0 || R == 0) return D2P_OK;
    D2P_REQUIRE(logits && labels && lens && den && dlogits, D2P_EINVAL, "xent bwd: null pointer");
    LabelView lab{labels, lrs, lts, lvs};
    long blocks = ((long)n_steps * R + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((xent_bwd_kernel<MODE>), dim3((int)blocks), dim3(256), 0, as_stream(stream), R,
                       V, G, n_steps, logits, lab, lens, den, scale, dlogits);
    D2P_LAUNCH_CHECK("xent_bwd");
    return D2P_OK;
}

extern "C" int d2p_softmax_xent_masked_fwd(int T, int R, int V, int G, int n_steps,
                                           const float* logits, const float* labels, long lrs,
                                           long lts, long lvs, const int* lens, float* loss_num,
                                           float* loss_den, void* ws, size_t ws_bytes,
                                           d2p_stream_t stream) {
    return xent_fwd<0>(T, R, V, G, n_steps, logits, labels, lrs, lts, lvs, lens, loss_num, loss_den,
                       ws, ws_bytes, stream);
}
extern "C" int d2p_sigmoid_xent_masked_fwd(int T, int R, int V, int G, int n_steps,
                                           const float* logits, const float* labels, long lrs,
                                           long lts, long lvs, const int* lens, float* loss_num,
                                           float* loss_den, void* ws, size_t ws_bytes,
                                           d2p_stream_t stream) {
    return xent_fwd<1>(T, R, V, G, n_steps, logits, labels, lrs, lts, lvs, lens, loss_num, loss_den,
                       ws, ws_bytes, stream);
}
extern "C" int d2p_softmax_xent_masked_bwd(int T, int R, int V, int G, int n_steps,
                                           const float* logits, const float* labels, long lrs,
                                           long lts, long lvs, const int* lens,
                                           const float* loss_den, float scale, float* dlogits,
                                           d2p_stream_t stream) {
    return xent_bwd<0>(T, R, V, G, n_steps, logits, labels, lrs, lts, lvs, lens, loss_den, scale,
                       dlogits, stream);
}
extern "C" int d2p_sigmoid_xent_masked_bwd(int T, int R, int V, int G, int n_steps,
                                           const float* logits, const float* labels, long lrs,
                                           long lts, long lvs, const int* lens,
                                           const float* loss_den, float scale, float* dlogits,
                                           d2p_stream_t stream) {
    return xent_bwd<1>(T, R, V, G, n_steps, logits, labels, lrs, lts, lvs, lens, loss_den, scale,
                       dlogits, stream);
}

// loss = sum_terms (1/G_j) sum_g num/den   (models/model_full.py:932,1035-1038,1078-1079)
struct TermGroups { int n; int g[8]; };

__global__ void loss_assemble_kernel(TermGroups tg, const float* nums, const float* dens,
                                     float* loss, float* term_losses) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = 0.f;
    int off = 0;
    for (int j = 0; j < tg.n; ++j) {
        float s = 0.f;
        for (int g = 0; g < tg.g[j]; ++g) s += nums[off + g] / dens[off + g];
        s /= (float)tg.g[j];
        if (term_losses) term_losses[j] = s;
        total += s;
        off += tg.g[j];
    }
    loss[0] = total;
}

extern "C" int d2p_loss_assemble(int n_terms, const int* groups, const float* nums,
                                 const float* dens, float* loss, float* term_losses,
                                 d2p_stream_t stream) {
    D2P_REQUIRE(n_terms > 0 && n_terms <= 8 && groups && nums && dens && loss, D2P_EINVAL,
                "loss_assemble: bad arguments (n_terms=%d)", n_terms);
    TermGroups tg;
    tg.n = n_terms;
    for (int i = 0; i < 8; ++i) tg.g[i] = i < n_terms ? groups[i] : 0;
    for (int i = 0; i < n_terms; ++i)
        D2P_REQUIRE(groups[i] > 0, D2P_EINVAL, "loss_assemble: groups[%d]=%d", i, groups[i]);
    hipLaunchKernelGGL(loss_assemble_kernel, dim3(1), dim3(64), 0, as_stream(stream), tg, nums, dens,
                       loss, term_losses);
    D2P_LAUNCH_CHECK("loss_assemble");
    return D2P_OK;
}

// The loss value from the per-workgroup sums d2p_xent_bwd_dhout_multi leaves behind (desc.loss_part): term j has G_j
// groups and nb_j workgroups; group sums in workgroup order (SUB strided partial sums per group, then those in order),
// then the assembly of loss_assemble_kernel.  One launch instead of three partial + three final + one assemble.
#define LFP_SUB 4
struct LfpArgs { int n; int g[3]; int nb[3]; const float* part[3]; };

__global__ void __launch_bounds__(256)
loss_from_partials_kernel(LfpArgs a, const float* dens, float* nums, float* loss, float* term_losses) {
    __shared__ float sub[64][LFP_SUB];
    __shared__ float num_s[64];
    const int tid = threadIdx.x, grp = tid / LFP_SUB, sb = tid % LFP_SUB;
    int j = 0, off = 0;
    while (j < a.n && grp >= off + a.g[j]) { off += a.g[j]; ++j; }
    if (j < a.n) {
        const int g = grp - off, G = a.g[j], nb = a.nb[j];
        const float* p = a.part[j] + g;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = sb;
        for (; b + 3 * LFP_SUB < nb; b += 4 * LFP_SUB) {
            s0 += p[(long)b * G]; s1 += p[(long)(b + LFP_SUB) * G]; s2 += p[(long)(b + 2 * LFP_SUB) * G];
            s3 += p[(long)(b + 3 * LFP_SUB) * G];
        }
        for (; b < nb; b += LFP_SUB) s0 += p[(long)b * G];
        sub[grp][sb] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (j < a.n && sb == 0) {
        float s = 0.f;
        for (int q = 0; q < LFP_SUB; ++q) s += sub[grp][q];
        num_s[grp] = s;
        if (nums) nums[grp] = s;
    }
    __syncthreads();
    if (tid == 0) {
        float total = 0.f;
        int o = 0;
        for (int t = 0; t < a.n; ++t) {
            float s = 0.f;
            for (int g = 0; g < a.g[t]; ++g) s += num_s[o + g] / dens[o + g];
            s /= (float)a.g[t];
            if (term_losses) term_losses[t] = s;
            total += s;
            o += a.g[t];
        }
        loss[0] = total;
    }
}

extern "C" int d2p_loss_from_partials(int n_terms, const int* groups, const int* nblocks, const float* const* parts,
                                      const float* dens, float* nums, float* loss, float* term_losses,
                                      d2p_stream_t stream) {
    D2P_REQUIRE(n_terms > 0 && n_terms <= 3 && groups && nblocks && parts && dens && loss, D2P_EINVAL,
                "loss_from_partials: bad arguments (n_terms=%d)", n_terms);
    LfpArgs a;
    a.n = n_terms;
    int total = 0;
    for (int i = 0; i < 3; ++i) {
        a.g[i] = i < n_terms ? groups[i] : 0;
        a.nb[i] = i < n_terms ? nblocks[i] : 0;
        a.part[i] = i < n_terms ? parts[i] : nullptr;
        if (i < n_terms) {
            D2P_REQUIRE(groups[i] > 0 && nblocks[i] >= 0 && (parts[i] || nblocks[i] == 0), D2P_EINVAL,
                        "loss_from_partials: term %d", i);
            total += groups[i];
        }
    }
    D2P_REQUIRE(total <= 64, D2P_EINVAL, "loss_from_partials: %d groups (at most 64)", total);
    hipLaunchKernelGGL(loss_from_partials_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, dens, nums, loss, term_losses);
    D2P_LAUNCH_CHECK("loss_from_partials");
    return D2P_OK;
}

// zero logits rows (t, r) with t >= min(T, max_{r' % G == r % G} lens[r'])
// (dynamic_decode stops at the longest sequence of that call; models/model_full.py:476-484)
__global__ void __launch_bounds__(256)
zero_past_group_steps_kernel(int T, int R, int V, int G, const int* lens, float* logits) {
    extern __shared__ int gmax[];
    for (int g = threadIdx.x; g < G; g += 256) {
        int m = 0;
        for (int r = g; r < R; r += G) m = max(m, lens[r]);
        gmax[g] = min(m, T);
    }
    __syncthreads();
    const long total = (long)T * R * V;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        const long row = idx / V;
        const int t = (int)(row / R), r = (int)(row - (long)t * R);
        if (t >= gmax[r % G]) logits[idx] = 0.f;
    }
}

extern "C" int d2p_zero_past_group_steps(int T, int R, int V, int G, const int* lens, float* logits,
                                         d2p_stream_t stream) {
    D2P_REQUIRE(T >= 0 && R >= 0 && V > 0 && G > 0 && G <= 4096, D2P_EINVAL, "zero_past: bad sizes");
    if (T == 0 || R == 0) return D2P_OK;
    D2P_REQUIRE(lens && logits, D2P_EINVAL, "zero_past: null pointer");
    long blocks = ((long)T * R * V + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(zero_past_group_steps_kernel, dim3((int)blocks), dim3(256), G * sizeof(int),
                       as_stream(stream), T, R, V, G, lens, logits);
    D2P_LAUNCH_CHECK("zero_past_group_steps");
    return D2P_OK;
}
