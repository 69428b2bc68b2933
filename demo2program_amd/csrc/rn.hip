// The relation networks' pointwise chains in four launches (round 5).
//
// rn_pool of the summarizer (models/model_full.py:333-349, both summaries rn_h / rn_c at once): per program b the k*k
// pairs (a, c) of demonstrations,
//     y1a[b,a,c] = lrelu(P[b,c] + Q[b,a] + bias1)        (fc1 on the concatenated pair, factorised: DESIGN 1)
//     y1 = batch norm(y1a)                                 (one group: all B*k*k rows)
//     y2a = lrelu(y1 . W2 + b2),  y2 = batch norm(y2a)     (the fc2 product stays a GEMM launch)
//     out[b] = mean over the pairs of y2[b] (+ mean over k of the features: the avg-pool branch)
// The chain around the two GEMMs was 9 launches forward (pair add, 2 x (partial sums, finalize, apply), feature mean,
// pair mean) and 11 backward (pair-mean backward, 2 x (partial sums, finalize, apply, column sums), pair backward), each
// 5-9 us, forward alone on the chip between the second encoder and the decoders.  Four launches now:
//   rn_fc1_fwd   y1a is a FUNCTION of 2k rows of P / Q per program, so the batch-norm sums are taken from recomputed values
//                (no pass over a materialised y1a), then y1a (backward needs it) and y1 are written once;
//   rn_fc2_fwd   batch norm is affine per column, so it commutes with the mean over a program's pairs:
//                out[b] = gamma (mean_pairs y2a[b] - mean) rstd + beta -- one read of y2a gives the batch-norm sums AND the
//                per-program sums; y2 is never written;
//   rn_fc2_bwd   the gradient of out is constant over a program's pairs, so the batch-norm backward's sums are closed
//                forms of d_out [B, U] and the per-program sums saved by rn_fc2_fwd: one pass y2a -> dpre2, no exchange;
//   rn_fc1_bwd   batch-norm backward of fc1 (these sums are real) and the pair backward: dP[b,c] = sum_a dpre1[b,a,c], dQ[b,a] =
//                sum_c dpre1[b,a,c] from registers -- dpre1 is never written.
// Workgroup = (summary, 128-column slice, program); the B workgroups of a (summary, slice) exchange their fp64 partial
// sums as the one-launch State_Encoder does (write-through partial sums, monotonic arrival tickets, every workgroup adds
// the B partial pairs in program order: the same bits everywhere), so all of a launch's workgroups must be co-resident:
// they are small (256 threads, < 32 KB of LDS) and the host checks the grid against the chip.  A workgroup that waits
// too long sets the status word of d2p_lstm_persist_error (code 0x7c).
#include "common.h"
#include "prof.h"

unsigned* d2p_persist_err_ptr();       // lstm_persist.hip

typedef int rn_i32x4 __attribute__((ext_vector_type(4)));

namespace {

#define RN_CS 128                  // columns per workgroup
#define RN_MAXG 64                 // (summary, slice) groups
#define RN_AUX_SC1 16
#define RN_MAXB 256
// monotonic tickets per [launch kind][programs B][group]: a generation is B arrivals, so launches with different B must
// not share a counter (as the one-launch State_Encoder's counters, one set per slice count)
__device__ unsigned long long g_rn_counters[4][RN_MAXB + 1][RN_MAXG];

struct RnGeom {
    int B, k, U, ncs;              // ncs = U / RN_CS
    int sc, cs, b;                 // this workgroup
    __device__ __forceinline__ void init(int B_, int k_, int U_) {
        B = B_; k = k_; U = U_; ncs = U_ / RN_CS;
        const int grp = blockIdx.x / B_;
        b = blockIdx.x - grp * B_;
        sc = grp / ncs;
        cs = grp - sc * ncs;
    }
    __device__ __forceinline__ int group() const { return sc * ncs + cs; }
};

// The exchange of a (summary, slice): threads ch < 128 hold this workgroup's (s0, s1); every thread ch < 128 leaves
// with the sums over the B workgroups in program order.  part: [group][B][128][2] doubles.
__device__ __forceinline__ void rn_exchange(double* part, unsigned long long* counters, unsigned* err, int group, int B, int b,
                                            double& s0, double& s1, int* flag, double* xred /* [2][128][2] doubles of LDS */) {
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(
        part + (long)group * B * RN_CS * 2, 0, B * RN_CS * 2 * (int)sizeof(double), 0x00020000);
    if (tid < RN_CS) {
        const double pr[2] = {s0, s1};
        rn_i32x4 pv;
        __builtin_memcpy(&pv, pr, 16);
        __builtin_amdgcn_raw_buffer_store_b128(pv, res, (b * RN_CS + tid) * 16, 0, RN_AUX_SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned long long* cnt = counters + group;
        const unsigned long long ticket = __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = (ticket / (unsigned)B + 1ull) * (unsigned)B;
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 800000u || ((spins & 1023u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(err, (0x7cu << 24) | 0x800000u | (blockIdx.x & 0xffffu), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *flag = ok;
    }
    __syncthreads();
    {
        // both thread halves read: half h the programs [h ceil(B/2), ...) of its column, sixteen 16-byte loads in flight
        // (four at a time by half the threads was eight memory round trips at B = 32); the two halves' sums meet in
        // LDS -- lower programs first: the same order in every workgroup
        const int ch = tid & (RN_CS - 1), half = tid >> 7, hb = (B + 1) >> 1;
        const int p0 = half * hb, p1 = p0 + hb < B ? p0 + hb : B;
        double t0 = 0.0, t1 = 0.0;
        for (int s = p0; s < p1; s += 16) {
            rn_i32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(res, s + u < p1 ? ((s + u) * RN_CS + ch) * 16 : 0x7fffff00, 0, RN_AUX_SC1);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                double pr[2];
                __builtin_memcpy(pr, &v[u], 16);
                t0 += pr[0];
                t1 += pr[1];
            }
        }
        xred[(half * RN_CS + ch) * 2] = t0;
        xred[(half * RN_CS + ch) * 2 + 1] = t1;
    }
    __syncthreads();
    if (tid < RN_CS) {
        s0 = xred[tid * 2] + xred[(RN_CS + tid) * 2];
        s1 = xred[tid * 2 + 1] + xred[(RN_CS + tid) * 2 + 1];
    }
}

// mean / rstd / var of (summary, column) from the sums, into LDS for both thread halves; workgroup b == 0 writes them out
// and applies the moving-average update (the arithmetic of bn_update_moving_kernel, one update: G = 1)
__device__ __forceinline__ void rn_stats(const RnGeom& g, int n, double s0, double s1, float* lmean, float* lrstd, float* mean,
                                         float* rstd, float* var, float* mm, float* mv, float decay, const unsigned* err) {
    const int tid = threadIdx.x;
    if (tid < RN_CS) {
        const double mu = s0 / n;
        double va = s1 / n - mu * mu;
        if (va < 0.0) va = 0.0;
        const float m = (float)mu, rs = (float)(1.0 / sqrt(va + 1e-3));
        lmean[tid] = m;
        lrstd[tid] = rs;
        if (g.b == 0) {
            const int u = g.sc * g.U + g.cs * RN_CS + tid;
            mean[u] = m;
            rstd[u] = rs;
            if (var) var[u] = (float)va;
            if (mm && *err == 0u) {
                mm[u] = decay * mm[u] + (1.f - decay) * m;
                mv[u] = decay * mv[u] + (1.f - decay) * (float)va;
            }
        }
    }
    __syncthreads();
}

struct RnFc1FwdArgs {
    int B, k, U;
    const float *P, *Q;            // [2, B*k, U] each
    const float *bias, *gamma, *beta; long pstride;
    float *y1a, *y1;               // [2, B*k*k, U]
    float *mean, *rstd, *var;      // [2, U]
    float *mm, *mv; float decay;   // moving statistics [2, U] (null: not tracked)
    double* part; unsigned long long* counters; unsigned* err;
};

__global__ void __launch_bounds__(256) rn_fc1_fwd_kernel(RnFc1FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    RnGeom g;
    g.init(a.B, a.k, a.U);
    const int tid = threadIdx.x, ch = tid & (RN_CS - 1), half = tid >> 7, k = a.k;
    const int u = g.cs * RN_CS + ch;
    float* Ps = lds;                           // [k][128]
    float* Qs = Ps + k * RN_CS;                // [k][128]
    double* red = reinterpret_cast<double*>(Qs + k * RN_CS);      // [2][128][2]
    float* lmean = reinterpret_cast<float*>(red + 4 * RN_CS);
    float* lrstd = lmean + RN_CS;
    int* flag = reinterpret_cast<int*>(lrstd + RN_CS);
    const long M = (long)a.B * k;
    for (int i = tid; i < k * RN_CS; i += 256) {
        const int r = i >> 7, c = i & (RN_CS - 1);
        const long src = ((long)g.sc * M + (long)g.b * k + r) * a.U + g.cs * RN_CS + c;
        Ps[i] = a.P[src];
        Qs[i] = a.Q[src];
    }
    const float bias = a.bias[g.sc * a.pstride + u];
    __syncthreads();
    const int a0 = half ? (k + 1) / 2 : 0, a1 = half ? k : (k + 1) / 2;
    double s0 = 0.0, s1 = 0.0;
    for (int aa = a0; aa < a1; ++aa) {
        const float qb = Qs[aa * RN_CS + ch] + bias;
        for (int c = 0; c < k; ++c) {
            const float v = d2p_lrelu(Ps[c * RN_CS + ch] + qb);
            s0 += (double)v;
            s1 += (double)v * (double)v;
        }
    }
    if (half) { red[ch * 2] = s0; red[ch * 2 + 1] = s1; }
    __syncthreads();
    if (!half) { s0 += red[ch * 2]; s1 += red[ch * 2 + 1]; }
    rn_exchange(a.part, a.counters, a.err, g.group(), a.B, g.b, s0, s1, flag, red);
    rn_stats(g, a.B * k * k, s0, s1, lmean, lrstd, a.mean, a.rstd, a.var, a.mm, a.mv, a.decay, a.err);
    const float mu = lmean[ch], rs = lrstd[ch];
    const float ga = a.gamma[g.sc * a.pstride + u], be = a.beta[g.sc * a.pstride + u];
    const long R = M * k;
    for (int aa = a0; aa < a1; ++aa) {
        const float qb = Qs[aa * RN_CS + ch] + bias;
        const long row0 = (long)g.sc * R + ((long)g.b * k + aa) * k;
        for (int c = 0; c < k; ++c) {
            const float v = d2p_lrelu(Ps[c * RN_CS + ch] + qb);
            a.y1a[(row0 + c) * a.U + u] = v;
            a.y1[(row0 + c) * a.U + u] = ga * (v - mu) * rs + be;
        }
    }
}

struct RnFc2FwdArgs {
    int B, k, U;
    const float* y2a;              // [2, B*k*k, U]
    const float *gamma, *beta; long pstride;
    const float* feat;             // [2, B*k, U] or null: + mean over k (the avg-pool branch)
    float *out, *psum;             // [2, B, U]: the summary; the per-program sums of y2a (backward)
    float *mean, *rstd, *var, *mm, *mv; float decay;
    double* part; unsigned long long* counters; unsigned* err;
};

__global__ void __launch_bounds__(256) rn_fc2_fwd_kernel(RnFc2FwdArgs a) {
    __shared__ double red[RN_CS * 4];
    __shared__ float lmean[RN_CS], lrstd[RN_CS];
    __shared__ int flag;
    RnGeom g;
    g.init(a.B, a.k, a.U);
    const int tid = threadIdx.x, ch = tid & (RN_CS - 1), half = tid >> 7, kk = a.k * a.k;
    const int u = g.cs * RN_CS + ch;
    const long R = (long)a.B * kk;
    const float* src = a.y2a + ((long)g.sc * R + (long)g.b * kk) * a.U + u;
    const int j0 = half ? (kk + 1) / 2 : 0, j1 = half ? kk : (kk + 1) / 2;
    double s0 = 0.0, s1 = 0.0;
    int j = j0;
    for (; j + 25 <= j1; j += 25) {           // (25 loads in flight: at one wave per SIMD a batch is a memory round trip)
        float v[25];
#pragma unroll
        for (int q = 0; q < 25; ++q) v[q] = src[(long)(j + q) * a.U];
#pragma unroll
        for (int q = 0; q < 25; ++q) { s0 += (double)v[q]; s1 += (double)v[q] * (double)v[q]; }
    }
    for (; j + 5 <= j1; j += 5) {
        float v[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) v[q] = src[(long)(j + q) * a.U];
#pragma unroll
        for (int q = 0; q < 5; ++q) { s0 += (double)v[q]; s1 += (double)v[q] * (double)v[q]; }
    }
    for (; j < j1; ++j) { const float v = src[(long)j * a.U]; s0 += (double)v; s1 += (double)v * (double)v; }
    if (half) { red[ch * 2] = s0; red[ch * 2 + 1] = s1; }
    __syncthreads();
    double own = 0.0;
    if (!half) { s0 += red[ch * 2]; s1 += red[ch * 2 + 1]; own = s0; }
    rn_exchange(a.part, a.counters, a.err, g.group(), a.B, g.b, s0, s1, &flag, red);
    rn_stats(g, a.B * kk, s0, s1, lmean, lrstd, a.mean, a.rstd, a.var, a.mm, a.mv, a.decay, a.err);
    if (!half) {
        const long o = ((long)g.sc * a.B + g.b) * a.U + u;
        const float pm = (float)(own / kk);
        float r = a.gamma[g.sc * a.pstride + u] * (pm - lmean[ch]) * lrstd[ch] + a.beta[g.sc * a.pstride + u];
        if (a.feat) {
            const float* f = a.feat + ((long)g.sc * a.B * a.k + (long)g.b * a.k) * a.U + u;
            float t = 0.f;
            for (int q = 0; q < a.k; ++q) t += f[(long)q * a.U];
            r += t / (float)a.k;
        }
        a.out[o] = r;
        a.psum[o] = (float)own;
    }
}

struct RnFc2BwdArgs {
    int B, k, U;
    const float *y2a, *dout, *psum;    // [2, B*k*k, U]; [2, B, U]; [2, B, U]
    const float* gamma; long pstride;
    const float *mean, *rstd;          // [2, U]
    float* dpre;                       // [2, B*k*k, U]: gradient of fc2's pre-activation
    float *dgamma, *dbeta;             // (+ sc * pstride)
    float* dbpart;                     // [2, B, U]: this program's column sums of dpre (folded by rn_fc1_bwd_kernel)
};

__global__ void __launch_bounds__(256) rn_fc2_bwd_kernel(RnFc2BwdArgs a) {
    __shared__ float red[RN_CS], cst[2][RN_CS];
    RnGeom g;
    g.init(a.B, a.k, a.U);
    const int tid = threadIdx.x, ch = tid & (RN_CS - 1), half = tid >> 7, kk = a.k * a.k;
    const int u = g.cs * RN_CS + ch;
    const long R = (long)a.B * kk;
    const float mu = a.mean[g.sc * a.U + u], rs = a.rstd[g.sc * a.U + u];
    if (!half) {
        // dy is d_out[b] / kk on every pair of program b: sum dy = sum_b d_out[b], sum dy x-hat = sum_b d_out[b] / kk *
        // (psum[b] - kk mean) rstd -- in program order, the same in every workgroup of the (summary, slice)
        double t1 = 0.0, t2 = 0.0;
        for (int b0 = 0; b0 < a.B; b0 += 8) {
            float dv[8], pv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const long o = ((long)g.sc * a.B + (b0 + q < a.B ? b0 + q : a.B - 1)) * a.U + u;
                dv[q] = a.dout[o];
                pv[q] = a.psum[o];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (b0 + q < a.B) {
                    const double d = (double)dv[q];
                    t1 += d;
                    t2 += d / kk * ((double)pv[q] - (double)kk * (double)mu) * (double)rs;
                }
            }
        }
        cst[0][ch] = (float)(t1 / (double)R);
        cst[1][ch] = (float)(t2 / (double)R);
        if (g.b == 0) {
            a.dgamma[g.sc * a.pstride + u] = (float)t2;
            a.dbeta[g.sc * a.pstride + u] = (float)t1;
        }
    }
    __syncthreads();
    const float m1 = cst[0][ch], m2 = cst[1][ch];
    const float k1 = a.gamma[g.sc * a.pstride + u] * rs;
    const float dy = a.dout[((long)g.sc * a.B + g.b) * a.U + u] / (float)kk;
    const long base = ((long)g.sc * R + (long)g.b * kk) * a.U + u;
    const int j0 = half ? (kk + 1) / 2 : 0, j1 = half ? kk : (kk + 1) / 2;
    float sum = 0.f;
    int j = j0;
    for (; j + 25 <= j1; j += 25) {
        float v[25];
#pragma unroll
        for (int q = 0; q < 25; ++q) v[q] = a.y2a[base + (long)(j + q) * a.U];
#pragma unroll
        for (int q = 0; q < 25; ++q) {
            const float xh = (v[q] - mu) * rs;
            float d = k1 * (dy - m1 - xh * m2);
            d *= d2p_lrelu_grad_from_out(v[q]);
            a.dpre[base + (long)(j + q) * a.U] = d;
            sum += d;
        }
    }
    for (; j < j1; ++j) {
        const float v = a.y2a[base + (long)j * a.U];
        const float xh = (v - mu) * rs;
        float d = k1 * (dy - m1 - xh * m2);
        d *= d2p_lrelu_grad_from_out(v);
        a.dpre[base + (long)j * a.U] = d;
        sum += d;
    }
    if (half) red[ch] = sum;
    __syncthreads();
    if (!half) a.dbpart[((long)g.sc * a.B + g.b) * a.U + u] = sum + red[ch];
}

struct RnFc1BwdArgs {
    int B, k, U;
    const float *y1a, *dy1;            // [2, B*k*k, U]
    const float* gamma; long pstride;
    const float *mean, *rstd;          // [2, U]
    float *dP, *dQ;                    // [2, B*k, U]
    float *dgamma, *dbeta, *dbias;     // fc1's (+ sc * pstride)
    const float* dbpart2; float* dbias2;   // rn_fc2_bwd_kernel's per-program column sums -> fc2's bias gradient
    float* dbpart;                     // [2, B, U] scratch: this launch's per-program column sums
    double* part; unsigned long long* counters; unsigned long long* tickets; unsigned* err;
};

// KT: k at compile time (0: run-time k <= 32).  With k known and small the thread's whole share of a pass -- ceil(k/2) rows of k
// pairs, two arrays: 100 loads at k = 10 -- is in flight at once; at one workgroup per CU (a wave per SIMD) every batch of
// loads is a full memory round trip, and the run-time form pays one per row of pairs.
template <int KT>
__global__ void __launch_bounds__(256) rn_fc1_bwd_kernel(RnFc1BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    RnGeom g;
    g.init(a.B, a.k, a.U);
    constexpr int KC = KT > 0 ? KT : 32;                    // pairs per row held in registers
    constexpr int NA = (KT > 0 && KT <= 12) ? (KT + 1) / 2 : 1;      // rows of pairs per batch
    const int tid = threadIdx.x, ch = tid & (RN_CS - 1), half = tid >> 7, k = KT > 0 ? KT : a.k, kk = k * k;
    const int u = g.cs * RN_CS + ch;
    double* red = reinterpret_cast<double*>(lds);                 // [2][128][2]
    float* cst = reinterpret_cast<float*>(red + 4 * RN_CS);       // [2][128]
    float* dPs = cst + 2 * RN_CS;                                 // [k][128]: the second half's share of dP
    int* flag = reinterpret_cast<int*>(dPs + k * RN_CS);
    const long R = (long)a.B * kk;
    const long base = ((long)g.sc * R + (long)g.b * kk) * a.U + u;
    const float mu = a.mean[g.sc * a.U + u], rs = a.rstd[g.sc * a.U + u];
    const int a0 = half ? (k + 1) / 2 : 0, a1 = half ? k : (k + 1) / 2;
    // pass 1: sum dy, sum dy x-hat over this program's pairs
    double s0 = 0.0, s1 = 0.0;
    for (int ab = a0; ab < a1; ab += NA) {
        float dv[NA][KC], yv[NA][KC];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c < k) {
                    const int aa = ab + i < a1 ? ab + i : a1 - 1;            // (past the half's rows: repeated, not added)
                    const long o = base + (long)(aa * k + c) * a.U;
                    dv[i][c] = a.dy1[o];
                    yv[i][c] = a.y1a[o];
                }
            }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c < k && ab + i < a1) {
                    const float xh = (yv[i][c] - mu) * rs;
                    s0 += (double)dv[i][c];
                    s1 += (double)dv[i][c] * (double)xh;
                }
            }
    }
    if (half) { red[ch * 2] = s0; red[ch * 2 + 1] = s1; }
    __syncthreads();
    if (!half) { s0 += red[ch * 2]; s1 += red[ch * 2 + 1]; }
    rn_exchange(a.part, a.counters, a.err, g.group(), a.B, g.b, s0, s1, flag, red);
    if (tid < RN_CS) {
        cst[ch] = (float)(s0 / (double)R);
        cst[RN_CS + ch] = (float)(s1 / (double)R);
        if (g.b == 0) {
            a.dgamma[g.sc * a.pstride + u] = (float)s1;
            a.dbeta[g.sc * a.pstride + u] = (float)s0;
            // fc2's bias gradient: the per-program column sums rn_fc2_bwd_kernel left, in program order
            float t = 0.f;
            for (int b0 = 0; b0 < a.B; b0 += 16) {
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = b0 + q < a.B ? a.dbpart2[((long)g.sc * a.B + b0 + q) * a.U + u] : 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) t += v[q];
            }
            a.dbias2[g.sc * a.pstride + u] = t;
        }
    }
    __syncthreads();
    // pass 2 (the rows are in L2 now): dpre1 = gamma rstd (dy - m1 - x-hat m2) lrelu'(y1a), summed over a into dP[c], over c
    // into dQ[a]; dpre1 itself is never written
    const float m1 = cst[ch], m2 = cst[RN_CS + ch], k1 = a.gamma[g.sc * a.pstride + u] * rs;
    const long M = (long)a.B * k;
    float tot = 0.f;
    // (dP accumulators over this half's a: in LDS columns of this thread -- k is a run-time value)
    float* myP = dPs + ch;                 // second half: its sums; first half adds them at the end
    float dpacc[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) dpacc[c] = 0.f;
    for (int ab = a0; ab < a1; ab += NA) {
        float dv[NA][KC], yv[NA][KC];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c < k) {
                    const int aa = ab + i < a1 ? ab + i : a1 - 1;
                    const long o = base + (long)(aa * k + c) * a.U;
                    dv[i][c] = a.dy1[o];
                    yv[i][c] = a.y1a[o];
                }
            }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (ab + i < a1) {
                float dq = 0.f;
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    if (c < k) {
                        const float v = yv[i][c];
                        const float xh = (v - mu) * rs;
                        float d = k1 * (dv[i][c] - m1 - xh * m2);
                        d *= d2p_lrelu_grad_from_out(v);
                        dq += d;
                        dpacc[c] += d;
                    }
                }
                a.dQ[((long)g.sc * M + (long)g.b * k + ab + i) * a.U + u] = dq;
                tot += dq;
            }
        }
    }
    if (half) {
#pragma unroll
        for (int c = 0; c < KC; ++c)
            if (c < k) myP[c * RN_CS] = dpacc[c];
        reinterpret_cast<float*>(red)[ch] = tot;
    }
    __syncthreads();
    if (!half) {
#pragma unroll
        for (int c = 0; c < KC; ++c)
            if (c < k) a.dP[((long)g.sc * M + (long)g.b * k + c) * a.U + u] = dpacc[c] + myP[c * RN_CS];
        // fc1's bias gradient = sum over all rows: per-program column sums (write-through, as the exchange's partial sums:
        // no fence -- a release fence writes back the XCD's whole L2), the LAST program to arrive adds them in order
        const __amdgpu_buffer_rsrc_t wres = __builtin_amdgcn_make_buffer_rsrc(
            a.dbpart + (long)g.sc * a.B * a.U, 0, a.B * a.U * (int)sizeof(float), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, tot + reinterpret_cast<float*>(red)[ch]), wres,
                                              (g.b * a.U + u) * 4, 0, RN_AUX_SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned long long t = __hip_atomic_fetch_add(a.tickets + g.group(), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = ((t + 1ull) % (unsigned)a.B == 0ull) ? 1 : 0;
    }
    __syncthreads();
    if (*flag && tid < RN_CS) {
        const __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(
            a.dbpart + (long)g.sc * a.B * a.U, 0, a.B * a.U * (int)sizeof(float), 0x00020000);
        float t = 0.f;
        for (int b0 = 0; b0 < a.B; b0 += 16) {        // sixteen loads in flight (one at a time: a memory round trip each)
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q)
                v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     res, b0 + q < a.B ? ((b0 + q) * a.U + u) * 4 : 0x7fffff00, 0, RN_AUX_SC1));
#pragma unroll
            for (int q = 0; q < 16; ++q) t += v[q];
        }
        a.dbias[g.sc * a.pstride + u] = t;
    }
}

bool rn_geom_ok(int B, int k, int U) {
    if (B < 1 || B > RN_MAXB || k < 1 || k > 32 || U < RN_CS || U % RN_CS != 0 || 2 * (U / RN_CS) > RN_MAXG) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return false;
    // the exchange needs every workgroup resident: at most two per CU are counted on, and only where two of the LARGEST
    // dynamic-LDS request of the four kernels (rn_fc1_fwd: 2 k 512 + 5136 bytes -- 37.9 KB at k = 32) fit a CU's LDS
    // beside each other (gfx950: 160 KB; a 64 KB part holds one such workgroup for k >= 27) -- ADVICE round 5
    int lds_cu = 0;
    if (hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || lds_cu <= 0)
        lds_cu = 64 * 1024;
    const long lds_wg = (long)(2 * k * RN_CS) * (long)sizeof(float) + 4 * RN_CS * (long)sizeof(double) +
                        2 * RN_CS * (long)sizeof(float) + 16;
    const long per_cu = lds_cu / lds_wg >= 2 ? 2 : (lds_cu / lds_wg >= 1 ? 1 : 0);
    return 2L * (U / RN_CS) * B <= per_cu * cus;
}
size_t rn_part_bytes(int B, int U) { return (size_t)2 * (U / RN_CS) * B * RN_CS * 2 * sizeof(double); }
unsigned long long* rn_counters(int kind, int B) {
    static unsigned long long* base = nullptr;
    if (!base && hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_rn_counters)) != hipSuccess) return nullptr;
    return base + ((size_t)kind * (RN_MAXB + 1) + B) * RN_MAXG;
}

}   // namespace

extern "C" size_t d2p_rn_ws_bytes(int B, int k, int U) {
    if (!rn_geom_ok(B, k, U)) return 0;
    // three exchanges' partial sums + two [2, B, U] column-sum scratch arrays
    return 3 * rn_part_bytes(B, U) + (size_t)2 * 2 * B * U * sizeof(float);
}

extern "C" int d2p_rn_fc1_fwd(int B, int k, int U, const float* P, const float* Q, const float* bias, const float* gamma,
                              const float* beta, long pstride, float* y1a, float* y1, float* mean, float* rstd, float* var,
                              float* moving_mean, float* moving_var, float decay, void* ws, size_t ws_bytes,
                              d2p_stream_t stream) {
    D2P_REQUIRE(rn_geom_ok(B, k, U), D2P_EINVAL, "rn fc1 fwd: B=%d k=%d U=%d not taken", B, k, U);
    D2P_REQUIRE(P && Q && bias && gamma && beta && y1a && y1 && mean && rstd, D2P_EINVAL, "rn fc1 fwd: null pointer");
    D2P_REQUIRE(ws && ws_bytes >= d2p_rn_ws_bytes(B, k, U) && ((uintptr_t)ws & 15) == 0, D2P_EWS, "rn fc1 fwd: workspace");
    RnFc1FwdArgs a{B, k, U, P, Q, bias, gamma, beta, pstride, y1a, y1, mean, rstd, var, moving_mean, moving_var, decay,
                   (double*)ws, rn_counters(0, B), d2p_persist_err_ptr()};
    D2P_REQUIRE(a.counters, D2P_EINVAL, "rn: counters");
    const size_t lds = (size_t)(2 * k * RN_CS) * sizeof(float) + 4 * RN_CS * sizeof(double) + 2 * RN_CS * sizeof(float) + 16;
    hipStream_t st = as_stream(stream);
    D2pProfScope prof(st, D2P_PROF_BN, 2.0 * 2.0 * B * k * k * U * sizeof(float));
    hipLaunchKernelGGL(rn_fc1_fwd_kernel, dim3(2 * (U / RN_CS) * B), dim3(256), lds, st, a);
    D2P_LAUNCH_CHECK("rn_fc1_fwd");
    return D2P_OK;
}

extern "C" int d2p_rn_fc2_fwd(int B, int k, int U, const float* y2a, const float* gamma, const float* beta, long pstride,
                              const float* feat, float* out, float* psum, float* mean, float* rstd, float* var,
                              float* moving_mean, float* moving_var, float decay, void* ws, size_t ws_bytes,
                              d2p_stream_t stream) {
    D2P_REQUIRE(rn_geom_ok(B, k, U), D2P_EINVAL, "rn fc2 fwd: B=%d k=%d U=%d not taken", B, k, U);
    D2P_REQUIRE(y2a && gamma && beta && out && psum && mean && rstd, D2P_EINVAL, "rn fc2 fwd: null pointer");
    D2P_REQUIRE(ws && ws_bytes >= d2p_rn_ws_bytes(B, k, U) && ((uintptr_t)ws & 15) == 0, D2P_EWS, "rn fc2 fwd: workspace");
    RnFc2FwdArgs a{B, k, U, y2a, gamma, beta, pstride, feat, out, psum, mean, rstd, var, moving_mean, moving_var, decay,
                   (double*)((char*)ws + rn_part_bytes(B, U)), rn_counters(1, B), d2p_persist_err_ptr()};
    D2P_REQUIRE(a.counters, D2P_EINVAL, "rn: counters");
    hipStream_t st = as_stream(stream);
    D2pProfScope prof(st, D2P_PROF_BN, 2.0 * B * k * k * U * sizeof(float));
    hipLaunchKernelGGL(rn_fc2_fwd_kernel, dim3(2 * (U / RN_CS) * B), dim3(256), 0, st, a);
    D2P_LAUNCH_CHECK("rn_fc2_fwd");
    return D2P_OK;
}

extern "C" int d2p_rn_fc2_bwd(int B, int k, int U, const float* y2a, const float* dout, const float* psum, const float* gamma,
                              long pstride, const float* mean, const float* rstd, float* dpre, float* dgamma, float* dbeta,
                              void* ws, size_t ws_bytes, d2p_stream_t stream) {
    D2P_REQUIRE(rn_geom_ok(B, k, U), D2P_EINVAL, "rn fc2 bwd: B=%d k=%d U=%d not taken", B, k, U);
    D2P_REQUIRE(y2a && dout && psum && gamma && mean && rstd && dpre && dgamma && dbeta, D2P_EINVAL, "rn fc2 bwd: null pointer");
    D2P_REQUIRE(ws && ws_bytes >= d2p_rn_ws_bytes(B, k, U), D2P_EWS, "rn fc2 bwd: workspace");
    float* dbpart = (float*)((char*)ws + 3 * rn_part_bytes(B, U));
    RnFc2BwdArgs a{B, k, U, y2a, dout, psum, gamma, pstride, mean, rstd, dpre, dgamma, dbeta, dbpart};
    hipStream_t st = as_stream(stream);
    D2pProfScope prof(st, D2P_PROF_BN, 2.0 * 2.0 * B * k * k * U * sizeof(float));
    hipLaunchKernelGGL(rn_fc2_bwd_kernel, dim3(2 * (U / RN_CS) * B), dim3(256), 0, st, a);
    D2P_LAUNCH_CHECK("rn_fc2_bwd");
    return D2P_OK;
}

extern "C" int d2p_rn_fc1_bwd(int B, int k, int U, const float* y1a, const float* dy1, const float* gamma, long pstride,
                              const float* mean, const float* rstd, float* dP, float* dQ, float* dgamma, float* dbeta,
                              float* dbias, float* dbias2, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    D2P_REQUIRE(rn_geom_ok(B, k, U), D2P_EINVAL, "rn fc1 bwd: B=%d k=%d U=%d not taken", B, k, U);
    D2P_REQUIRE(y1a && dy1 && gamma && mean && rstd && dP && dQ && dgamma && dbeta && dbias && dbias2, D2P_EINVAL,
                "rn fc1 bwd: null pointer");
    D2P_REQUIRE(ws && ws_bytes >= d2p_rn_ws_bytes(B, k, U) && ((uintptr_t)ws & 15) == 0, D2P_EWS, "rn fc1 bwd: workspace");
    float* dbpart2 = (float*)((char*)ws + 3 * rn_part_bytes(B, U));
    float* dbpart = dbpart2 + (size_t)2 * B * U;
    RnFc1BwdArgs a{B, k, U, y1a, dy1, gamma, pstride, mean, rstd, dP, dQ, dgamma, dbeta, dbias, dbpart2, dbias2, dbpart,
                   (double*)((char*)ws + 2 * rn_part_bytes(B, U)), rn_counters(2, B), rn_counters(3, B), d2p_persist_err_ptr()};
    D2P_REQUIRE(a.counters && a.tickets, D2P_EINVAL, "rn: counters");
    const size_t lds = 4 * RN_CS * sizeof(double) + (size_t)(2 * RN_CS + k * RN_CS) * sizeof(float) + 16;
    hipStream_t st = as_stream(stream);
    D2pProfScope prof(st, D2P_PROF_BN, 2.0 * 2.0 * B * k * k * U * sizeof(float));
    if (k == 10) hipLaunchKernelGGL(rn_fc1_bwd_kernel<10>, dim3(2 * (U / RN_CS) * B), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(rn_fc1_bwd_kernel<0>, dim3(2 * (U / RN_CS) * B), dim3(256), lds, st, a);
    D2P_LAUNCH_CHECK("rn_fc1_bwd");
    return D2P_OK;
}
