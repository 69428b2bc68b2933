// K2: training-mode batch norm over row groups, forward and backward (include/d2p.h).
// Replaces tf.contrib.layers.batch_norm(is_training=True) at models/ops.py:20-23.
//
// x is [R, C] row-major; group(r) = (r / inner) % G.  The reference runs the encoder once
// per demonstration index (models/model_full.py:373-379), so the batched product path
// needs one set of statistics per index: G = k, inner = rows per (program, demo).
// Reductions are two-stage and deterministic: (G x S) workgroups write fp64 partial sums,
// a finalize kernel folds them in a fixed order.  fp64 accumulation makes
// E[x^2] - E[x]^2 safe for the un-normalised 0..255 ViZDoom activations.
#include "common.h"
#include "prof.h"

#define BN_EPS 1e-3   // [TF-1.3] contrib.layers.batch_norm default epsilon

// status word of the persistent recurrent kernels: while it is set (a launch gave up a hand-off and the
// activations downstream of it are garbage) the moving statistics are left alone -- the guarded optimizer
// step skips its update under the same condition (adam.hip) and the trainer re-runs the step
unsigned* d2p_persist_err_ptr();

struct BnPlan {
    int lanes_c;    // threads along channels
    int row_lanes;  // threads along rows
    int S;          // row splits per group
};

// With C % 4 == 0 each thread owns 4 consecutive channels (one 16-byte load per row), so the
// "channel lanes" count is C/4; otherwise one channel per thread.
static int g_bn_wg_target = 1024;   // workgroups a partial-sum launch aims at (d2p_bn_set_fold bits 8..: experiment)
static inline BnPlan bn_plan(int R, int C, int G) {
    BnPlan p;
    const int cl = (C % 4 == 0) ? C / 4 : C;
    p.lanes_c = cl < 256 ? cl : 256;
    p.row_lanes = 256 / p.lanes_c;
    const int n = G > 0 ? R / G : 0;                      // rows per group
    int s = g_bn_wg_target / (G > 0 ? G : 1);
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    int cap = (n + p.row_lanes * 8 - 1) / (p.row_lanes * 8);
    if (cap < 1) cap = 1;
    if (s > cap) s = cap;
    p.S = s;
    return p;
}

// Several independent, equally shaped BN problems in one launch (grid.z = problem): every pointer argument of
// problem b is the first problem's plus b times a stride -- xs for the [R, C] inputs (x, dy), ys for the [R, C]
// outputs (y, dx), ps for per-channel parameters and their gradients (gamma, beta, dgamma, dbeta, the bias
// gradient), ms for the moving statistics, ss = G*C for mean / rstd / var, wsb BYTES for the workspace.
// All zero for a single problem.
struct BnBatch { long xs, ys, ps, ms, ss, wsb; };
#define BN_SHIFT(ptr, stride) ptr = (ptr) ? (ptr) + (long)blockIdx.z * (stride) : (ptr)
#define BN_SHIFT_WS(ptr, T) ptr = (T*)((char*)(ptr) + (long)blockIdx.z * bb.wsb)

static inline int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

// grid of the backward apply pass when it also produces column sums: thread count a multiple
// of the channel-vector count so that every thread keeps one column
static inline int bn_sum_blocks(long R, int C, int vec) {
    const int CV = C / vec;
    const int unit = CV / gcd_i(CV, 256);              // blocks must be a multiple of this
    long want = (R * CV + 255) / 256;
    if (want > 1024) want = 1024;
    long b = (want + unit - 1) / unit * unit;
    return (int)(b < unit ? unit : b);
}

extern "C" size_t d2p_bn_ws_bytes(int R, int C, int G) {
    if (R <= 0 || C <= 0 || G <= 0) return 0;
    BnPlan p = bn_plan(R, C, G);
    const int vec = (C % 4 == 0) ? 4 : 1;
    const size_t stats = (size_t)G * p.S * C * 2 * sizeof(double) + (size_t)G * C * 2 * sizeof(double) +
                         align_up((size_t)G * C * 2 * sizeof(float), 16);
    return stats + (size_t)bn_sum_blocks(R, C, vec) * C * sizeof(float);
}

// ---- finalize folded into the partial-sum launch (round 3) ------------------------------------------------------
// The separate finalize kernels ran ONE wavefront per channel over all groups: 9-11 us each for the conv layers (16-48
// channels, 10 groups) on the step's critical path, six per step.  With few partials per group (S * C small) the LAST
// workgroup of a group to arrive (a ticket, cdna_hip_programming.md Guideline 16: stores -> vmcnt(0) -> barrier ->
// agent release -> relaxed fetch_add; the last arriver acquires) folds that group's S partials in s order -- the
// same fp64 sums in a fixed order, deterministic -- and the last GROUP to finish (a second ticket) does what needs
// every group: the moving-average updates in group order (forward), dgamma / dbeta (backward).
// Tickets live in a small device-global pool, one slot per call in flight (round-robin), zero at load and reset by
// each last arriver.
#define BN_TICKET_SLOTS 64
#define BN_TICKET_WORDS 64
__device__ unsigned g_bn_tickets[BN_TICKET_SLOTS * BN_TICKET_WORDS];
static unsigned* bn_ticket_slot() {
    static unsigned* base = nullptr;
    static unsigned next = 0;
    if (!base) (void)hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_bn_tickets));
    return base ? base + (size_t)(next++ % BN_TICKET_SLOTS) * BN_TICKET_WORDS : nullptr;
}
static int g_bn_gc = 1;          // one wavefront per (group, channel) in the finalize launches (bit 1 of d2p_bn_set_fold: off)
static int g_bn_fold = 0;        // 1: fold the finalize steps (d2p_bn_set_fold); measured equal-to-slower, off by default
extern "C" int d2p_bn_set_fold(int on) {
    g_bn_fold = (on & 1) ? 1 : 0;
    g_bn_gc = (on & 2) ? 0 : 1;
    g_bn_wg_target = (on >> 8) > 0 ? (on >> 8) : 1024;
    return D2P_OK;
}
static inline bool bn_fold_ok(int nb, int G, int S, int C) {
    return g_bn_fold && (long)S * C <= 4096 && nb * (G + 1) <= BN_TICKET_WORDS;
}

struct BnFold {
    unsigned* tickets;          // null: no fold (a finalize launch follows)
    // forward
    float* mean; float* rstd; float* var_out; float* moving_mean; float* moving_var; float decay;
    const unsigned* err;
    // backward
    float* m12; float* dgamma; float* dbeta; double* gsum;
};

// Ticket of this workgroup for counter `t` out of `total` arrivals; true for the last arriver (whole workgroup
// agrees).  `flag` is a spare LDS word of the caller's one shared array.
// Everything a later arriver reads travels as WRITE-THROUGH (sc1) stores and is read with sc1 loads (bn_st / bn_ld:
// relaxed agent-scope atomics of <= 8 bytes lower to exactly that), so no release / acquire fence is needed -- a
// fence here writes back the whole L2 of the XCD, once per workgroup, right after a conv kernel left megabytes of
// dirty activations in it (measured: the fenced form made the step 0.05 ms SLOWER than the separate launches).
template <typename T>
__device__ __forceinline__ void bn_st(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T bn_ld(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool bn_last_arriver(unsigned* t, unsigned total, double* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned k = __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = (k == total - 1u);
        if (last) __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the slot's next call
        *flag = last ? 1.0 : 0.0;
    }
    __syncthreads();
    return *flag != 0.0;
}

// partial[((g*S + s)*C + c)*2 + {0,1}] = sum over this block's rows of (a, b) where
//   MODE 0 (fwd):  a = x,   b = x*x
//   MODE 1 (bwd):  a = dy,  b = dy * xhat
// VEC = 4: thread owns channels 4*cl .. 4*cl+3 (16-byte loads); VEC = 1: one channel.
template <int MODE, int VEC>
__global__ void __launch_bounds__(256)
bn_partial_kernel(int n, int C, int G, int inner, int lanes_c, int row_lanes, const float* x,
                  const float* dy, const float* mean, const float* rstd, double* partial, BnBatch bb, BnFold fo) {
    BN_SHIFT(x, bb.xs); BN_SHIFT(dy, bb.xs); BN_SHIFT(mean, bb.ss); BN_SHIFT(rstd, bb.ss); BN_SHIFT_WS(partial, double);
    __shared__ double red[2][VEC][256 + 1];
    const int g = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const int tid = threadIdx.x;
    const int cl = tid % lanes_c, rl = tid / lanes_c;
    const bool active = rl < row_lanes;
    const int CL = C / VEC;                       // channel lanes in total
    for (int c0 = 0; c0 < CL; c0 += lanes_c) {
        const int c = (c0 + cl) * VEC;
        double a[VEC], b[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = b[e] = 0.0;
        if (active && c < C) {
            float mu[VEC], rs[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                mu[e] = MODE == 1 ? mean[g * C + c + e] : 0.f;
                rs[e] = MODE == 1 ? rstd[g * C + c + e] : 0.f;
            }
            // rows in batches of four: the four loads are independent and in flight together (one load per trip
            // left every row's ~1 us of latency exposed: 8 trips = most of this kernel's 8-10 us)
            const int step = S * row_lanes;
            for (int j0 = s * row_lanes + rl; j0 < n; j0 += 4 * step) {
                float xv[4][VEC], dv[4][VEC];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u * step;
                    ok[u] = j < n;
                    const int jc = ok[u] ? j : j0;
                    const int o = jc / inner, ii = jc - o * inner;
                    const long r = ((long)o * G + g) * inner + ii;
                    if (VEC == 4) {
                        const float4 t = *reinterpret_cast<const float4*>(x + r * C + c);
                        xv[u][0] = t.x; xv[u][1 % VEC] = t.y; xv[u][2 % VEC] = t.z; xv[u][3 % VEC] = t.w;
                        if (MODE == 1) {
                            const float4 d = *reinterpret_cast<const float4*>(dy + r * C + c);
                            dv[u][0] = d.x; dv[u][1 % VEC] = d.y; dv[u][2 % VEC] = d.z; dv[u][3 % VEC] = d.w;
                        }
                    } else {
                        xv[u][0] = x[r * C + c];
                        if (MODE == 1) dv[u][0] = dy[r * C + c];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!ok[u]) continue;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        if (MODE == 0) {
                            a[e] += (double)xv[u][e];
                            b[e] += (double)xv[u][e] * (double)xv[u][e];
                        } else {
                            a[e] += (double)dv[u][e];
                            b[e] += (double)dv[u][e] * (double)((xv[u][e] - mu[e]) * rs[e]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            red[0][e][tid] = a[e];
            red[1][e][tid] = b[e];
        }
        __syncthreads();
        // block reduction over the row lanes in a fixed order; with many row lanes (narrow layers: 64 of them for 16
        // channels) in two levels -- eight lanes per channel lane sum every eighth partial, then one sums the eight --
        // instead of one thread walking 64 dependent LDS reads
        const int RG = row_lanes >= 16 ? 8 : 1;
        double aa[VEC], bb[VEC];
        if (rl < RG && c < C) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                aa[e] = 0.0; bb[e] = 0.0;
                for (int q = rl; q < row_lanes; q += RG) {
                    aa[e] += red[0][e][q * lanes_c + cl];
                    bb[e] += red[1][e][q * lanes_c + cl];
                }
            }
        }
        if (RG > 1) {
            __syncthreads();
            if (rl < RG && c < C) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    red[0][e][rl * lanes_c + cl] = aa[e];
                    red[1][e][rl * lanes_c + cl] = bb[e];
                }
            }
            __syncthreads();
            if (rl == 0 && c < C) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    aa[e] = 0.0; bb[e] = 0.0;
                    for (int q = 0; q < RG; ++q) {
                        aa[e] += red[0][e][q * lanes_c + cl];
                        bb[e] += red[1][e][q * lanes_c + cl];
                    }
                }
            }
        }
        if (rl == 0 && c < C) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                double* out = partial + (((long)g * S + s) * C + c + e) * 2;
                if (fo.tickets) {
                    bn_st(out, aa[e]);
                    bn_st(out + 1, bb[e]);
                } else {
                    out[0] = aa[e];
                    out[1] = bb[e];
                }
            }
        }
        __syncthreads();
    }
    if (!fo.tickets) return;
    // ---- folded finalize: last workgroup of group g, then last group of the problem
    unsigned* tk = fo.tickets + (long)blockIdx.z * (G + 1);
    double* flag = &red[0][0][256];
    if (!bn_last_arriver(tk + g, (unsigned)S, flag)) return;
    if (MODE == 0) {
        float* mean_o = fo.mean + (long)blockIdx.z * bb.ss;
        float* rstd_o = fo.rstd + (long)blockIdx.z * bb.ss;
        float* var_o = fo.var_out ? fo.var_out + (long)blockIdx.z * bb.ss : nullptr;
        double* gs = fo.gsum + (long)blockIdx.z * (bb.wsb / (long)sizeof(double));       // [G][C][2]: (mean, var) in fp64
        for (int c = tid; c < C; c += 256) {
            double a = 0.0, b = 0.0;
            for (int q = 0; q < S; ++q) {
                const double* p = partial + (((long)g * S + q) * C + c) * 2;
                a += bn_ld(p);
                b += bn_ld(p + 1);
            }
            const double mu = a / n;
            double var = b / n - mu * mu;   // biased variance
            if (var < 0.0) var = 0.0;
            mean_o[g * C + c] = (float)mu;
            rstd_o[g * C + c] = (float)(1.0 / sqrt(var + BN_EPS));
            if (var_o) var_o[g * C + c] = (float)var;
            bn_st(gs + ((long)g * C + c) * 2 + 0, mu);
            bn_st(gs + ((long)g * C + c) * 2 + 1, var);
        }
        if (!fo.moving_mean) return;
        if (!bn_last_arriver(tk + G, (unsigned)G, flag)) return;
        // one moving-average update per group, in group order (same arithmetic as bn_update_moving_kernel)
        if (*fo.err != 0u) return;
        float* mmp = fo.moving_mean + (long)blockIdx.z * bb.ms;
        float* mvp = fo.moving_var + (long)blockIdx.z * bb.ms;
        for (int c = tid; c < C; c += 256) {
            float mm = mmp[c], mv = mvp[c];
            for (int q = 0; q < G; ++q) {
                mm = fo.decay * mm + (1.f - fo.decay) * (float)bn_ld(gs + ((long)q * C + c) * 2 + 0);
                mv = fo.decay * mv + (1.f - fo.decay) * (float)bn_ld(gs + ((long)q * C + c) * 2 + 1);
            }
            mmp[c] = mm;
            mvp[c] = mv;
        }
    } else {
        float* m12 = (float*)((char*)fo.m12 + (long)blockIdx.z * bb.wsb);
        double* gs = (double*)((char*)fo.gsum + (long)blockIdx.z * bb.wsb);               // [G][C][2]: the group sums
        for (int c = tid; c < C; c += 256) {
            double a = 0.0, b = 0.0;
            for (int q = 0; q < S; ++q) {
                const double* p = partial + (((long)g * S + q) * C + c) * 2;
                a += bn_ld(p);
                b += bn_ld(p + 1);
            }
            m12[((long)g * C + c) * 2 + 0] = (float)(a / n);
            m12[((long)g * C + c) * 2 + 1] = (float)(b / n);
            bn_st(gs + ((long)g * C + c) * 2 + 0, a);
            bn_st(gs + ((long)g * C + c) * 2 + 1, b);
        }
        if (!bn_last_arriver(tk + G, (unsigned)G, flag)) return;
        float* dgp = fo.dgamma ? fo.dgamma + (long)blockIdx.z * bb.ps : nullptr;
        float* dbp = fo.dbeta ? fo.dbeta + (long)blockIdx.z * bb.ps : nullptr;
        for (int c = tid; c < C; c += 256) {
            double sb = 0.0, sg = 0.0;
            for (int q = 0; q < G; ++q) {
                sb += bn_ld(gs + ((long)q * C + c) * 2 + 0);
                sg += bn_ld(gs + ((long)q * C + c) * 2 + 1);
            }
            if (dgp) dgp[c] = (float)sg;
            if (dbp) dbp[c] = (float)sb;
        }
    }
}

// One wavefront per CHANNEL, looping over the groups in group order: lanes stride over the S
// partials of a (group, channel), wave-reduce in fp64.  Doing all groups of a channel in one wave
// lets the same launch finish what needs them in order: the moving-average updates (one per
// group = one per reference BN call, models/model_full.py:373-379) in the forward kernel, and
// dgamma / dbeta (sums over the groups) in the backward kernel.
__global__ void __launch_bounds__(256)
bn_finalize_fwd_kernel(int n, int C, int G, int S, const double* partial, float* mean, float* rstd,
                       float* var_out, float* moving_mean, float* moving_var, float decay, const unsigned* err,
                       BnBatch bb) {
    BN_SHIFT_WS(partial, const double); BN_SHIFT(mean, bb.ss); BN_SHIFT(rstd, bb.ss); BN_SHIFT(var_out, bb.ss);
    BN_SHIFT(moving_mean, bb.ms); BN_SHIFT(moving_var, bb.ms);
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    float mm = moving_mean ? moving_mean[c] : 0.f, mv = moving_var ? moving_var[c] : 0.f;
    for (int g = 0; g < G; ++g) {
        double a = 0.0, b = 0.0;
        for (int s = lane; s < S; s += 64) {
            const double* p = partial + (((long)g * S + s) * C + c) * 2;
            a += p[0];
            b += p[1];
        }
        a = wave_reduce_sum(a);
        b = wave_reduce_sum(b);
        const double mu = a / n;
        double var = b / n - mu * mu;   // biased variance
        if (var < 0.0) var = 0.0;
        if (lane == 0) {
            const int idx = g * C + c;
            mean[idx] = (float)mu;
            rstd[idx] = (float)(1.0 / sqrt(var + BN_EPS));
            if (var_out) var_out[idx] = (float)var;
        }
        // one moving-average update per group, in group order (same arithmetic as bn_update_moving_kernel)
        mm = decay * mm + (1.f - decay) * (float)mu;
        mv = decay * mv + (1.f - decay) * (float)var;
    }
    if (lane == 0 && moving_mean && *err == 0u) {
        moving_mean[c] = mm;
        moving_var[c] = mv;
    }
}

// m12[(g*C+c)*2 + {0,1}] = (mean_g(dy), mean_g(dy*xhat)); dgamma / dbeta = sums over the groups
__global__ void __launch_bounds__(256)
bn_finalize_bwd_kernel(int n, int C, int G, int S, const double* partial, float* m12, float* dgamma,
                       float* dbeta, BnBatch bb) {
    BN_SHIFT_WS(partial, const double); BN_SHIFT_WS(m12, float); BN_SHIFT(dgamma, bb.ps); BN_SHIFT(dbeta, bb.ps);
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double sb = 0.0, sg = 0.0;
    for (int g = 0; g < G; ++g) {
        double a = 0.0, b = 0.0;
        for (int s = lane; s < S; s += 64) {
            const double* p = partial + (((long)g * S + s) * C + c) * 2;
            a += p[0];
            b += p[1];
        }
        a = wave_reduce_sum(a);
        b = wave_reduce_sum(b);
        if (lane == 0) {
            m12[((long)g * C + c) * 2 + 0] = (float)(a / n);
            m12[((long)g * C + c) * 2 + 1] = (float)(b / n);
        }
        sb += a;
        sg += b;
    }
    if (lane == 0) {
        if (dgamma) dgamma[c] = (float)sg;
        if (dbeta) dbeta[c] = (float)sb;
    }
}

// Round 3: one wavefront per (group, channel) instead of one per channel walking the groups in turn (10 dependent
// rounds of loads + two fp64 wave reductions: 9-11 us for the conv layers, six of them per step on the critical
// path).  What needs the groups in order -- the moving-average updates -- moves to one extra workgroup of the
// apply launch, which reads the (mean, var) pairs this kernel leaves in fp64 at gs[(g*C + c)*2].
__global__ void __launch_bounds__(256)
bn_finalize_fwd_gc_kernel(int n, int C, int G, int S, const double* partial, float* mean, float* rstd,
                          float* var_out, double* gs, BnBatch bb) {
    BN_SHIFT_WS(partial, const double); BN_SHIFT(mean, bb.ss); BN_SHIFT(rstd, bb.ss); BN_SHIFT(var_out, bb.ss);
    BN_SHIFT_WS(gs, double);
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= G * C) return;
    const int g = idx / C, c = idx - g * C;
    double a = 0.0, b = 0.0;
    for (int s = lane; s < S; s += 64) {
        const double* p = partial + (((long)g * S + s) * C + c) * 2;
        a += p[0];
        b += p[1];
    }
    a = wave_reduce_sum(a);
    b = wave_reduce_sum(b);
    const double mu = a / n;
    double var = b / n - mu * mu;   // biased variance
    if (var < 0.0) var = 0.0;
    if (lane == 0) {
        mean[idx] = (float)mu;
        rstd[idx] = (float)(1.0 / sqrt(var + BN_EPS));
        if (var_out) var_out[idx] = (float)var;
        gs[(long)idx * 2 + 0] = mu;
        gs[(long)idx * 2 + 1] = var;
    }
}
// the extra workgroup of the apply launch: one moving-average update per group, in group order
__device__ __forceinline__ void bn_moving_update(int C, int G, const double* gs, float* moving_mean, float* moving_var,
                                                 float decay, const unsigned* err) {
    if (*err != 0u) return;
    for (int c = threadIdx.x; c < C; c += 256) {
        float mm = moving_mean[c], mv = moving_var[c];
        for (int g = 0; g < G; ++g) {
            mm = decay * mm + (1.f - decay) * (float)gs[((long)g * C + c) * 2 + 0];
            mv = decay * mv + (1.f - decay) * (float)gs[((long)g * C + c) * 2 + 1];
        }
        moving_mean[c] = mm;
        moving_var[c] = mv;
    }
}
struct BnMoving {        // null moving_mean: no update (the extra workgroup is not launched)
    const double* gs; float* moving_mean; float* moving_var; float decay; const unsigned* err; int nblk;
};
// backward: m12 per (group, channel) + the group sums in fp64 at gs; dgamma / dbeta = sums over the groups are
// taken by the column-sum finalize launch that follows the apply pass
__global__ void __launch_bounds__(256)
bn_finalize_bwd_gc_kernel(int n, int C, int G, int S, const double* partial, float* m12, double* gs, BnBatch bb) {
    BN_SHIFT_WS(partial, const double); BN_SHIFT_WS(m12, float); BN_SHIFT_WS(gs, double);
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= G * C) return;
    const int g = idx / C, c = idx - g * C;
    double a = 0.0, b = 0.0;
    for (int s = lane; s < S; s += 64) {
        const double* p = partial + (((long)g * S + s) * C + c) * 2;
        a += p[0];
        b += p[1];
    }
    a = wave_reduce_sum(a);
    b = wave_reduce_sum(b);
    if (lane == 0) {
        m12[(long)idx * 2 + 0] = (float)(a / n);
        m12[(long)idx * 2 + 1] = (float)(b / n);
        gs[(long)idx * 2 + 0] = a;
        gs[(long)idx * 2 + 1] = b;
    }
}

__global__ void __launch_bounds__(256)
bn_apply_fwd_kernel(long R, int C, int G, int inner, const float* x, const float* gamma,
                    const float* beta, const float* mean, const float* rstd, float* y, BnBatch bb, BnMoving mo) {
    if ((int)blockIdx.x == mo.nblk) {
        bn_moving_update(C, G, (const double*)((const char*)mo.gs + (long)blockIdx.z * bb.wsb),
                         mo.moving_mean + (long)blockIdx.z * bb.ms, mo.moving_var + (long)blockIdx.z * bb.ms, mo.decay, mo.err);
        return;
    }
    BN_SHIFT(x, bb.xs); BN_SHIFT(gamma, bb.ps); BN_SHIFT(beta, bb.ps); BN_SHIFT(mean, bb.ss); BN_SHIFT(rstd, bb.ss);
    BN_SHIFT(y, bb.ys);
    const long total = R * C;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)mo.nblk * 256L) {
        const long r = idx / C;
        const int c = (int)(idx - r * C);
        const int g = (int)((r / inner) % G);
        y[idx] = gamma[c] * (x[idx] - mean[g * C + c]) * rstd[g * C + c] + beta[c];
    }
}

__global__ void __launch_bounds__(256)
bn_apply_fwd_vec4_kernel(long R, int C, int G, int inner, const float* x, const float* gamma,
                         const float* beta, const float* mean, const float* rstd, float* y, BnBatch bb, BnMoving mo) {
    if ((int)blockIdx.x == mo.nblk) {
        bn_moving_update(C, G, (const double*)((const char*)mo.gs + (long)blockIdx.z * bb.wsb),
                         mo.moving_mean + (long)blockIdx.z * bb.ms, mo.moving_var + (long)blockIdx.z * bb.ms, mo.decay, mo.err);
        return;
    }
    BN_SHIFT(x, bb.xs); BN_SHIFT(gamma, bb.ps); BN_SHIFT(beta, bb.ps); BN_SHIFT(mean, bb.ss); BN_SHIFT(rstd, bb.ss);
    BN_SHIFT(y, bb.ys);
    const int C4 = C >> 2;
    const long total = R * C4;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)mo.nblk * 256L) {
        const long r = idx / C4;
        const int c = (int)(idx - r * C4) * 4;
        const int g = (int)((r / inner) % G);
        const float4 xv = *reinterpret_cast<const float4*>(x + r * C + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(beta + c);
        const float4 mu = *reinterpret_cast<const float4*>(mean + g * C + c);
        const float4 rs = *reinterpret_cast<const float4*>(rstd + g * C + c);
        float4 o;
        o.x = ga.x * (xv.x - mu.x) * rs.x + be.x;
        o.y = ga.y * (xv.y - mu.y) * rs.y + be.y;
        o.z = ga.z * (xv.z - mu.z) * rs.z + be.z;
        o.w = ga.w * (xv.w - mu.w) * rs.w + be.w;
        *reinterpret_cast<float4*>(y + r * C + c) = o;
    }
}

// dx = gamma * rstd * (dy - m1 - xhat * m2) [* lrelu'(x)].  VEC consecutive channels per thread;
// no division in the loop (row / channel-vector advance incrementally).  With SUM the grid is
// sized so that a thread keeps ONE channel vector for the whole grid-stride loop and leaves the
// column sum of its dx values in colpart[thread][VEC] -- that is the bias gradient of the layer
// under the BN (conv / fc bias sits before lrelu + BN), which otherwise costs one more full
// read of dx by a separate column-sum pass.
template <int VEC, bool SUM>
__global__ void __launch_bounds__(256)
bn_apply_bwd_kernel(int R, int C, int G, int inner, const float* __restrict__ x,
                    const float* __restrict__ dy, const float* __restrict__ gamma,
                    const float* __restrict__ mean, const float* __restrict__ rstd,
                    const float* __restrict__ m12, int act_bwd, float* __restrict__ dx,
                    float* __restrict__ colpart, BnBatch bb, unsigned* sum_ticket, float* __restrict__ colsum_out) {
    BN_SHIFT(x, bb.xs); BN_SHIFT(dy, bb.xs); BN_SHIFT(gamma, bb.ps); BN_SHIFT(mean, bb.ss); BN_SHIFT(rstd, bb.ss);
    BN_SHIFT_WS(m12, const float); BN_SHIFT(dx, bb.ys);
    if (colpart) BN_SHIFT_WS(colpart, float);
    const int CV = C / VEC;
    const long tid = blockIdx.x * 256L + threadIdx.x;
    const long stride = gridDim.x * 256L;
    const int dr = (int)(stride / CV), dc = (int)(stride % CV);   // dc == 0 when SUM
    int r = (int)(tid / CV), cv = (int)(tid % CV);
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    float gm[VEC];
    if (SUM) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) gm[j] = gamma[cv * VEC + j];
    }
    while (r < R) {
        const int g = (r / inner) % G;
        const long e = (long)r * C + cv * VEC;
        float xv[VEC], dv[VEC], o[VEC];
        if (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(x + e), d = *reinterpret_cast<const float4*>(dy + e);
            xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
            dv[0] = d.x; dv[1] = d.y; dv[2] = d.z; dv[3] = d.w;
        } else {
            xv[0] = x[e]; dv[0] = dy[e];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = cv * VEC + j;
            const float rs = rstd[g * C + c];
            const float xhat = (xv[j] - mean[g * C + c]) * rs;
            const float m1 = m12[((long)g * C + c) * 2], m2 = m12[((long)g * C + c) * 2 + 1];
            float d = (SUM ? gm[j] : gamma[c]) * rs * (dv[j] - m1 - xhat * m2);
            if (act_bwd) d *= d2p_lrelu_grad_from_out(xv[j]);
            o[j] = d;
            s[j] += d;
        }
        if (VEC == 4) *reinterpret_cast<float4*>(dx + e) = make_float4(o[0], o[1], o[2], o[3]);
        else dx[e] = o[0];
        r += dr; cv += dc;
        if (cv >= CV) { cv -= CV; ++r; }
    }
    if (SUM) {
        // block-level column sums in thread order, then one row of C floats per block
        __shared__ float red[256][VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) red[threadIdx.x][j] = s[j];
        __syncthreads();
        const int base = (int)((blockIdx.x * 256L) % CV);
        for (int c = threadIdx.x; c < C; c += 256) {
            const int cvv = c / VEC, j = c % VEC;
            float t = 0.f;
            for (int tt = (cvv - base + CV) % CV; tt < 256; tt += CV) t += red[tt][j];
            if (sum_ticket) bn_st(colpart + (long)blockIdx.x * C + c, t);
            else colpart[(long)blockIdx.x * C + c] = t;
        }
        if (!sum_ticket) return;
        // folded column-sum finalize (few blocks x channels): the last workgroup adds the per-block rows in block
        // order -- thread (c, j) takes blocks j, j + P, ..., the P strided sums of a channel meet in LDS, fixed order
        __shared__ double red2[256 + 1];
        if (!bn_last_arriver(sum_ticket + blockIdx.z, gridDim.x, &red2[256])) return;
        float* out = colsum_out + (long)blockIdx.z * bb.ps;
        int Cp = 1;
        while (Cp < C) Cp <<= 1;                          // C <= 256 (host checks)
        const int P = 256 / Cp;
        const int c = threadIdx.x % Cp, j = threadIdx.x / Cp;
        double acc = 0.0;
        if (c < C)
            for (int b = j; b < (int)gridDim.x; b += P) acc += (double)bn_ld(colpart + (long)b * C + c);
        red2[threadIdx.x] = acc;
        __syncthreads();
        if (j == 0 && c < C) {
            double t = 0.0;
            for (int q = 0; q < P; ++q) t += red2[q * Cp + c];
            out[c] = (float)t;
        }
    }
}

// out[c] = sum over blocks of colpart[block][c] (fixed order: deterministic)
__global__ void __launch_bounds__(256)
bn_colsum_finalize_kernel(int C, int nblocks, const float* __restrict__ colpart, float* __restrict__ out, BnBatch bb,
                          int G, const double* gs, float* dgamma, float* dbeta) {
    BN_SHIFT_WS(colpart, const float); BN_SHIFT(out, bb.ps);
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double acc = 0.0;
    for (int b = lane; b < nblocks; b += 64) acc += (double)colpart[(long)b * C + c];
    acc = wave_reduce_sum(acc);
    if (lane == 0) out[c] = (float)acc;
    if (gs && lane == 0) {            // dgamma / dbeta: the group sums of bn_finalize_bwd_gc_kernel, in group order
        BN_SHIFT_WS(gs, const double);
        double sb = 0.0, sg = 0.0;
        for (int g = 0; g < G; ++g) {
            sb += gs[((long)g * C + c) * 2 + 0];
            sg += gs[((long)g * C + c) * 2 + 1];
        }
        if (dgamma) (dgamma + (long)blockIdx.z * bb.ps)[c] = (float)sg;
        if (dbeta) (dbeta + (long)blockIdx.z * bb.ps)[c] = (float)sb;
    }
}

static int bn_check(int R, int C, int G, int inner) {
    D2P_REQUIRE(R >= 0 && C > 0 && G > 0 && inner > 0, D2P_EINVAL,
                "bn: bad sizes R=%d C=%d G=%d inner=%d", R, C, G, inner);
    D2P_REQUIRE(R % (G * inner) == 0, D2P_EINVAL,
                "bn: R=%d is not a multiple of G*inner=%d", R, G * inner);
    return D2P_OK;
}

static inline int ew_blocks(long total) {
    long b = (total + 255) / 256;
    if (b > 2048) b = 2048;   // 256 CUs x 8 blocks, grid-stride the rest
    if (b < 1) b = 1;
    return (int)b;
}

static int bn_fwd_impl(int nb, long xs, long ys, long ps, long ms, int R, int C, int G, int inner, const float* x,
                       const float* gamma, const float* beta, float* y, float* mean, float* rstd, float* var_out,
                       float* moving_mean, float* moving_var, float decay, void* ws, size_t ws_bytes,
                       d2p_stream_t stream) {
    int rc = bn_check(R, C, G, inner);
    if (rc) return rc;
    if (R == 0) return D2P_OK;
    D2P_REQUIRE(nb >= 1 && nb <= 65535, D2P_EINVAL, "bn fwd: batch of %d problems", nb);
    D2P_REQUIRE(x && gamma && beta && y && mean && rstd, D2P_EINVAL, "bn fwd: null pointer");
    D2P_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), D2P_EINVAL,
                "bn fwd: moving_mean and moving_var go together");
    const size_t ws1 = align_up(d2p_bn_ws_bytes(R, C, G), 256);
    D2P_REQUIRE(ws && ws_bytes >= (nb > 1 ? ws1 * nb : d2p_bn_ws_bytes(R, C, G)), D2P_EWS,
                "bn fwd: workspace too small (%zu < %zu)", ws_bytes, ws1 * nb);
    hipStream_t st = as_stream(stream);
    D2pProfScope prof(st, D2P_PROF_BN, 3.0 * nb * R * C * sizeof(float));       // two reads, one write
    BnPlan p = bn_plan(R, C, G);
    const int n = R / G;
    const BnBatch bb{xs, ys, ps, ms, nb > 1 ? (long)G * C : 0L, nb > 1 ? (long)ws1 : 0L};
    double* partial = (double*)ws;
    const bool al = ((xs | ys | ps) % 4) == 0;
    const bool vec4 = (C % 4 == 0) && (((uintptr_t)x & 15) == 0) && al;
    // (the workspace stride of a single problem is 0 in bb: the fold needs none, it has one problem)
    BnFold fo{};
    if (bn_fold_ok(nb, G, p.S, C)) {
        fo.tickets = bn_ticket_slot();
        fo.mean = mean; fo.rstd = rstd; fo.var_out = var_out; fo.moving_mean = moving_mean; fo.moving_var = moving_var;
        fo.decay = decay; fo.err = (const unsigned*)d2p_persist_err_ptr();
        fo.gsum = partial + (size_t)G * p.S * C * 2;
    }
    if (vec4)
        hipLaunchKernelGGL((bn_partial_kernel<0, 4>), dim3(G, p.S, nb), dim3(256), 0, st, n, C, G, inner,
                           p.lanes_c, p.row_lanes, x, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, partial, bb, fo);
    else
        hipLaunchKernelGGL((bn_partial_kernel<0, 1>), dim3(G, p.S, nb), dim3(256), 0, st, n, C, G, inner,
                           (C < 256 ? C : 256), 256 / (C < 256 ? C : 256), x, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, partial, bb, fo);
    D2P_LAUNCH_CHECK("bn_partial_fwd");
    double* gsum = partial + (size_t)G * p.S * C * 2;
    BnMoving mo{gsum, nullptr, nullptr, decay, (const unsigned*)d2p_persist_err_ptr(), 0};
    if (!fo.tickets) {
        if (g_bn_gc) {
            hipLaunchKernelGGL(bn_finalize_fwd_gc_kernel, dim3(ceil_div(G * C, 4), 1, nb), dim3(256), 0, st, n, C, G, p.S,
                               partial, mean, rstd, var_out, gsum, bb);
            mo.moving_mean = moving_mean;           // the apply launch's extra workgroup updates them
            mo.moving_var = moving_var;
        } else {
            hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(ceil_div(C, 4), 1, nb), dim3(256), 0, st, n, C, G,
                               p.S, partial, mean, rstd, var_out, moving_mean, moving_var, decay,
                               (const unsigned*)d2p_persist_err_ptr(), bb);
        }
        D2P_LAUNCH_CHECK("bn_finalize_fwd");
    }
    const bool vec = (C % 4 == 0) && al && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma |
                                              (uintptr_t)beta | (uintptr_t)mean | (uintptr_t)rstd) & 15) == 0);
    const int extra = mo.moving_mean ? 1 : 0;
    if (vec) {
        mo.nblk = ew_blocks((long)R * C / 4);
        hipLaunchKernelGGL(bn_apply_fwd_vec4_kernel, dim3(mo.nblk + extra, 1, nb), dim3(256), 0,
                           st, (long)R, C, G, inner, x, gamma, beta, mean, rstd, y, bb, mo);
    } else {
        mo.nblk = ew_blocks((long)R * C);
        hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(mo.nblk + extra, 1, nb), dim3(256), 0, st,
                           (long)R, C, G, inner, x, gamma, beta, mean, rstd, y, bb, mo);
    }
    D2P_LAUNCH_CHECK("bn_apply_fwd");
    return D2P_OK;
}

// ---- statistics from a producer's partial sums; the apply pass alone (round 5) --------------------------------------
// One wavefront per (group, channel): mean / rstd / var from partial[((g*S + s)*C + c)*2 + {0,1}] (what a folding conv
// launch leaves, d2p_conv2d_nhwc_s2_same_fwd_bn), and the folded affine of the NEXT consumer: scale = gamma * rstd,
// shift = beta - mean * scale, and its PAD PIXEL -shift / scale -- the input value the affine maps to zero, which a
// folding conv kernel loads for its out-of-image taps instead of masking them.  |scale| is kept >= 1e-20 so that the pad
// pixel exists (an fp32 gamma never trains that close to zero; every consumer of the affine sees the same scale).
__global__ void __launch_bounds__(256)
bn_stats_from_partials_kernel(int n, int C, int G, int S, const double* partial, const float* gamma, const float* beta,
                              float* mean, float* rstd, float* var_out, float* scale, float* shift, float* pad) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= G * C) return;
    const int g = idx / C, c = idx - g * C;
    double a = 0.0, b = 0.0;
    for (int s = lane; s < S; s += 64) {
        const double* p = partial + (((long)g * S + s) * C + c) * 2;
        a += p[0];
        b += p[1];
    }
    a = wave_reduce_sum(a);
    b = wave_reduce_sum(b);
    const double mu = a / n;
    double var = b / n - mu * mu;   // biased variance
    if (var < 0.0) var = 0.0;
    if (lane == 0) {
        const float m = (float)mu, rs = (float)(1.0 / sqrt(var + BN_EPS));
        mean[idx] = m;
        rstd[idx] = rs;
        if (var_out) var_out[idx] = (float)var;
        if (scale) {
            float sc = gamma[c] * rs;
            if (!(fabsf(sc) >= 1e-20f)) sc = sc < 0.f ? -1e-20f : 1e-20f;
            const float sf = beta[c] - m * sc;
            scale[idx] = sc;
            shift[idx] = sf;
            if (pad) pad[idx] = -sf / sc;
        }
    }
}
extern "C" int d2p_bn_stats_from_partials(int n_per_group, int C, int G, int S, const double* partial, const float* gamma,
                                          const float* beta, float* mean, float* rstd, float* var, float* scale,
                                          float* shift, float* pad, d2p_stream_t stream) {
    D2P_REQUIRE(n_per_group > 0 && C > 0 && G > 0 && S > 0, D2P_EINVAL, "bn stats: bad sizes n=%d C=%d G=%d S=%d", n_per_group, C, G, S);
    D2P_REQUIRE(partial && mean && rstd, D2P_EINVAL, "bn stats: null pointer");
    D2P_REQUIRE((scale == nullptr) == (shift == nullptr) && (!scale || (gamma && beta)) && (!pad || scale), D2P_EINVAL,
                "bn stats: scale / shift go together (pad with them) and need gamma / beta");
    D2pProfScope prof(as_stream(stream), D2P_PROF_BN, (double)G * S * C * 2 * sizeof(double));
    hipLaunchKernelGGL(bn_stats_from_partials_kernel, dim3(ceil_div(G * C, 4)), dim3(256), 0, as_stream(stream), n_per_group, C,
                       G, S, partial, gamma, beta, mean, rstd, var, scale, shift, pad);
    D2P_LAUNCH_CHECK("bn_stats_from_partials");
    return D2P_OK;
}
// the apply pass of d2p_bn_group_fwd alone, with statistics the caller already has
extern "C" int d2p_bn_apply_fwd(int R, int C, int G, int inner, const float* x, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, float* y, d2p_stream_t stream) {
    int rc = bn_check(R, C, G, inner);
    if (rc) return rc;
    if (R == 0) return D2P_OK;
    D2P_REQUIRE(x && gamma && beta && mean && rstd && y, D2P_EINVAL, "bn apply: null pointer");
    const BnBatch bb{0, 0, 0, 0, 0, 0};
    D2pProfScope prof(as_stream(stream), D2P_PROF_BN, 2.0 * R * C * sizeof(float));
    BnMoving mo{nullptr, nullptr, nullptr, 0.f, (const unsigned*)d2p_persist_err_ptr(), 0};
    const bool vec = (C % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)mean |
                                        (uintptr_t)rstd) & 15) == 0);
    if (vec) {
        mo.nblk = ew_blocks((long)R * C / 4);
        hipLaunchKernelGGL(bn_apply_fwd_vec4_kernel, dim3(mo.nblk, 1, 1), dim3(256), 0, as_stream(stream), (long)R, C, G, inner,
                           x, gamma, beta, mean, rstd, y, bb, mo);
    } else {
        mo.nblk = ew_blocks((long)R * C);
        hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(mo.nblk, 1, 1), dim3(256), 0, as_stream(stream), (long)R, C, G, inner, x,
                           gamma, beta, mean, rstd, y, bb, mo);
    }
    D2P_LAUNCH_CHECK("bn_apply_fwd");
    return D2P_OK;
}

extern "C" int d2p_bn_group_fwd(int R, int C, int G, int inner, const float* x, const float* gamma,
                                const float* beta, float* y, float* mean, float* rstd,
                                float* var_out, float* moving_mean, float* moving_var, float decay,
                                void* ws, size_t ws_bytes, d2p_stream_t stream) {
    return bn_fwd_impl(1, 0, 0, 0, 0, R, C, G, inner, x, gamma, beta, y, mean, rstd, var_out, moving_mean, moving_var,
                       decay, ws, ws_bytes, stream);
}
extern "C" size_t d2p_bn_batched_ws_bytes(int nb, int R, int C, int G) {
    return nb <= 0 ? 0 : align_up(d2p_bn_ws_bytes(R, C, G), 256) * (size_t)nb;
}
extern "C" int d2p_bn_group_fwd_batched(int nb, long xs, long ys, long ps, long ms, int R, int C, int G, int inner,
                                        const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                        float* rstd, float* var_out, float* moving_mean, float* moving_var,
                                        float decay, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    return bn_fwd_impl(nb, xs, ys, ps, ms, R, C, G, inner, x, gamma, beta, y, mean, rstd, var_out, moving_mean,
                       moving_var, decay, ws, ws_bytes, stream);
}

// sums / S_sums (optional, nb == 1): the partial sums [G][S_sums][C][2] fp64 = (sum dy, sum dy * xhat) the PRODUCER of dy left
// behind (an input-gradient conv launch: ConvDgradBn) -- the partial-sum pass over (x, dy) is then not run
static int bn_bwd_impl(int nb, long xs, long ys, long ps, int R, int C, int G, int inner, const float* x,
                       const float* dy, const float* gamma, const float* mean, const float* rstd, int act_bwd,
                       float* dx, float* dgamma, float* dbeta, float* dx_colsum, void* ws, size_t ws_bytes,
                       d2p_stream_t stream, const double* sums = nullptr, int S_sums = 0) {
    int rc = bn_check(R, C, G, inner);
    if (rc) return rc;
    if (R == 0) return D2P_OK;
    D2P_REQUIRE(nb >= 1 && nb <= 65535, D2P_EINVAL, "bn bwd: batch of %d problems", nb);
    D2P_REQUIRE(x && dy && gamma && mean && rstd && dx, D2P_EINVAL, "bn bwd: null pointer");
    const size_t ws1 = align_up(d2p_bn_ws_bytes(R, C, G), 256);
    D2P_REQUIRE(ws && ws_bytes >= (nb > 1 ? ws1 * nb : d2p_bn_ws_bytes(R, C, G)), D2P_EWS,
                "bn bwd: workspace too small (%zu < %zu)", ws_bytes, ws1 * nb);
    hipStream_t st = as_stream(stream);
    BnPlan p = bn_plan(R, C, G);
    const int n = R / G;
    D2pProfScope prof(st, D2P_PROF_BN, 5.0 * nb * R * C * sizeof(float));       // (x, dy) twice, dx once
    const BnBatch bb{xs, ys, ps, 0L, nb > 1 ? (long)G * C : 0L, nb > 1 ? (long)ws1 : 0L};
    double* partial = (double*)ws;
    double* gsum = partial + (size_t)G * p.S * C * 2;
    float* m12 = (float*)(gsum + (size_t)G * C * 2);
    const bool al = ((xs | ys | ps) % 4) == 0;
    const bool vec4 = (C % 4 == 0) && al && ((((uintptr_t)x | (uintptr_t)dy) & 15) == 0);
    BnFold fo{};
    unsigned* tickets = nullptr;
    D2P_REQUIRE(!sums || (nb == 1 && S_sums >= 1), D2P_EINVAL, "bn bwd: sums need one problem and S_sums >= 1");
    const int S_use = sums ? S_sums : p.S;
    if (!sums && bn_fold_ok(nb, G, p.S, C)) {
        tickets = fo.tickets = bn_ticket_slot();
        fo.m12 = m12; fo.dgamma = dgamma; fo.dbeta = dbeta; fo.gsum = gsum;
    }
    if (sums) {
        partial = const_cast<double*>(sums);
    } else if (vec4)
        hipLaunchKernelGGL((bn_partial_kernel<1, 4>), dim3(G, p.S, nb), dim3(256), 0, st, n, C, G, inner,
                           p.lanes_c, p.row_lanes, x, dy, mean, rstd, partial, bb, fo);
    else
        hipLaunchKernelGGL((bn_partial_kernel<1, 1>), dim3(G, p.S, nb), dim3(256), 0, st, n, C, G, inner,
                           (C < 256 ? C : 256), 256 / (C < 256 ? C : 256), x, dy, mean, rstd, partial, bb, fo);
    D2P_LAUNCH_CHECK("bn_partial_bwd");
    // dgamma / dbeta by the column-sum finalize launch (it exists when the bias gradient is asked for) from the fp64
    // group sums, m12 by one wavefront per (group, channel)
    const bool gc = !fo.tickets && g_bn_gc && dx_colsum != nullptr;
    if (gc) {
        hipLaunchKernelGGL(bn_finalize_bwd_gc_kernel, dim3(ceil_div(G * C, 4), 1, nb), dim3(256), 0, st, n, C, G, S_use,
                           partial, m12, gsum, bb);
        D2P_LAUNCH_CHECK("bn_finalize_bwd");
    } else if (!fo.tickets) {
        hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(ceil_div(C, 4), 1, nb), dim3(256), 0, st, n, C, G,
                           S_use, partial, m12, dgamma, dbeta, bb);
        D2P_LAUNCH_CHECK("bn_finalize_bwd");
    }
    const bool v4 = (C % 4 == 0) && al && ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0);
    float* colpart = (float*)((char*)m12 + align_up((size_t)G * C * 2 * sizeof(float), 16));
    if (dx_colsum) {
        const int vec = v4 ? 4 : 1;
        const int blocks = bn_sum_blocks(R, C, vec);
        // column-sum finalize folded into the apply launch when its input is small (conv layers: <= 1024 blocks x
        // 16-48 channels); the ticket sits behind the partial-sum tickets of the same slot (or in a slot of its own)
        unsigned* sum_ticket = nullptr;
        // (not in gc mode: there the column-sum finalize LAUNCH is the one writer of dgamma / dbeta -- folding it away
        //  left them unwritten for layers whose partial sums are too many for the ticket fold)
        if (g_bn_fold && !gc && C <= 256 && (long)blocks * C <= 65536 && nb * (G + 2) <= BN_TICKET_WORDS)
            sum_ticket = (tickets ? tickets : bn_ticket_slot()) + nb * (G + 1);
        if (v4)
            hipLaunchKernelGGL((bn_apply_bwd_kernel<4, true>), dim3(blocks, 1, nb), dim3(256), 0, st, R, C, G, inner, x,
                               dy, gamma, mean, rstd, m12, act_bwd, dx, colpart, bb, sum_ticket, dx_colsum);
        else
            hipLaunchKernelGGL((bn_apply_bwd_kernel<1, true>), dim3(blocks, 1, nb), dim3(256), 0, st, R, C, G, inner, x,
                               dy, gamma, mean, rstd, m12, act_bwd, dx, colpart, bb, sum_ticket, dx_colsum);
        D2P_LAUNCH_CHECK("bn_apply_bwd");
        if (!sum_ticket) {
            hipLaunchKernelGGL(bn_colsum_finalize_kernel, dim3(ceil_div(C, 4), 1, nb), dim3(256), 0, st, C, blocks, colpart,
                               dx_colsum, bb, G, gc ? (const double*)gsum : (const double*)nullptr, dgamma, dbeta);
            D2P_LAUNCH_CHECK("bn_colsum_finalize");
        }
    } else {
        if (v4)
            hipLaunchKernelGGL((bn_apply_bwd_kernel<4, false>), dim3(ew_blocks((long)R * C / 4), 1, nb), dim3(256), 0, st,
                               R, C, G, inner, x, dy, gamma, mean, rstd, m12, act_bwd, dx, (float*)nullptr, bb,
                               (unsigned*)nullptr, (float*)nullptr);
        else
            hipLaunchKernelGGL((bn_apply_bwd_kernel<1, false>), dim3(ew_blocks((long)R * C), 1, nb), dim3(256), 0, st, R,
                               C, G, inner, x, dy, gamma, mean, rstd, m12, act_bwd, dx, (float*)nullptr, bb,
                               (unsigned*)nullptr, (float*)nullptr);
        D2P_LAUNCH_CHECK("bn_apply_bwd");
    }
    return D2P_OK;
}

// ---- batch-norm backward as per-(group, channel) COEFFICIENTS (round 5) -----------------------------------------------
// dx = gamma * rstd * (dy - m1 - xhat * m2) [* lrelu'(x)] is affine in (dy, x) per (group, channel):
//   dx = (k1 * dy + k2 * x + k3) [* lrelu'(x)],  k1 = gamma * rstd,  k2 = -k1 * rstd * m2,  k3 = -k1 * m1 - k2 * mean,
// so a CONSUMER of dx (the first conv layer's weight gradient, d2p_conv2d_nhwc_s2_same_wgrad_bnbwd) can form it on load
// from x and dy instead of reading a materialised dx: the apply pass (a read of x and dy and a write of dx) disappears.
// This entry runs the two-stage sums of d2p_bn_group_bwd (same kernels, same order) and leaves coef [G, C, 4] =
// (k1, k2, k3, 0) plus dgamma / dbeta; the bias gradient (column sums of dx) comes from the consumer.
__global__ void __launch_bounds__(256)
bn_bwd_coef_kernel(int C, int G, const float* gamma, const float* mean, const float* rstd, const float* m12,
                   const double* gs, float* coef, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double sb = 0.0, sg = 0.0;
    for (int g = 0; g < G; ++g) {
        const int idx = g * C + c;
        const float rs = rstd[idx], k1 = gamma[c] * rs, k2 = -k1 * rs * m12[(long)idx * 2 + 1];
        const float k3 = -k1 * m12[(long)idx * 2] - k2 * mean[idx];
        *reinterpret_cast<float4*>(coef + (long)idx * 4) = make_float4(k1, k2, k3, 0.f);
        sb += gs[(long)idx * 2 + 0];
        sg += gs[(long)idx * 2 + 1];
    }
    if (dgamma) dgamma[c] = (float)sg;
    if (dbeta) dbeta[c] = (float)sb;
}
extern "C" int d2p_bn_group_bwd_coef(int R, int C, int G, int inner, const float* x, const float* dy, const float* gamma,
                                     const float* mean, const float* rstd, float* coef, float* dgamma, float* dbeta,
                                     const double* sums, int S_sums, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    int rc = bn_check(R, C, G, inner);
    if (rc) return rc;
    D2P_REQUIRE(R > 0 && (sums || (x && dy)) && gamma && mean && rstd && coef, D2P_EINVAL, "bn bwd coef: null pointer or no rows");
    D2P_REQUIRE(!sums || S_sums >= 1, D2P_EINVAL, "bn bwd coef: S_sums = %d", S_sums);
    D2P_REQUIRE(((uintptr_t)coef & 15) == 0, D2P_EALIGN, "bn bwd coef: coef must be 16-byte aligned");
    D2P_REQUIRE(ws && ws_bytes >= d2p_bn_ws_bytes(R, C, G), D2P_EWS, "bn bwd coef: workspace too small");
    hipStream_t st = as_stream(stream);
    D2pProfScope prof(st, D2P_PROF_BN, sums ? (double)G * S_sums * C * 2 * sizeof(double) : 2.0 * R * C * sizeof(float));
    BnPlan p = bn_plan(R, C, G);
    const int n = R / G;
    const BnBatch bb{0, 0, 0, 0, 0, 0};
    double* partial = (double*)ws;
    double* gsum = partial + (size_t)G * p.S * C * 2;
    float* m12 = (float*)(gsum + (size_t)G * C * 2);
    const bool vec4 = (C % 4 == 0) && ((((uintptr_t)x | (uintptr_t)dy) & 15) == 0);
    BnFold fo{};
    if (sums) {
        // (the producer of dy already left the partial sums: d2p_conv2d_nhwc_s2_same_dgrad_bn)
    } else if (vec4)
        hipLaunchKernelGGL((bn_partial_kernel<1, 4>), dim3(G, p.S, 1), dim3(256), 0, st, n, C, G, inner, p.lanes_c, p.row_lanes,
                           x, dy, mean, rstd, partial, bb, fo);
    else
        hipLaunchKernelGGL((bn_partial_kernel<1, 1>), dim3(G, p.S, 1), dim3(256), 0, st, n, C, G, inner, (C < 256 ? C : 256),
                           256 / (C < 256 ? C : 256), x, dy, mean, rstd, partial, bb, fo);
    D2P_LAUNCH_CHECK("bn_partial_bwd");
    hipLaunchKernelGGL(bn_finalize_bwd_gc_kernel, dim3(ceil_div(G * C, 4), 1, 1), dim3(256), 0, st, n, C, G,
                       sums ? S_sums : p.S, sums ? sums : (const double*)partial, m12, gsum, bb);
    D2P_LAUNCH_CHECK("bn_finalize_bwd");
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, C, G, gamma, mean, rstd, m12, gsum, coef,
                       dgamma, dbeta);
    D2P_LAUNCH_CHECK("bn_bwd_coef");
    return D2P_OK;
}

extern "C" int d2p_bn_group_bwd(int R, int C, int G, int inner, const float* x, const float* dy,
                                const float* gamma, const float* mean, const float* rstd,
                                int act_bwd, float* dx, float* dgamma, float* dbeta, float* dx_colsum,
                                void* ws, size_t ws_bytes, d2p_stream_t stream) {
    return bn_bwd_impl(1, 0, 0, 0, R, C, G, inner, x, dy, gamma, mean, rstd, act_bwd, dx, dgamma, dbeta, dx_colsum, ws,
                       ws_bytes, stream);
}
extern "C" int d2p_bn_group_bwd_sums(int R, int C, int G, int inner, const float* x, const float* dy,
                                     const float* gamma, const float* mean, const float* rstd,
                                     int act_bwd, float* dx, float* dgamma, float* dbeta, float* dx_colsum,
                                     const double* sums, int S_sums, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    D2P_REQUIRE(sums && S_sums >= 1, D2P_EINVAL, "bn bwd (sums): null sums or S_sums = %d", S_sums);
    return bn_bwd_impl(1, 0, 0, 0, R, C, G, inner, x, dy, gamma, mean, rstd, act_bwd, dx, dgamma, dbeta, dx_colsum, ws,
                       ws_bytes, stream, sums, S_sums);
}
extern "C" int d2p_bn_group_bwd_batched(int nb, long xs, long ys, long ps, int R, int C, int G, int inner,
                                        const float* x, const float* dy, const float* gamma, const float* mean,
                                        const float* rstd, int act_bwd, float* dx, float* dgamma, float* dbeta,
                                        float* dx_colsum, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    return bn_bwd_impl(nb, xs, ys, ps, R, C, G, inner, x, dy, gamma, mean, rstd, act_bwd, dx, dgamma, dbeta, dx_colsum,
                       ws, ws_bytes, stream);
}

__global__ void __launch_bounds__(256)
bn_update_moving_kernel(int C, int G, float decay, const float* mean, const float* var,
                        float* mm, float* mv) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = mm[c], b = mv[c];
    for (int g = 0; g < G; ++g) {   // one update per Demo_Encoder call, in call order
        a = decay * a + (1.f - decay) * mean[g * C + c];
        b = decay * b + (1.f - decay) * var[g * C + c];
    }
    mm[c] = a;
    mv[c] = b;
}

extern "C" int d2p_bn_update_moving(int C, int G, float decay, const float* mean, const float* var,
                                    float* moving_mean, float* moving_var, d2p_stream_t stream) {
    D2P_REQUIRE(C > 0 && G > 0, D2P_EINVAL, "bn moving: bad sizes C=%d G=%d", C, G);
    D2P_REQUIRE(mean && var && moving_mean && moving_var, D2P_EINVAL, "bn moving: null pointer");
    hipLaunchKernelGGL(bn_update_moving_kernel, dim3(ceil_div(C, 256)), dim3(256), 0,
                       as_stream(stream), C, G, decay, mean, var, moving_mean, moving_var);
    D2P_LAUNCH_CHECK("bn_update_moving");
    return D2P_OK;
}

// ---- inference mode (is_training=False: evaler.py:61 builds Model(config, is_train=False)) ----
// y = (x - moving_mean) * rsqrt(moving_var + eps) * gamma + beta, per channel.
template <int VEC>
__global__ void __launch_bounds__(256)
bn_inference_kernel(long total_v, int C, const float* __restrict__ x, const float* __restrict__ gamma,
                    const float* __restrict__ beta, const float* __restrict__ mm, const float* __restrict__ mv,
                    float eps, float* __restrict__ y) {
    const int CV = C / VEC;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total_v; i += (long)gridDim.x * 256L) {
        const int c0 = (int)(i % CV) * VEC;
        float xv[VEC], o[VEC];
        if (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(x + i * 4);
            xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
        } else {
            xv[0] = x[i];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = c0 + j;
            o[j] = (xv[j] - mm[c]) * rsqrtf(mv[c] + eps) * gamma[c] + beta[c];
        }
        if (VEC == 4) *reinterpret_cast<float4*>(y + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
        else y[i] = o[0];
    }
}

extern "C" int d2p_bn_inference_fwd(int R, int C, const float* x, const float* gamma, const float* beta,
                                    const float* moving_mean, const float* moving_var, float* y,
                                    d2p_stream_t stream) {
    D2P_REQUIRE(R >= 0 && C > 0, D2P_EINVAL, "bn inference: bad sizes R=%d C=%d", R, C);
    if (R == 0) return D2P_OK;
    D2P_REQUIRE(x && gamma && beta && moving_mean && moving_var && y, D2P_EINVAL, "bn inference: null pointer");
    hipStream_t st = as_stream(stream);
    const bool v4 = (C % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    if (v4)
        hipLaunchKernelGGL((bn_inference_kernel<4>), dim3(ew_blocks((long)R * C / 4)), dim3(256), 0, st,
                           (long)R * C / 4, C, x, gamma, beta, moving_mean, moving_var, 1e-3f, y);
    else
        hipLaunchKernelGGL((bn_inference_kernel<1>), dim3(ew_blocks((long)R * C)), dim3(256), 0, st, (long)R * C, C,
                           x, gamma, beta, moving_mean, moving_var, 1e-3f, y);
    D2P_LAUNCH_CHECK("bn_inference");
    return D2P_OK;
}
