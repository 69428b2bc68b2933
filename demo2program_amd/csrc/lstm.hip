// K3/K4 + sequence drivers: BasicLSTMCell gates and the recurrent loop (include/d2p.h).
// Replaces rnn.BasicLSTMCell + tf.nn.dynamic_rnn (models/model_full.py:244-256,265-276)
// and the BasicDecoder/TrainingHelper loop (models/model_full.py:413,465-471).
//
// [TF-1.3] BasicLSTMCell: z = [x,h]·W + b split as i, j, f, o;
//   c' = c*sigmoid(f + 1.0) + sigmoid(i)*tanh(j);  h' = tanh(c')*sigmoid(o).
// dynamic_rnn(sequence_length): for t >= len the emitted output is 0 and (c,h) copy through.
//
// The input projection x·Wx + b is hoisted out of the loop by the caller (one large GEMM
// over all steps); per step only the recurrent h·Wh GEMM (accumulating into the stored
// pre-activations) and the pointwise gate kernel run.  The pre-activations are kept for
// backward; activations are recomputed there (saves 3/4 of the saved-state traffic).
#include "common.h"
#include "prof.h"

#include "lstm_math.h"
#include "lstm_internal.h"

__global__ void __launch_bounds__(256)
lstm_gate_fwd_kernel(int M, int U, const float* __restrict__ z, long zrs,
                     const float* __restrict__ c_prev, const float* __restrict__ h_prev,
                     const int* __restrict__ lens, int t, float* __restrict__ c_out,
                     float* __restrict__ hs_out, float* __restrict__ h_out) {
    const int U4 = U >> 2;
    const long total = (long)M * U4;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        const int r = (int)(idx / U4);
        const int u = (int)(idx - (long)r * U4) * 4;
        const long o = (long)r * U + u;
        const bool active = lens ? (t < lens[r]) : true;
        f4 cp = c_prev ? ldf4(c_prev + o) : zero4();
        if (active) {
            const float* zr = z + (long)r * zrs + u;
            const f4 zi = ldf4(zr), zj = ldf4(zr + U), zf = ldf4(zr + 2 * U), zo = ldf4(zr + 3 * U);
            f4 cn, hn;
            lstm_gate_fwd4(zi, zj, zf, zo, cp, cn, hn);
            stf4(c_out + o, cn);
            if (hs_out) stf4(hs_out + o, hn);
            stf4(h_out + o, hn);
        } else {
            stf4(c_out + o, cp);
            if (hs_out) stf4(hs_out + o, h_prev ? ldf4(h_prev + o) : zero4());
            stf4(h_out + o, zero4());
        }
    }
}

__global__ void __launch_bounds__(256)
lstm_gate_bwd_kernel(int M, int U, const float* __restrict__ z, long zrs,
                     const float* __restrict__ c_prev, const float* __restrict__ c,
                     const float* __restrict__ dh_in, const float* __restrict__ dh_out_grad,
                     const int* __restrict__ lens, int t, float* __restrict__ dc,
                     float* __restrict__ dz, long dzrs, float* __restrict__ dh_pass) {
    const int U4 = U >> 2;
    const long total = (long)M * U4;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        const int r = (int)(idx / U4);
        const int u = (int)(idx - (long)r * U4) * 4;
        const long o = (long)r * U + u;
        const bool active = lens ? (t < lens[r]) : true;
        float* dzr = dz + (long)r * dzrs + u;
        if (active) {
            const float* zr = z + (long)r * zrs + u;
            const f4 zi = ldf4(zr), zj = ldf4(zr + U), zf = ldf4(zr + 2 * U), zo = ldf4(zr + 3 * U);
            const f4 cp = c_prev ? ldf4(c_prev + o) : zero4();
            const f4 cc = ldf4(c + o);
            f4 dh = dh_in ? ldf4(dh_in + o) : zero4();
            if (dh_out_grad) {
                const f4 e = ldf4(dh_out_grad + o);
#pragma unroll
                for (int q = 0; q < 4; ++q) dh.v[q] += e.v[q];
            }
            f4 dcv = ldf4(dc + o);
            f4 gi, gj, gf, go, dcn;
            lstm_gate_bwd4(zi, zj, zf, zo, cp, cc, dh, dcv, gi, gj, gf, go, dcn);
            stf4(dzr, gi);
            stf4(dzr + U, gj);
            stf4(dzr + 2 * U, gf);
            stf4(dzr + 3 * U, go);
            stf4(dc + o, dcn);
            if (dh_pass) stf4(dh_pass + o, zero4());
        } else {
            const f4 zz = zero4();
            stf4(dzr, zz);
            stf4(dzr + U, zz);
            stf4(dzr + 2 * U, zz);
            stf4(dzr + 3 * U, zz);
            if (dh_pass) stf4(dh_pass + o, dh_in ? ldf4(dh_in + o) : zero4());
        }
    }
}

static inline int gate_blocks(long total) {
    long b = (total + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

static int gate_check(int M, int U, long zrs, const void* z) {
    D2P_REQUIRE(M >= 0 && U > 0, D2P_EINVAL, "lstm gate: bad sizes M=%d U=%d", M, U);
    D2P_REQUIRE(U % 4 == 0 && zrs % 4 == 0 && (((uintptr_t)z & 15) == 0), D2P_EALIGN,
                "lstm gate: needs U %% 4 == 0, row stride %% 4 == 0 and 16-byte aligned z");
    return D2P_OK;
}

extern "C" int d2p_lstm_gate_fwd(int M, int U, const float* z, long z_row_stride,
                                 const float* c_prev, const float* h_prev, const int* lens, int t,
                                 float* c_out, float* h_state_out, float* h_out,
                                 d2p_stream_t stream) {
    int rc = gate_check(M, U, z_row_stride, z);
    if (rc) return rc;
    if (M == 0) return D2P_OK;
    D2P_REQUIRE(z && c_out && h_out, D2P_EINVAL, "lstm gate fwd: null pointer");
    // algorithmic bytes/row: read 4U pre-activations + U c_prev, write U c + U h
    D2pProfScope prof(as_stream(stream), D2P_PROF_GATE_FWD, 7.0 * U * 4.0 * M);
    hipLaunchKernelGGL(lstm_gate_fwd_kernel, dim3(gate_blocks((long)M * U / 4)), dim3(256), 0,
                       as_stream(stream), M, U, z, z_row_stride, c_prev, h_prev, lens, t, c_out,
                       h_state_out, h_out);
    D2P_LAUNCH_CHECK("lstm_gate_fwd");
    return D2P_OK;
}

extern "C" int d2p_lstm_gate_bwd(int M, int U, const float* z, long z_row_stride,
                                 const float* c_prev, const float* c, const float* dh_in,
                                 const float* dh_out_grad, const int* lens, int t, float* dc,
                                 float* dz, long dz_row_stride, float* dh_pass,
                                 d2p_stream_t stream) {
    int rc = gate_check(M, U, z_row_stride, z);
    if (rc) return rc;
    if (M == 0) return D2P_OK;
    D2P_REQUIRE(z && c && dc && dz, D2P_EINVAL, "lstm gate bwd: null pointer");
    D2P_REQUIRE(dz_row_stride % 4 == 0 && (((uintptr_t)dz & 15) == 0), D2P_EALIGN,
                "lstm gate bwd: dz must be 16-byte aligned with row stride %% 4 == 0");
    // algorithmic bytes/row: read dh, dc, 4U pre-activations, c_prev, c (8U); write 4U dz + U dc
    D2pProfScope prof(as_stream(stream), D2P_PROF_GATE_BWD, 13.0 * U * 4.0 * M);
    hipLaunchKernelGGL(lstm_gate_bwd_kernel, dim3(gate_blocks((long)M * U / 4)), dim3(256), 0,
                       as_stream(stream), M, U, z, z_row_stride, c_prev, c, dh_in, dh_out_grad, lens,
                       t, dc, dz, dz_row_stride, dh_pass);
    D2P_LAUNCH_CHECK("lstm_gate_bwd");
    return D2P_OK;
}

// ---- sequence drivers ------------------------------------------------------------------
// Fused recurrent-step path (lstm_step.hip); the unfused path below (generic GEMM + gate
// kernel per step) remains as the reference implementation and for unsupported sizes.
bool d2p_lstm_fused_eligible(int M, int U);
size_t d2p_lstm_fused_ws_bytes(int M, int U);
int d2p_lstm_fused_fwd(int M, int U, int n_steps, float* z, long zrs, long zts, const float* Wh,
                       const float* h0, const float* c0, const int* lens, float* hout, float* cs,
                       float* h_final, float* c_final, float* ws, hipStream_t st);
int d2p_lstm_fused_bwd(int M, int U, int n_steps, const float* z, long zrs, long zts, const float* Wh,
                       const float* c0, const int* lens, const float* cs, const float* dhout,
                       const float* dh_final, const float* dc_final, float* dz, float* dh0,
                       float* dc0, float* ws, hipStream_t st);

static int g_lstm_fused = 1;

extern "C" int d2p_lstm_set_fused(int on) {
    g_lstm_fused = on ? 1 : 0;
    return D2P_OK;
}
int d2p_lstm_is_fused_enabled() { return g_lstm_fused; }

static size_t unfused_ws_bytes(int M, int U) { return (size_t)3 * M * U * sizeof(float); }

extern "C" size_t d2p_lstm_ws_bytes(int M, int U) {
    if (M <= 0 || U <= 0) return 0;
    size_t a = unfused_ws_bytes(M, U);
    if (d2p_lstm_fused_eligible(M, U)) {
        size_t b = d2p_lstm_fused_ws_bytes(M, U);
        if (b > a) a = b;
        b = d2p_lstm_persist_ws_bytes(M, U);
        if (b > a) a = b;
    }
    return a;
}

static bool use_fused(int M, int U, long zrs, const void* z, size_t ws_bytes) {
    return g_lstm_fused && d2p_lstm_fused_eligible(M, U) && ws_bytes >= d2p_lstm_fused_ws_bytes(M, U) &&
           (zrs % 4 == 0) && (((uintptr_t)z & 15) == 0);
}

static int copy_or_zero(float* dst, const float* src, size_t n, hipStream_t st) {
    if (!dst) return D2P_OK;
    if (src) {
        if (src != dst) D2P_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        D2P_HIP(hipMemsetAsync(dst, 0, n * sizeof(float), st));
    }
    return D2P_OK;
}

// Two sequences of a multi call as ONE persistent launch (lstm_persist.hip): taken when both would go to the
// persistent back end on their own and sharing the chip is expected to pay.  Returns false when the caller
// should issue them one after the other.
bool d2p_lstm_try_pair_fwd(const d2p_lstm_fwd_desc* d, hipStream_t st, int* rc) {
    for (int i = 0; i < 2; ++i) {
        const d2p_lstm_fwd_desc& q = d[i];
        if (q.M <= 0 || q.n_steps <= 0 || !(q.z && q.Wh && q.hout && q.cs && q.ws)) return false;
        if (!use_fused(q.M, q.U, q.z_row_stride, q.z, q.ws_bytes) || q.ws_bytes < d2p_lstm_persist_ws_bytes(q.M, q.U))
            return false;
    }
    if (d[0].U != d[1].U || d[0].ws == d[1].ws) return false;
    if (!d2p_lstm_persist_fwd_pair_ok(d[0].M, d[0].n_steps, d[1].M, d[1].n_steps, d[0].U)) return false;
    PsFwdCall c[2];
    for (int i = 0; i < 2; ++i)
        c[i] = PsFwdCall{d[i].M, d[i].U, d[i].n_steps, d[i].z, d[i].z_row_stride, d[i].z_t_stride, d[i].Wh, d[i].h0,
                         d[i].c0, d[i].lens, d[i].hout, d[i].cs, d[i].h_final, d[i].c_final, (float*)d[i].ws,
                         d[i].flags, d[i].epoch, d[i].wpack};
    *rc = d2p_lstm_persist_fwd_pair(c[0], c[1], st);
    return true;
}
// One to three sequences as ONE launch of the wide-tile persistent kernel (lstm_persist.hip): direct launches only.
bool d2p_lstm_try_wide_fwd(int n, const d2p_lstm_fwd_desc* d, hipStream_t st, int* rc) {
    if (n < 1 || n > 3) return false;
    PsFwdCall c[3];
    for (int i = 0; i < n; ++i) {
        const d2p_lstm_fwd_desc& q = d[i];
        if (q.M <= 0 || q.n_steps <= 0 || !(q.z && q.Wh && q.hout && q.cs && q.ws && q.flags)) return false;
        if (!use_fused(q.M, q.U, q.z_row_stride, q.z, q.ws_bytes) || q.ws_bytes < d2p_lstm_persist_ws_bytes(q.M, q.U))
            return false;
        c[i] = PsFwdCall{q.M, q.U, q.n_steps, q.z, q.z_row_stride, q.z_t_stride, q.Wh, q.h0, q.c0, q.lens, q.hout, q.cs,
                         q.h_final, q.c_final, (float*)q.ws, q.flags, q.epoch, q.wpack, q.rowmap, q.slab_steps};
    }
    if (!d2p_lstm_persist_fwd_wide_ok(n, c)) return false;
    *rc = d2p_lstm_persist_fwd_wide(n, c, st);
    return true;
}
bool d2p_lstm_try_pair_bwd(const d2p_lstm_bwd_desc* d, hipStream_t st, int* rc) {
    for (int i = 0; i < 2; ++i) {
        const d2p_lstm_bwd_desc& q = d[i];
        if (q.M <= 0 || q.n_steps <= 0 || !(q.z && q.Wh && q.cs && q.dz && q.ws)) return false;
        if (!use_fused(q.M, q.U, q.z_row_stride, q.z, q.ws_bytes) || (((uintptr_t)q.dz & 15) != 0) ||
            q.ws_bytes < d2p_lstm_persist_ws_bytes(q.M, q.U))
            return false;
    }
    if (d[0].U != d[1].U || d[0].ws == d[1].ws) return false;
    if (!d2p_lstm_persist_bwd_pair_ok(d[0].M, d[0].n_steps, d[1].M, d[1].n_steps, d[0].U)) return false;
    PsBwdCall c[2];
    for (int i = 0; i < 2; ++i)
        c[i] = PsBwdCall{d[i].M, d[i].U, d[i].n_steps, d[i].z, d[i].z_row_stride, d[i].z_t_stride, d[i].Wh, d[i].c0,
                         d[i].lens, d[i].cs, d[i].dhout, d[i].dh_final, d[i].dc_final, d[i].dz, d[i].dh0, d[i].dc0,
                         (float*)d[i].ws, d[i].db, d[i].flags, d[i].epoch, d[i].wpack, d[i].rowmap,
                         d[i].slab_steps};
    *rc = d2p_lstm_persist_bwd_pair(c[0], c[1], st);
    return true;
}

bool d2p_lstm_try_triple_bwd(const d2p_lstm_bwd_desc* d, hipStream_t st, int* rc) {
    int M[3], T[3];
    for (int i = 0; i < 3; ++i) {
        const d2p_lstm_bwd_desc& q = d[i];
        if (q.M <= 0 || q.n_steps <= 0 || !(q.z && q.Wh && q.cs && q.dz && q.ws)) return false;
        if (!use_fused(q.M, q.U, q.z_row_stride, q.z, q.ws_bytes) || (((uintptr_t)q.dz & 15) != 0) ||
            q.ws_bytes < d2p_lstm_persist_ws_bytes(q.M, q.U) || q.U != d[0].U)
            return false;
        M[i] = q.M; T[i] = q.n_steps;
    }
    if (d[0].ws == d[1].ws || d[0].ws == d[2].ws || d[1].ws == d[2].ws) return false;
    if (!d2p_lstm_persist_bwd_triple_ok(M, T, d[0].U)) return false;
    PsBwdCall c[3];
    for (int i = 0; i < 3; ++i)
        c[i] = PsBwdCall{d[i].M, d[i].U, d[i].n_steps, d[i].z, d[i].z_row_stride, d[i].z_t_stride, d[i].Wh, d[i].c0,
                         d[i].lens, d[i].cs, d[i].dhout, d[i].dh_final, d[i].dc_final, d[i].dz, d[i].dh0, d[i].dc0,
                         (float*)d[i].ws, d[i].db, d[i].flags, d[i].epoch, d[i].wpack, d[i].rowmap,
                         d[i].slab_steps};
    *rc = d2p_lstm_persist_bwd_triple(c, st);
    return true;
}

static int seq_fwd_impl(int M, int U, int n_steps, float* z, long z_row_stride,
                        long z_t_stride, const float* Wh, const float* h0, const float* c0,
                        const int* lens, float* hout, float* cs, float* h_final,
                        float* c_final, void* ws, size_t ws_bytes, d2p_stream_t stream, unsigned* flags, unsigned epoch,
                        const float* wpack = nullptr);

extern "C" int d2p_lstm_seq_fwd(int M, int U, int n_steps, float* z, long z_row_stride,
                                long z_t_stride, const float* Wh, const float* h0, const float* c0,
                                const int* lens, float* hout, float* cs, float* h_final,
                                float* c_final, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    return seq_fwd_impl(M, U, n_steps, z, z_row_stride, z_t_stride, Wh, h0, c0, lens, hout, cs, h_final, c_final, ws,
                        ws_bytes, stream, nullptr, 0u);
}
int d2p_lstm_seq_fwd_desc(const d2p_lstm_fwd_desc* q, d2p_stream_t stream) {
    return seq_fwd_impl(q->M, q->U, q->n_steps, q->z, q->z_row_stride, q->z_t_stride, q->Wh, q->h0, q->c0, q->lens,
                        q->hout, q->cs, q->h_final, q->c_final, q->ws, q->ws_bytes, stream, q->flags, q->epoch, q->wpack);
}

static int seq_fwd_impl(int M, int U, int n_steps, float* z, long z_row_stride,
                        long z_t_stride, const float* Wh, const float* h0, const float* c0,
                        const int* lens, float* hout, float* cs, float* h_final,
                        float* c_final, void* ws, size_t ws_bytes, d2p_stream_t stream, unsigned* flags, unsigned epoch,
                        const float* wpack) {
    D2P_REQUIRE(M >= 0 && U > 0 && n_steps >= 0, D2P_EINVAL, "lstm seq fwd: bad sizes");
    hipStream_t st = as_stream(stream);
    const size_t MU = (size_t)M * U;
    if (M == 0) return D2P_OK;
    D2P_REQUIRE(n_steps == 0 || (z && Wh && hout && cs), D2P_EINVAL, "lstm seq fwd: null pointer");
    D2P_REQUIRE(ws && ws_bytes >= unfused_ws_bytes(M, U), D2P_EWS,
                "lstm seq fwd: workspace too small (%zu < %zu)", ws_bytes, d2p_lstm_ws_bytes(M, U));
    if (n_steps > 0 && use_fused(M, U, z_row_stride, z, ws_bytes) && d2p_lstm_persist_fwd_ok(M, U, n_steps) &&
        ws_bytes >= d2p_lstm_persist_ws_bytes(M, U))
        return d2p_lstm_persist_fwd(PsFwdCall{M, U, n_steps, z, z_row_stride, z_t_stride, Wh, h0, c0, lens, hout, cs,
                                              h_final, c_final, (float*)ws, flags, epoch, wpack}, st);
    if (n_steps > 0 && use_fused(M, U, z_row_stride, z, ws_bytes))
        return d2p_lstm_fused_fwd(M, U, n_steps, z, z_row_stride, z_t_stride, Wh, h0, c0, lens, hout,
                                  cs, h_final, c_final, (float*)ws, st);
    float* hs[2] = {(float*)ws, (float*)ws + MU};
    const float* h_prev = h0;
    const float* c_prev = c0;
    d2p_prof_set_tag(1);   // sub-tag 1 = launches inside the recurrence
    for (int t = 0; t < n_steps; ++t) {
        float* zt = z + (long)t * z_t_stride;
        if (h_prev) {
            int rc = d2p_gemm_f32_nn(M, 4 * U, U, h_prev, U, Wh, 4L * U, zt, z_row_stride, nullptr,
                                     0, /*accumulate=*/1, nullptr, 0, stream);
            if (rc) return rc;
        }
        float* hs_out = lens ? hs[t & 1] : nullptr;
        int rc = d2p_lstm_gate_fwd(M, U, zt, z_row_stride, c_prev, h_prev, lens, t, cs + t * MU,
                                   hs_out, hout + t * MU, stream);
        if (rc) return rc;
        h_prev = lens ? hs_out : hout + t * MU;
        c_prev = cs + t * MU;
    }
    d2p_prof_set_tag(0);
    int rc = copy_or_zero(h_final, h_prev, MU, st);
    if (rc) return rc;
    return copy_or_zero(c_final, c_prev, MU, st);
}

static int seq_bwd_impl(int M, int U, int n_steps, const float* z, long z_row_stride,
                        long z_t_stride, const float* Wh, const float* c0, const int* lens,
                        const float* cs, const float* dhout, const float* dh_final,
                        const float* dc_final, float* dz, float* dh0, float* dc0, void* ws,
                        size_t ws_bytes, d2p_stream_t stream, float* db, bool* db_done, unsigned* flags = nullptr,
                        unsigned epoch = 0u, const float* wpack = nullptr, const int* rowmap = nullptr,
                        const int* slab_steps = nullptr);

extern "C" int d2p_lstm_seq_bwd(int M, int U, int n_steps, const float* z, long z_row_stride,
                                long z_t_stride, const float* Wh, const float* c0, const int* lens,
                                const float* cs, const float* dhout, const float* dh_final,
                                const float* dc_final, float* dz, float* dh0, float* dc0, void* ws,
                                size_t ws_bytes, d2p_stream_t stream) {
    bool done = false;
    return seq_bwd_impl(M, U, n_steps, z, z_row_stride, z_t_stride, Wh, c0, lens, cs, dhout, dh_final, dc_final, dz,
                        dh0, dc0, ws, ws_bytes, stream, nullptr, &done);
}

// The bias gradient of a sequence whose back end did not produce it inside its launches (per-step kernels,
// GEMM + gate path): db = column sums of dz over the n_steps*M rows.
int d2p_lstm_db_colsum(const d2p_lstm_bwd_desc* q, d2p_stream_t stream) {
    if (!q->db) return D2P_OK;
    if (q->M <= 0 || q->n_steps <= 0) {
        D2P_HIP(hipMemsetAsync(q->db, 0, (size_t)4 * q->U * sizeof(float), as_stream(stream)));
        return D2P_OK;
    }
    D2P_REQUIRE(q->z_t_stride == (long)q->M * q->z_row_stride, D2P_EINVAL,
                "lstm bwd: the bias gradient needs dz rows at one stride across steps");
    const int rows = q->n_steps * q->M, cols = 4 * q->U;
    D2P_REQUIRE(q->ws_bytes >= d2p_colsum_ws_bytes(rows, cols), D2P_EWS, "lstm bwd: workspace too small for the bias gradient");
    return d2p_colsum_f32(rows, cols, q->dz, q->z_row_stride, q->db, q->ws, q->ws_bytes, stream);
}

int d2p_lstm_seq_bwd_desc(const d2p_lstm_bwd_desc* q, d2p_stream_t stream) {
    bool done = false;
    int rc = seq_bwd_impl(q->M, q->U, q->n_steps, q->z, q->z_row_stride, q->z_t_stride, q->Wh, q->c0, q->lens, q->cs,
                          q->dhout, q->dh_final, q->dc_final, q->dz, q->dh0, q->dc0, q->ws, q->ws_bytes, stream, q->db,
                          &done, q->flags, q->epoch, q->wpack, q->rowmap, q->slab_steps);
    if (rc || done) return rc;
    return d2p_lstm_db_colsum(q, stream);
}

static int seq_bwd_impl(int M, int U, int n_steps, const float* z, long z_row_stride,
                        long z_t_stride, const float* Wh, const float* c0, const int* lens,
                        const float* cs, const float* dhout, const float* dh_final,
                        const float* dc_final, float* dz, float* dh0, float* dc0, void* ws,
                        size_t ws_bytes, d2p_stream_t stream, float* db, bool* db_done, unsigned* flags, unsigned epoch,
                        const float* wpack, const int* rowmap, const int* slab_steps) {
    D2P_REQUIRE(M >= 0 && U > 0 && n_steps >= 0, D2P_EINVAL, "lstm seq bwd: bad sizes");
    hipStream_t st = as_stream(stream);
    const size_t MU = (size_t)M * U;
    if (M == 0) return D2P_OK;
    D2P_REQUIRE(n_steps == 0 || (z && Wh && cs && dz), D2P_EINVAL, "lstm seq bwd: null pointer");
    D2P_REQUIRE(ws && ws_bytes >= unfused_ws_bytes(M, U), D2P_EWS,
                "lstm seq bwd: workspace too small (%zu < %zu)", ws_bytes, d2p_lstm_ws_bytes(M, U));
    if (n_steps > 0 && use_fused(M, U, z_row_stride, z, ws_bytes) && (((uintptr_t)dz & 15) == 0) &&
        d2p_lstm_persist_bwd_ok(M, U, n_steps) && ws_bytes >= d2p_lstm_persist_ws_bytes(M, U)) {
        *db_done = true;
        return d2p_lstm_persist_bwd(PsBwdCall{M, U, n_steps, z, z_row_stride, z_t_stride, Wh, c0, lens, cs, dhout,
                                              dh_final, dc_final, dz, dh0, dc0, (float*)ws, db, flags, epoch, wpack,
                                              rowmap, slab_steps}, st);
    }
    if (n_steps > 0 && use_fused(M, U, z_row_stride, z, ws_bytes) && (((uintptr_t)dz & 15) == 0))
        return d2p_lstm_fused_bwd(M, U, n_steps, z, z_row_stride, z_t_stride, Wh, c0, lens, cs, dhout,
                                  dh_final, dc_final, dz, dh0, dc0, (float*)ws, st);
    float* dHbuf[2] = {(float*)ws, (float*)ws + MU};
    float* dC = (float*)ws + 2 * MU;
    int rc = copy_or_zero(dC, dc_final, MU, st);
    if (rc) return rc;
    const float* dh_in = dh_final;   // may be null (= 0)
    d2p_prof_set_tag(1);
    for (int t = n_steps - 1; t >= 0; --t) {
        const float* c_prev = t ? cs + (size_t)(t - 1) * MU : c0;
        const bool last = (t == 0);
        if (last && !dh0) {
            // caller does not need dh0: still need dz[0] and dc0
            rc = d2p_lstm_gate_bwd(M, U, z + (long)t * z_t_stride, z_row_stride, c_prev, cs + t * MU,
                                   dh_in, dhout ? dhout + t * MU : nullptr, lens, t, dC,
                                   dz + (long)t * z_t_stride, z_row_stride, nullptr, stream);
            if (rc) return rc;
            dh_in = nullptr;
            break;
        }
        float* target = last ? dh0 : dHbuf[t & 1];
        rc = d2p_lstm_gate_bwd(M, U, z + (long)t * z_t_stride, z_row_stride, c_prev, cs + t * MU,
                               dh_in, dhout ? dhout + t * MU : nullptr, lens, t, dC,
                               dz + (long)t * z_t_stride, z_row_stride, lens ? target : nullptr,
                               stream);
        if (rc) return rc;
        // dh_prev = dz[t] · Wh^T (+ pass-through of masked rows)
        rc = d2p_gemm_f32_nt(M, U, 4 * U, dz + (long)t * z_t_stride, z_row_stride, Wh, 4L * U, target,
                             U, nullptr, 0, /*accumulate=*/lens ? 1 : 0, nullptr, 0, stream);
        if (rc) return rc;
        dh_in = target;
    }
    d2p_prof_set_tag(0);
    if (n_steps == 0) {
        rc = copy_or_zero(dh0, dh_final, MU, st);
        if (rc) return rc;
    }
    return copy_or_zero(dc0, dC, MU, st);
}
