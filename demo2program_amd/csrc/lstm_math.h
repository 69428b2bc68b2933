// BasicLSTMCell pointwise math shared by the standalone gate kernels (lstm.hip) and the
// fused recurrent-step kernels (lstm_step.hip).  [TF-1.3] rnn.BasicLSTMCell.call
// (models/model_full.py:244-246): gate order i, j, f, o; forget_bias = 1.0.
#pragma once
#include "common.h"

#define D2P_FORGET_BIAS 1.0f

struct f4 { float v[4]; };
__device__ __forceinline__ f4 ldf4(const float* p) {
    float4 t = *reinterpret_cast<const float4*>(p);
    f4 r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r;
}
__device__ __forceinline__ void stf4(float* p, const f4& a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
__device__ __forceinline__ f4 zero4() { f4 r; r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.f; return r; }

// One cell.  c' = c*sigmoid(f+1) + sigmoid(i)*tanh(j);  h' = tanh(c')*sigmoid(o).
// Contraction is pinned (one explicit fma) so that every kernel that calls this -- vectorised or
// not -- rounds identically: the persistent and the per-step recurrences are compared bit for bit.
__device__ __forceinline__ void lstm_cell_fwd(float zi, float zj, float zf, float zo, float cp, float& cn,
                                              float& hn) {
#pragma clang fp contract(off)
    const float ij = d2p_sigmoid(zi) * d2p_tanh(zj);
    const float c1 = fmaf(cp, d2p_sigmoid(zf + D2P_FORGET_BIAS), ij);
    cn = c1;
    hn = d2p_tanh(c1) * d2p_sigmoid(zo);
}
__device__ __forceinline__ void lstm_gate_fwd4(const f4& zi, const f4& zj, const f4& zf, const f4& zo,
                                               const f4& cp, f4& cn, f4& hn) {
#pragma unroll
    for (int q = 0; q < 4; ++q) lstm_cell_fwd(zi.v[q], zj.v[q], zf.v[q], zo.v[q], cp.v[q], cn.v[q], hn.v[q]);
}

// Given pre-activations z, c_prev, c (= c after the step), the total gradient dh wrt the
// emitted/next-state h and dc wrt the state c: pre-activation gradients and dc wrt c_prev.
__device__ __forceinline__ void lstm_cell_bwd(float zi, float zj, float zf, float zo, float cp, float cc,
                                              float dh, float dcv, float& gi, float& gj, float& gf, float& go,
                                              float& dcn) {
#pragma clang fp contract(off)
    const float i = d2p_sigmoid(zi);
    const float j = d2p_tanh(zj);
    const float f = d2p_sigmoid(zf + D2P_FORGET_BIAS);
    const float og = d2p_sigmoid(zo);
    const float tc = d2p_tanh(cc);
    const float d_o = dh * tc;
    const float dct = fmaf(dh * og, 1.f - tc * tc, dcv);
    gi = dct * j * i * (1.f - i);
    gj = dct * i * (1.f - j * j);
    gf = dct * cp * f * (1.f - f);
    go = d_o * og * (1.f - og);
    dcn = dct * f;
}
__device__ __forceinline__ void lstm_gate_bwd4(const f4& zi, const f4& zj, const f4& zf, const f4& zo,
                                               const f4& cp, const f4& cc, const f4& dh, const f4& dcv,
                                               f4& gi, f4& gj, f4& gf, f4& go, f4& dcn) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        lstm_cell_bwd(zi.v[q], zj.v[q], zf.v[q], zo.v[q], cp.v[q], cc.v[q], dh.v[q], dcv.v[q], gi.v[q], gj.v[q],
                      gf.v[q], go.v[q], dcn.v[q]);
}

// The same cell backward in two parts for kernels that know z, c before they know dh (the persistent
// backward kernel computes part 1 while the dz.Wh^T product is still running): part 1 = everything that
// does not depend on dh / dc, part 2 = six multiplies.  Products are grouped differently from
// lstm_cell_bwd (1-ulp differences).
struct LstmCellBwdPre { float a_dct, b_i, b_j, b_f, b_o, f; };
__device__ __forceinline__ LstmCellBwdPre lstm_cell_bwd_pre(float zi, float zj, float zf, float zo, float cp, float cc) {
#pragma clang fp contract(off)
    const float i = d2p_sigmoid(zi);
    const float j = d2p_tanh(zj);
    const float f = d2p_sigmoid(zf + D2P_FORGET_BIAS);
    const float og = d2p_sigmoid(zo);
    const float tc = d2p_tanh(cc);
    LstmCellBwdPre q;
    q.a_dct = og * (1.f - tc * tc);      // d c_t / d h
    q.b_i = j * i * (1.f - i);
    q.b_j = i * (1.f - j * j);
    q.b_f = cp * f * (1.f - f);
    q.b_o = tc * og * (1.f - og);
    q.f = f;
    return q;
}
__device__ __forceinline__ void lstm_cell_bwd_post(const LstmCellBwdPre& q, float dh, float dcv, float& gi, float& gj,
                                                   float& gf, float& go, float& dcn) {
#pragma clang fp contract(off)
    const float dct = fmaf(dh, q.a_dct, dcv);
    gi = dct * q.b_i;
    gj = dct * q.b_j;
    gf = dct * q.b_f;
    go = dh * q.b_o;
    dcn = dct * q.f;
}
