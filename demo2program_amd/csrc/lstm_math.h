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

// c' = c*sigmoid(f+1) + sigmoid(i)*tanh(j);  h' = tanh(c')*sigmoid(o)
__device__ __forceinline__ void lstm_gate_fwd4(const f4& zi, const f4& zj, const f4& zf, const f4& zo,
                                               const f4& cp, f4& cn, f4& hn) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float c1 = cp.v[q] * d2p_sigmoid(zf.v[q] + D2P_FORGET_BIAS) +
                         d2p_sigmoid(zi.v[q]) * d2p_tanh(zj.v[q]);
        cn.v[q] = c1;
        hn.v[q] = d2p_tanh(c1) * d2p_sigmoid(zo.v[q]);
    }
}

// Given pre-activations z, c_prev, c (= c after the step), the total gradient dh wrt the
// emitted/next-state h and dc wrt the state c: pre-activation gradients and dc wrt c_prev.
__device__ __forceinline__ void lstm_gate_bwd4(const f4& zi, const f4& zj, const f4& zf, const f4& zo,
                                               const f4& cp, const f4& cc, const f4& dh, const f4& dcv,
                                               f4& gi, f4& gj, f4& gf, f4& go, f4& dcn) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float i = d2p_sigmoid(zi.v[q]);
        const float j = d2p_tanh(zj.v[q]);
        const float f = d2p_sigmoid(zf.v[q] + D2P_FORGET_BIAS);
        const float og = d2p_sigmoid(zo.v[q]);
        const float tc = d2p_tanh(cc.v[q]);
        const float d_o = dh.v[q] * tc;
        const float dct = dcv.v[q] + dh.v[q] * og * (1.f - tc * tc);
        gi.v[q] = dct * j * i * (1.f - i);
        gj.v[q] = dct * i * (1.f - j * j);
        gf.v[q] = dct * cp.v[q] * f * (1.f - f);
        go.v[q] = d_o * og * (1.f - og);
        dcn.v[q] = dct * f;
    }
}
