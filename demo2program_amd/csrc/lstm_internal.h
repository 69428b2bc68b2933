// Internal interface between the three LSTM sequence back ends behind d2p_lstm_seq_fwd/_bwd
// (include/d2p.h): lstm.hip (dispatch + generic GEMM/gate path), lstm_step.hip (one fused launch
// per step) and lstm_persist.hip (one persistent launch per sequence).
#pragma once
#include "common.h"

// fragment-major operand packing (lstm_step.hip)
int d2p_lstm_pack_w_fwd(int U, const float* Wh, float* Wf, hipStream_t st);
int d2p_lstm_pack_w_bwd(int U, const float* Wh, float* Wb, hipStream_t st);
int d2p_lstm_pack_rows(int M, int K, int total_rs, const float* X, float* Af, hipStream_t st);

// fragment-major offset (in floats) of (row, k..k+3), k % 4 == 0, for a matrix with KCx 16-wide
// k chunks: a [16 rows x 16 k] block is 64 lanes x float4, lane l = (row l&15, k 4*(l>>4)..+3)
__host__ __device__ __forceinline__ long d2p_frag_off(int row, int k, int KCx) {
    return ((((long)(row >> 4) * KCx + (k >> 4)) * 64) + (((k & 15) >> 2) << 4) + (row & 15)) * 4;
}

// persistent-sequence back end (lstm_persist.hip)
int d2p_lstm_is_persistent_enabled();
bool d2p_lstm_persist_fwd_ok(int M, int U, int n_steps);
bool d2p_lstm_persist_bwd_ok(int M, int U, int n_steps);
size_t d2p_lstm_persist_ws_bytes(int M, int U);
int d2p_lstm_persist_fwd(int M, int U, int n_steps, float* z, long zrs, long zts, const float* Wh,
                         const float* h0, const float* c0, const int* lens, float* hout, float* cs,
                         float* h_final, float* c_final, float* ws, hipStream_t st);
int d2p_lstm_persist_bwd(int M, int U, int n_steps, const float* z, long zrs, long zts, const float* Wh,
                         const float* c0, const int* lens, const float* cs, const float* dhout,
                         const float* dh_final, const float* dc_final, float* dz, float* dh0,
                         float* dc0, float* ws, hipStream_t st);
