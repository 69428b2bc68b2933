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

// One element (a float4) of the packed operands; idx is the float4 index in the packed image.
// Forward B operand: Wf[((ct*KC + kc)*2 + cs)*64 + lane], ct = 8-unit column tile, local column
// j = cs*16 + (lane&15) -> gate j>>3, unit ct*8 + (j&7); k = kc*16 + 4*(lane>>4) + jj
__device__ __forceinline__ float4 d2p_pack_w_fwd_elem(int U, const float* __restrict__ Wh, long idx) {
    const int KC = U >> 4;
    const int lane = (int)(idx & 63);
    const int cs = (int)((idx >> 6) & 1);
    const long r = idx >> 7;
    const int kc = (int)(r % KC), ct = (int)(r / KC);
    const int j = cs * 16 + (lane & 15);
    const long col = (long)(j >> 3) * U + ct * 8 + (j & 7);
    const int k = kc * 16 + 4 * (lane >> 4);
    const long ld = 4L * U;
    return make_float4(Wh[(k + 0) * ld + col], Wh[(k + 1) * ld + col], Wh[(k + 2) * ld + col], Wh[(k + 3) * ld + col]);
}
// Backward B operand (Wh^T): Wb[(nt*KC4 + kc)*64 + lane], n = nt*16 + (lane&15) (unit),
// k = kc*16 + 4*(lane>>4) (gate column): value Wh[n][k..k+3] -- a straight float4 copy.
__device__ __forceinline__ float4 d2p_pack_w_bwd_elem(int U, const float* __restrict__ Wh, long idx) {
    const int KC4 = U >> 2;
    const int lane = (int)(idx & 63);
    const long r = idx >> 6;
    const int kc = (int)(r % KC4), nt = (int)(r / KC4);
    const int n = nt * 16 + (lane & 15), k = kc * 16 + 4 * (lane >> 4);
    return *reinterpret_cast<const float4*>(Wh + (long)n * 4 * U + k);
}
// A operand from a row-major [M, K] matrix: Af[(rs*KCx + kc)*64 + lane], row = rs*16 + (lane&15);
// rows >= M are zeros
__device__ __forceinline__ float4 d2p_pack_rows_elem(int M, int K, const float* __restrict__ X, long idx) {
    const int KCx = K >> 4;
    const int lane = (int)(idx & 63);
    const long r = idx >> 6;
    const int kc = (int)(r % KCx), rs = (int)(r / KCx);
    const int row = rs * 16 + (lane & 15), k = kc * 16 + 4 * (lane >> 4);
    return row < M ? *reinterpret_cast<const float4*>(X + (long)row * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// persistent-sequence back end (lstm_persist.hip)
int d2p_lstm_is_persistent_enabled();
bool d2p_lstm_persist_fwd_ok(int M, int U, int n_steps);
bool d2p_lstm_persist_bwd_ok(int M, int U, int n_steps);
size_t d2p_lstm_persist_ws_bytes(int M, int U);
// one sequence's arguments (those of d2p_lstm_seq_fwd / _bwd, include/d2p.h; ws holds d2p_lstm_persist_ws_bytes)
struct PsFwdCall {
    int M, U, n_steps;
    float* z; long zrs, zts;
    const float* Wh; const float* h0; const float* c0; const int* lens;
    float* hout; float* cs; float* h_final; float* c_final;
    float* ws;
    // optional ("direct" launch: no preparation launch): a dedicated flag buffer of D2P_LSTM_FLAG_WORDS words,
    // zeroed ONCE by the caller, and the caller's epoch for it (raised by more than n_steps per launch)
    unsigned* flags; unsigned epoch;
    const float* wpack;      // optional with flags: the caller's packed forward image of Wh (d2p_lstm_pack_weights)
    // optional with flags and lens (length-sorted launch of the wide-tile kernel, as PsBwdCall): rowmap[v] (device, M
    // entries) = the caller's row at position v of the order by decreasing length; slab_steps (HOST, ceil(M/16) entries)
    const int* rowmap; const int* slab_steps;
};
struct PsBwdCall {
    int M, U, n_steps;
    const float* z; long zrs, zts;
    const float* Wh; const float* c0; const int* lens; const float* cs;
    const float* dhout; const float* dh_final; const float* dc_final;
    float* dz; float* dh0; float* dc0;
    float* ws;
    float* db;          // optional: bias gradient [4U] = column sums of dz, produced inside the launch
    unsigned* flags; unsigned epoch;      // as PsFwdCall
    const float* wpack;                   // ... the packed backward (Wh^T) image
    // optional with flags (length-sorted launch): rowmap[v] (device, M entries) = the caller's row at position v of the
    // order by decreasing length; slab_steps (HOST, ceil(M/16) entries) = the longest length among rows 16s .. 16s+15 of
    // that order.  Domains then run only as many passes as their longest row needs.
    const int* rowmap; const int* slab_steps;
};
int d2p_lstm_persist_fwd(const PsFwdCall& q, hipStream_t st);
int d2p_lstm_persist_bwd(const PsBwdCall& q, hipStream_t st);
// two independent sequences (same U) on disjoint workgroups of ONE launch; *_pair_ok: both shapes are taken and
// sharing the chip is expected to beat the two launches back to back
bool d2p_lstm_persist_fwd_pair_ok(int Ma, int Ta, int Mb, int Tb, int U);
bool d2p_lstm_persist_bwd_pair_ok(int Ma, int Ta, int Mb, int Tb, int U);
int d2p_lstm_persist_fwd_pair(const PsFwdCall& qa, const PsFwdCall& qb, hipStream_t st);
// wide-tile forward kernel (16 units per column tile, 8 row domains at U = 512): one to three sequences per launch,
// length-sorted where a sequence brings rowmap / slab_steps; direct launches only (flags)
bool d2p_lstm_persist_fwd_wide_ok(int n, const PsFwdCall* q);
int d2p_lstm_persist_fwd_wide(int n, const PsFwdCall* q, hipStream_t st);
bool d2p_lstm_try_wide_fwd(int n, const d2p_lstm_fwd_desc* d, hipStream_t st, int* rc);
int d2p_lstm_persist_bwd_pair(const PsBwdCall& qa, const PsBwdCall& qb, hipStream_t st);
// the backward kernel also takes three (the three decoders)
bool d2p_lstm_persist_bwd_triple_ok(const int M[3], const int T[3], int U);
int d2p_lstm_persist_bwd_triple(const PsBwdCall q[3], hipStream_t st);
// multi entry points: two sequences as one persistent launch when that is possible and expected to pay (lstm.hip)
bool d2p_lstm_try_pair_fwd(const d2p_lstm_fwd_desc* d, hipStream_t st, int* rc);
bool d2p_lstm_try_pair_bwd(const d2p_lstm_bwd_desc* d, hipStream_t st, int* rc);
bool d2p_lstm_try_triple_bwd(const d2p_lstm_bwd_desc* d, hipStream_t st, int* rc);
// one sequence by descriptor (d2p_lstm_seq_bwd + the optional bias gradient)
int d2p_lstm_seq_bwd_desc(const d2p_lstm_bwd_desc* d, d2p_stream_t stream);
int d2p_lstm_seq_fwd_desc(const d2p_lstm_fwd_desc* d, d2p_stream_t stream);
int d2p_lstm_db_colsum(const d2p_lstm_bwd_desc* d, d2p_stream_t stream);
