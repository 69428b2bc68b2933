// Greedy decoding (evaluation path): BasicDecoder + GreedyEmbeddingHelper under
// dynamic_decode(impute_finished=False, maximum_iterations=L)
// (models/model_full.py:424-435,465-490; reached from run_test / evaler, not the train step).
//
// [TF-1.3] semantics restated: start input = embedding[start_id] (start_id = token_dim, the last
// in-range row); per step logits = Dense(cell(x)), sample = argmax(logits) (first index on
// ties); every row -- finished or not -- feeds embedding[sample] to the next step; row r's
// length is the first step (1-based) at which it sampled end_id (L if never); the loop stops
// when all rows are finished and the emitted logits are zero-padded to L afterwards.
//
// The token table is tiny (<= 51 rows), so the input projection of EVERY possible token,
// table_proj = embedding·Wx + b  [V+1, 4U], is computed once by the caller and each step just
// gathers rows of it: no per-step input GEMM.  All L steps are issued without host round trips;
// a finalize kernel derives the lengths, the number of steps TF would have run, and zeroes the
// logits / ids past it -- identical results, no device->host synchronisation inside the loop.
#include "common.h"

__global__ void __launch_bounds__(256)
greedy_gather_kernel(int M, int W4, int rows, const int* ids, int fixed_id, const float4* table,
                     float4* z) {
    const long total = (long)M * W4;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        const int m = (int)(idx / W4), e = (int)(idx - (long)m * W4);
        int id = ids ? ids[m] : fixed_id;
        id = id < 0 ? 0 : (id >= rows ? rows - 1 : id);
        z[idx] = table[(long)id * W4 + e];
    }
}

// one wave per row: first index of the maximum (tf.argmax tie rule)
__global__ void __launch_bounds__(256)
argmax_rows_kernel(int rows, int V, const float* x, long ld, int* out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
        const float f = x[(long)r * ld + v];
        if (f > best || (f == best && v < bi)) { best = f; bi = v; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) out[r] = bi;
}

extern "C" int d2p_argmax_rows(int rows, int V, const float* x, long ld, int* out,
                               d2p_stream_t stream) {
    D2P_REQUIRE(rows >= 0 && V > 0, D2P_EINVAL, "argmax_rows: bad sizes");
    if (rows == 0) return D2P_OK;
    D2P_REQUIRE(x && out, D2P_EINVAL, "argmax_rows: null pointer");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, as_stream(stream),
                       rows, V, x, ld, out);
    D2P_LAUNCH_CHECK("argmax_rows");
    return D2P_OK;
}

// lengths[m] = 1 + first t with ids[t][m] == end_id (L if none); n_run = max_m lengths
__global__ void __launch_bounds__(256)
greedy_lengths_kernel(int M, int L, int end_id, const int* ids, int* lengths, int* n_run) {
    __shared__ int red[256];
    int mx = 0;
    for (int m = threadIdx.x; m < M; m += 256) {
        int len = L;
        for (int t = 0; t < L; ++t)
            if (ids[(long)t * M + m] == end_id) { len = t + 1; break; }
        lengths[m] = len;
        mx = max(mx, len);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) n_run[0] = red[0];
}

__global__ void __launch_bounds__(256)
greedy_zero_tail_kernel(int M, int L, int V, const int* n_run, float* logits, int* ids) {
    const int n = n_run[0];
    const long total = (long)L * M * V;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        const int t = (int)(idx / ((long)M * V));
        if (t >= n) {
            logits[idx] = 0.f;
            if (idx % V == 0) ids[idx / V] = 0;
        }
    }
}

extern "C" size_t d2p_greedy_ws_bytes(int M, int U, int V) {
    (void)V;
    if (M <= 0 || U <= 0) return 0;
    return ((size_t)M * 4 * U + 4 * (size_t)M * U) * sizeof(float) + 64;
}

extern "C" int d2p_greedy_decode(int M, int U, int V, int L, const float* table_proj,
                                 const float* Wh, const float* proj, const float* h0,
                                 const float* c0, int start_id, int end_id, float* logits, int* ids,
                                 int* lengths, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    D2P_REQUIRE(M >= 0 && U > 0 && V > 0 && L >= 0, D2P_EINVAL, "greedy: bad sizes");
    if (M == 0 || L == 0) return D2P_OK;
    D2P_REQUIRE(table_proj && Wh && proj && h0 && c0 && logits && ids && lengths, D2P_EINVAL,
                "greedy: null pointer");
    D2P_REQUIRE(U % 4 == 0 && ((((uintptr_t)table_proj | (uintptr_t)ws) & 15) == 0), D2P_EALIGN,
                "greedy: needs U %% 4 == 0 and 16-byte aligned table / workspace");
    D2P_REQUIRE(ws && ws_bytes >= d2p_greedy_ws_bytes(M, U, V), D2P_EWS, "greedy: workspace too small");
    hipStream_t st = as_stream(stream);
    const size_t MU = (size_t)M * U;
    float* z = (float*)ws;
    float* hb[2] = {z + (size_t)M * 4 * U, z + (size_t)M * 4 * U + MU};
    float* cb[2] = {hb[1] + MU, hb[1] + 2 * MU};
    int* n_run = (int*)(cb[1] + MU);
    const float* h_prev = h0;
    const float* c_prev = c0;
    const int W4 = U;    // 4U floats = U float4
    long gb = ((long)M * W4 + 255) / 256;
    if (gb > 2048) gb = 2048;
    for (int t = 0; t < L; ++t) {
        hipLaunchKernelGGL(greedy_gather_kernel, dim3((int)gb), dim3(256), 0, st, M, W4, V + 1,
                           t ? ids + (size_t)(t - 1) * M : (const int*)nullptr, start_id,
                           (const float4*)table_proj, (float4*)z);
        D2P_LAUNCH_CHECK("greedy_gather");
        int rc = d2p_gemm_f32_nn(M, 4 * U, U, h_prev, U, Wh, 4L * U, z, 4L * U, nullptr, 0, 1, nullptr, 0,
                                 stream);
        if (rc) return rc;
        rc = d2p_lstm_gate_fwd(M, U, z, 4L * U, c_prev, nullptr, nullptr, t, cb[t & 1], nullptr,
                               hb[t & 1], stream);
        if (rc) return rc;
        float* lg = logits + (size_t)t * M * V;
        rc = d2p_gemm_f32_nn(M, V, U, hb[t & 1], U, proj, V, lg, V, nullptr, 0, 0, nullptr, 0, stream);
        if (rc) return rc;
        rc = d2p_argmax_rows(M, V, lg, V, ids + (size_t)t * M, stream);
        if (rc) return rc;
        h_prev = hb[t & 1];
        c_prev = cb[t & 1];
    }
    hipLaunchKernelGGL(greedy_lengths_kernel, dim3(1), dim3(256), 0, st, M, L, end_id, ids, lengths,
                       n_run);
    D2P_LAUNCH_CHECK("greedy_lengths");
    long zb = ((long)L * M * V + 255) / 256;
    if (zb > 2048) zb = 2048;
    hipLaunchKernelGGL(greedy_zero_tail_kernel, dim3((int)zb), dim3(256), 0, st, M, L, V, n_run, logits,
                       ids);
    D2P_LAUNCH_CHECK("greedy_zero_tail");
    return D2P_OK;
}

// ---- scheduled sampling (models/model_full.py:59-67,414-423; seq2seq.ScheduledEmbeddingTrainingHelper)
// For each decoder row after step t:  with probability p_sample the next input is a draw from
// Categorical(softmax(logits_t)), otherwise the ground-truth token of step t+1.  The draw is
// Gumbel-max over counter-based Philox4x32-10 noise keyed by (seed, step counter, row, token):
// no RNG state to carry, replays of a captured graph get fresh noise because the step counter
// lives in device memory.  p_sample is read from device memory too (it follows
// polynomial_decay(global_step) and must change between graph replays).
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) {      // (0, 1): never 0 or 1
    return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// one wave per row; rng[0] = seed, rng[1] = step counter (both uint64 in device memory)
__global__ void __launch_bounds__(256)
sched_sample_kernel(int M, int V, const float* __restrict__ logits, const int* __restrict__ gt_next,
                    const float* __restrict__ p_sample, const unsigned long long* __restrict__ rng, int t,
                    int* __restrict__ next_ids, int* __restrict__ sampled_flag) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const unsigned long long seed = rng[0], ctr = rng[1];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    // Bernoulli draw: stream (ctr, t, row, 0xFFFFFFFF)
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32) ^ ((uint32_t)t << 16), (uint32_t)r, 0xFFFFFFFFu};
    philox4x32_10(c, k0, k1);
    const bool take = u01(c[0]) < p_sample[0];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
        uint32_t d[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32) ^ ((uint32_t)t << 16), (uint32_t)r, (uint32_t)v};
        philox4x32_10(d, k0, k1);
        const float g = -logf(-logf(u01(d[0])));
        const float f = logits[(long)r * V + v] + g;
        if (f > best || (f == best && v < bi)) { best = f; bi = v; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
        next_ids[r] = take ? bi : gt_next[r];
        if (sampled_flag) sampled_flag[r] = take ? 1 : 0;
    }
}

extern "C" int d2p_sched_sample(int M, int V, const float* logits, const int* gt_next, const float* p_sample_dev,
                                const void* rng_dev, int t, int* next_ids, int* sampled_flag,
                                d2p_stream_t stream) {
    D2P_REQUIRE(M >= 0 && V > 0 && t >= 0 && t < 65536, D2P_EINVAL, "sched_sample: bad sizes");
    if (M == 0) return D2P_OK;
    D2P_REQUIRE(logits && gt_next && p_sample_dev && rng_dev && next_ids, D2P_EINVAL, "sched_sample: null pointer");
    hipLaunchKernelGGL(sched_sample_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, as_stream(stream), M, V, logits,
                       gt_next, p_sample_dev, (const unsigned long long*)rng_dev, t, next_ids, sampled_flag);
    D2P_LAUNCH_CHECK("sched_sample");
    return D2P_OK;
}
