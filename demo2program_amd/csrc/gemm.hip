// K5: dense fp32 MFMA GEMM entry points + column sum (include/d2p.h).
#define D2P_GEMM_CORUN_TILES
#include "gemm_core.h"

static inline int vec_ok(const float* p, long ld) {
    return (((uintptr_t)p & 15) == 0 && (ld % 4) == 0) ? 1 : 0;
}

static int check_gemm_args(int M, int N, int K, const float* A, const float* B, float* C,
                           int act) {
    D2P_REQUIRE(M >= 0 && N >= 0 && K >= 0, D2P_EINVAL, "gemm: negative dimension M=%d N=%d K=%d", M, N, K);
    D2P_REQUIRE(act == 0 || act == 1, D2P_EINVAL, "gemm: unknown act %d", act);
    if (M == 0 || N == 0) return D2P_OK;
    D2P_REQUIRE(C != nullptr, D2P_EINVAL, "gemm: C is null");
    D2P_REQUIRE(K == 0 || (A != nullptr && B != nullptr), D2P_EINVAL, "gemm: A or B is null");
    return D2P_OK;
}

extern "C" int d2p_gemm_set_option(int bk32) {
    g_gemm_bk32 = (bk32 & 1) ? 1 : 0;
    g_gemm_small_ksr = (bk32 & 2) ? 0 : 1;     // bit 1: switch the small-problem 32x32 tile off
    g_gemm_nosel = (bk32 & 4) ? 0 : 1;         // bit 2: keep the select-at-store loaders for every K
    g_gemm_no_bk32 = (bk32 & 16) ? 1 : 0;      // bit 4 (experiment): never the 32-deep slabs
    g_gemm_fold = (bk32 & 32) ? 0 : 1;         // bit 5: split-K combine as a separate launch (round 2's form)
    g_gemm_dma_big = (bk32 & 8) ? 1 : 0;       // bit 3 (experiment): large dense GEMMs on the persistent LDS-DMA kernel
    g_gemm_dma_grid = bk32 >> 8;               // bits 8..: persistent grid of the LDS-DMA kernel (0 = automatic)
    return D2P_OK;
}
extern "C" int d2p_gemm_set_corun(int on) {
    g_gemm_corun = on ? 1 : 0;
    return D2P_OK;
}
extern "C" int d2p_gemm_force_plan(int tile, int splits) {
    g_gemm_force_tile = tile;
    g_gemm_force_split = splits;
    return D2P_OK;
}

extern "C" size_t d2p_gemm_ws_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return d2p_plan_ws_bytes(M, N, K);
}

extern "C" int d2p_gemm_f32_nn(int M, int N, int K, const float* A, long lda, const float* B,
                               long ldb, float* C, long ldc, const float* bias, int act,
                               int accumulate, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    int rc = check_gemm_args(M, N, K, A, B, C, act);
    if (rc) return rc;
    DenseKC al{A, lda, M, vec_ok(A, lda)};
    DenseXC bl{B, ldb, N, vec_ok(B, ldb)};
    EpiDense ep{C, ldc, bias, act, accumulate};
    return d2p_launch_gemm(al, bl, ep, M, N, K, ws, ws_bytes, as_stream(stream), "gemm_nn");
}

// Strided batch of nb1 x nb0 equally shaped problems in ONE launch (grid.y): problem (i, j) uses
// A + i*sA1 + j*sA0 etc.  kind: 0 = nn, 1 = nt, 2 = tn.  No split-K (no workspace).
extern "C" int d2p_gemm_f32_batched(int kind, int nb1, int nb0, int M, int N, int K, const float* A, long lda,
                                    long sA1, long sA0, const float* B, long ldb, long sB1, long sB0, float* C,
                                    long ldc, long sC1, long sC0, const float* bias, long sbias1, long sbias0,
                                    int act, int accumulate, d2p_stream_t stream) {
    int rc = check_gemm_args(M, N, K, A, B, C, act);
    if (rc) return rc;
    D2P_REQUIRE(kind >= 0 && kind <= 2 && nb1 > 0 && nb0 > 0 && (long)nb1 * nb0 <= 65535, D2P_EINVAL,
                "gemm_batched: kind=%d batch=%dx%d", kind, nb1, nb0);
    const int va = vec_ok(A, lda) && sA1 % 4 == 0 && sA0 % 4 == 0;
    const int vb = vec_ok(B, ldb) && sB1 % 4 == 0 && sB0 % 4 == 0;
    EpiDense ep{C, ldc, bias, act, accumulate, sC1, sC0, sbias1, sbias0, nb0};
    const int batch = nb1 * nb0;
    hipStream_t st = as_stream(stream);
    if (kind == 0) {
        DenseKC al{A, lda, M, va, sA1, sA0, nb0};
        DenseXC bl{B, ldb, N, vb, sB1, sB0, nb0};
        return d2p_launch_gemm(al, bl, ep, M, N, K, nullptr, 0, st, "gemm_batched_nn", D2P_PROF_GEMM, batch);
    }
    if (kind == 1) {
        DenseKC al{A, lda, M, va, sA1, sA0, nb0};
        DenseKC bl{B, ldb, N, vb, sB1, sB0, nb0};
        return d2p_launch_gemm(al, bl, ep, M, N, K, nullptr, 0, st, "gemm_batched_nt", D2P_PROF_GEMM, batch);
    }
    DenseXC al{A, lda, M, va, sA1, sA0, nb0};
    DenseXC bl{B, ldb, N, vb, sB1, sB0, nb0};
    return d2p_launch_gemm(al, bl, ep, M, N, K, nullptr, 0, st, "gemm_batched_tn", D2P_PROF_GEMM, batch);
}

extern "C" int d2p_gemm_f32_nt(int M, int N, int K, const float* A, long lda, const float* B,
                               long ldb, float* C, long ldc, const float* bias, int act,
                               int accumulate, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    int rc = check_gemm_args(M, N, K, A, B, C, act);
    if (rc) return rc;
    DenseKC al{A, lda, M, vec_ok(A, lda)};
    DenseKC bl{B, ldb, N, vec_ok(B, ldb)};   // B is [N,K]: K contiguous
    EpiDense ep{C, ldc, bias, act, accumulate};
    return d2p_launch_gemm(al, bl, ep, M, N, K, ws, ws_bytes, as_stream(stream), "gemm_nt");
}

// Products over a LIST of rows: row x of the result is computed from row rows[x] of A and stored at row rows[x]
// of C (M = number of listed rows; rows of C that are not listed are left untouched).  kind 0: C = A . B (B [K, N]),
// kind 1: C = A . B^T (B [N, K]).  For the active rows of padded time-major batches.
extern "C" int d2p_gemm_f32_rows(int kind, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                                 float* C, long ldc, const float* bias, const int* rows, void* ws, size_t ws_bytes,
                                 d2p_stream_t stream) {
    int rc = check_gemm_args(M, N, K, A, B, C, 0);
    if (rc) return rc;
    D2P_REQUIRE(kind == 0 || kind == 1, D2P_EINVAL, "gemm_rows: kind %d", kind);
    if (M == 0 || N == 0) return D2P_OK;
    D2P_REQUIRE(rows != nullptr, D2P_EINVAL, "gemm_rows: null row list");
    GatherKC al{A, lda, M, vec_ok(A, lda), rows};
    EpiScatterRows ep{C, ldc, bias, rows};
    if (kind == 0) {
        DenseXC bl{B, ldb, N, vec_ok(B, ldb)};
        return d2p_launch_gemm(al, bl, ep, M, N, K, ws, ws_bytes, as_stream(stream), "gemm_rows_nn");
    }
    DenseKC bl{B, ldb, N, vec_ok(B, ldb)};
    return d2p_launch_gemm(al, bl, ep, M, N, K, ws, ws_bytes, as_stream(stream), "gemm_rows_nt");
}

extern "C" int d2p_gemm_f32_tn(int M, int N, int K, const float* A, long lda, const float* B,
                               long ldb, float* C, long ldc, const float* bias, int act,
                               int accumulate, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    int rc = check_gemm_args(M, N, K, A, B, C, act);
    if (rc) return rc;
    DenseXC al{A, lda, M, vec_ok(A, lda)};   // A is [K,M]: M contiguous
    DenseXC bl{B, ldb, N, vec_ok(B, ldb)};
    EpiDense ep{C, ldc, bias, act, accumulate};
    return d2p_launch_gemm(al, bl, ep, M, N, K, ws, ws_bytes, as_stream(stream), "gemm_tn");
}

// C = A^T B over a LIST of K rows: C[m, n] (+)= sum_x A[rowsA[x], m] * B[rowsB[x], n], x < K (A and B row-major,
// M resp. N contiguous).  Weight gradients X^T dZ of a padded time-major batch run over the rows inside their
// sequences only (the others are zeros in dZ); rowsA != rowsB serves dWh = sum_t h[t-1]^T dz[t] (rowsA = rowsB - M).
// Fast path (16-byte loads, no select in the K loop) when K is a multiple of 32: callers pad the lists with the
// index of a row that is zero in B and finite in A.
extern "C" int d2p_gemm_f32_tn_rows(int M, int N, int K, const float* A, long lda, const int* rowsA, const float* B,
                                    long ldb, const int* rowsB, float* C, long ldc, int accumulate, void* ws,
                                    size_t ws_bytes, d2p_stream_t stream) {
    int rc = check_gemm_args(M, N, K, A, B, C, 0);
    if (rc) return rc;
    if (M == 0 || N == 0) return D2P_OK;
    D2P_REQUIRE(K == 0 || (rowsA && rowsB), D2P_EINVAL, "gemm_tn_rows: null row list");
    GatherXC al{A, lda, M, vec_ok(A, lda), rowsA};
    GatherXC bl{B, ldb, N, vec_ok(B, ldb), rowsB};
    EpiDense ep{C, ldc, nullptr, 0, accumulate};
    return d2p_launch_gemm(al, bl, ep, M, N, K, ws, ws_bytes, as_stream(stream), "gemm_tn_rows");
}

// ---- column sum (bias gradients): two-stage, deterministic --------------------------
// stage 1: block (cb, s) sums rows r = s*4+rl, step S*4, of 64 columns -> part[s][c];
// stage 2: block per 16 columns x 16 lanes over the S partials, fixed-order tree.
// S adapts so that ~2048 workgroups stream the matrix (HBM-bound: rows*cols*4 bytes).
static inline int colsum_S(int rows, int cols) {
    const int cb = ceil_div(cols, 64);
    int S = 2048 / cb;
    const int cap = ceil_div(rows, 32);          // >= 8 rows per thread
    if (S > cap) S = cap;
    if (S > 256) S = 256;
    if (S < 1) S = 1;
    return S;
}

__global__ void __launch_bounds__(256)
colsum_stage1(int rows, int cols, const float* X, long ld, float* part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < cols) {
        const int step = gridDim.y * 4;
        int r = blockIdx.y * 4 + rl;
        for (; r + 3 * step < rows; r += 4 * step) {      // 4 independent loads in flight
            s0 += X[(long)r * ld + c];
            s1 += X[(long)(r + step) * ld + c];
            s2 += X[(long)(r + 2 * step) * ld + c];
            s3 += X[(long)(r + 3 * step) * ld + c];
        }
        for (; r < rows; r += step) s0 += X[(long)r * ld + c];
    }
    red[rl][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < cols)
        part[(long)blockIdx.y * cols + c] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// fold > 1: the matrix was viewed as [rows/fold, cols*fold]; out[c] = sum_g colsum[g*cols + c]
// Block = 16 columns x 16 partial-row lanes (all S*fold loads of a column are independent and in
// flight together; 128 workgroups for 2048 columns instead of 32), fixed-order tree in LDS.
__global__ void __launch_bounds__(256)
colsum_stage2(int cols, int fold, int S, const float* part, float* out) {
    __shared__ float red[16][17];
    const int wc = cols * fold;                       // width of the partial rows
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < cols) {
        const int n = S * fold;                       // partial entries of this column
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int e = pl;
        for (; e + 48 < n; e += 64) {
            s0 += part[(long)(e / fold) * wc + (e % fold) * cols + c];
            s1 += part[(long)((e + 16) / fold) * wc + ((e + 16) % fold) * cols + c];
            s2 += part[(long)((e + 32) / fold) * wc + ((e + 32) % fold) * cols + c];
            s3 += part[(long)((e + 48) / fold) * wc + ((e + 48) % fold) * cols + c];
        }
        for (; e < n; e += 16) s0 += part[(long)(e / fold) * wc + (e % fold) * cols + c];
        s = (s0 + s1) + (s2 + s3);
    }
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        out[c] = t;
    }
}

extern "C" size_t d2p_colsum_ws_bytes(int rows, int cols) {
    (void)rows;
    return cols > 0 ? (size_t)256 * cols * sizeof(float) : 0;
}

extern "C" int d2p_colsum_f32(int rows, int cols, const float* X, long ld, float* out, void* ws,
                              size_t ws_bytes, d2p_stream_t stream) {
    D2P_REQUIRE(rows >= 0 && cols >= 0, D2P_EINVAL, "colsum: negative size");
    if (cols == 0) return D2P_OK;
    D2P_REQUIRE(out && (rows == 0 || X), D2P_EINVAL, "colsum: null pointer");
    D2P_REQUIRE(ws && ws_bytes >= d2p_colsum_ws_bytes(rows, cols), D2P_EWS,
                "colsum: workspace too small (%zu < %zu)", ws_bytes, d2p_colsum_ws_bytes(rows, cols));
    hipStream_t st = as_stream(stream);
    float* part = (float*)ws;
    // narrow contiguous matrices (conv channels 16/32): view [rows, cols] as
    // [rows/fold, cols*fold] so that all 64 lanes of a wave read one contiguous 256 B row
    int fold = 1;
    if (cols < 64 && 64 % cols == 0 && ld == cols && rows % (64 / cols) == 0) fold = 64 / cols;
    const int vrows = rows / fold, vcols = cols * fold;
    const int S = colsum_S(vrows, vcols);
    hipLaunchKernelGGL(colsum_stage1, dim3(ceil_div(vcols, 64), S), dim3(256), 0, st, vrows, vcols, X,
                       (long)(fold > 1 ? vcols : ld), part);
    D2P_LAUNCH_CHECK("colsum_stage1");
    hipLaunchKernelGGL(colsum_stage2, dim3(ceil_div(cols, 16)), dim3(256), 0, st, cols, fold, S, part,
                       out);
    D2P_LAUNCH_CHECK("colsum_stage2");
    return D2P_OK;
}

// ---- K6 backward: embedding scatter-add as a one-hot TN GEMM ---------------------------
// dtable[v, e] = sum_i [ids[i] == v] * dout[i, e]  =  OneHot^T · dout.
// Replaces the IndexedSlices gradient of tf.nn.embedding_lookup (models/model_full.py:294).
// Out-of-range ids (the <s> id token_dim+1) match no row -> no gradient, as in TF.
// The products are 1.0*x (exact) and the sum runs on the MFMA pipe with split-K over the id
// list, combined in a fixed order: deterministic, unlike float atomics.
struct OneHotXC {   // A operand, A^T·B form: x = table row v, k = id position i
    static constexpr bool KCONTIG = false;
    const int* ids;
    int rows;
    bool fast_ok(int K) const { return false; }
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        const int id = (k < klim) ? ids[k] : -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (x + j < rows && id == x + j) ? 1.f : 0.f;
        return true;
    }
};

extern "C" size_t d2p_embedding_scatter_ws_bytes(int n, int rows, int E) {
    if (n <= 0 || rows <= 0 || E <= 0) return 0;
    return d2p_plan_ws_bytes(rows, E, n);
}

extern "C" int d2p_embedding_scatter_add_oob0(int n, int rows, int E, const int* ids,
                                              const float* dout, float* dtable, void* ws,
                                              size_t ws_bytes, d2p_stream_t stream) {
    D2P_REQUIRE(n >= 0 && rows > 0 && E > 0, D2P_EINVAL, "embedding scatter: bad sizes");
    D2P_REQUIRE(dtable && (n == 0 || (ids && dout)), D2P_EINVAL, "embedding scatter: null pointer");
    OneHotXC al{ids, rows};
    DenseXC bl{dout, E, E, vec_ok(dout, E)};
    EpiDense ep{dtable, E, nullptr, 0, 0};
    return d2p_launch_gemm(al, bl, ep, rows, E, n, ws, ws_bytes, as_stream(stream), "embedding_scatter");
}
