// K5: dense fp32 MFMA GEMM entry points + column sum (include/d2p.h).
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


// ---- A^T B straight from memory into the MFMA operand registers (round 4) -------------------------------------------
// C[M, N] (+)= sum_k A[rowsA[k], :M]^T B[rowsB[k], :N] (the weight gradients dWx = X^T dZ, dWh = H^T dZ over the rows
// inside their sequences).  Both operands are stored with the OUTPUT index contiguous, and that is the layout a
// 16x16x4 MFMA wants its operands in: lane (c = l & 15, kq = l >> 4) supplies A[k + kq][m] and B[k + kq][n].  So a lane
// loads 16 bytes -- four consecutive m (n) of row k + kq -- straight from memory and uses element i (j) as the operand of
// the MFMA that owns output rows {m0 + 4 c' + i} (columns {n0 + 4 c'' + j}): two 16-byte loads per lane feed sixteen
// MFMAs of a 64 x 64 wave tile, with no LDS staging, no barrier and no fragment read in the loop (the staged kernel
// spends 8 ds_read_b32 per 4 MFMAs on these layouts and meets at a barrier every 32 rows of K).  The KW waves of a
// workgroup own the SAME output tile and an eighth of K each (a ring of D row groups in flight per wave, a lane's row
// index of a group fetched one ring round before the group's data); they meet once, at the end, in a fixed-order
// tree through LDS.  Deterministic; not bit-identical to the staged kernel (another order of the K sum).
typedef float tnd_f32x4 __attribute__((ext_vector_type(4)));
#define TND_D 4           // row groups (of four K rows) in flight per wave

template <int KW, bool GATHER>
__global__ void __launch_bounds__(64 * KW)
gemm_tn_direct_kernel(const float* __restrict__ A, int lda, const int* __restrict__ rowsA, const float* __restrict__ B,
                      int ldb, const int* __restrict__ rowsB, float* __restrict__ C, long ldc, int M, int N, int K, int kps,
                      int accumulate, float* __restrict__ partial, unsigned* __restrict__ tickets) {
    extern __shared__ __attribute__((aligned(16))) float tnd_red[];      // [KW / 2][64 values][64 lanes]
    const int lane = threadIdx.x & 63, c = lane & 15, kq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbn = (N + 63) / 64;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const int m0 = bm * 64, n0 = bn * 64;
    // (round 5) gridDim.y workgroups share an output tile and split K between them (outputs with fewer than 128 tiles:
    // 48 x 2048, 512 x 512): wave w of slice blockIdx.y walks rows [(y KW + w) kps, + kps)
    const int kbeg = ((int)blockIdx.y * KW + w) * kps;
    const int kend = kbeg + kps < K ? kbeg + kps : K;
    const int ng = kend > kbeg ? (kend - kbeg) >> 2 : 0;                  // groups of four rows; a multiple of TND_D
    // columns past the matrix: loaded from its last four (never stored)
    const float* pa = A + (m0 + 4 * c < M - 4 ? m0 + 4 * c : M - 4);
    const float* pb = B + (n0 + 4 * c < N - 4 ? n0 + 4 * c : N - 4);

    tnd_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = tnd_f32x4{0.f, 0.f, 0.f, 0.f};

    if (ng > 0) {
        // The loads and their waits are written out (inline asm): left to itself hipcc sinks the ring's refills to half
        // a round in front of their use and waits for fresh loads at the loop head.  Volatile asm statements keep their
        // order, so the counts are exact.  Per slot and round, in this order: the two data loads of the group D ahead,
        // then (row lists) the two index loads of the group 2 D ahead -- each lane its own K quarter's row, one dword --
        // so when a slot is consumed the oldest NL of the NL D operations in flight are its own: vmcnt(NL (D - 1)).
        constexpr int NL = GATHER ? 4 : 2;
        tnd_f32x4 av[TND_D], bv[TND_D];
        int ia[TND_D], ib[TND_D];                    // this lane's rows for the slot's NEXT data loads
        const int* qia = rowsA + kbeg + kq;          // (GATHER) this lane's entry of group 0
        const int* qib = rowsB + kbeg + kq;
        auto load_idx = [&](int d, int g) {          // rows of group g (clamped to the wave's last group) into slot d
            const int gc = g < ng ? g : ng - 1;
            if constexpr (GATHER) {
                asm volatile("global_load_dword %0, %1, off" : "=v"(ia[d]) : "v"(qia + 4 * gc) : "memory");
                asm volatile("global_load_dword %0, %1, off" : "=v"(ib[d]) : "v"(qib + 4 * gc) : "memory");
            } else {
                ia[d] = ib[d] = kbeg + 4 * gc + kq;
            }
        };
        auto load_data = [&](int d) {
            const float* qa = pa + (long)ia[d] * lda;
            const float* qb = pb + (long)ib[d] * ldb;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(av[d]) : "v"(qa) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[d]) : "v"(qb) : "memory");
        };
#pragma unroll
        for (int d = 0; d < TND_D; ++d) load_idx(d, d);
        if constexpr (GATHER) {
#pragma unroll
            for (int d = 0; d < TND_D; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ia[d]), "+v"(ib[d]) :: "memory");
        }
#pragma unroll
        for (int d = 0; d < TND_D; ++d) {
            load_data(d);
            load_idx(d, TND_D + d);
        }
        for (int g0 = 0; g0 < ng; g0 += TND_D) {
#pragma unroll
            for (int d = 0; d < TND_D; ++d) {
                if constexpr (GATHER)
                    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(av[d]), "+v"(bv[d]), "+v"(ia[d]), "+v"(ib[d])
                                 : "n"(NL * (TND_D - 1)) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(av[d]), "+v"(bv[d]) : "n"(NL * (TND_D - 1)) : "memory");
                const tnd_f32x4 a4 = av[d], b4 = bv[d];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], b4[j], acc[i][j], 0, 0, 0);
                load_data(d);                        // group g0 + d + D into the slot just consumed
                load_idx(d, g0 + d + 2 * TND_D);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the clamped refills of the last round)
    }

    // the KW partial tiles: fixed-order tree through LDS (value e of lane l at [slot][e][l]: conflict-free)
#pragma unroll
    for (int step = KW / 2; step >= 1; step >>= 1) {
        if (w >= step && w < 2 * step) {
            float* dst = tnd_red + (size_t)(w - step) * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[((i * 4 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (w < step) {
            const float* src = tnd_red + (size_t)w * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += src[((i * 4 + j) * 4 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (gridDim.y > 1) {
        // K split between workgroups: each leaves its 64 x 64 partial tile ([value][lane], write-through), takes a ticket of
        // the tile, and the LAST one adds all slices in slice order (sc1 loads; its own included: the same order whoever is
        // last) -- wave w the sixteen-byte pieces (i = w / 2, r in {2 (w % 2), + 1}) -- and stores C
        const unsigned tile_floats = 4096u, nt = gridDim.x;
        const __amdgpu_buffer_rsrc_t pres = d2p_wt_rsrc(partial, gridDim.y * nt * tile_floats * 4u);
        if (w == 0) {
            const int base = (int)((blockIdx.y * nt + blockIdx.x) * tile_floats) + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) d2p_st_wt(pres, base + ((i * 4 + j) * 4 + r) * 64, acc[i][j][r]);
        }
        if (!d2p_last_slice(tickets + blockIdx.x, gridDim.y, reinterpret_cast<int*>(tnd_red))) return;
        const int col = n0 + 4 * c;
        if (KW == 8) {
            const int i = w >> 1;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = 2 * (w & 1) + rr;
                tnd_f32x4 o = {0.f, 0.f, 0.f, 0.f};
                for (unsigned y = 0; y < gridDim.y; ++y) {
                    const int base = (int)((y * nt + blockIdx.x) * tile_floats) + lane;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += d2p_ld_wt(pres, base + ((i * 4 + j) * 4 + r) * 64);
                }
                const int row = m0 + 16 * kq + 4 * r + i;
                if (row < M && col < N) {
                    float* dst = C + (long)row * ldc + col;
                    if (accumulate) o += *reinterpret_cast<const tnd_f32x4*>(dst);
                    *reinterpret_cast<tnd_f32x4*>(dst) = o;
                }
            }
        }
        return;
    }
    if (w != 0) return;
    // lane (c, kq) holds, for MFMA (i, j) and register r: output row m0 + 4 (4 kq + r) + i, column n0 + 4 c + j
    const int col = n0 + 4 * c;
    if (col >= N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + 16 * kq + 4 * r + i;
            if (row < M) {
                float* dst = C + (long)row * ldc + col;
                tnd_f32x4 o = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                if (accumulate) o += *reinterpret_cast<const tnd_f32x4*>(dst);
                *reinterpret_cast<tnd_f32x4*>(dst) = o;
            }
        }
}

// The same product on a 128 x 64 output tile per wave (round 6): a second 16-byte load of A (rows m0 + 64 ..) feeds
// sixteen more MFMAs with the SAME 16 bytes of B -- three loads per 32 MFMAs instead of two per 16 (21 flop per byte
// instead of 16; 128 accumulator registers): 1024 x 2048 x 4480 in 160 us against 178 (117 TFLOP/s).  For outputs of at
// least 256 such tiles only: a 512 x 2048 output has 128, and with K split between two workgroups per tile (the in-launch
// ticket combine) the fill / tree / combine of half a K each ate the gain (94.1 against 93.3 us).  The LSTM kernel
// gradients reach 256 tiles as [X | H]^T dZ (d2p_gemm_f32_tn_rows2).
template <int KW, bool GATHER>
__global__ void __launch_bounds__(64 * KW)
gemm_tn_direct128_kernel(const float* __restrict__ A, int lda, const int* __restrict__ rowsA, const float* __restrict__ B,
                         int ldb, const int* __restrict__ rowsB, float* __restrict__ C, long ldc, int M, int N, int K, int kps,
                         int accumulate, float* __restrict__ partial, unsigned* __restrict__ tickets,
                         const float* __restrict__ A2, int lda2, int msplit, const float* __restrict__ B2, int ldb2,
                         float* __restrict__ C2) {
    // (output rows from msplit on come from a SECOND operand A2 -- column m - msplit of its rows -- through the same row
    //  list: [X | H]^T dZ, the input and the recurrent half of an LSTM's kernel gradient, as one product of 256 tiles;
    //  msplit is a multiple of 128, or >= M: no second operand.  With B2 / C2 the rows from msplit on are a SECOND,
    //  independent product A2^T B2 -> C2 of the same K through the same row lists: two 512 x 2048 kernel gradients -- the
    //  action and the perception decoder's recurrent halves -- fill the chip as 2 x 128 tiles)
    extern __shared__ __attribute__((aligned(16))) float tnd_red[];      // [KW / 2][128 values][64 lanes]
    const int lane = threadIdx.x & 63, c = lane & 15, kq = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbn = (N + 63) / 64;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const int m0 = bm * 128, n0 = bn * 64;
    const int kbeg = ((int)blockIdx.y * KW + w) * kps;
    const int kend = kbeg + kps < K ? kbeg + kps : K;
    const int ng = kend > kbeg ? (kend - kbeg) >> 2 : 0;                  // groups of four rows; a multiple of TND_D
    const bool second = m0 >= msplit;
    const float* Ab = second ? A2 : A;
    const int ma = second ? m0 - msplit : m0, Ma = second ? M - msplit : (msplit < M ? msplit : M);
    lda = second ? lda2 : lda;
    if (second && B2) { B = B2; ldb = ldb2; }
    // output rows: of C from m0 on, or (a second product) of C2 from ma on
    float* Co = (second && C2) ? C2 : C;
    const int mo = (second && C2) ? ma : m0, Mo = (second && C2) ? Ma : M;
    const float* pa0 = Ab + (ma + 4 * c < Ma - 4 ? ma + 4 * c : Ma - 4);
    const float* pa1 = Ab + (ma + 64 + 4 * c < Ma - 4 ? ma + 64 + 4 * c : Ma - 4);
    const float* pb = B + (n0 + 4 * c < N - 4 ? n0 + 4 * c : N - 4);

    tnd_f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = tnd_f32x4{0.f, 0.f, 0.f, 0.f};

    if (ng > 0) {
        // (the ring of gemm_tn_direct_kernel with three data loads per slot)
        constexpr int NL = GATHER ? 5 : 3;
        tnd_f32x4 av0[TND_D], av1[TND_D], bv[TND_D];
        int ia[TND_D], ib[TND_D];
        const int* qia = rowsA + kbeg + kq;
        const int* qib = rowsB + kbeg + kq;
        auto load_idx = [&](int d, int g) {
            const int gc = g < ng ? g : ng - 1;
            if constexpr (GATHER) {
                asm volatile("global_load_dword %0, %1, off" : "=v"(ia[d]) : "v"(qia + 4 * gc) : "memory");
                asm volatile("global_load_dword %0, %1, off" : "=v"(ib[d]) : "v"(qib + 4 * gc) : "memory");
            } else {
                ia[d] = ib[d] = kbeg + 4 * gc + kq;
            }
        };
        auto load_data = [&](int d) {
            const long oa = (long)ia[d] * lda;
            const float* qb = pb + (long)ib[d] * ldb;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(av0[d]) : "v"(pa0 + oa) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(av1[d]) : "v"(pa1 + oa) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bv[d]) : "v"(qb) : "memory");
        };
#pragma unroll
        for (int d = 0; d < TND_D; ++d) load_idx(d, d);
        if constexpr (GATHER) {
#pragma unroll
            for (int d = 0; d < TND_D; ++d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ia[d]), "+v"(ib[d]) :: "memory");
        }
#pragma unroll
        for (int d = 0; d < TND_D; ++d) {
            load_data(d);
            load_idx(d, TND_D + d);
        }
        for (int g0 = 0; g0 < ng; g0 += TND_D) {
#pragma unroll
            for (int d = 0; d < TND_D; ++d) {
                if constexpr (GATHER)
                    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(av0[d]), "+v"(av1[d]), "+v"(bv[d]), "+v"(ia[d]), "+v"(ib[d])
                                 : "n"(NL * (TND_D - 1)) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(av0[d]), "+v"(av1[d]), "+v"(bv[d]) : "n"(NL * (TND_D - 1)) : "memory");
                const tnd_f32x4 a4 = av0[d], a5 = av1[d], b4 = bv[d];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], b4[j], acc[i][j], 0, 0, 0);
                        acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a5[i], b4[j], acc[4 + i][j], 0, 0, 0);
                    }
                load_data(d);
                load_idx(d, g0 + d + 2 * TND_D);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // the KW partial tiles: fixed-order tree through LDS (value e of lane l at [slot][e][l])
#pragma unroll
    for (int step = KW / 2; step >= 1; step >>= 1) {
        if (w >= step && w < 2 * step) {
            float* dst = tnd_red + (size_t)(w - step) * 8192;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[((i * 4 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (w < step) {
            const float* src = tnd_red + (size_t)w * 8192;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += src[((i * 4 + j) * 4 + r) * 64 + lane];
        }
        __syncthreads();
    }
    // lane (c, kq) holds, for MFMA (i, j) and register r: output row m0 + 64 (i / 4) + 4 (4 kq + r) + i % 4, column n0 + 4 c + j
    const int col = n0 + 4 * c;
    if (w != 0 || col >= N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = mo + 64 * (i >> 2) + 16 * kq + 4 * r + (i & 3);
            if (row < Mo) {
                float* dst = Co + (long)row * ldc + col;
                tnd_f32x4 o = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                if (accumulate) o += *reinterpret_cast<const tnd_f32x4*>(dst);
                *reinterpret_cast<tnd_f32x4*>(dst) = o;
            }
        }
}

static int g_gemm_tn_direct = 1;          // d2p_gemm_set_option bit 6 switches it off (A/B)
static int g_rows_by_key = 1;             // d2p_gemm_set_option bit 7 switches it off (A/B, tests)

// K slices between workgroups for outputs of fewer than 128 tiles (1: none): enough slices to fill the chip, at least 512
// rows of K per workgroup
static int tn_direct_slices(long tiles, int K) {
    if (tiles >= 128) return 1;
    int ks = (int)(256 / tiles);
    while (ks > 1 && K / ks < 512) --ks;
    return ks > 16 ? 16 : ks;
}
// the 128 x 64 tile form (gemm_tn_direct128_kernel): outputs of at least 256 tiles of 64 x 64 in whole 128 x 64 tiles, a K
// long enough for two slices
static int g_tn_direct_tall = 1;          // d2p_gemm_set_option bit 8 switches it off (A/B, tests)
static bool tn_direct_tall(int M, int N, int K) {
    return g_tn_direct_tall && M % 128 == 0 && N % 64 == 0 && (long)(M / 128) * (N / 64) >= 256 && (long)(M / 128) * (N / 64) <= 65535 &&
           K >= 1024;
}
static size_t tn_direct_ws_bytes(int M, int N, int K) {
    if (tn_direct_tall(M, N, K)) return 0;
    const long tiles = (long)ceil_div(M, 64) * ceil_div(N, 64);
    const int ks = tn_direct_slices(tiles, K);
    return ks > 1 ? (size_t)ks * tiles * 4096 * sizeof(float) : 0;
}
template <int KW, bool GATHER>
static bool tn_direct_tall_launch(int M, int N, int K, const float* A, long lda, const int* rowsA, const float* B, long ldb,
                                  const int* rowsB, float* C, long ldc, int accumulate, void* ws, size_t ws_bytes, hipStream_t st,
                                  const float* A2 = nullptr, long lda2 = 0, int msplit = 0x7fffffff, const float* B2 = nullptr,
                                  long ldb2 = 0, float* C2 = nullptr) {
    const long tiles = (long)(M / 128) * (N / 64);
    const int ks = 1;
    unsigned* tickets = nullptr;
    const int kps = (ceil_div(K, KW * ks) + 4 * TND_D - 1) / (4 * TND_D) * (4 * TND_D);
    constexpr int lds = (KW / 2) * 8192 * (int)sizeof(float);       // 128 KB: never beside a recurrence's workgroup (see below)
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)gemm_tn_direct128_kernel<KW, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return false;
        attr = true;
    }
    D2pProfScope prof(st, D2P_PROF_GEMM, 2.0 * M * N * K);
    hipLaunchKernelGGL((gemm_tn_direct128_kernel<KW, GATHER>), dim3((unsigned)tiles, (unsigned)ks), dim3(64 * KW), lds, st, A,
                       (int)lda, rowsA, B, (int)ldb, rowsB, C, ldc, M, N, K, kps, accumulate, (float*)ws, tickets, A2, (int)lda2,
                       msplit, B2, (int)ldb2, C2);
    return true;
}
// true: launched.  Shapes it takes: whole 16-byte pieces everywhere, K in whole ring rounds, enough tiles (x K slices) to
// fill the chip
template <int KW, bool GATHER>
static bool tn_direct_launch_kw(int M, int N, int K, const float* A, long lda, const int* rowsA, const float* B, long ldb,
                                const int* rowsB, float* C, long ldc, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!g_gemm_tn_direct || M < 4 || N < 4 || (M | N) % 4 != 0 || K % (4 * TND_D) != 0 || K < 768) return false;
    if (!vec_ok(A, lda) || !vec_ok(B, ldb) || !vec_ok(C, ldc)) return false;
    const long tiles = (long)ceil_div(M, 64) * ceil_div(N, 64);
    if (tiles < 16 || tiles > 65535 || lda > 0x7fffffffL || ldb > 0x7fffffffL) return false;
    if (KW == 8 && tn_direct_tall(M, N, K) &&
        tn_direct_tall_launch<KW, GATHER>(M, N, K, A, lda, rowsA, B, ldb, rowsB, C, ldc, accumulate, ws, ws_bytes, st))
        return true;
    const int ks = tn_direct_slices(tiles, K);
    if (tiles * ks < 128) return false;
    unsigned* tickets = nullptr;
    if (ks > 1) {
        if (!ws || ws_bytes < tn_direct_ws_bytes(M, N, K) || ((uintptr_t)ws & 15)) return false;
        tickets = d2p_gemm_take_tickets((unsigned)tiles);
        if (!tickets) return false;
    }
    const int kps = (ceil_div(K, KW * ks) + 4 * TND_D - 1) / (4 * TND_D) * (4 * TND_D);
    // LDS: the tree's (KW / 2) x 16 KB -- and never less than 100 KB: with 64 KB these eight-wave workgroups become
    // resident BESIDE a persistent recurrence of the other stream (64.5 / 81 KB of the CU's 160) and slow it down by
    // more than they gain (two-stream profile: backward recurrence 343 us on average instead of 235, the step
    // unchanged although the product itself is 15 % faster); at 100 KB they wait for the recurrence's workgroup to
    // leave the CU, as the four-wave staged kernel does for want of a free SIMD (tools/corun_probe.py)
    constexpr int lds_tree = (KW / 2) * 4096 * (int)sizeof(float);
    // (100 KB: the optimum of a sweep of the request in the step, profiles/r04_tnd_lds_request_sweep.log)
    constexpr int lds_req = 100 * 1024;
    const int lds = lds_tree > lds_req ? lds_tree : lds_req;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)gemm_tn_direct_kernel<KW, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return false;
        attr = true;
    }
    D2pProfScope prof(st, D2P_PROF_GEMM, 2.0 * M * N * K);
    hipLaunchKernelGGL((gemm_tn_direct_kernel<KW, GATHER>), dim3((unsigned)tiles, (unsigned)ks), dim3(64 * KW), lds, st, A, (int)lda,
                       rowsA, B, (int)ldb, rowsB, C, ldc, M, N, K, kps, accumulate, (float*)ws, tickets);
    return true;
}

template <bool GATHER>
static bool tn_direct_launch(int M, int N, int K, const float* A, long lda, const int* rowsA, const float* B, long ldb,
                             const int* rowsB, float* C, long ldc, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    return tn_direct_launch_kw<8, GATHER>(M, N, K, A, lda, rowsA, B, ldb, rowsB, C, ldc, accumulate, ws, ws_bytes, st);
}

extern "C" int d2p_gemm_set_option(int bk32) {
    g_gemm_bk32 = (bk32 & 1) ? 1 : 0;
    g_gemm_small_ksr = (bk32 & 2) ? 0 : 1;     // bit 1: switch the small-problem 32x32 tile off
    g_gemm_nosel = (bk32 & 4) ? 0 : 1;         // bit 2: keep the select-at-store loaders for every K
    g_gemm_no_bk32 = (bk32 & 16) ? 1 : 0;      // bit 4 (experiment): never the 32-deep slabs
    g_gemm_tn_direct = (bk32 & 64) ? 0 : 1;    // bit 6: the A^T B products on the staged kernel (no register-direct form)
    g_rows_by_key = (bk32 & 128) ? 0 : 1;      // bit 7: the embedding gradient as a one-hot GEMM (no rows_by_key_kernel)
    g_gemm_fold = (bk32 & 32) ? 0 : 1;         // bit 5: split-K combine as a separate launch (round 2's form)
    g_tn_direct_tall = (bk32 & 256) ? 0 : 1;   // bit 8: the large A^T B products on 64 x 64 tiles (no 128 x 64 form)
    g_gemm_dma_big = (bk32 & 8) ? 1 : 0;       // bit 3 (experiment): large dense GEMMs on the persistent LDS-DMA kernel
    g_gemm_dma_grid = bk32 >> 16;              // bits 16..: persistent grid of the LDS-DMA kernel (0 = automatic)
    return D2P_OK;
}
extern "C" int d2p_gemm_force_plan(int tile, int splits) {
    g_gemm_force_tile = tile;
    g_gemm_force_split = splits;
    return D2P_OK;
}

extern "C" size_t d2p_gemm_ws_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const size_t a = d2p_plan_ws_bytes(M, N, K), b = tn_direct_ws_bytes(M, N, K);     // (whichever kernel takes the product)
    return a > b ? a : b;
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
    if (!bias && act == 0 && M > 0 && N > 0 &&
        tn_direct_launch<false>(M, N, K, A, lda, nullptr, B, ldb, nullptr, C, ldc, accumulate, ws, ws_bytes, as_stream(stream))) {
        D2P_LAUNCH_CHECK("gemm_tn_direct");
        return D2P_OK;
    }
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
    if (tn_direct_launch<true>(M, N, K, A, lda, rowsA, B, ldb, rowsB, C, ldc, accumulate, ws, ws_bytes, as_stream(stream))) {
        D2P_LAUNCH_CHECK("gemm_tn_direct");
        return D2P_OK;
    }
    GatherXC al{A, lda, M, vec_ok(A, lda), rowsA};
    GatherXC bl{B, ldb, N, vec_ok(B, ldb), rowsB};
    EpiDense ep{C, ldc, nullptr, 0, accumulate};
    return d2p_launch_gemm(al, bl, ep, M, N, K, ws, ws_bytes, as_stream(stream), "gemm_tn_rows");
}

// C[:M0] (+)= A0^T B and C[M0 : M0 + M1] (+)= A1^T B over ONE pair of row lists: the two halves of an LSTM's kernel
// gradient (dWx = X^T dZ, dWh = H^T dZ; they are adjacent row blocks of the same gradient tensor and read the same dZ rows).
// 512 + 512 rows x 2048 columns are 256 tiles of 128 x 64: one launch of gemm_tn_direct128_kernel without a K split; any
// other geometry (or d2p_gemm_set_option bit 8) runs the two products one after the other -- the same values bit for bit
// (the same K partition between the waves, the same tree).
extern "C" int d2p_gemm_f32_tn_rows2(int M0, int M1, int N, int K, const float* A0, long lda0, const float* A1, long lda1,
                                     const int* rowsA, const float* B, long ldb, const int* rowsB, float* C, long ldc,
                                     int accumulate, void* ws, size_t ws_bytes, d2p_stream_t stream) {
    D2P_REQUIRE(M0 >= 0 && M1 >= 0, D2P_EINVAL, "gemm_tn_rows2: negative M");
    if (M0 > 0 && M1 > 0 && N > 0 && K > 0 && g_gemm_tn_direct && M0 % 128 == 0 && tn_direct_tall(M0 + M1, N, K) && K % (4 * TND_D) == 0 && A0 && A1 && B && C && rowsA && rowsB &&
        vec_ok(A0, lda0) && vec_ok(A1, lda1) && vec_ok(B, ldb) && vec_ok(C, ldc) && lda0 <= 0x7fffffffL && lda1 <= 0x7fffffffL &&
        ldb <= 0x7fffffffL &&
        tn_direct_tall_launch<8, true>(M0 + M1, N, K, A0, lda0, rowsA, B, ldb, rowsB, C, ldc, accumulate, ws, ws_bytes,
                                       as_stream(stream), A1, lda1, M0)) {
        D2P_LAUNCH_CHECK("gemm_tn_direct128");
        return D2P_OK;
    }
    int rc = M0 > 0 ? d2p_gemm_f32_tn_rows(M0, N, K, A0, lda0, rowsA, B, ldb, rowsB, C, ldc, accumulate, ws, ws_bytes, stream) : D2P_OK;
    if (rc) return rc;
    return M1 > 0 ? d2p_gemm_f32_tn_rows(M1, N, K, A1, lda1, rowsA, B, ldb, rowsB, C + (long)M0 * ldc, ldc, accumulate, ws, ws_bytes,
                                        stream) : D2P_OK;
}

// Two independent products of ONE shape through one pair of row lists, C0 (+)= A0^T B0 and C1 (+)= A1^T B1 (the recurrent
// halves of the action and the perception decoder's kernel gradients: 512 x 2048 each, the same rows of their own dZ):
// 2 x 128 tiles of 128 x 64 in one launch of gemm_tn_direct128_kernel; any other geometry, or d2p_gemm_set_option bit 8:
// one after the other, the same values bit for bit.
extern "C" int d2p_gemm_f32_tn_rows_x2(int M, int N, int K, const float* A0, long lda0, const float* B0, long ldb0, float* C0,
                                       const float* A1, long lda1, const float* B1, long ldb1, float* C1, long ldc,
                                       const int* rowsA, const int* rowsB, int accumulate, void* ws, size_t ws_bytes,
                                       d2p_stream_t stream) {
    D2P_REQUIRE(M >= 0, D2P_EINVAL, "gemm_tn_rows_x2: negative M");
    if (M > 0 && N > 0 && K > 0 && g_gemm_tn_direct && M % 128 == 0 && tn_direct_tall(2 * M, N, K) && K % (4 * TND_D) == 0 &&
        A0 && A1 && B0 && B1 && C0 && C1 && rowsA && rowsB && vec_ok(A0, lda0) && vec_ok(A1, lda1) && vec_ok(B0, ldb0) &&
        vec_ok(B1, ldb1) && vec_ok(C0, ldc) && vec_ok(C1, ldc) && lda0 <= 0x7fffffffL && lda1 <= 0x7fffffffL &&
        ldb0 <= 0x7fffffffL && ldb1 <= 0x7fffffffL &&
        tn_direct_tall_launch<8, true>(2 * M, N, K, A0, lda0, rowsA, B0, ldb0, rowsB, C0, ldc, accumulate, ws, ws_bytes,
                                       as_stream(stream), A1, lda1, M, B1, ldb1, C1)) {
        D2P_LAUNCH_CHECK("gemm_tn_direct128");
        return D2P_OK;
    }
    int rc = d2p_gemm_f32_tn_rows(M, N, K, A0, lda0, rowsA, B0, ldb0, rowsB, C0, ldc, accumulate, ws, ws_bytes, stream);
    if (rc) return rc;
    return d2p_gemm_f32_tn_rows(M, N, K, A1, lda1, rowsA, B1, ldb1, rowsB, C1, ldc, accumulate, ws, ws_bytes, stream);
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

// ---- rows summed by key: out[v] = sum of the rows r of X with ids[r] == v (the embedding gradient) ------------------------
// The one-hot GEMM above spends a matrix-pipe launch on what is a read of X (52 MB for the action decoder's dz: 57 us at
// 7-8 table rows, one tile row of workgroups walking all of K).  Here a workgroup streams a slice of rows of 64 columns:
// thread (cl, rl) reads 16 bytes of the rows rl, rl + 16, ... (four in flight) and adds each into ITS OWN cell
// [rl][ids[row]][cl] of an LDS table -- no two threads share a cell, so the order of every sum is the row order;
// the sixteen row lanes are then added in order and the slices by the deterministic combine kernel.
// 256 threads (16 row lanes x 16 column lanes of 16 bytes): with fewer waves the workgroups become resident beside a
// persistent recurrence of the other stream (tools/corun_probe.py) and their stream of HBM reads slows it down -- the first
// version (128 threads) was 23 us faster than the one-hot GEMM alone and gained 5 us in the step.  So the table must
// fit 16 row lanes: up to 32 keys (the action decoder's 8; the program decoder's 52 stay on the one-hot GEMM, which is
// as fast there).
#define RBK_RL 16

__global__ void __launch_bounds__(256)
rows_by_key_kernel(int n, int nkeys, int E, int rows_per_slice, const int* __restrict__ ids, const float* __restrict__ X,
                   float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float rbk_tab[];      // [RBK_RL][nkeys][64]
    const int tid = threadIdx.x, cl = tid & 15, rl = tid >> 4;
    const int c0 = blockIdx.x * 64 + 4 * cl;
    const int cc = c0 < E - 4 ? c0 : E - 4;                               // (E % 4 == 0; columns past E: loaded, never stored)
    const int r0 = blockIdx.y * rows_per_slice;
    const int r1 = r0 + rows_per_slice < n ? r0 + rows_per_slice : n;
    for (int i = tid; i < RBK_RL * nkeys * 16; i += 256) reinterpret_cast<tnd_f32x4*>(rbk_tab)[i] = tnd_f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    float* mine = rbk_tab + (size_t)rl * nkeys * 64 + 4 * cl;
    for (int r = r0 + rl; r < r1; r += 4 * RBK_RL) {
        tnd_f32x4 v[4];
        int key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * RBK_RL < r1 ? r + u * RBK_RL : r;
            key[u] = r + u * RBK_RL < r1 ? ids[rr] : -1;
            v[u] = *reinterpret_cast<const tnd_f32x4*>(X + (long)rr * E + cc);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (key[u] >= 0 && key[u] < nkeys) {                          // (ids outside the table: dropped, as the gather reads zeros)
                tnd_f32x4* cell = reinterpret_cast<tnd_f32x4*>(mine + (size_t)key[u] * 64);
                *cell = *cell + v[u];
            }
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.y * nkeys * E;
    for (int i = tid; i < nkeys * 16; i += 256) {
        const int key = i >> 4, c4 = i & 15;
        tnd_f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < RBK_RL; ++q) sum += *reinterpret_cast<const tnd_f32x4*>(rbk_tab + ((size_t)q * nkeys + key) * 64 + 4 * c4);
        const int col = blockIdx.x * 64 + 4 * c4;
        if (col < E) *reinterpret_cast<tnd_f32x4*>(out + (size_t)key * E + col) = sum;
    }
}

// ---- S = A^T dz for the perception decoder's factored input, from the structure of A ---------------------------------
// A (d2p_per_affine_rows) has P + 1 non-zeros per row: per[r, 0..P) in the columns of the row's demonstration index
// g = r % G and a 1 behind them -- so S [NCp, E] is, per index, P + 1 weighted sums of the index's dz rows: a read of dz
// (52 MB) instead of a 60 x 2048 x 6400 matrix-pipe launch (34 us).  Row lane = index: thread (column lane, g) walks the
// rows r = g, g + G, ... of its slice (four in flight) with P + 1 accumulators of four columns in registers; slices are
// added by the deterministic combine kernel.  G <= 16, P <= 8, rows a multiple of G.
#define PRT_MAXP 8
__global__ void __launch_bounds__(256)
per_rows_tn_kernel(int rows, int G, int P, int E, int NCp, int rows_per_slice, const float* __restrict__ per,
                   const float* __restrict__ dz, float* __restrict__ partial) {
    const int tid = threadIdx.x, cl = tid & 15, g = tid >> 4;
    float* out = partial + (size_t)blockIdx.y * NCp * E;
    const int c0 = blockIdx.x * 64 + 4 * cl;
    if (g >= G) {
        // the spare row lanes write the pad rows of S (NC .. NCp) as zeros
        for (int i = (g - G) * 16 + cl; i < (NCp - G * (P + 1)) * 16; i += (16 - G) * 16) {
            const int row = G * (P + 1) + i / 16, c = blockIdx.x * 64 + 4 * (i % 16);
            if (c < E) *reinterpret_cast<tnd_f32x4*>(out + (size_t)row * E + c) = tnd_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    const int cc = c0 < E - 4 ? c0 : E - 4;
    const int r0 = blockIdx.y * rows_per_slice;                      // (a multiple of G)
    const int r1 = r0 + rows_per_slice < rows ? r0 + rows_per_slice : rows;
    tnd_f32x4 acc[PRT_MAXP + 1];
#pragma unroll
    for (int j = 0; j <= PRT_MAXP; ++j) acc[j] = tnd_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + g; r < r1; r += 4 * G) {
        tnd_f32x4 v[4];
        float w[4][PRT_MAXP];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = r + u * G < r1;
            const int rr = ok ? r + u * G : r;
            v[u] = *reinterpret_cast<const tnd_f32x4*>(dz + (long)rr * E + cc);
            if (!ok) v[u] = tnd_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < PRT_MAXP; ++j) w[u][j] = j < P ? per[(long)rr * P + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < PRT_MAXP; ++j)
                if (j < P) acc[j] += w[u][j] * v[u];
            acc[PRT_MAXP] += v[u];
        }
    }
    if (c0 < E) {
#pragma unroll
        for (int j = 0; j < PRT_MAXP; ++j)
            if (j < P) *reinterpret_cast<tnd_f32x4*>(out + (size_t)(g * (P + 1) + j) * E + c0) = acc[j];
        *reinterpret_cast<tnd_f32x4*>(out + (size_t)(g * (P + 1) + P) * E + c0) = acc[PRT_MAXP];
    }
}

// The forward counterpart: z [rows, E] = A . HWx + bias from the structure of A (P + 1 non-zeros per row): a write of z
// (52 MB) with the six HWx rows of the row's index in registers, instead of a K = 60 matrix-pipe launch (34 us).
// Workgroup = (256 columns, index g, 32 rows of that index); thread (column lane, row lane).
__global__ void __launch_bounds__(256)
per_rows_nn_kernel(int rows, int G, int P, int E, const float* __restrict__ per, const float* __restrict__ HWx,
                   const float* __restrict__ bias, float* __restrict__ z) {
    const int tid = threadIdx.x, cl = tid & 63, rl = tid >> 6;
    const int c0 = blockIdx.x * 256 + 4 * cl;
    const int g = blockIdx.y, i0 = blockIdx.z * 32;
    if (c0 >= E) return;
    tnd_f32x4 hw[PRT_MAXP + 1];
#pragma unroll
    for (int j = 0; j < PRT_MAXP; ++j)
        hw[j] = j < P ? *reinterpret_cast<const tnd_f32x4*>(HWx + (size_t)(g * (P + 1) + j) * E + c0) : tnd_f32x4{0.f, 0.f, 0.f, 0.f};
    hw[PRT_MAXP] = *reinterpret_cast<const tnd_f32x4*>(HWx + (size_t)(g * (P + 1) + P) * E + c0);
    if (bias) hw[PRT_MAXP] += *reinterpret_cast<const tnd_f32x4*>(bias + c0);
#pragma unroll 2
    for (int i = i0 + rl; i < i0 + 32; i += 4) {
        const long r = (long)i * G + g;
        if (r >= rows) break;
        tnd_f32x4 o = hw[PRT_MAXP];
#pragma unroll
        for (int j = 0; j < PRT_MAXP; ++j)
            if (j < P) o += per[r * P + j] * hw[j];
        *reinterpret_cast<tnd_f32x4*>(z + r * E + c0) = o;
    }
}
// (declared in include/d2p.h)
extern "C" int d2p_per_rows_nn(int rows, int G, int P, int NCp, int E, const float* per, const float* HWx, const float* bias,
                               float* z, d2p_stream_t stream) {
    D2P_REQUIRE(rows > 0 && G > 0 && G <= 65535 && P > 0 && P <= PRT_MAXP && NCp >= G * (P + 1) && E >= 4 && E % 4 == 0 &&
                rows % G == 0, D2P_EINVAL, "per_rows_nn: rows=%d G=%d P=%d NCp=%d E=%d", rows, G, P, NCp, E);
    D2P_REQUIRE(per && HWx && z, D2P_EINVAL, "per_rows_nn: null pointer");
    D2P_REQUIRE((((uintptr_t)HWx | (uintptr_t)z | (uintptr_t)bias) & 15) == 0, D2P_EALIGN, "per_rows_nn: 16-byte alignment");
    const int per_group = rows / G;
    hipLaunchKernelGGL(per_rows_nn_kernel, dim3((E + 255) / 256, G, (per_group + 31) / 32), dim3(256), 0, as_stream(stream), rows,
                       G, P, E, per, HWx, bias, z);
    D2P_LAUNCH_CHECK("per_rows_nn");
    return D2P_OK;
}

static int prt_slices(int rows, int G) {
    int s = rows / (G * 24);                // ~24 rows per thread
    if (s > 32) s = 32;
    return s < 1 ? 1 : s;
}
extern "C" size_t d2p_per_rows_tn_ws_bytes(int rows, int G, int NCp, int E) {
    if (rows <= 0 || G <= 0 || NCp <= 0 || E <= 0) return 0;
    return (size_t)prt_slices(rows, G) * NCp * E * sizeof(float);
}
// (declared in include/d2p.h)
extern "C" int d2p_per_rows_tn(int rows, int G, int P, int NCp, int E, const float* per, const float* dz, float* S,
                               void* ws, size_t ws_bytes, d2p_stream_t stream) {
    D2P_REQUIRE(rows > 0 && G > 0 && G <= 15 && P > 0 && P <= PRT_MAXP && NCp >= G * (P + 1) && E >= 64 && E % 4 == 0 &&
                rows % G == 0, D2P_EINVAL, "per_rows_tn: rows=%d G=%d P=%d NCp=%d E=%d", rows, G, P, NCp, E);
    D2P_REQUIRE(per && dz && S && ws && ws_bytes >= d2p_per_rows_tn_ws_bytes(rows, G, NCp, E), D2P_EWS,
                "per_rows_tn: null pointer or workspace too small");
    D2P_REQUIRE((((uintptr_t)dz | (uintptr_t)S | (uintptr_t)ws) & 15) == 0, D2P_EALIGN, "per_rows_tn: 16-byte alignment");
    hipStream_t st = as_stream(stream);
    const int slices = prt_slices(rows, G);
    int rps = (rows + slices - 1) / slices;
    rps = (rps + G - 1) / G * G;
    hipLaunchKernelGGL(per_rows_tn_kernel, dim3((E + 63) / 64, slices), dim3(256), 0, st, rows, G, P, E, NCp, rps, per, dz,
                       (float*)ws);
    D2P_LAUNCH_CHECK("per_rows_tn");
    EpiDense ep{S, E, nullptr, 0, 0};
    const long total = (long)NCp * E;
    if (slices <= 16) {
        int blocks = (int)((total + 255) / 256);
        hipLaunchKernelGGL((gemm_splitk_reduce_flat_kernel<EpiDense>), dim3(blocks > 2048 ? 2048 : blocks), dim3(256), 0, st, ep,
                           (const float*)ws, NCp, E, slices);
    } else {
        int blocks = (int)((total * 16 + 255) / 256);
        hipLaunchKernelGGL((gemm_splitk_reduce_kernel<EpiDense>), dim3(blocks > 4096 ? 4096 : blocks), dim3(256), 0, st, ep,
                           (const float*)ws, NCp, E, slices);
    }
    D2P_LAUNCH_CHECK("per_rows_tn_combine");
    return D2P_OK;
}

static int rbk_slices(int n) {
    int s = (n + 199) / 200;              // ~200 rows per workgroup
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}
static bool rbk_ok(int n, int rows, int E, const float* dout, const float* dtable) {
    return g_rows_by_key && n >= 256 && rows <= 32 && E % 4 == 0 && E >= 64 && (((uintptr_t)dout | (uintptr_t)dtable) & 15) == 0;
}

extern "C" size_t d2p_embedding_scatter_ws_bytes(int n, int rows, int E) {
    if (n <= 0 || rows <= 0 || E <= 0) return 0;
    const size_t a = d2p_plan_ws_bytes(rows, E, n), b = (size_t)rbk_slices(n) * rows * E * sizeof(float);
    return a > b ? a : b;
}

extern "C" int d2p_embedding_scatter_add_oob0(int n, int rows, int E, const int* ids,
                                              const float* dout, float* dtable, void* ws,
                                              size_t ws_bytes, d2p_stream_t stream) {
    D2P_REQUIRE(n >= 0 && rows > 0 && E > 0, D2P_EINVAL, "embedding scatter: bad sizes");
    D2P_REQUIRE(dtable && (n == 0 || (ids && dout)), D2P_EINVAL, "embedding scatter: null pointer");
    const int slices = rbk_slices(n);
    if (rbk_ok(n, rows, E, dout, dtable) && ws && ws_bytes >= (size_t)slices * rows * E * sizeof(float)) {
        hipStream_t st = as_stream(stream);
        const int lds = RBK_RL * rows * 64 * (int)sizeof(float);
        static bool attr = false;
        if (!attr) {
            D2P_HIP(hipFuncSetAttribute((const void*)rows_by_key_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr = true;
        }
        const int rps = (n + slices - 1) / slices;
        hipLaunchKernelGGL(rows_by_key_kernel, dim3((E + 63) / 64, slices), dim3(256), lds, st, n, rows, E, rps, ids, dout,
                           (float*)ws);
        D2P_LAUNCH_CHECK("rows_by_key");
        EpiDense ep{dtable, E, nullptr, 0, 0};
        const long total = (long)rows * E;
        if (slices <= 16) {
            int blocks = (int)((total + 255) / 256);
            hipLaunchKernelGGL((gemm_splitk_reduce_flat_kernel<EpiDense>), dim3(blocks > 2048 ? 2048 : blocks), dim3(256), 0, st, ep,
                               (const float*)ws, rows, E, slices);
        } else {
            int blocks = (int)((total * 16 + 255) / 256);
            hipLaunchKernelGGL((gemm_splitk_reduce_kernel<EpiDense>), dim3(blocks > 4096 ? 4096 : blocks), dim3(256), 0, st, ep,
                               (const float*)ws, rows, E, slices);
        }
        D2P_LAUNCH_CHECK("rows_by_key_combine");
        return D2P_OK;
    }
    OneHotXC al{ids, rows};
    DenseXC bl{dout, E, E, vec_ok(dout, E)};
    EpiDense ep{dtable, E, nullptr, 0, 0};
    return d2p_launch_gemm(al, bl, ep, rows, E, n, ws, ws_bytes, as_stream(stream), "embedding_scatter");
}
