// fp32 MFMA GEMM core for gfx950: C[M,N] = A[M,K] * B[K,N] with pluggable operand
// loaders (dense / implicit im2col) and epilogues.
//
// Machine mapping (MI355X_MICROARCH.md, cdna_hip_programming.md §3):
//  - v_mfma_f32_32x32x2_f32: per wave a 32x32 fp32 tile, K=2 per instruction, exact
//    f32 (bitwise an fmaf chain), 64 cycles/SIMD -> 157 TF chip peak.
//    A operand: lane l holds A[i=l&31][k=l>>5]; B: lane l holds B[k=l>>5][j=l&31];
//    C/D: 16 regs/lane, col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
//  - 256-thread workgroups = 4 waves (one per SIMD) arranged 2x2 over a BMxBN tile,
//    each wave owning TMxTN MFMA tiles (TM=BM/64, TN=BN/64).
//  - K is consumed in BK=16 slabs staged through double-buffered LDS; one barrier per
//    slab; the next slab's global loads are issued before the MFMAs of the current one.
//  - An operand whose K index is contiguous in memory ("KCONTIG", e.g. row-major A) is
//    stored in LDS as [x][BK+4] and read with one ds_read_b128 per 4 MFMAs: lanes 0-31
//    take k..k+3, lanes 32-63 take k+4..k+7, and the j-th MFMA of the group consumes
//    component j from every lane (any pairing of k indices is valid as long as A and B
//    agree).  Row stride 20 floats = 5 sixteen-byte slots (odd) -> conflict-free b128.
//  - An operand whose x index is contiguous ("XCONTIG", e.g. row-major B) is stored as
//    [BK][BX] and read with ds_read_b32: lanes 0-31 consecutive floats -> conflict-free.
#pragma once
#include "common.h"
#include "prof.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define D2P_GEMM_BK 16

// ------------------------------------------------------------------------------------
// Loaders.  load4(x, k, klim, v):
//   KCONTIG:  v[j] = elem(x, k+j)  for k+j < klim, x < X, else 0
//   XCONTIG:  v[j] = elem(x+j, k)  for x+j < X, k < klim, else 0
// ------------------------------------------------------------------------------------
struct DenseKC {  // elem(x,k) = p[x*ld + k]
    static constexpr bool KCONTIG = true;
    const float* p;
    long ld;
    int X;
    int vec;  // ld % 4 == 0 and p 16-byte aligned
    // strided batch (grid.y): problem b reads from p + (b / nb0) * bs1 + (b % nb0) * bs0
    long bs1 = 0, bs0 = 0;
    int nb0 = 1;
    __device__ __forceinline__ void shift(int b) { p += (long)(b / nb0) * bs1 + (long)(b % nb0) * bs0; }
    // FAST: vec && K % 4 == 0: one unconditional 16-byte load from a clamped address, then a
    // select -- no control flow, so hipcc keeps counted vmcnt waits across the K loop.
    bool fast_ok(int K) const { return vec && K >= 4 && (K % 4) == 0 && X > 0; }
    __device__ __forceinline__ const float* row_ptr(int x) const { return p + (long)x * ld; }
    // Returns false when the caller must treat v as zeros (FAST defers the select to the
    // LDS-store point so that nothing consumes the load result early).
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        if (FAST) {
            const bool ok = (x < X) & (k < klim);
            const float4 t = *reinterpret_cast<const float4*>(p + (ok ? (long)x * ld + k : 0L));
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            return ok;
        }
        if (x >= X) {
            v[0] = v[1] = v[2] = v[3] = 0.f;
            return true;
        }
        const float* q = p + (long)x * ld + k;
        if (vec && k + 3 < klim) {
            float4 t = *reinterpret_cast<const float4*>(q);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (k + j < klim) ? q[j] : 0.f;
        }
        return true;
    }
};

// Rows through an index list: elem(x,k) = p[idx[x]*ld + k].  For products over the ACTIVE rows of a padded
// time-major batch (rows past a sequence's length are zeros and their outputs are never read): the GEMM runs over
// the listed rows only, its epilogue (EpiScatterRows) puts row x of the result at idx[x].
struct GatherKC {
    static constexpr bool KCONTIG = true;
    const float* p;
    long ld;
    int X;
    int vec;
    const int* idx;
    __device__ __forceinline__ void shift(int) {}
    bool fast_ok(int K) const { return vec && K >= 4 && (K % 4) == 0 && X > 0; }
    __device__ __forceinline__ const float* row_ptr(int x) const { return p + (long)idx[x] * ld; }
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        if (FAST) {
            const bool ok = (x < X) & (k < klim);
            const float4 t = *reinterpret_cast<const float4*>(ok ? row_ptr(x) + k : p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            return ok;
        }
        if (x >= X) {
            v[0] = v[1] = v[2] = v[3] = 0.f;
            return true;
        }
        const float* q = row_ptr(x) + k;
        if (vec && k + 3 < klim) {
            float4 t = *reinterpret_cast<const float4*>(q);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (k + j < klim) ? q[j] : 0.f;
        }
        return true;
    }
};

struct DenseXC {  // elem(x,k) = p[k*ld + x]
    static constexpr bool KCONTIG = false;
    const float* p;
    long ld;
    int X;
    int vec;
    long bs1 = 0, bs0 = 0;   // strided batch, as DenseKC
    int nb0 = 1;
    __device__ __forceinline__ void shift(int b) { p += (long)(b / nb0) * bs1 + (long)(b % nb0) * bs0; }
    bool fast_ok(int K) const { return vec && X >= 4 && (X % 4) == 0 && K > 0; }
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        if (FAST) {
            const bool ok = (x < X) & (k < klim);
            const float4 t = *reinterpret_cast<const float4*>(p + (ok ? (long)k * ld + x : 0L));
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            return ok;
        }
        if (k >= klim) {
            v[0] = v[1] = v[2] = v[3] = 0.f;
            return true;
        }
        const float* q = p + (long)k * ld + x;
        if (vec && x + 3 < X) {
            float4 t = *reinterpret_cast<const float4*>(q);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (x + j < X) ? q[j] : 0.f;
        }
        return true;
    }
};

// K through an index list: elem(x,k) = p[idx[k]*ld + x].  For 'tn' products (C = A^T B: weight gradients
// X^T dZ) over the ACTIVE rows of a padded time-major batch only: both operands are read through the SAME list of
// K row indices; the rows it leaves out are zeros in dZ and would add nothing (30 % of the demonstration rows, 45 %
// of the program rows in the bench's batches).  The list is padded by the caller to whole K slabs with the index
// of any such zero row.
struct GatherXC {
    static constexpr bool KCONTIG = false;
    const float* p;
    long ld;
    int X;
    int vec;
    const int* idx;
    __device__ __forceinline__ void shift(int) {}
    bool fast_ok(int K) const { return vec && X >= 4 && (X % 4) == 0 && K > 0; }
    template <bool FAST>
    __device__ __forceinline__ bool load4(int x, int k, int klim, float (&v)[4]) const {
        if (FAST) {
            const bool ok = (x < X) & (k < klim);
            const float4 t = *reinterpret_cast<const float4*>(p + (ok ? (long)idx[k] * ld + x : 0L));
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            return ok;
        }
        if (k >= klim) {
            v[0] = v[1] = v[2] = v[3] = 0.f;
            return true;
        }
        const float* q = p + (long)idx[k] * ld + x;
        if (vec && x + 3 < X) {
            float4 t = *reinterpret_cast<const float4*>(q);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (x + j < X) ? q[j] : 0.f;
        }
        return true;
    }
};
template <class L> struct d2p_kgather { static constexpr bool value = false; };
template <> struct d2p_kgather<GatherXC> { static constexpr bool value = true; };

// Loaders whose out-of-range rows / columns may hold ANYTHING (they only feed output rows /
// columns that are never stored) and whose only zeros are k >= klim: with K a multiple of the
// slab depth such a loader needs no select at all between the global load and the LDS store.
template <class L> struct d2p_nosel_ok { static constexpr bool value = false; };
template <> struct d2p_nosel_ok<DenseKC> { static constexpr bool value = true; };
template <> struct d2p_nosel_ok<DenseXC> { static constexpr bool value = true; };
template <> struct d2p_nosel_ok<GatherKC> { static constexpr bool value = true; };
template <> struct d2p_nosel_ok<GatherXC> { static constexpr bool value = true; };
// strided-batched launches (grid.y > 1) exist for dense operands with the dense epilogue only
template <class EP> struct d2p_batch_ok { static constexpr bool value = false; };

// ------------------------------------------------------------------------------------
// Epilogues.  operator()(row, col, value): called once per in-range output element.
// ------------------------------------------------------------------------------------
// Interface used by the kernel epilogue (so that per-column values are loaded once per lane and
// the C reads of an accumulating GEMM are issued together instead of one-load-one-wait):
//   col_value(col)      value added to every element of the column (bias), 0 if none
//   has_c()             whether C_old must be added
//   c_value(row, col)   C_old
//   store(row, col, v)  v already holds acc + C_old + col_value; applies act and writes
struct EpiDense {
    float* C;
    long ldc;
    const float* bias;  // [N] or null
    int act;            // 0 none, 1 lrelu(0.2)
    int accumulate;     // C = act(v + C_old + bias)
    long cs1 = 0, cs0 = 0, bb1 = 0, bb0 = 0;   // strided batch: C and bias offsets of problem b
    int nb0 = 1;
    __device__ __forceinline__ void shift(int b) {
        C += (long)(b / nb0) * cs1 + (long)(b % nb0) * cs0;
        if (bias) bias += (long)(b / nb0) * bb1 + (long)(b % nb0) * bb0;
    }
    __device__ __forceinline__ float col_value(int col) const { return bias ? bias[col] : 0.f; }
    __device__ __forceinline__ bool has_c() const { return accumulate != 0; }
    __device__ __forceinline__ float c_value(int row, int col) const { return C[(long)row * ldc + col]; }
    __device__ __forceinline__ void store(int row, int col, float v) const {
        if (act == 1) v = d2p_lrelu(v);
        C[(long)row * ldc + col] = v;
    }
    __device__ __forceinline__ void operator()(int row, int col, float v) const {
        if (accumulate) v += c_value(row, col);
        store(row, col, v + col_value(col));
    }
};

// Row x of the product lands at row idx[x] of C (see GatherKC); bias, no accumulate.
struct EpiScatterRows {
    float* C;
    long ldc;
    const float* bias;
    const int* idx;
    __device__ __forceinline__ void shift(int) {}
    __device__ __forceinline__ float col_value(int col) const { return bias ? bias[col] : 0.f; }
    __device__ __forceinline__ bool has_c() const { return false; }
    __device__ __forceinline__ float c_value(int, int) const { return 0.f; }
    __device__ __forceinline__ void store(int row, int col, float v) const { C[(long)idx[row] * ldc + col] = v; }
    __device__ __forceinline__ void operator()(int row, int col, float v) const { store(row, col, v + col_value(col)); }
};

// ------------------------------------------------------------------------------------
// Tile shapes: BM x BN block tile, waves arranged WM x WN (WM*WN = 4), each wave owning
// TM x TN MFMA tiles of 32x32 (TM = BM/WM/32, TN = BN/WN/32):
//   64x64  (2x2, 1x1)  default for small problems
//   128x128 (2x2, 2x2) large dense GEMMs
//   128x32 (4x1, 1x1)  skinny N (conv Cout / Cin = 16..32): no wasted MFMA columns
//   256x32 (4x1, 2x1)  very tall skinny N
// ------------------------------------------------------------------------------------
// KS ("K split across waves") mode, for tiny outputs with a huge K (conv weight gradients:
// 36x16 .. 432x48 outputs over up to 10M rows): all four waves own the SAME BM x BN tile and
// each consumes one 8-deep chunk of every 32-deep K slab; every wave writes its own split-K
// slab (slab index blockIdx.z*4 + wave), so the ordinary deterministic combine kernel finishes
// the job.  Without it 3 of 4 waves idle on a one-tile output.
// ---- split-K combine inside the GEMM launch (round 3) -------------------------------------------------------------
// A split-K launch used to be followed by a combine launch (gemm_splitk_reduce_*): 22 of them per training step, 5 us
// each alone -- but 20-60 us (up to 350) on the side stream beside a persistent recurrent kernel, i.e. 0.6 ms of
// side-stream time per step.  With `tickets` the K slices of an output tile write their partial tiles with
// WRITE-THROUGH stores, take a ticket (cdna_hip_programming.md Guideline 16, the sc1 form: stores -> vmcnt(0) ->
// barrier -> relaxed agent fetch_add), and the LAST one to arrive adds all slices of the tile in slice order (sc1
// loads; the same order as the flat combine kernel, whichever slice happens to be last: deterministic) and applies
// the epilogue.  Tickets come from a zero-initialised per-module pool and are reset by their last arriver.
#define D2P_GEMM_TICKETS 65536
static __device__ unsigned g_gemm_tickets[D2P_GEMM_TICKETS];
// (buffer instructions with the sc1 policy, not relaxed atomics: the compiler keeps atomic loads in program order, one
//  L2 round trip after the other -- 64 of them per lane made the folded launch SLOWER than launch + combine)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t d2p_wt_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void d2p_st_wt(__amdgpu_buffer_rsrc_t r, int idx, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, idx * 4, 0, 16);
}
__device__ __forceinline__ float d2p_ld_wt(__amdgpu_buffer_rsrc_t r, int idx) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, idx * 4, 0, 16));
}
// true in every thread of the workgroup that drew the last ticket of its tile (flag: one LDS word that is free)
__device__ __forceinline__ bool d2p_last_slice(unsigned* ticket, unsigned nz, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t == nz - 1u;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last ? 1 : 0;
    }
    __syncthreads();
    return *flag != 0;
}

template <int BM, int BN, int WM, int WN, int BK, bool FAST, bool KS, class AL, class BL, class EP, bool NOSEL = false>
__global__ void __launch_bounds__(256)
gemm_mfma_kernel(AL al, BL bl, EP ep, int M, int N, int K, int k_per_split, float* partial, unsigned* tickets) {
    static_assert(!NOSEL || FAST, "NOSEL is a refinement of the fast loaders");
    static_assert(KS || WM * WN == 4, "4 waves per workgroup");
    // threads: one wave per wave tile
    constexpr int NT = KS ? 256 : 64 * WM * WN;
    static_assert(!KS || (WM == 1 && WN == 1 && BK == 32), "KS: one tile for all waves, 4 chunks per slab");
    static_assert(BK % 8 == 0, "K slab is consumed in chunks of 8");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
    // KS on a single 32x32 tile: the four waves' partial tiles are summed through LDS, so a
    // workgroup writes finished values (or ONE split-K slab) -- the small-problem path
    constexpr bool KSR = KS && BM == 32 && BN == 32;
    constexpr int A_LD = AL::KCONTIG ? (BK + 4) : BM;
    constexpr int B_LD = BL::KCONTIG ? (BK + 4) : BN;
    constexpr int A_SZ = AL::KCONTIG ? BM * (BK + 4) : BK * BM;
    constexpr int B_SZ = BL::KCONTIG ? BN * (BK + 4) : BK * BN;
    constexpr int CA = BM * BK / 4, CB = BN * BK / 4;           // float4 slots per slab
    constexpr int IA = (CA + NT - 1) / NT, IB = (CB + NT - 1) / NT; // per-thread staging loads
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_SZ + B_SZ)];

    if constexpr (d2p_nosel_ok<AL>::value && d2p_nosel_ok<BL>::value && d2p_batch_ok<EP>::value) {
        if (gridDim.y > 1) {          // strided batch: this workgroup's problem
            al.shift(blockIdx.y);
            bl.shift(blockIdx.y);
            ep.shift(blockIdx.y);
        }
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, hi = lane >> 5;
    const int wm = KS ? 0 : wave / WN, wn = KS ? 0 : wave % WN;

    const int nbn = (N + BN - 1) / BN;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int kbeg = blockIdx.z * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two register staging sets: the loads of K-slab kt+2 are issued before the MFMAs of slab
    // kt and only written to LDS after the MFMAs of slab kt+1, so each global load has two
    // slabs of MFMA time (plus whatever other resident workgroups contribute) to land.
    float ra0[IA][4], rb0[IB][4], ra1[IA][4], rb1[IB][4];
    bool oa0[IA], ob0[IB], oa1[IA], ob1[IB];

    // NOSEL: this thread's operand pointers for slab 0, rows / columns clamped into range ONCE (an
    // out-of-range row or column only feeds outputs that are never stored); a slab then costs one
    // 64-bit add per load -- the slab offset is wave-uniform -- instead of the multiply / compare /
    // select chain of the generic loaders (4.7 VALU instructions per MFMA before, SQ_INSTS_VALU).
    const float* pa[IA];
    const float* pb[IB];
    long astep = 0, bstep = 0;                                   // floats per K slab
    // K-gathering loaders (GatherXC): this thread's k index in slab 0 and two sets of row indices -- the set of
    // slab kt+1 is requested at the START of the loads of slab kt (before its data loads, so that waiting for it
    // one call later does not wait for them); the two sets swap roles like the staging registers
    int ka[IA], kb[IB], ja0[IA], jb0[IB], ja1[IA], jb1[IB];
    if constexpr (NOSEL) {
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int q = tid + i * NT;
            pa[i] = al.p;
            if (CA % NT == 0 || q < CA) {
                if constexpr (AL::KCONTIG) pa[i] = al.row_ptr(min(m0 + q / (BK / 4), al.X - 1)) + kbeg + (q % (BK / 4)) * 4;
                else if constexpr (d2p_kgather<AL>::value) pa[i] = al.p + min(m0 + (q % (BM / 4)) * 4, al.X - 4);
                else pa[i] = al.p + (long)(kbeg + q / (BM / 4)) * al.ld + min(m0 + (q % (BM / 4)) * 4, al.X - 4);
            }
            ka[i] = kbeg + ((CA % NT == 0 || q < CA) ? q / (BM / 4) : 0);
            if constexpr (d2p_kgather<AL>::value) ja0[i] = al.idx[ka[i]];
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int q = tid + i * NT;
            pb[i] = bl.p;
            if (CB % NT == 0 || q < CB) {
                if constexpr (BL::KCONTIG) pb[i] = bl.row_ptr(min(n0 + q / (BK / 4), bl.X - 1)) + kbeg + (q % (BK / 4)) * 4;
                else if constexpr (d2p_kgather<BL>::value) pb[i] = bl.p + min(n0 + (q % (BN / 4)) * 4, bl.X - 4);
                else pb[i] = bl.p + (long)(kbeg + q / (BN / 4)) * bl.ld + min(n0 + (q % (BN / 4)) * 4, bl.X - 4);
            }
            kb[i] = kbeg + ((CB % NT == 0 || q < CB) ? q / (BN / 4) : 0);
            if constexpr (d2p_kgather<BL>::value) jb0[i] = bl.idx[kb[i]];
        }
        astep = AL::KCONTIG ? (long)BK : (long)BK * al.ld;
        bstep = BL::KCONTIG ? (long)BK : (long)BK * bl.ld;
    }

#define D2P_GLOAD(RA, RB, OA, OB, kt, JAC, JBC, JAN, JBN)                                          \
    if constexpr (NOSEL) {                                                                         \
        const long so_ = (long)min((int)(kt), nk - 1);          /* prefetches past the end re-read the last slab */ \
        if constexpr (d2p_kgather<AL>::value) {                                                    \
            const int sn_ = min((int)(kt) + 1, nk - 1) * BK;                                       \
            _Pragma("unroll") for (int i = 0; i < IA; ++i) JAN[i] = al.idx[ka[i] + sn_];           \
        }                                                                                          \
        if constexpr (d2p_kgather<BL>::value) {                                                    \
            const int sn_ = min((int)(kt) + 1, nk - 1) * BK;                                       \
            _Pragma("unroll") for (int i = 0; i < IB; ++i) JBN[i] = bl.idx[kb[i] + sn_];           \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < IA; ++i) {                                           \
            const float* src_ = pa[i] + so_ * astep;                                               \
            if constexpr (d2p_kgather<AL>::value) src_ = pa[i] + (long)JAC[i] * al.ld;             \
            const float4 t_ = *reinterpret_cast<const float4*>(src_);                              \
            RA[i][0] = t_.x; RA[i][1] = t_.y; RA[i][2] = t_.z; RA[i][3] = t_.w; OA[i] = true;      \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < IB; ++i) {                                           \
            const float* src_ = pb[i] + so_ * bstep;                                               \
            if constexpr (d2p_kgather<BL>::value) src_ = pb[i] + (long)JBC[i] * bl.ld;             \
            const float4 t_ = *reinterpret_cast<const float4*>(src_);                              \
            RB[i][0] = t_.x; RB[i][1] = t_.y; RB[i][2] = t_.z; RB[i][3] = t_.w; OB[i] = true;      \
        }                                                                                          \
    } else {                                                                                       \
        const int k0 = kbeg + (kt) * BK;                                                           \
        _Pragma("unroll") for (int i = 0; i < IA; ++i) {                                           \
            const int q = tid + i * NT;                                                           \
            if (CA % NT == 0 || q < CA) {                                                         \
                if (AL::KCONTIG) OA[i] = al.template load4<FAST>(m0 + q / (BK / 4), k0 + (q % (BK / 4)) * 4, kend, RA[i]); \
                else OA[i] = al.template load4<FAST>(m0 + (q % (BM / 4)) * 4, k0 + q / (BM / 4), kend, RA[i]); \
            }                                                                                      \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < IB; ++i) {                                           \
            const int q = tid + i * NT;                                                           \
            if (CB % NT == 0 || q < CB) {                                                         \
                if (BL::KCONTIG) OB[i] = bl.template load4<FAST>(n0 + q / (BK / 4), k0 + (q % (BK / 4)) * 4, kend, RB[i]); \
                else OB[i] = bl.template load4<FAST>(n0 + (q % (BN / 4)) * 4, k0 + q / (BN / 4), kend, RB[i]); \
            }                                                                                      \
        }                                                                                          \
    }
#define D2P_SSTORE(RA, RB, OA, OB, buf)                                                            \
    {                                                                                              \
        float* As = smem + (buf) * (A_SZ + B_SZ);                                                  \
        float* Bs = As + A_SZ;                                                                     \
        _Pragma("unroll") for (int i = 0; i < IA; ++i) {                                           \
            const int q = tid + i * NT;                                                           \
            if (CA % NT == 0 || q < CA) {                                                         \
                const bool k_ = NOSEL || OA[i];                                                    \
                float4 t = make_float4(k_ ? RA[i][0] : 0.f, k_ ? RA[i][1] : 0.f,                   \
                                       k_ ? RA[i][2] : 0.f, k_ ? RA[i][3] : 0.f);                  \
                if (AL::KCONTIG) *reinterpret_cast<float4*>(&As[(q / (BK / 4)) * A_LD + (q % (BK / 4)) * 4]) = t; \
                else *reinterpret_cast<float4*>(&As[(q / (BM / 4)) * A_LD + (q % (BM / 4)) * 4]) = t; \
            }                                                                                      \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < IB; ++i) {                                           \
            const int q = tid + i * NT;                                                           \
            if (CB % NT == 0 || q < CB) {                                                         \
                const bool k_ = NOSEL || OB[i];                                                    \
                float4 t = make_float4(k_ ? RB[i][0] : 0.f, k_ ? RB[i][1] : 0.f,                   \
                                       k_ ? RB[i][2] : 0.f, k_ ? RB[i][3] : 0.f);                  \
                if (BL::KCONTIG) *reinterpret_cast<float4*>(&Bs[(q / (BK / 4)) * B_LD + (q % (BK / 4)) * 4]) = t; \
                else *reinterpret_cast<float4*>(&Bs[(q / (BN / 4)) * B_LD + (q % (BN / 4)) * 4]) = t; \
            }                                                                                      \
        }                                                                                          \
    }
    auto compute = [&](int buf) {
        const float* As = smem + buf * (A_SZ + B_SZ);
        const float* Bs = As + A_SZ;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            if (KS && kk != wave) continue;
            float fa[TM][4], fb[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int x = wm * (BM / WM) + i * 32 + l32;
                if (AL::KCONTIG) {
                    float4 t = *reinterpret_cast<const float4*>(&As[x * A_LD + kk * 8 + 4 * hi]);
                    fa[i][0] = t.x; fa[i][1] = t.y; fa[i][2] = t.z; fa[i][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fa[i][j] = As[(kk * 8 + 4 * hi + j) * A_LD + x];
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int x = wn * (BN / WN) + i * 32 + l32;
                if (BL::KCONTIG) {
                    float4 t = *reinterpret_cast<const float4*>(&Bs[x * B_LD + kk * 8 + 4 * hi]);
                    fb[i][0] = t.x; fb[i][1] = t.y; fb[i][2] = t.z; fb[i][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[i][j] = Bs[(kk * 8 + 4 * hi + j) * B_LD + x];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][j], fb[jj][j],
                                                                          acc[i][jj], 0, 0, 0);
        }
    };

    // FAST loaders (no control flow around a load): the steady-state loop is straight-line -- slabs
    // are consumed in pairs with unconditional prefetches (loads past the end of K come from a
    // clamped address and are zeroed or never used), an odd last slab is finished after the loop.
    // With branches in the loop hipcc (a) waits vmcnt(0) at the loop head and (b) for tiles with
    // several accumulators shuffles them between AGPRs and VGPRs every slab (64 v_accvgpr moves
    // per 32 MFMAs in the 128x64 im2col kernel).  For NOSEL nothing sits between a global load
    // and its LDS store; with a select the compiler may pull it (and a wait) up behind the load,
    // which costs the dense kernels 1-3 % -- they take the NOSEL variant whenever K allows.
    // (Fencing the prefetch above and the stores below each slab's MFMAs with
    // __builtin_amdgcn_sched_barrier gives the textbook two-slab distance in the ISA but measured
    // equal on small grids and 5-15 % slower on large ones: left to the scheduler.)
    // The general loaders (scalar tails, branches inside) keep the guarded form.
    if (nk > 0) {
        D2P_GLOAD(ra0, rb0, oa0, ob0, 0, ja0, jb0, ja1, jb1)
        D2P_SSTORE(ra0, rb0, oa0, ob0, 0)
        D2P_GLOAD(ra1, rb1, oa1, ob1, 1, ja1, jb1, ja0, jb0)
        __syncthreads();
        if constexpr (FAST) {
            const int npairs = nk >> 1;
            for (int pr = 0; pr < npairs; ++pr) {
                const int kt = 2 * pr;
                D2P_GLOAD(ra0, rb0, oa0, ob0, kt + 2, ja0, jb0, ja1, jb1)  // in flight during compute(kt) and compute(kt+1)
                compute(0);
                D2P_SSTORE(ra1, rb1, oa1, ob1, 1)
                __syncthreads();
                D2P_GLOAD(ra1, rb1, oa1, ob1, kt + 3, ja1, jb1, ja0, jb0)
                compute(1);
                D2P_SSTORE(ra0, rb0, oa0, ob0, 0)
                __syncthreads();
            }
            if (nk & 1) compute(0);
        } else {
            // loads past the end of K return zeros (klim checks): prefetch unconditionally;
            // slabs nk and nk+1 are never stored
            for (int kt = 0; kt < nk; kt += 2) {
                D2P_GLOAD(ra0, rb0, oa0, ob0, kt + 2, ja0, jb0, ja1, jb1)
                compute(0);
                if (kt + 1 < nk) D2P_SSTORE(ra1, rb1, oa1, ob1, 1)
                __syncthreads();
                if (kt + 1 < nk) {
                    D2P_GLOAD(ra1, rb1, oa1, ob1, kt + 3, ja1, jb1, ja0, jb0)
                    compute(1);
                    if (kt + 2 < nk) D2P_SSTORE(ra0, rb0, oa0, ob0, 0)
                    __syncthreads();
                }
            }
        }
    }
#undef D2P_GLOAD
#undef D2P_SSTORE

    if constexpr (KSR) {
        static_assert(!KSR || 2 * (A_SZ + B_SZ) >= 4 * 1024, "LDS holds the four 32x32 partial tiles");
        __syncthreads();                                   // every wave is done with the last slab
#pragma unroll
        for (int r = 0; r < 16; ++r) smem[wave * 1024 + r * 64 + lane] = acc[0][0][r];
        __syncthreads();
        const bool to_partial = gridDim.z > 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + e * 256;
            const float s = ((smem[idx] + smem[1024 + idx]) + smem[2048 + idx]) + smem[3072 + idx];
            const int r = idx >> 6, ln = idx & 63;
            const int col = n0 + (ln & 31);
            const int row = m0 + 4 * (ln >> 5) + (r & 3) + 8 * (r >> 2);
            if (row < M && col < N) {
                if (to_partial) {
                    if (tickets) d2p_st_wt(d2p_wt_rsrc(partial, (unsigned)gridDim.z * M * N * 4u), ((int)blockIdx.z * M + row) * N + col, s);
                    else partial[((long)blockIdx.z * M + row) * N + col] = s;
                } else ep.store(row, col, s + (ep.has_c() ? ep.c_value(row, col) : 0.f) + ep.col_value(col));
            }
        }
        if (to_partial && tickets) {
            __syncthreads();                               // smem is read above: free from here
            if (!d2p_last_slice(tickets + blockIdx.x, gridDim.z, reinterpret_cast<int*>(smem))) return;
            const int nz = gridDim.z;
            const __amdgpu_buffer_rsrc_t pr = d2p_wt_rsrc(partial, (unsigned)nz * M * N * 4u);
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < nz; ++z) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = tid + e * 256;
                    const int r = idx >> 6, ln = idx & 63;
                    const int col = min(n0 + (ln & 31), N - 1);
                    const int row = min(m0 + 4 * (ln >> 5) + (r & 3) + 8 * (r >> 2), M - 1);
                    s4[e] += d2p_ld_wt(pr, (z * M + row) * N + col);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = tid + e * 256;
                const int r = idx >> 6, ln = idx & 63;
                const int col = n0 + (ln & 31);
                const int row = m0 + 4 * (ln >> 5) + (r & 3) + 8 * (r >> 2);
                if (row < M && col < N) ep(row, col, s4[e]);
            }
        }
        return;
    }
    const bool split = KS || gridDim.z > 1;
    const int slab = KS ? blockIdx.z * 4 + wave : blockIdx.z;
    const bool with_c = !split && ep.has_c();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + l32;
            const int rbase = m0 + wm * (BM / WM) + i * 32 + 4 * hi;
            if (col >= N) continue;
            const float cv = split ? 0.f : ep.col_value(col);       // once per lane and column
            float cold[16];
            if (with_c) {                                           // all C reads in flight together
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    cold[r] = row < M ? ep.c_value(row, col) : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < M) {
                    if (split) {
                        if (!KS && tickets) d2p_st_wt(d2p_wt_rsrc(partial, (unsigned)gridDim.z * M * N * 4u), (slab * M + row) * N + col, acc[i][j][r]);
                        else partial[((long)slab * M + row) * N + col] = acc[i][j][r];
                    } else ep.store(row, col, acc[i][j][r] + (with_c ? cold[r] : 0.f) + cv);
                }
            }
        }
    if constexpr (!KS) {
        if (split && tickets) {
            if (!d2p_last_slice(tickets + blockIdx.x, gridDim.z, reinterpret_cast<int*>(smem))) return;
            const int nz = gridDim.z;
            const __amdgpu_buffer_rsrc_t pr = d2p_wt_rsrc(partial, (unsigned)nz * M * N * 4u);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * (BN / WN) + j * 32 + l32;
                    const int rbase = m0 + wm * (BM / WM) + i * 32 + 4 * hi;
                    if (col >= N) continue;
                    // slice by slice, the 16 rows of a slice in flight together (clamped rows: loaded, never stored)
                    float s16[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) s16[r] = 0.f;
                    for (int z = 0; z < nz; ++z) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = min(rbase + (r & 3) + 8 * (r >> 2), M - 1);
                            s16[r] += d2p_ld_wt(pr, (z * M + row) * N + col);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row < M) ep(row, col, s16[r]);
                    }
                }
        }
    }
}

template <> struct d2p_batch_ok<EpiDense> { static constexpr bool value = true; };

// Deterministic split-K combine: 16 lanes per output element each sum slabs z = l, l+16, ...
// then a fixed-order xor-shuffle tree combines them; EP is applied by lane 0 of the group.
template <class EP>
__global__ void __launch_bounds__(256)
gemm_splitk_reduce_kernel(EP ep, const float* partial, int M, int N, int nz) {
    const long total = (long)M * N;
    const int sub = threadIdx.x & 15;
    for (long base = (blockIdx.x * 256L + threadIdx.x) >> 4; base < ((total + 15) & ~15L);
         base += ((long)gridDim.x * 256L) >> 4) {
        float s = 0.f;
        if (base < total)
            for (int z = sub; z < nz; z += 16) s += partial[(long)z * total + base];
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (sub == 0 && base < total) ep((int)(base / N), (int)(base % N), s);
    }
}

// Few slabs (weight gradients with 4-8 splits over a large output): one thread per element.
template <class EP>
__global__ void __launch_bounds__(256)
gemm_splitk_reduce_flat_kernel(EP ep, const float* partial, int M, int N, int nz) {
    const long total = (long)M * N;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z) s += partial[(long)z * total + idx];
        ep((int)(idx / N), (int)(idx % N), s);
    }
}

// ------------------------------------------------------------------------------------
// LDS-DMA pipeline for the large dense GEMMs (K a multiple of 32, 16-byte aligned operands).
//
// The kernel above stages every operand slab global -> VGPR -> ds_write -> LDS, and hipcc sinks the
// global loads next to their LDS stores: a wave waits a full L2 round trip per slab and the MFMA
// pipe is kept busy only by the other resident waves (0.64 of the fp32 MFMA peak at best, every
// tile shape).  Here the operand slabs go global -> LDS directly (global_load_lds_dwordx4: 64 lanes
// x 16 bytes land at a wave-uniform LDS base + lane*16) into a ring of STAGES slabs, so a slab is
// requested STAGES-1 slabs of MFMA time before it is consumed, costs no VGPRs and no ds_write
// issue, and the only waits in the K loop are one counted vmcnt + one barrier per 32-deep slab.
//
// LDS layouts (unpadded -- the DMA destination is linear in the lane index):
//   KCONTIG operand: [x][32 k] rows of 128 bytes; the 16-byte chunk c of row x sits at chunk
//     position c ^ ((x >> 1) & 7).  A DMA instruction covers 8 rows; its 8 lanes per row read one
//     whole 128-byte line (permuted) -> coalesced; ds_read_b128 of chunk c over 32 rows touches
//     every 16-byte slot of the 256-byte bank row once per 16-lane service group
//     (MI355X_MICROARCH.md, LDS: groups {0-3,12-15,20-27} ...) -> conflict-free.
//   XCONTIG operand: [32 k][BX]; a DMA instruction covers 256 consecutive floats; ds_read_b32 with
//     lanes 0-31 on consecutive floats -> conflict-free.
// The k indices are consumed in the same order as in gemm_mfma_kernel, so the two kernels give
// bit-identical results for the same split of K.
// ------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void d2p_lds_void;
typedef __attribute__((address_space(1))) const void d2p_glb_void;

template <int N>
__device__ __forceinline__ void d2p_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// A fifth wave issues every DMA of the workgroup: hipcc's wait-count pass cannot tell which LDS bytes an
// outstanding global_load_lds will write and puts s_waitcnt vmcnt(0) in front of every ds_read that
// follows one IN THE SAME WAVE -- the prefetch distance collapses to zero.  The producer wave reads no
// LDS; the four MFMA waves issue no DMA.
//
// Workgroups are PERSISTENT: workgroup w takes the work items (output tile x K split) w, w+G, w+2G, ...
// and its producer streams their slabs back to back through the ring, so the first slabs of the next tile
// land while the MFMA waves store the current one: the per-tile prologue (address setup + an L2/HBM
// round trip with an idle matrix pipe) is paid once per workgroup, not once per tile.
#define D2P_DMA_THREADS 320
template <int BM, int BN, int WM, int WN, int STAGES, class AL, class BL, class EP>
__global__ void __launch_bounds__(D2P_DMA_THREADS)
gemm_dma_kernel(AL al, BL bl, EP ep, int M, int N, int K, int k_per_split, int splits, float* partial) {
    static_assert(WM * WN == 4, "4 MFMA waves per workgroup");
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    constexpr int BK = 32;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_SZ = BM * BK, B_SZ = BN * BK, ST = A_SZ + B_SZ;   // floats per stage
    constexpr int NA = BM / 8, NB = BN / 8;        // DMA instructions per slab (64 lanes x 16 bytes = 256 floats each)
    constexpr int NI = NA + NB;
    static_assert((STAGES - 1) * NI <= 96, "the producer keeps at most STAGES-1 slabs of requests in flight");
    extern __shared__ __attribute__((aligned(16))) float dma_smem[];

    if constexpr (d2p_batch_ok<EP>::value) {
        if (gridDim.y > 1) {
            al.shift(blockIdx.y);
            bl.shift(blockIdx.y);
            ep.shift(blockIdx.y);
        }
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int nbn = (N + BN - 1) / BN;
    const int ntiles = ((M + BM - 1) / BM) * nbn;
    const int nitems = ntiles * splits;            // item = split * ntiles + tile
    const int G = gridDim.x;

    if (wave == 4) {
        // ---- producer
        const float* pa[NA];
        const float* pb[NB];
        const long astep = AL::KCONTIG ? (long)BK : (long)BK * al.ld;     // floats per slab
        const long bstep = BL::KCONTIG ? (long)BK : (long)BK * bl.ld;
        int item = blockIdx.x, slab = 0, nk = 0;   // cursor of the NEXT slab to request
        auto open_item = [&]() {                   // per-lane source pointers of the item's slab-0 requests
            const int tile = item % ntiles, sp = item / ntiles;
            const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
            const int kbeg = sp * k_per_split;
            nk = (min(K, kbeg + k_per_split) - kbeg) / BK;
            slab = 0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int u = i * 64 + lane;                              // 16-byte unit of the stage
                if (AL::KCONTIG) {
                    const int row = u >> 3, c = (u & 7) ^ ((row >> 1) & 7);
                    pa[i] = al.p + (long)min(m0 + row, al.X - 1) * al.ld + kbeg + 4 * c;
                } else {
                    const int k = u / (BM / 4), xq = u % (BM / 4);
                    pa[i] = al.p + (long)(kbeg + k) * al.ld + min(m0 + 4 * xq, al.X - 4);
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int u = i * 64 + lane;
                if (BL::KCONTIG) {
                    const int row = u >> 3, c = (u & 7) ^ ((row >> 1) & 7);
                    pb[i] = bl.p + (long)min(n0 + row, bl.X - 1) * bl.ld + kbeg + 4 * c;
                } else {
                    const int k = u / (BN / 4), xq = u % (BN / 4);
                    pb[i] = bl.p + (long)(kbeg + k) * bl.ld + min(n0 + 4 * xq, bl.X - 4);
                }
            }
        };
        int pbuf = 0;
        // requests the cursor's slab into stage pbuf and advances; returns false past the last item
        auto issue = [&]() -> bool {
            if (item >= nitems) return false;
            float* As = dma_smem + pbuf * ST;
            float* Bs = As + A_SZ;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                __builtin_amdgcn_global_load_lds((d2p_glb_void*)(pa[i] + slab * astep), (d2p_lds_void*)(As + i * 256), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < NB; ++i)
                __builtin_amdgcn_global_load_lds((d2p_glb_void*)(pb[i] + slab * bstep), (d2p_lds_void*)(Bs + i * 256), 16, 0, 0);
            pbuf = (pbuf + 1 == STAGES) ? 0 : pbuf + 1;
            if (++slab == nk) {
                item += G;
                if (item < nitems) open_item();
            }
            return true;
        };
        // total slabs of this workgroup
        int Q = 0;
        for (int it = blockIdx.x; it < nitems; it += G) {
            const int kbeg = (it / ntiles) * k_per_split;
            Q += (min(K, kbeg + k_per_split) - kbeg) / BK;
        }
        open_item();
        int ahead = 0;                             // slabs requested and not yet handed over
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s) ahead += issue() ? 1 : 0;
        for (int q = 0; q < Q; ++q) {
            // slab q must have landed: the requests behind it are those of the ahead-1 later slabs
            if (ahead >= STAGES - 1) d2p_wait_vmcnt<(STAGES - 2) * NI>();
            else if (STAGES >= 4 && ahead == 2) d2p_wait_vmcnt<NI>();
            else d2p_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // hand it over; the MFMA waves are done with slab q-1
            ahead += (issue() ? 1 : 0) - 1;        // into the stage slab q-1 occupied
        }
        return;
    }

    const int l32 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // fragment addresses inside a stage (floats)
    const int g8 = (l32 >> 1) & 7;
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int x = wm * (BM / WM) + i * 32 + l32;
        aoff[i] = AL::KCONTIG ? x * BK : x;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int x = wn * (BN / WN) + i * 32 + l32;
        boff[i] = A_SZ + (BL::KCONTIG ? x * BK : x);
    }
    int cp[BK / 8];
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) cp[kk] = ((2 * kk + hi) ^ g8) * 4;

    int buf = 0;
    for (int item = blockIdx.x; item < nitems; item += G) {
        const int tile = item % ntiles, sp = item / ntiles;
        const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;
        const int kbeg = sp * k_per_split;
        const int nk = (min(K, kbeg + k_per_split) - kbeg) / BK;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int s = 0; s < nk; ++s) {
            __builtin_amdgcn_s_barrier();          // slab s is in LDS (the producer waited for it)
            const float* S = dma_smem + buf * ST;
            float fa[2][TM][4], fb[2][TN][4];      // fragments of chunk kk in set kk & 1: read one chunk ahead
            auto read_chunk = [&](int kk, int set) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (AL::KCONTIG) {
                        const float4 t = *reinterpret_cast<const float4*>(&S[aoff[i] + cp[kk]]);
                        fa[set][i][0] = t.x; fa[set][i][1] = t.y; fa[set][i][2] = t.z; fa[set][i][3] = t.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) fa[set][i][j] = S[aoff[i] + (kk * 8 + 4 * hi + j) * BM];
                    }
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    if (BL::KCONTIG) {
                        const float4 t = *reinterpret_cast<const float4*>(&S[boff[i] + cp[kk]]);
                        fb[set][i][0] = t.x; fb[set][i][1] = t.y; fb[set][i][2] = t.z; fb[set][i][3] = t.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) fb[set][i][j] = S[boff[i] + (kk * 8 + 4 * hi + j) * BN];
                    }
                }
            };
            read_chunk(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                if (kk + 1 < BK / 8) read_chunk(kk + 1, (kk + 1) & 1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj)
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i][j], fb[kk & 1][jj][j],
                                                                              acc[i][jj], 0, 0, 0);
                // the next chunk's LDS reads go out BEFORE this chunk's MFMAs (left alone, hipcc sinks them behind
                // the MFMAs and the wave idles an LDS round trip per chunk)
                if (kk + 1 < BK / 8) __builtin_amdgcn_sched_group_barrier(0x100, 4 * (TM + TN), 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
            }
            buf = (buf + 1 == STAGES) ? 0 : buf + 1;
        }

        const bool split = splits > 1;
        const bool with_c = !split && ep.has_c();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (BN / WN) + j * 32 + l32;
                const int rbase = m0 + wm * (BM / WM) + i * 32 + 4 * hi;
                if (col >= N) continue;
                const float cv = split ? 0.f : ep.col_value(col);
                float cold[16];
                if (with_c) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        cold[r] = row < M ? ep.c_value(row, col) : 0.f;
                    }
                }
                // every load of the epilogue is complete before its first store: the K loop of the next item
                // then never waits on vmcnt (on gfx9 that counter also holds the stores -- a wait there would
                // stall the next tile on this tile's write-back)
                __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < M) {
                        if (split) partial[((long)sp * M + row) * N + col] = acc[i][j][r];
                        else ep.store(row, col, acc[i][j][r] + (with_c ? cold[r] : 0.f) + cv);
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------
// Host-side launch policy shared by gemm.hip and conv.hip.
// ------------------------------------------------------------------------------------
static int g_gemm_bk32 = 0;   // per translation unit; toggled through d2p_gemm_set_option
static int g_gemm_nosel = 1;  // 0: always keep the select between global load and LDS store
static int g_gemm_force_tile = -1;   // -1: automatic; else a GemmTile value (tuning experiments)
static int g_gemm_force_split = 0;   // 0: automatic
static int g_gemm_dma_grid = 0;      // 0: CUs x resident workgroups per CU; else the persistent grid (tuning)
static inline int d2p_gemm_num_cus() {
    static int n = 0;
    if (n <= 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else return 256;
    }
    return n;
}

enum GemmTile { TILE_64x64 = 0, TILE_128x128 = 1, TILE_128x32 = 2, TILE_256x32 = 3, TILE_128x64 = 4,
                TILE_64x32_KS = 5, TILE_64x64_KS = 6, TILE_32x32_KSR = 7,
                // LDS-DMA pipeline (gemm_dma_kernel), dense operands with K % 32 == 0 only
                TILE_DMA_64x64_S4 = 8, TILE_DMA_64x64_S3 = 9, TILE_DMA_128x64_S3 = 10, TILE_DMA_128x128_S2 = 11,
                TILE_DMA_128x128_S3 = 12, TILE_COUNT = 13 };
static inline bool d2p_tile_is_dma(int t) { return t >= TILE_DMA_64x64_S4 && t < TILE_COUNT; }
static inline int d2p_tile_stages(int t) {
    return t == TILE_DMA_64x64_S4 ? 4 : (t == TILE_DMA_128x128_S2 ? 2 : 3);
}

struct GemmPlan {
    int tile;     // GemmTile
    int bm, bn;
    int splits;   // grid.z
    int k_per_split;
    int slabs;    // partial slabs written (splits, or 4*splits in KS mode); 1 = no combine pass
};

static inline GemmPlan d2p_plan_gemm_base(int M, int N, int K, bool allow_split) {
    GemmPlan p;
    const long t128 = (long)ceil_div(M, 128) * ceil_div(N, 128);
    // Measured on MI355X (tools/bench_gemm.py): every tile shape tops out near 100 TFLOP/s, so
    // the choice is about wave quantisation over 256 CUs -- 6400x2048 runs 134 us on 64x64
    // (3200 workgroups) vs 164 us on 128x128 (800 workgroups = 3.1 per CU -> 4 rounds).
    if (N <= 32) {                       // skinny outputs: keep every MFMA column useful
        const long t256 = (long)ceil_div(M, 256);
        if (t256 >= 1024) { p.tile = TILE_256x32; p.bm = 256; p.bn = 32; }
        else { p.tile = TILE_128x32; p.bm = 128; p.bn = 32; }
    } else if (N <= 64 && (long)ceil_div(M, 128) >= 512) {   // e.g. conv Cout = 48
        p.tile = TILE_128x64; p.bm = 128; p.bn = 64;
    } else if (t128 >= 2048) {           // >= 8 full rounds at the large tile: L2 traffic wins
        p.tile = TILE_128x128; p.bm = 128; p.bn = 128;
    } else if (N <= 512 && M >= 4096) {  // tall, medium-width outputs (dX = dZ Wx^T)
        p.tile = TILE_128x64; p.bm = 128; p.bn = 64;
    } else {
        p.tile = TILE_64x64; p.bm = 64; p.bn = 64;
    }
    // tiny output, huge K: K split across the waves of a workgroup as well as across workgroups
    if (allow_split && K >= 8192 && N <= 64 &&
        (long)ceil_div(M, 64) * ceil_div(N, N <= 32 ? 32 : 64) <= 8) {
        if (N <= 32) { p.tile = TILE_64x32_KS; p.bm = 64; p.bn = 32; }
        else { p.tile = TILE_64x64_KS; p.bm = 64; p.bn = 64; }
    }
    if (g_gemm_force_tile >= 0) {
        p.tile = g_gemm_force_tile;
        const int bms[TILE_COUNT] = {64, 128, 128, 256, 128, 64, 64, 32, 64, 64, 128, 128, 128};
        const int bns[TILE_COUNT] = {64, 128, 32, 32, 64, 32, 64, 32, 64, 64, 64, 128, 128};
        p.bm = bms[p.tile]; p.bn = bns[p.tile];
    }
    if (p.tile == TILE_32x32_KSR) {                      // forced (tuning): one slab per workgroup
        int s = (allow_split && g_gemm_force_split > 1) ? g_gemm_force_split : 1;
        int kps = ((K + s - 1) / s + 31) / 32 * 32;
        p.k_per_split = kps;
        p.splits = (K + kps - 1) / kps;
        p.slabs = p.splits;
        return p;
    }
    const bool ks = (p.tile == TILE_64x32_KS || p.tile == TILE_64x64_KS);
    p.splits = 1;
    p.k_per_split = K;
    p.slabs = ks ? 4 : 1;
    const long tiles = (long)ceil_div(M, p.bm) * ceil_div(N, p.bn);
    if (ks) {
        long s = 1024 / tiles;                           // ~4 workgroups per CU
        const long maxs = K / 1024;                      // >= 32 slabs of 32 per workgroup
        if (s > maxs) s = maxs;
        if (s < 1) s = 1;
        int kps = (int)((K + s - 1) / s);
        kps = (kps + 31) / 32 * 32;
        p.k_per_split = kps;
        p.splits = (K + kps - 1) / kps;
        p.slabs = 4 * p.splits;
        return p;
    }
    if (allow_split && g_gemm_force_split > 1) {
        int kps = (K + g_gemm_force_split - 1) / g_gemm_force_split;
        kps = (kps + 31) / 32 * 32;
        p.k_per_split = kps;
        p.splits = (K + kps - 1) / kps;
        p.slabs = p.splits;
        return p;
    }
    if (allow_split && tiles <= 512 && K >= 1024) {
        long want = (1024 + tiles - 1) / tiles;          // aim at ~4 workgroups per CU
        long maxs = K / 512;                             // keep >= 512 of K per split
        long s = want < maxs ? want : maxs;
        if (s > 1024) s = 1024;                          // bound the combine pass
        if (s > 1) {
            int kps = (int)((K + s - 1) / s);
            kps = (kps + 31) / 32 * 32;
            p.k_per_split = kps;
            p.splits = (K + kps - 1) / kps;
            p.slabs = p.splits;
        }
    }
    return p;
}

static int g_gemm_small_ksr = 1;   // 0: never pick the 32x32 wave-split tile automatically
static int g_gemm_fold = 1;        // split-K combine inside the GEMM launch (0: the separate combine launches)
// a slice of this module's ticket pool (ONE cursor for every instantiation of the launcher: GEMMs of two streams may be
// in flight together and must not share tickets; a slice is reused after the pool has gone round once)
static inline unsigned* d2p_gemm_take_tickets(unsigned ntiles) {
    static unsigned* pool = nullptr;
    static unsigned next = 0;
    if (!pool) (void)hipGetSymbolAddress((void**)&pool, HIP_SYMBOL(g_gemm_tickets));
    if (!pool || ntiles > D2P_GEMM_TICKETS / 4) return nullptr;
    if (next + ntiles > D2P_GEMM_TICKETS) next = 0;
    unsigned* t = pool + next;
    next += ntiles;
    return t;
}
static int g_gemm_no_bk32 = 0;     // experiment: never the 32-deep slabs (16 KB of LDS per workgroup instead of 32)
static int g_gemm_dma_big = 0;     // experiment: large dense GEMMs (>= 2 GFLOP) on the persistent LDS-DMA kernel

static inline GemmPlan d2p_plan_gemm(int M, int N, int K, bool allow_split) {
    GemmPlan p = d2p_plan_gemm_base(M, N, K, allow_split);
    if (g_gemm_force_tile >= 0 || g_gemm_force_split > 1 || !g_gemm_small_ksr) return p;
    // Small problems: the plan above would occupy fewer than half the CUs, each workgroup walking a
    // long K loop alone (measured 19 us for 320x512x512 on 40 workgroups).  32x32 tiles give 4x
    // the workgroups, the four waves of each split K between them and combine through LDS.
    const long wgs = (long)ceil_div(M, p.bm) * ceil_div(N, p.bn) * p.splits;
    // (a tall output of <= 64 columns with a long K -- the first encoder's input gradient, 4480 x 48 over K = 2048 --
    //  is 35 tiles of 128x64 x 4 slices = 140 workgroups: 34 us; 280 tiles of 32x32 x 4 slices: 23 us)
    const long t32 = (long)ceil_div(M, 32) * ceil_div(N, 32);
    // (only while the 32x32 tiles still get a K split, t32 < 512: past that each of ~1000 small workgroups would walk
    //  the whole K alone -- the one measured shape is 4480x48x2048)
    const bool tall_skinny = N <= 64 && K >= 1024 && wgs < 256 && allow_split && t32 < 512;
    if ((wgs >= 128 && !tall_skinny) || K < 128) return p;
    long s = 1;
    if (allow_split && K >= 1024 && (t32 < 256 || (tall_skinny && t32 < 512))) {
        s = ((t32 < 256 ? 512 : 1024) + t32 - 1) / t32;  // ~2 (4) workgroups per CU
        const long maxs = K / 256;                       // >= 64 of K per wave
        if (s > maxs) s = maxs;
        if (s < 1) s = 1;
    }
    p.tile = TILE_32x32_KSR; p.bm = 32; p.bn = 32;
    int kps = (int)(((K + s - 1) / s + 31) / 32 * 32);
    p.k_per_split = kps;
    p.splits = (K + kps - 1) / kps;
    p.slabs = p.splits;
    return p;
}

static inline size_t d2p_plan_ws_bytes(int M, int N, int K) {
    GemmPlan p = d2p_plan_gemm(M, N, K, true);
    return p.slabs > 1 ? (size_t)p.slabs * M * N * sizeof(float) : 0;
}

template <int BM, int BN, int WM, int WN, int BK, bool KS = false, class AL, class BL, class EP>
static void d2p_launch_tile(const AL& al, const BL& bl, const EP& ep, int M, int N, int K,
                            const GemmPlan& p, bool fast, float* partial, hipStream_t st, int batch = 1,
                            unsigned* tickets = nullptr) {
    dim3 grid(ceil_div(M, BM) * ceil_div(N, BN), batch, p.splits);
    constexpr int NT = KS ? 256 : 64 * WM * WN;
    constexpr bool NOSEL_OK = d2p_nosel_ok<AL>::value && d2p_nosel_ok<BL>::value;
    if (NOSEL_OK && fast && g_gemm_nosel && K % BK == 0 && (p.splits == 1 || p.k_per_split % BK == 0)) {
        if constexpr (NOSEL_OK)
            hipLaunchKernelGGL((gemm_mfma_kernel<BM, BN, WM, WN, BK, true, KS, AL, BL, EP, true>), grid, dim3(NT), 0,
                               st, al, bl, ep, M, N, K, p.k_per_split, partial, tickets);
    } else if (fast)
        hipLaunchKernelGGL((gemm_mfma_kernel<BM, BN, WM, WN, BK, true, KS, AL, BL, EP>), grid, dim3(NT), 0, st,
                           al, bl, ep, M, N, K, p.k_per_split, partial, tickets);
    else
        hipLaunchKernelGGL((gemm_mfma_kernel<BM, BN, WM, WN, BK, false, KS, AL, BL, EP>), grid, dim3(NT), 0, st,
                           al, bl, ep, M, N, K, p.k_per_split, partial, tickets);
}

template <int BM, int BN, int WM, int WN, int STAGES, class AL, class BL, class EP>
static int d2p_launch_dma_tile(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, const GemmPlan& p,
                               float* partial, hipStream_t st, int batch) {
    constexpr int lds_bytes = STAGES * (BM + BN) * 32 * (int)sizeof(float);
    auto kern = gemm_dma_kernel<BM, BN, WM, WN, STAGES, AL, BL, EP>;
    static bool attr_set = false;
    if (!attr_set) {
        D2P_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    // persistent workgroups: as many as fit the chip at once (LDS-bound), never more than the work items
    const int per_cu = (160 * 1024) / lds_bytes;
    const long items = (long)ceil_div(M, BM) * ceil_div(N, BN) * p.splits;
    long g = (long)d2p_gemm_num_cus() * per_cu;
    if (g_gemm_dma_grid > 0) g = g_gemm_dma_grid;
    if (g > items) g = items;
    dim3 grid((unsigned)g, batch, 1);
    hipLaunchKernelGGL(kern, grid, dim3(D2P_DMA_THREADS), lds_bytes, st, al, bl, ep, M, N, K, p.k_per_split, p.splits,
                       partial);
    return D2P_OK;
}

// dense operands the DMA pipeline can take: 16-byte aligned rows, whole 32-deep slabs in every split, and at
// least STAGES-1 slabs in the shortest split
template <class AL, class BL>
static inline bool d2p_dma_shape_ok(const AL& al, const BL& bl, int K, const GemmPlan& p, int stages) {
    if (!(al.fast_ok(K) && bl.fast_ok(K)) || K % 32 != 0 || p.k_per_split % 32 != 0) return false;
    const int last = K - (p.splits - 1) * p.k_per_split;
    return last >= (stages - 1) * 32 && p.k_per_split >= (stages - 1) * 32;
}

template <class AL, class BL, class EP>
static int d2p_launch_gemm(const AL& al, const BL& bl, const EP& ep, int M, int N, int K,
                           void* ws, size_t ws_bytes, hipStream_t st, const char* name,
                           int prof_family = D2P_PROF_GEMM, int batch = 1) {
    if (M <= 0 || N <= 0 || batch <= 0) return D2P_OK;
    D2pProfScope prof(st, prof_family, 2.0 * M * N * K * batch);
    // a strided batch (grid.y) never splits K: the slabs of different problems would need their own scratch
    GemmPlan p = d2p_plan_gemm(M, N, K, ws != nullptr && batch == 1);
    if (g_gemm_dma_big && g_gemm_force_tile < 0 && 2.0 * M * N * K >= 2e9 && p.bm == 64 && p.bn == 64)
        p.tile = TILE_DMA_64x64_S4;
    float* partial = nullptr;
    if (p.slabs > 1) {
        const size_t need = (size_t)p.slabs * M * N * sizeof(float);
        if (ws_bytes < need) {   // not enough scratch: fall back to a single pass
            p = d2p_plan_gemm(M, N, K, false);
        } else {
            partial = (float*)ws;
        }
    }
    const bool fast = al.fast_ok(K) && bl.fast_ok(K);
    // split-K combine inside the launch: plain and 32x32 wave-split tiles (the KS tiles write one slab per WAVE and keep
    // the combine pass); a slice of the module's ticket pool, round-robin
    unsigned* tickets = nullptr;
    if (p.slabs > 1 && p.slabs <= 16 && (double)p.slabs * M * N * 4.0 < 2e9 && g_gemm_fold && !d2p_tile_is_dma(p.tile) &&
        p.tile != TILE_64x32_KS && p.tile != TILE_64x64_KS) {
        tickets = d2p_gemm_take_tickets((unsigned)(ceil_div(M, p.bm) * ceil_div(N, p.bn)));
    }
    if (d2p_tile_is_dma(p.tile)) {
        bool done = false;
        if constexpr (d2p_nosel_ok<AL>::value && d2p_nosel_ok<BL>::value && d2p_batch_ok<EP>::value) {
            if (d2p_dma_shape_ok(al, bl, K, p, d2p_tile_stages(p.tile))) {
                int rc = D2P_OK;
                switch (p.tile) {
                    case TILE_DMA_64x64_S4: rc = d2p_launch_dma_tile<64, 64, 2, 2, 4>(al, bl, ep, M, N, K, p, partial, st, batch); break;
                    case TILE_DMA_64x64_S3: rc = d2p_launch_dma_tile<64, 64, 2, 2, 3>(al, bl, ep, M, N, K, p, partial, st, batch); break;
                    case TILE_DMA_128x64_S3: rc = d2p_launch_dma_tile<128, 64, 2, 2, 3>(al, bl, ep, M, N, K, p, partial, st, batch); break;
                    case TILE_DMA_128x128_S2: rc = d2p_launch_dma_tile<128, 128, 2, 2, 2>(al, bl, ep, M, N, K, p, partial, st, batch); break;
                    default: rc = d2p_launch_dma_tile<128, 128, 2, 2, 3>(al, bl, ep, M, N, K, p, partial, st, batch); break;
                }
                if (rc) return rc;
                done = true;
            }
        }
        if (!done) p.tile = (p.bm == 64) ? TILE_64x64 : (p.bn == 64 ? TILE_128x64 : TILE_128x128);   // staged form
    }
    switch (p.tile) {
        case TILE_DMA_64x64_S4: case TILE_DMA_64x64_S3: case TILE_DMA_128x64_S3: case TILE_DMA_128x128_S2:
        case TILE_DMA_128x128_S3: break;            // launched above
        case TILE_128x128: d2p_launch_tile<128, 128, 2, 2, 16>(al, bl, ep, M, N, K, p, fast, partial, st, batch, tickets); break;
        case TILE_128x32: d2p_launch_tile<128, 32, 4, 1, 16>(al, bl, ep, M, N, K, p, fast, partial, st, batch, tickets); break;
        case TILE_256x32: d2p_launch_tile<256, 32, 4, 1, 16>(al, bl, ep, M, N, K, p, fast, partial, st, batch, tickets); break;
        case TILE_128x64: d2p_launch_tile<128, 64, 2, 2, 16>(al, bl, ep, M, N, K, p, fast, partial, st, batch, tickets); break;
        case TILE_64x32_KS: d2p_launch_tile<64, 32, 1, 1, 32, true>(al, bl, ep, M, N, K, p, fast, partial, st, batch); break;
        case TILE_64x64_KS: d2p_launch_tile<64, 64, 1, 1, 32, true>(al, bl, ep, M, N, K, p, fast, partial, st, batch); break;
        case TILE_32x32_KSR: d2p_launch_tile<32, 32, 1, 1, 32, true>(al, bl, ep, M, N, K, p, fast, partial, st, batch, tickets); break;
        default:
            // 32-deep slabs (half the barriers, 128-byte runs): +3-5 % on the large grids, a loss when
            // the workgroups are few (each then walks its K loop with less overlap)
            if (!g_gemm_no_bk32 && (g_gemm_bk32 || (long)ceil_div(M, 64) * ceil_div(N, 64) * p.splits >= 1024) && K >= 256 && fast)
                d2p_launch_tile<64, 64, 2, 2, 32>(al, bl, ep, M, N, K, p, fast, partial, st, batch, tickets);
            else
                d2p_launch_tile<64, 64, 2, 2, 16>(al, bl, ep, M, N, K, p, fast, partial, st, batch, tickets);
            break;
    }
    D2P_LAUNCH_CHECK(name);
    if (p.slabs > 1 && !tickets) {
        const long total = (long)M * N;
        if (p.slabs <= 16) {
            int blocks = (int)((total + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL((gemm_splitk_reduce_flat_kernel<EP>), dim3(blocks), dim3(256), 0, st, ep,
                               partial, M, N, p.slabs);
        } else {
            int blocks = (int)((total * 16 + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL((gemm_splitk_reduce_kernel<EP>), dim3(blocks), dim3(256), 0, st, ep,
                               partial, M, N, p.slabs);
        }
        D2P_LAUNCH_CHECK("gemm_splitk_reduce");
    }
    return D2P_OK;
}
