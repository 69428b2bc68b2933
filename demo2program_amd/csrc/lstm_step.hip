// Fused recurrent-step kernels: per timestep ONE launch computes h·Wh (fp32 MFMA), adds the
// hoisted input projection, applies the BasicLSTMCell gates and the dynamic_rnn length mask
// (forward), or computes dz[t+1]·Wh^T and the gate backward of step t (backward).
// Replaces the per-step matmul + 5 elementwise TF ops inside tf.nn.dynamic_rnn /
// seq2seq.dynamic_decode (models/model_full.py:254-256,274-276,469-471).
//
// Why this shape (MI355X_MICROARCH.md price list): every step needs ALL of h, so the step
// boundary is an all-to-all seam; a kernel boundary (~1.5 us) is cheaper than a grid barrier
// (4-7 us), so the loop stays one launch per step.  Inside a launch the problem
// ([320 x 512] x [512 x 2048], 0.67 GFLOP = 4.3 us at fp32-MFMA peak) is latency-bound, so:
//   * 256 workgroups (one per CU), each owning a [~80 rows] x [8 units x 4 gates] output tile;
//     block -> tile mapping keeps a tile column's row-tiles on one XCD (shared L2 slice of Wh);
//   * the 4 waves of a workgroup split K (no operand is shared between waves, so nothing is
//     staged through LDS); each wave issues ALL loads of its K slice up front (<= 56 x 1 KiB
//     in flight per wave) and then runs its v_mfma_f32_16x16x4_f32 chain;
//   * operands are stored "fragment-major": a [16 rows x 16 k] block is 64 lanes x 4 floats
//     laid out so that lane l's float4 is (row l&15, k = 4*(l>>4)..+3).  A wave loads a block
//     with one perfectly coalesced dwordx4 (1 KiB) and uses component j in the j-th of 4
//     MFMAs (A and B agree on the k permutation).  Wh is packed once per sequence call; h / dz
//     are emitted in this layout by the previous step's epilogue;
//   * the 4 K-partials are combined through LDS (row stride 36 floats: conflict-free b32
//     writes, 16-byte aligned b128 reads) and the epilogue handles (row, 4 units) per thread
//     with 16-byte global accesses.
#include "common.h"
#include "lstm_math.h"
#include "lstm_internal.h"
#include "prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FWD_RSMAX 5     // row sub-tiles (16 rows) per workgroup, forward
#define BWD_RSMAX 3     // backward
// Both kernels are held under 256 registers (VGPR+AGPR) so that TWO workgroups are
// resident per CU: one workgroup's load / LDS-combine / epilogue phases then overlap the
// other's MFMA chain (a single resident workgroup leaves the matrix pipe idle ~60 % of a step).
#define P_LD 36         // LDS partial-tile row stride (floats)

// ---------------------------------------------------------------------------------------
// Packing kernels (HBM-bound copies, once per sequence call / per initial state)
// ---------------------------------------------------------------------------------------
// (element formulas: lstm_internal.h)
__global__ void __launch_bounds__(256)
pack_w_fwd_kernel(int U, const float* __restrict__ Wh, float4* __restrict__ Wf) {
    const long total = (long)(U >> 3) * (U >> 4) * 2 * 64;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L)
        Wf[idx] = d2p_pack_w_fwd_elem(U, Wh, idx);
}
__global__ void __launch_bounds__(256)
pack_w_bwd_kernel(int U, const float* __restrict__ Wh, float4* __restrict__ Wb) {
    const long total = (long)(U >> 4) * (U >> 2) * 64;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L)
        Wb[idx] = d2p_pack_w_bwd_elem(U, Wh, idx);
}
__global__ void __launch_bounds__(256)
pack_rows_kernel(int M, int K, int total_rs, const float* __restrict__ X, float4* __restrict__ Af) {
    const long total = (long)total_rs * (K >> 4) * 64;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L)
        Af[idx] = d2p_pack_rows_elem(M, K, X, idx);
}

static inline int pack_blocks(long total) {
    long b = (total + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

int d2p_lstm_pack_w_fwd(int U, const float* Wh, float* Wf, hipStream_t st) {
    hipLaunchKernelGGL(pack_w_fwd_kernel, dim3(pack_blocks((long)U * U / 4)), dim3(256), 0, st, U, Wh, (float4*)Wf);
    D2P_LAUNCH_CHECK("pack_w_fwd");
    return D2P_OK;
}
int d2p_lstm_pack_w_bwd(int U, const float* Wh, float* Wb, hipStream_t st) {
    hipLaunchKernelGGL(pack_w_bwd_kernel, dim3(pack_blocks((long)U * U / 4)), dim3(256), 0, st, U, Wh, (float4*)Wb);
    D2P_LAUNCH_CHECK("pack_w_bwd");
    return D2P_OK;
}
int d2p_lstm_pack_rows(int M, int K, int total_rs, const float* X, float* Af, hipStream_t st) {
    hipLaunchKernelGGL(pack_rows_kernel, dim3(pack_blocks((long)total_rs * 16 * K / 4)), dim3(256), 0, st, M, K,
                       total_rs, X, (float4*)Af);
    D2P_LAUNCH_CHECK("pack_rows");
    return D2P_OK;
}

#define frag_off d2p_frag_off

// row sub-tile range of row tile rt when total_rs sub-tiles are spread over RT tiles
__device__ __forceinline__ void rt_range(int rt, int total_rs, int RT, int& rs0, int& nrs) {
    const int base = total_rs / RT, rem = total_rs % RT;
    rs0 = rt * base + min(rt, rem);
    nrs = base + (rt < rem ? 1 : 0);
}

// ---------------------------------------------------------------------------------------
// Forward step
// ---------------------------------------------------------------------------------------
struct StepFwdArgs {
    int M, U, total_rs, RT, t, has_h, skip_epi;
    const float4* hfrag_in;    // A operand: state h before this step (fragment-major)
    const float4* Wf;          // packed Wh (forward layout)
    float* z; long zrs;        // in: x·Wx + b ; out: full pre-activations (row stride zrs)
    const float* c_prev;       // [M,U] or null (zeros)
    const float* hs_prev;      // [M,U] state h (row-major), only read for masked rows; may be null
    const int* lens;           // [M] or null
    float* c_out;              // [M,U]
    float* hout;               // [M,U] emitted output (0 for masked rows)
    float* hs_out;             // [M,U] state h (row-major) or null when lens == null
    float* hfrag_out;          // state h after this step, fragment-major
};

// K-slice GEMM of one wave: NRS row sub-tiles x 2 column sub-tiles, all loads issued up
// front, K-partials written to LDS.
template <int CPW, int NRS>
__device__ __forceinline__ void fwd_gemm(const f32x4* __restrict__ Af, const f32x4* __restrict__ Bf,
                                         int rs0, int ct, int wave, int lane,
                                         float (*P)[FWD_RSMAX * 16][P_LD]) {
    constexpr int KC = 4 * CPW;
    f32x4 acc[NRS][2];
#pragma unroll
    for (int i = 0; i < NRS; ++i) {
        acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 av[CPW][NRS], bv[CPW][2];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int kc = wave * CPW + c;
#pragma unroll
        for (int s = 0; s < 2; ++s) bv[c][s] = Bf[(((long)ct * KC + kc) * 2 + s) * 64 + lane];
#pragma unroll
        for (int i = 0; i < NRS; ++i) av[c][i] = Af[((long)(rs0 + i) * KC + kc) * 64 + lane];
    }
    // keep every load above the MFMA chain: hipcc's occupancy-driven scheduler otherwise sinks
    // each load next to its first use (58 VGPRs, one load in flight at a time)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < NRS; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i][jj], bv[c][0][jj], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][i][jj], bv[c][1][jj], acc[i][1], 0, 0, 0);
            }
    // C/D layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int i = 0; i < NRS; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                P[wave][i * 16 + (lane >> 4) * 4 + r][s * 16 + (lane & 15)] = acc[i][s][r];
}

// Register-lean variant: two stages of FWD_CB chunks in flight (<= 256 registers, so two
// workgroups -- e.g. of two independent LSTMs on different streams -- can share a CU).
#define FWD_CB 2
template <int CPW, int NRS>
__device__ __forceinline__ void fwd_gemm_pipe(const f32x4* __restrict__ Af, const f32x4* __restrict__ Bf,
                                              int rs0, int ct, int wave, int lane,
                                              float (*P)[FWD_RSMAX * 16][P_LD]) {
    constexpr int KC = 4 * CPW;
    constexpr int CB = CPW < FWD_CB ? CPW : FWD_CB;
    constexpr int NB = CPW / CB;
    f32x4 acc[NRS][2];
#pragma unroll
    for (int i = 0; i < NRS; ++i) {
        acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 a0[CB][NRS], b0[CB][2], a1[CB][NRS], b1[CB][2];
#define FWD_LOAD(AV, BV, nb)                                                                  \
    {                                                                                         \
        _Pragma("unroll") for (int c = 0; c < CB; ++c) {                                      \
            const int kc = wave * CPW + (nb) * CB + c;                                        \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                     \
                BV[c][s] = Bf[(((long)ct * KC + kc) * 2 + s) * 64 + lane];                    \
            _Pragma("unroll") for (int i = 0; i < NRS; ++i)                                   \
                AV[c][i] = Af[((long)(rs0 + i) * KC + kc) * 64 + lane];                       \
        }                                                                                     \
    }
#define FWD_COMPUTE(AV, BV)                                                                   \
    {                                                                                         \
        _Pragma("unroll") for (int c = 0; c < CB; ++c)                                        \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                      \
        _Pragma("unroll") for (int i = 0; i < NRS; ++i) {                                     \
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[c][i][jj], BV[c][0][jj], acc[i][0], 0, 0, 0); \
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[c][i][jj], BV[c][1][jj], acc[i][1], 0, 0, 0); \
        }                                                                                     \
    }
    FWD_LOAD(a0, b0, 0)
#pragma unroll
    for (int nb = 0; nb < NB; nb += 2) {
        if (nb + 1 < NB) {
            FWD_LOAD(a1, b1, nb + 1)
            __builtin_amdgcn_sched_barrier(0);
        }
        FWD_COMPUTE(a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        if (nb + 1 < NB) {
            if (nb + 2 < NB) {
                FWD_LOAD(a0, b0, nb + 2)
                __builtin_amdgcn_sched_barrier(0);
            }
            FWD_COMPUTE(a1, b1)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef FWD_LOAD
#undef FWD_COMPUTE
#pragma unroll
    for (int i = 0; i < NRS; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                P[wave][i * 16 + (lane >> 4) * 4 + r][s * 16 + (lane & 15)] = acc[i][s][r];
}

// Up to D2P_MAX_SEQ independent LSTMs (same U) advance one step in ONE launch ("horizontal
// fusion": e.g. the action, perception and program decoders).  Workgroups [0, nb[0]) serve
// sequence 0, the next nb[1] sequence 1, ...  Fewer launches, and workgroups of different
// LSTMs share a CU (<= 256 registers each), overlapping each other's memory and MFMA phases
// without any extra operand traffic.
#define D2P_MAX_SEQ 3
struct StepFwdMulti {
    StepFwdArgs s[D2P_MAX_SEQ];
    int nb[D2P_MAX_SEQ];
};

template <int CPW, bool PIPE>   // 16-wide k chunks per wave; U = 64*CPW
__global__ void __launch_bounds__(256, PIPE ? 2 : 1) lstm_step_fwd_kernel(StepFwdMulti mm) {
    constexpr int KC = 4 * CPW;
    __shared__ __attribute__((aligned(16))) float P[4][FWD_RSMAX * 16][P_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x, si = 0;
    if (bid >= mm.nb[0]) { bid -= mm.nb[0]; si = 1; if (bid >= mm.nb[1]) { bid -= mm.nb[1]; si = 2; } }
    const StepFwdArgs& a = mm.s[si];
    const int U = a.U;
    // block -> (ct, rt): all row tiles of a column tile on one XCD (block b runs on XCD b % 8;
    // every nb[] is a multiple of 8, so bid % 8 == blockIdx.x % 8)
    const int nct = U >> 3;
    int ct, rt;
    {
        const int b = bid;
        if ((nct & 7) == 0) {
            const int per_xcd = nct >> 3, xcd = b & 7, slot = b >> 3;
            ct = xcd * per_xcd + slot / a.RT;
            rt = slot % a.RT;
        } else {
            ct = b / a.RT;
            rt = b % a.RT;
        }
    }
    int rs0, nrs;
    rt_range(rt, a.total_rs, a.RT, rs0, nrs);

    // Epilogue operands of this thread's (row, 4 units) item are requested BEFORE the GEMM so
    // that their latency hides under the MFMA chain (one item per thread: <= 160 items).
    const int nrows = nrs * 16;
    const int item = tid;
    const bool has_item = item < nrows * 2 && (rs0 * 16 + (item >> 1)) < a.M && !a.skip_epi;
    const int r = item >> 1, q = item & 1;
    const int row = rs0 * 16 + r;
    const int u = ct * 8 + q * 4;
    const long o = (long)row * U + u;
    bool active = false;
    f4 cp = zero4(), zz[4], hprev = zero4();
#pragma unroll
    for (int g = 0; g < 4; ++g) zz[g] = zero4();
    if (has_item) {
        active = a.lens ? (a.t < a.lens[row]) : true;
        if (a.c_prev) cp = ldf4(a.c_prev + o);
        if (active) {
            const float* zr = a.z + (long)row * a.zrs + u;
#pragma unroll
            for (int g = 0; g < 4; ++g) zz[g] = ldf4(zr + (long)g * U);
        } else if (a.hs_prev) {
            hprev = ldf4(a.hs_prev + o);
        }
    }

    if (a.has_h) {
        const f32x4* Af = reinterpret_cast<const f32x4*>(a.hfrag_in);
        const f32x4* Bf = reinterpret_cast<const f32x4*>(a.Wf);
        // the row sub-tile count is a compile-time constant inside each case: runtime
        // predicates around the MFMAs force accumulator shuffles and serialise the chain
        if (PIPE) {
            switch (nrs) {
                case 1: fwd_gemm_pipe<CPW, 1>(Af, Bf, rs0, ct, wave, lane, P); break;
                case 2: fwd_gemm_pipe<CPW, 2>(Af, Bf, rs0, ct, wave, lane, P); break;
                case 3: fwd_gemm_pipe<CPW, 3>(Af, Bf, rs0, ct, wave, lane, P); break;
                case 4: fwd_gemm_pipe<CPW, 4>(Af, Bf, rs0, ct, wave, lane, P); break;
                default: fwd_gemm_pipe<CPW, 5>(Af, Bf, rs0, ct, wave, lane, P); break;
            }
        } else {
            switch (nrs) {
                case 1: fwd_gemm<CPW, 1>(Af, Bf, rs0, ct, wave, lane, P); break;
                case 2: fwd_gemm<CPW, 2>(Af, Bf, rs0, ct, wave, lane, P); break;
                case 3: fwd_gemm<CPW, 3>(Af, Bf, rs0, ct, wave, lane, P); break;
                case 4: fwd_gemm<CPW, 4>(Af, Bf, rs0, ct, wave, lane, P); break;
                default: fwd_gemm<CPW, 5>(Af, Bf, rs0, ct, wave, lane, P); break;
            }
        }
    }
    __syncthreads();

    if (has_item) {
        f4 hstate;
        if (active) {
            if (a.has_h) {
                float* zr = a.z + (long)row * a.zrs + u;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const f4 p = ldf4(&P[w][r][g * 8 + q * 4]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) zz[g].v[e] += p.v[e];
                    }
                    stf4(zr + (long)g * U, zz[g]);      // full pre-activation, kept for backward
                }
            }
            f4 cn, hn;
            lstm_gate_fwd4(zz[0], zz[1], zz[2], zz[3], cp, cn, hn);
            stf4(a.c_out + o, cn);
            stf4(a.hout + o, hn);
            hstate = hn;
        } else {
            stf4(a.c_out + o, cp);
            stf4(a.hout + o, zero4());
            hstate = hprev;
        }
        if (a.hs_out) stf4(a.hs_out + o, hstate);
        stf4(a.hfrag_out + frag_off(row, u, U >> 4), hstate);
    }
}

// ---------------------------------------------------------------------------------------
// Backward step:  G = dz[t+1]·Wh^T (skipped when has_gemm == 0), then
//   mode 0: gate backward of step t -> dz[t] (row-major + fragment-major), dC in place
//   mode 1: final call: dh0 = G + pass-through
// ---------------------------------------------------------------------------------------
struct StepBwdArgs {
    int M, U, total_rs, RT, t, n_steps, has_gemm, mode, skip_epi;
    const float4* dzfrag_in;   // A operand: dz[t+1], fragment-major over K = 4U
    const float4* Wb;          // packed Wh^T
    const float* z; long zrs;  // pre-activations of step t
    const float* c_prev;       // c before step t, or null
    const float* c;            // c after step t
    const float* dhout;        // gradient wrt emitted output of step t, or null
    const float* dh_final;     // [M,U] or null
    const int* lens;
    float* dC;                 // [M,U] in/out
    float* dz; long dzrs;      // out: dz[t] row-major
    float* dzfrag_out;         // out: dz[t] fragment-major
    float* dh0;                // mode 1 output
};

#define BWD_CB 4   // chunks per software-pipeline stage

// K-slice GEMM of one wave for NRS row sub-tiles x 1 column sub-tile; two register stages of
// BWD_CB chunks each are kept in flight (manual double buffering with static names).
template <int NRS>
__device__ __forceinline__ void bwd_gemm(const f32x4* __restrict__ Af, const f32x4* __restrict__ Bf,
                                         int rs0, int KC4, int cpw, int wave, int lane,
                                         float (*P)[BWD_RSMAX * 16][P_LD]) {
    f32x4 acc[NRS];
#pragma unroll
    for (int i = 0; i < NRS; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a0[BWD_CB][NRS], b0[BWD_CB], a1[BWD_CB][NRS], b1[BWD_CB];
    const int kbeg = wave * cpw;
    const int NB = cpw / BWD_CB;
#define BWD_LOAD(AV, BV, nb)                                                               \
    {                                                                                      \
        _Pragma("unroll") for (int c = 0; c < BWD_CB; ++c) {                               \
            const int kc = kbeg + (nb) * BWD_CB + c;                                       \
            BV[c] = Bf[(long)kc * 64 + lane];                                              \
            _Pragma("unroll") for (int i = 0; i < NRS; ++i)                                \
                AV[c][i] = Af[((long)(rs0 + i) * KC4 + kc) * 64 + lane];                   \
        }                                                                                  \
    }
#define BWD_COMPUTE(AV, BV)                                                                \
    {                                                                                      \
        _Pragma("unroll") for (int c = 0; c < BWD_CB; ++c)                                 \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                   \
        _Pragma("unroll") for (int i = 0; i < NRS; ++i)                                    \
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[c][i][jj], BV[c][jj], acc[i], 0, 0, 0); \
    }
    // Steady state has UNCONDITIONAL prefetches: a conditional load makes hipcc merge the
    // vmcnt state at the join conservatively (vmcnt(0) before every compute stage, i.e. no
    // overlap).  The tail is peeled instead.
    BWD_LOAD(a0, b0, 0)
    int nb = 0;
    for (; nb + 2 < NB; nb += 2) {
        BWD_LOAD(a1, b1, nb + 1)
        __builtin_amdgcn_sched_barrier(0);
        BWD_COMPUTE(a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        BWD_LOAD(a0, b0, nb + 2)
        __builtin_amdgcn_sched_barrier(0);
        BWD_COMPUTE(a1, b1)
        __builtin_amdgcn_sched_barrier(0);
    }
    if (nb + 1 < NB) {
        BWD_LOAD(a1, b1, nb + 1)
        __builtin_amdgcn_sched_barrier(0);
        BWD_COMPUTE(a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        BWD_COMPUTE(a1, b1)
    } else {
        BWD_COMPUTE(a0, b0)
    }
#undef BWD_LOAD
#undef BWD_COMPUTE
#pragma unroll
    for (int i = 0; i < NRS; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[wave][i * 16 + (lane >> 4) * 4 + r][lane & 15] = acc[i][r];
}

struct StepBwdMulti {
    StepBwdArgs s[D2P_MAX_SEQ];
    int nb[D2P_MAX_SEQ];
};

template <int DUMMY>
__global__ void __launch_bounds__(256, 2) lstm_step_bwd_kernel(StepBwdMulti mm) {
    __shared__ __attribute__((aligned(16))) float P[4][BWD_RSMAX * 16][P_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x, si = 0;
    if (bid >= mm.nb[0]) { bid -= mm.nb[0]; si = 1; if (bid >= mm.nb[1]) { bid -= mm.nb[1]; si = 2; } }
    const StepBwdArgs& a = mm.s[si];
    const int U = a.U;
    const int KC4 = U >> 2;                 // 16-wide chunks over K = 4U
    const int cpw = KC4 >> 2;               // chunks per wave (U % 64 == 0 -> multiple of 4)
    const int nnt = U >> 4;
    int nt, rt;
    {
        const int b = bid;
        if ((nnt & 7) == 0) {
            const int per_xcd = nnt >> 3, xcd = b & 7, slot = b >> 3;
            nt = xcd * per_xcd + slot / a.RT;
            rt = slot % a.RT;
        } else {
            nt = b / a.RT;
            rt = b % a.RT;
        }
    }
    int rs0, nrs;
    rt_range(rt, a.total_rs, a.RT, rs0, nrs);

    // One (row, 4 units) epilogue item per thread (<= 64 rows x 4 quads); its operands are
    // requested before the GEMM so their latency hides under the MFMA chain.
    const int nrows = nrs * 16;
    const int r = tid >> 2, q = tid & 3;
    const int row = rs0 * 16 + r;
    const bool has_item = tid < nrows * 4 && row < a.M && !a.skip_epi;
    const int u = nt * 16 + q * 4;
    const long o = (long)row * U + u;
    int len = a.n_steps;
    bool cur_active = false, next_active = false;
    f4 zi = zero4(), zj = zero4(), zf = zero4(), zo = zero4(), cp = zero4(), cc = zero4();
    f4 dhx = zero4(), dcv = zero4();           // dhx = dh_final pass-through + dhout[t]
    if (has_item) {
        if (a.lens) len = a.lens[row];
        next_active = (a.t + 1 < a.n_steps) && (a.t + 1 < len);
        cur_active = (a.mode == 0) && (a.t < len);
        if (!next_active && a.dh_final) dhx = ldf4(a.dh_final + o);
        if (cur_active) {
            const float* zr = a.z + (long)row * a.zrs + u;
            zi = ldf4(zr); zj = ldf4(zr + U); zf = ldf4(zr + 2L * U); zo = ldf4(zr + 3L * U);
            if (a.c_prev) cp = ldf4(a.c_prev + o);
            cc = ldf4(a.c + o);
            dcv = ldf4(a.dC + o);
            if (a.dhout) {
                const f4 e4 = ldf4(a.dhout + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) dhx.v[e] += e4.v[e];
            }
        }
    }

    if (a.has_gemm) {
        const f32x4* Af = reinterpret_cast<const f32x4*>(a.dzfrag_in);
        const f32x4* Bf = reinterpret_cast<const f32x4*>(a.Wb) + (long)nt * KC4 * 64;
        switch (nrs) {
            case 1: bwd_gemm<1>(Af, Bf, rs0, KC4, cpw, wave, lane, P); break;
            case 2: bwd_gemm<2>(Af, Bf, rs0, KC4, cpw, wave, lane, P); break;
            default: bwd_gemm<3>(Af, Bf, rs0, KC4, cpw, wave, lane, P); break;
        }
    }
    __syncthreads();

    if (has_item) {
        // dH_t = dz[t+1]·Wh^T (+ dh_final for rows that are not active at t+1) (+ dhout[t])
        f4 dH = dhx;
        if (a.has_gemm) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f4 p = ldf4(&P[w][r][q * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) dH.v[e] += p.v[e];
            }
        }
        if (a.mode == 1) {
            stf4(a.dh0 + o, dH);
        } else {
            float* dzr = a.dz + (long)row * a.dzrs + u;
            const int KCx = U >> 2;
            if (cur_active) {
                f4 g[4], dcn;
                lstm_gate_bwd4(zi, zj, zf, zo, cp, cc, dH, dcv, g[0], g[1], g[2], g[3], dcn);
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    stf4(dzr + (long)gg * U, g[gg]);
                    stf4(a.dzfrag_out + frag_off(row, gg * U + u, KCx), g[gg]);
                }
                stf4(a.dC + o, dcn);
            } else {
                const f4 z0 = zero4();
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    stf4(dzr + (long)gg * U, z0);
                    stf4(a.dzfrag_out + frag_off(row, gg * U + u, KCx), z0);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Host drivers (called from d2p_lstm_seq_fwd / d2p_lstm_seq_bwd in lstm.hip)
// ---------------------------------------------------------------------------------------
// Ablation knobs for tools/bench_lstm_step.py (results are WRONG when set): bit 0 skips the
// MFMA part, bit 1 skips the epilogue.  Runtime flags, so nothing is dead-code eliminated.
static int g_step_debug = 0;
extern "C" int d2p_lstm_debug_flags(int flags) {
    g_step_debug = flags & 3;
    return D2P_OK;
}

static int g_fwd_pipe = 1;    // 1: register-lean pipelined forward kernel (2 workgroups per CU)
static int g_fwd_wgs = 256, g_bwd_wgs = 256;   // workgroups per launch aimed at (256 CUs)
extern "C" int d2p_lstm_set_tiling(int fwd_wgs, int bwd_wgs, int fwd_pipelined) {
    if (fwd_wgs > 0) g_fwd_wgs = fwd_wgs;
    if (bwd_wgs > 0) g_bwd_wgs = bwd_wgs;
    if (fwd_pipelined >= 0) g_fwd_pipe = fwd_pipelined ? 1 : 0;
    return D2P_OK;
}

static inline int pick_rt(int total_rs, int col_tiles, int rsmax, int target_wgs) {
    int rt = (target_wgs + col_tiles / 2) / col_tiles;
    if (rt < 1) rt = 1;
    const int need = (total_rs + rsmax - 1) / rsmax;      // keep tiles within the register budget
    if (rt < need) rt = need;
    if (rt > total_rs) rt = total_rs;
    return rt;
}

bool d2p_lstm_fused_eligible(int M, int U) {
    return M > 0 && (U == 64 || U == 128 || U == 256 || U == 512);
}

size_t d2p_lstm_fused_ws_bytes(int M, int U) {
    const size_t Mp = (size_t)((M + 15) / 16) * 16;
    // max over forward (Wf + 2 hfrag + 2 hs) and backward (Wb + 2 dzfrag + dC)
    const size_t fwd = (size_t)4 * U * U + 2 * Mp * U + 2 * (size_t)M * U;
    const size_t bwd = (size_t)4 * U * U + 2 * Mp * 4 * U + (size_t)M * U;
    return (fwd > bwd ? fwd : bwd) * sizeof(float);
}

template <int CPW>
static void launch_fwd(const StepFwdMulti& m, int blocks, hipStream_t st) {
    if (g_fwd_pipe)
        hipLaunchKernelGGL((lstm_step_fwd_kernel<CPW, true>), dim3(blocks), dim3(256), 0, st, m);
    else
        hipLaunchKernelGGL((lstm_step_fwd_kernel<CPW, false>), dim3(blocks), dim3(256), 0, st, m);
}

// ---- forward ---------------------------------------------------------------------------
struct FwdSeq {
    int M, U, n_steps, total_rs, RT, blocks;
    size_t MU;
    float* z; long zrs, zts;
    const float *Wh, *h0, *c0; const int* lens;
    float *hout, *cs, *h_final, *c_final;
    float *Wf, *hfrag[2], *hs[2];
    const float *h_prev, *c_prev;
};

static int fwd_prepare(FwdSeq& q, float* ws, hipStream_t st) {
    q.MU = (size_t)q.M * q.U;
    q.total_rs = (q.M + 15) / 16;
    const size_t Mp = (size_t)q.total_rs * 16;
    const int U = q.U;
    q.Wf = ws;
    q.hfrag[0] = q.Wf + (size_t)4 * U * U;
    q.hfrag[1] = q.hfrag[0] + Mp * U;
    q.hs[0] = q.hfrag[1] + Mp * U;
    q.hs[1] = q.hs[0] + q.MU;
    hipLaunchKernelGGL(pack_w_fwd_kernel, dim3(pack_blocks((long)U * U / 4)), dim3(256), 0, st, U, q.Wh,
                       (float4*)q.Wf);
    D2P_LAUNCH_CHECK("pack_w_fwd");
    if (q.h0) {
        hipLaunchKernelGGL(pack_rows_kernel, dim3(pack_blocks((long)Mp * U / 4)), dim3(256), 0, st, q.M, U,
                           q.total_rs, q.h0, (float4*)q.hfrag[0]);
        D2P_LAUNCH_CHECK("pack_rows(h0)");
    }
    const int nct = U / 8;
    q.RT = pick_rt(q.total_rs, nct, FWD_RSMAX, g_fwd_wgs);
    q.blocks = nct * q.RT;
    q.h_prev = q.h0;
    q.c_prev = q.c0;
    return D2P_OK;
}

static StepFwdArgs fwd_args(FwdSeq& q, int t) {
    StepFwdArgs a;
    a.M = q.M; a.U = q.U; a.total_rs = q.total_rs; a.RT = q.RT; a.t = t;
    a.has_h = (t > 0 || q.h0) ? 1 : 0;
    if (g_step_debug & 1) a.has_h = 0;
    a.skip_epi = (g_step_debug & 2) ? 1 : 0;
    a.hfrag_in = (const float4*)q.hfrag[t & 1];
    a.Wf = (const float4*)q.Wf;
    a.z = q.z + (long)t * q.zts; a.zrs = q.zrs;
    a.c_prev = q.c_prev;
    a.hs_prev = q.lens ? q.h_prev : nullptr;
    a.lens = q.lens;
    a.c_out = q.cs + t * q.MU;
    a.hout = q.hout + t * q.MU;
    a.hs_out = q.lens ? q.hs[t & 1] : nullptr;
    a.hfrag_out = q.hfrag[(t + 1) & 1];
    q.h_prev = q.lens ? q.hs[t & 1] : q.hout + t * q.MU;
    q.c_prev = q.cs + t * q.MU;
    return a;
}

static int fwd_finish(FwdSeq& q, hipStream_t st) {
    if (q.h_final) {
        if (q.h_prev) D2P_HIP(hipMemcpyAsync(q.h_final, q.h_prev, q.MU * sizeof(float), hipMemcpyDeviceToDevice, st));
        else D2P_HIP(hipMemsetAsync(q.h_final, 0, q.MU * sizeof(float), st));
    }
    if (q.c_final) {
        if (q.c_prev) D2P_HIP(hipMemcpyAsync(q.c_final, q.c_prev, q.MU * sizeof(float), hipMemcpyDeviceToDevice, st));
        else D2P_HIP(hipMemsetAsync(q.c_final, 0, q.MU * sizeof(float), st));
    }
    return D2P_OK;
}

// All sequences must share U.  Launch j serves step j of every sequence that still has one.
int d2p_lstm_fused_fwd_multi(int nseq, FwdSeq* seqs, float** ws, hipStream_t st) {
    int max_n = 0;
    for (int i = 0; i < nseq; ++i) {
        int rc = fwd_prepare(seqs[i], ws[i], st);
        if (rc) return rc;
        if (seqs[i].n_steps > max_n) max_n = seqs[i].n_steps;
    }
    const int U = seqs[0].U;
    for (int t = 0; t < max_n; ++t) {
        StepFwdMulti m;
        int blocks = 0, slot = 0;
        double work = 0.0;
        for (int i = 0; i < D2P_MAX_SEQ; ++i) m.nb[i] = 0;
        for (int i = 0; i < nseq; ++i) {
            if (t >= seqs[i].n_steps) continue;
            m.s[slot] = fwd_args(seqs[i], t);
            m.nb[slot] = seqs[i].blocks;
            blocks += seqs[i].blocks;
            work += 2.0 * seqs[i].M * 4.0 * U * U;
            ++slot;
        }
        for (int i = slot; i < D2P_MAX_SEQ; ++i) m.s[i] = m.s[0];
        {
            D2pProfScope prof(st, D2P_PROF_LSTM_STEP_FWD, work);
            switch (U) {
                case 64: launch_fwd<1>(m, blocks, st); break;
                case 128: launch_fwd<2>(m, blocks, st); break;
                case 256: launch_fwd<4>(m, blocks, st); break;
                default: launch_fwd<8>(m, blocks, st); break;
            }
        }
        D2P_LAUNCH_CHECK("lstm_step_fwd");
    }
    for (int i = 0; i < nseq; ++i) {
        int rc = fwd_finish(seqs[i], st);
        if (rc) return rc;
    }
    return D2P_OK;
}

int d2p_lstm_fused_fwd(int M, int U, int n_steps, float* z, long zrs, long zts, const float* Wh,
                       const float* h0, const float* c0, const int* lens, float* hout, float* cs,
                       float* h_final, float* c_final, float* ws, hipStream_t st) {
    FwdSeq q;
    q.M = M; q.U = U; q.n_steps = n_steps; q.z = z; q.zrs = zrs; q.zts = zts; q.Wh = Wh; q.h0 = h0;
    q.c0 = c0; q.lens = lens; q.hout = hout; q.cs = cs; q.h_final = h_final; q.c_final = c_final;
    return d2p_lstm_fused_fwd_multi(1, &q, &ws, st);
}

// ---- backward --------------------------------------------------------------------------
struct BwdSeq {
    int M, U, n_steps, total_rs, RT, blocks;
    size_t MU;
    const float* z; long zrs, zts;
    const float *Wh, *c0; const int* lens; const float *cs, *dhout, *dh_final, *dc_final;
    float *dz, *dh0, *dc0;
    float *Wb, *dzfrag[2], *dC;
};

static int bwd_prepare(BwdSeq& q, float* ws, hipStream_t st) {
    q.MU = (size_t)q.M * q.U;
    q.total_rs = (q.M + 15) / 16;
    const size_t Mp = (size_t)q.total_rs * 16;
    const int U = q.U;
    q.Wb = ws;
    q.dzfrag[0] = q.Wb + (size_t)4 * U * U;
    q.dzfrag[1] = q.dzfrag[0] + Mp * 4 * U;
    q.dC = q.dzfrag[1] + Mp * 4 * U;
    hipLaunchKernelGGL(pack_w_bwd_kernel, dim3(pack_blocks((long)U * U / 4)), dim3(256), 0, st, U, q.Wh,
                       (float4*)q.Wb);
    D2P_LAUNCH_CHECK("pack_w_bwd");
    if (q.dc_final) D2P_HIP(hipMemcpyAsync(q.dC, q.dc_final, q.MU * sizeof(float), hipMemcpyDeviceToDevice, st));
    else D2P_HIP(hipMemsetAsync(q.dC, 0, q.MU * sizeof(float), st));
    const int nnt = U / 16;
    q.RT = pick_rt(q.total_rs, nnt, BWD_RSMAX, g_bwd_wgs);
    q.blocks = nnt * q.RT;
    return D2P_OK;
}

static StepBwdArgs bwd_args(const BwdSeq& q, int t) {
    StepBwdArgs a;
    a.M = q.M; a.U = q.U; a.total_rs = q.total_rs; a.RT = q.RT; a.t = t; a.n_steps = q.n_steps;
    a.has_gemm = (t + 1 < q.n_steps) ? 1 : 0;
    if (g_step_debug & 1) a.has_gemm = 0;
    a.skip_epi = (g_step_debug & 2) ? 1 : 0;
    a.mode = t < 0 ? 1 : 0;
    a.dzfrag_in = (const float4*)q.dzfrag[(t + 1) & 1];
    a.Wb = (const float4*)q.Wb;
    a.z = t >= 0 ? q.z + (long)t * q.zts : nullptr; a.zrs = q.zrs;
    a.c_prev = t > 0 ? q.cs + (size_t)(t - 1) * q.MU : q.c0;
    a.c = t >= 0 ? q.cs + (size_t)t * q.MU : nullptr;
    a.dhout = (q.dhout && t >= 0) ? q.dhout + (size_t)t * q.MU : nullptr;
    a.dh_final = q.dh_final;
    a.lens = q.lens;
    a.dC = q.dC;
    a.dz = t >= 0 ? q.dz + (long)t * q.zts : nullptr; a.dzrs = q.zrs;
    a.dzfrag_out = q.dzfrag[t & 1];
    a.dh0 = q.dh0;
    return a;
}

// Launch j serves step t_i = n_i - 1 - j of every sequence (down to the final dh0 pass t = -1).
int d2p_lstm_fused_bwd_multi(int nseq, BwdSeq* seqs, float** ws, hipStream_t st) {
    int max_launches = 0;
    for (int i = 0; i < nseq; ++i) {
        int rc = bwd_prepare(seqs[i], ws[i], st);
        if (rc) return rc;
        const int n = seqs[i].n_steps + (seqs[i].dh0 ? 1 : 0);
        if (n > max_launches) max_launches = n;
    }
    const int U = seqs[0].U;
    for (int j = 0; j < max_launches; ++j) {
        StepBwdMulti m;
        int blocks = 0, slot = 0;
        double work = 0.0;
        for (int i = 0; i < D2P_MAX_SEQ; ++i) m.nb[i] = 0;
        for (int i = 0; i < nseq; ++i) {
            const int t = seqs[i].n_steps - 1 - j;
            if (t < -1 || (t < 0 && !seqs[i].dh0)) continue;
            m.s[slot] = bwd_args(seqs[i], t);
            m.nb[slot] = seqs[i].blocks;
            blocks += seqs[i].blocks;
            if (m.s[slot].has_gemm) work += 2.0 * seqs[i].M * 4.0 * U * U;
            ++slot;
        }
        if (slot == 0) break;
        for (int i = slot; i < D2P_MAX_SEQ; ++i) m.s[i] = m.s[0];
        {
            D2pProfScope prof(st, D2P_PROF_LSTM_STEP_BWD, work);
            hipLaunchKernelGGL((lstm_step_bwd_kernel<0>), dim3(blocks), dim3(256), 0, st, m);
        }
        D2P_LAUNCH_CHECK("lstm_step_bwd");
    }
    for (int i = 0; i < nseq; ++i)
        if (seqs[i].dc0)
            D2P_HIP(hipMemcpyAsync(seqs[i].dc0, seqs[i].dC, seqs[i].MU * sizeof(float), hipMemcpyDeviceToDevice, st));
    return D2P_OK;
}

int d2p_lstm_fused_bwd(int M, int U, int n_steps, const float* z, long zrs, long zts, const float* Wh,
                       const float* c0, const int* lens, const float* cs, const float* dhout,
                       const float* dh_final, const float* dc_final, float* dz, float* dh0,
                       float* dc0, float* ws, hipStream_t st) {
    BwdSeq q;
    q.M = M; q.U = U; q.n_steps = n_steps; q.z = z; q.zrs = zrs; q.zts = zts; q.Wh = Wh; q.c0 = c0;
    q.lens = lens; q.cs = cs; q.dhout = dhout; q.dh_final = dh_final; q.dc_final = dc_final;
    q.dz = dz; q.dh0 = dh0; q.dc0 = dc0;
    return d2p_lstm_fused_bwd_multi(1, &q, &ws, st);
}

// ---- C ABI: several independent sequences per launch (include/d2p.h) ----------------------
static bool multi_fused_ok(int nseq, const int* Ms, const int* Us, const size_t* wsb, const void* const* wsp) {
    extern int d2p_lstm_is_fused_enabled();
    if (!d2p_lstm_is_fused_enabled() || nseq < 1 || nseq > D2P_MAX_SEQ) return false;
    for (int i = 0; i < nseq; ++i)
        if (!d2p_lstm_fused_eligible(Ms[i], Us[i]) || Us[i] != Us[0] || !wsp[i] ||
            wsb[i] < d2p_lstm_fused_ws_bytes(Ms[i], Us[i]))
            return false;
    return true;
}

extern "C" int d2p_lstm_seq_fwd_multi(int nseq, const d2p_lstm_fwd_desc* d, d2p_stream_t stream) {
    D2P_REQUIRE(nseq >= 1 && d, D2P_EINVAL, "lstm fwd multi: bad arguments");
    int Ms[D2P_MAX_SEQ], Us[D2P_MAX_SEQ];
    size_t wsb[D2P_MAX_SEQ];
    const void* wsp[D2P_MAX_SEQ];
    bool ok = nseq <= D2P_MAX_SEQ;
    for (int i = 0; ok && i < nseq; ++i) {
        Ms[i] = d[i].M; Us[i] = d[i].U; wsb[i] = d[i].ws_bytes; wsp[i] = d[i].ws;
        ok = d[i].n_steps > 0 && d[i].z_row_stride % 4 == 0 && (((uintptr_t)d[i].z & 15) == 0);
    }
    // with the persistent back end on, every sequence gets its own persistent launch (the sequence
    // entry point falls back to per-step launches for shapes it does not take)
    if (ok && !d2p_lstm_is_persistent_enabled() && multi_fused_ok(nseq, Ms, Us, wsb, wsp)) {
        FwdSeq q[D2P_MAX_SEQ];
        float* ws[D2P_MAX_SEQ];
        for (int i = 0; i < nseq; ++i) {
            q[i].M = d[i].M; q[i].U = d[i].U; q[i].n_steps = d[i].n_steps; q[i].z = d[i].z;
            q[i].zrs = d[i].z_row_stride; q[i].zts = d[i].z_t_stride; q[i].Wh = d[i].Wh;
            q[i].h0 = d[i].h0; q[i].c0 = d[i].c0; q[i].lens = d[i].lens; q[i].hout = d[i].hout;
            q[i].cs = d[i].cs; q[i].h_final = d[i].h_final; q[i].c_final = d[i].c_final;
            ws[i] = (float*)d[i].ws;
        }
        return d2p_lstm_fused_fwd_multi(nseq, q, ws, as_stream(stream));
    }
    if (nseq <= 3 && d2p_lstm_is_persistent_enabled()) {     // all of them in one launch of the wide-tile kernel
        int rc = D2P_OK;
        if (d2p_lstm_try_wide_fwd(nseq, d, as_stream(stream), &rc)) return rc;
    }
    if (nseq == 2 && d2p_lstm_is_persistent_enabled()) {     // two sequences sharing one persistent launch
        int rc = D2P_OK;
        if (d2p_lstm_try_pair_fwd(d, as_stream(stream), &rc)) return rc;
    }
    for (int i = 0; i < nseq; ++i) {     // generic path: one sequence after the other
        int rc = d2p_lstm_seq_fwd_desc(d + i, stream);
        if (rc) return rc;
    }
    return D2P_OK;
}

extern "C" int d2p_lstm_seq_bwd_multi(int nseq, const d2p_lstm_bwd_desc* d, d2p_stream_t stream) {
    D2P_REQUIRE(nseq >= 1 && d, D2P_EINVAL, "lstm bwd multi: bad arguments");
    int Ms[D2P_MAX_SEQ], Us[D2P_MAX_SEQ];
    size_t wsb[D2P_MAX_SEQ];
    const void* wsp[D2P_MAX_SEQ];
    bool ok = nseq <= D2P_MAX_SEQ;
    for (int i = 0; ok && i < nseq; ++i) {
        Ms[i] = d[i].M; Us[i] = d[i].U; wsb[i] = d[i].ws_bytes; wsp[i] = d[i].ws;
        ok = d[i].n_steps > 0 && d[i].z_row_stride % 4 == 0 && (((uintptr_t)d[i].z & 15) == 0) &&
             (((uintptr_t)d[i].dz & 15) == 0);
    }
    if (ok && !d2p_lstm_is_persistent_enabled() && multi_fused_ok(nseq, Ms, Us, wsb, wsp)) {
        BwdSeq q[D2P_MAX_SEQ];
        float* ws[D2P_MAX_SEQ];
        for (int i = 0; i < nseq; ++i) {
            q[i].M = d[i].M; q[i].U = d[i].U; q[i].n_steps = d[i].n_steps; q[i].z = d[i].z;
            q[i].zrs = d[i].z_row_stride; q[i].zts = d[i].z_t_stride; q[i].Wh = d[i].Wh; q[i].c0 = d[i].c0;
            q[i].lens = d[i].lens; q[i].cs = d[i].cs; q[i].dhout = d[i].dhout;
            q[i].dh_final = d[i].dh_final; q[i].dc_final = d[i].dc_final; q[i].dz = d[i].dz;
            q[i].dh0 = d[i].dh0; q[i].dc0 = d[i].dc0;
            ws[i] = (float*)d[i].ws;
        }
        int rc = d2p_lstm_fused_bwd_multi(nseq, q, ws, as_stream(stream));
        for (int i = 0; !rc && i < nseq; ++i) rc = d2p_lstm_db_colsum(d + i, stream);
        return rc;
    }
    if (nseq == 2 && d2p_lstm_is_persistent_enabled()) {
        int rc = D2P_OK;
        if (d2p_lstm_try_pair_bwd(d, as_stream(stream), &rc)) return rc;
    }
    if (nseq == 3 && d2p_lstm_is_persistent_enabled()) {      // all three in one launch, else the first alone + a pair
        int rc = D2P_OK;
        if (d2p_lstm_try_triple_bwd(d, as_stream(stream), &rc)) return rc;
        rc = d2p_lstm_seq_bwd_desc(d, stream);
        if (rc) return rc;
        if (d2p_lstm_try_pair_bwd(d + 1, as_stream(stream), &rc)) return rc;
        for (int i = 1; i < 3; ++i) {
            rc = d2p_lstm_seq_bwd_desc(d + i, stream);
            if (rc) return rc;
        }
        return D2P_OK;
    }
    for (int i = 0; i < nseq; ++i) {
        int rc = d2p_lstm_seq_bwd_desc(d + i, stream);
        if (rc) return rc;
    }
    return D2P_OK;
}
