// Persistent LSTM sequence kernels: ONE launch runs every time step of a recurrence.
// Replaces the per-step loop of tf.nn.dynamic_rnn(BasicLSTMCell) / BasicDecoder+TrainingHelper
// (models/model_full.py:243-258,260-277,465-471) -- same arithmetic as lstm_step.hip, which
// launches one kernel per step and pays, per step, a kernel boundary (~1.7 us), a cold re-fetch of
// the 4 MB recurrent weight and a ~224 KB operand load per workgroup before 4.3 us of MFMA work.
//
// Structure (MI355X: 256 CUs, 8 XCDs with private L2s, no grid barrier cheaper than 4 us):
//   * rows (sequences) never mix inside a recurrence, so the M rows are cut into RT independent
//     DOMAINS; only the workgroups of one domain exchange data, and there is no grid-wide barrier.
//     A workgroup owns a [domain's rows] x [8 units x 4 gates] (forward) / [16 units] (backward)
//     output tile for ALL steps; its slice of the recurrent weight lives in VGPRs for the whole
//     sequence (64 / 128 registers per wave at U = 512) and is read from memory once.
//   * a domain's rows are processed one 16-row sub-tile ("phase") at a time.  While a workgroup
//     computes phase p of step t, the other phases' results of step t / t-1 travel between the
//     workgroups of the domain, so the hand-off latency (write-through store -> flag -> poll ->
//     operand load, ~2.5 us) hides behind the MFMA work of the other phases (5 phases at M = 320).
//   * wave roles: waves 0-3 = MFMA waves (K split four ways, partial tiles combined through LDS);
//     wave 4 = epilogue wave (gate math, state, all global stores, publishes the new h / dz rows).
//     The MFMA waves never wait for a store or for the epilogue; they prefetch the next phase's
//     operands into a second register set behind a flag poll whose load was issued half a phase
//     earlier.
//   * hand-off protocol (cdna_hip_programming.md Guideline 16, form R1): payload = 16-byte
//     write-through (sc1) stores by ONE wave -> s_waitcnt vmcnt(0) -> one relaxed agent-scope flag
//     store per (domain, phase, producer); consumers poll the flags of exactly the producers their
//     K slice needs with relaxed agent-scope loads and read the payload with sc1 loads (L1 bypass).
//     Correct under any workgroup -> CU/XCD placement; the block -> tile map only tries to keep a
//     domain on few XCDs.  Every spin is bounded: on timeout an error word is set, polling stops
//     everywhere and the kernel runs to completion (results invalid, d2p_lstm_persist_error() != 0).
//   * needs all workgroups co-resident: grid <= number of CUs, one workgroup per CU.
#include "common.h"
#include "lstm_internal.h"
#include "lstm_math.h"
#include "prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define PS_NRS_MAX 8        // 16-row phases per domain (M <= RT * 128)
#define PS_PLD 36           // LDS partial-tile row stride (floats)
#define PS_THREADS 320      // 4 MFMA waves + 1 epilogue wave
#define PS_SPIN_LIMIT 60000 // ~50 ms of polling before giving up
#define PS_AUX_SC1 16       // buffer-instruction cache policy: sc1 (agent scope, bypasses the CU's L1)

// ---- error word ---------------------------------------------------------------------------
__device__ unsigned g_ps_err;          // sticky: 0 ok, else (code << 24) | block
static unsigned* ps_err_ptr() {
    static unsigned* p = nullptr;
    if (!p) (void)hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_ps_err));
    return p;
}
extern "C" int d2p_lstm_persist_error(int reset) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ps_err), sizeof(v)) != hipSuccess) return -1;
    if (reset && v) {
        const unsigned z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ps_err), &z, sizeof(z));
    }
    return (int)v;
}

static int g_persist = 1;
extern "C" int d2p_lstm_set_persistent(int on) {
    g_persist = on ? 1 : 0;
    return D2P_OK;
}
int d2p_lstm_is_persistent_enabled() { return g_persist; }

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ unsigned ps_ld_flag(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ps_st_flag(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ps_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ps_ld_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, PS_AUX_SC1);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void ps_st_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, float a, float b, float c, float d) {
    const f32x4 v = {a, b, c, d};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, byte_off, 0, PS_AUX_SC1);
}
__device__ __forceinline__ void ps_barrier() {
    // LDS writes of this wave done, then the workgroup barrier; deliberately NOT __syncthreads():
    // its release fence would also drain the operand prefetch (vmcnt) that must stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Wave-uniform: wait until every polled flag is >= need.  `fv` is the value of an earlier
// (asynchronous) read of this lane's flag.  Gives up after PS_SPIN_LIMIT polls or as soon as any
// workgroup reported an error.
__device__ __forceinline__ void ps_wait_flags(const unsigned* f, unsigned need, unsigned fv, unsigned* err,
                                              unsigned code) {
    if (__all((int)(fv >= need))) return;
    unsigned spins = 0;
    for (;;) {
        __builtin_amdgcn_s_sleep(4);
        fv = ps_ld_flag(f);
        if (__all((int)(fv >= need))) return;
        ++spins;
        if ((spins & 63u) == 0u) {
            if (ps_ld_flag(err) != 0u) return;
            if (spins > PS_SPIN_LIMIT) {
                if ((threadIdx.x & 63) == 0) ps_st_flag(err, (code << 24) | (blockIdx.x & 0xffffffu) | 0x800000u);
                return;
            }
        }
    }
}

// block -> (row domain rt, column tile ct): consecutive tiles of a domain on the same XCD
// (block b runs on XCD b % 8 -- a speed hint only, nothing depends on it)
__device__ __forceinline__ void ps_block_tile(int ncol, int& rt, int& ct) {
    const int g = gridDim.x, b = blockIdx.x;
    int L = b;
    if ((g & 7) == 0) L = (b & 7) * (g >> 3) + (b >> 3);
    rt = L / ncol;
    ct = L - rt * ncol;
}
__device__ __forceinline__ void ps_rt_range(int rt, int total_rs, int RT, int& rs0, int& nrs) {
    const int base = total_rs / RT, rem = total_rs % RT;
    rs0 = rt * base + min(rt, rem);
    nrs = base + (rt < rem ? 1 : 0);
}

// =============================================================================================
// Forward
// =============================================================================================
struct PsFwdArgs {
    int M, U, T, total_rs, RT, has_h0;
    const float4* Wf;       // packed Wh (lstm_step.hip forward layout)
    float* hfrag;           // 2 ping-pong buffers of Mp*U floats, fragment-major; [0] = state before step 0
    unsigned hfrag_bytes;   // bytes of ONE buffer
    float* z; long zrs, zts;
    const float* h0; const float* c0; const int* lens;
    float* hout; float* cs; float* h_final; float* c_final;
    unsigned* flags;        // [RT][PS_NRS_MAX][U/8], zeroed before the launch
    unsigned* err;
};

template <int CPW>
__device__ __forceinline__ void ps_fwd_chain(const f32x4 (&av)[CPW], const f32x4 (&bv)[CPW][2], int c0, int c1,
                                             f32x4& acc0, f32x4& acc1) {
#pragma unroll
    for (int c = c0; c < c1; ++c)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][jj], bv[c][0][jj], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][jj], bv[c][1][jj], acc1, 0, 0, 0);
        }
}

// One phase of one MFMA wave: K-slice product of the 16 rows in `cur` with the resident weight,
// prefetch of the next phase's rows into `nxt` behind the flag poll.
template <int CPW>
__device__ __forceinline__ void ps_fwd_tick(const f32x4 (&cur)[CPW], f32x4 (&nxt)[CPW], const f32x4 (&bv)[CPW][2],
                                            bool do_gemm, const unsigned* fl, unsigned need,
                                            __amdgpu_buffer_rsrc_t hres, int next_off, float* Pw, int lane,
                                            unsigned* err) {
    constexpr int H = CPW / 2;
    // the flag read is unconditional (need == 0 when nothing has to be waited for): a conditional
    // load would be waited for at the join, i.e. before the MFMA chain instead of behind it
    const unsigned fv = ps_ld_flag(fl);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (do_gemm) ps_fwd_chain<CPW>(cur, bv, 0, H, acc0, acc1);
    ps_wait_flags(fl, need, fv, err, 1);
#pragma unroll
    for (int c = 0; c < CPW; ++c) nxt[c] = ps_ld_sc1(hres, next_off + c * 1024);
    __builtin_amdgcn_sched_barrier(0);
    if (do_gemm) ps_fwd_chain<CPW>(cur, bv, H, CPW, acc0, acc1);
    // C/D layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Pw[((lane >> 4) * 4 + r) * PS_PLD + (lane & 15)] = acc0[r];
        Pw[((lane >> 4) * 4 + r) * PS_PLD + 16 + (lane & 15)] = acc1[r];
    }
    ps_barrier();
}

template <int CPW>   // U = 64 * CPW
__global__ void __launch_bounds__(PS_THREADS) lstm_persist_fwd_kernel(PsFwdArgs a) {
    constexpr int KC = 4 * CPW;
    // ONE shared array (a second __shared__ object de-pipelines loads, cdna_hip_programming.md)
    __shared__ __attribute__((aligned(16))) float lds[2 * 4 * 16 * PS_PLD + PS_NRS_MAX * 64 * 5];
    float* P = lds;                                            // [2][4][16][PS_PLD]
    float* stc = lds + 2 * 4 * 16 * PS_PLD;                    // [NRS][64][2] cell state
    float* sth = stc + PS_NRS_MAX * 64 * 2;                    // [NRS][64][2] hidden state
    int* stl = reinterpret_cast<int*>(sth + PS_NRS_MAX * 64 * 2);   // [NRS][64] row length
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int U = a.U, nct = U >> 3;
    int rt, ct;
    ps_block_tile(nct, rt, ct);
    int rs0, nrs;
    ps_rt_range(rt, a.total_rs, a.RT, rs0, nrs);
    const int nticks = nrs * a.T;
    unsigned* fbase = a.flags + (long)rt * PS_NRS_MAX * nct;

    if (wave < 4) {
        // ---------------- MFMA waves ----------------
        f32x4 bv[CPW][2];
        const f32x4* Bf = reinterpret_cast<const f32x4*>(a.Wf);
#pragma unroll
        for (int c = 0; c < CPW; ++c)
#pragma unroll
            for (int s = 0; s < 2; ++s) bv[c][s] = Bf[(((long)ct * KC + wave * CPW + c) * 2 + s) * 64 + lane];
        const __amdgpu_buffer_rsrc_t hres = ps_rsrc(a.hfrag, 2u * a.hfrag_bytes);
        // byte offset of this lane's float4 in block (rs, kc = wave*CPW) of buffer `b`
        const int lane_off = (wave * CPW * 64 + lane) * 16;
        const unsigned* fl = fbase + 2 * CPW * wave + (lane & (2 * CPW - 1));
        f32x4 a0[CPW], a1[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) a0[c] = ps_ld_sc1(hres, rs0 * KC * 1024 + lane_off + c * 1024);
        int p = 0, t = 0;
        for (int n = 0; n < nticks; n += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half == 1 && n + 1 >= nticks) break;
                // next tick (clamped to this one at the very end: the load is unconditional)
                int p1 = p + 1, t1 = t;
                if (p1 == nrs) { p1 = 0; t1 = t + 1; }
                const bool last = (n + half + 1 >= nticks);
                if (last) { p1 = p; t1 = t; }
                const bool do_gemm = (t > 0) || a.has_h0;
                const unsigned need = last ? 0u : (unsigned)t1;      // version t1 = published after step t1-1
                const int next_off = (int)((t1 & 1) * a.hfrag_bytes) + (rs0 + p1) * KC * 1024 + lane_off;
                float* Pw = P + (((n + half) & 1) * 4 + wave) * 16 * PS_PLD;
                if (half == 0)
                    ps_fwd_tick<CPW>(a0, a1, bv, do_gemm, fl + p1 * nct, need, hres, next_off, Pw, lane, a.err);
                else
                    ps_fwd_tick<CPW>(a1, a0, bv, do_gemm, fl + p1 * nct, need, hres, next_off, Pw, lane, a.err);
                p = p1; t = t1;
            }
        }
    } else {
        // ---------------- epilogue wave ----------------
        const int r = lane >> 2, pr = lane & 3;
        const int u = ct * 8 + pr * 2;
        const __amdgpu_buffer_rsrc_t hres = ps_rsrc(a.hfrag, 2u * a.hfrag_bytes);
        for (int p = 0; p < nrs; ++p) {
            const int row = (rs0 + p) * 16 + r;
            const bool valid = row < a.M;
            float2 c = make_float2(0.f, 0.f), h = make_float2(0.f, 0.f);
            int len = 0x7fffffff;
            if (valid) {
                if (a.c0) c = *reinterpret_cast<const float2*>(a.c0 + (long)row * U + u);
                if (a.h0) h = *reinterpret_cast<const float2*>(a.h0 + (long)row * U + u);
                if (a.lens) len = a.lens[row];
            }
            *reinterpret_cast<float2*>(stc + (p * 64 + lane) * 2) = c;
            *reinterpret_cast<float2*>(sth + (p * 64 + lane) * 2) = h;
            stl[p * 64 + lane] = len;
        }
        int p = 0, t = 0;
        float2 zin[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) zin[g] = make_float2(0.f, 0.f);
        {   // pre-activation inputs of the first phase
            const int row = rs0 * 16 + r;
            if (row < a.M && 0 < stl[lane]) {
                const float* zr = a.z + (long)row * a.zrs + u;
#pragma unroll
                for (int g = 0; g < 4; ++g) zin[g] = *reinterpret_cast<const float2*>(zr + (long)g * U);
            }
        }
        for (int n = 0; n < nticks; ++n) {
            ps_barrier();          // partial tiles of tick n are in P[n & 1]
            const int row = (rs0 + p) * 16 + r;
            const bool valid = row < a.M;
            const int len = stl[p * 64 + lane];
            const bool active = t < len;
            const bool has_gemm = (t > 0) || a.has_h0;
            const long o = (long)row * U + u;
            float2 zz[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) zz[g] = zin[g];
            if (has_gemm) {
                const float* Pb = P + (n & 1) * 4 * 16 * PS_PLD + r * PS_PLD + pr * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float2 q = *reinterpret_cast<const float2*>(Pb + w * 16 * PS_PLD + g * 8);
                        zz[g].x += q.x;
                        zz[g].y += q.y;
                    }
            }
            const float2 cp = *reinterpret_cast<const float2*>(stc + (p * 64 + lane) * 2);
            const float2 hp = *reinterpret_cast<const float2*>(sth + (p * 64 + lane) * 2);
            float2 cn = cp, hs = hp, ho = make_float2(0.f, 0.f);
            if (active) {
                // c' = c*sigmoid(f+1) + sigmoid(i)*tanh(j);  h' = tanh(c')*sigmoid(o)   (lstm_math.h)
                cn.x = cp.x * d2p_sigmoid(zz[2].x + D2P_FORGET_BIAS) + d2p_sigmoid(zz[0].x) * d2p_tanh(zz[1].x);
                cn.y = cp.y * d2p_sigmoid(zz[2].y + D2P_FORGET_BIAS) + d2p_sigmoid(zz[0].y) * d2p_tanh(zz[1].y);
                ho.x = d2p_tanh(cn.x) * d2p_sigmoid(zz[3].x);
                ho.y = d2p_tanh(cn.y) * d2p_sigmoid(zz[3].y);
                hs = ho;
            }
            // publish first: the new state rows in fragment-major layout, 16 bytes per even lane
            const float px = __shfl_xor(hs.x, 1, 64), py = __shfl_xor(hs.y, 1, 64);
            if (valid && !(pr & 1)) {
                const int off = (int)(((t + 1) & 1) * a.hfrag_bytes) +
                                (int)(d2p_frag_off(row, ct * 8 + (pr >> 1) * 4, U >> 4) * 4);
                ps_st_sc1(hres, off, hs.x, hs.y, px, py);
            }
            if (valid) {
                if (active && has_gemm) {
                    float* zr = a.z + (long)t * a.zts + (long)row * a.zrs + u;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<float2*>(zr + (long)g * U) = zz[g];
                }
                *reinterpret_cast<float2*>(a.cs + (size_t)t * a.M * U + o) = cn;
                *reinterpret_cast<float2*>(a.hout + (size_t)t * a.M * U + o) = ho;
            }
            *reinterpret_cast<float2*>(stc + (p * 64 + lane) * 2) = cn;
            *reinterpret_cast<float2*>(sth + (p * 64 + lane) * 2) = hs;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) ps_st_flag(fbase + p * nct + ct, (unsigned)(t + 1));
            // next phase + its pre-activation inputs (land while this wave waits at the barrier)
            if (++p == nrs) { p = 0; ++t; }
            if (n + 1 < nticks) {
                const int row1 = (rs0 + p) * 16 + r;
                if (row1 < a.M && t < stl[p * 64 + lane]) {
                    const float* zr = a.z + (long)t * a.zts + (long)row1 * a.zrs + u;
#pragma unroll
                    for (int g = 0; g < 4; ++g) zin[g] = *reinterpret_cast<const float2*>(zr + (long)g * U);
                }
            }
        }
        for (int q = 0; q < nrs; ++q) {
            const int row = (rs0 + q) * 16 + r;
            if (row < a.M) {
                if (a.h_final) *reinterpret_cast<float2*>(a.h_final + (long)row * U + u) = *reinterpret_cast<const float2*>(sth + (q * 64 + lane) * 2);
                if (a.c_final) *reinterpret_cast<float2*>(a.c_final + (long)row * U + u) = *reinterpret_cast<const float2*>(stc + (q * 64 + lane) * 2);
            }
        }
    }
}

// =============================================================================================
// Backward:  dH_t = dz[t+1]·Wh^T, gate backward of step t -> dz[t]; last pass (t = -1): dh0
// =============================================================================================
struct PsBwdArgs {
    int M, U, T, total_rs, RT, want_dh0;
    const float4* Wb;       // packed Wh^T (lstm_step.hip backward layout)
    float* dzfrag;          // 2 ping-pong buffers of Mp*4U floats, fragment-major over K = 4U
    unsigned dzfrag_bytes;  // bytes of ONE buffer
    const float* z; long zrs, zts;
    const float* c0; const float* cs; const int* lens;
    const float* dhout; const float* dh_final; const float* dc_final;
    float* dz; float* dh0; float* dc0;
    unsigned* flags;        // [RT][PS_NRS_MAX][U/16]
    unsigned* err;
};

template <int CB, int CPWB>
__device__ __forceinline__ void ps_bwd_stage(const f32x4 (&sv)[CB], const f32x4 (&bw)[CPWB], int base, f32x4& acc0,
                                             f32x4& acc1) {
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (c & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[c][jj], bw[base + c][jj], acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[c][jj], bw[base + c][jj], acc0, 0, 0, 0);
        }
}

template <int CPW>   // U = 64 * CPW; each MFMA wave owns one gate's K range = 4*CPW chunks of 16
__global__ void __launch_bounds__(PS_THREADS) lstm_persist_bwd_kernel(PsBwdArgs a) {
    constexpr int CPWB = 4 * CPW;                // chunks per wave
    constexpr int NB = CPWB >= 32 ? 4 : 2;       // register stages per phase (even)
    constexpr int CB = CPWB / NB;                // chunks per stage
    constexpr int KC4 = 4 * CPWB;                // chunks over K = 4U
    __shared__ __attribute__((aligned(16))) float lds[2 * 4 * 16 * PS_PLD + PS_NRS_MAX * 64 * 5];
    float* P = lds;                                            // [2][4][16][PS_PLD] (16 columns used)
    float* stdc = lds + 2 * 4 * 16 * PS_PLD;                   // [NRS][64][4] dC state
    int* stl = reinterpret_cast<int*>(stdc + PS_NRS_MAX * 64 * 4);   // [NRS][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int U = a.U, nnt = U >> 4;
    int rt, nt;
    ps_block_tile(nnt, rt, nt);
    int rs0, nrs;
    ps_rt_range(rt, a.total_rs, a.RT, rs0, nrs);
    const int J = a.T + (a.want_dh0 ? 1 : 0);    // passes: t = T-1 .. 0 (, -1)
    const int nticks = nrs * J;
    unsigned* fbase = a.flags + (long)rt * PS_NRS_MAX * nnt;

    if (wave < 4) {
        // ---------------- MFMA waves ----------------
        f32x4 bw[CPWB];
        const f32x4* Bf = reinterpret_cast<const f32x4*>(a.Wb) + ((long)nt * KC4 + wave * CPWB) * 64;
#pragma unroll
        for (int c = 0; c < CPWB; ++c) bw[c] = Bf[(long)c * 64 + lane];
        const __amdgpu_buffer_rsrc_t dres = ps_rsrc(a.dzfrag, 2u * a.dzfrag_bytes);
        const int lane_off = (wave * CPWB * 64 + lane) * 16;
        const unsigned* fl = fbase + (lane & (nnt - 1));
        f32x4 s0[CB], s1[CB];
        // pass j = 0 has no product (there is no dz[T]); its loads are harmless reads of the buffer
        int p = 0, j = 0;
        {
            const int off0 = (int)(((a.T - j) & 1) * a.dzfrag_bytes) + (rs0 + p) * KC4 * 1024 + lane_off;
#pragma unroll
            for (int c = 0; c < CB; ++c) s0[c] = ps_ld_sc1(dres, off0 + c * 1024);
        }
        for (int n = 0; n < nticks; ++n) {
            int p1 = p + 1, j1 = j;
            if (p1 == nrs) { p1 = 0; j1 = j + 1; }
            const bool last = (n + 1 >= nticks);
            if (last) { p1 = p; j1 = j; }
            const bool do_gemm = j > 0;
            const unsigned need = last ? 0u : (unsigned)j1;
            // pass j consumes dz[t+1] with t = T-1-j, i.e. the buffer written in pass j-1: (T-j) & 1
            const int off = (int)(((a.T - j) & 1) * a.dzfrag_bytes) + (rs0 + p) * KC4 * 1024 + lane_off;
            const int off1 = (int)(((a.T - j1) & 1) * a.dzfrag_bytes) + (rs0 + p1) * KC4 * 1024 + lane_off;
            const unsigned* fl1 = fl + p1 * nnt;
            unsigned fv = 0u;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < NB; ++st) {
                if (st == NB - 2) fv = ps_ld_flag(fl1);        // unconditional, see ps_fwd_tick
                if (st == NB - 1) ps_wait_flags(fl1, need, fv, a.err, 2);
                const int noff = (st < NB - 1) ? off + (st + 1) * CB * 1024 : off1;
                if ((st & 1) == 0) {
#pragma unroll
                    for (int c = 0; c < CB; ++c) s1[c] = ps_ld_sc1(dres, noff + c * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                    if (do_gemm) ps_bwd_stage<CB, CPWB>(s0, bw, st * CB, acc0, acc1);
                } else {
#pragma unroll
                    for (int c = 0; c < CB; ++c) s0[c] = ps_ld_sc1(dres, noff + c * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                    if (do_gemm) ps_bwd_stage<CB, CPWB>(s1, bw, st * CB, acc0, acc1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float* Pw = P + ((n & 1) * 4 + wave) * 16 * PS_PLD;
#pragma unroll
            for (int r = 0; r < 4; ++r) Pw[((lane >> 4) * 4 + r) * PS_PLD + (lane & 15)] = acc0[r] + acc1[r];
            ps_barrier();
            p = p1; j = j1;
        }
    } else {
        // ---------------- epilogue wave: one (row, 4 units) item per lane ----------------
        const int r = lane >> 2, q = lane & 3;
        const int u = nt * 16 + q * 4;
        const int KCx = U >> 2;
        const __amdgpu_buffer_rsrc_t dres = ps_rsrc(a.dzfrag, 2u * a.dzfrag_bytes);
        for (int p = 0; p < nrs; ++p) {
            const int row = (rs0 + p) * 16 + r;
            f4 d = zero4();
            int len = a.T;
            if (row < a.M) {
                if (a.dc_final) d = ldf4(a.dc_final + (long)row * U + u);
                if (a.lens) len = a.lens[row];
            }
            stf4(stdc + (p * 64 + lane) * 4, d);
            stl[p * 64 + lane] = len;
        }
        int p = 0, j = 0;
        for (int n = 0; n < nticks; ++n) {
            // operands of this item that do not depend on the product: requested before the barrier
            const int t = a.T - 1 - j;
            const int row = (rs0 + p) * 16 + r;
            const bool valid = row < a.M;
            const long o = (long)row * U + u;
            const int len = stl[p * 64 + lane];
            const bool next_active = (t + 1 < a.T) && (t + 1 < len);
            const bool cur_active = (t >= 0) && (t < len);
            const bool has_gemm = j > 0;
            f4 zi = zero4(), zj = zero4(), zf = zero4(), zo = zero4(), cp = zero4(), cc = zero4(), dhx = zero4();
            if (valid) {
                if (!next_active && a.dh_final) dhx = ldf4(a.dh_final + o);
                if (cur_active) {
                    const float* zr = a.z + (long)t * a.zts + (long)row * a.zrs + u;
                    zi = ldf4(zr); zj = ldf4(zr + U); zf = ldf4(zr + 2L * U); zo = ldf4(zr + 3L * U);
                    if (t > 0) cp = ldf4(a.cs + (size_t)(t - 1) * a.M * U + o);
                    else if (a.c0) cp = ldf4(a.c0 + o);
                    cc = ldf4(a.cs + (size_t)t * a.M * U + o);
                    if (a.dhout) {
                        const f4 e4 = ldf4(a.dhout + (size_t)t * a.M * U + o);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dhx.v[e] += e4.v[e];
                    }
                }
            }
            ps_barrier();          // partial tiles of tick n are in P[n & 1]
            f4 dH = dhx;
            if (has_gemm) {
                const float* Pb = P + (n & 1) * 4 * 16 * PS_PLD + r * PS_PLD + q * 4;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const f4 pp = ldf4(Pb + w * 16 * PS_PLD);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dH.v[e] += pp.v[e];
                }
            }
            if (t < 0) {
                if (valid) stf4(a.dh0 + o, dH);
            } else {
                f4 g[4];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) g[gg] = zero4();
                if (cur_active) {
                    const f4 dcv = ldf4(stdc + (p * 64 + lane) * 4);
                    f4 dcn;
                    lstm_gate_bwd4(zi, zj, zf, zo, cp, cc, dH, dcv, g[0], g[1], g[2], g[3], dcn);
                    stf4(stdc + (p * 64 + lane) * 4, dcn);
                }
                if (valid) {
                    const int fo = (int)((t & 1) * a.dzfrag_bytes);
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg)
                        ps_st_sc1(dres, fo + (int)(d2p_frag_off(row, gg * U + u, KCx) * 4), g[gg].v[0], g[gg].v[1],
                                  g[gg].v[2], g[gg].v[3]);
                    float* dzr = a.dz + (long)t * a.zts + (long)row * a.zrs + u;
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) stf4(dzr + (long)gg * U, g[gg]);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) ps_st_flag(fbase + p * nnt + nt, (unsigned)(j + 1));
            if (++p == nrs) { p = 0; ++j; }
        }
        if (a.dc0)
            for (int qq = 0; qq < nrs; ++qq) {
                const int row = (rs0 + qq) * 16 + r;
                if (row < a.M) stf4(a.dc0 + (long)row * U + u, ldf4(stdc + (qq * 64 + lane) * 4));
            }
    }
}

// =============================================================================================
// Host side
// =============================================================================================
static int ps_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 1;
    }
    return n;
}

// row domains for `ncol` column tiles: as many as fit the chip, at most one per 16-row sub-tile
static int ps_pick_rt(int total_rs, int ncol) {
    int rt = ps_num_cus() / ncol;
    if (rt > total_rs) rt = total_rs;
    return rt;
}
static bool ps_shape_ok(int M, int U, int n_steps, int ncol) {
    if (M <= 0 || n_steps <= 0 || !(U == 64 || U == 128 || U == 256 || U == 512)) return false;
    const int total_rs = (M + 15) / 16;
    const int rt = ps_pick_rt(total_rs, ncol);
    if (rt < 1) return false;
    const long Mp = (long)total_rs * 16;
    if (2L * Mp * 4 * U * 4 > 0x7fffffffL) return false;        // 32-bit buffer offsets
    return (total_rs + rt - 1) / rt <= PS_NRS_MAX;
}
bool d2p_lstm_persist_fwd_ok(int M, int U, int n_steps) {
    return g_persist && ps_shape_ok(M, U, n_steps, U / 8);
}
bool d2p_lstm_persist_bwd_ok(int M, int U, int n_steps) {
    return g_persist && ps_shape_ok(M, U, n_steps, U / 16);
}

#define PS_FLAG_WORDS 4096   // >= RT * PS_NRS_MAX * ncol for any grid <= 512 workgroups

size_t d2p_lstm_persist_ws_bytes(int M, int U) {
    const size_t Mp = (size_t)((M + 15) / 16) * 16;
    // packed weight + 2 fragment buffers over K = 4U (backward; forward needs U) + flags
    return ((size_t)4 * U * U + 2 * Mp * 4 * U) * sizeof(float) + PS_FLAG_WORDS * sizeof(unsigned);
}

int d2p_lstm_persist_fwd(int M, int U, int n_steps, float* z, long zrs, long zts, const float* Wh,
                         const float* h0, const float* c0, const int* lens, float* hout, float* cs,
                         float* h_final, float* c_final, float* ws, hipStream_t st) {
    PsFwdArgs a;
    a.M = M; a.U = U; a.T = n_steps;
    a.total_rs = (M + 15) / 16;
    const int nct = U / 8;
    a.RT = ps_pick_rt(a.total_rs, nct);
    a.has_h0 = h0 ? 1 : 0;
    const size_t Mp = (size_t)a.total_rs * 16;
    float* Wf = ws;
    a.Wf = (const float4*)Wf;
    a.hfrag = Wf + (size_t)4 * U * U;
    a.hfrag_bytes = (unsigned)(Mp * U * sizeof(float));
    a.flags = (unsigned*)(a.hfrag + 2 * Mp * U);
    a.err = ps_err_ptr();
    a.z = z; a.zrs = zrs; a.zts = zts; a.h0 = h0; a.c0 = c0; a.lens = lens;
    a.hout = hout; a.cs = cs; a.h_final = h_final; a.c_final = c_final;
    int rc = d2p_lstm_pack_w_fwd(U, Wh, Wf, st);
    if (rc) return rc;
    if (h0) {
        rc = d2p_lstm_pack_rows(M, U, a.total_rs, h0, a.hfrag, st);
        if (rc) return rc;
    }
    D2P_HIP(hipMemsetAsync(a.flags, 0, (size_t)a.RT * PS_NRS_MAX * nct * sizeof(unsigned), st));
    const int blocks = nct * a.RT;
    {
        D2pProfScope prof(st, D2P_PROF_LSTM_STEP_FWD, 2.0 * M * 4.0 * U * U * (n_steps - (h0 ? 0 : 1)));
        switch (U) {
            case 64: hipLaunchKernelGGL((lstm_persist_fwd_kernel<1>), dim3(blocks), dim3(PS_THREADS), 0, st, a); break;
            case 128: hipLaunchKernelGGL((lstm_persist_fwd_kernel<2>), dim3(blocks), dim3(PS_THREADS), 0, st, a); break;
            case 256: hipLaunchKernelGGL((lstm_persist_fwd_kernel<4>), dim3(blocks), dim3(PS_THREADS), 0, st, a); break;
            default: hipLaunchKernelGGL((lstm_persist_fwd_kernel<8>), dim3(blocks), dim3(PS_THREADS), 0, st, a); break;
        }
    }
    D2P_LAUNCH_CHECK("lstm_persist_fwd");
    return D2P_OK;
}

int d2p_lstm_persist_bwd(int M, int U, int n_steps, const float* z, long zrs, long zts, const float* Wh,
                         const float* c0, const int* lens, const float* cs, const float* dhout,
                         const float* dh_final, const float* dc_final, float* dz, float* dh0,
                         float* dc0, float* ws, hipStream_t st) {
    PsBwdArgs a;
    a.M = M; a.U = U; a.T = n_steps;
    a.total_rs = (M + 15) / 16;
    const int nnt = U / 16;
    a.RT = ps_pick_rt(a.total_rs, nnt);
    a.want_dh0 = dh0 ? 1 : 0;
    const size_t Mp = (size_t)a.total_rs * 16;
    float* Wb = ws;
    a.Wb = (const float4*)Wb;
    a.dzfrag = Wb + (size_t)4 * U * U;
    a.dzfrag_bytes = (unsigned)(Mp * 4 * U * sizeof(float));
    a.flags = (unsigned*)(a.dzfrag + 2 * Mp * 4 * U);
    a.err = ps_err_ptr();
    a.z = z; a.zrs = zrs; a.zts = zts; a.c0 = c0; a.cs = cs; a.lens = lens;
    a.dhout = dhout; a.dh_final = dh_final; a.dc_final = dc_final;
    a.dz = dz; a.dh0 = dh0; a.dc0 = dc0;
    int rc = d2p_lstm_pack_w_bwd(U, Wh, Wb, st);
    if (rc) return rc;
    D2P_HIP(hipMemsetAsync(a.flags, 0, (size_t)a.RT * PS_NRS_MAX * nnt * sizeof(unsigned), st));
    const int blocks = nnt * a.RT;
    {
        D2pProfScope prof(st, D2P_PROF_LSTM_STEP_BWD, 2.0 * M * 4.0 * U * U * (n_steps - 1 + (dh0 ? 1 : 0)));
        switch (U) {
            case 64: hipLaunchKernelGGL((lstm_persist_bwd_kernel<1>), dim3(blocks), dim3(PS_THREADS), 0, st, a); break;
            case 128: hipLaunchKernelGGL((lstm_persist_bwd_kernel<2>), dim3(blocks), dim3(PS_THREADS), 0, st, a); break;
            case 256: hipLaunchKernelGGL((lstm_persist_bwd_kernel<4>), dim3(blocks), dim3(PS_THREADS), 0, st, a); break;
            default: hipLaunchKernelGGL((lstm_persist_bwd_kernel<8>), dim3(blocks), dim3(PS_THREADS), 0, st, a); break;
        }
    }
    D2P_LAUNCH_CHECK("lstm_persist_bwd");
    return D2P_OK;
}
