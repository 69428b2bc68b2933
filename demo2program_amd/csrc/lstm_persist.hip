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
//   * wave roles: waves 0-3 = MFMA waves: K split four ways, partial tiles combined through LDS, and
//     after the combine barrier each wave does the gate math of 4 of the phase's 16 rows ITSELF.
//     (fp32 MFMA executes on the SIMD's vector ALUs: a separate epilogue wave, at any priority, got
//     one VALU issue slot per 32-clock MFMA of its SIMD-mate -- measured 3400-4000 clocks for ~100
//     instructions, 1.7x the phase's whole MFMA time.  In the MFMA wave's own stream the same
//     instructions cost their issue time.)  Their global stores are unconditional (masked rows go to
//     a dump line), so the compiler's counted vmcnt waits never turn into a drain.
//     wave 4 = publish + prefetch wave: after the epilogue barrier it writes the new h / dz rows the
//     MFMA waves staged in LDS with write-through stores, drains them (the only wave that ever waits
//     for a store) and raises the flag; it also streams everything the epilogue reads that does NOT
//     depend on the recurrence (hoisted input projections; saved z / c / dhout in backward)
//     PS_PF_D phases ahead into an LDS ring with LDS-DMA loads (HBM latency, ~1-2 us under load,
//     would otherwise sit on the critical hand-off path).  The MFMA waves fetch the next phase's
//     operands into a second register set one phase ahead, behind a flag whose read was issued one
//     more phase earlier (an sc1 read of a freshly written line takes ~0.8 us).
//   * hand-off protocol (cdna_hip_programming.md Guideline 16, form R1): payload = 16-byte
//     write-through (sc1) stores by ONE wave -> s_waitcnt vmcnt(K), K = the DMA loads issued behind them -> one relaxed agent-scope flag
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
#define PS_THREADS 320      // 4 MFMA waves + the publish / prefetch wave
#define PS_PF_D 3           // ticks the operand prefetch runs ahead of the epilogue
#define PS_PF_R (PS_PF_D + 2)   // LDS ring slots (one more than the distance: a deferred epilogue reads its slot a tick late)
#define PS_SPIN_LIMIT 60000 // ~50 ms of polling before giving up
#define PS_AUX_SC1 16       // buffer-instruction cache policy: sc1 (agent scope, bypasses the CU's L1)
#define PS_FLAG_WORDS 4096   // >= RT * PS_NRS_MAX * ncol for any grid <= 512 workgroups
#define PS_TICKET_WORDS 64   // bias-gradient tickets of the backward kernel, one per column tile (zeroed with the flags)
#define PS_RT_TAB 8          // row domains a length-sorted launch can describe (backward, U = 512: 8 domains)

// ---- error word ---------------------------------------------------------------------------
__device__ unsigned g_ps_err;          // sticky: 0 ok, else (code << 24) | block
static unsigned* ps_err_ptr() {
    static unsigned* p = nullptr;
    if (!p) (void)hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_ps_err));
    return p;
}
unsigned* d2p_persist_err_ptr() { return ps_err_ptr(); }     // adam.hip, bn.hip: the guarded optimizer step
// Test hook: sets the status word as a timed-out hand-off would (code 0x7f), without a real failure.
extern "C" int d2p_lstm_persist_inject_error(void) {
    const unsigned v = (0x7fu << 24) | 0x800000u;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ps_err), &v, sizeof(v)) == hipSuccess ? D2P_OK : D2P_EINVAL;
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
static int ps_num_cus();
extern "C" int d2p_lstm_set_persistent(int on) {
    g_persist = on ? 1 : 0;
    if (on) (void)ps_num_cus();
    return D2P_OK;
}
int d2p_lstm_is_persistent_enabled() { return g_persist; }

// ---- optional timeline trace (tools/trace_lstm_persist.py) -----------------------------------
// One workgroup records shader-clock stamps of every tick: role 0 = MFMA wave 0, role 1 = epilogue.
#define PS_TR_MAXT 512
#define PS_TR_K 8
static unsigned long long* g_ps_trace = nullptr;
static int g_ps_trace_block = 0;
extern "C" int d2p_lstm_persist_set_trace(void* buf, size_t bytes, int block) {
    D2P_REQUIRE(!buf || bytes >= (size_t)2 * PS_TR_MAXT * PS_TR_K * sizeof(unsigned long long), D2P_EWS,
                "lstm persist trace: buffer too small");
    g_ps_trace = (unsigned long long*)buf;
    g_ps_trace_block = block;
    return D2P_OK;
}
struct PsTrace {
    unsigned long long* buf;   // null: off
    unsigned long long st[PS_TR_K];
    __device__ __forceinline__ void stamp(int k) {
        if (buf) st[k] = __builtin_readcyclecounter();
    }
    __device__ __forceinline__ void flush(int role, int n, int lane) {
        if (buf && n < PS_TR_MAXT && lane < PS_TR_K) {
            unsigned long long v = st[0];
#pragma unroll
            for (int k = 1; k < PS_TR_K; ++k) v = (lane == k) ? st[k] : v;
            buf[((long)role * PS_TR_MAXT + n) * PS_TR_K + lane] = v;
        }
    }
};

// ---- launch-level stamps (tools/lstm_launch_stamps.py; a DIAGNOSTIC build of this file with -DD2P_PS_STAMPS,
// linked into its own library -- the product library never carries them: PS_STAMP expands to nothing) ------------
// Wave 0 (role 0: the first MFMA wave) and wave 4 (role 1: the publish / prefetch wave) of EVERY workgroup leave the
// 100 MHz s_memrealtime counter (one clock for the whole chip, unlike the shader clock of PsTrace) at the boundaries
// of a launch: 0 entry, 1 tables / initial state in LDS (weights requested), 3 last tick done, 4 bias-gradient
// exchange done, 5 final stores issued, 6 final stores out; 7 = geometry word.  Nothing inside the tick loops: a
// stamp there changes hipcc's schedule of the loop (the backward launches ran 8 % slower with one).
#ifdef D2P_PS_STAMPS
#define PS_ST_K 8
#define PS_ST_MAXB 512
__device__ unsigned long long* g_ps_stamp_buf = nullptr;    // [slot][PS_ST_MAXB blocks][2 roles][PS_ST_K]
__device__ int g_ps_stamp_slot = -1;
static int g_ps_stamp_slots = 0, g_ps_stamp_next = 0;
__global__ void ps_stamp_slot_kernel(unsigned long long* buf, int slot) {
    g_ps_stamp_buf = buf;
    g_ps_stamp_slot = slot;
}
static unsigned long long* g_ps_stamp_host_buf = nullptr;
extern "C" int d2p_lstm_persist_set_stamps(void* buf, size_t bytes) {
    g_ps_stamp_host_buf = (unsigned long long*)buf;
    g_ps_stamp_slots = buf ? (int)(bytes / ((size_t)PS_ST_MAXB * 2 * PS_ST_K * sizeof(unsigned long long))) : 0;
    g_ps_stamp_next = 0;
    return D2P_OK;
}
extern "C" int d2p_lstm_persist_stamp_launches(void) { return g_ps_stamp_next; }
// in stream order in front of a persistent launch: the slot its workgroups write (-1: none left / off)
static void ps_stamp_before_launch(hipStream_t st) {
    const int slot = (g_ps_stamp_host_buf && g_ps_stamp_next < g_ps_stamp_slots) ? g_ps_stamp_next++ : -1;
    hipLaunchKernelGGL(ps_stamp_slot_kernel, dim3(1), dim3(1), 0, st, g_ps_stamp_host_buf, slot);
}
__device__ __forceinline__ void ps_stamp(int role, int k, unsigned long long v = 0ull) {
    unsigned long long* b = g_ps_stamp_buf;
    const int s = g_ps_stamp_slot;
    if (b && s >= 0 && (threadIdx.x & 63) == 0 && blockIdx.x < PS_ST_MAXB)
        b[(((long)s * PS_ST_MAXB + blockIdx.x) * 2 + role) * PS_ST_K + k] = k == 7 ? v : __builtin_amdgcn_s_memrealtime();
}
#define PS_STAMP(role, k) ps_stamp(role, k);
#define PS_STAMP_W0(k) if (wave == 0) ps_stamp(0, k);
#define PS_STAMP_ENTRY(geom) if (wave == 0 || wave == 4) { ps_stamp(wave >> 2, 0); ps_stamp(wave >> 2, 7, geom); }
#define PS_STAMP_EXIT_W0 if (wave == 0) { ps_stamp(0, 5); ps_wait_vmcnt<0>(); ps_stamp(0, 6); }
#define PS_STAMP_LAUNCH(st) ps_stamp_before_launch(st);
#else   // (every macro expands to NOTHING -- not even an empty statement: an empty `if` in front of the tick loops changed
        //  hipcc's code for the backward kernels, 54 012 -> 53 704 bytes for U = 512)
#define PS_STAMP(role, k)
#define PS_STAMP_W0(k)
#define PS_STAMP_ENTRY(geom)
#define PS_STAMP_EXIT_W0
#define PS_STAMP_LAUNCH(st)
#endif

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
// The polls are PIPELINED: two reads of the flag are in flight at any time (write-through-scope buffer loads, which the
// compiler lets overlap -- relaxed atomic loads are completed one by one), so a flag that is raised between two polls is
// seen half a round trip earlier than by read / wait / sleep / read.  `base`: the flag area (for the descriptor).
// the poll word of a single-phase domain's waits: bit 1 -> bit 0, and the pauses of bits 5-7 instead of 2-4 when bit 8 is set
__device__ __forceinline__ int ps_poll_single(int w) {
    const int naps = (w & 256) ? ((w >> 5) & 7) : ((w >> 2) & 7);
    return ((w >> 1) & 1) | (naps << 2);
}
static int g_ps_poll_pipelined = 16;     // bits 0-1 (pipelined polls): off -- faster kernels in isolation, a slower training step;
                                         // bits 2-4 (extra pauses between polls): 4 -- fewer polls leave the other queue's GEMMs more of L2
__device__ __forceinline__ unsigned ps_ld_flag_buf(__amdgpu_buffer_rsrc_t r, int off) {
    const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, PS_AUX_SC1);
    asm volatile("" ::: "memory");          // a new read every time: never merged with the previous one
    return v;
}
template <bool REREAD = false>
__device__ __forceinline__ void ps_wait_flags(const unsigned* f, unsigned need, unsigned fv, unsigned* err,
                                              unsigned code, const unsigned* base, int pipelined) {
    if (__all((int)(fv >= need))) return;
    if (REREAD) {
        // `fv` was read a while ago: look again before the first pause (a pause is ~0.5 us)
        fv = ps_ld_flag(f);
        if (__all((int)(fv >= need))) return;
    }
    unsigned spins = 0;
    if (pipelined & 3) {
        const __amdgpu_buffer_rsrc_t r = ps_rsrc(base, (unsigned)((PS_FLAG_WORDS + PS_TICKET_WORDS) * sizeof(unsigned)));
        const int off = (int)((const char*)f - (const char*)base);
        unsigned a = ps_ld_flag_buf(r, off);
        for (;;) {
            const unsigned b = ps_ld_flag_buf(r, off);
            if (__all((int)(a >= need))) return;
            a = ps_ld_flag_buf(r, off);
            if (__all((int)(b >= need))) return;
            spins += 2;
            if ((spins & 63u) == 0u) {
                if (ps_ld_flag(err) != 0u) return;
                if (spins > PS_SPIN_LIMIT) {
                    if ((threadIdx.x & 63) == 0) ps_st_flag(err, (code << 24) | (blockIdx.x & 0xffffffu) | 0x800000u);
                    return;
                }
            }
        }
    }
    const int naps = (pipelined >> 2) & 7;        // pauses between two polls: (1 + naps) x s_sleep(4) (bits 2-4 of the poll word)
    for (;;) {
        __builtin_amdgcn_s_sleep(4);
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(4);
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
// b, g: the workgroup's index in and the size of ITS sequence's group of the launch (a launch carries one
// sequence, or two independent ones on disjoint workgroups)
__device__ __forceinline__ void ps_block_tile(int b, int g, int ncol, int& rt, int& ct) {
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
// Shared pieces
// =============================================================================================
typedef __attribute__((address_space(3))) void ps_lds_void;
typedef __attribute__((address_space(1))) const void ps_glb_void;

// 64 lanes x 16 bytes from per-lane global addresses straight into LDS at dst + lane*16
__device__ __forceinline__ void ps_dma16(const float* src, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((ps_glb_void*)src, (ps_lds_void*)lds_dst, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void ps_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct PsTick {     // (phase, step/pass) of a tick index, advanced incrementally
    int p, t;
    __device__ __forceinline__ void next(int nrs) {
        if (++p == nrs) { p = 0; ++t; }
    }
};

// =============================================================================================
// Forward
// =============================================================================================
struct PsFwdArgs {
    int M, U, T, total_rs, RT, has_h0;
    int bid0, gsz;          // this sequence's workgroups are blocks [bid0, bid0 + gsz) of the launch
    const float4* Wf;       // packed Wh (lstm_step.hip forward layout)
    float* hfrag;           // 2 ping-pong buffers of Mp*U floats, fragment-major; [0] = state before step 0
    unsigned hfrag_bytes;   // bytes of ONE buffer
    float* z; long zrs, zts;
    const float* h0; const float* c0; const int* lens;
    float* hout; float* cs; float* h_final; float* c_final;
    unsigned* flags;        // [RT][PS_NRS_MAX][U/8], zeroed before the launch
    float* dump;            // 64 floats nobody reads: target of masked lanes' stores
    unsigned* err;
    unsigned long long* trace; int trace_block;
    // "direct" launches (round 3: no preparation launch in front of the kernel):
    //   * the recurrent weight is read straight from the row-major Wh (wh_raw) into the registers,
    //   * the operands of step 0 come from the row-major initial state through a second buffer descriptor (h0_raw,
    //     h0_bytes = M*U*4; 0 bytes without an initial state: every load returns zero, as do rows >= M),
    //   * the flags are never reset: every launch publishes epoch + t + 1 and waits for epoch + t, and the caller
    //     raises the epoch by more than T from launch to launch (a dedicated, once-zeroed flag buffer).
    int direct;
    const float* wh_raw; const float* h0_raw; unsigned h0_bytes; unsigned epoch;
    int packed;             // direct launch with a caller-kept packed weight image in Wf (d2p_lstm_pack_weights)
    int poll;               // bit 0: pipelined flag polls (ps_wait_flags) in domains that look ahead, bit 1: in single-phase ones too
};

// where a phase's 16 rows of the running state are read from: descriptor, byte offset of chunk 0, bytes per chunk
struct PsSrc {
    __amdgpu_buffer_rsrc_t r;
    int off, cs;
};
__device__ __forceinline__ f32x4 ps_ld_src(const PsSrc& s, int c) { return ps_ld_sc1(s.r, s.off + c * s.cs); }
// step t, phase p of a forward sequence: the fragment-major ping-pong buffer, or (direct launches, t = 0) the
// row-major initial state: lane l = (row l&15, k 4*(l>>4)..+3) of chunk kc = wave*CPW + c -- a select, no branch
__device__ __forceinline__ PsSrc ps_fwd_src(const PsFwdArgs& a, __amdgpu_buffer_rsrc_t hres, __amdgpu_buffer_rsrc_t hres0,
                                            int t, int rs, int KC, int lane_off, int rm_off) {
    const bool first = a.direct && t == 0;
    PsSrc s;
    s.r = first ? hres0 : hres;
    s.off = first ? rs * 16 * a.U * 4 + rm_off : (int)((t & 1) * a.hfrag_bytes) + rs * KC * 1024 + lane_off;
    s.cs = first ? 64 : 1024;
    return s;
}
__device__ __forceinline__ unsigned ps_need(unsigned epoch, int t) { return t == 0 ? 0u : epoch + (unsigned)t; }

template <int CPW>
__device__ __forceinline__ void ps_fwd_chain(const f32x4 (&av)[CPW], const f32x4 (&bv)[CPW][2], f32x4& acc0,
                                             f32x4& acc1, int c0 = 0, int c1 = CPW) {
#pragma unroll
    for (int c = c0; c < c1; ++c)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][jj], bv[c][0][jj], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][jj], bv[c][1][jj], acc1, 0, 0, 0);
        }
}

#define PS_FWD_SLOT 512     // floats per prefetch ring slot: 16 rows x 4 gates x 8 units

struct PsFwdEpi {           // per-lane constants of an MFMA wave's share of the epilogue
    int rr, un, u;          // row within the phase, unit within the tile, global unit
    bool lane_on;
    float *P, *stc, *sth, *stage, *ring, *spare;   // P, stage: two buffers each (tick parity)
    int* stl;
};
#define PS_P_FLOATS (4 * 16 * PS_PLD)

// Gate math of this wave's 4 rows x 8 units of phase (p, t): lanes 0-31 one cell each, in two parts so
// that the deferred form below can issue the LDS reads early in the next phase's MFMA chain and do the
// arithmetic late in it (a read next to its use makes the in-order wave -- MFMAs included -- wait out the
// LDS latency every time).  No branch in here.
struct PsFwdEpiIn {
    float zin[4], part[4][4], cp, hp;
    int len;
};
__device__ __forceinline__ void ps_fwd_epilogue_load(const PsFwdEpi& e, int p, int slot, int par, PsFwdEpiIn& in) {
    const int sidx = (p * 16 + e.rr) * 8 + e.un;
    const float* zs = e.ring + slot * PS_FWD_SLOT + e.rr * 32 + e.un;
#pragma unroll
    for (int g = 0; g < 4; ++g) in.zin[g] = zs[g * 8];
    // the partial tiles are zero when the step has no product (t = 0 without an initial state)
    const float* Pb = e.P + par * PS_P_FLOATS + e.rr * PS_PLD + e.un;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int w = 0; w < 4; ++w) in.part[g][w] = Pb[w * 16 * PS_PLD + g * 8];
    in.cp = e.stc[sidx];
    in.hp = e.sth[sidx];
    in.len = e.stl[sidx];
}
__device__ __forceinline__ void ps_fwd_epilogue_finish(const PsFwdArgs& a, const PsFwdEpi& e, int rs0, int p, int t,
                                                       int par, const PsFwdEpiIn& in) {
    const int U = a.U;
    const int row = (rs0 + p) * 16 + e.rr;
    const bool valid = e.lane_on && row < a.M;
    const int sidx = (p * 16 + e.rr) * 8 + e.un;
    const bool active = t < in.len;
    float zz[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        zz[g] = in.zin[g];
#pragma unroll
        for (int w = 0; w < 4; ++w) zz[g] += in.part[g][w];
    }
    float cn, hn;
    lstm_cell_fwd(zz[0], zz[1], zz[2], zz[3], in.cp, cn, hn);
    const float c_out = active ? cn : in.cp;
    const float h_out = active ? hn : 0.f;        // emitted output: 0 past the row's length
    const float h_state = active ? hn : in.hp;    // (c, h) copy through
    // lanes 32-63 duplicate lanes 0-31 and write a spare word instead
    *(e.lane_on ? e.stc + sidx : e.spare) = c_out;
    *(e.lane_on ? e.sth + sidx : e.spare) = h_state;
    // staged in fragment-major order: the publish wave's lane l = (quad l>>4, row l&15) reads a float4
    *(e.lane_on ? e.stage + par * 128 + (((e.un >> 2) << 4) + e.rr) * 4 + (e.un & 3) : e.spare) = h_state;
    // unconditional stores (masked lanes -> dump line; rows that keep their input projection write
    // it back unchanged): the counted waits for the operand prefetch then never wait for a store.
    // Addresses are formed for a clamped row and only the final offset is selected: with the whole
    // address expression under the select the compiler branches around it (and splits the block).
    const int rowc = min(row, a.M - 1);
    const long o = (long)rowc * U + e.u;
    const long zo = (long)t * a.zts + (long)rowc * a.zrs + e.u;
    const long so = (long)t * a.M * U + o;
    const long dz = a.dump - a.z, dc = a.dump - a.cs, dh = a.dump - a.hout;    // wave-uniform
    float* zr = a.z + (valid ? zo : dz);
    const long zg = valid ? (long)U : 0L;
    const bool keep = !(active && ((t > 0) || a.has_h0));
#pragma unroll
    for (int g = 0; g < 4; ++g) zr[g * zg] = keep ? in.zin[g] : zz[g];
    a.cs[valid ? so : dc] = c_out;
    a.hout[valid ? so : dh] = h_out;
}
__device__ __forceinline__ void ps_fwd_epilogue(const PsFwdArgs& a, const PsFwdEpi& e, int rs0, int p, int t, int slot,
                                                int par) {
    PsFwdEpiIn in;
    ps_fwd_epilogue_load(e, p, slot, par, in);
    ps_fwd_epilogue_finish(a, e, rs0, p, t, par, in);
}

__device__ __forceinline__ void ps_fwd_write_partials(float* Pw, int lane, const f32x4& acc0, const f32x4& acc1) {
    // C/D layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Pw[((lane >> 4) * 4 + r) * PS_PLD + (lane & 15)] = acc0[r];
        Pw[((lane >> 4) * 4 + r) * PS_PLD + 16 + (lane & 15)] = acc1[r];
    }
}

// One phase of one MFMA wave, epilogue right behind its own product (domains with < 4 phases).  With
// look-ahead (la: >= 2 phases) the rows of the NEXT phase are requested at the start of this one,
// behind a flag whose read `fv` was issued one phase earlier, and the flag of the phase after that is
// read for the next call.  A single-phase domain's next tick depends on THIS tick's epilogue: nothing
// can be fetched ahead, `cur` is loaded behind a blocking poll instead.
template <int CPW, bool la>
__device__ __forceinline__ void ps_fwd_tick(f32x4 (&cur)[CPW], f32x4 (&nxt)[CPW], const f32x4 (&bv)[CPW][2],
                                            const PsFwdArgs& a, const PsFwdEpi& e, PsTick k0,
                                            const unsigned* fl_cur, const PsSrc& cur_src, const unsigned* fl1,
                                            unsigned need1, const PsSrc& src1, const unsigned* fl2, unsigned& fv,
                                            int wave, int rs0, int slot, int lane, PsTrace& tr) {
    tr.stamp(0);
    if (la) {
        ps_wait_flags(fl1, need1, fv, a.err, 1, a.flags, a.poll);
    } else {
        ps_wait_flags(fl_cur, ps_need(a.epoch, k0.t), ps_ld_flag(fl_cur), a.err, 3, a.flags, ps_poll_single(a.poll));
#pragma unroll
        for (int c = 0; c < CPW; ++c) cur[c] = ps_ld_src(cur_src, c);
    }
    tr.stamp(1);
#pragma unroll
    for (int c = 0; c < CPW; ++c) nxt[c] = ps_ld_src(src1, c);
    fv = ps_ld_flag(fl2);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if ((k0.t > 0) || a.has_h0) ps_fwd_chain<CPW>(cur, bv, acc0, acc1);
    tr.stamp(2);
    ps_fwd_write_partials(e.P + wave * 16 * PS_PLD, lane, acc0, acc1);
    ps_barrier();           // A: all four partial tiles of this phase are in LDS
    tr.stamp(3);
    ps_fwd_epilogue(a, e, rs0, k0.p, k0.t, slot, 0);
    ps_barrier();           // B: new rows staged, P and the ring slot free again
    tr.stamp(4);
}

// Deferred form (domains with >= 4 phases): the gate math of the PREVIOUS phase runs inside this
// phase's MFMA chain -- one basic block, and the scheduler is told to put two VALU and one LDS read
// behind every MFMA (three VALU / SALU / LDS instructions per MFMA): the matrix pipe and the vector ALU are separate, so in the MFMA wave's own
// instruction stream the epilogue costs (almost) nothing, where behind the chain it cost 1300 of a
// phase's 4750 clocks.  One barrier per phase; the new rows are published one phase later, which a
// domain with >= 4 phases in flight can afford.
template <int CPW>
__device__ __forceinline__ void ps_fwd_tick_defer(const f32x4 (&cur)[CPW], f32x4 (&nxt)[CPW], const f32x4 (&bv)[CPW][2],
                                                  const PsFwdArgs& a, const PsFwdEpi& e, PsTick kprev, int slot_prev,
                                                  const unsigned* fl1, unsigned need1, const PsSrc& src1,
                                                  const unsigned* fl2, unsigned& fv, int wave, int rs0,
                                                  int par, int lane, PsTrace& tr) {
    tr.stamp(0);
    ps_wait_flags(fl1, need1, fv, a.err, 1, a.flags, a.poll);
    tr.stamp(1);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int C1 = CPW >= 4 ? CPW / 4 : 0, C2 = CPW >= 4 ? CPW / 2 : 0;
    // first quarter of the chain with the next phase's operand loads (~60 clocks of issue each) and
    // the previous phase's LDS reads issued into it ...
#pragma unroll
    for (int c = 0; c < CPW; ++c) nxt[c] = ps_ld_src(src1, c);
    fv = ps_ld_flag(fl2);
    PsFwdEpiIn in;
    ps_fwd_epilogue_load(e, kprev.p, slot_prev, par ^ 1, in);
    ps_fwd_chain<CPW>(cur, bv, acc0, acc1, 0, C1);
#pragma unroll
    for (int i = 0; i < 8 * C1; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one operand load
        __builtin_amdgcn_sched_group_barrier(0x186, 3, 0);      // three of VALU / SALU / LDS read
    }
    __builtin_amdgcn_sched_barrier(0);
    // ... a quarter for them to land ...
    ps_fwd_chain<CPW>(cur, bv, acc0, acc1, C1, C2);
    __builtin_amdgcn_sched_barrier(0);
    // ... and the arithmetic and stores spread over the second half
    ps_fwd_chain<CPW>(cur, bv, acc0, acc1, C2, CPW);
    ps_fwd_epilogue_finish(a, e, rs0, kprev.p, kprev.t, par ^ 1, in);
#pragma unroll
    for (int i = 0; i < 8 * (CPW - C2); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x3d6, 4, 0);      // four of VALU / SALU / LDS / VMEM write
    }
    __builtin_amdgcn_sched_barrier(0);
    tr.stamp(2);
    ps_fwd_write_partials(e.P + par * PS_P_FLOATS + wave * 16 * PS_PLD, lane, acc0, acc1);
    ps_barrier();           // partial tiles of this phase and the staged rows of the previous one are in LDS
    tr.stamp(3);
    tr.stamp(4);
}

// All ticks of one MFMA wave.  Two ticks per iteration with the two operand register sets swapping
// roles, so they are never copied.  MODE 0: no look-ahead (1 phase), 1: look-ahead, 2: look-ahead +
// deferred epilogue (>= 4 phases).
template <int CPW, int MODE>
__device__ __forceinline__ void ps_fwd_mfma_wave(const PsFwdArgs& a, const PsFwdEpi& e, const f32x4 (&bv)[CPW][2],
                                                 __amdgpu_buffer_rsrc_t hres, __amdgpu_buffer_rsrc_t hres0,
                                                 const unsigned* fl, int nct, int rs0,
                                                 int nrs, int nticks, int lane_off, int rm_off, int wave, int lane) {
    constexpr int KC = 4 * CPW;
    PsTrace tr;
    tr.buf = (blockIdx.x == a.trace_block) ? a.trace : nullptr;
    f32x4 a0[CPW], a1[CPW];
    {
        const PsSrc s0 = ps_fwd_src(a, hres, hres0, 0, rs0, KC, lane_off, rm_off);
#pragma unroll
        for (int c = 0; c < CPW; ++c) a0[c] = ps_ld_src(s0, c);
    }
    PsTick kp = {0, 0}, k0 = {0, 0}, k1 = {0, 0}, k2 = {0, 0};      // ticks n-1, n, n+1, n+2
    k1.next(nrs);
    k2.next(nrs); k2.next(nrs);
    unsigned fv = ps_ld_flag(fl + k1.p * nct);
    int slot = 0, slot_prev = 0;
#define PS_FWD_ONE_TICK(CUR, NXT, m)                                                                          \
    {                                                                                                         \
        /* ticks past the end are clamped to the last one: their loads are issued unconditionally */         \
        const bool e1 = (m) + 1 >= nticks, e2 = (m) + 2 >= nticks;                                            \
        const PsTick q1 = e1 ? k0 : k1, q2 = e2 ? (e1 ? k0 : k1) : k2;                                        \
        const unsigned need1 = e1 ? 0u : ps_need(a.epoch, q1.t);     /* version t = published after step t-1 */ \
        const PsSrc cur_src = ps_fwd_src(a, hres, hres0, k0.t, rs0 + k0.p, KC, lane_off, rm_off);             \
        const PsSrc src1 = ps_fwd_src(a, hres, hres0, q1.t, rs0 + q1.p, KC, lane_off, rm_off);                \
        if (MODE == 2)                                                                                        \
            ps_fwd_tick_defer<CPW>(CUR, NXT, bv, a, e, kp, slot_prev, fl + q1.p * nct, need1, src1, fl + q2.p * nct, \
                                   fv, wave, rs0, (m) & 1, lane, tr);                                         \
        else                                                                                                  \
            ps_fwd_tick<CPW, MODE != 0>(CUR, NXT, bv, a, e, k0, fl + k0.p * nct, cur_src, fl + q1.p * nct, need1, \
                                        src1, fl + q2.p * nct, fv, wave, rs0, slot, lane, tr);                \
        if (wave == 0) tr.flush(0, (m), lane);                                                                \
        kp = k0; k0 = k1; k1 = k2; k2.next(nrs);                                                              \
        slot_prev = slot;                                                                                     \
        if (++slot == PS_PF_R) slot = 0;                                                                      \
    }
    int n = 0;
    if (MODE == 2) {
        // first tick: nothing to finish yet (plain product, one barrier)
        const unsigned need1 = ps_need(a.epoch, k1.t);
        const PsSrc src1 = ps_fwd_src(a, hres, hres0, k1.t, rs0 + k1.p, KC, lane_off, rm_off);
        ps_wait_flags(fl + k1.p * nct, need1, fv, a.err, 1, a.flags, a.poll);
#pragma unroll
        for (int c = 0; c < CPW; ++c) a1[c] = ps_ld_src(src1, c);
        fv = ps_ld_flag(fl + k2.p * nct);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        ps_fwd_chain<CPW>(a0, bv, acc0, acc1);
        ps_fwd_write_partials(e.P + wave * 16 * PS_PLD, lane, acc0, acc1);
        ps_barrier();
        kp = k0; k0 = k1; k1 = k2; k2.next(nrs);
        slot_prev = slot;
        ++slot;
        n = 1;
#pragma unroll 1
        for (; n + 1 < nticks; n += 2) {
            PS_FWD_ONE_TICK(a1, a0, n)
            PS_FWD_ONE_TICK(a0, a1, n + 1)
        }
        if (n < nticks) PS_FWD_ONE_TICK(a1, a0, n)
        // the last phase's gate math (kp is the last tick now)
        ps_fwd_epilogue(a, e, rs0, kp.p, kp.t, slot_prev, (nticks - 1) & 1);
    } else {
#pragma unroll 1
        for (; n + 1 < nticks; n += 2) {
            PS_FWD_ONE_TICK(a0, a1, n)
            PS_FWD_ONE_TICK(a1, a0, n + 1)
        }
        if (n < nticks) PS_FWD_ONE_TICK(a0, a1, n)
    }
#undef PS_FWD_ONE_TICK
}

template <int CPW>   // U = 64 * CPW
__global__ void __launch_bounds__(PS_THREADS) lstm_persist_fwd_kernel(PsFwdArgs a0, PsFwdArgs a1) {
    // two independent sequences may share a launch (a1.gsz > 0): the first a0.gsz workgroups run a0, the rest
    // a1 -- nothing is exchanged between the two groups, they only have to be resident together
    const PsFwdArgs a = ((int)blockIdx.x >= a0.gsz) ? a1 : a0;
    constexpr int KC = 4 * CPW;
    // ONE shared array (a second __shared__ object de-pipelines loads, cdna_hip_programming.md)
    __shared__ __attribute__((aligned(16))) float lds[2 * PS_P_FLOATS + PS_NRS_MAX * 384 + 64 + 2 * 128 + PS_PF_R * PS_FWD_SLOT];
    float* P = lds;                                            // [2][4][16][PS_PLD]
    float* stc = lds + 2 * PS_P_FLOATS;                        // [NRS][16 rows][8 units] cell state
    float* sth = stc + PS_NRS_MAX * 128;                       // [NRS][16][8] hidden state
    int* stl = reinterpret_cast<int*>(sth + PS_NRS_MAX * 128); // [NRS][16][8] row length
    float* spare = stc + PS_NRS_MAX * 384;                     // [64] write target of the duplicate lanes
    float* stage = spare + 64;                                 // [2][2 quads][16 rows][4]: new h rows of a phase
    float* ring = stage + 2 * 128;                             // [PS_PF_R][16 rows][4 gates][8 units]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int U = a.U, nct = U >> 3;
    int rt, ct;
    ps_block_tile((int)blockIdx.x - a.bid0, a.gsz, nct, rt, ct);
    int rs0, nrs;
    ps_rt_range(rt, a.total_rs, a.RT, rs0, nrs);
    const int nticks = nrs * a.T;
    unsigned* fbase = a.flags + (long)rt * PS_NRS_MAX * nct;
    // deferred epilogue: the domain must have phases to spare for the later publication, and every
    // step must have a product (the host zero-fills the state buffer when there is no h0)
    // slack between a publication and the phase that asks for it, in phases: nrs - 1 without look-ahead,
    // nrs - 2 with it, nrs - 3 with the deferred epilogue on top; a hand-off takes ~1.5 phases, so the deferred
    // form needs 5 phases (measured per step at 20 steps, U = 512: 4 phases deferred 16.8 us, with plain
    // look-ahead 8.9 us; 5 phases deferred 9.1 us -- tools/lstm_persist_rows.py)
    const bool defer = nrs >= 5;

    if (wave < 4) {
        // ---------------- MFMA waves ----------------
        f32x4 bv[CPW][2];
        const f32x4* Bf = reinterpret_cast<const f32x4*>(a.Wf);
        if (a.direct && !a.packed) {
            // straight from the row-major Wh: the element the packed image would hold at this index (four 4-byte loads
            // per register instead of one 16-byte load, once per launch -- and no pack pass over 4 MB in front of it)
#pragma unroll
            for (int c = 0; c < CPW; ++c)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float4 w = d2p_pack_w_fwd_elem(U, a.wh_raw, (((long)ct * KC + wave * CPW + c) * 2 + s) * 64 + lane);
                    bv[c][s] = f32x4{w.x, w.y, w.z, w.w};
                }
        } else {
#pragma unroll
            for (int c = 0; c < CPW; ++c)
#pragma unroll
                for (int s = 0; s < 2; ++s) bv[c][s] = Bf[(((long)ct * KC + wave * CPW + c) * 2 + s) * 64 + lane];
        }
        const __amdgpu_buffer_rsrc_t hres = ps_rsrc(a.hfrag, 2u * a.hfrag_bytes);
        const __amdgpu_buffer_rsrc_t hres0 = ps_rsrc(a.h0_raw ? (const void*)a.h0_raw : (const void*)a.hfrag, a.h0_bytes);
        // byte offset of this lane's float4 in block (rs = 0, kc = wave*CPW)
        const int lane_off = (wave * CPW * 64 + lane) * 16;
        // ... and in the row-major initial state: row lane&15 of the phase, k = (wave*CPW + c)*16 + 4*(lane>>4)
        const int rm_off = ((lane & 15) * U + wave * CPW * 16 + 4 * (lane >> 4)) * 4;
        // the 2*CPW producers (column tiles) whose units this wave's K slice covers
        const unsigned* fl = fbase + 2 * CPW * wave + (lane & (2 * CPW - 1));
        PsFwdEpi e;
        e.rr = wave * 4 + ((lane >> 3) & 3);
        e.un = lane & 7;
        e.u = ct * 8 + e.un;
        e.lane_on = lane < 32;
        e.P = P; e.stc = stc; e.sth = sth; e.stl = stl; e.stage = stage; e.ring = ring; e.spare = spare + lane;
        for (int p = 0; p < nrs; ++p) {          // this wave's cells: initial state, row lengths
            const int row = (rs0 + p) * 16 + e.rr;
            float c = 0.f, h = 0.f;
            int len = 0x7fffffff;
            if (row < a.M) {
                if (a.c0) c = a.c0[(long)row * U + e.u];
                if (a.h0) h = a.h0[(long)row * U + e.u];
                if (a.lens) len = a.lens[row];
            }
            if (e.lane_on) {
                stc[(p * 16 + e.rr) * 8 + e.un] = c;
                sth[(p * 16 + e.rr) * 8 + e.un] = h;
                stl[(p * 16 + e.rr) * 8 + e.un] = len;
            }
        }
        if (defer) ps_fwd_mfma_wave<CPW, 2>(a, e, bv, hres, hres0, fl, nct, rs0, nrs, nticks, lane_off, rm_off, wave, lane);
        // look-ahead asks for the NEXT phase's rows at the start of a phase: with two phases those were
        // published by the phase just finished, so the full hand-off latency sat in front of every phase
        // (3.8 us per phase); 1- and 2-phase domains poll for their own rows instead (3 phases: look-ahead
        // measured better, 8.9 against 13.1 us per step for a 3,2,2,2-phase split)
        else if (nrs >= 3) ps_fwd_mfma_wave<CPW, 1>(a, e, bv, hres, hres0, fl, nct, rs0, nrs, nticks, lane_off, rm_off, wave, lane);
        else ps_fwd_mfma_wave<CPW, 0>(a, e, bv, hres, hres0, fl, nct, rs0, nrs, nticks, lane_off, rm_off, wave, lane);
        if (e.lane_on)
            for (int q = 0; q < nrs; ++q) {
                const int row = (rs0 + q) * 16 + e.rr;
                if (row < a.M) {
                    if (a.h_final) a.h_final[(long)row * U + e.u] = sth[(q * 16 + e.rr) * 8 + e.un];
                    if (a.c_final) a.c_final[(long)row * U + e.u] = stc[(q * 16 + e.rr) * 8 + e.un];
                }
            }
    } else {
        // ---------------- publish + prefetch wave ----------------
        // prefetch: 2 DMA instructions per tick; quad = i*64 + lane = (row, gate, half): 16 bytes = 4 units
        const float* src[2];
        int rowoff[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int quad = i * 64 + lane;
            rowoff[i] = quad >> 3;
            src[i] = a.z + (long)((quad >> 1) & 3) * U + ct * 8 + (quad & 1) * 4;
        }
        PsTick kp = {0, 0};        // next tick to prefetch
        int pslot = 0;
        auto issue = [&]() {
            const PsTick q = kp;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = min((rs0 + q.p) * 16 + rowoff[i], a.M - 1);
                ps_dma16(src[i] + (long)q.t * a.zts + (long)row * a.zrs, ring + pslot * PS_FWD_SLOT + i * 256);
            }
            // past the last tick the same rows are fetched again: the counted waits below stay exact
            if (q.p + 1 < nrs || q.t + 1 < a.T) kp.next(nrs);
            if (++pslot == PS_PF_R) pslot = 0;
        };
        for (int d = 0; d < PS_PF_D; ++d) issue();
        ps_wait_vmcnt<(PS_PF_D - 1) * 2>();         // tick 0's inputs have landed
        const __amdgpu_buffer_rsrc_t hres = ps_rsrc(a.hfrag, 2u * a.hfrag_bytes);
        PsTrace tr;
        tr.buf = (blockIdx.x == a.trace_block) ? a.trace : nullptr;
        PsTick k = {0, 0};
        for (int n = 0; n < nticks; ++n) {
            tr.stamp(0);
            ps_barrier();          // A (deferred form: the only barrier of the tick)
            if (!defer) ps_barrier();          // B: the phase's new h rows are staged
            tr.stamp(1);
            // deferred form: tick n-1's rows were staged during tick n and are published now
            const bool pub = !defer || n > 0;
            const int par = defer ? ((n - 1) & 1) : 0;
            const f32x4 hv = *reinterpret_cast<const f32x4*>(stage + par * 128 + (lane & 31) * 4);
            const int row = (rs0 + k.p) * 16 + (lane & 15);
            const int off = (int)(((k.t + 1) & 1) * a.hfrag_bytes) +
                            (int)(d2p_frag_off(row, ct * 8 + ((lane >> 4) & 1) * 4, U >> 4) * 4);
            asm volatile("" ::: "memory");
            if (lane < 32 && pub) ps_st_sc1(hres, off, hv[0], hv[1], hv[2], hv[3]);
            tr.stamp(2);
            ps_wait_vmcnt<0>();                      // the store is out (and the loads of the last tick have landed)
            tr.stamp(3);
            if (lane == 0 && pub) ps_st_flag(fbase + k.p * nct + ct, a.epoch + (unsigned)(k.t + 1));
            asm volatile("" ::: "memory");
            issue();                                 // tick n + D into a free ring slot, off the hand-off path
            tr.flush(1, n, lane);
            if (pub) k.next(nrs);
        }
        ps_wait_vmcnt<0>();
    }
}

// =============================================================================================
// Forward, wide column tiles (round 4): a workgroup owns [16 units x 4 gates] = 64 gate columns
// =============================================================================================
// The kernel above gives a workgroup 8 units (32 gate columns): 64 column tiles, so 256 CUs hold 4 row domains and a
// phase is 64 MFMAs per wave (2048 clocks) beside ~1450 clocks of everything else.  Here a phase is 128 MFMAs (4096
// clocks) beside about the same non-MFMA work -- the epilogue computes one cell per lane on all 64 lanes instead of 32
// (the narrow tile's lanes 32-63 are idle duplicates), the operand loads, the flag polls, the barrier and the
// publication are per phase, not per column -- and 32 column tiles leave room for 8 row domains:
//   * up to THREE sequences per launch (the three decoders: 3 + 3 + 2 domains), as the backward kernel takes them;
//   * length-sorted launches for the encoders (rowmap / rs_start / tdom as in the backward kernel): a domain holds
//     rows of similar length and runs only its longest row's steps; what the skipped steps would have written
//     (zeros in hout, the carried cell state in cs) is filled in at the end, so every array is bit-identical to the
//     unsorted launch;
//   * with 8 domains of 32 tiles the block -> tile map puts ONE domain on ONE XCD when workgroup b lands on XCD b % 8
//     (observed, not promised).  Every workgroup publishes its HW_REG_XCC_ID with its first rows; a domain whose 32
//     workgroups all read the same id exchanges through that XCD's L2 from then on (plain stores: the lines stay in
//     L2, the consumers' sc1 loads hit there) instead of writing through to memory -- decided per domain at run
//     time, correct under any placement.
// Same arithmetic and summation order as the narrow kernel and lstm_step.hip (K split four ways over the waves, chunks
// and k in ascending order inside a wave, partial tiles added in wave order): outputs are bit-identical.
// Direct launches only (the caller's flag buffer and epoch); everything else goes to the kernel above.
#ifndef PSW_DESC_PREFETCH
#define PSW_DESC_PREFETCH 1               // the z prefetch through a buffer descriptor (0: 64-bit pointers; build-time A/B)
#endif
#ifndef PSW_DESC_STORES
#define PSW_DESC_STORES 1                 // the epilogue's global stores through buffer descriptors (0: 64-bit pointers)
#endif
#define PSW_PLD 80                        // partial-tile row stride (floats): 64 columns, rows rr / rr+1 on disjoint banks
#define PSW_P_FLOATS (4 * 16 * PSW_PLD)   // one set of four partial tiles
#define PSW_SLOT 1024                     // floats per prefetch ring slot: 4 gates x 16 rows x 16 units
#define PSW_CELLS 256                     // cells of a phase: 16 rows x 16 units
#define PSW_XCC_WORDS 256                 // one word per (domain, column tile) behind the tickets: the workgroup's XCC id + 1

__device__ unsigned g_psw_local_wgs;     // workgroups so far that found their row domain on one XCD (a statistic)
struct PsFwdWArgs {
    int M, U, T, total_rs, RT, has_h0;
    int dom0;               // this sequence's row domains are domains [dom0, dom0 + RT) of the launch
    float* hfrag;           // 2 ping-pong buffers of Mp*U floats, fragment-major
    unsigned hfrag_bytes;   // bytes of ONE buffer
    float* z; long zrs, zts;
    const float* c0; const int* lens;
    float* hout; float* cs; float* h_final; float* c_final;
    unsigned* flags;        // [RT][PS_NRS_MAX][U/16] on epochs, then tickets, then PSW_XCC_WORDS placement words
    float* dump;
    unsigned* err;
    unsigned long long* trace; int trace_block;
    const float* wh_raw; const float* h0_raw; unsigned h0_bytes; unsigned epoch;
    int poll;
    int la_from, defer_from;     // phases per domain from which a domain looks ahead / runs the deferred form
    int la_q;                    // quarters of a phase's chain in front of the look-ahead request (2 or 3)
    int lds_nb;                  // 2 when some domain of the launch defers (double-buffered partial tiles / staged rows)
    int xcd_local;               // 1: domains found on one XCD exchange through its L2 (0: always write-through)
    // length-sorted launch (as PsBwdArgs)
    const int* rowmap;
    int sorted, Tfull;
    int rs_start[PS_RT_TAB + 1], tdom[PS_RT_TAB];
};

template <int CPW>
__device__ __forceinline__ void psw_chain(const f32x4 (&av)[CPW], const f32x4 (&bv)[CPW][4], f32x4 (&acc)[4], int c0 = 0,
                                          int c1 = CPW) {
#pragma unroll
    for (int c = c0; c < c1; ++c)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][jj], bv[c][g][jj], acc[g], 0, 0, 0);
}

struct PswEpi {             // per-lane constants of an MFMA wave's share of the epilogue: one cell per lane
    int rr, un, u;          // row within the phase, unit within the tile, global unit
    float *P, *stc, *sth, *stage, *ring;
    int* stl;
    // the global stores of the epilogue go through buffer descriptors: per-lane byte offset (row, unit) in a VGPR, the
    // step's and the gate's offset in an SGPR, masked lanes past the descriptor's range (dropped by the bounds check)
    // -- a handful of VALU instructions instead of ~45 for six 64-bit addresses (a VALU instruction beside an fp32
    // MFMA chain is paid in full: measured ~8 clocks each, tools/trace_lstm_wide.py)
    __amdgpu_buffer_rsrc_t zres, cres, hres_out;
};
struct PswEpiIn {
    float zin[4], part[4][4], cp, hp;
    int lw;                 // length | row of the caller's arrays << 16
};
#define PSW_OOB 0x7fffff00   // a byte offset past every descriptor (sizes are checked on the host: < 2^31 bytes)
__device__ __forceinline__ void psw_epilogue_load(const PswEpi& e, int p, int slot, int par, PswEpiIn& in) {
    const int sidx = (p * 16 + e.rr) * 16 + e.un;
    const float* zs = e.ring + slot * PSW_SLOT + e.rr * 16 + e.un;
#pragma unroll
    for (int g = 0; g < 4; ++g) in.zin[g] = zs[g * 256];
    const float* Pb = e.P + par * PSW_P_FLOATS + e.rr * PSW_PLD + e.un;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int w = 0; w < 4; ++w) in.part[g][w] = Pb[w * 16 * PSW_PLD + g * 16];
    in.cp = e.stc[sidx];
    in.hp = e.sth[sidx];
    in.lw = e.stl[p * 16 + e.rr];
}
__device__ __forceinline__ void psw_epilogue_finish(const PsFwdWArgs& a, const PswEpi& e, int rs0, int p, int t, int par,
                                                    const PswEpiIn& in) {
    const int U = a.U;
    const int vrow = (rs0 + p) * 16 + e.rr;
    const bool valid = vrow < a.M;
    const int prow = in.lw >> 16;
    const int sidx = (p * 16 + e.rr) * 16 + e.un;
    const bool active = t < (in.lw & 0xffff);
    float zz[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        zz[g] = in.zin[g];
#pragma unroll
        for (int w = 0; w < 4; ++w) zz[g] += in.part[g][w];
    }
    float cn, hn;
    lstm_cell_fwd(zz[0], zz[1], zz[2], zz[3], in.cp, cn, hn);
    const float c_out = active ? cn : in.cp;
    const float h_out = active ? hn : 0.f;        // emitted output: 0 past the row's length
    const float h_state = active ? hn : in.hp;    // (c, h) copy through
    e.stc[sidx] = c_out;
    e.sth[sidx] = h_state;
    // staged in fragment-major order: the publish wave's lane l = (quad l>>4, row l&15) reads a float4
    e.stage[par * PSW_CELLS + (((e.un >> 2) << 4) + e.rr) * 4 + (e.un & 3)] = h_state;
    if constexpr (!PSW_DESC_STORES) {
        const long o = (long)prow * U + e.u;
        const long zo = (long)t * a.zts + (long)prow * a.zrs + e.u;
        const long so = (long)t * a.M * U + o;
        const long dz = a.dump - a.z, dc = a.dump - a.cs, dh = a.dump - a.hout;    // wave-uniform
        float* zr = a.z + (valid ? zo : dz);
        const long zg = valid ? (long)U : 0L;
        const bool keep_ = !(active && ((t > 0) || a.has_h0));
#pragma unroll
        for (int g = 0; g < 4; ++g) zr[g * zg] = keep_ ? in.zin[g] : zz[g];
        a.cs[valid ? so : dc] = c_out;
        a.hout[valid ? so : dh] = h_out;
        return;
    }
    // unconditional stores (rows that keep their input projection write it back unchanged; masked lanes out of range)
    const int u4 = e.u * 4;
    const int vz = valid ? (int)__umul24((unsigned)prow, (unsigned)(a.zrs * 4)) + u4 : PSW_OOB;
    const int vo = valid ? (int)__umul24((unsigned)prow, (unsigned)(U * 4)) + u4 : PSW_OOB;
    const int sz = t * (int)(a.zts * 4), so = t * (a.M * U * 4);       // wave-uniform: scalar offsets of the step
    const bool keep = !(active && ((t > 0) || a.has_h0));
#pragma unroll
    for (int g = 0; g < 4; ++g)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, keep ? in.zin[g] : zz[g]), e.zres, vz, sz + g * U * 4, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, c_out), e.cres, vo, so, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, h_out), e.hres_out, vo, so, 0);
}
__device__ __forceinline__ void psw_write_partials(float* Pw, int lane, const f32x4 (&acc)[4]) {
    // C/D layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) Pw[((lane >> 4) * 4 + r) * PSW_PLD + g * 16 + (lane & 15)] = acc[g][r];
}

// where step t, phase p of a sequence reads its 16 rows: the fragment-major ping-pong buffer, or (t = 0) the row-major
// initial state through its own descriptor (zero-length without one); prow0 = this lane's row of the caller's arrays
__device__ __forceinline__ PsSrc psw_src(const PsFwdWArgs& a, __amdgpu_buffer_rsrc_t hres, __amdgpu_buffer_rsrc_t hres0, int t,
                                         int rs, int KC, int lane_off, int prow0, int k_off) {
    const bool first = t == 0;
    PsSrc s;
    s.r = first ? hres0 : hres;
    s.off = first ? (prow0 * a.U + k_off) * 4 : (int)((t & 1) * a.hfrag_bytes) + rs * KC * 1024 + lane_off;
    s.cs = first ? 64 : 1024;
    return s;
}

// One phase, epilogue right behind its own product.  MODE 0 (single-phase domains): the phase's own rows behind a
// blocking poll, nothing fetched ahead (the next tick's rows come out of this tick's epilogue).  MODE 1 (look-ahead): the
// NEXT phase's rows are requested three quarters into this phase's chain -- their flag is read at the start of the
// chain and looked at again if it was not up yet (a 2-phase domain's rows were published by the tick before this one:
// that much slack their hand-off needs) -- and the loads go between the MFMAs of the last quarter (a VMEM issue beside
// an MFMA costs ~15 clocks, behind the chain ~60); they land during the partial-tile exchange and the gate math.
template <int CPW, int MODE>     // MODE 3: as 1, with the request in the middle of the chain
__device__ __forceinline__ void psw_tick(f32x4 (&cur)[CPW], f32x4 (&nxt)[CPW], const f32x4 (&bv)[CPW][4],
                                         const PsFwdWArgs& a, const PswEpi& e, PsTick k0, const unsigned* fl_cur,
                                         const PsSrc& cur_src, const unsigned* fl1, unsigned need1, const PsSrc& src1,
                                         int wave, int rs0, int slot, int lane, PsTrace& tr) {
    constexpr int C3 = CPW >= 4 ? (MODE == 3 ? CPW / 2 : (3 * CPW) / 4) : CPW;
    tr.stamp(0);
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        ps_wait_flags(fl_cur, ps_need(a.epoch, k0.t), ps_ld_flag(fl_cur), a.err, 3, a.flags, ps_poll_single(a.poll));
#pragma unroll
        for (int c = 0; c < CPW; ++c) cur[c] = ps_ld_src(cur_src, c);
        tr.stamp(1);
        psw_chain<CPW>(cur, bv, acc);
    } else {
        const unsigned fvn = ps_ld_flag(fl1);
        psw_chain<CPW>(cur, bv, acc, 0, C3);
        __builtin_amdgcn_sched_barrier(0);
        ps_wait_flags<true>(fl1, need1, fvn, a.err, 1, a.flags, a.poll);
        tr.stamp(1);
#pragma unroll
        for (int c = 0; c < CPW; ++c) nxt[c] = ps_ld_src(src1, c);
        if (C3 < CPW) {
            psw_chain<CPW>(cur, bv, acc, C3, CPW);
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, (16 * (CPW - C3)) / CPW, 0);      // MFMAs
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                            // one operand load
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    tr.stamp(2);
    psw_write_partials(e.P + wave * 16 * PSW_PLD, lane, acc);
    ps_barrier();           // A: all four partial tiles of this phase are in LDS
    tr.stamp(3);
    PswEpiIn in;
    psw_epilogue_load(e, k0.p, slot, 0, in);
    psw_epilogue_finish(a, e, rs0, k0.p, k0.t, 0, in);
    ps_barrier();           // B: new rows staged, P and the ring slot free again
    tr.stamp(4);
}

// Deferred form (domains with >= defer_from phases): the gate math of the PREVIOUS phase inside this phase's MFMA chain
// (one basic block: LDS reads and the next phase's operand loads in the first quarter, a quarter for them to land,
// arithmetic and stores spread over the second half), one barrier per phase, rows published one phase later.
template <int CPW>
__device__ __forceinline__ void psw_tick_defer(const f32x4 (&cur)[CPW], f32x4 (&nxt)[CPW], const f32x4 (&bv)[CPW][4],
                                               const PsFwdWArgs& a, const PswEpi& e, PsTick kprev, int slot_prev,
                                               const unsigned* fl1, unsigned need1, const PsSrc& src1,
                                               const unsigned* fl2, unsigned& fv, int wave, int rs0, int par, int lane,
                                               PsTrace& tr) {
    tr.stamp(0);
    ps_wait_flags(fl1, need1, fv, a.err, 1, a.flags, a.poll);
    tr.stamp(1);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int C1 = CPW >= 4 ? CPW / 4 : 0, C2 = CPW >= 4 ? CPW / 2 : 0;
#pragma unroll
    for (int c = 0; c < CPW; ++c) nxt[c] = ps_ld_src(src1, c);
    fv = ps_ld_flag(fl2);
    PswEpiIn in;
    psw_epilogue_load(e, kprev.p, slot_prev, par ^ 1, in);
    psw_chain<CPW>(cur, bv, acc, 0, C1);
#pragma unroll
    for (int i = 0; i < 8 * C1; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // two MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one operand load
        __builtin_amdgcn_sched_group_barrier(0x186, 3, 0);      // three of VALU / SALU / LDS read
    }
    __builtin_amdgcn_sched_barrier(0);
    psw_chain<CPW>(cur, bv, acc, C1, C2);
    __builtin_amdgcn_sched_barrier(0);
    psw_chain<CPW>(cur, bv, acc, C2, CPW);
    psw_epilogue_finish(a, e, rs0, kprev.p, kprev.t, par ^ 1, in);
#pragma unroll
    for (int i = 0; i < 8 * (CPW - C2); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // two MFMAs
        __builtin_amdgcn_sched_group_barrier(0x3d6, 4, 0);      // four of VALU / SALU / LDS / VMEM write
    }
    __builtin_amdgcn_sched_barrier(0);
    tr.stamp(2);
    psw_write_partials(e.P + par * PSW_P_FLOATS + wave * 16 * PSW_PLD, lane, acc);
    ps_barrier();           // partial tiles of this phase and the staged rows of the previous one are in LDS
    tr.stamp(3);
    tr.stamp(4);
}

// All ticks of one MFMA wave; two ticks per iteration with the two operand register sets swapping roles.
template <int CPW, int MODE>
__device__ __forceinline__ void psw_mfma_wave(const PsFwdWArgs& a, const PswEpi& e, const f32x4 (&bv)[CPW][4],
                                              __amdgpu_buffer_rsrc_t hres, __amdgpu_buffer_rsrc_t hres0,
                                              const unsigned* fl, int nnt, int rs0, int nrs, int nticks, int lane_off,
                                              int k_off, int wave, int lane) {
    constexpr int KC = 4 * CPW;
    PsTrace tr;
    tr.buf = (blockIdx.x == a.trace_block) ? a.trace : nullptr;
    f32x4 a0[CPW], a1[CPW];
    const int* rowtab = e.stl + (lane & 15);          // this lane's operand row of phase p: rowtab[p*16] >> 16
    {
        const PsSrc s0 = psw_src(a, hres, hres0, 0, rs0, KC, lane_off, rowtab[0] >> 16, k_off);
#pragma unroll
        for (int c = 0; c < CPW; ++c) a0[c] = ps_ld_src(s0, c);
    }
    PsTick kp = {0, 0}, k0 = {0, 0}, k1 = {0, 0}, k2 = {0, 0};      // ticks n-1, n, n+1, n+2
    k1.next(nrs);
    k2.next(nrs); k2.next(nrs);
    unsigned fv = ps_ld_flag(fl + k1.p * nnt);
    int slot = 0, slot_prev = 0;
#define PSW_ONE_TICK(CUR, NXT, m)                                                                             \
    {                                                                                                         \
        const bool e1 = (m) + 1 >= nticks, e2 = (m) + 2 >= nticks;                                            \
        const PsTick q1 = e1 ? k0 : k1, q2 = e2 ? (e1 ? k0 : k1) : k2;                                        \
        const unsigned need1 = e1 ? 0u : ps_need(a.epoch, q1.t);                                              \
        int pr0 = 0, pr1 = 0;          /* the row-major initial state is read in step 0 only */              \
        if (k0.t == 0) pr0 = rowtab[k0.p * 16] >> 16;                                                         \
        if (q1.t == 0) pr1 = rowtab[q1.p * 16] >> 16;                                                         \
        const PsSrc cur_src = psw_src(a, hres, hres0, k0.t, rs0 + k0.p, KC, lane_off, pr0, k_off);            \
        const PsSrc src1 = psw_src(a, hres, hres0, q1.t, rs0 + q1.p, KC, lane_off, pr1, k_off);               \
        if (MODE == 2)                                                                                        \
            psw_tick_defer<CPW>(CUR, NXT, bv, a, e, kp, slot_prev, fl + q1.p * nnt, need1, src1, fl + q2.p * nnt, \
                                fv, wave, rs0, (m) & 1, lane, tr);                                            \
        else                                                                                                  \
            psw_tick<CPW, MODE>(CUR, NXT, bv, a, e, k0, fl + k0.p * nnt, cur_src, fl + q1.p * nnt, need1, src1, \
                                wave, rs0, slot, lane, tr);                                                   \
        if (wave == 0) tr.flush(0, (m), lane);                                                                \
        kp = k0; k0 = k1; k1 = k2; k2.next(nrs);                                                              \
        slot_prev = slot;                                                                                     \
        if (++slot == PS_PF_R) slot = 0;                                                                      \
    }
    int n = 0;
    if (MODE == 2) {
        // first tick: nothing to finish yet (plain product, one barrier)
        const unsigned need1 = ps_need(a.epoch, k1.t);
        const PsSrc src1 = psw_src(a, hres, hres0, k1.t, rs0 + k1.p, KC, lane_off, rowtab[k1.p * 16] >> 16, k_off);
        ps_wait_flags(fl + k1.p * nnt, need1, fv, a.err, 1, a.flags, a.poll);
#pragma unroll
        for (int c = 0; c < CPW; ++c) a1[c] = ps_ld_src(src1, c);
        fv = ps_ld_flag(fl + k2.p * nnt);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        psw_chain<CPW>(a0, bv, acc);
        psw_write_partials(e.P + wave * 16 * PSW_PLD, lane, acc);
        ps_barrier();
        kp = k0; k0 = k1; k1 = k2; k2.next(nrs);
        slot_prev = slot;
        ++slot;
        n = 1;
#pragma unroll 1
        for (; n + 1 < nticks; n += 2) {
            PSW_ONE_TICK(a1, a0, n)
            PSW_ONE_TICK(a0, a1, n + 1)
        }
        if (n < nticks) PSW_ONE_TICK(a1, a0, n)
        // the last phase's gate math (kp is the last tick now)
        PswEpiIn in;
        psw_epilogue_load(e, kp.p, slot_prev, (nticks - 1) & 1, in);
        psw_epilogue_finish(a, e, rs0, kp.p, kp.t, (nticks - 1) & 1, in);
    } else {
#pragma unroll 1
        for (; n + 1 < nticks; n += 2) {
            PSW_ONE_TICK(a0, a1, n)
            PSW_ONE_TICK(a1, a0, n + 1)
        }
        if (n < nticks) PSW_ONE_TICK(a0, a1, n)
    }
#undef PSW_ONE_TICK
}

struct PsFwdWLaunch { PsFwdWArgs s[3]; };   // (indexed in the kernel-argument segment: selecting between three by-value
                                            //  argument blocks costs ~100 scalar selects per field group at kernel entry)
template <int CPW>   // U = 64 * CPW
__global__ void __launch_bounds__(PS_THREADS) lstm_persist_fwdw_kernel(PsFwdWLaunch L) {
    constexpr int KC = 4 * CPW;
    const int nnt = KC;                                        // column tiles: U / 16
    // block -> (domain of the launch, column tile): consecutive tiles of a domain on one XCD if block b runs on XCD
    // b % 8 -- with 8 domains of 32 tiles, one whole domain per XCD
    int dom, nt;
    ps_block_tile((int)blockIdx.x, (int)gridDim.x, nnt, dom, nt);
    const int seq = (dom >= L.s[0].RT + L.s[1].RT) ? 2 : ((dom >= L.s[0].RT) ? 1 : 0);
    const PsFwdWArgs& a = L.s[seq];
    const int rt = dom - a.dom0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nb = a.lds_nb;
    float* P = lds;                                            // [nb][4][16][PSW_PLD]
    float* stc = lds + nb * PSW_P_FLOATS;                      // [NRS][16 rows][16 units] cell state
    float* sth = stc + PS_NRS_MAX * PSW_CELLS;                 // [NRS][16][16] hidden state
    int* stl = reinterpret_cast<int*>(sth + PS_NRS_MAX * PSW_CELLS);    // [NRS][16] length | row << 16
    int* strow = stl + PS_NRS_MAX * 16;                        // [NRS][16] the prefetch wave's copy of the rows
    int* xcl = strow + PS_NRS_MAX * 16;                        // [4 + 1] the waves' verdicts on the domain's placement
    float* stage = reinterpret_cast<float*>(xcl + 16);         // [nb][4 quads][16 rows][4]: new h rows of a phase
    float* ring = stage + nb * PSW_CELLS;                      // [PS_PF_R][4 gates][16 rows][16 units]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int U = a.U;
    int rs0, nrs, T = a.T;
    ps_rt_range(rt, a.total_rs, a.RT, rs0, nrs);
    if (a.sorted) {
        int r0 = 0, r1 = 0, td = 1;
#pragma unroll
        for (int i = 0; i < PS_RT_TAB; ++i) {
            r0 = (i == rt) ? a.rs_start[i] : r0;
            r1 = (i == rt) ? a.rs_start[i + 1] : r1;
            td = (i == rt) ? a.tdom[i] : td;
        }
        rs0 = r0; nrs = r1 - r0;
        T = td;
    }
    const int nticks = nrs * T;
    unsigned* fbase = a.flags + (long)rt * PS_NRS_MAX * nnt;
    unsigned* xccw = a.flags + PS_FLAG_WORDS + PS_TICKET_WORDS + rt * nnt;     // this domain's placement words
    const bool defer = nrs >= a.defer_from && a.lds_nb == 2;
    const unsigned my_xcc = (unsigned)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID[3:0]
    PS_STAMP_ENTRY((unsigned long long)nrs | ((unsigned long long)T << 8) | ((unsigned long long)dom << 16) |
                                     ((unsigned long long)nt << 24) | ((unsigned long long)my_xcc << 32) | ((unsigned long long)seq << 40))

    if (wave < 4) {
        // ---------------- MFMA waves ----------------
        f32x4 bv[CPW][4];
        {
            // straight from the row-major Wh [U, 4U]: k = kc*16 + 4*(lane>>4) + jj, column g*U + nt*16 + (lane&15)
            const long ld = 4L * U;
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int k = (wave * CPW + c) * 16 + 4 * (lane >> 4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float* w = a.wh_raw + (long)k * ld + (long)g * U + nt * 16 + (lane & 15);
                    bv[c][g] = f32x4{w[0], w[ld], w[2 * ld], w[3 * ld]};
                }
            }
        }
        const __amdgpu_buffer_rsrc_t hres = ps_rsrc(a.hfrag, 2u * a.hfrag_bytes);
        const __amdgpu_buffer_rsrc_t hres0 = ps_rsrc(a.h0_raw ? (const void*)a.h0_raw : (const void*)a.hfrag, a.h0_bytes);
        const int lane_off = (wave * CPW * 64 + lane) * 16;
        const int k_off = wave * CPW * 16 + 4 * (lane >> 4);
        // the CPW producers (column tiles) whose units this wave's K slice covers
        const unsigned* fl = fbase + CPW * wave + (lane & (CPW - 1));
        PswEpi e;
        e.rr = wave * 4 + (lane >> 4);
        e.un = lane & 15;
        e.u = nt * 16 + e.un;
        e.P = P; e.stc = stc; e.sth = sth; e.stl = stl; e.stage = stage; e.ring = ring;
        e.zres = ps_rsrc(a.z, (unsigned)(((size_t)(a.Tfull - 1) * a.zts + (size_t)(a.M - 1) * a.zrs + 4 * (size_t)U) * sizeof(float)));
        e.cres = ps_rsrc(a.cs, (unsigned)((size_t)a.Tfull * a.M * U * sizeof(float)));
        e.hres_out = ps_rsrc(a.hout, (unsigned)((size_t)a.Tfull * a.M * U * sizeof(float)));
        for (int p = 0; p < nrs; ++p) {          // this wave's cells: initial state, row lengths, rows
            const int vrow = (rs0 + p) * 16 + e.rr;
            int row = min(vrow, a.M - 1);
            if (a.rowmap && vrow < a.M) row = a.rowmap[vrow];
            float c = 0.f, h = 0.f;
            int len = 0xffff;
            if (vrow < a.M) {
                if (a.c0) c = a.c0[(long)row * U + e.u];
                if (a.h0_raw && a.h0_bytes) h = a.h0_raw[(long)row * U + e.u];
                if (a.lens) len = min(max(a.lens[row], 0), 0xffff);
            }
            stc[(p * 16 + e.rr) * 16 + e.un] = c;
            sth[(p * 16 + e.rr) * 16 + e.un] = h;
            if (e.un == 0) stl[p * 16 + e.rr] = len | (row << 16);
        }
        ps_barrier();                             // (stl rows of the other waves: the operand rows of step 0)
        PS_STAMP_W0(1)
        if (defer) psw_mfma_wave<CPW, 2>(a, e, bv, hres, hres0, fl, nnt, rs0, nrs, nticks, lane_off, k_off, wave, lane);
        else if (nrs >= a.la_from && a.la_q == 2) psw_mfma_wave<CPW, 3>(a, e, bv, hres, hres0, fl, nnt, rs0, nrs, nticks, lane_off, k_off, wave, lane);
        else if (nrs >= a.la_from) psw_mfma_wave<CPW, 1>(a, e, bv, hres, hres0, fl, nnt, rs0, nrs, nticks, lane_off, k_off, wave, lane);
        else psw_mfma_wave<CPW, 0>(a, e, bv, hres, hres0, fl, nnt, rs0, nrs, nticks, lane_off, k_off, wave, lane);
        PS_STAMP_W0(3)
        for (int q = 0; q < nrs; ++q) {
            const int vrow = (rs0 + q) * 16 + e.rr;
            const int row = stl[q * 16 + e.rr] >> 16;
            if (vrow < a.M) {
                const float hs = sth[(q * 16 + e.rr) * 16 + e.un], cc = stc[(q * 16 + e.rr) * 16 + e.un];
                if (a.h_final) a.h_final[(long)row * U + e.u] = hs;
                if (a.c_final) a.c_final[(long)row * U + e.u] = cc;
                // the steps this domain did not run: every row is past its length there -- zero output, carried cell state
                for (int t = T; t < a.Tfull; ++t) {
                    a.hout[(long)t * a.M * U + (long)row * U + e.u] = 0.f;
                    a.cs[(long)t * a.M * U + (long)row * U + e.u] = cc;
                }
            }
        }
        PS_STAMP_EXIT_W0
    } else {
        // ---------------- publish + prefetch wave ----------------
        // prefetch: 4 DMA instructions per tick (one per gate): lane = (row r, quad q), 16 bytes = 4 units
        const int r = lane >> 2, q4 = lane & 3;
        for (int i = lane; i < nrs * 16; i += 64) {        // this wave's own table of the domain's rows
            const int vrow = rs0 * 16 + i;
            int row = min(vrow, a.M - 1);
            if (a.rowmap && vrow < a.M) row = a.rowmap[vrow];
            strow[i] = row;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        ps_barrier();                             // (pairs with the MFMA waves' barrier behind their tables)
        // (through a buffer descriptor: the lane's byte offset (row, quad) in one VGPR, step and gate in the scalar offset
        //  -- this wave shares a SIMD with an MFMA wave, whose chain pays for every VALU instruction issued here)
        const __amdgpu_buffer_rsrc_t zres =
            ps_rsrc(a.z, (unsigned)(((size_t)(a.Tfull - 1) * a.zts + (size_t)(a.M - 1) * a.zrs + 4 * (size_t)U) * sizeof(float)));
        const int vbase = (nt * 16 + q4 * 4) * 4, zrs4 = (int)(a.zrs * 4), zts4 = (int)(a.zts * 4);
        PsTick kp = {0, 0};        // next tick to prefetch
        int pslot = 0;
        const float* zsrc = a.z + nt * 16 + q4 * 4;
        auto issue = [&]() {
            const PsTick q = kp;
            float* dst = ring + pslot * PSW_SLOT;
            if constexpr (PSW_DESC_PREFETCH) {
                const int voff = (int)__umul24((unsigned)strow[q.p * 16 + r], (unsigned)zrs4) + vbase;
                const int soff = q.t * zts4;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(zres, (ps_lds_void*)(dst + g * 256), 16, voff, soff + g * U * 4, 0, 0);
            } else {
                const float* zr = zsrc + (long)q.t * a.zts + (long)strow[q.p * 16 + r] * a.zrs;
#pragma unroll
                for (int g = 0; g < 4; ++g) ps_dma16(zr + (long)g * U, dst + g * 256);
            }
            // past the last tick the same rows are fetched again: the counted waits below stay exact
            if (q.p + 1 < nrs || q.t + 1 < T) kp.next(nrs);
            if (++pslot == PS_PF_R) pslot = 0;
        };
        for (int d = 0; d < PS_PF_D; ++d) issue();
        ps_wait_vmcnt<(PS_PF_D - 1) * 4>();         // tick 0's inputs have landed
        PS_STAMP(1, 1)
        const __amdgpu_buffer_rsrc_t hres = ps_rsrc(a.hfrag, 2u * a.hfrag_bytes);
        PsTrace tr;
        tr.buf = (blockIdx.x == a.trace_block) ? a.trace : nullptr;
        // placement: every workgroup of the domain leaves its XCC id (+ 1 + epoch-free: the words are rewritten by every
        // launch before its first flag) in front of its first publication; after the first complete hand-off this wave
        // reads all of them -- all equal to its own: the domain exchanges through this XCD's L2 from then on
        if (lane == 0)
            __builtin_amdgcn_raw_buffer_store_b32((int)(a.epoch * 16u + my_xcc + 1u),
                                                  ps_rsrc(xccw, (unsigned)(nnt * sizeof(unsigned))), nt * 4, 0, PS_AUX_SC1);
        bool local = false;
        int checked = 0;
        PsTick k = {0, 0};
        for (int n = 0; n < nticks; ++n) {
            tr.stamp(0);
            ps_barrier();          // A (deferred form: the only barrier of the tick)
            if (!defer) ps_barrier();          // B: the phase's new h rows are staged
            tr.stamp(1);
            // deferred form: tick n-1's rows were staged during tick n and are published now
            const bool pub = !defer || n > 0;
            const int par = defer ? ((n - 1) & 1) : 0;
            const f32x4 hv = *reinterpret_cast<const f32x4*>(stage + par * PSW_CELLS + lane * 4);
            const int off = (int)(((k.t + 1) & 1) * a.hfrag_bytes) + (((rs0 + k.p) * KC + nt) * 64 + lane) * 16;
            asm volatile("" ::: "memory");
            if (pub) {
                if (local)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, hv), hres, off, 0, 0);
                else
                    ps_st_sc1(hres, off, hv[0], hv[1], hv[2], hv[3]);
            }
            tr.stamp(2);
            ps_wait_vmcnt<0>();                      // the store is out (and the loads of the last tick have landed)
            tr.stamp(3);
            if (lane == 0 && pub) ps_st_flag(fbase + k.p * nnt + nt, a.epoch + (unsigned)(k.t + 1));
            asm volatile("" ::: "memory");
            issue();                                 // tick n + D into a free ring slot, off the hand-off path
            if (a.xcd_local && !checked && pub && k.t >= 1) {
                // every workgroup of the domain has published step 0 (this workgroup consumed those rows before it
                // produced step 1's), so every placement word of this launch is visible
                const unsigned w = lane < nnt ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(
                                                    ps_rsrc(xccw, (unsigned)(nnt * sizeof(unsigned))), lane * 4, 0, PS_AUX_SC1)
                                              : a.epoch * 16u + my_xcc + 1u;
                local = __all((int)(w == a.epoch * 16u + my_xcc + 1u)) != 0;
                checked = 1;
                if (local && lane == 0) atomicAdd(&g_psw_local_wgs, 1u);
            }
            tr.flush(1, n, lane);
            if (pub) k.next(nrs);
        }
        PS_STAMP(1, 3)
        ps_wait_vmcnt<0>();
        PS_STAMP(1, 6)
    }
}

// =============================================================================================
// Backward:  dH_t = dz[t+1]·Wh^T, gate backward of step t -> dz[t]; last pass (t = -1): dh0
// =============================================================================================
struct PsBwdArgs {
    int M, U, T, total_rs, RT, want_dh0;
    int bid0, gsz;          // this sequence's workgroups are blocks [bid0, bid0 + gsz) of the launch
    const float4* Wb;       // packed Wh^T (lstm_step.hip backward layout)
    float* dzfrag;          // 2 ping-pong buffers of Mp*4U floats, fragment-major over K = 4U
    unsigned dzfrag_bytes;  // bytes of ONE buffer
    const float* z; long zrs, zts;
    const float* c0; const float* cs; const int* lens;
    const float* dhout; const float* dh_final; const float* dc_final;
    float* dz; float* dh0; float* dc0;
    unsigned* flags;        // [RT][PS_NRS_MAX][U/16]
    float* dump;
    unsigned* err;
    unsigned long long* trace; int trace_block;
    // bias gradient db[4U] = column sums of dz over all rows and steps (optional): every workgroup sums its own
    // 64 columns over its domain's rows, the last of a column tile's RT workgroups (a ticket) adds the RT partial
    // sums in domain order -- no separate column-sum pass over the 52 MB of dz
    float* db; float* dbpart; unsigned* dbtick;
    int defer_from;         // phases per domain from which the deferred form runs (a large value: never)
    int lds_nb;             // 1, or 2 when the launch may defer (double-buffered partial tiles / staged rows)
    // "direct" launches (no preparation launch, as PsFwdArgs): Wh^T fragments straight from the row-major Wh (one
    // 16-byte load each: the packed image is a permutation of its float4s), pass 0's all-zero operand from a
    // zero-length buffer descriptor, flags on epochs, bias-gradient tickets reset by their last arriver
    int direct;
    const float* wh_raw; unsigned epoch;
    int packed;             // as PsFwdArgs: Wb is the caller's packed Wh^T image
    // Length-sorted launches (direct ones only).  The kernel works on VIRTUAL rows -- the caller's rows ordered by
    // decreasing length, rowmap[v] = the row of the caller's arrays -- so that a domain (a contiguous range of
    // 16-row sub-tiles, rs_start[d] .. rs_start[d+1]) holds rows of similar length and runs only tdom[d] = its
    // longest row's passes; the host sizes the domains so that (phases x passes) is balanced.  Everything the
    // caller sees stays in ITS row order; dz of the passes a domain skips is zero-filled (what the masked passes
    // would have written).
    const int* rowmap;
    int sorted, Tfull;
    int rs_start[PS_RT_TAB + 1], tdom[PS_RT_TAB];
    int poll;               // bit 0: pipelined flag polls (ps_wait_flags) in domains that look ahead, bit 1: in single-phase ones too
    int use_desc;           // 1: stores and prefetches through buffer descriptors with 32-bit offsets (sizes checked on the host)
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

#define PS_BWD_NOP 8                     // prefetched operands per tick: z (4 gates), c_prev, c, dhout, dh_final
#define PS_BWD_SLOT (PS_BWD_NOP * 256)   // floats per ring slot: 16 rows x 16 units each

struct PsBwdEpi {
    int rr, un, u;
    float *P, *stdc, *stage, *ring;
    int* stl;
    // global stores through buffer descriptors (round 4, as the wide forward kernel): the lane's byte offset (row, unit)
    // in a VGPR, pass and gate in the scalar offset, masked lanes out of the descriptor's range
    __amdgpu_buffer_rsrc_t dzres, dh0res;
};

// Gate backward of this wave's 4 rows x 16 units of phase p, pass j (t = T-1-j; t = -1: dh0 pass), in
// two parts: everything that does not depend on the dz.Wh^T product (ring operands, the five
// non-linearities, the gate derivative factors) is computed inside the product's last MFMA stage; behind
// the combine barrier only the sum of the partial tiles, six multiplies and the stores remain.
struct PsBwdEpiPre {
    LstmCellBwdPre q;
    float dhx, dcv;
    bool cur_active;
    int prow;               // the row of the caller's arrays (clamped into range)
};
__device__ __forceinline__ void ps_bwd_epilogue_pre(const PsBwdArgs& a, const PsBwdEpi& e, int p, int j, int slot,
                                                    PsBwdEpiPre& pre) {
    const int t = a.T - 1 - j;
    const int lw = e.stl[p * 16 + e.rr];         // length | row << 16
    const int len = lw & 0xffff;
    pre.prow = lw >> 16;
    const bool next_active = (t + 1 < a.T) && (t + 1 < len);
    pre.cur_active = (t >= 0) && (t < len);
    const float* rs = e.ring + slot * PS_BWD_SLOT + e.rr * 16 + e.un;
    // dH_t = dz[t+1]·Wh^T (+ dh_final for rows that are not active at t+1) (+ dhout[t])
    float dhx = (!next_active && a.dh_final) ? rs[7 * 256] : 0.f;
    dhx += (pre.cur_active && a.dhout) ? rs[6 * 256] : 0.f;
    pre.dhx = dhx;
    const float cp = (t > 0 || a.c0) ? rs[4 * 256] : 0.f;
    pre.q = lstm_cell_bwd_pre(rs[0], rs[256], rs[512], rs[768], cp, rs[5 * 256]);
    pre.dcv = e.stdc[(p * 16 + e.rr) * 16 + e.un];
}
#define PS_BWD_P_FLOATS (4 * 16 * PS_PLD)
template <bool DESC>
__device__ __forceinline__ void ps_bwd_epilogue_post(const PsBwdArgs& a, const PsBwdEpi& e, int rs0, int p, int j,
                                                     const PsBwdEpiPre& pre, float (&dbacc)[4], int par = 0,
                                                     bool live = true) {
    const int U = a.U;
    const int t = a.T - 1 - j;
    const int row = (rs0 + p) * 16 + e.rr;
    const bool valid = row < a.M && live;        // (!live: the deferred form's dummy in front of tick 0 -- dump line only)
    float dH = pre.dhx;
    {   // the partial tiles are zero in pass 0 (the host zero-fills dz[T])
        const float* Pb = e.P + par * PS_BWD_P_FLOATS + e.rr * PS_PLD + e.un;
#pragma unroll
        for (int w = 0; w < 4; ++w) dH += Pb[w * 16 * PS_PLD];
    }
    const int sidx = (p * 16 + e.rr) * 16 + e.un;
    float g[4], dcn;
    lstm_cell_bwd_post(pre.q, dH, pre.dcv, g[0], g[1], g[2], g[3], dcn);
    e.stdc[sidx] = pre.cur_active ? dcn : pre.dcv;
    const bool act = pre.cur_active && valid;     // (rows past M repeat row M-1's operands: not part of any sum)
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
        g[gg] = act ? g[gg] : 0.f;
        dbacc[gg] += g[gg];
        // staged fragment-major per gate: the publish wave's lane l = (quad l>>4, row l&15) reads a float4
        e.stage[par * 1024 + gg * 256 + (((e.un >> 2) << 4) + e.rr) * 4 + (e.un & 3)] = g[gg];
    }
    // unconditional stores: dz[t] row-major (the dh0 pass writes dh0 with the first and dumps the rest);
    // addresses for a clamped row, only the final offset selected (see the forward epilogue)
    const bool fin = t < 0;
    const int rowc = pre.prow;
    if constexpr (DESC) {
        const int u4 = e.u * 4;
        const int vz = (valid && !fin) ? (int)__umul24((unsigned)rowc, (unsigned)(a.zrs * 4)) + u4 : PSW_OOB;
        const int vh = (valid && fin) ? (int)__umul24((unsigned)rowc, (unsigned)(U * 4)) + u4 : PSW_OOB;
        const int sz = max(t, 0) * (int)(a.zts * 4);
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, g[gg]), e.dzres, vz, sz + gg * U * 4, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, dH), e.dh0res, vh, 0, 0);
        return;
    }
    const long zo = (long)max(t, 0) * a.zts + (long)rowc * a.zrs + e.u;
    const long dd = a.dump - a.dz;
    float* dzr = a.dz + ((valid && !fin) ? zo : dd);
    const long zg = (valid && !fin) ? (long)U : 0L;
    float* q0 = (valid && fin) ? a.dh0 + (long)rowc * U + e.u : dzr;
    *q0 = fin ? dH : g[0];
#pragma unroll
    for (int gg = 1; gg < 4; ++gg) dzr[gg * zg] = g[gg];
}

// All ticks of one backward MFMA wave (LA: look-ahead, i.e. the domain has >= 2 phases).  Pass 0 has
// no product (there is no dz[T]): the host zero-fills that buffer and the chain runs unconditionally.
// DEFER (domains with >= 5 phases, as in the forward kernel): the product-DEPENDENT half of the previous phase's gate
// backward (partial-tile sum, six multiplies, state / staging writes, the dz stores) runs inside this phase's first
// MFMA stage instead of behind its own product -- one barrier per phase, the new rows published one phase later.
template <int CPW, bool DESC, bool LA, bool DEFER = false>
__device__ __forceinline__ void ps_bwd_mfma_wave(const PsBwdArgs& a, const PsBwdEpi& e, const f32x4 (&bw)[4 * CPW],
                                                 __amdgpu_buffer_rsrc_t dres, const unsigned* fl, int nnt, int rs0,
                                                 int nrs, int nticks, int lane_off, float* P, int wave, int lane,
                                                 float (&dbacc)[4]) {
    constexpr int CPWB = 4 * CPW;                // chunks per wave
    constexpr int NB = CPWB >= 32 ? 4 : 2;       // register stages per phase (even)
    constexpr int CB = CPWB / NB;                // chunks per stage
    constexpr int KC4 = 4 * CPWB;                // chunks over K = 4U
    PsTrace tr;
    tr.buf = (blockIdx.x == a.trace_block) ? a.trace : nullptr;
    f32x4 s0[CB], s1[CB];
    PsTick k0 = {0, 0}, k1 = {0, 0}, k2 = {0, 0};      // ticks n, n+1, n+2 (t = pass index j)
    k1.next(nrs);
    k2.next(nrs); k2.next(nrs);
    // pass j consumes dz[t+1] with t = T-1-j, i.e. the buffer written in pass j-1: (T-j) & 1
    // (pass 0 has no dz[T]: legacy launches read a zero-filled buffer, direct ones a zero-length descriptor)
    const __amdgpu_buffer_rsrc_t dres0 = a.direct ? ps_rsrc(a.dzfrag, 0u) : dres;
    {
        const int off0 = (int)((a.T & 1) * a.dzfrag_bytes) + rs0 * KC4 * 1024 + lane_off;
#pragma unroll
        for (int c = 0; c < CB; ++c) s0[c] = ps_ld_sc1(dres0, off0 + c * 1024);
    }
    unsigned fv = ps_ld_flag(fl + k1.p * nnt);
    int slot = 0;
    PsBwdEpiPre pre_prev;
    pre_prev.q = LstmCellBwdPre{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    pre_prev.dhx = 0.f; pre_prev.dcv = 0.f; pre_prev.cur_active = false; pre_prev.prow = 0;
    PsTick kprev = {0, 0};
#pragma unroll 1
    for (int n = 0; n < nticks; ++n) {
        const int par = DEFER ? (n & 1) : 0;
        float* Pw = P + par * PS_BWD_P_FLOATS + wave * 16 * PS_PLD;
        const bool e1 = n + 1 >= nticks, e2 = n + 2 >= nticks;
        const PsTick q1 = e1 ? k0 : k1, q2 = e2 ? (e1 ? k0 : k1) : k2;
        const unsigned need1 = e1 ? 0u : ps_need(a.epoch, q1.t);
        const int off = (int)(((a.T - k0.t) & 1) * a.dzfrag_bytes) + (rs0 + k0.p) * KC4 * 1024 + lane_off;
        const int off1 = (int)(((a.T - q1.t) & 1) * a.dzfrag_bytes) + (rs0 + q1.p) * KC4 * 1024 + lane_off;
        const __amdgpu_buffer_rsrc_t rk0 = k0.t == 0 ? dres0 : dres, rq1 = q1.t == 0 ? dres0 : dres;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        PsBwdEpiPre pre;
        tr.stamp(0);
        if (!LA) {
            ps_wait_flags(fl + k0.p * nnt, ps_need(a.epoch, k0.t), ps_ld_flag(fl + k0.p * nnt), a.err, 4, a.flags, ps_poll_single(a.poll));
#pragma unroll
            for (int c = 0; c < CB; ++c) s0[c] = ps_ld_sc1(rk0, off + c * 1024);
        }
#pragma unroll
        for (int st = 0; st < NB; ++st) {
            if (st == NB - 1) {
                // the next tick's first stage: behind its flags, read one tick ago
                tr.stamp(1);
                if (LA) ps_wait_flags(fl + q1.p * nnt, need1, fv, a.err, 2, a.flags, a.poll);
                tr.stamp(2);
            }
            const int noff = (st < NB - 1) ? off + (st + 1) * CB * 1024 : off1;
            const __amdgpu_buffer_rsrc_t rn = (st < NB - 1) ? rk0 : rq1;
            if ((st & 1) == 0) {
#pragma unroll
                for (int c = 0; c < CB; ++c) s1[c] = ps_ld_sc1(rn, noff + c * 1024);
                // deferred: the previous phase's product-dependent half, spread over this stage's MFMAs (tick 0: a
                // dummy on zeros whose stores go to the dump line -- no branch in the chain)
                // (phase index PS_NRS_MAX = a spare state slot nobody reads)
                if (DEFER && st == 0)
                    ps_bwd_epilogue_post<DESC>(a, e, rs0, n > 0 ? kprev.p : PS_NRS_MAX, kprev.t, pre_prev, dbacc, par ^ 1, n > 0);
                ps_bwd_stage<CB, CPWB>(s0, bw, st * CB, acc0, acc1);
            } else {
#pragma unroll
                for (int c = 0; c < CB; ++c) s0[c] = ps_ld_sc1(rn, noff + c * 1024);
                if (st == NB - 1) {
                    fv = ps_ld_flag(fl + q2.p * nnt);
                    ps_bwd_epilogue_pre(a, e, k0.p, k0.t, slot, pre);      // interleaved with this stage's MFMAs
                }
                ps_bwd_stage<CB, CPWB>(s1, bw, st * CB, acc0, acc1);
            }
            // the next stage's operand loads (~60 clocks of issue each) and, in the last stage, the
            // product-independent half of the gate math go between this stage's MFMAs
#pragma unroll
            for (int i = 0; i < 2 * CB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // two MFMAs
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one load
                __builtin_amdgcn_sched_group_barrier(0x186, 4, 0);      // VALU / SALU / LDS reads
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Pw[((lane >> 4) * 4 + r) * PS_PLD + (lane & 15)] = acc0[r] + acc1[r];
        ps_barrier();          // A: the four partial tiles are in LDS (deferred: and the previous phase's dz is staged)
        tr.stamp(3);
        if (!DEFER) {
            ps_bwd_epilogue_post<DESC>(a, e, rs0, k0.p, k0.t, pre, dbacc);
            ps_barrier();          // B: dz of the phase staged, P and the ring slot free again
        } else {
            pre_prev = pre;
            kprev = k0;
        }
        tr.stamp(4);
        if (wave == 0) tr.flush(0, n, lane);
        k0 = k1; k1 = k2; k2.next(nrs);
        if (++slot == PS_PF_R) slot = 0;
    }
    // the last phase's product-dependent half (its rows go nowhere: no publication)
    if (DEFER) ps_bwd_epilogue_post<DESC>(a, e, rs0, kprev.p, kprev.t, pre_prev, dbacc, (nticks - 1) & 1);
}

template <int CPW, bool DESC>   // U = 64 * CPW; each MFMA wave owns one gate's K range = 4*CPW chunks of 16; DESC: stores and
                                // prefetches through buffer descriptors (a compile-time choice: a run-time branch around the
                                // prefetch loads makes the compiler wait for them at the join)
__global__ void __launch_bounds__(PS_THREADS) lstm_persist_bwd_kernel(PsBwdArgs a0, PsBwdArgs a1, PsBwdArgs a2) {
    // up to three independent sequences per launch (the three decoders), on disjoint workgroups as in the forward kernel
    // (selected between three by-value argument blocks: indexing one array of them in the kernel-argument segment, as the
    //  wide forward kernel does, measured 35 us SLOWER per three-decoder launch here -- tools/lstm_bwd_desc_ab.py)
    PsBwdArgs a = ((int)blockIdx.x >= a0.gsz + a1.gsz) ? a2 : (((int)blockIdx.x >= a0.gsz) ? a1 : a0);
    constexpr int CPWB = 4 * CPW;                // chunks per wave
    constexpr int NB = CPWB >= 32 ? 4 : 2;       // register stages per phase (even)
    constexpr int CB = CPWB / NB;                // chunks per stage
    constexpr int KC4 = 4 * CPWB;                // chunks over K = 4U
    // dynamic LDS, sized by the host: the partial tiles and the staged rows are double-buffered only when some domain of
    // the launch runs the deferred form (a.lds_nb = 2: 77 KB; else 64 KB -- what this kernel leaves of the CU's 160 KB
    // decides how many GEMM workgroups of the side stream fit beside it)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nb = a.lds_nb;
    float* P = lds;                                            // [nb][4][16][PS_PLD] (16 columns used; tick parity)
    float* stdc = lds + nb * PS_BWD_P_FLOATS;                  // [NRS + 1 spare][16 rows][16 units] dC state
    int* stl = reinterpret_cast<int*>(stdc + (PS_NRS_MAX + 1) * 256);     // [NRS][16] row length | row << 16
    int* strow = stl + PS_NRS_MAX * 16;                               // [NRS][16] the prefetch wave's copy of the rows
    float* stage = stdc + (PS_NRS_MAX + 1) * 256 + 2 * PS_NRS_MAX * 16;  // [nb][4 gates][4 quads][16 rows][4]: dz of a phase
    float* ring = stage + nb * 1024;                           // [PS_PF_R][8][16 rows][16 units]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int U = a.U, nnt = U >> 4;
    int rt, nt;
    ps_block_tile((int)blockIdx.x - a.bid0, a.gsz, nnt, rt, nt);
    int rs0, nrs;
    ps_rt_range(rt, a.total_rs, a.RT, rs0, nrs);
    if (a.sorted) {
        // this domain's range and pass count (selects, not a dynamic index into the argument block)
        int r0 = 0, r1 = 0, td = 1;
#pragma unroll
        for (int i = 0; i < PS_RT_TAB; ++i) {
            r0 = (i == rt) ? a.rs_start[i] : r0;
            r1 = (i == rt) ? a.rs_start[i + 1] : r1;
            td = (i == rt) ? a.tdom[i] : td;
        }
        rs0 = r0; nrs = r1 - r0;
        a.T = td;
    }
    const int J = a.T + (a.want_dh0 ? 1 : 0);    // passes: t = T-1 .. 0 (, -1)
    const int nticks = nrs * J;
    unsigned* fbase = a.flags + (long)rt * PS_NRS_MAX * nnt;
    const bool defer = nrs >= a.defer_from && a.lds_nb == 2;
    PS_STAMP_ENTRY((unsigned long long)nrs | ((unsigned long long)J << 8) | ((unsigned long long)rt << 16) |
                                     ((unsigned long long)nt << 24) |
                                     ((unsigned long long)(((int)blockIdx.x >= a0.gsz + a1.gsz) ? 2 : (((int)blockIdx.x >= a0.gsz) ? 1 : 0)) << 40))

    if (wave < 4) {
        // ---------------- MFMA waves ----------------
        f32x4 bw[CPWB];
        const f32x4* Bf = reinterpret_cast<const f32x4*>(a.Wb) + ((long)nt * KC4 + wave * CPWB) * 64;
        if (a.direct && !a.packed) {
#pragma unroll
            for (int c = 0; c < CPWB; ++c) {
                const float4 w = d2p_pack_w_bwd_elem(U, a.wh_raw, ((long)nt * KC4 + wave * CPWB + c) * 64 + lane);
                bw[c] = f32x4{w.x, w.y, w.z, w.w};
            }
        } else {
#pragma unroll
            for (int c = 0; c < CPWB; ++c) bw[c] = Bf[(long)c * 64 + lane];
        }
        const __amdgpu_buffer_rsrc_t dres = ps_rsrc(a.dzfrag, 2u * a.dzfrag_bytes);
        const int lane_off = (wave * CPWB * 64 + lane) * 16;
        const unsigned* fl = fbase + (lane & (nnt - 1));      // every producer of the domain writes this gate
        PsBwdEpi e;
        e.rr = wave * 4 + (lane >> 4);
        e.un = lane & 15;
        e.u = nt * 16 + e.un;
        e.P = P; e.stdc = stdc; e.stl = stl; e.stage = stage; e.ring = ring;
        e.dzres = ps_rsrc(a.dz, (unsigned)(((size_t)(a.Tfull - 1) * a.zts + (size_t)(a.M - 1) * a.zrs + 4 * (size_t)U) * sizeof(float)));
        e.dh0res = ps_rsrc(a.dh0 ? a.dh0 : a.dump, a.dh0 ? (unsigned)((size_t)a.M * U * sizeof(float)) : 0u);
        for (int p = 0; p < nrs; ++p) {
            const int vrow = (rs0 + p) * 16 + e.rr;
            int row = min(vrow, a.M - 1);
            if (a.rowmap && vrow < a.M) row = a.rowmap[vrow];
            float d = 0.f;
            int len = a.T;
            if (vrow < a.M) {
                if (a.dc_final) d = a.dc_final[(long)row * U + e.u];
                if (a.lens) len = a.lens[row];
            }
            stdc[(p * 16 + e.rr) * 16 + e.un] = d;
            if (e.un == 0) stl[p * 16 + e.rr] = min(len, 0xffff) | (row << 16);
        }
        float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
        PS_STAMP_W0(1)
        // slack between a publication and the phase that asks for it: nrs - 2 phases with look-ahead, nrs - 3 with the
        // deferred epilogue on top; a hand-off takes ~1.5 phases (as in the forward kernel: deferred from 5 phases)
        if (defer) ps_bwd_mfma_wave<CPW, DESC, true, true>(a, e, bw, dres, fl, nnt, rs0, nrs, nticks, lane_off, P, wave, lane, dbacc);
        else if (nrs >= 2) ps_bwd_mfma_wave<CPW, DESC, true>(a, e, bw, dres, fl, nnt, rs0, nrs, nticks, lane_off, P, wave, lane, dbacc);
        else ps_bwd_mfma_wave<CPW, DESC, false>(a, e, bw, dres, fl, nnt, rs0, nrs, nticks, lane_off, P, wave, lane, dbacc);
        PS_STAMP_W0(3)
        if (a.db) {
            // this workgroup's 4 gates x 16 units: the four row lanes of a wave by two shuffles, the four waves
            // through LDS in wave order (P is free: the last tick's barrier B is behind every wave)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                float v = dbacc[gg];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (lane < 16) P[(wave * 4 + gg) * 16 + lane] = v;
            }
            ps_barrier();
            if (wave == 0) {
                const int gg = lane >> 4, un = lane & 15;
                const float v = ((P[(0 * 4 + gg) * 16 + un] + P[(1 * 4 + gg) * 16 + un]) + P[(2 * 4 + gg) * 16 + un]) +
                                P[(3 * 4 + gg) * 16 + un];
                const __amdgpu_buffer_rsrc_t pres = ps_rsrc(a.dbpart, (unsigned)(a.RT * nnt * 64 * sizeof(float)));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), pres, ((rt * nnt + nt) * 64 + lane) * 4, 0,
                                                      PS_AUX_SC1);
                ps_wait_vmcnt<0>();                                    // the write-through store is out
                unsigned ticket = 0;
                if (lane == 0) ticket = __hip_atomic_fetch_add(a.dbtick + nt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
                if (ticket == (unsigned)(a.RT - 1)) {                  // last of this column tile: fold, in domain order
                    if (lane == 0) __hip_atomic_store(a.dbtick + nt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    float sum = 0.f;
                    for (int r = 0; r < a.RT; ++r)
                        sum += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             pres, ((r * nnt + nt) * 64 + lane) * 4, 0, PS_AUX_SC1));
                    a.db[(long)gg * U + nt * 16 + un] = sum;
                }
            }
        }
        PS_STAMP_W0(4)
        if (a.dc0)
            for (int qq = 0; qq < nrs; ++qq) {
                const int vrow = (rs0 + qq) * 16 + e.rr;
                const int row = stl[qq * 16 + e.rr] >> 16;
                if (vrow < a.M) a.dc0[(long)row * U + e.u] = stdc[(qq * 16 + e.rr) * 16 + e.un];
            }
        if (a.sorted && a.T < a.Tfull)
            // the passes this domain did not run: every row is past its length there, dz = 0
            for (int qq = 0; qq < nrs; ++qq) {
                const int vrow = (rs0 + qq) * 16 + e.rr;
                const int row = stl[qq * 16 + e.rr] >> 16;
                if (vrow < a.M)
                    for (int t = a.T; t < a.Tfull; ++t) {
                        float* zr = a.dz + (long)t * a.zts + (long)row * a.zrs + e.u;
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg) zr[(long)gg * U] = 0.f;
                    }
            }
        PS_STAMP_EXIT_W0
    } else {
        // ---------------- publish + prefetch wave ----------------
        // prefetch: 8 DMA instructions per tick, lane = (row r, quad q): z gates i, j, f, o, c before the
        // step, c after it, dhout, dh_final.  Absent operands fetch a valid dummy (never read).
        const int r = lane >> 2, q = lane & 3;
        const int u = nt * 16 + q * 4;
        for (int i = lane; i < nrs * 16; i += 64) {        // this wave's own table of the domain's rows
            const int vrow = rs0 * 16 + i;
            int row = min(vrow, a.M - 1);
            if (a.rowmap && vrow < a.M) row = a.rowmap[vrow];
            strow[i] = row;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        PsTick kp = {0, 0};
        int pslot = 0;
        const __amdgpu_buffer_rsrc_t zres_b =
            ps_rsrc(a.z, (unsigned)(((size_t)(a.Tfull - 1) * a.zts + (size_t)(a.M - 1) * a.zrs + 4 * (size_t)U) * sizeof(float)));
        const unsigned mu_bytes = (unsigned)((size_t)a.M * U * sizeof(float));
        const __amdgpu_buffer_rsrc_t cres_b = ps_rsrc(a.cs, (unsigned)((size_t)a.Tfull * a.M * U * sizeof(float)));
        const __amdgpu_buffer_rsrc_t c0res_b = ps_rsrc(a.c0 ? a.c0 : a.cs, mu_bytes);
        const __amdgpu_buffer_rsrc_t dhres_b =
            ps_rsrc(a.dhout ? a.dhout : a.cs, (unsigned)((size_t)(a.dhout ? a.Tfull : 1) * a.M * U * sizeof(float)));
        const __amdgpu_buffer_rsrc_t dhfres_b = ps_rsrc(a.dh_final ? a.dh_final : a.cs, mu_bytes);
        const int zrs4 = (int)(a.zrs * 4), zts4 = (int)(a.zts * 4);
        auto issue_desc = [&]() {
            const PsTick kk = kp;
            const int t = max(a.T - 1 - kk.t, 0);                       // the dh0 pass fetches step 0 again (unused)
            const unsigned row = (unsigned)strow[kk.p * 16 + r];
            const int vz = (int)__umul24(row, (unsigned)zrs4) + u * 4, vo = (int)__umul24(row, (unsigned)(U * 4)) + u * 4;
            float* dst = ring + pslot * PS_BWD_SLOT;
            const int sz = t * zts4, sc = t * (int)mu_bytes;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(zres_b, (ps_lds_void*)(dst + g * 256), 16, vz, sz + g * U * 4, 0, 0);
            // c before the step: cs[t-1], or c0, or (no initial state) a valid dummy
            const bool first = t == 0;
            const __amdgpu_buffer_rsrc_t cpres = first ? c0res_b : cres_b;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(cpres, (ps_lds_void*)(dst + 4 * 256), 16, vo, first ? 0 : sc - (int)mu_bytes, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(cres_b, (ps_lds_void*)(dst + 5 * 256), 16, vo, sc, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dhres_b, (ps_lds_void*)(dst + 6 * 256), 16, vo, a.dhout ? sc : 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dhfres_b, (ps_lds_void*)(dst + 7 * 256), 16, vo, 0, 0, 0);
            if (kk.p + 1 < nrs || kk.t + 1 < J) kp.next(nrs);
            if (++pslot == PS_PF_R) pslot = 0;
        };
        auto issue_ptr = [&]() {
            const PsTick kk = kp;
            const int t = max(a.T - 1 - kk.t, 0);                       // the dh0 pass fetches step 0 again (unused)
            const int row = strow[kk.p * 16 + r];
            const long o = (long)row * U + u;
            float* dst = ring + pslot * PS_BWD_SLOT;
            const float* zr = a.z + (long)t * a.zts + (long)row * a.zrs + u;
#pragma unroll
            for (int g = 0; g < 4; ++g) ps_dma16(zr + (long)g * U, dst + g * 256);
            const float* cc = a.cs + (size_t)t * a.M * U + o;
            const float* cp = t > 0 ? a.cs + (size_t)(t - 1) * a.M * U + o : (a.c0 ? a.c0 + o : cc);
            ps_dma16(cp, dst + 4 * 256);
            ps_dma16(cc, dst + 5 * 256);
            ps_dma16(a.dhout ? a.dhout + (size_t)t * a.M * U + o : cc, dst + 6 * 256);
            ps_dma16(a.dh_final ? a.dh_final + o : cc, dst + 7 * 256);
            if (kk.p + 1 < nrs || kk.t + 1 < J) kp.next(nrs);
            if (++pslot == PS_PF_R) pslot = 0;
        };
        auto issue = [&]() {
            if constexpr (DESC) issue_desc();
            else issue_ptr();
        };
        for (int d = 0; d < PS_PF_D; ++d) issue();
        ps_wait_vmcnt<(PS_PF_D - 1) * PS_BWD_NOP>();
        PS_STAMP(1, 1)
        const __amdgpu_buffer_rsrc_t dres = ps_rsrc(a.dzfrag, 2u * a.dzfrag_bytes);
        const int KCx = U >> 2;
        PsTrace tr;
        tr.buf = (blockIdx.x == a.trace_block) ? a.trace : nullptr;
        PsTick k = {0, 0};
        for (int n = 0; n < nticks; ++n) {
            tr.stamp(0);
            ps_barrier();          // A (deferred form: the only barrier of the tick)
            if (!defer) ps_barrier();          // B: dz of the phase is staged
            tr.stamp(1);
            // deferred form: tick n-1's rows were staged during tick n and are published now
            const bool pub = !defer || n > 0;
            const int par = defer ? ((n - 1) & 1) : 0;
            const int t = a.T - 1 - k.t;
            const int row = (rs0 + k.p) * 16 + (lane & 15);
            const int fo = (int)((t & 1) * a.dzfrag_bytes);
            f32x4 gv[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) gv[gg] = *reinterpret_cast<const f32x4*>(stage + par * 1024 + gg * 256 + lane * 4);
            asm volatile("" ::: "memory");
            if (t >= 0 && pub) {
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
                    ps_st_sc1(dres, fo + (int)(d2p_frag_off(row, gg * U + nt * 16 + (lane >> 4) * 4, KCx) * 4), gv[gg][0],
                              gv[gg][1], gv[gg][2], gv[gg][3]);
            }
            tr.stamp(2);
            ps_wait_vmcnt<0>();                      // the stores are out (and the loads of the last tick have landed)
            tr.stamp(3);
            if (lane == 0 && pub) ps_st_flag(fbase + k.p * nnt + nt, a.epoch + (unsigned)(k.t + 1));
            asm volatile("" ::: "memory");
            issue();                                 // off the hand-off path
            tr.flush(1, n, lane);
            if (pub) k.next(nrs);
        }
        PS_STAMP(1, 3)
        ps_wait_vmcnt<0>();
        PS_STAMP(1, 6)
        if (a.db) ps_barrier();                      // the MFMA waves' bias-gradient exchange
    }
}

// =============================================================================================
// Host side
// =============================================================================================
// CUs the planners may fill (d2p_lstm_persist_set_cu_budget; 0 = every CU): a launch planned for fewer leaves whole CUs to
// the other queue -- a four-wave workgroup never becomes resident on a CU that holds a recurrence's five 256-register waves
// (DESIGN_APPENDIX: corun_probe), so beside a launch that fills the chip the side queue only WAITS
static int g_ps_cu_budget = 0;
extern "C" int d2p_lstm_persist_set_cu_budget(int cus) {
    g_ps_cu_budget = cus > 0 ? cus : 0;
    return D2P_OK;
}
static int ps_num_cus() {
    // hipDeviceGetAttribute is legal under stream capture (hipGetDeviceProperties is not: a first call from
    // inside a captured training step would silently disable the persistent path); retried until it works
    static int n = 0;
    if (n <= 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
            n = v;
        else
            (void)hipGetLastError();
    }
    const int have = n > 0 ? n : 1;
    return g_ps_cu_budget > 0 && g_ps_cu_budget < have ? g_ps_cu_budget : have;
}

// row domains for `ncol` column tiles: as many as fit the chip, at most one per 16-row sub-tile
// Packed weight images of up to 8 cells in ONE launch: forward (B operand of h.Wh) and backward (Wh^T) fragments of
// each Wh [U, 4U].  The trainer's model runs it on the side stream at the start of a step, beside the conv / batch-norm
// chain, and hands the images to the recurrences through d2p_lstm_*_desc.wpack: their prologues then read contiguous
// 16-byte fragments instead of gathering from the row-major matrix.
struct PsPackArgs { int n, U; const float* wh[8]; float4* wf[8]; float4* wb[8]; };
__global__ void __launch_bounds__(256) ps_pack_weights_kernel(PsPackArgs a) {
    const long per = (long)a.U * a.U;                   // float4s per image
    const int cell = blockIdx.y;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < 2 * per; idx += (long)gridDim.x * 256L) {
        if (idx < per) {
            if (a.wf[cell]) a.wf[cell][idx] = d2p_pack_w_fwd_elem(a.U, a.wh[cell], idx);
        } else if (a.wb[cell]) {
            a.wb[cell][idx - per] = d2p_pack_w_bwd_elem(a.U, a.wh[cell], idx - per);
        }
    }
}
extern "C" int d2p_lstm_pack_weights(int n, int U, const float* const* Wh, float* const* Wf, float* const* Wb,
                                     d2p_stream_t stream) {
    D2P_REQUIRE(n >= 1 && n <= 8 && (U == 64 || U == 128 || U == 256 || U == 512) && Wh && Wf && Wb, D2P_EINVAL,
                "lstm pack weights: 1..8 cells, U in {64,128,256,512}");
    PsPackArgs a;
    a.n = n; a.U = U;
    for (int i = 0; i < 8; ++i) {
        a.wh[i] = i < n ? Wh[i] : nullptr;
        a.wf[i] = i < n ? (float4*)Wf[i] : nullptr;
        a.wb[i] = i < n ? (float4*)Wb[i] : nullptr;
        D2P_REQUIRE(i >= n || a.wh[i], D2P_EINVAL, "lstm pack weights: null Wh");
    }
    long blocks = (2L * U * U + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(ps_pack_weights_kernel, dim3((unsigned)blocks, n), dim3(256), 0, as_stream(stream), a);
    D2P_LAUNCH_CHECK("lstm_pack_weights");
    return D2P_OK;
}
static int g_ps_direct = 1;                  // 0: always the preparation launch (d2p_lstm_persist_set_direct: A/B switch)
extern "C" int d2p_lstm_persist_set_direct(int on) {
    g_ps_direct = on ? 1 : 0;
    return D2P_OK;
}
extern "C" size_t d2p_lstm_flag_words(void) { return (size_t)PS_FLAG_WORDS + PS_TICKET_WORDS + PSW_XCC_WORDS; }
extern "C" int d2p_lstm_persist_set_poll(int pipelined) {
    g_ps_poll_pipelined = pipelined & 511;
    return D2P_OK;
}
static int g_ps_sorted = 1;                  // 0: ignore the length-sorted description of a launch (A/B switch)
extern "C" int d2p_lstm_persist_set_sorted(int on) {
    g_ps_sorted = on ? 1 : 0;
    return D2P_OK;
}
static double g_ps_cost_ph = 3.3, g_ps_cost_fl = 6.3;   // backward: us per phase and the hand-off floor per pass (the planners' model)
extern "C" int d2p_lstm_persist_set_plan_cost(double us_per_phase, double floor_us) {
    if (us_per_phase > 0.0) g_ps_cost_ph = us_per_phase;
    if (floor_us > 0.0) g_ps_cost_fl = floor_us;
    return D2P_OK;
}
static int g_ps_bwd_desc = 0;                // 1: descriptors (measured 5-10 % SLOWER per launch: off); 0: the backward kernel's stores / prefetches by 64-bit pointers (round 3's form; A/B switch)
extern "C" int d2p_lstm_persist_set_bwd_desc(int on) {
    g_ps_bwd_desc = on ? 1 : 0;
    return D2P_OK;
}
static int g_ps_bwd_defer_from = 1 << 20;    // backward: deferred form from this many phases per domain (d2p_lstm_persist_set_bwd_defer);
                                             // measured -4 % per phase at 5-7 phases in isolation, nothing in the step: off by default
extern "C" int d2p_lstm_persist_set_bwd_defer(int from_phases) {
    g_ps_bwd_defer_from = from_phases > 0 ? from_phases : 1 << 20;
    return D2P_OK;
}
static int g_ps_wgs_per_cu[2] = {1, 1};      // forward, backward (experiment knob)
extern "C" int d2p_lstm_persist_set_wgs_per_cu(int fwd, int bwd) {
    if (fwd > 0) g_ps_wgs_per_cu[0] = fwd;
    if (bwd > 0) g_ps_wgs_per_cu[1] = bwd;
    return D2P_OK;
}
static int ps_pick_rt(int total_rs, int ncol, int dir = 0) {
    int rt = ps_num_cus() * g_ps_wgs_per_cu[dir] / ncol;
    if (rt > total_rs) rt = total_rs;
    return rt;
}
static bool ps_shape_ok(int M, int U, int n_steps, int ncol, int dir = 0) {
    if (M <= 0 || n_steps <= 0 || !(U == 64 || U == 128 || U == 256 || U == 512)) return false;
    const int total_rs = (M + 15) / 16;
    const int rt = ps_pick_rt(total_rs, ncol, dir);
    if (rt < 1) return false;
    const long Mp = (long)total_rs * 16;
    if (2L * Mp * 4 * U * 4 > 0x7fffffffL) return false;        // 32-bit buffer offsets
    return (total_rs + rt - 1) / rt <= PS_NRS_MAX;
}
bool d2p_lstm_persist_fwd_ok(int M, int U, int n_steps) {
    return g_persist && ps_shape_ok(M, U, n_steps, U / 8);
}
bool d2p_lstm_persist_bwd_ok(int M, int U, int n_steps) {
    return g_persist && ps_shape_ok(M, U, n_steps, U / 16, 1);
}

#define PS_DUMP_FLOATS 64
#define PS_DBPART_FLOATS (512 * 64)   // backward: one 64-float partial bias gradient per workgroup

size_t d2p_lstm_persist_ws_bytes(int M, int U) {
    const size_t Mp = (size_t)((M + 15) / 16) * 16;
    // packed weight + 2 fragment buffers over K = 4U (backward; forward needs U) + flags
    return ((size_t)4 * U * U + 2 * Mp * 4 * U) * sizeof(float) + (PS_FLAG_WORDS + PS_TICKET_WORDS) * sizeof(unsigned) +
           (PS_DUMP_FLOATS + PS_DBPART_FLOATS) * sizeof(float);
}

// Everything a persistent launch needs prepared, in ONE launch instead of three (weight pack, initial-state
// pack or zero fill, flag reset): the recurrences sit on the critical path of the step and every extra
// dependent launch in front of them costs its few microseconds plus a kernel boundary.
//   [0, nW)            packed weights (forward or backward image)
//   [nW, nW + nA)      float4s of the initial operand buffer: rows of X in fragment order, or zeros if X is null
//   [.., + nF)         flag words (one uint4 = 4 flags per index)
__global__ void __launch_bounds__(256)
ps_prep_kernel(int bwd, int U, const float* __restrict__ Wh, float4* __restrict__ Wp, long nW, int M,
               const float* __restrict__ X, float4* __restrict__ Af, long nA, uint4* __restrict__ flags, long nF) {
    const long total = nW + nA + nF;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        if (idx < nW) {
            Wp[idx] = bwd ? d2p_pack_w_bwd_elem(U, Wh, idx) : d2p_pack_w_fwd_elem(U, Wh, idx);
        } else if (idx < nW + nA) {
            const long i = idx - nW;
            Af[i] = X ? d2p_pack_rows_elem(M, U, X, i) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            flags[idx - nW - nA] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}
static int ps_prep(int bwd, int U, const float* Wh, float* Wp, int M, const float* X, float* Af, size_t af_bytes,
                   unsigned* flags, size_t nflags, hipStream_t st) {
    const long nW = (long)U * U;                       // 4U^2 floats
    const long nA = (long)(af_bytes / 16);
    const long nF = (long)((nflags + 3) / 4);          // the flag area is PS_FLAG_WORDS (a multiple of 4) long
    long blocks = (nW + nA + nF + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(ps_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, st, bwd, U, Wh, (float4*)Wp, nW, M, X,
                       (float4*)Af, nA, (uint4*)flags, nF);
    D2P_LAUNCH_CHECK("lstm_persist_prep");
    return D2P_OK;
}

// ---- one sequence's share of a launch -----------------------------------------------------------
static int ps_fwd_setup(const PsFwdCall& q, int RT, int bid0, PsFwdArgs& a, hipStream_t st) {
    const int M = q.M, U = q.U;
    a.M = M; a.U = U; a.T = q.n_steps;
    a.total_rs = (M + 15) / 16;
    const int nct = U / 8;
    a.RT = RT;
    a.bid0 = bid0; a.gsz = nct * RT;
    a.has_h0 = q.h0 ? 1 : 0;
    const size_t Mp = (size_t)a.total_rs * 16;
    float* Wf = q.ws;
    a.Wf = (const float4*)Wf;
    a.hfrag = Wf + (size_t)4 * U * U;
    a.hfrag_bytes = (unsigned)(Mp * U * sizeof(float));
    a.flags = (unsigned*)(a.hfrag + 2 * Mp * U);
    a.dump = (float*)(a.flags + PS_FLAG_WORDS + PS_TICKET_WORDS);
    a.err = ps_err_ptr();
    a.trace = g_ps_trace; a.trace_block = g_ps_trace_block;
    a.z = q.z; a.zrs = q.zrs; a.zts = q.zts; a.h0 = q.h0; a.c0 = q.c0; a.lens = q.lens;
    a.hout = q.hout; a.cs = q.cs; a.h_final = q.h_final; a.c_final = q.c_final;
    a.direct = 0; a.wh_raw = q.Wh; a.h0_raw = q.h0; a.h0_bytes = 0; a.epoch = 0; a.packed = 0;
    a.poll = g_ps_poll_pipelined;
    if (q.flags && g_ps_direct) {
        // direct launch: nothing to prepare -- weights and the initial state are read where they lie, the flags
        // (the caller's once-zeroed buffer) run on epochs
        a.direct = 1;
        a.flags = q.flags;
        a.epoch = q.epoch;
        a.h0_bytes = q.h0 ? (unsigned)((size_t)M * U * sizeof(float)) : 0u;
        if (q.wpack) {               // the caller's packed image (kept up to date off the critical path)
            a.packed = 1;
            a.Wf = (const float4*)q.wpack;
        }
        return D2P_OK;
    }
    // packed weights; h0 in fragment order (without one the deferred-epilogue form still multiplies in step 0:
    // an all-zero operand makes that product an exact zero); flags reset
    return ps_prep(0, U, q.Wh, Wf, M, q.h0, a.hfrag, a.hfrag_bytes, a.flags, (size_t)RT * PS_NRS_MAX * nct, st);
}
static int ps_fwd_launch(const PsFwdArgs& a0, const PsFwdArgs& a1, int U, double flops, hipStream_t st) {
    const int blocks = a0.gsz + a1.gsz;
    D2pProfScope prof(st, D2P_PROF_LSTM_STEP_FWD, flops);
    switch (U) {
        case 64: hipLaunchKernelGGL((lstm_persist_fwd_kernel<1>), dim3(blocks), dim3(PS_THREADS), 0, st, a0, a1); break;
        case 128: hipLaunchKernelGGL((lstm_persist_fwd_kernel<2>), dim3(blocks), dim3(PS_THREADS), 0, st, a0, a1); break;
        case 256: hipLaunchKernelGGL((lstm_persist_fwd_kernel<4>), dim3(blocks), dim3(PS_THREADS), 0, st, a0, a1); break;
        default: hipLaunchKernelGGL((lstm_persist_fwd_kernel<8>), dim3(blocks), dim3(PS_THREADS), 0, st, a0, a1); break;
    }
    D2P_LAUNCH_CHECK("lstm_persist_fwd");
    return D2P_OK;
}
static inline double ps_fwd_flops(const PsFwdCall& q) {
    return 2.0 * q.M * 4.0 * q.U * q.U * (q.n_steps - (q.h0 ? 0 : 1));
}

int d2p_lstm_persist_fwd(const PsFwdCall& q, hipStream_t st) {
    PsFwdArgs a, none;
    int rc = ps_fwd_setup(q, ps_pick_rt((q.M + 15) / 16, q.U / 8), 0, a, st);
    if (rc) return rc;
    none = a;
    none.gsz = 0;
    return ps_fwd_launch(a, none, q.U, ps_fwd_flops(q), st);
}

// ---- two sequences in one launch ---------------------------------------------------------------------
// Row domains are dealt out between the two so that the slower one finishes as early as possible under a
// simple per-step model (a step costs its phases, but never less than the hand-off latency of one phase);
// the pair is taken only when that beats the two launches back to back.  Measured at U = 512 (us):
// forward 1.6 per phase, floor 3.3; backward 3.3 per phase, floor 6.3.
static bool ps_plan_pair(int trs_a, int T_a, int trs_b, int T_b, int ncol, int dir, int& RTa, int& RTb) {
    const double ph = dir ? 3.3 : 1.6, fl = dir ? 6.3 : 3.3;
    auto cost = [&](int trs, int T, int RT) {
        const int nrs = (trs + RT - 1) / RT;
        const double step = ph * nrs;
        return T * (step > fl ? step : fl);
    };
    const int budget = ps_num_cus() / ncol;                 // row domains the chip holds at one workgroup per CU
    const int fa = ps_pick_rt(trs_a, ncol, dir), fb = ps_pick_rt(trs_b, ncol, dir);
    if (fa < 1 || fb < 1) return false;
    const double serial = cost(trs_a, T_a, fa) + cost(trs_b, T_b, fb);
    double best = serial * 0.9;                             // must win by a margin
    bool found = false;
    for (int rb = 1; rb < budget; ++rb) {
        int ra = budget - rb;
        if (ra > trs_a) ra = trs_a;
        if (rb > trs_b || ra < 1) continue;
        if ((trs_a + ra - 1) / ra > PS_NRS_MAX || (trs_b + rb - 1) / rb > PS_NRS_MAX) continue;
        const double ca = cost(trs_a, T_a, ra), cb = cost(trs_b, T_b, rb);
        const double c = ca > cb ? ca : cb;
        if (c < best) { best = c; RTa = ra; RTb = rb; found = true; }
    }
    return found;
}

bool d2p_lstm_persist_fwd_pair_ok(int Ma, int Ta, int Mb, int Tb, int U) {
    int ra, rb;
    return g_persist && d2p_lstm_persist_fwd_ok(Ma, U, Ta) && d2p_lstm_persist_fwd_ok(Mb, U, Tb) &&
           ps_plan_pair((Ma + 15) / 16, Ta, (Mb + 15) / 16, Tb, U / 8, 0, ra, rb);
}
static int g_ps_pair_launches = 0;
extern "C" int d2p_lstm_persist_pair_launches(void) { return g_ps_pair_launches; }
int d2p_lstm_persist_fwd_pair(const PsFwdCall& qa, const PsFwdCall& qb, hipStream_t st) {
    int ra = 0, rb = 0;
    if (!ps_plan_pair((qa.M + 15) / 16, qa.n_steps, (qb.M + 15) / 16, qb.n_steps, qa.U / 8, 0, ra, rb))
        return D2P_EINVAL;
    PsFwdArgs a, b;
    int rc = ps_fwd_setup(qa, ra, 0, a, st);
    if (rc) return rc;
    rc = ps_fwd_setup(qb, rb, a.gsz, b, st);
    if (rc) return rc;
    ++g_ps_pair_launches;
    return ps_fwd_launch(a, b, qa.U, ps_fwd_flops(qa) + ps_fwd_flops(qb), st);
}

// Length-sorted launch: cut the (sorted, longest first) sub-tiles into RT contiguous domains so that the slowest
// domain -- passes of its first sub-tile x the per-pass cost of its phases -- is as fast as possible (dynamic
// programme over the cut positions; steps[s] = passes sub-tile s needs = the length of its longest row).
static bool ps_plan_sorted(int trs, int RT, const int* steps, int extra_pass, double ph, double fl, int* rs_start,
                           int* tdom, double (*step_cost)(int) = nullptr) {
    if (RT < 1 || RT > PS_RT_TAB || trs < RT || trs > 256) return false;
    auto cost = [&](int s0, int s1) {          // domain of sub-tiles [s0, s1)
        const double step = ph * (s1 - s0);
        const double per = step_cost ? step_cost(s1 - s0) : (step > fl ? step : fl);
        return (double)(steps[s0] + extra_pass) * per;
    };
    // best[d][s]: smallest possible maximum over the first d domains covering sub-tiles [0, s)
    static thread_local double best[PS_RT_TAB + 1][257];
    static thread_local int cut[PS_RT_TAB + 1][257];
    for (int d = 0; d <= RT; ++d)
        for (int s2 = 0; s2 <= trs; ++s2) best[d][s2] = 1e30;
    best[0][0] = 0.0;
    for (int d = 1; d <= RT; ++d)
        for (int s2 = d; s2 <= trs - (RT - d); ++s2)
            for (int s1 = (s2 - PS_NRS_MAX > d - 1 ? s2 - PS_NRS_MAX : d - 1); s1 < s2; ++s1) {
                if (best[d - 1][s1] >= 1e30) continue;
                const double c = cost(s1, s2), m = c > best[d - 1][s1] ? c : best[d - 1][s1];
                if (m < best[d][s2]) { best[d][s2] = m; cut[d][s2] = s1; }
            }
    if (best[RT][trs] >= 1e30) return false;
    int s2 = trs;
    for (int d = RT; d >= 1; --d) {
        rs_start[d] = s2;
        s2 = cut[d][s2];
    }
    rs_start[0] = 0;
    for (int d = 0; d < RT; ++d) {
        int t = steps[rs_start[d]];
        tdom[d] = t < 1 ? 1 : t;
    }
    for (int d = RT; d < PS_RT_TAB; ++d) { rs_start[d + 1] = trs; tdom[d] = 1; }
    return true;
}

static int ps_bwd_setup(const PsBwdCall& q, int RT, int bid0, PsBwdArgs& a, hipStream_t st, double* flops = nullptr) {
    const int M = q.M, U = q.U, n_steps = q.n_steps;
    a.M = M; a.U = U; a.T = n_steps;
    a.total_rs = (M + 15) / 16;
    const int nnt = U / 16;
    a.RT = RT;
    a.bid0 = bid0; a.gsz = nnt * RT;
    a.want_dh0 = q.dh0 ? 1 : 0;
    const size_t Mp = (size_t)a.total_rs * 16;
    float* Wb = q.ws;
    a.Wb = (const float4*)Wb;
    a.dzfrag = Wb + (size_t)4 * U * U;
    a.dzfrag_bytes = (unsigned)(Mp * 4 * U * sizeof(float));
    a.flags = (unsigned*)(a.dzfrag + 2 * Mp * 4 * U);
    a.dbtick = a.flags + PS_FLAG_WORDS;
    a.dump = (float*)(a.flags + PS_FLAG_WORDS + PS_TICKET_WORDS);
    a.dbpart = a.dump + PS_DUMP_FLOATS;
    a.db = q.db;
    a.defer_from = g_ps_bwd_defer_from;
    a.lds_nb = (a.total_rs + RT - 1) / RT >= g_ps_bwd_defer_from ? 2 : 1;
    a.direct = 0; a.wh_raw = q.Wh; a.epoch = 0; a.packed = 0;
    a.poll = g_ps_poll_pipelined;
    // 32-bit byte offsets into z / dz (rows at zrs, steps at zts), cs, dhout; 24-bit row strides in bytes; rows < 2^15
    a.use_desc = (g_ps_bwd_desc && M <= 32767 && q.zrs * 4 < (1L << 24) && q.zts > 0 && q.zrs >= 4L * U &&
                  ((long)(n_steps - 1) * q.zts + (long)(M - 1) * q.zrs + 4L * U) * 4 < 0x7fffff00L &&
                  (long)n_steps * M * U * 4 < 0x7fffff00L) ? 1 : 0;
    a.err = ps_err_ptr();
    a.trace = g_ps_trace; a.trace_block = g_ps_trace_block;
    a.z = q.z; a.zrs = q.zrs; a.zts = q.zts; a.c0 = q.c0; a.cs = q.cs; a.lens = q.lens;
    a.dhout = q.dhout; a.dh_final = q.dh_final; a.dc_final = q.dc_final;
    a.dz = q.dz; a.dh0 = q.dh0; a.dc0 = q.dc0;
    a.rowmap = nullptr; a.sorted = 0; a.Tfull = n_steps;
    for (int d = 0; d <= PS_RT_TAB; ++d) a.rs_start[d] = 0;
    for (int d = 0; d < PS_RT_TAB; ++d) a.tdom[d] = 1;
    if (flops) *flops = 2.0 * M * 4.0 * U * U * (n_steps - 1 + (q.dh0 ? 1 : 0));
    if (q.flags && g_ps_direct) {
        a.direct = 1;
        a.flags = q.flags;
        a.dbtick = q.flags + PS_FLAG_WORDS;
        a.epoch = q.epoch;
        if (q.wpack) {
            a.packed = 1;
            a.Wb = (const float4*)q.wpack;
        }
        if (q.rowmap && q.slab_steps && g_ps_sorted &&
            ps_plan_sorted(a.total_rs, RT, q.slab_steps, q.dh0 ? 1 : 0, g_ps_cost_ph, g_ps_cost_fl, a.rs_start, a.tdom)) {
            a.rowmap = q.rowmap;
            a.sorted = 1;
            for (int d = 0; d < RT; ++d)
                if (a.tdom[d] > n_steps) a.tdom[d] = n_steps;
            // the deferred form's LDS follows the longest domain of THIS cut
            int nmax = 0;
            for (int d = 0; d < RT; ++d) nmax = a.rs_start[d + 1] - a.rs_start[d] > nmax ? a.rs_start[d + 1] - a.rs_start[d] : nmax;
            a.lds_nb = nmax >= g_ps_bwd_defer_from ? 2 : 1;
            if (flops) {                     // executed: each domain's rows x its own passes
                double f = 0.0;
                for (int d = 0; d < RT; ++d) {
                    int rows = (a.rs_start[d + 1] - a.rs_start[d]) * 16;
                    if (a.rs_start[d + 1] * 16 > M) rows -= a.rs_start[d + 1] * 16 - M;
                    f += 2.0 * rows * 4.0 * U * U * (a.tdom[d] - 1 + (q.dh0 ? 1 : 0));
                }
                *flops = f;
            }
        }
        return D2P_OK;
    }
    // packed Wh^T; pass 0 has no product -- the chain runs on an all-zero dz[T]; flags reset
    // (the whole flag area + the bias-gradient tickets behind it: 1040 uint4 next to 262144 weight float4s)
    return ps_prep(1, U, q.Wh, Wb, M, nullptr, (float*)((char*)a.dzfrag + (size_t)(n_steps & 1) * a.dzfrag_bytes),
                   a.dzfrag_bytes, a.flags, (size_t)PS_FLAG_WORDS + PS_TICKET_WORDS, st);
}
static int ps_bwd_launch(const PsBwdArgs& a0, const PsBwdArgs& a1, const PsBwdArgs& a2, int U, double flops,
                         hipStream_t st) {
    const int blocks = a0.gsz + a1.gsz + a2.gsz;
    // every sequence of the launch gets the same LDS layout (one dynamic size per launch)
    PsBwdArgs b0 = a0, b1 = a1, b2 = a2;
    const int nb = ((a0.gsz > 0 && a0.lds_nb == 2) || (a1.gsz > 0 && a1.lds_nb == 2) || (a2.gsz > 0 && a2.lds_nb == 2)) ? 2 : 1;
    b0.lds_nb = b1.lds_nb = b2.lds_nb = nb;
    const bool desc = (a0.gsz == 0 || a0.use_desc) && (a1.gsz == 0 || a1.use_desc) && (a2.gsz == 0 || a2.use_desc);
    const size_t lds = (size_t)(nb * PS_BWD_P_FLOATS + (PS_NRS_MAX + 1) * 256 + 2 * PS_NRS_MAX * 16 + nb * 1024 +
                                PS_PF_R * PS_BWD_SLOT) * sizeof(float);
    PS_STAMP_LAUNCH(st)
    D2pProfScope prof(st, D2P_PROF_LSTM_STEP_BWD, flops);
#define PS_BWD_LAUNCH(CPW)                                                                                              \
    {                                                                                                                   \
        static bool attr = false;                                                                                       \
        if (!attr) {                                                                                                    \
            (void)hipFuncSetAttribute((const void*)lstm_persist_bwd_kernel<CPW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      96 * 1024);                                                                       \
            (void)hipFuncSetAttribute((const void*)lstm_persist_bwd_kernel<CPW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      96 * 1024);                                                                       \
            attr = true;                                                                                                \
        }                                                                                                               \
        if (desc) hipLaunchKernelGGL((lstm_persist_bwd_kernel<CPW, true>), dim3(blocks), dim3(PS_THREADS), lds, st, b0, b1, b2);  \
        else hipLaunchKernelGGL((lstm_persist_bwd_kernel<CPW, false>), dim3(blocks), dim3(PS_THREADS), lds, st, b0, b1, b2);      \
    }
    switch (U) {
        case 64: PS_BWD_LAUNCH(1) break;
        case 128: PS_BWD_LAUNCH(2) break;
        case 256: PS_BWD_LAUNCH(4) break;
        default: PS_BWD_LAUNCH(8) break;
    }
#undef PS_BWD_LAUNCH
    D2P_LAUNCH_CHECK("lstm_persist_bwd");
    return D2P_OK;
}

int d2p_lstm_persist_bwd(const PsBwdCall& q, hipStream_t st) {
    PsBwdArgs a, none;
    double flops = 0.0;
    int rc = ps_bwd_setup(q, ps_pick_rt((q.M + 15) / 16, q.U / 16, 1), 0, a, st, &flops);
    if (rc) return rc;
    none = a;
    none.gsz = 0;
    return ps_bwd_launch(a, none, none, q.U, flops, st);
}
bool d2p_lstm_persist_bwd_pair_ok(int Ma, int Ta, int Mb, int Tb, int U) {
    int ra, rb;
    return g_persist && d2p_lstm_persist_bwd_ok(Ma, U, Ta) && d2p_lstm_persist_bwd_ok(Mb, U, Tb) &&
           ps_plan_pair((Ma + 15) / 16, Ta, (Mb + 15) / 16, Tb, U / 16, 1, ra, rb);
}
int d2p_lstm_persist_bwd_pair(const PsBwdCall& qa, const PsBwdCall& qb, hipStream_t st) {
    int ra = 0, rb = 0;
    if (!ps_plan_pair((qa.M + 15) / 16, qa.n_steps, (qb.M + 15) / 16, qb.n_steps, qa.U / 16, 1, ra, rb))
        return D2P_EINVAL;
    PsBwdArgs a, b;
    double fa = 0.0, fb = 0.0;
    int rc = ps_bwd_setup(qa, ra, 0, a, st, &fa);
    if (rc) return rc;
    rc = ps_bwd_setup(qb, rb, a.gsz, b, st, &fb);
    if (rc) return rc;
    ++g_ps_pair_launches;
    PsBwdArgs none = b;
    none.gsz = 0;
    return ps_bwd_launch(a, b, none, qa.U, fa + fb, st);
}

// three sequences (the three decoders' backward recurrences): row domains dealt out by the same cost model; taken
// when it beats the best of {three launches, one launch + a pair}
static double ps_bwd_cost(int trs, int T, int RT) {
    const int nrs = (trs + RT - 1) / RT;
    const double step = 3.3 * nrs;
    return T * (step > 6.3 ? step : 6.3);
}
static bool ps_plan_triple_bwd(const int trs[3], const int T[3], int ncol, int R[3]) {
    const int budget = ps_num_cus() / ncol;
    double single[3];
    for (int i = 0; i < 3; ++i) {
        const int f = ps_pick_rt(trs[i], ncol, 1);
        if (f < 1) return false;
        single[i] = ps_bwd_cost(trs[i], T[i], f) + 25.0;                     // + the fixed cost of a launch
    }
    double alt = single[0] + single[1] + single[2];
    for (int i = 0; i < 3; ++i) {                                             // one alone + the other two as a pair
        const int j = (i + 1) % 3, l = (i + 2) % 3;
        int ra, rb;
        if (ps_plan_pair(trs[j], T[j], trs[l], T[l], ncol, 1, ra, rb)) {
            const double cj = ps_bwd_cost(trs[j], T[j], ra), cl = ps_bwd_cost(trs[l], T[l], rb);
            const double c = single[i] + (cj > cl ? cj : cl) + 25.0;
            if (c < alt) alt = c;
        }
    }
    double best = alt * 0.92;
    bool found = false;
    for (int r0 = 1; r0 < budget; ++r0)
        for (int r1 = 1; r0 + r1 < budget; ++r1) {
            int r[3] = {r0, r1, budget - r0 - r1};
            bool ok = true;
            double c = 0.0;
            for (int i = 0; i < 3 && ok; ++i) {
                if (r[i] > trs[i]) r[i] = trs[i];
                ok = r[i] >= 1 && (trs[i] + r[i] - 1) / r[i] <= PS_NRS_MAX;
                const double ci = ps_bwd_cost(trs[i], T[i], r[i]);
                if (ci > c) c = ci;
            }
            if (ok && c + 25.0 < best) { best = c + 25.0; R[0] = r[0]; R[1] = r[1]; R[2] = r[2]; found = true; }
        }
    return found;
}
bool d2p_lstm_persist_bwd_triple_ok(const int M[3], const int T[3], int U) {
    if (!g_persist) return false;
    int trs[3], R[3];
    for (int i = 0; i < 3; ++i) {
        if (!d2p_lstm_persist_bwd_ok(M[i], U, T[i])) return false;
        trs[i] = (M[i] + 15) / 16;
    }
    return ps_plan_triple_bwd(trs, T, U / 16, R);
}
int d2p_lstm_persist_bwd_triple(const PsBwdCall q[3], hipStream_t st) {
    int trs[3], T[3], R[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) { trs[i] = (q[i].M + 15) / 16; T[i] = q[i].n_steps; }
    if (!ps_plan_triple_bwd(trs, T, q[0].U / 16, R)) return D2P_EINVAL;
    PsBwdArgs a[3];
    int bid0 = 0;
    double flops = 0.0;
    for (int i = 0; i < 3; ++i) {
        double f = 0.0;
        int rc = ps_bwd_setup(q[i], R[i], bid0, a[i], st, &f);
        if (rc) return rc;
        bid0 += a[i].gsz;
        flops += f;
    }
    g_ps_pair_launches += 2;          // counts as two fusions (tests: the fused path must not be skipped silently)
    return ps_bwd_launch(a[0], a[1], a[2], q[0].U, flops, st);
}

// =============================================================================================
// Forward, wide column tiles: host side
// =============================================================================================
static int g_psw_on = 1;                     // 0: every forward launch goes to the narrow-tile kernel (A/B switch)
static int g_psw_la_from = 3, g_psw_defer_from = 5, g_psw_xcd_local = 1, g_psw_la_q = 2;
static double g_psw_cost_ph = 2.7, g_psw_cost_fl = 4.6;      // planner knobs: scale of the per-phase costs / of the single-phase step (psw_step_cost)
extern "C" int d2p_lstm_persist_set_fwd_wide(int on, int la_from, int defer_from, int xcd_local) {
    g_psw_on = on ? 1 : 0;
    if (la_from >= 2) g_psw_la_from = la_from;
    if (defer_from >= 3) g_psw_defer_from = defer_from;
    if (xcd_local >= 0) g_psw_xcd_local = (xcd_local & 1) ? 1 : 0;
    if (xcd_local >= 0 && (xcd_local & 6)) g_psw_la_q = (xcd_local & 2) ? 2 : 3;      // bits 1 / 2: request after 2 / 3 quarters
    return D2P_OK;
}
extern "C" int d2p_lstm_persist_set_fwd_plan_cost(double us_per_phase, double floor_us) {
    if (us_per_phase > 0.0) g_psw_cost_ph = us_per_phase;
    if (floor_us > 0.0) g_psw_cost_fl = floor_us;
    return D2P_OK;
}
static int g_psw_launches[4] = {0, 0, 0, 0};     // wide launches so far that carried 1, 2, 3 sequences ([0]: sorted ones)
extern "C" int d2p_lstm_persist_wide_launches(int nseq) {
    return (nseq >= 0 && nseq <= 3) ? g_psw_launches[nseq] : 0;
}

extern "C" int d2p_lstm_persist_wide_local_wgs(int reset) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_psw_local_wgs), sizeof(v)) != hipSuccess) return -1;
    if (reset) {
        const unsigned z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_psw_local_wgs), &z, sizeof(z));
    }
    return (int)(v & 0x7fffffffu);
}
// us per step of a row domain with nrs phases, as measured IN THE TRAINING STEP (tools/lstm_launch_stamps.py, round 6: the
// tick loops of every domain of the three forward launches; a single phase is the bare hand-off chain, two phases poll for
// their own rows with nothing fetched ahead, from three the next rows are requested ahead, from defer_from the gate math
// rides in the next phase's chain), scaled by the two knobs.  Rounds 4-5 planned with a stand-alone sweep's 4.6 / 7.0 /
// 2.83 n / 2.65 n: 8 % low for one phase, 13 % high for two -- the cuts of the sorted launches follow the table
// (2.585 -> 2.573 ms per step, profiles/r06_ab_fwd_planner.log).
static double psw_step_cost(int nrs) {
    const double ph = g_psw_cost_ph / 2.7, fl = g_psw_cost_fl / 4.6;
    static const double tab[PS_NRS_MAX + 1] = {5.0, 5.0, 6.2, 7.9, 10.6, 12.4, 14.7, 17.2, 19.6};
    if (nrs <= 1) return tab[1] * fl;
    return tab[nrs > PS_NRS_MAX ? PS_NRS_MAX : nrs] * ph * (nrs > PS_NRS_MAX ? (double)nrs / PS_NRS_MAX : 1.0);
}
static double psw_cost(int trs, int T, int RT) { return T * psw_step_cost((trs + RT - 1) / RT); }
// row domains of a launch with n sequences: all of the chip's (num_cus / column tiles) domains, dealt out so that the
// slowest sequence finishes as early as possible under the per-step cost model
static bool psw_plan(int n, const int* trs, const int* T, int ncol, int* R) {
    const int budget = ps_num_cus() / ncol > PS_RT_TAB ? PS_RT_TAB : ps_num_cus() / ncol;
    if (budget < n) return false;
    auto ok = [&](int i, int r) { return r >= 1 && r <= trs[i] && (trs[i] + r - 1) / r <= PS_NRS_MAX; };
    auto cap = [&](int i, int r) { return r > trs[i] ? trs[i] : r; };
    if (n == 1) {
        R[0] = cap(0, budget);
        return ok(0, R[0]);
    }
    double best = 1e30;
    bool found = false;
    if (n == 2) {
        for (int r0 = 1; r0 < budget; ++r0) {
            const int r[2] = {cap(0, r0), cap(1, budget - r0)};
            if (!ok(0, r[0]) || !ok(1, r[1])) continue;
            const double c0 = psw_cost(trs[0], T[0], r[0]), c1 = psw_cost(trs[1], T[1], r[1]);
            const double c = c0 > c1 ? c0 : c1;
            if (c < best) { best = c; R[0] = r[0]; R[1] = r[1]; found = true; }
        }
        return found;
    }
    for (int r0 = 1; r0 < budget; ++r0)
        for (int r1 = 1; r0 + r1 < budget; ++r1) {
            const int r[3] = {cap(0, r0), cap(1, r1), cap(2, budget - r0 - r1)};
            if (!ok(0, r[0]) || !ok(1, r[1]) || !ok(2, r[2])) continue;
            double c = 0.0;
            for (int i = 0; i < 3; ++i) {
                const double ci = psw_cost(trs[i], T[i], r[i]);
                if (ci > c) c = ci;
            }
            if (c < best) { best = c; R[0] = r[0]; R[1] = r[1]; R[2] = r[2]; found = true; }
        }
    return found;
}

bool d2p_lstm_persist_fwd_wide_ok(int n, const PsFwdCall* q) {
    if (!g_persist || !g_psw_on || !g_ps_direct || n < 1 || n > 3) return false;
    int trs[3], T[3], R[3];
    for (int i = 0; i < n; ++i) {
        const int U = q[i].U;
        if (q[i].M <= 0 || q[i].n_steps <= 0 || q[i].n_steps >= 0xffff || q[i].M > 32767 || U != q[0].U) return false;
        if (!(U == 64 || U == 128 || U == 256 || U == 512) || !q[i].flags) return false;
        const long Mp = (long)((q[i].M + 15) / 16) * 16;
        if (2L * Mp * U * 4 > 0x7fffffffL) return false;
        // 32-bit byte offsets into z (rows at zrs, steps at zts), cs and hout; 24-bit row strides in bytes
        if (((long)(q[i].n_steps - 1) * q[i].zts + (long)(q[i].M - 1) * q[i].zrs + 4L * U) * 4 >= 0x7fffff00L ||
            (long)q[i].n_steps * q[i].M * U * 4 >= 0x7fffff00L || q[i].zrs * 4 >= (1L << 24) || q[i].zts <= 0 ||
            q[i].zrs < 4L * U)
            return false;
        for (int j = 0; j < i; ++j)
            if (q[j].ws == q[i].ws || q[j].flags == q[i].flags) return false;
        trs[i] = (q[i].M + 15) / 16;
        T[i] = q[i].n_steps;
    }
    return psw_plan(n, trs, T, q[0].U / 16, R);
}

int d2p_lstm_persist_fwd_wide(int n, const PsFwdCall* q, hipStream_t st) {
    int trs[3] = {0, 0, 0}, T[3] = {0, 0, 0}, R[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) { trs[i] = (q[i].M + 15) / 16; T[i] = q[i].n_steps; }
    const int U = q[0].U, nnt = U / 16;
    if (!psw_plan(n, trs, T, nnt, R)) return D2P_EINVAL;
    PsFwdWArgs a[3];
    int dom = 0, nb = 1;
    double flops = 0.0;
    bool any_sorted = false;
    for (int i = 0; i < n; ++i) {
        const PsFwdCall& c = q[i];
        PsFwdWArgs& x = a[i];
        x.M = c.M; x.U = U; x.T = c.n_steps; x.total_rs = trs[i]; x.RT = R[i]; x.has_h0 = c.h0 ? 1 : 0;
        x.dom0 = dom;
        dom += R[i];
        const size_t Mp = (size_t)trs[i] * 16;
        x.hfrag = c.ws + (size_t)4 * U * U;              // (the workspace layout of the narrow kernel: its packed weights first)
        x.hfrag_bytes = (unsigned)(Mp * U * sizeof(float));
        x.z = c.z; x.zrs = c.zrs; x.zts = c.zts; x.c0 = c.c0; x.lens = c.lens;
        x.hout = c.hout; x.cs = c.cs; x.h_final = c.h_final; x.c_final = c.c_final;
        x.flags = c.flags;
        x.dump = (float*)((unsigned*)(x.hfrag + 2 * Mp * U) + PS_FLAG_WORDS + PS_TICKET_WORDS);
        x.err = ps_err_ptr();
        x.trace = g_ps_trace; x.trace_block = g_ps_trace_block;
        x.wh_raw = c.Wh; x.h0_raw = c.h0; x.h0_bytes = c.h0 ? (unsigned)((size_t)c.M * U * sizeof(float)) : 0u;
        x.epoch = c.epoch;
        x.poll = g_ps_poll_pipelined;
        x.la_from = g_psw_la_from; x.defer_from = g_psw_defer_from; x.la_q = g_psw_la_q;
        x.xcd_local = g_psw_xcd_local;
        x.rowmap = nullptr; x.sorted = 0; x.Tfull = c.n_steps;
        for (int d = 0; d <= PS_RT_TAB; ++d) x.rs_start[d] = 0;
        for (int d = 0; d < PS_RT_TAB; ++d) x.tdom[d] = 1;
        int nmax = (trs[i] + R[i] - 1) / R[i];
        double f = 2.0 * c.M * 4.0 * U * U * (c.n_steps - (c.h0 ? 0 : 1));
        if (c.rowmap && c.slab_steps && c.lens && g_ps_sorted &&
            ps_plan_sorted(trs[i], R[i], c.slab_steps, 0, g_psw_cost_ph, g_psw_cost_fl, x.rs_start, x.tdom, psw_step_cost)) {
            x.rowmap = c.rowmap;
            x.sorted = 1;
            any_sorted = true;
            nmax = 0;
            f = 0.0;
            for (int d = 0; d < R[i]; ++d) {
                if (x.tdom[d] > c.n_steps) x.tdom[d] = c.n_steps;
                const int ph = x.rs_start[d + 1] - x.rs_start[d];
                nmax = ph > nmax ? ph : nmax;
                int rows = ph * 16;
                if (x.rs_start[d + 1] * 16 > c.M) rows -= x.rs_start[d + 1] * 16 - c.M;
                f += 2.0 * rows * 4.0 * U * U * (x.tdom[d] - (c.h0 ? 0 : 1));
            }
        }
        if (nmax >= g_psw_defer_from) nb = 2;
        flops += f;
    }
    for (int i = n; i < 3; ++i) { a[i] = a[0]; a[i].RT = 0; a[i].dom0 = dom; }
    for (int i = 0; i < 3; ++i) a[i].lds_nb = nb;
    PsFwdWLaunch L;
    for (int i = 0; i < 3; ++i) L.s[i] = a[i];
    const size_t lds = (size_t)(nb * PSW_P_FLOATS + 2 * PS_NRS_MAX * PSW_CELLS + 2 * PS_NRS_MAX * 16 + 16 + nb * PSW_CELLS +
                                PS_PF_R * PSW_SLOT) * sizeof(float);
    const int blocks = dom * nnt;
    ++g_psw_launches[n];
    if (any_sorted) ++g_psw_launches[0];
    PS_STAMP_LAUNCH(st)
    D2pProfScope prof(st, D2P_PROF_LSTM_STEP_FWD, flops);
#define PSW_LAUNCH(CPW)                                                                                                 \
    {                                                                                                                   \
        static bool attr = false;                                                                                       \
        if (!attr) {                                                                                                    \
            (void)hipFuncSetAttribute((const void*)lstm_persist_fwdw_kernel<CPW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      96 * 1024);                                                                       \
            attr = true;                                                                                                \
        }                                                                                                               \
        hipLaunchKernelGGL((lstm_persist_fwdw_kernel<CPW>), dim3(blocks), dim3(PS_THREADS), lds, st, L);                \
    }
    switch (U) {
        case 64: PSW_LAUNCH(1) break;
        case 128: PSW_LAUNCH(2) break;
        case 256: PSW_LAUNCH(4) break;
        default: PSW_LAUNCH(8) break;
    }
#undef PSW_LAUNCH
    D2P_LAUNCH_CHECK("lstm_persist_fwdw");
    return D2P_OK;
}
