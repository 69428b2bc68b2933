// Optional per-launch HIP-event timing used by bench.py's roofline leg (include/d2p.h,
// "profiling").  Disabled by default: when off, the hooks are two predictable branches.
// Events are recorded on the SAME stream the kernel is launched on, so elapsed time is the
// kernel's device duration (plus event overhead), independent of what torch's current
// stream is.  Not for use under hipGraph capture.
#pragma once
#include <hip/hip_runtime.h>

// kernel families (key = family * 8 + sub-tag)
#define D2P_PROF_GEMM 1        // dense fp32 MFMA GEMM       work = FLOP
#define D2P_PROF_CONV 2        // implicit-im2col conv GEMM  work = FLOP
#define D2P_PROF_GATE_FWD 3    // LSTM gate pointwise fwd    work = algorithmic bytes
#define D2P_PROF_GATE_BWD 4    // LSTM gate pointwise bwd    work = algorithmic bytes
#define D2P_PROF_BN 5          // batch norm (stats+apply)   work = algorithmic bytes
#define D2P_PROF_ADAM 6        // l2norm + clip + Adam       work = algorithmic bytes
#define D2P_PROF_LSTM_STEP_FWD 7   // fused recurrent step fwd  work = FLOP
#define D2P_PROF_LSTM_STEP_BWD 8   // fused recurrent step bwd  work = FLOP

bool d2p_prof_on();
int d2p_prof_tag();
void d2p_prof_begin(hipStream_t st, int family, double work);
void d2p_prof_end(hipStream_t st);

struct D2pProfScope {
    hipStream_t st;
    bool on;
    D2pProfScope(hipStream_t s, int family, double work) : st(s), on(d2p_prof_on()) {
        if (on) d2p_prof_begin(st, family, work);
    }
    ~D2pProfScope() {
        if (on) d2p_prof_end(st);
    }
};
