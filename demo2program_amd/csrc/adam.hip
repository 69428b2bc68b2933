// K8: global-norm clip + Adam over one flat parameter buffer (include/d2p.h).
// Replaces tf.contrib.layers.optimize_loss(clip_gradients=20.0, AdamOptimizer) at
// trainer.py:102-109: [TF-1.3] clip_by_global_norm(grads, 20) then Adam
//   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= lr_t * m / (sqrt(v) + eps),
//   lr_t = lr*sqrt(1-b2^t)/(1-b1^t)  (computed by the caller, who knows t).
// The same flat buffer is what the data-parallel all-reduce operates on; `prescale`
// (= 1/world_size) turns the all-reduced SUM into the mean inside these kernels, so the
// averaged gradient is never written back to HBM.  The norm stays on the device (fp64).
#include "common.h"

#define L2_BLOCKS 1024

extern "C" size_t d2p_l2norm_ws_bytes(size_t n) {
    (void)n;
    return (size_t)L2_BLOCKS * sizeof(double);
}

__global__ void __launch_bounds__(256)
l2norm_partial_kernel(size_t n, const float* g, double* partial) {
    __shared__ double red[4];
    double s = 0.0;
    const size_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = blockIdx.x * 256UL + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256UL) {
        const float4 v = g4[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        s += (double)v * v;
    }
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256)
l2norm_final_kernel(int nb, const double* partial, double prescale2, double* sumsq) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sumsq[0] = (red[0] + red[1] + red[2] + red[3]) * prescale2;
}

extern "C" int d2p_l2norm_flat(size_t n, const float* g, float prescale, double* sumsq, void* ws,
                               size_t ws_bytes, d2p_stream_t stream) {
    D2P_REQUIRE(sumsq && (n == 0 || g), D2P_EINVAL, "l2norm: null pointer");
    D2P_REQUIRE(((uintptr_t)g & 15) == 0, D2P_EALIGN, "l2norm: g must be 16-byte aligned");
    D2P_REQUIRE(ws && ws_bytes >= d2p_l2norm_ws_bytes(n), D2P_EWS, "l2norm: workspace too small");
    hipStream_t st = as_stream(stream);
    size_t want = (n / 4 + 255) / 256;
    int nb = (int)(want < 1 ? 1 : (want > L2_BLOCKS ? L2_BLOCKS : want));
    hipLaunchKernelGGL(l2norm_partial_kernel, dim3(nb), dim3(256), 0, st, n, g, (double*)ws);
    D2P_LAUNCH_CHECK("l2norm_partial");
    hipLaunchKernelGGL(l2norm_final_kernel, dim3(1), dim3(256), 0, st, nb, (const double*)ws,
                       (double)prescale * (double)prescale, sumsq);
    D2P_LAUNCH_CHECK("l2norm_final");
    return D2P_OK;
}

// The status word of the persistent recurrent kernels (lstm_persist.hip): non-zero once a launch gave up a
// hand-off -- the gradients of that step, and of every step until the host resets the word, are invalid.
unsigned* d2p_persist_err_ptr();

__global__ void step_status_publish_kernel(const unsigned* err, float* slot) {
    if (threadIdx.x == 0) slot[0] = (*err != 0u) ? 1.f : 0.f;
}

extern "C" int d2p_step_status_publish(float* slot, d2p_stream_t stream) {
    D2P_REQUIRE(slot, D2P_EINVAL, "step status: null slot");
    hipLaunchKernelGGL(step_status_publish_kernel, dim3(1), dim3(64), 0, as_stream(stream), d2p_persist_err_ptr(), slot);
    D2P_LAUNCH_CHECK("step_status_publish");
    return D2P_OK;
}

// err / fail_slot / counters: the guarded form (d2p_adam_clip_flat_guarded).  The update is SKIPPED -- parameters
// and moments untouched -- when this device's status word is set or the (all-reduced) slot says some rank's is;
// counters[0] counts applied steps, counters[1] skipped ones; mirror (optional, host-mapped pinned memory): a copy of
// both after this step, so that the host can follow them without a copy operation on the stream.
__global__ void __launch_bounds__(256)
adam_clip_kernel(size_t n, float* p, const float* g, float* m, float* v, const double* sumsq,
                 float prescale, float clip, float lr_t_host, const float* lr_t_dev, float b1,
                 float b2, float eps, const unsigned* err, const float* fail_slot, unsigned long long* counters,
                 unsigned long long* mirror) {
    const bool bad = (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ||
                     (fail_slot && !(fail_slot[0] == 0.f));          // NaN counts as set
    if (counters && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long a = counters[0] + (bad ? 0ULL : 1ULL), k = counters[1] + (bad ? 1ULL : 0ULL);
        counters[0] = a;
        counters[1] = k;
        if (mirror) {
            mirror[0] = a;
            mirror[1] = k;
        }
    }
    if (bad) return;
    const float lr_t = lr_t_dev ? lr_t_dev[0] : lr_t_host;
    const double norm = sqrt(sumsq[0]);
    // [TF-1.3] clip_by_global_norm: g * clip / max(norm, clip)
    const float scale = prescale * (float)((double)clip / (norm > (double)clip ? norm : (double)clip));
    const size_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (size_t i = blockIdx.x * 256UL + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256UL) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#define D2P_ADAM1(c)                                              \
        {                                                         \
            const float gs = gg.c * scale;                        \
            mm.c = b1 * mm.c + (1.f - b1) * gs;                   \
            vv.c = b2 * vv.c + (1.f - b2) * gs * gs;              \
            pp.c -= lr_t * mm.c / (sqrtf(vv.c) + eps);            \
        }
        D2P_ADAM1(x) D2P_ADAM1(y) D2P_ADAM1(z) D2P_ADAM1(w)
#undef D2P_ADAM1
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        const float gs = g[i] * scale;
        const float mn = b1 * m[i] + (1.f - b1) * gs;
        const float vn = b2 * v[i] + (1.f - b2) * gs * gs;
        m[i] = mn; v[i] = vn;
        p[i] -= lr_t * mn / (sqrtf(vn) + eps);
    }
}

extern "C" int d2p_adam_clip_flat(size_t n, float* p, const float* g, float* m, float* v,
                                  const double* sumsq, float prescale, float clip, float lr_t,
                                  const float* lr_t_dev, float beta1, float beta2, float eps,
                                  d2p_stream_t stream) {
    if (n == 0) return D2P_OK;
    D2P_REQUIRE(p && g && m && v && sumsq, D2P_EINVAL, "adam: null pointer");
    D2P_REQUIRE(((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0), D2P_EALIGN,
                "adam: buffers must be 16-byte aligned");
    size_t want = (n / 4 + 255) / 256;
    int nb = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(adam_clip_kernel, dim3(nb), dim3(256), 0, as_stream(stream), n, p, g, m, v,
                       sumsq, prescale, clip, lr_t, lr_t_dev, beta1, beta2, eps, (const unsigned*)nullptr,
                       (const float*)nullptr, (unsigned long long*)nullptr, (unsigned long long*)nullptr);
    D2P_LAUNCH_CHECK("adam_clip");
    return D2P_OK;
}

extern "C" int d2p_adam_clip_flat_guarded(size_t n, float* p, const float* g, float* m, float* v,
                                          const double* sumsq, float prescale, float clip, float lr_t,
                                          const float* lr_t_dev, float beta1, float beta2, float eps,
                                          const float* fail_slot, unsigned long long* counters,
                                          unsigned long long* mirror, d2p_stream_t stream) {
    D2P_REQUIRE(counters, D2P_EINVAL, "adam guarded: null counters");
    D2P_REQUIRE(n == 0 || (p && g && m && v && sumsq), D2P_EINVAL, "adam: null pointer");
    D2P_REQUIRE(((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0), D2P_EALIGN,
                "adam: buffers must be 16-byte aligned");
    size_t want = (n / 4 + 255) / 256;
    int nb = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(adam_clip_kernel, dim3(nb), dim3(256), 0, as_stream(stream), n, p, g, m, v,
                       sumsq, prescale, clip, lr_t, lr_t_dev, beta1, beta2, eps,
                       (const unsigned*)d2p_persist_err_ptr(), fail_slot, counters, mirror);
    D2P_LAUNCH_CHECK("adam_clip_guarded");
    return D2P_OK;
}
