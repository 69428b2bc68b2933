// Micro-benchmark: does an fp32 MFMA stream (v_mfma_f32_16x16x4_f32) leave VALU issue slots to a
// second wave on the same SIMD?  One 512-thread workgroup (2 waves per SIMD): waves 0-3 run an MFMA
// chain (mode bit 0), waves 4-7 a VALU chain (bit 1: dependent fma chain; bit 2: 4 independent
// chains; bit 3: transcendental chain), each timed with s_memtime.  Build: hipcc --offload-arch=gfx950
// -O3 mfma_valu_contention.hip -o mvc && ./mvc
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k(int mode, int prio, int n_mfma, int n_valu, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    unsigned long long t0 = 0, t1 = 0;
    float r = 0.f;
    __syncthreads();
    if (wave < 4) {
        if (mode & 1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 1e-6f;
            t0 = __builtin_readcyclecounter();
            for (int i = 0; i < n_mfma; i += 2) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            }
            r = a0[0] + a1[1];
            asm volatile("" : "+v"(r));
            t1 = __builtin_readcyclecounter();
        }
    } else {
        if (prio) __builtin_amdgcn_s_setprio(3);
        float v0 = threadIdx.x * 0.5f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
        const float m = 1.0000001f, c = 1e-7f;
        if (mode & 2) {
            t0 = __builtin_readcyclecounter();
            for (int i = 0; i < n_valu; ++i) v0 = fmaf(v0, m, c);
            asm volatile("" : "+v"(v0));
            t1 = __builtin_readcyclecounter();
        } else if (mode & 4) {
            t0 = __builtin_readcyclecounter();
            for (int i = 0; i < n_valu; i += 4) {
                v0 = fmaf(v0, m, c); v1 = fmaf(v1, m, c); v2 = fmaf(v2, m, c); v3 = fmaf(v3, m, c);
            }
            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            t1 = __builtin_readcyclecounter();
        } else if (mode & 8) {
            t0 = __builtin_readcyclecounter();
            for (int i = 0; i < n_valu; i += 2) v0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v0));
            asm volatile("" : "+v"(v0));
            t1 = __builtin_readcyclecounter();
        }
        r = v0 + v1 + v2 + v3;
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 12345.678f) sink[0] = r;
}

int main() {
    unsigned long long* d; float* s;
    hipMalloc(&d, 8 * 8); hipMalloc(&s, 4);
    const int NM = 4096, NV = 2048;
    struct { const char* name; int mode, prio; } cases[] = {
        {"MFMA alone", 1, 0}, {"VALU dependent alone", 2, 0}, {"VALU 4-way ILP alone", 4, 0}, {"VALU exp/rcp chain alone", 8, 0},
        {"MFMA + VALU dependent", 3, 0}, {"MFMA + VALU dependent, prio 3", 3, 1},
        {"MFMA + VALU 4-way ILP", 5, 0}, {"MFMA + VALU 4-way ILP, prio 3", 5, 1},
        {"MFMA + exp/rcp chain", 9, 0}, {"MFMA + exp/rcp chain, prio 3", 9, 1}};
    for (auto& c : cases) {
        unsigned long long h[8];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, c.mode, c.prio, NM, NV, d, s);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-34s MFMA wave0: %7.1f clk/MFMA   VALU wave4: %7.2f clk/op\n", c.name, (double)h[0] / NM, (double)h[4] / NV);
    }
    return 0;
}
