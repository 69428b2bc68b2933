// Micro-benchmark (round 5): can the LSTM gate math of a phase run on a SECOND wave of the SIMD that is busy with the
// phase's fp32 MFMA chain -- the question behind VERDICT round 4, item 1(a)?  Unlike mfma_valu_contention.hip (whose
// timed loops are a handful of instructions around a taken branch) this runs the real shapes: an MFMA wave issues the
// 128 v_mfma_f32_16x16x4_f32 of a backward phase (two accumulators) and writes its partial tile to LDS; a gate wave on the
// same SIMD reads four partial tiles + its cell's operands from LDS, runs the forward cell math of lstm_math.h (three
// sigmoids, two tanh: exp2 / rcp on the transcendental unit, ~45 VALU instructions with ILP ~4 inside a cell), writes
// h to LDS and c / h to memory.  One 512-thread workgroup = 2 waves per SIMD (wave w and w + 4 share SIMD w).
//
// mode bits: 1 = MFMA waves run, 2 = gate waves run, 4 = one workgroup barrier per phase (as the kernel would need),
//            8 = the MFMA waves do the gate math THEMSELVES behind their chain (today's non-deferred form; gate waves idle),
//            16 = gate waves do two cells per lane (two independent cells: more ILP)
// Build: hipcc --offload-arch=gfx950 -O3 mfma_gate_corun.hip -o mgc && ./mgc
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tnh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f; }

__device__ __forceinline__ void cell(const float* P, const float* ring, float* hl, float* gout, int lane, int ph) {
    // partial tiles [4 waves][16 rows][36] (+ gate stride), ring [5][256]
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float s = ring[g * 256 + lane];
#pragma unroll
        for (int w = 0; w < 4; ++w) s += P[(w * 16 + (lane >> 4) + 4 * g) * 36 + (lane & 15)];
        z[g] = s;
    }
    const float cp = ring[4 * 256 + lane];
    const float ij = sigm(z[0]) * tnh(z[1]);
    const float c1 = fmaf(cp, sigm(z[2] + 1.0f), ij);
    const float h = tnh(c1) * sigm(z[3]);
    hl[lane] = h;
    gout[(ph & 63) * 128 + lane] = c1;
    gout[(ph & 63) * 128 + 64 + lane] = h;
}

__global__ void __launch_bounds__(512) k(int mode, int phases, unsigned long long* out, float* sink) {
    __shared__ float P[4 * 16 * 36 + 64];
    __shared__ float ring[2][5 * 256];
    __shared__ float hl[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4 * 16 * 36 + 64; i += 512) P[i] = 0.001f * i;
    for (int i = threadIdx.x; i < 2 * 5 * 256; i += 512) (&ring[0][0])[i] = 0.002f * i - 1.0f;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    float r = 0.f;
    float* gout = sink + 64 + wave * 64 * 128;
    if (wave < 4) {
        if (mode & 1) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 1e-6f;
            t0 = __builtin_readcyclecounter();
#pragma unroll 1
            for (int ph = 0; ph < phases; ++ph) {
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) P[(wave * 16 + (lane >> 4) * 4 + q) * 36 + (lane & 15)] = (a0[q] + a1[q]) * 1e-30f;
                if (mode & 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
                if (mode & 8) {
                    cell(P, ring[ph & 1], hl[wave], gout, lane, ph);
                    if (mode & 4) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
                }
            }
            r = a0[0] + a1[1];
            asm volatile("" : "+v"(r));
            t1 = __builtin_readcyclecounter();
        } else if (mode & 4) {
            // (barrier partner for the gate waves when they run alone)
            for (int ph = 0; ph < phases; ++ph) { __builtin_amdgcn_s_barrier(); }
        }
    } else {
        if (mode & 2) {
            t0 = __builtin_readcyclecounter();
#pragma unroll 1
            for (int ph = 0; ph < phases; ++ph) {
                if (mode & 4) { __builtin_amdgcn_s_barrier(); }
                cell(P, ring[ph & 1], hl[wave - 4], gout, lane, ph);
                if (mode & 16) cell(P + 1, ring[(ph + 1) & 1], hl[wave - 4], gout + 32 * 128, lane, ph);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t1 = __builtin_readcyclecounter();
        } else if ((mode & 4) && (mode & 1)) {
            const int nb = (mode & 8) ? 2 * phases : phases;
            for (int ph = 0; ph < nb; ++ph) { __builtin_amdgcn_s_barrier(); }
        }
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 12345.678f) sink[0] = r;
}

int main() {
    unsigned long long* d; float* s;
    hipMalloc(&d, 8 * 8 * 256); hipMalloc(&s, (64 + 8 * 64 * 128) * 4);
    const int PH = 256;
    struct { const char* name; int mode; } cases[] = {
        {"MFMA waves alone (128 MFMA + partial-tile write per phase)", 1},
        {"gate waves alone, 1 cell per lane", 2},
        {"gate waves alone, 2 cells per lane", 2 | 16},
        {"MFMA + gate waves (no barrier), 1 cell per lane", 3},
        {"MFMA + gate waves (no barrier), 2 cells per lane", 3 | 16},
        {"MFMA alone, barrier per phase", 1 | 4},
        {"MFMA + gate waves, barrier per phase, 1 cell per lane", 3 | 4},
        {"MFMA + gate waves, barrier per phase, 2 cells per lane", 3 | 4 | 16},
        {"MFMA waves do the gate math themselves (no barrier)", 1 | 8},
        {"MFMA waves do the gate math themselves, two barriers per phase", 1 | 8 | 4}};
    for (int blocks = 1; blocks <= 256; blocks *= 256) {
        printf("---- %d workgroup(s) of 512 threads, %d phases; clocks per phase ----\n", blocks, PH);
        for (auto& c : cases) {
            unsigned long long h[8];
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, c.mode, PH, d, s);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-66s MFMA wave0 %8.1f  wave1 %8.1f | gate wave4 %8.1f  wave5 %8.1f\n", c.name, (double)h[0] / PH,
                   (double)h[1] / PH, (double)h[4] / PH, (double)h[5] / PH);
        }
    }
    return 0;
}
