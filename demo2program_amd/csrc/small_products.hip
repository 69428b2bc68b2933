// The decoders' SMALL gradient products in one launch (round 6; VERDICT round 5, item 5: "fourteen products under 0.2
// GFLOP at 8-19 us each").  Behind the backward recurrences of the three decoders (models/model_full.py:497-599: program,
// action, perception; layers_core.Dense projections :463-464, Token_Embedding :282-296, Per_Encoder :308-316) sat, per
// decoder, a chain of launches whose work is a few dozen MFLOP each and whose time is launch latency and a K walk by a
// handful of workgroups -- ~180 us of the side queue per step at config 2, which is what the end of the step waits for:
//   (the projections' weight gradients gproj [U, V] = hout^T dlogits, K = all step rows, stay three split-K GEMM launches +
//    combines: a grouped two-launch form -- 128-row slices through LDS, slices added in order -- measured 55 against 41 us)
//   * with S [R <= 256 rows, 4U] = the decoder's dz rows summed by input token / perception column:
//       G1 [U, 4U] = A^T S   (A = the embedding table or the perception rows' matrix H: the input half of the cell's kernel gradient)
//       G2 [R, U]  = S Wx^T  (the embedding gradient / the Q of d2p_per_fc_bn_bwd)
//     two GEMM launches per decoder                                       -> d2p_small_pair_products: one launch for all
// Deterministic (fixed summation orders, no atomics on data).
#include "common.h"
#include "prof.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SP_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define SP_MAXP 4               // problems per launch

namespace {

__device__ __forceinline__ f32x4 sp_ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// ---- the pair of small products behind the rows summed by key ------------------------------------------------------
struct SpProb {
    int R, U, N4;               // rows of S in use (<= 256), units, 4U
    int blk0, nblk1;            // first workgroup; workgroups of the G1 part (the G2 part follows)
    const float* S; const float* A; const float* Wx; float* G1; float* G2;
};
struct SpArgs {
    int n;
    SpProb p[SP_MAXP];
};

__global__ void __launch_bounds__(256) small_pair_products_kernel(SpArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 32 + 4 * 256];
    int d = 0;
#pragma unroll
    for (int i = 1; i < SP_MAXP; ++i) d = (i < a.n && (int)blockIdx.x >= a.p[i].blk0) ? i : d;
    SpProb q = a.p[0];
#pragma unroll
    for (int i = 1; i < SP_MAXP; ++i) q = (d == i) ? a.p[i] : q;
    const int b = (int)blockIdx.x - q.blk0;
    const int tid = threadIdx.x;
    if (b < q.nblk1) {
        // ---- G1[i, c] = sum_r A[r, i] S[r, c]: workgroup = (256 columns c, 32 rows i); per 64 rows of S a thread keeps
        //      its column in registers, the A tile sits transposed in LDS ([i][r]: every lane reads the same 16 bytes)
        const int nct = q.N4 / 256;
        const int it = b / nct, ct = b - it * nct;
        float* At = lds;                                        // [32][64]
        const int c = ct * 256 + tid;
        float acc[32];
#pragma unroll
        for (int ii = 0; ii < 32; ++ii) acc[ii] = 0.f;
        for (int r0 = 0; r0 < q.R; r0 += 64) {
            __syncthreads();
            for (int i = tid; i < 32 * 64; i += 256) {
                const int ii = i >> 6, r = r0 + (i & 63);
                At[i] = r < q.R ? q.A[(long)r * q.U + it * 32 + ii] : 0.f;
            }
            float sreg[64];
#pragma unroll
            for (int r = 0; r < 64; ++r) sreg[r] = q.S[(long)min(r0 + r, q.R - 1) * q.N4 + c];   // (rows >= R meet zeros of At)
            __syncthreads();
#pragma unroll
            for (int ii = 0; ii < 32; ++ii) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int r = 0; r < 64; r += 8) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(At + ii * 64 + r);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(At + ii * 64 + r + 4);
                    s0 += a0.x * sreg[r] + a0.y * sreg[r + 1] + a0.z * sreg[r + 2] + a0.w * sreg[r + 3];
                    s1 += a1.x * sreg[r + 4] + a1.y * sreg[r + 5] + a1.z * sreg[r + 6] + a1.w * sreg[r + 7];
                }
                acc[ii] += s0 + s1;
            }
        }
#pragma unroll
        for (int ii = 0; ii < 32; ++ii) q.G1[(long)(it * 32 + ii) * q.N4 + c] = acc[ii];
        return;
    }
    // ---- G2[r, i] = sum_c S[r, c] Wx[i, c]: workgroup = one 16 x 16 tile, K = 4U split over the four waves
    //      (v_mfma_f32_16x16x4_f32; a lane's float4 = 4 consecutive c, component j feeds MFMA j), partial tiles added in
    //      wave order through LDS
    const int b2 = b - q.nblk1;
    const int nit = q.U / 16;
    const int rt = b2 / nit, itile = b2 - rt * nit;
    const int lane = tid & 63, wave = tid >> 6, p = lane & 15, qq = lane >> 4;
    const int rrow = min(rt * 16 + p, q.R - 1);
    const float* sp = q.S + (long)rrow * q.N4 + 4 * qq;
    const float* wp = q.Wx + (long)(itile * 16 + p) * q.N4 + 4 * qq;
    const int kper = q.N4 / 4;                                  // this wave's share of K
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    // (eight 16-deep chunks of both operands in flight at a time: a chunk-by-chunk loop waits out a memory round trip per
    //  8 MFMAs -- 50 us for the whole launch)
    for (int k = wave * kper; k < (wave + 1) * kper; k += 128) {
        f32x4 sv[8], wv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            sv[c] = sp_ldg4(sp + k + 16 * c);
            wv[c] = sp_ldg4(wp + k + 16 * c);
        }
#pragma unroll
        for (int c = 0; c < 8; c += 2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = SP_MFMA16(sv[c][j], wv[c][j], acc0);
                acc1 = SP_MFMA16(sv[c + 1][j], wv[c + 1][j], acc1);
            }
    }
    float* red = lds + 64 * 32;                                 // [4 waves][64 lanes][4]
    *reinterpret_cast<f32x4*>(red + (wave * 64 + lane) * 4) = acc0 + acc1;
    __syncthreads();
    if (wave == 0) {
        f32x4 t = *reinterpret_cast<const f32x4*>(red + lane * 4);
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) t += *reinterpret_cast<const f32x4*>(red + (wv * 64 + lane) * 4);
        // D[i = 4 qq + r][j = p] = G2[row rt*16 + 4 qq + r][unit itile*16 + p]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + 4 * qq + r;
            if (row < q.R) q.G2[(long)row * q.U + itile * 16 + p] = t[r];
        }
    }
}

}   // namespace

extern "C" int d2p_small_pair_products(int n, const d2p_pair_products_desc* d, d2p_stream_t stream) {
    D2P_REQUIRE(n >= 1 && n <= SP_MAXP && d, D2P_EINVAL, "pair products: 1..%d problems", SP_MAXP);
    SpArgs a;
    a.n = n;
    int blk = 0;
    double fl = 0.0;
    for (int i = 0; i < n; ++i) {
        D2P_REQUIRE(d[i].R >= 1 && d[i].R <= 256 && d[i].U >= 32 && d[i].U % 32 == 0 && d[i].N4 >= 512 && d[i].N4 % 512 == 0,
                    D2P_EINVAL, "pair products: problem %d: R=%d U=%d N4=%d (R <= 256, U a multiple of 32, N4 of 512)", i, d[i].R,
                    d[i].U, d[i].N4);
        D2P_REQUIRE(d[i].S && d[i].A && d[i].Wx && d[i].G1 && d[i].G2, D2P_EINVAL, "pair products: null pointer");
        D2P_REQUIRE((((uintptr_t)d[i].S | (uintptr_t)d[i].Wx) & 15) == 0, D2P_EALIGN, "pair products: S, Wx must be 16-byte aligned");
        SpProb& q = a.p[i];
        q.R = d[i].R; q.U = d[i].U; q.N4 = d[i].N4;
        q.S = d[i].S; q.A = d[i].A; q.Wx = d[i].Wx; q.G1 = d[i].G1; q.G2 = d[i].G2;
        q.blk0 = blk;
        q.nblk1 = (q.N4 / 256) * (q.U / 32);
        blk += q.nblk1 + ((q.R + 15) / 16) * (q.U / 16);
        fl += 2.0 * q.R * q.U * q.N4 * 2;
    }
    for (int i = n; i < SP_MAXP; ++i) { a.p[i] = a.p[0]; a.p[i].blk0 = 0x7fffffff; }
    hipStream_t st = as_stream(stream);
    D2pProfScope prof(st, D2P_PROF_GEMM, fl);
    hipLaunchKernelGGL(small_pair_products_kernel, dim3(blk), dim3(256), 0, st, a);
    D2P_LAUNCH_CHECK("small_pair_products");
    return D2P_OK;
}
