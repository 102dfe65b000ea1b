// lv_gemm_f32.hip -- exact-f32 MFMA GEMM for the dense contractions of the LSTM-VAE hot path.
//
// Replaces, on the aggressive inner step, every aten::mm / aten::addmm the reference reaches through
// nn.LSTM's input projection (modules/encoders/enc_lstm.py:60, modules/decoders/dec_lstm.py:104),
// nn.Linear (enc_lstm.py:62, dec_lstm.py:99,109) and their autograd backward (text.py:384).
//
// Design (gfx950): 128x128x16 workgroup tile, 4 waves as 2x2, each wave 64x64 = 2x2 v_mfma_f32_32x32x2_f32
// accumulators (exact f32: a k-ordered fmaf chain, so parity with the CPU oracle is f32-roundoff class).
// Operands are staged K-major in LDS (As[k][m], Bs[k][n], row pitch 132 floats) so a fragment read is 32
// consecutive floats per half-wave (conflict-free ds_read_b32) whatever the global layout; K-contiguous
// global operands are transposed on the LDS write (2-way write conflict = free on gfx950), M/N-contiguous
// ones are copied with ds_write_b128.  Global->LDS is register-staged and double-buffered (one barrier per
// K-tile).  blockIdx is remapped XCD-aware (block b runs on XCD b%8; each XCD gets a contiguous span of the
// grouped tile order so neighbouring tiles share A/B panels in that XCD's private L2).
#include "lv_device.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = BM + 4;

struct GemmP {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    float alpha;
    int accumulate;
    const float* add1; long ld1; int mod1;
    const float* add2; long ld2; int mod2;
    int tilesM, tilesN;
};

__device__ __forceinline__ float4 lv_load4(const float* __restrict__ base, long row, long col, long ld,
                                           long nrows, long ncols, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows && col < ncols) {
        const float* p = base + row * ld + col;
        if (vec && col + 3 < ncols) {
            v = *reinterpret_cast<const float4*>(p);
        } else {
            v.x = p[0];
            if (col + 1 < ncols) v.y = p[1];
            if (col + 2 < ncols) v.z = p[2];
            if (col + 3 < ncols) v.w = p[3];
        }
    }
    return v;
}

// KC = operand is contiguous along the contraction index (stored [rows][K]); otherwise stored [K][rows].
template <bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int rows, int K, int r0, int k0,
                                          bool vec, int t, float4 (&reg)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int f = t + 256 * i;
        if (KC) {
            int m = f >> 2, kq = f & 3;
            reg[i] = lv_load4(P, r0 + m, k0 + 4 * kq, ld, rows, K, vec);
        } else {
            int k = f >> 5, mq = f & 31;
            reg[i] = lv_load4(P, k0 + k, r0 + 4 * mq, ld, K, rows, vec);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float (*S)[LDT], int t, const float4 (&reg)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int f = t + 256 * i;
        if (KC) {
            int m = f >> 2, kq = f & 3;
            S[4 * kq + 0][m] = reg[i].x;
            S[4 * kq + 1][m] = reg[i].y;
            S[4 * kq + 2][m] = reg[i].z;
            S[4 * kq + 3][m] = reg[i].w;
        } else {
            int k = f >> 5, mq = f & 31;
            *reinterpret_cast<float4*>(&S[k][4 * mq]) = reg[i];
        }
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void lv_gemm_f32_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDT];

    // XCD-aware bijective remap + grouped (8 row-tiles) ordering.
    const int nblk = p.tilesM * p.tilesN;
    const int bid = (int)blockIdx.x;
    const int xcd = bid % 8, q = nblk / 8, r = nblk % 8;
    const int s = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    const int G = 8;
    const int nig = G * p.tilesN;
    const int group = s / nig;
    const int first_m = group * G;
    const int gsz = (p.tilesM - first_m) < G ? (p.tilesM - first_m) : G;
    const int tm = first_m + (s % nig) % gsz;
    const int tn = (s % nig) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int t = (int)threadIdx.x;
    const int l = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;

    const bool vecA = (p.lda % 4 == 0) && ((((uintptr_t)p.A) & 15) == 0);
    const bool vecB = (p.ldb % 4 == 0) && ((((uintptr_t)p.B) & 15) == 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[2], rb[2];
    const int nk = (p.K + BK - 1) / BK;
    load_tile<A_KC>(p.A, p.lda, p.M, p.K, m0, 0, vecA, t, ra);
    load_tile<B_KC>(p.B, p.ldb, p.N, p.K, n0, 0, vecB, t, rb);
    store_tile<A_KC>(As[0], t, ra);
    store_tile<B_KC>(Bs[0], t, rb);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            load_tile<A_KC>(p.A, p.lda, p.M, p.K, m0, (kt + 1) * BK, vecA, t, ra);
            load_tile<B_KC>(p.B, p.ldb, p.N, p.K, n0, (kt + 1) * BK, vecB, t, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (l >> 5);
            const float a0 = As[buf][kr][wm * 64 + (l & 31)];
            const float a1 = As[buf][kr][wm * 64 + 32 + (l & 31)];
            const float b0 = Bs[buf][kr][wn * 64 + (l & 31)];
            const float b1 = Bs[buf][kr][wn * 64 + 32 + (l & 31)];
            acc[0][0] = lv_mfma_32x32x2(a0, b0, acc[0][0]);
            acc[0][1] = lv_mfma_32x32x2(a0, b1, acc[0][1]);
            acc[1][0] = lv_mfma_32x32x2(a1, b0, acc[1][0]);
            acc[1][1] = lv_mfma_32x32x2(a1, b1, acc[1][1]);
        }
        if (kt + 1 < nk) {
            store_tile<A_KC>(As[buf ^ 1], t, ra);
            store_tile<B_KC>(Bs[buf ^ 1], t, rb);
        }
        __syncthreads();
    }

    // Epilogue: D[row=(e&3)+8*(e>>2)+4*(l>>5)][col=l&31] per 32x32 accumulator.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (l & 31);
            if (col >= p.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
                if (row >= p.M) continue;
                float v = p.alpha * acc[i][j][e];
                if (p.add1) v += p.add1[(long)(row % p.mod1) * p.ld1 + col];
                if (p.add2) v += p.add2[(long)(row % p.mod2) * p.ld2 + col];
                float* c = p.C + (long)row * p.ldc + col;
                if (p.accumulate) v += *c;
                *c = v;
            }
        }
}

}  // namespace

// C[M,N] (ldc) = alpha * op(A)[M,K] * op(B)[K,N]  (+ add1[(row % mod1)*ld1 + col]) (+ add2[...]) (+ C if accumulate)
// transA = 0: A stored [M][K] (lda >= K);  transA = 1: A stored [K][M] (lda >= M)
// transB = 0: B stored [K][N] (ldb >= N);  transB = 1: B stored [N][K] (ldb >= K)
extern "C" int lv_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                           const float* A, long lda, const float* B, long ldb,
                           float* C, long ldc, int accumulate,
                           const float* add1, long ld1, int mod1,
                           const float* add2, long ld2, int mod2, void* stream) {
    if (M < 0 || N < 0 || K < 0) return LV_ERR_SHAPE;
    if (M == 0 || N == 0) return LV_OK;
    if (!A || !B || !C) return LV_ERR_ARG;
    if ((add1 && mod1 <= 0) || (add2 && mod2 <= 0)) return LV_ERR_ARG;
    if (lda < (transA ? M : K) || ldb < (transB ? K : N) || ldc < N) return LV_ERR_SHAPE;
    GemmP p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.alpha = alpha; p.accumulate = accumulate;
    p.add1 = add1; p.ld1 = ld1; p.mod1 = mod1 > 0 ? mod1 : 1;
    p.add2 = add2; p.ld2 = ld2; p.mod2 = mod2 > 0 ? mod2 : 1;
    p.tilesM = lv_cdiv(M, BM); p.tilesN = lv_cdiv(N, BN);
    dim3 grid((unsigned)(p.tilesM * p.tilesN)), block(256);
    const bool akc = !transA, bkc = transB != 0;
    if (akc && bkc) LV_LAUNCH((lv_gemm_f32_kernel<true, true>), grid, block, 0, stream, p);
    else if (akc && !bkc) LV_LAUNCH((lv_gemm_f32_kernel<true, false>), grid, block, 0, stream, p);
    else if (!akc && bkc) LV_LAUNCH((lv_gemm_f32_kernel<false, true>), grid, block, 0, stream, p);
    else LV_LAUNCH((lv_gemm_f32_kernel<false, false>), grid, block, 0, stream, p);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
