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

constexpr int BK = 16;

struct GemmP {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    float alpha;
    int accumulate;
    const float* add1; long ld1; int mod1;
    const float* add2; long ld2; int mod2;
    int tilesM, tilesN;
    int splits, kt_per_split;      // split-K: blockIdx.y = slice, output slab ws[slice][M][N]
    float* ws;
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
// WT = 32-wide MFMA tiles per wave per dimension: block tile = (64*WT) x (64*WT), WT float4 per thread per operand.
template <bool KC, int WT>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int rows, int K, int r0, int k0,
                                          bool vec, int t, float4 (&reg)[WT]) {
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        const int f = t + 256 * i;
        if (KC) {
            const int m = f >> 2, kq = f & 3;
            reg[i] = lv_load4(P, r0 + m, k0 + 4 * kq, ld, rows, K, vec);
        } else {
            const int k = f / (16 * WT), mq = f % (16 * WT);
            reg[i] = lv_load4(P, k0 + k, r0 + 4 * mq, ld, K, rows, vec);
        }
    }
}

// Branch-free staging for the common case (16-byte aligned operand, complete K tile; for [K][rows] operands also a tile
// that lies completely inside the operand): addresses are resolved once per K tile and every load is an unconditional
// float4.  Rows of a K-contiguous operand beyond its last row are CLAMPED to it instead of predicated: an A row only
// reaches the C row of the same index (a B row the C column) and those are never written.
template <bool KC, int WT>
__device__ __forceinline__ void load_tile_fast(const float* __restrict__ P, long ld, int rows, int r0, int k0, int t,
                                               float4 (&reg)[WT]) {
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        const int f = t + 256 * i;
        if (KC) {
            int row = r0 + (f >> 2);
            if (row > rows - 1) row = rows - 1;
            reg[i] = *reinterpret_cast<const float4*>(P + (long)row * ld + k0 + 4 * (f & 3));
        } else {
            const int k = f / (16 * WT), mq = f % (16 * WT);
            reg[i] = *reinterpret_cast<const float4*>(P + (long)(k0 + k) * ld + r0 + 4 * mq);
        }
    }
}

template <bool KC, int WT>
__device__ __forceinline__ void store_tile(float (*S)[64 * WT + 4], int t, const float4 (&reg)[WT]) {
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        const int f = t + 256 * i;
        if (KC) {
            const int m = f >> 2, kq = f & 3;
            S[4 * kq + 0][m] = reg[i].x;
            S[4 * kq + 1][m] = reg[i].y;
            S[4 * kq + 2][m] = reg[i].z;
            S[4 * kq + 3][m] = reg[i].w;
        } else {
            const int k = f / (16 * WT), mq = f % (16 * WT);
            *reinterpret_cast<float4*>(&S[k][4 * mq]) = reg[i];
        }
    }
}

template <bool A_KC, bool B_KC, int WT>
__global__ __launch_bounds__(256) void lv_gemm_f32_kernel(GemmP p) {
    constexpr int BT = 64 * WT, LDT = BT + 4;
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
    const int m0 = tm * BT, n0 = tn * BT;

    const int t = (int)threadIdx.x;
    const int l = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;

    const bool vecA = (p.lda % 4 == 0) && ((((uintptr_t)p.A) & 15) == 0);
    const bool vecB = (p.ldb % 4 == 0) && ((((uintptr_t)p.B) & 15) == 0);

    f32x16 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[WT], rb[WT];
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = (int)blockIdx.y * p.kt_per_split;
    int kt1 = kt0 + p.kt_per_split;
    if (kt1 > nk_all) kt1 = nk_all;
    const int nfull = p.K / BK;                                       // K tiles [0, nfull) are complete
    const bool fastA = vecA && (A_KC || m0 + BT <= p.M);
    const bool fastB = vecB && (B_KC || n0 + BT <= p.N);
    auto stage = [&](int kt) {
        if (fastA && kt < nfull) load_tile_fast<A_KC, WT>(p.A, p.lda, p.M, m0, kt * BK, t, ra);
        else load_tile<A_KC, WT>(p.A, p.lda, p.M, p.K, m0, kt * BK, vecA, t, ra);
        if (fastB && kt < nfull) load_tile_fast<B_KC, WT>(p.B, p.ldb, p.N, n0, kt * BK, t, rb);
        else load_tile<B_KC, WT>(p.B, p.ldb, p.N, p.K, n0, kt * BK, vecB, t, rb);
    };
    stage(kt0);
    store_tile<A_KC, WT>(As[0], t, ra);
    store_tile<B_KC, WT>(Bs[0], t, rb);
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) stage(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (l >> 5);
            float a[WT], b[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) a[i] = As[buf][kr][wm * 32 * WT + i * 32 + (l & 31)];
#pragma unroll
            for (int j = 0; j < WT; ++j) b[j] = Bs[buf][kr][wn * 32 * WT + j * 32 + (l & 31)];
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j) acc[i][j] = lv_mfma_32x32x2(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < kt1) {
            store_tile<A_KC, WT>(As[buf ^ 1], t, ra);
            store_tile<B_KC, WT>(Bs[buf ^ 1], t, rb);
        }
        __syncthreads();
    }

    // Epilogue: D[row=(e&3)+8*(e>>2)+4*(l>>5)][col=l&31] per 32x32 accumulator.
    const bool split = p.splits > 1;
    float* const out = split ? p.ws + (long)blockIdx.y * p.M * p.N : p.C;
    const long ldo = split ? p.N : p.ldc;
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int col = n0 + wn * 32 * WT + j * 32 + (l & 31);
            if (col >= p.N) continue;
            const int rbase = m0 + wm * 32 * WT + i * 32 + 4 * (l >> 5);
            const int q1 = p.add1 ? rbase % p.mod1 : 0, q2 = p.add2 ? rbase % p.mod2 : 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);
                const int row = rbase + ro;
                if (row >= p.M) continue;
                float* c = out + (long)row * ldo + col;
                if (split) { *c = acc[i][j][e]; continue; }
                float v = p.alpha * acc[i][j][e];
                // (addends loaded in place: gathering them first, as lv_gemm_b16's epilogue does, costs this kernel 27-48 more live
                //  registers and a third to a half of its occupancy)
                if (p.add1) v += p.add1[(long)lv_wrap_row(q1, ro, p.mod1) * p.ld1 + col];
                if (p.add2) v += p.add2[(long)lv_wrap_row(q2, ro, p.mod2) * p.ld2 + col];
                if (p.accumulate) v += *c;
                *c = v;
            }
        }
}

template <int NF>
__device__ __forceinline__ float sum_pieces(const float* first, long stride, int n) {
    float s = 0.f;
    for (int k0 = 0; k0 < n; k0 += NF) {
        float pv[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) pv[u] = first[(long)(k0 + u < n ? k0 + u : 0) * stride];      // clamped: the surplus is never added
#pragma unroll
        for (int u = 0; u < NF; ++u)
            if (k0 + u < n) s += pv[u];
    }
    return s;
}

// C = alpha * sum_s ws[s] (+ addends) (+ C): fixed summation order -> deterministic
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmP p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long MN = (long)p.M * p.N;
    if (idx >= MN) return;
    const int row = (int)(idx / p.N), col = (int)(idx % p.N);
    // the pieces in flight together (2, 4 or 8 at a time by the split count), added in piece order
    float s = 0.f;
    if (p.splits <= 2) s = sum_pieces<2>(p.ws + idx, MN, p.splits);
    else if (p.splits <= 4) s = sum_pieces<4>(p.ws + idx, MN, p.splits);
    else s = sum_pieces<8>(p.ws + idx, MN, p.splits);
    float v = p.alpha * s;
    if (p.add1) v += p.add1[(long)(row % p.mod1) * p.ld1 + col];
    if (p.add2) v += p.add2[(long)(row % p.mod2) * p.ld2 + col];
    float* c = p.C + (long)row * p.ldc + col;
    if (p.accumulate) v += *c;
    *c = v;
}

template <int WT>
void launch_gemm(const GemmP& p, bool akc, bool bkc, void* stream) {
    dim3 grid((unsigned)(p.tilesM * p.tilesN), (unsigned)p.splits), block(256);
    if (akc && bkc) LV_LAUNCH((lv_gemm_f32_kernel<true, true, WT>), grid, block, 0, stream, p);
    else if (akc && !bkc) LV_LAUNCH((lv_gemm_f32_kernel<true, false, WT>), grid, block, 0, stream, p);
    else if (!akc && bkc) LV_LAUNCH((lv_gemm_f32_kernel<false, true, WT>), grid, block, 0, stream, p);
    else LV_LAUNCH((lv_gemm_f32_kernel<false, false, WT>), grid, block, 0, stream, p);
}

}  // namespace

// C[M,N] (ldc) = alpha * op(A)[M,K] * op(B)[K,N]  (+ add1[(row % mod1)*ld1 + col]) (+ add2[...]) (+ C if accumulate)
// transA = 0: A stored [M][K] (lda >= K);  transA = 1: A stored [K][M] (lda >= M)
// transB = 0: B stored [K][N] (ldb >= N);  transB = 1: B stored [N][K] (ldb >= K)
// ws / ws_floats: optional caller-owned scratch for deterministic split-K (used when the output has too few tiles to
// fill 256 CUs and K is long: wgrad GEMMs, the M = batch "skinny" GEMMs); NULL disables split-K.
extern "C" int lv_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                           const float* A, long lda, const float* B, long ldb,
                           float* C, long ldc, int accumulate,
                           const float* add1, long ld1, int mod1,
                           const float* add2, long ld2, int mod2,
                           float* ws, long ws_floats, void* stream) {
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
    p.ws = ws;
    const int nk = lv_cdiv(K, BK);
    const long t128 = (long)lv_cdiv(M, 128) * lv_cdiv(N, 128);
    // Tile choice.  Measured on MI355X these GEMMs are bound by L2-miss (Infinity Cache) traffic, which scales with
    // 1/tile: prefer 128x128 tiles and, when they are too few to fill 256 CUs but K is long, split K across
    // workgroups (deterministic slab reduce) instead of shrinking the tile; 64x64 tiles only for small outputs.
    int splits = 1;
    bool big = t128 >= 1024;
    if (!big && ws && t128 >= 48 && nk >= 128) {
        long s = lv_cdiv(1024, t128);
        if (s > nk / 32) s = nk / 32;
        const long cap = ws_floats / ((long)M * N);
        if (s > cap) s = cap;
        if (s >= 2) { big = true; splits = (int)s; }
    }
    const int BT = big ? 128 : 64;
    p.tilesM = lv_cdiv(M, BT); p.tilesN = lv_cdiv(N, BT);
    const long tiles = (long)p.tilesM * p.tilesN;
    if (!big && ws && tiles < 256 && nk >= 16) {
        long s = lv_cdiv(512, tiles);
        if (s > nk / 8) s = nk / 8;
        if (s > 64) s = 64;
        const long cap = ws_floats / ((long)M * N);
        if (s > cap) s = cap;
        if (s > 1) splits = (int)s;
    }
    p.kt_per_split = lv_cdiv(nk > 0 ? nk : 1, splits);
    splits = lv_cdiv(nk > 0 ? nk : 1, p.kt_per_split);
    p.splits = splits;
    const bool akc = !transA, bkc = transB != 0;
    if (big) launch_gemm<2>(p, akc, bkc, stream);
    else launch_gemm<1>(p, akc, bkc, stream);
    if (splits > 1)
        LV_LAUNCH(splitk_reduce_kernel, dim3((unsigned)lv_cdiv((long)M * N, 256)), dim3(256), 0, stream, p);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
