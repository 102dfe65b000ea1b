// lv_gemm_bf16.hip -- bf16-MFMA GEMM (f32 storage in HBM, f32 accumulate) for the throughput configuration.
//
// Same contract and call sites as lv_gemm_f32 (lv_gemm_f32.hip): the dense contractions of the LSTM-VAE step.
// BASELINE.json's GPU configurations are bf16 ("Yelp/Yahoo LSTM-VAE bf16"); parity (1e-4, f32) is proven on the f32
// kernel, this one trades input precision for the 16x faster matrix pipe: operands are read as f32 from HBM,
// rounded to bf16 (RNE) while being staged into LDS, multiplied on v_mfma_f32_32x32x16_bf16 and accumulated in f32;
// master weights, activations and gradients stay f32 in HBM, so nothing else in the pipeline changes.
//
// Tile (64*WT)^2 x 32, 4 waves 2x2.  LDS image is chunk-major: S[c][row] = 8 consecutive-k bf16 (16 B), chunk c of 4,
// row pitch 16 B, chunk pitch (BT+2)*16 B: a fragment read is ds_read_b128 with 32 consecutive rows per half-wave
// (conflict-free), the K-contiguous staging write (float4 -> 4 bf16 = ds_write_b64) is conflict-free thanks to the
// +2 row pad, the transposing write for [K][rows]-stored operands is 2-way.  Loads keep full 128 B lines per row
// (K-contiguous) or 256 B contiguous per row (rows-contiguous) -- see profiles/r01_microbench_stride_probe.txt.
#include "lv_device.h"

namespace {

constexpr int BK = 32;

struct GemmP {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    float alpha;
    int accumulate;
    const float* add1; long ld1; int mod1;
    const float* add2; long ld2; int mod2;
    int tilesM, tilesN;
    int splits, kt_per_split;
    float* ws;
};

// Stage one operand tile (BT rows x 32 k) into registers as packed bf16: U = 2*WT units per thread, 2 dwords each.
// KC  (stored [rows][K]): unit = (row m = f>>3, k-quad kq = f&7): one float4 along k.
// !KC (stored [K][rows]): unit = (row m = f % BT, k-quad kq = f / BT): four dwords from rows k..k+3.
template <bool KC, int WT>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int rows, int K, int r0, int k0,
                                          bool vec, int t, uint2 (&reg)[2 * WT]) {
    constexpr int BT = 64 * WT;
#pragma unroll
    for (int i = 0; i < 2 * WT; ++i) {
        const int f = t + 256 * i;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (KC) {
            const int m = f >> 3, kq = f & 7;
            const long row = r0 + m;
            const int k = k0 + 4 * kq;
            if (row < rows && k < K) {
                const float* p = P + row * ld + k;
                if (vec && k + 3 < K) {
                    const float4 q = *reinterpret_cast<const float4*>(p);
                    v0 = q.x; v1 = q.y; v2 = q.z; v3 = q.w;
                } else {
                    v0 = p[0];
                    if (k + 1 < K) v1 = p[1];
                    if (k + 2 < K) v2 = p[2];
                    if (k + 3 < K) v3 = p[3];
                }
            }
        } else {
            const int m = f % BT, kq = f / BT;
            const long col = r0 + m;
            const int k = k0 + 4 * kq;
            if (col < rows) {
                const float* p = P + (long)k * ld + col;
                if (k < K) v0 = p[0];
                if (k + 1 < K) v1 = p[ld];
                if (k + 2 < K) v2 = p[2 * ld];
                if (k + 3 < K) v3 = p[3 * ld];
            }
        }
        reg[i].x = lv_pack_bf16x2(v0, v1);
        reg[i].y = lv_pack_bf16x2(v2, v3);
    }
}

// Branch-free staging for the common case (see lv_gemm_f32.hip: 16-byte aligned K-contiguous operand with clamped rows,
// or a [K][rows] operand whose tile lies inside it; complete K tile): unconditional loads, conversion as above.
template <bool KC, int WT>
__device__ __forceinline__ void load_tile_fast(const float* __restrict__ P, long ld, int rows, int r0, int k0, int t,
                                               uint2 (&reg)[2 * WT]) {
    constexpr int BT = 64 * WT;
#pragma unroll
    for (int i = 0; i < 2 * WT; ++i) {
        const int f = t + 256 * i;
        float v0, v1, v2, v3;
        if (KC) {
            int row = r0 + (f >> 3);
            if (row > rows - 1) row = rows - 1;
            const float4 q = *reinterpret_cast<const float4*>(P + (long)row * ld + k0 + 4 * (f & 7));
            v0 = q.x; v1 = q.y; v2 = q.z; v3 = q.w;
        } else {
            const float* p = P + (long)(k0 + 4 * (f / BT)) * ld + r0 + (f % BT);
            v0 = p[0]; v1 = p[ld]; v2 = p[2 * ld]; v3 = p[3 * ld];
        }
        reg[i].x = lv_pack_bf16x2(v0, v1);
        reg[i].y = lv_pack_bf16x2(v2, v3);
    }
}

template <bool KC, int WT>
__device__ __forceinline__ void store_tile(uint4 (*S)[64 * WT + 2], int t, const uint2 (&reg)[2 * WT]) {
    constexpr int BT = 64 * WT;
#pragma unroll
    for (int i = 0; i < 2 * WT; ++i) {
        const int f = t + 256 * i;
        const int m = KC ? (f >> 3) : (f % BT);
        const int kq = KC ? (f & 7) : (f / BT);
        uint2* dst = reinterpret_cast<uint2*>(&S[kq >> 1][m]) + (kq & 1);
        *dst = reg[i];
    }
}

template <bool A_KC, bool B_KC, int WT>
__global__ __launch_bounds__(256) void lv_gemm_bf16_kernel(GemmP p) {
    constexpr int BT = 64 * WT;
    __shared__ __attribute__((aligned(16))) uint4 As[2][4][BT + 2];
    __shared__ __attribute__((aligned(16))) uint4 Bs[2][4][BT + 2];

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
    const int li = l & 31, lh = l >> 5;

    const bool vecA = (p.lda % 4 == 0) && ((((uintptr_t)p.A) & 15) == 0);
    const bool vecB = (p.ldb % 4 == 0) && ((((uintptr_t)p.B) & 15) == 0);

    f32x16 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint2 ra[2 * WT], rb[2 * WT];
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = (int)blockIdx.y * p.kt_per_split;
    int kt1 = kt0 + p.kt_per_split;
    if (kt1 > nk_all) kt1 = nk_all;
    const int nfull = p.K / BK;                                       // K tiles [0, nfull) are complete
    const bool fastA = A_KC ? vecA : (m0 + BT <= p.M);
    const bool fastB = B_KC ? vecB : (n0 + BT <= p.N);
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
        for (int ks = 0; ks < 2; ++ks) {
            const int c = 2 * ks + lh;
            uint4 a[WT], b[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) a[i] = As[buf][c][wm * 32 * WT + i * 32 + li];
#pragma unroll
            for (int j = 0; j < WT; ++j) b[j] = Bs[buf][c][wn * 32 * WT + j * 32 + li];
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j) acc[i][j] = lv_mfma_32x32x16_bf16(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < kt1) {
            store_tile<A_KC, WT>(As[buf ^ 1], t, ra);
            store_tile<B_KC, WT>(Bs[buf ^ 1], t, rb);
        }
        __syncthreads();
    }

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
                if (p.add1) v += p.add1[(long)lv_wrap_row(q1, ro, p.mod1) * p.ld1 + col];
                if (p.add2) v += p.add2[(long)lv_wrap_row(q2, ro, p.mod2) * p.ld2 + col];
                if (p.accumulate) v += *c;
                *c = v;
            }
        }
}

__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(GemmP p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long MN = (long)p.M * p.N;
    if (idx >= MN) return;
    const int row = (int)(idx / p.N), col = (int)(idx % p.N);
    float s = 0.f;
    for (int k = 0; k < p.splits; ++k) s += p.ws[(long)k * MN + idx];
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
    if (akc && bkc) LV_LAUNCH((lv_gemm_bf16_kernel<true, true, WT>), grid, block, 0, stream, p);
    else if (akc && !bkc) LV_LAUNCH((lv_gemm_bf16_kernel<true, false, WT>), grid, block, 0, stream, p);
    else if (!akc && bkc) LV_LAUNCH((lv_gemm_bf16_kernel<false, true, WT>), grid, block, 0, stream, p);
    else LV_LAUNCH((lv_gemm_bf16_kernel<false, false, WT>), grid, block, 0, stream, p);
}

}  // namespace

// Same semantics and arguments as lv_gemm_f32; operands are rounded to bf16 on the way into the matrix pipe.
extern "C" int lv_gemm_bf16(int transA, int transB, int M, int N, int K, float alpha,
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
    // same tile policy as lv_gemm_f32 (traffic-bound: prefer 128x128 tiles + split-K over smaller tiles)
    int splits = 1;
    bool big = t128 >= 512;
    if (!big && ws && t128 >= 48 && nk >= 64) {
        long s = lv_cdiv(1024, t128);
        if (s > nk / 16) s = nk / 16;
        const long cap = ws_floats / ((long)M * N);
        if (s > cap) s = cap;
        if (s >= 2) { big = true; splits = (int)s; }
    }
    const int BT = big ? 128 : 64;
    p.tilesM = lv_cdiv(M, BT); p.tilesN = lv_cdiv(N, BT);
    const long tiles = (long)p.tilesM * p.tilesN;
    if (!big && ws && tiles < 256 && nk >= 8) {
        long s = lv_cdiv(512, tiles);
        if (s > nk / 4) s = nk / 4;
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
        LV_LAUNCH(splitk_reduce_bf16_kernel, dim3((unsigned)lv_cdiv((long)M * N, 256)), dim3(256), 0, stream, p);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
