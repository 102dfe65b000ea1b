// lv_gemm_b16.hip -- bf16-MFMA GEMM over operands that are ALREADY bf16 in HBM (f32 accumulate, f32 output).
//
// The three vocabulary-sized contractions of the decoder (logits = O.Wp^T, dO = dlogits.Wp, dWp = dlogits^T.O;
// reference modules/decoders/dec_lstm.py:117,140-146 and their autograd) are L2->LDS-traffic- and latency-bound in
// lv_gemm_bf16 (f32 sources: 32 KB of loads per 128x128x32 step, 5 integer ops per element to round).  Here the
// producers hand over bf16 images -- lv_cvt_bf16_f32 for the small operands (O, Wp and their transposes),
// lv_softmax_nll_bwd_b16 for dlogits -- with exactly the rounding lv_gemm_bf16 applies on the fly (RNE), so the
// two routes give bit-identical results while this one moves half the bytes, converts nothing, and takes K in
// steps of 64 (16 MFMAs per wave between a tile's loads and its LDS store).
//
// C[M,N] = alpha * A . B^T-form (+ epilogue), B stored [N][K] (K-contiguous) always;
// A stored [M][K] (transA = 0) or [K][M] (transA = 1: the weight gradient reads dlogits as it lies).
//
// Tile 128x128x64, 4 waves 2x2, wave tile 64x64 = 2x2 v_mfma_f32_32x32x16_bf16.  LDS image S[c][row ^ 2c] = 8
// consecutive-k bf16 of one row (16 B), chunk c of 8, pitch 128 rows: 2 operands x 2 buffers = 64 KB, 2 WG/CU.
// The XOR keeps the K-contiguous staging write (16 lanes = 8 chunks x 2 rows -> 64 distinct banks) and the fragment
// read (16 lanes = 16 consecutive rows of one chunk) conflict-free without padding.  For A stored [K][M] a thread
// loads 16 B (eight adjacent rows at one k) for 4 consecutive k and transposes the 8x4 block in registers (16 integer
// ops) into the eight rows' k-quads; row e of octet m8 is kept in LDS row 16*e + m8 so that the lanes of one store hit
// consecutive LDS rows, and the epilogue undoes that permutation for free in its row index.
#ifndef LV_B16_T256_DMA
#define LV_B16_T256_DMA 1
#endif
#if defined(LV_TRACE)
#define lv_trace_buf lv_trace_buf_gemm      // (one trace pointer per translation unit: device code is not relocatable)
#endif
#include "lv_device.h"
#if defined(LV_TRACE) && !defined(LV_EMU)
__device__ unsigned long long* lv_trace_buf = nullptr;
extern "C" int lv_trace_set_gemm(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(lv_trace_buf), &p, sizeof(p)); }
#endif
#include <type_traits>

namespace {

#ifndef LV_B16_ABL
#define LV_B16_ABL 0      // ablation switches for profiles/microbench only: 1 = no MFMA, 2 = no global loads in the loop, 4 = no fragment reads
#endif
#ifndef LV_B16_SPLIT_TARGET
#define LV_B16_SPLIT_TARGET 512     // workgroups a split-K launch aims for (2 per CU x 2 rounds); A/B knob of the microbench
#endif

constexpr int BK = 64;
constexpr int BT = 128;
constexpr int NCH = BK / 8;

#ifndef LV_NLL_ABL
#define LV_NLL_ABL 0                     // measurement only (profiles/microbench/gemm_pp_probe.py): 1 = no logits store, 2 = no statistics, 4 = no epilogue
#endif
struct GemmQ {
    const uint16_t* A; const uint16_t* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    float alpha;
    int accumulate;
    const float* add1; long ld1; int mod1;
    const float* add2; long ld2; int mod2;
    int tilesM, tilesN;
    int splits, kt_per_split;
    float* ws;
    // fused vocabulary projection + token-NLL statistics (lv_gemm_b16_nll): C is written as binary16 and never as f32
    uint16_t* C16; long ldc16;
    const int64_t* ids; long ids_stride; int tgt_off; int Bsz;
    float2* part; int nparts;          // [M][nparts] (max, sum exp(x - max)) of each 64-column piece of a row
    float* tgt;                        // [M] the target token's logit
    // nn.Dropout backward folded into the reduction stage (lv_gemm_b16_keep): C *= keep[(row % Bsz) * keepT + row / Bsz][col] ? kscale : 0
    const uint8_t* keep; float kscale; int keepT;
    // sum of squares of the product emitted by the kernels that hold its final values (lv_gemm_b16_sumsq: the 256 x 256 tile's
    // epilogue and the tail reduce; one partial per wave, fixed slots: deterministic); sq_only: C itself is not written
    float* sq; int sq_only;
    // two destinations (lv_gemm_b16_dual: split-K products only): columns >= nsplit of the product go to C2 [M][ldc2] (column - nsplit)
    float* C2; long ldc2; int nsplit;
};

__device__ __forceinline__ uint4 load_chunk(const uint16_t* __restrict__ p, int valid) {
    if (valid >= 8) return *reinterpret_cast<const uint4*>(p);
    uint32_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = e < valid ? (uint32_t)p[e] : 0u;
    uint4 q;
    q.x = h[0] | (h[1] << 16); q.y = h[2] | (h[3] << 16); q.z = h[4] | (h[5] << 16); q.w = h[6] | (h[7] << 16);
    return q;
}

// Staging is branch-free inside the K loop: every thread resolves its addresses once (rows beyond the operand are
// CLAMPED to a valid row rather than predicated -- an A row only ever reaches the C row of the same index and a B row
// the C column, and those are not written), full K tiles are loaded unconditionally (FULL), and only the single
// ragged tile at the end of K takes the predicated path that zero-fills k >= K.

// K-contiguous operand ([rows][K]): unit i of thread t = (row m = f>>3, chunk c = f&7), f = t + 256 i: one 16 B load,
// 8 lanes cover a row's 128 B.
struct KcPtr { const uint16_t* p[4]; };

__device__ __forceinline__ KcPtr kc_setup(const uint16_t* __restrict__ P, long ld, int rows, int r0, int t) {
    KcPtr q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = t + 256 * i;
        int row = r0 + (f >> 3);
        if (row > rows - 1) row = rows - 1;
        q.p[i] = P + (long)row * ld + 8 * (f & 7);
    }
    return q;
}

template <bool FULL>
__device__ __forceinline__ void kc_load(const KcPtr& q, int k0, int K, int t, uint4 (&reg)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (FULL) {
            reg[i] = *reinterpret_cast<const uint4*>(q.p[i] + k0);
        } else {
            const int k = k0 + 8 * ((t + 256 * i) & 7);
            reg[i] = k < K ? load_chunk(q.p[i] + k0, K - k) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

__device__ __forceinline__ void store_kc(uint4 (*S)[BT], int t, const uint4 (&reg)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = t + 256 * i;
        const int m = f >> 3, c = f & 7;
        S[c][m ^ (2 * c)] = reg[i];
    }
}

// rows-contiguous operand ([K][rows]): one unit per thread = (row octet m8 = t&15 -> rows 8*m8 .. 8*m8+7; k-quad
// kq = t>>4), four 16 B loads (k .. k+3; a wave-level load = 4 k-rows x 256 contiguous bytes), transposed in registers
// into the 8 rows' k-quads.
struct McPtr { const uint16_t* p; long ld; };

__device__ __forceinline__ McPtr mc_setup(const uint16_t* __restrict__ P, long ld, int r0, int t) {
    long col = r0 + 8 * (t & 15);
    if (col > ld - 8) col = ld - 8;                 // stay inside the row pitch (ld % 8 == 0)
    McPtr q;
    q.p = P + col + (long)(4 * (t >> 4)) * ld;
    q.ld = ld;
    return q;
}

template <bool FULL>
__device__ __forceinline__ void mc_load(const McPtr& q, int k0, int K, int t, uint4 (&reg)[4]) {
    const uint16_t* base = q.p + (long)k0 * q.ld;
    const int k = k0 + 4 * (t >> 4);
    uint4 d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (FULL || k + j < K) d[j] = *reinterpret_cast<const uint4*>(base + (long)j * q.ld);
        else d[j] = make_uint4(0u, 0u, 0u, 0u);
    }
    const uint32_t w[4][4] = {{d[0].x, d[1].x, d[2].x, d[3].x}, {d[0].y, d[1].y, d[2].y, d[3].y},
                              {d[0].z, d[1].z, d[2].z, d[3].z}, {d[0].w, d[1].w, d[2].w, d[3].w}};
    uint32_t o[16];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {                                    // row pair (2*pr, 2*pr+1) of the octet
        o[4 * pr + 0] = (w[pr][0] & 0xFFFFu) | (w[pr][1] << 16);        // even row: k, k+1
        o[4 * pr + 1] = (w[pr][2] & 0xFFFFu) | (w[pr][3] << 16);        //           k+2, k+3
        o[4 * pr + 2] = (w[pr][0] >> 16) | (w[pr][1] & 0xFFFF0000u);    // odd row : k, k+1
        o[4 * pr + 3] = (w[pr][2] >> 16) | (w[pr][3] & 0xFFFF0000u);    //           k+2, k+3
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) reg[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
}

// Row e of octet m8 lives in LDS row e*16 + m8, so that for each e the 16 lanes of an octet group write 16 consecutive
// LDS rows; the epilogue maps LDS row rho back to tile row 8*(rho & 15) + (rho >> 4).
__device__ __forceinline__ void store_mc(uint4 (*S)[BT], int t, const uint4 (&reg)[4]) {
    const int m8 = t & 15, kq = t >> 4;
    const int c = kq >> 1, half = kq & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint2* even = reinterpret_cast<uint2*>(&S[c][((2 * i) * 16 + m8) ^ (2 * c)]) + half;
        uint2* odd = reinterpret_cast<uint2*>(&S[c][((2 * i + 1) * 16 + m8) ^ (2 * c)]) + half;
        *even = make_uint2(reg[i].x, reg[i].y);
        *odd = make_uint2(reg[i].z, reg[i].w);
    }
}

template <bool A_KC>
__global__ __launch_bounds__(256) void lv_gemm_b16_kernel(GemmQ p) {
    __shared__ __attribute__((aligned(16))) uint4 As[2][NCH][BT];
    __shared__ __attribute__((aligned(16))) uint4 Bs[2][NCH][BT];

    // XCD-aware bijective remap + grouped (8 tile-rows) ordering, as in lv_gemm_f32 / lv_gemm_bf16
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

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 ra[4], rb[4];
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = (int)blockIdx.y * p.kt_per_split;
    int kt1 = kt0 + p.kt_per_split;
    if (kt1 > nk_all) kt1 = nk_all;
    const int nfull = p.K / BK;                        // K tiles [0, nfull) are complete

    const KcPtr qb = kc_setup(p.B, p.ldb, p.N, n0, t);
    KcPtr qa;
    McPtr qm;
    if (A_KC) qa = kc_setup(p.A, p.lda, p.M, m0, t);
    else qm = mc_setup(p.A, p.lda, m0, t);

    auto stage = [&](int kt) {                         // global -> registers for K tile kt
        const int k0 = kt * BK;
        if (kt < nfull) {
            if (A_KC) kc_load<true>(qa, k0, p.K, t, ra);
            else mc_load<true>(qm, k0, p.K, t, ra);
            kc_load<true>(qb, k0, p.K, t, rb);
        } else {
            if (A_KC) kc_load<false>(qa, k0, p.K, t, ra);
            else mc_load<false>(qm, k0, p.K, t, ra);
            kc_load<false>(qb, k0, p.K, t, rb);
        }
    };
    auto commit = [&](int buf) {                       // registers -> LDS image
        if (A_KC) store_kc(As[buf], t, ra);
        else store_mc(As[buf], t, ra);
        store_kc(Bs[buf], t, rb);
    };

    if (kt0 < kt1) stage(kt0);
    commit(0);
    __syncthreads();

    const int arow = wm * 64 + li, brow = wn * 64 + li;
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1 && !(LV_B16_ABL & 2)) stage(kt + 1);
        // fragments of k-step ks+1 are read while the MFMAs of k-step ks run (register double buffer)
        uint4 fa[2][2], fb[2][2];
        {
            const int c = lh;
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[0][i] = As[buf][c][(arow + i * 32) ^ (2 * c)];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[0][j] = Bs[buf][c][(brow + j * 32) ^ (2 * c)];
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < BK / 16 && !(LV_B16_ABL & 4)) {
                const int c = 2 * (ks + 1) + lh;
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[nxt][i] = As[buf][c][(arow + i * 32) ^ (2 * c)];
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[nxt][j] = Bs[buf][c][(brow + j * 32) ^ (2 * c)];
            } else if (ks + 1 < BK / 16) {
                fa[nxt][0] = fa[nxt][1] = ra[ks & 3]; fb[nxt][0] = fb[nxt][1] = rb[ks & 3];
            }
            if (LV_B16_ABL & 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j][0] += (float)((fa[cur][i].x ^ fb[cur][j].y) & 0xFFu);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = lv_mfma_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j]);
            }
        }
        if (kt + 1 < kt1) commit(buf ^ 1);
        __syncthreads();
    }

    const bool split = p.splits > 1;
    float* const out = split ? p.ws + (long)blockIdx.y * p.M * p.N : p.C;
    const long ldo = split ? p.N : p.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (l & 31);
            if (col >= p.N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);   // row within the wave's 64
                const int rho = wm * 64 + rr;                                    // LDS row of this accumulator row
                const int row = m0 + (A_KC ? rho : 8 * (rho & 15) + (rho >> 4));
                if (row >= p.M) continue;
                float* c = out + (long)row * ldo + col;
                if (split) { *c = acc[i][j][e]; continue; }
                float v = p.alpha * acc[i][j][e];
                if (p.add1) v += p.add1[(long)(row % p.mod1) * p.ld1 + col];
                if (p.add2) v += p.add2[(long)(row % p.mod2) * p.ld2 + col];
                if (p.accumulate) v += *c;
                *c = v;
            }
        }
}

#ifndef LV_B16_PP_DMA
#define LV_B16_PP_DMA 0        // ping-pong schedule: units of the next tile's DMA per LOAD segment of k-steps 0..3: 0 = 2,2,0,0; 1 = 2,1,1,0; 2 = 1,1,1,1
#endif
#ifndef LV_B16_PP_ABL
#define LV_B16_PP_ABL 0        // ablation switches for profiles/microbench only (results are wrong): 1 = no DMA in the loop, 2 = no MFMA, 4 = no fragment reads, 8 = no barriers, 16 = a second tile's DMA stays in flight (vmcnt(8)), 32 = only the A tile is staged, 64 = every workgroup stages tile (0, 0), 128 = no source-side swizzle
#endif
#ifndef LV_B16_PP_WAIT
#define LV_B16_PP_WAIT 1       // ping-pong schedule: group 0 waits for its DMA at the end of the MFMA segment of k-step 3 (0: both groups at the end of the LOAD segment)
#endif
#ifndef LV_B16_PP_PRIO
#define LV_B16_PP_PRIO 1       // ping-pong schedule: s_setprio 1 around the MFMA segment
#endif
#ifndef LV_B16_T256_ORDER
#define LV_B16_T256_ORDER 1   // 256 x 256 kernel: fragments of k-step ks+1 requested before (0) / in the middle of (1) the MFMAs of k-step ks
                              // (1 since the DMA moved into the first half of the tile: logits 310 -> 298 us, dO 304 -> 295, fused NLL 325 -> 317;
                              //  DMA ahead of the fragment reads, or one unit behind each MFMA group, measured the same)
#endif
#ifndef LV_B16_SINGLE_WAVES
#define LV_B16_SINGLE_WAVES 4   // single-buffer 128 x 128 kernel: waves per SIMD its register budget is cut to (0 = unconstrained: 161 registers, 3 workgroups per CU; 4: 123-128 registers, 4 per CU -- the 1600-tile input projection then needs two rounds instead of three: 47-53 -> 43 us)
#endif
#ifndef LV_B16_TN_SWZ
#define LV_B16_TN_SWZ 2    // TN image: 16-byte slot of k row k is permuted by s ^ SWZ*(k & 3) (A/B knob of the microbench)
#endif
#ifndef LV_B16_GLDS
#define LV_B16_GLDS 1       // NT form through LDS-DMA staging (0: register staging for both forms; A/B knob of the microbench)
#endif

// NT form (both operands K-contiguous) with LDS-DMA staging: global_load_lds_dwordx4 moves a K tile straight into LDS, so
// the 32 KB per K tile never pass through VGPRs and the ds_write pass (13 issue cycles per 1 KB, the busiest LDS port of
// the register-staged kernel) disappears.  The DMA fixes the LDS image: a wave instruction fills 1 KB in lane order, i.e.
// 8 rows x 128 B of the row-major tile [128 rows][8 slots of 16 B].  A fragment read (32 lanes = 32 consecutive rows of one
// k-chunk) on that image would be a 16-way bank conflict, so the swizzle goes on the SOURCE side: slot s of row r holds
// k-chunk c = s ^ ((r >> 1) & 7) -- the 8 lanes of a row still read that row's one 128-byte line (permuted), and the 16
// lanes of every ds_read_b128 lane group land on 16 distinct 16-byte bank slots.  The single ragged K tile at the end of
// a K that is not a multiple of 64 is staged through registers (predicated, zero-filled) into the same image.
// 16 bytes of a row whose last `valid` (1..7, or >= 8) elements are inside K: one aligned load, the tail masked to zero
// (rows are 16-byte aligned and ld % 8 == 0, so the load stays inside the row pitch)
__device__ __forceinline__ uint4 load_chunk_masked(const uint16_t* __restrict__ p, int valid) {
    uint4 q = *reinterpret_cast<const uint4*>(p);
    if (valid >= 8) return q;
    uint32_t wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        wv[i] &= (2 * i < valid ? 0xFFFFu : 0u) | (2 * i + 1 < valid ? 0xFFFF0000u : 0u);
    return make_uint4(wv[0], wv[1], wv[2], wv[3]);
}

// (max, sum exp(x - max)) over the first `valid` of a 64-column piece of logits; the piece inside N takes the branch without
// per-element bounds.  exp through the hardware exp2 (lv_exp_fast): the statistics epilogue is VALU-bound -- 64 precise expf per
// thread and piece cost as much as the tile's whole K loop at K = 1024 -- and 1-2 ulp on terms that are summed 20 001 at a time
// and enter the loss through a logarithm are far inside the binary16 rounding of the logits themselves.
__device__ __forceinline__ void nll_piece_stats(const float (&v)[64], int valid, float& mx, float& sm) {
    mx = -INFINITY;
    sm = 0.f;
    if (valid >= 64) {
#pragma unroll
        for (int k = 0; k < 64; ++k) mx = fmaxf(mx, v[k]);
#pragma unroll
        for (int k = 0; k < 64; ++k) sm += lv_exp_fast(v[k] - mx);
    } else {
#pragma unroll
        for (int k = 0; k < 64; ++k)
            if (k < valid) mx = fmaxf(mx, v[k]);
#pragma unroll
        for (int k = 0; k < 64; ++k)
            if (k < valid) sm += lv_exp_fast(v[k] - mx);
    }
}

// One 32 x 32 accumulator fragment -> C (alpha, the two row-cyclic addends, accumulate).  The addend loads are gathered before
// the stores (a bias -- mod 1 -- is ONE load per fragment; a per-batch-row addend is 16 independent loads): with a load and a
// modulo per element inside the store loop the bias cost the K = 512 input projection 55 -> 72 us and the decoder's z addend
// 55 -> 98 us.  One copy of this code per fragment and no second, unchecked variant: the epilogue is straight-line code run once
// per workgroup, and doubling it for interior tiles made the 256 x 256 kernel SLOWER (instruction fetch: Gx 48 -> 64 us).
#ifndef LV_B16_EPI_ABL
#define LV_B16_EPI_ABL 0        // measurement only (profiles/microbench/gemm_lstm_shapes.py; results are wrong): 1 = the f32 epilogue stores nothing
#endif
__device__ __forceinline__ void store_frag_f32(const GemmQ& p, const f32x16& a, int rbase, int col) {
    if (col >= p.N) return;
#if LV_B16_EPI_ABL & 1
    if (a[0] != 1.2345e-30f) return;
#endif
    if (!p.add1 && !p.add2 && !p.accumulate) {           // plain store: no per-element branches
        float* cb = p.C + (long)rbase * p.ldc + col;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ro = (e & 3) + 8 * (e >> 2);
            if (rbase + ro < p.M) cb[(long)ro * p.ldc] = p.alpha * a[e];
        }
        return;
    }
    float ad[16];
    if (p.add1) {
        if (p.mod1 == 1) {
            const float t = p.add1[col];
#pragma unroll
            for (int e = 0; e < 16; ++e) ad[e] = t;
        } else {
            const int q1 = rbase % p.mod1;
#pragma unroll
            for (int e = 0; e < 16; ++e) ad[e] = p.add1[(long)lv_wrap_row(q1, (e & 3) + 8 * (e >> 2), p.mod1) * p.ld1 + col];
        }
    }
    // The second addend (the decoder's z projection: every Gx of the decoder) is gathered like the first one -- 16 loads issued
    // together -- and INTO the same registers (ad = add1 + add2): loaded in place behind `if (rbase + ro < M)` each one was its own
    // memory round trip (Gx with the z addend: 61 us against 43 us plain), while 16 more live registers cost the 128 x 128 kernel
    // its occupancy (all of its GEMMs 40-55 % slower: measured).
    if (p.add2) {
        const int q2 = rbase % p.mod2;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float t2 = p.add2[(long)lv_wrap_row(q2, (e & 3) + 8 * (e >> 2), p.mod2) * p.ld2 + col];
            ad[e] = p.add1 ? ad[e] + t2 : t2;
        }
    }
    float* cb = p.C + (long)rbase * p.ldc + col;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int ro = (e & 3) + 8 * (e >> 2);
        if (rbase + ro >= p.M) continue;
        float* c = cb + (long)ro * p.ldc;
        float v = p.alpha * a[e];
        if (p.add1 || p.add2) v += ad[e];
        if (p.accumulate) v += *c;
        *c = v;
    }
}

typedef uint4 LdsTile[BT][NCH];
template <bool SINGLE> struct SecondPair { LdsTile a, b; };
template <> struct SecondPair<true> { int unused; };

// SINGLE = false: two buffer pairs (64 KB, 2 workgroups per CU), the DMA of tile k+1 runs under the MFMAs of tile k -- best
// for long K loops (dO: 384 vs 478 us).  SINGLE = true: one pair (32 KB, 3 workgroups per CU), load -> barrier -> MFMA ->
// barrier with the other workgroups hiding the transfer -- best when a workgroup only sees a few K tiles and prologue /
// epilogue dominate (Gx, K = 512: 55 vs 63 us; dX under split-K: 55 vs 60 us).  Measured: profiles/r02e_gemm_shapes.txt.
// TN = true (weight gradients: A stored [K][M], M-contiguous): the A tile arrives by the same DMA as [64 k][16 slots of 8 m] --
// a wave instruction fills 4 k rows of 256 B -- with the 16-byte slots of row k permuted by s ^ 2(k & 3) on the source side,
// and the K-contiguous MFMA fragment is taken out of that M-contiguous image by ds_read_b64_tr_b16 (two per fragment; the 16
// addresses of a lane group are 4 k rows x 32 contiguous bytes, on distinct banks thanks to the permutation).  The
// register-staged lv_gemm_b16_kernel<false> did this transposition with 16 integer ops per 8 x 4 block (dW_pred: 464 us).
template <bool SINGLE, bool NLL = false, bool TN = false, bool F16 = false>
__global__ __launch_bounds__(256, (SINGLE && !NLL && LV_B16_SINGLE_WAVES) ? LV_B16_SINGLE_WAVES : 1) void lv_gemm_b16_nt_glds_kernel(GemmQ p) {
    // F16: the operands are IEEE binary16 images (lv_cvt_h16_f32) and the products run on v_mfma_f32_32x32x16_f16 -- staging, LDS
    // images, swizzles and the epilogue are format-blind (16-bit elements)
    // separate LDS objects per buffer: the compiler orders an LDS read behind every in-flight LDS-DMA it cannot prove
    // disjoint (with one double-buffered array it put an s_waitcnt vmcnt(0) between the DMA issue and the first fragment read)
    __shared__ __attribute__((aligned(1024))) LdsTile As0, Bs0;
    __shared__ __attribute__((aligned(1024))) SecondPair<SINGLE> second;

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

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = (int)blockIdx.y * p.kt_per_split;
    int kt1 = kt0 + p.kt_per_split;
    if (kt1 > nk_all) kt1 = nk_all;
    const int nfull = p.K / BK;

    // staging units of this thread: wave w fills row blocks u = 4w + i (8 rows each); lane = (row l>>3, slot l&7)
    const uint16_t* ga[4];
    const uint16_t* gb[4];
    int kch[4];                                        // k offset (elements) of the chunk this lane fetches
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (4 * w + i) + (l >> 3);
        const int c = (l & 7) ^ ((row >> 1) & 7);
        kch[i] = 8 * c;
        int ra = m0 + row, rb = n0 + row;
        if (ra > p.M - 1) ra = p.M - 1;                // clamped, not predicated: such rows only reach C rows / columns
        if (rb > p.N - 1) rb = p.N - 1;                // that are never written
        ga[i] = p.A + (long)ra * p.lda + kch[i];
        gb[i] = p.B + (long)rb * p.ldb + kch[i];
        if constexpr (TN) {
            // A stored [K][M]: this lane fetches, for k row 4(4w + i) + (l >> 4) of a tile, the 8 m values of slot l & 15
            const int krow = 4 * (4 * w + i) + (l >> 4);
            long mc = m0 / 8 + ((l & 15) ^ (LV_B16_TN_SWZ * (krow & 3)));
            if (mc > p.lda / 8 - 1) mc = p.lda / 8 - 1;             // clamped inside the row pitch: such m only reach unwritten C rows
            ga[i] = p.A + (long)krow * p.lda + 8 * mc;
        }
    }
    auto stage_dma = [&](int kt, LdsTile& Ad, LdsTile& Bd) {        // a complete K tile: LDS-DMA
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (TN) lv_glds16(ga[i] + (long)k0 * p.lda, reinterpret_cast<char*>(&Ad[0][0]) + 1024 * (4 * w + i));
            else lv_glds16(ga[i] + k0, &Ad[8 * (4 * w + i)][0]);
            lv_glds16(gb[i] + k0, &Bd[8 * (4 * w + i)][0]);
        }
    };
    auto stage_ragged = [&](int kt, LdsTile& Ad, LdsTile& Bd) {     // the ragged last tile: masked loads through registers
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (4 * w + i) + (l >> 3);
            const int k = k0 + kch[i];
            const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (TN) {
                const int krow = 4 * (4 * w + i) + (l >> 4);
                reinterpret_cast<uint4*>(&Ad[0][0])[64 * (4 * w + i) + l] =
                    k0 + krow < p.K ? *reinterpret_cast<const uint4*>(ga[i] + (long)k0 * p.lda) : z4;
            } else {
                Ad[row][l & 7] = k < p.K ? load_chunk_masked(ga[i] + k0, p.K - k) : z4;
            }
            Bd[row][l & 7] = k < p.K ? load_chunk_masked(gb[i] + k0, p.K - k) : z4;
        }
    };

    const int arow = wm * 64 + li, brow = wn * 64 + li;
    const int ax0 = (arow >> 1) & 7, ax1 = ((arow + 32) >> 1) & 7;       // swizzle keys of this lane's fragment rows
    const int bx0 = (brow >> 1) & 7, bx1 = ((brow + 32) >> 1) & 7;
    // 16 MFMAs over the K tile in (Ac, Bc), then the hand-over: this wave's DMA into the other pair has landed
    // (vmcnt(0)), everybody's has and everybody is done reading this pair (barrier)
    // TN: byte offset, inside the [64 k][256 B] image, of the 8 bytes this lane supplies to the transpose read of A fragment i2
    // at k-step 0, first half: k row 8 lh + (r >> 2), m = wm*64 + 32 i2 + 16 (lane group & 1) + 4 (r & 3), r = l & 15
    int atr[2] = {0, 0};
    if constexpr (TN) {
        const int r = l & 15, kr = r >> 2;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int mloc = wm * 64 + 32 * i2 + 16 * ((l >> 4) & 1) + 4 * (r & 3);
            atr[i2] = (8 * lh + kr) * 256 + (((mloc >> 3) ^ (LV_B16_TN_SWZ * kr)) * 16) + (mloc & 7) * 2;
        }
    }
    auto a_frag = [&](LdsTile& Ac, int ks, int i2) -> uint4 {
        if constexpr (TN) {
            const char* base = reinterpret_cast<const char*>(&Ac[0][0]) + atr[i2] + ks * 16 * 256;
            const uint2 lo = lv_ds_read_tr16_b64(base), hi = lv_ds_read_tr16_b64(base + 4 * 256);
            return make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
            const int c = 2 * ks + lh;
            return i2 ? Ac[arow + 32][c ^ ax1] : Ac[arow][c ^ ax0];
        }
    };
    auto mma_tile = [&](LdsTile& Ac, LdsTile& Bc, bool hand_over = true) {
        uint4 fa[2][2], fb[2][2];
        {
            const int c = lh;
            fa[0][0] = a_frag(Ac, 0, 0); fa[0][1] = a_frag(Ac, 0, 1);
            fb[0][0] = Bc[brow][c ^ bx0]; fb[0][1] = Bc[brow + 32][c ^ bx1];
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < BK / 16) {
                const int c = 2 * (ks + 1) + lh;
                fa[nxt][0] = a_frag(Ac, ks + 1, 0); fa[nxt][1] = a_frag(Ac, ks + 1, 1);
                fb[nxt][0] = Bc[brow][c ^ bx0]; fb[nxt][1] = Bc[brow + 32][c ^ bx1];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = F16 ? lv_mfma_32x32x16_f16(fa[cur][i], fb[cur][j], acc[i][j]) : lv_mfma_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j]);
        }
        if (hand_over) LV_WAIT_VMEM();
        __syncthreads();
    };

    const int ntiles = kt1 - kt0;                                    // K tiles of this workgroup; tile i lives in pair i & 1
    const int nmain = (kt1 < nfull ? kt1 : nfull) - kt0;             // ... of which the first nmain are complete
    if constexpr (SINGLE) {
    for (int i = 0; i < ntiles; ++i) {
        if (i < nmain) stage_dma(kt0 + i, As0, Bs0);
        else stage_ragged(kt0 + i, As0, Bs0);
        LV_WAIT_VMEM();
        __syncthreads();
        mma_tile(As0, Bs0, false);
    }
    } else {
    LdsTile& As1 = second.a;
    LdsTile& Bs1 = second.b;
    if (ntiles > 0) {
        if (nmain > 0) stage_dma(kt0, As0, Bs0);
        else stage_ragged(kt0, As0, Bs0);
    }
    LV_WAIT_VMEM();
    __syncthreads();
    int i = 0;
    // hot loop: branch-free pairs of tiles, each followed by another complete tile (a conditional staging path inside this
    // loop made the compiler shuttle all 64 accumulators AGPR -> VGPR -> AGPR every iteration)
    for (; i + 2 < nmain; i += 2) {
        stage_dma(kt0 + i + 1, As1, Bs1);
        LV_SCHED_BARRIER();          // keep the DMA issue AHEAD of the tile's MFMAs (left alone, the scheduler sinks it to
        mma_tile(As0, Bs0);          // just before the vmcnt(0) and the whole transfer latency is exposed)
        stage_dma(kt0 + i + 2, As0, Bs0);
        LV_SCHED_BARRIER();
        mma_tile(As1, Bs1);
    }
    for (; i < ntiles; ++i) {                                        // the last <= 3 tiles
        const int nx = i + 1;
        if ((i & 1) == 0) {
            if (nx < nmain) stage_dma(kt0 + nx, As1, Bs1);
            else if (nx < ntiles) stage_ragged(kt0 + nx, As1, Bs1);
            mma_tile(As0, Bs0);
        } else {
            if (nx < nmain) stage_dma(kt0 + nx, As0, Bs0);
            else if (nx < ntiles) stage_ragged(kt0 + nx, As0, Bs0);
            mma_tile(As1, Bs1);
        }
    }

    }

    if constexpr (NLL) {
        // ---- fused epilogue of the vocabulary projection: the 128 x 128 tile goes through LDS as binary16 (the K loop's
        // 32 KB are free now: rows 0..63 in As0, 64..127 in Bs0), so that (i) the logits leave as 16-byte row segments
        // instead of 4-byte pieces -- half the bytes of the f32 image, which is never written -- and (ii) every thread owns
        // half a tile row (64 consecutive logits) and emits its (max, sum exp) for the online-softmax merge, plus the
        // target token's logit if it falls into its piece.  Statistics are taken from the ROUNDED values: forward and
        // backward then see the same logits and sum_c softmax = 1 holds exactly for the gradient.
        uint16_t* const tile0 = reinterpret_cast<uint16_t*>(&As0[0][0]);
        uint16_t* const tile1 = reinterpret_cast<uint16_t*>(&Bs0[0][0]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wn * 64 + j * 32 + (l & 31);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
                    (rr < 64 ? tile0 : tile1)[(rr & 63) * BT + cl] = lv_f32_to_f16_bits(p.alpha * acc[i][j][e]);
                }
            }
        __syncthreads();
        const int rr = t >> 1, half = t & 1;
        const int row = m0 + rr;
        if (row < p.M) {
            const uint16_t* src = (rr < 64 ? tile0 : tile1) + (rr & 63) * BT + 64 * half;
            const int c0 = n0 + 64 * half;
            uint4 q[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) q[k] = reinterpret_cast<const uint4*>(src)[k];
            uint16_t* dst = p.C16 + (long)row * p.ldc16 + c0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c0 + 8 * k + 8 <= p.ldc16) reinterpret_cast<uint4*>(dst)[k] = q[k];
            float v[64];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t wv[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
                for (int h2 = 0; h2 < 4; ++h2) {
                    v[8 * k + 2 * h2] = lv_f16_bits_to_f32((uint16_t)(wv[h2] & 0xFFFFu));
                    v[8 * k + 2 * h2 + 1] = lv_f16_bits_to_f32((uint16_t)(wv[h2] >> 16));
                }
            }
            float mx, sm;
            nll_piece_stats(v, p.N - c0, mx, sm);
            p.part[(long)row * p.nparts + 2 * tn + half] = make_float2(mx, sm);
            if (tn == p.tilesN - 1)                     // pieces counted in 256-column tiles: the (empty) ones beyond this tile
                for (int k = 2 * p.tilesN + half; k < p.nparts; k += 2) p.part[(long)row * p.nparts + k] = make_float2(-INFINITY, 0.f);
            const int tt = row / p.Bsz, bb = row % p.Bsz;
            long tg = p.ids[(long)bb * p.ids_stride + tt + p.tgt_off];
            if (tg < 0) tg = 0;
            if (tg >= p.N) tg = p.N - 1;
            const int tl = (int)tg - c0;
            if (tl >= 0 && tl < 64) p.tgt[row] = lv_f16_bits_to_f32(src[tl]);
        }
        return;
    }
    if (p.splits > 1) {
        float* const out = p.ws + (long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + (l & 31);
                if (col >= p.N) continue;
                const int rbase = m0 + wm * 64 + i * 32 + 4 * (l >> 5);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = rbase + (e & 3) + 8 * (e >> 2);
                    if (row < p.M) out[(long)row * p.N + col] = acc[i][j][e];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            store_frag_f32(p, acc[i][j], m0 + wm * 64 + i * 32 + 4 * (l >> 5), n0 + wn * 64 + j * 32 + (l & 31));
}

// ---- 256 x 256 x 64 tile, 8 waves (2 along M x 4 along N, wave tile 128 x 64 = 4 x 2 v_mfma_f32_32x32x16_bf16), one workgroup per
// CU: the same LDS-DMA staging, images and swizzles as the 128 x 128 kernel above at twice the tile edge, i.e. half the LDS
// fragment bytes and half the L2 -> LDS bytes per MFMA and 32 MFMAs per wave between barriers instead of 16.  LDS: two buffer
// pairs of 2 x 32 KB = 128 KB of the CU's 160.  For the large products only (the three vocabulary-sized GEMMs of the decoder):
// a tile is 8.4 MFLOP per K step, so the grid is cut to the 256 CUs explicitly -- whole rounds of 256 tiles run their full K
// range and write C directly; the TAIL (tiles % 256) is split along K into `tail_s` pieces per tile so that it fills a round too
// (dW_pred: 316 tiles = 256 + 60 x 4 pieces; dO: 100 tiles x 5 pieces = two rounds), the pieces go to the workspace as dense
// 256 x 256 slabs and tail_reduce_t256_kernel adds them in piece order (deterministic) and applies the epilogue.
constexpr int BT2 = 256;
typedef uint4 LdsTile2[BT2][NCH];

struct Tail256 {
    int full;          // tiles [0, full) run the whole K range (full % 256 == 0 or the launch has no tail)
    int tail;          // tiles [full, full + tail): the last, partial round
    int tail_s;        // pieces per tail tile (1: the tail tiles run whole as well)
    int kt_per_piece;  // K tiles per piece
};

#ifndef LV_T256_G
#define LV_T256_G 4                      // M tiles per group of the 256-tile kernel's tile order: the 32 workgroups an XCD runs at a time
                                         // then cover 4 x 8 tiles, whose operand slices fit its 4 MB L2 better than 8 x 4 (8: plain
                                         // logits 312 us, dO 300, 8192^3 930; 4: 284 / 290 / 890-915; 2, 3, 6, 16 in between or worse)
#endif
__device__ __forceinline__ void t256_tile_of(const GemmQ& p, int s, int& tm, int& tn) {
    const int G = LV_T256_G;
    const int nig = G * p.tilesN;
    const int group = s / nig;
    const int first_m = group * G;
    const int gsz = (p.tilesM - first_m) < G ? (p.tilesM - first_m) : G;
    tm = first_m + (s % nig) % gsz;
    tn = (s % nig) / gsz;
}

// Which tile (and which K piece of it) workgroup `bid` of a 256 x 256 launch computes.
struct T256Pick { int tile, kt0, kt1, piece, tm, tn; };
__device__ __forceinline__ T256Pick t256_pick(const GemmQ& p, const Tail256& q, int bid) {
    T256Pick k;
    const int nk_all = (p.K + BK - 1) / BK;
    k.piece = -1;
    if (bid < q.full) {
        // whole rounds: consecutive workgroup ids land on different XCDs; give each XCD (own L2) a contiguous range of tiles
        const int xcd = bid % 8, per = q.full / 8;
        k.tile = xcd * per + bid / 8;
        k.kt0 = 0; k.kt1 = nk_all;
    } else {
        // tail workgroups: XCD x (= workgroup id % 8, own L2) gets a contiguous range of (piece, tile) pairs in piece-major order,
        // i.e. neighbouring tiles over the SAME K range, so that its 32 co-resident workgroups share A and B tiles in L2 (with the
        // pieces of a tile on consecutive workgroup ids -- eight different XCDs -- dO fetched 1030 MB per launch for 296 MB of operands)
        const int r = bid - q.full, nw = q.tail * q.tail_s;
        const int xcd = r % 8, qq = nw / 8, rem = nw % 8;
        const int idx = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + r / 8;
        k.tile = q.full + idx % q.tail;
        k.piece = idx / q.tail;
        k.kt0 = k.piece * q.kt_per_piece;
        k.kt1 = k.kt0 + q.kt_per_piece;
        if (k.kt1 > nk_all) k.kt1 = nk_all;
        if (q.tail_s == 1) k.piece = -1;
    }
    t256_tile_of(p, k.tile, k.tm, k.tn);
    return k;
}

// What a workgroup of the 256 x 256 kernels does with its finished accumulators (wave (wm, wn) of 2 x 4 holds the 128 x 64 block
// at (128 wm, 64 wn) as 4 x 2 fragments): the fused NLL epilogue, a K piece's slab, or C.  Every wave of the workgroup has passed
// a barrier behind its last read of the K tiles, and no LDS-DMA is in flight.
template <bool NLL>
__device__ __forceinline__ void t256_epilogue(const GemmQ& p, const Tail256& q, f32x16 (&acc)[4][2], LdsTile2& As0, LdsTile2& Bs0,
                                              LdsTile2& As1, LdsTile2& Bs1, int tile, int piece, int tn, int m0, int n0, int t, int l,
                                              int wm, int wn, int lh) {
    constexpr int NJ = 2;
    if constexpr (NLL) {
#if LV_NLL_ABL & 4
        return;
#endif
        // fused epilogue of the vocabulary projection (see the 128 x 128 kernel): the 256 x 256 tile goes through the 128 KB of
        // LDS as binary16 (64 rows of 512 B per buffer, 16-byte chunks permuted by chunk ^ (row & 15) so that the row-per-lane
        // reads below are conflict-free), thread (row rr = t & 255, half = t >> 8) owns 128 consecutive logits = two 64-column
        // pieces of the statistics
        auto rowptr = [&](int rr) -> char* {            // 64 tile rows of 512 B per buffer
            const int b = rr >> 6;
            char* base = b == 0 ? reinterpret_cast<char*>(&As0[0][0]) : b == 1 ? reinterpret_cast<char*>(&Bs0[0][0])
                       : b == 2 ? reinterpret_cast<char*>(&As1[0][0]) : reinterpret_cast<char*>(&Bs1[0][0]);
            return base + (rr & 63) * 512;
        };
        // Write phase: a lane holds ONE column of 16 rows per fragment, its neighbour (lane ^ 1) the next column.  Rows are taken
        // in pairs (e, e + 1): the even lane sends its row e + 1 and keeps row e, the odd lane the other way round (one DPP move),
        // so every lane stores one packed pair (two adjacent columns of one row) -- 64 ds_write_b32 per thread instead of 128
        // ds_write_b16, one conversion instruction per pair.  Address = buffer + row * 512 + (chunk ^ (row & 15)) * 16 + ...; with
        // row & 15 = (e & 3 | odd) | 4 lh | 8 (e >> 2 & 1) the lane part and the per-e part of the XOR separate.
        const int odd = l & 1;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            char* const bufp = rowptr(wm * 128 + i2 * 32);                  // row (i2 & 1) * 32 of its buffer
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int clp = (wn * 64 + j * 32 + (l & 31)) & ~1;          // first column of this lane's pair
                const int lane_off = (4 * lh + odd) * 512 + ((((clp >> 3) ^ (4 * lh) ^ odd) << 4) | ((clp & 7) << 1));
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const float mine0 = acc[i2][j][e], mine1 = acc[i2][j][e + 1];
                    const float got = lv_lane_xor1(odd ? mine0 : mine1);
                    const uint32_t pk = odd ? lv_pack_f16x2(got, mine1) : lv_pack_f16x2(mine0, got);
                    const int key = (e & 3) | (8 * ((e >> 2) & 1));          // compile-time part of row & 15
                    const int roff = ((e & 3) + 8 * (e >> 2)) * 512;
                    *reinterpret_cast<uint32_t*>(bufp + roff + (lane_off ^ (key << 4))) = pk;
                }
            }
        }
        __syncthreads();
#if !(LV_NLL_ABL & 1)
        // the binary16 logits leave in ROW order: half a wave per 512-byte tile row (32 lanes x 16 B = four full lines), 16 rows per
        // step of the workgroup.  (From the row-per-lane registers of the statistics below every store instruction touched 64
        // different lines, 16 bytes each -- 8192 line requests per tile instead of 1024 -- and the stores were 57 of the kernel's 302 us.)
        {
            const int c = l & 31;
#pragma unroll 4
            for (int s16 = 0; s16 < 16; ++s16) {
                const int r2 = 16 * s16 + 2 * (t >> 6) + (l >> 5);
                const uint4 qv = *reinterpret_cast<const uint4*>(rowptr(r2) + ((c ^ (r2 & 15)) << 4));
                const int col0 = n0 + 8 * c;
                // (non-temporal stores measured the same: 274 vs 278 us for the launch, nothing in the step)
                if (m0 + r2 < p.M && col0 + 8 <= p.ldc16) *reinterpret_cast<uint4*>(p.C16 + (long)(m0 + r2) * p.ldc16 + col0) = qv;
            }
        }
#endif
        const int rr = t & 255, half = t >> 8;
        const int row = m0 + rr;
        if (row < p.M) {
            const char* rowp = rowptr(rr);
            const int tt = row / p.Bsz, bb = row % p.Bsz;
            long tg = p.ids[(long)bb * p.ids_stride + tt + p.tgt_off];
            if (tg < 0) tg = 0;
            if (tg >= p.N) tg = p.N - 1;
#pragma unroll 1
            for (int pc = 0; pc < 2; ++pc) {
                const int cb = 128 * half + 64 * pc;       // first column of this piece inside the tile
                const int c0 = n0 + cb;
                uint4 qv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) qv[k] = *reinterpret_cast<const uint4*>(rowp + ((((cb >> 3) + k) ^ (rr & 15)) << 4));
#if LV_NLL_ABL & 2
                if (qv[0].x == 0x12345678u) p.tgt[row] = 1.f;
                continue;
#endif
                float v[64];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t wv[4] = {qv[k].x, qv[k].y, qv[k].z, qv[k].w};
#pragma unroll
                    for (int h2 = 0; h2 < 4; ++h2) {
                        v[8 * k + 2 * h2] = lv_f16_bits_to_f32((uint16_t)(wv[h2] & 0xFFFFu));
                        v[8 * k + 2 * h2 + 1] = lv_f16_bits_to_f32((uint16_t)(wv[h2] >> 16));
                    }
                }
                float mx, sm;
                nll_piece_stats(v, p.N - c0, mx, sm);
                p.part[(long)row * p.nparts + 4 * tn + 2 * half + pc] = make_float2(mx, sm);
                const int tl = (int)tg - c0;
                if (tl >= 0 && tl < 64) {
                    const int cl = cb + tl;
                    p.tgt[row] = lv_f16_bits_to_f32(*reinterpret_cast<const uint16_t*>(rowp + ((((cl >> 3) ^ (rr & 15)) << 4) | ((cl & 7) << 1))));
                }
            }
        }
        return;
    }
    if (piece >= 0) {
        // a K piece of a tail tile: dense 256 x 256 slab (no bounds: the reduce reads only what is inside C)
        if (p.sq && piece == 0 && l == 0) p.sq[tile * 8 + (t >> 6)] = 0.f;          // this tile's squares come from the reduce
        float* slab = p.ws + ((long)(tile - q.full) * q.tail_s + piece) * (BT2 * BT2);
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = wn * (32 * NJ) + j * 32 + (l & 31);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = wm * 128 + i2 * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
                    slab[rr * BT2 + col] = acc[i2][j][e];
                }
            }
        return;
    }
    if (p.sq) {
        // the gradient norm's share of this tile (what lies inside C, as stored: alpha * acc), one partial per wave
        float ss = 0.f;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int rbase = m0 + wm * 128 + i2 * 32 + 4 * (l >> 5), col = n0 + wn * (32 * NJ) + j * 32 + (l & 31);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = p.alpha * acc[i2][j][e];
                    if (col < p.N && rbase + (e & 3) + 8 * (e >> 2) < p.M) ss += v * v;
                }
            }
        ss = lv_wave_sum(ss);
        if (l == 0) p.sq[tile * 8 + (t >> 6)] = ss;
        if (p.sq_only) return;
    }
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            store_frag_f32(p, acc[i2][j], m0 + wm * 128 + i2 * 32 + 4 * (l >> 5), n0 + wn * (32 * NJ) + j * 32 + (l & 31));
}

// PP ("ping-pong", round 4): the K loop runs as two wave groups half a k-step apart.  Waves w and w + 4 share SIMD w & 3 and belong
// to different groups (wm = 0 / 1); a k-step of a wave is a LOAD segment (its 6 fragment reads of that k-step + its share of the
// next tile's LDS-DMA) and an MFMA segment (8 MFMAs = 256 cycles of the SIMD's matrix pipe), each closed by a bare s_barrier, and
// group 1 enters the loop one barrier late: whenever one wave of a SIMD multiplies, its partner reads / stages, so the matrix pipe
// of every SIMD sees ONE instruction stream of back-to-back MFMAs while only four waves at a time compete for the LDS port.
// Ordering of the LDS-DMA: a tile's 8 DMA instructions go out in the LOAD segments of k-steps 0 and 1 of the tile before it, every
// wave waits for its own (vmcnt(0): nothing newer is in flight then) before the barrier behind which group 0 starts to read the new
// tile (group 1 at the end of its LOAD segment of k-step 3, group 0 one segment later), and a LOAD segment ends with lgkmcnt(0)
// BEFORE its barrier, so the buffer a DMA overwrites (the tile before the current one) has been read completely by both groups
// when the first DMA instruction of the tile after the current one is issued.  Measured (profiles/r04a_gemm_pingpong_probe.txt):
// 8192^3 1204 -> 1241 TF, dO 290 -> 280 us, logits 282 -> 272 us, dW_pred unchanged -- the ablations of
// profiles/r04b_gemm_pingpong_ablation.txt say why it is not more: without the DMA the same loop runs at 2100 TF, the DMA alone
// (no MFMA, no fragment reads) takes 87 % of the full kernel's time: a CU gets ~45 GB/s out of L2 into LDS, i.e. 64 KB per 1.4 us
// against 0.85 us of matrix-pipe time per K tile.
template <bool NLL, bool TN, bool PP = false>
__global__ __launch_bounds__(512) void lv_gemm_b16_t256_kernel(GemmQ p, Tail256 q) {
    constexpr int WN = 4;                               // 8 waves as 2 (M) x 4 (N), wave tile 128 x 64
    constexpr int NJ = 2;                               // 32-column fragments per wave along N
    constexpr int UN = 4;                               // 1 KB staging units per wave and image
    __shared__ __attribute__((aligned(1024))) LdsTile2 As0, Bs0;
    __shared__ __attribute__((aligned(1024))) LdsTile2 As1, Bs1;

    const int t = (int)threadIdx.x;
    const int l = t & 63, w = lv_wave_uniform(t >> 6);   // wave id in SGPRs: the LDS-DMA destinations below are scalar (M0)
    const int wm = w / WN, wn = w % WN;
    const int li = l & 31, lh = l >> 5;
    const int nfull = p.K / BK;

    const T256Pick pk = t256_pick(p, q, (int)blockIdx.x);
    const int kt0 = pk.kt0, kt1 = pk.kt1;
    const int m0 = pk.tm * BT2, n0 = pk.tn * BT2;

    f32x16 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // staging units of this thread: wave w fills the 1 KB pieces u = 4w + i of each image (8 rows of 128 B; TN A image: 2 k rows
    // of 512 B)
    uint32_t oa[UN], ob[UN];                           // element offsets from p.A / p.B (the launcher checks they fit 32 bits)
    int kch[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) {
        const int u = UN * w + i;
        const int row = 8 * u + (l >> 3);
#if LV_B16_PP_ABL & 128
        const int c = (l & 7);
#else
        const int c = (l & 7) ^ ((row >> 1) & 7);
#endif
        kch[i] = 8 * c;
#if LV_B16_PP_ABL & 64
        int ra = row, rb = row;
#else
        int ra = m0 + row, rb = n0 + row;
#endif
        if (ra > p.M - 1) ra = p.M - 1;                // clamped, not predicated (see the 128 x 128 kernel)
        if (rb > p.N - 1) rb = p.N - 1;
        oa[i] = (uint32_t)((long)ra * p.lda + kch[i]);
        ob[i] = (uint32_t)((long)rb * p.ldb + kch[i]);
        if constexpr (TN) {
            // A stored [K][M]: k row 2u + (l >> 5) of the tile, 16-byte slot (l & 31) ^ 4 (k & 3) of its 32
            const int krow = 2 * u + (l >> 5);
            long mc = m0 / 8 + ((l & 31) ^ (4 * (krow & 3)));
            if (mc > p.lda / 8 - 1) mc = p.lda / 8 - 1;
            oa[i] = (uint32_t)((long)krow * p.lda + 8 * mc);
        }
    }
    auto stage_unit = [&](int kt, int i, LdsTile2& Ad, LdsTile2& Bd) {     // 2 of the 8 DMA instructions of a K tile
        const uint32_t k0 = (uint32_t)(kt * BK);
        const uint32_t ka = TN ? k0 * (uint32_t)p.lda : k0;
        lv_glds16(p.A + (size_t)(oa[i] + ka), reinterpret_cast<char*>(&Ad[0][0]) + 1024 * (UN * w + i));
#if !(LV_B16_PP_ABL & 32)
        lv_glds16(p.B + (size_t)(ob[i] + k0), reinterpret_cast<char*>(&Bd[0][0]) + 1024 * (UN * w + i));
#endif
    };
    auto stage_dma = [&](int kt, LdsTile2& Ad, LdsTile2& Bd) {
#pragma unroll
        for (int i = 0; i < UN; ++i) stage_unit(kt, i, Ad, Bd);
    };
    auto stage_ragged = [&](int kt, LdsTile2& Ad, LdsTile2& Bd) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            const int u = UN * w + i;
            const int k = k0 + kch[i];
            const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (TN) {
                const int krow = 2 * u + (l >> 5);
                reinterpret_cast<uint4*>(&Ad[0][0])[64 * u + l] =
                    k0 + krow < p.K ? *reinterpret_cast<const uint4*>(p.A + (size_t)(oa[i] + (uint32_t)k0 * (uint32_t)p.lda)) : z4;
            } else {
                reinterpret_cast<uint4*>(&Ad[0][0])[64 * u + l] = k < p.K ? load_chunk_masked(p.A + (size_t)(oa[i] + (uint32_t)k0), p.K - k) : z4;
            }
            reinterpret_cast<uint4*>(&Bd[0][0])[64 * u + l] = k < p.K ? load_chunk_masked(p.B + (size_t)(ob[i] + (uint32_t)k0), p.K - k) : z4;
        }
    };

    const int arow = wm * 128 + li, brow = wn * (32 * NJ) + li;
    const int sx = (li >> 1) & 7;                      // swizzle key of this lane's fragment rows (rows differ by multiples of 32)
    int atr[4] = {0, 0, 0, 0};
    if constexpr (TN) {
        const int r = l & 15, kr = r >> 2;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            const int mloc = wm * 128 + 32 * i2 + 16 * ((l >> 4) & 1) + 4 * (r & 3);
            atr[i2] = (8 * lh + kr) * 512 + (((mloc >> 3) ^ (4 * kr)) * 16) + (mloc & 7) * 2;
        }
    }
    auto a_frag = [&](LdsTile2& Ac, int ks, int i2) -> uint4 {
        if constexpr (TN) {
            const char* base = reinterpret_cast<const char*>(&Ac[0][0]) + atr[i2] + ks * 16 * 512;
            const uint2 lo = lv_ds_read_tr16_b64(base), hi = lv_ds_read_tr16_b64(base + 4 * 512);
            return make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
            return Ac[arow + 32 * i2][(2 * ks + lh) ^ sx];
        }
    };
    // One K tile (Ac, Bc) while the next one (kt_next) streams into (Ad, Bd).  Pinned with scheduling barriers: the fragments of
    // k-step ks + 1 are requested BEFORE the 8 MFMAs of k-step ks (left alone the compiler reads each fragment group right before
    // its MFMAs and waits out the LDS latency eight times per tile), and the tile's 8 DMA instructions go out two per k-step BETWEEN
    // the MFMAs (an LDS-DMA instruction costs 60-180 issue cycles; at the top of the tile they would all run with the pipe empty).
    LV_TRACE_ONLY(int tr_kt = 0;)
    auto mma_tile = [&](auto staging, LdsTile2& Ac, LdsTile2& Bc, int kt_next, LdsTile2& Ad, LdsTile2& Bd) {
        constexpr bool STAGE = decltype(staging)::value;
        LV_TRACE_MARK(tr_kt, 0);
        uint4 fa[2][4], fb[2][NJ];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) fa[0][i2] = a_frag(Ac, 0, i2);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[0][j] = Bc[brow + 32 * j][lh ^ sx];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
#if LV_B16_T256_ORDER == 0
            if (ks + 1 < BK / 16) {
                const int c = 2 * (ks + 1) + lh;
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[nxt][j] = Bc[brow + 32 * j][c ^ sx];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) fa[nxt][i2] = a_frag(Ac, ks + 1, i2);
            }
            LV_SCHED_BARRIER();
#endif
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = lv_mfma_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j]);
            LV_SCHED_BARRIER();
#if LV_B16_T256_ORDER == 1
            if (ks + 1 < BK / 16) {
                const int c = 2 * (ks + 1) + lh;
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[nxt][j] = Bc[brow + 32 * j][c ^ sx];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) fa[nxt][i2] = a_frag(Ac, ks + 1, i2);
            }
#endif
            // the next tile's 8 DMA instructions: two UNITS (4 instructions) behind each of the first two k-steps' first MFMA groups, so
            // that the last of them has half a tile of MFMAs to land under (one unit per k-step left the last pair ~600 cycles short
            // at the end-of-tile vmcnt(0): 8192^3 1090 -> 1185 TF, dW_pred 290 -> 276 us, dO 307 -> 300; all four units at the top, or
            // 1-2-1, measured worse: profiles/microbench/gemm_b16_shapes.py with -DLV_B16_T256_DMA=0/2/3/4)
#if LV_B16_T256_DMA == 1
            if constexpr (STAGE) {
                if (ks < 2) {
#pragma unroll
                    for (int i = 0; i < UN / 2; ++i) stage_unit(kt_next, (UN / 2) * ks + i, Ad, Bd);
                }
            }
#elif LV_B16_T256_DMA == 2
            if constexpr (STAGE) { if (ks == 0) { stage_unit(kt_next, 0, Ad, Bd); stage_unit(kt_next, 1, Ad, Bd); } else if (ks < 3) stage_unit(kt_next, ks + 1, Ad, Bd); }
#elif LV_B16_T256_DMA == 3
            if constexpr (STAGE) { if (ks == 0) { stage_unit(kt_next, 0, Ad, Bd); stage_unit(kt_next, 1, Ad, Bd); stage_unit(kt_next, 2, Ad, Bd); } else if (ks == 1) stage_unit(kt_next, 3, Ad, Bd); }
#elif LV_B16_T256_DMA == 4
            if constexpr (STAGE) { if (ks == 0) stage_unit(kt_next, 0, Ad, Bd); else if (ks == 1) { stage_unit(kt_next, 1, Ad, Bd); stage_unit(kt_next, 2, Ad, Bd); } else if (ks == 2) stage_unit(kt_next, 3, Ad, Bd); }
#else
            if constexpr (STAGE) stage_unit(kt_next, ks, Ad, Bd);
#endif
            LV_SCHED_BARRIER();
#pragma unroll
            for (int i = 2; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = lv_mfma_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j]);
            LV_SCHED_BARRIER();
        }
        LV_TRACE_MARK(tr_kt, 1);
        if constexpr (STAGE) LV_WAIT_VMEM();
        LV_TRACE_MARK(tr_kt, 2);
        __syncthreads();
        LV_TRACE_MARK(tr_kt, 3);
        LV_TRACE_ONLY(++tr_kt;)
    };

    // The ragged tile at the end of a K that is not a multiple of 64 goes FIRST (masked loads through registers while the
    // accumulators are still zero and not live: behind the main loop its staging code spilled them), then the complete tiles.
    const int nmain = (kt1 < nfull ? kt1 : nfull) - kt0;
    if (kt1 > kt0 + nmain) {
        stage_ragged(kt1 - 1, As0, Bs0);
        __syncthreads();
        mma_tile(std::false_type{}, As0, Bs0, -1, As1, Bs1);
    }
    if (nmain > 0) stage_dma(kt0, As0, Bs0);
    LV_WAIT_VMEM();
    __syncthreads();
    // branch-free staging: the tile after the last one is the last one again (a harmless reload into the idle buffer), so the
    // loop body is the only copy of the K-tile code besides the ragged prologue
    const int klast = kt0 + nmain - 1;
    if constexpr (PP) {
        auto pp_tile = [&](LdsTile2& Ac, LdsTile2& Bc, int kt_next, LdsTile2& Ad, LdsTile2& Bd) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                // LOAD segment
                uint4 fa[4], fb[NJ];
#if LV_B16_PP_ABL & 4
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = make_uint4(oa[j], ob[j], oa[j] + ks, 0x3f803f80u);
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) fa[i2] = make_uint4(ob[i2], oa[i2], 0x3f803f80u, ob[i2] + ks);
#else
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = Bc[brow + 32 * j][(2 * ks + lh) ^ sx];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) fa[i2] = a_frag(Ac, ks, i2);
#endif
                LV_SCHED_BARRIER();
#if LV_B16_PP_ABL & 1
#elif LV_B16_PP_DMA == 0
                if (ks < 2) {
#pragma unroll
                    for (int i = 0; i < UN / 2; ++i) stage_unit(kt_next, (UN / 2) * ks + i, Ad, Bd);
                }
#elif LV_B16_PP_DMA == 1
                if (ks == 0) { stage_unit(kt_next, 0, Ad, Bd); stage_unit(kt_next, 1, Ad, Bd); }
                else if (ks < 3) stage_unit(kt_next, ks + 1, Ad, Bd);
#else
                stage_unit(kt_next, ks, Ad, Bd);
#endif
                LV_SCHED_BARRIER();
#if LV_B16_PP_ABL & 16
                if (ks == BK / 16 - 1) LV_WAIT_VMEM_N(8);
#else
                // the DMA wait: every wave before the SAME barrier -- the one behind which group 0 starts reading the next tile --, i.e.
                // group 1 at the end of this LOAD segment, group 0 one segment later, at the end of its MFMA segment (LV_B16_PP_WAIT 1)
                if (ks == BK / 16 - 1 && (!LV_B16_PP_WAIT || wm == 1)) LV_WAIT_VMEM();
#endif
                LV_WAIT_LDS();
                if (!(LV_B16_PP_ABL & 8)) LV_S_BARRIER();
                LV_SCHED_BARRIER();
                // MFMA segment
                if (LV_B16_PP_PRIO) LV_SETPRIO(1);
#if LV_B16_PP_ABL & 2
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j][0] += (float)((fa[i].x ^ fb[j].y) & 0xFFu);
#else
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = lv_mfma_32x32x16_bf16(fa[i], fb[j], acc[i][j]);
#endif
                if (LV_B16_PP_PRIO) LV_SETPRIO(0);
                LV_SCHED_BARRIER();
                if (LV_B16_PP_WAIT && ks == BK / 16 - 1 && wm == 0) LV_WAIT_VMEM();
                if (!(LV_B16_PP_ABL & 8)) LV_S_BARRIER();
                LV_SCHED_BARRIER();
            }
        };
        if (wm == 1) LV_S_BARRIER();                 // group 1 runs one segment behind group 0
        for (int i = 0; i < nmain; i += 2) {
            pp_tile(As0, Bs0, kt0 + i + 1 < klast ? kt0 + i + 1 : klast, As1, Bs1);
            if (i + 1 >= nmain) break;
            pp_tile(As1, Bs1, kt0 + i + 2 < klast ? kt0 + i + 2 : klast, As0, Bs0);
        }
        if (wm == 0) LV_S_BARRIER();                 // ... and is met again here: behind this barrier nobody reads the K tiles any more
    } else {
    for (int i = 0; i < nmain; i += 2) {
        mma_tile(std::true_type{}, As0, Bs0, kt0 + i + 1 < klast ? kt0 + i + 1 : klast, As1, Bs1);
        if (i + 1 >= nmain) break;
        mma_tile(std::true_type{}, As1, Bs1, kt0 + i + 2 < klast ? kt0 + i + 2 : klast, As0, Bs0);
    }
    }

    t256_epilogue<NLL>(p, q, acc, As0, Bs0, As1, Bs1, pk.tile, pk.piece, pk.tn, m0, n0, t, l, wm, wn, lh);
}

// ---- the same tile with a QUADRANT-ordered ping-pong schedule and a continuous LDS-DMA stream (round 4) -----------------------
// The k-step schedules above stage a K tile as two bursts of 32 KB and then leave the memory pipeline empty until the next tile may
// be fetched (its buffer is read until the last k-step): a CU holds ~32 KB of DMA in flight and a burst takes one L2 round trip, so
// the tile period is two round trips + the gap.  Here a wave's 128 x 64 block is walked as four QUADRANTS (64 x 32: 2 fragments x 1
// fragment x all four k-steps = 8 MFMAs), in the order (a, b) = (0,0) (0,1) (1,1) (1,0): quadrant rows a / columns b are exactly one
// HALF of the A / B tile (A-half a = rows 128 wm + 64 a + [0, 64) for both wm, B-half b = rows 64 wn + 32 b + [0, 32) for all four
// wn), so a half-tile is read in ONE phase -- A0 and B0 in phase 0, B1 in phase 1, A1 in phase 2, nothing in phase 3 -- and its LDS
// can be refilled right behind that phase, 1.25 to 1.75 tiles ahead.  Every LOAD segment therefore issues one half-tile (2 DMA
// instructions per wave, 16 KB per workgroup): in the phases 1, 2, 3, 0 (of the next tile) the halves A0, B0, B1, A1 of the tile
// after next, into the buffer the current tile is being read from.  The stream never pauses: the DMA statements are inline assembly
// (lv_glds16_uncounted: hipcc would otherwise drain the stream in front of every read of a buffer that has DMA in flight), the
// waits are counted by hand (vmcnt(10): the five half-tiles issued behind the one about to be read stay in flight) and every wave places its wait before the barrier behind which
// group 0 starts reading that half (group 1: end of its LOAD segment, group 0: end of its MFMA segment).  Fragment registers: A
// half 8 x 4, both B halves 2 x 4 x 4 = 64 (the k-step schedules: 24 / 48).  The LDS images of B and of a K-contiguous A are the
// ones above (a half-tile is a set of whole 8-row DMA units); the M-contiguous A of the weight-gradient form is kept per half as
// [64 k][16 slots of 16 B] (slot = (8 wm + m / 8) ^ 4 (k & 3): the swizzle stays inside the half).
// The K loop of the quadrant schedule as a function of (tile origin, K-tile range): the kernel below runs it once per workgroup, the
// grouped stream-K kernel (lv_gemm_b16_t256g_kernel) once per segment of a workgroup's unit range.  On return every wave has passed a
// barrier behind its last read of the K tiles and no LDS-DMA is in flight: the four LDS buffers are free.
template <bool TN>
__device__ __forceinline__ void t256q_run(const GemmQ& p, int m0, int n0, int kt0, int kt1, f32x16 (&acc)[4][2], LdsTile2& As0, LdsTile2& Bs0,
                                          LdsTile2& As1, LdsTile2& Bs1, int l, int w) {
    const int wm = w >> 2, wn = w & 3;
    const int li = l & 31, lh = l >> 5;

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nfull = p.K / BK;

    // This wave's DMA items: for each half-tile two 1 KB units.  item h in {A0, A1, B0, B1} x u in {0, 1}: source element offset
    // (per lane), LDS byte offset inside the operand's tile buffer (wave-uniform), and what the ragged tile needs (the lane's k).
    uint32_t osrc[4][2];
    int odst[4][2], okk[4][2];                  // okk: NT = k offset of the lane's chunk inside the tile, TN A = the lane's k row
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // B-half a (b = a): 8-row unit 8 (w >> 1) + 4 b + 2 (w & 1) + u
            {
                const int ub = 8 * (w >> 1) + 4 * a + 2 * (w & 1) + u;
                const int row = 8 * ub + (l >> 3);
                const int c = (l & 7) ^ ((row >> 1) & 7);
                int rb = n0 + row;
                if (rb > p.N - 1) rb = p.N - 1;                  // clamped, not predicated
                osrc[2 + a][u] = (uint32_t)((long)rb * p.ldb + 8 * c);
                odst[2 + a][u] = 1024 * ub;
                okk[2 + a][u] = 8 * c;
            }
            if constexpr (TN) {
                // A stored [K][M]: unit 2 w + u of half a = k rows 4 (2w + u) + (l >> 4), 16-byte slot l & 15 of the half's 256-byte row
                const int krow = 4 * (2 * w + u) + (l >> 4);
                const int s16 = (l & 15) ^ (4 * (krow & 3));
                long mc = m0 / 8 + 16 * (s16 >> 3) + 8 * a + (s16 & 7);      // 8-element chunk index along M
                if (mc > p.lda / 8 - 1) mc = p.lda / 8 - 1;
                osrc[a][u] = (uint32_t)((long)krow * p.lda + 8 * mc);
                odst[a][u] = 16384 * a + 1024 * (2 * w + u);
                okk[a][u] = krow;
            } else {
                const int ub = 16 * (w >> 2) + 8 * a + 2 * (w & 3) + u;
                const int row = 8 * ub + (l >> 3);
                const int c = (l & 7) ^ ((row >> 1) & 7);
                int ra = m0 + row;
                if (ra > p.M - 1) ra = p.M - 1;
                osrc[a][u] = (uint32_t)((long)ra * p.lda + 8 * c);
                odst[a][u] = 1024 * ub;
                okk[a][u] = 8 * c;
            }
        }
    // one half-tile (h: 0 = A0, 1 = A1, 2 = B0, 3 = B1) of K tile kt into the buffer pair (Ad, Bd): 2 DMA instructions
    auto stage_half = [&](int kt, int h, LdsTile2& Ad, LdsTile2& Bd) {
        const uint32_t k0 = (uint32_t)(kt * BK);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (h < 2) {
                const uint32_t ka = TN ? k0 * (uint32_t)p.lda : k0;
                lv_glds16_uncounted(p.A + (size_t)(osrc[h][u] + ka), reinterpret_cast<char*>(&Ad[0][0]) + odst[h][u]);
            } else {
                lv_glds16_uncounted(p.B + (size_t)(osrc[h][u] + k0), reinterpret_cast<char*>(&Bd[0][0]) + odst[h][u]);
            }
        }
    };
    auto stage_ragged = [&](int kt, LdsTile2& Ad, LdsTile2& Bd) {      // the ragged last K tile: masked loads through registers
        const int k0 = kt * BK;
        const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                char* dst = reinterpret_cast<char*>(h < 2 ? &Ad[0][0] : &Bd[0][0]) + odst[h][u] + 16 * l;
                if (h < 2 && TN) {
                    *reinterpret_cast<uint4*>(dst) = k0 + okk[h][u] < p.K
                        ? *reinterpret_cast<const uint4*>(p.A + (size_t)(osrc[h][u] + (uint32_t)k0 * (uint32_t)p.lda)) : z4;
                } else {
                    const uint16_t* src = (h < 2 ? p.A : p.B) + (size_t)(osrc[h][u] + (uint32_t)k0);
                    const int k = k0 + okk[h][u];
                    *reinterpret_cast<uint4*>(dst) = k < p.K ? load_chunk_masked(src, p.K - k) : z4;
                }
            }
    };

    const int arow = wm * 128 + li, brow = wn * 64 + li;
    const int sx = (li >> 1) & 7;
    int atr[4] = {0, 0, 0, 0};
    if constexpr (TN) {
        const int r = l & 15, kr = r >> 2;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            const int mh = 32 * (i2 & 1) + 16 * ((l >> 4) & 1) + 4 * (r & 3);          // m inside the (wm, half) block of 64
            const int s16 = 8 * wm + (mh >> 3);
            atr[i2] = 16384 * (i2 >> 1) + (8 * lh + kr) * 256 + ((s16 ^ (4 * kr)) * 16) + (mh & 7) * 2;
        }
    }
    auto a_frag = [&](LdsTile2& Ac, int ks, int i2) -> uint4 {
        if constexpr (TN) {
            const char* base = reinterpret_cast<const char*>(&Ac[0][0]) + atr[i2] + ks * 16 * 256;
            const uint2 lo = lv_ds_read_tr16_b64(base), hi = lv_ds_read_tr16_b64(base + 4 * 256);
            return make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
            return Ac[arow + 32 * i2][(2 * ks + lh) ^ sx];
        }
    };
    auto b_frag = [&](LdsTile2& Bc, int ks, int j) -> uint4 { return Bc[brow + 32 * j][(2 * ks + lh) ^ sx]; };
    // 8 MFMAs of one quadrant: accumulators (2a, b) and (2a + 1, b) alternate, all four k-steps
    auto mma_quadrant = [&](const uint4 (&fa)[4][2], const uint4 (&fb)[4], int a, int b) {
        LV_SETPRIO(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[2 * a + i][b] = lv_mfma_32x32x16_bf16(fa[ks][i], fb[ks], acc[2 * a + i][b]);
        LV_SETPRIO(0);
    };
    // the DMA wait of a phase: before the barrier behind which group 0 reads what it covers -- group 1 at the end of its LOAD
    // segment, group 0 at the end of its MFMA segment; 10 = the DMA instructions of the five half-tiles issued behind the needed one
#define LV_Q_WAIT() LV_WAIT_VMEM_N(10)
    auto seg_close = [&](bool wait) {           // end of a LOAD segment
        LV_SCHED_BARRIER();
        if (wait && wm == 1) LV_Q_WAIT();
        LV_WAIT_LDS();
        LV_S_BARRIER();
        LV_SCHED_BARRIER();
    };
    auto mfma_close = [&](bool wait) {          // end of an MFMA segment
        LV_SCHED_BARRIER();
        if (wait && wm == 0) LV_Q_WAIT();
        LV_S_BARRIER();
        LV_SCHED_BARRIER();
    };
    // One K tile out of (Ac, Bc); kn1 = the tile after it (its half A1 is still to be issued: into (Ao, Bo), the OTHER pair), kn2 =
    // the tile after that (halves A0, B0, B1: into (Ac, Bc) itself, behind the phases that read them).
    auto q_tile = [&](LdsTile2& Ac, LdsTile2& Bc, LdsTile2& Ao, LdsTile2& Bo, int kn1, int kn2) {
        uint4 fa[4][2], fb0[4], fb1[4];
        // phase 0: quadrant (0, 0) -- reads A0 and B0, issues A1 of the next tile
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb0[ks] = b_frag(Bc, ks, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa[ks][0] = a_frag(Ac, ks, 0); fa[ks][1] = a_frag(Ac, ks, 1); }
        LV_SCHED_BARRIER();
        stage_half(kn1, 1, Ao, Bo);
        seg_close(true);                        // ... and the wait that covers B1 of this tile (read in phase 1)
        mma_quadrant(fa, fb0, 0, 0);
        mfma_close(true);
        // phase 1: quadrant (0, 1) -- reads B1, issues A0 of the tile after next (A0 of this pair was read in phase 0 by everybody)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb1[ks] = b_frag(Bc, ks, 1);
        LV_SCHED_BARRIER();
        stage_half(kn2, 0, Ac, Bc);
        seg_close(true);                        // covers A1 of this tile (phase 2)
        mma_quadrant(fa, fb1, 0, 1);
        mfma_close(true);
        // phase 2: quadrant (1, 1) -- reads A1, issues B0 of the tile after next
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fa[ks][0] = a_frag(Ac, ks, 2); fa[ks][1] = a_frag(Ac, ks, 3); }
        LV_SCHED_BARRIER();
        stage_half(kn2, 2, Ac, Bc);
        seg_close(false);
        mma_quadrant(fa, fb1, 1, 1);
        mfma_close(false);
        // phase 3: quadrant (1, 0) -- reads nothing, issues B1 of the tile after next
        stage_half(kn2, 3, Ac, Bc);
        seg_close(true);                        // covers A0 and B0 of the next tile (its phase 0)
        mma_quadrant(fa, fb0, 1, 0);
        mfma_close(true);
    };

    // The ragged tile at the end of a K that is not a multiple of 64 goes FIRST (masked loads through registers, lockstep)
    const int nmain = (kt1 < nfull ? kt1 : nfull) - kt0;
    if (kt1 > kt0 + nmain) {
        stage_ragged(kt1 - 1, As0, Bs0);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 ra[4], rb[2];
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) ra[i2] = a_frag(As0, ks, i2);
#pragma unroll
            for (int j = 0; j < 2; ++j) rb[j] = b_frag(Bs0, ks, j);
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i2][j] = lv_mfma_32x32x16_bf16(ra[i2], rb[j], acc[i2][j]);
        }
        __syncthreads();
    }
    if (nmain > 0) {
        const int klast = kt0 + nmain - 1;
        auto tile_at = [&](int i) { return kt0 + i < klast ? kt0 + i : klast; };      // beyond the end: the last tile again (a harmless reload)
        // prologue: the issue order of the steady state up to the first phase -- all of tile 0, then A0, B0, B1 of tile 1
        stage_half(tile_at(0), 0, As0, Bs0); stage_half(tile_at(0), 2, As0, Bs0); stage_half(tile_at(0), 3, As0, Bs0);
        stage_half(tile_at(0), 1, As0, Bs0);
        stage_half(tile_at(1), 0, As1, Bs1); stage_half(tile_at(1), 2, As1, Bs1); stage_half(tile_at(1), 3, As1, Bs1);
        LV_Q_WAIT();                            // A0, B0 of tile 0 (10 newer instructions behind them)
        __syncthreads();
        if (wm == 1) LV_S_BARRIER();            // group 1 runs one segment behind group 0
        for (int i = 0; i < nmain; i += 2) {
            q_tile(As0, Bs0, As1, Bs1, tile_at(i + 1), tile_at(i + 2));
            if (i + 1 >= nmain) break;
            q_tile(As1, Bs1, As0, Bs0, tile_at(i + 2), tile_at(i + 3));
        }
        if (wm == 0) LV_S_BARRIER();            // ... and is met again here
        LV_WAIT_VMEM();                         // the reloads issued in the last two tiles: nothing may still land in LDS
        __syncthreads();
    }
#undef LV_Q_WAIT
}

template <bool NLL, bool TN>
__global__ __launch_bounds__(512) void lv_gemm_b16_t256q_kernel(GemmQ p, Tail256 q) {
    __shared__ __attribute__((aligned(1024))) LdsTile2 As0, Bs0;
    __shared__ __attribute__((aligned(1024))) LdsTile2 As1, Bs1;

    const T256Pick pk = t256_pick(p, q, (int)blockIdx.x);
    const int m0 = pk.tm * BT2, n0 = pk.tn * BT2;
    const int t = (int)threadIdx.x;
    const int l = t & 63, w = lv_wave_uniform(t >> 6);

    f32x16 acc[4][2];
    t256q_run<TN>(p, m0, n0, pk.kt0, pk.kt1, acc, As0, Bs0, As1, Bs1, l, w);
    t256_epilogue<NLL>(p, q, acc, As0, Bs0, As1, Bs1, pk.tile, pk.piece, pk.tn, m0, n0, t, l, w >> 2, w & 3, l >> 5);
}

// ---- grouped stream-K launch over the quadrant K loop (round 6) ---------------------------------------------------------------
// The LSTM-sized gradient products do not fill 256 CUs with 256 x 256 tiles (dW_ih | dW_hh as one product: 96 tiles; dX: 50), which
// is why they ran on the 128 x 128 kernel (twice the L2 -> LDS bytes per flop) under split-K with a reduction launch each.  Here ONE
// launch of exactly one workgroup per CU works on up to TWO independent products: product j owns `rows_j` of the 32 workgroup rows
// (8 workgroups each, one per XCD), its tiles x K tiles are laid out as one line of UNITS in tile-major order, and logical workgroup s
// of its 8 rows_j takes units [s U / n, (s + 1) U / n) -- a run of SEGMENTS (tile, K-tile range), each through t256q_run.  Where the
// unit count per workgroup divides the K tiles of a tile the segments are aligned pieces (the weight gradients at the bench shapes:
// two halves per tile, and an XCD's workgroups -- consecutive logical ids -- walk neighbouring tiles over the same K range, sharing
// operand tiles in its L2); where it does not (dX on the rows that are left) a workgroup ends one tile and starts the next.
// A segment that covers its tile's whole K range goes straight to C.  Any other is handed over inside the launch: the accumulators
// leave as a 256 KB slab of write-through stores, one lane draws a ticket from the tile's arrival counter, and whichever workgroup
// arrives LAST adds the tile's slabs in K order (its own included, read back: the order of the additions never depends on who is
// last) and writes C.  Nobody waits for anybody: no residency or dispatch-order assumption, nothing to time out, and on a sequential
// executor (the CI emulator) the last workgroup to run simply finds all slabs there.  The counters live in a zero-initialised device
// array; the last arriver puts its counter back to zero, and concurrent launches take different slots of the array.
#ifndef LV_SK_ABL
#define LV_SK_ABL 0       // measurement only (profiles/microbench/gemm_pair_probe.py; results are wrong): 1 = partial tiles are dropped (no slab, no ticket, no sum), 2 = no C stores
#endif
#ifndef LV_SK_SPIN
#define LV_SK_SPIN LV_ARRIVAL_POLLS     // polls the closing contributor of a tile spends looking for the others before it hands its own piece over too; 0 = never
#endif
#ifndef LV_SK_SKEW
#define LV_SK_SKEW 3      // K tiles the closing piece of an aligned tile is longer (and its first piece shorter) than an even split; 0 = even (A/B knob)
#endif
#ifndef LV_SK_ROWS0
#define LV_SK_ROWS0 0     // measurement only: workgroup rows of the first product (0 = by the cost model)
#endif
constexpr int SK_SLOTS = 32;                 // launches that may hold arrival counters at the same time (round-robin)
constexpr int SK_TILES = 1024;               // 256 x 256 tiles per launch, both products together
__device__ unsigned lv_sk_arrivals[SK_SLOTS * SK_TILES];

struct SkProb {
    GemmQ q;              // operands, shape, tilesM / tilesN (256-tiles), C / ldc (+ C2 / ldc2 / nsplit: two destinations)
    int rows;             // workgroup rows of the launch that work on this product (0: none); workgroups = 8 rows
    int nk;               // K tiles
    long units;           // tilesM * tilesN * nk
    float* slabs;         // 2 slabs of 256 x 256 floats per workgroup: [2 s] the partial tile its range starts in, [2 s + 1] a later one
    int cnt0;             // index of this product's first arrival counter (one per tile) in lv_sk_arrivals
    // skew > 0 (only where the ranges are `pieces` whole aligned pieces of `plen` K tiles per tile): every boundary INSIDE a tile lies
    // `skew` K tiles early -- the first piece of a tile is that much shorter, the last (closing) one that much longer, so that the
    // others' slabs are on their way while the closing workgroup still multiplies and it finds them arrived when it looks
    int skew, pieces, plen;
};
struct GemmG { SkProb pr[2]; };

// first unit of logical workgroup s (s = nwg: one past the last unit), and the workgroup whose range holds unit u
__device__ __forceinline__ long sk_start(const SkProb& P, int nwg, int s) {
    const long b = (long)s * P.units / nwg;
    return (P.skew > 0 && s % P.pieces != 0) ? b - P.skew : b;
}
__device__ __forceinline__ int sk_owner(const SkProb& P, int nwg, long u) {
    if (P.skew > 0) {
        const int tile = (int)(u / P.nk), k = (int)(u - (long)tile * P.nk);
        int j = k / P.plen;
        if (j + 1 < P.pieces && k >= (j + 1) * P.plen - P.skew) ++j;
        return tile * P.pieces + j;
    }
    return (int)(((u + 1) * nwg - 1) / P.units);
}

// a finished 256 x 256 tile -> C (columns >= nsplit of a two-destination product -> C2)
__device__ __forceinline__ void sk_store_tile(const GemmQ& p, const f32x16 (&acc)[4][2], int m0, int n0, int l, int w) {
    const int wm = w >> 2, wn = w & 3;
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (l & 31);
            if (col >= p.N) continue;
            const int rbase = m0 + wm * 128 + i2 * 32 + 4 * (l >> 5);
            float* cb = (p.nsplit > 0 && col >= p.nsplit) ? p.C2 + (long)rbase * p.ldc2 + (col - p.nsplit) : p.C + (long)rbase * p.ldc + col;
            const long ld = (p.nsplit > 0 && col >= p.nsplit) ? p.ldc2 : p.ldc;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = (e & 3) + 8 * (e >> 2);
                if (rbase + ro < p.M) cb[(long)ro * ld] = p.alpha * acc[i2][j][e];
            }
        }
}

// acc += the slab at sl (this thread's 32 float4 of it), four float4 in flight at a time
__device__ __forceinline__ void sk_add_slab(f32x16 (&acc)[4][2], const float4* sl) {
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 v[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) v[a] = sl[((i2 * 2 + j) * 4 + a) * 512];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[i2][j][4 * a] += v[a].x; acc[i2][j][4 * a + 1] += v[a].y;
                acc[i2][j][4 * a + 2] += v[a].z; acc[i2][j][4 * a + 3] += v[a].w;
            }
        }
}

template <bool TN>
__device__ __forceinline__ void sk_work(const SkProb& P, int s, LdsTile2& As0, LdsTile2& Bs0, LdsTile2& As1, LdsTile2& Bs1, int t, int l, int w) {
    const int nwg = 8 * P.rows;
    const long u0 = sk_start(P, nwg, s), u1 = sk_start(P, nwg, s + 1);
    const int first_tile = (int)(u0 / P.nk);
    unsigned* const flag = reinterpret_cast<unsigned*>(&As0[0][0]);          // free between two segments (see t256q_run)
    unsigned* const cnt = lv_sk_arrivals + P.cnt0;
    // The segments of the range from its END: the piece of a tile this workgroup runs last is then the one that closes that tile's K
    // range (every other contributor of the tile either ran its piece first, long ago, or finishes at the same time).
    for (long e = u1; e > u0;) {
        const int tile = (int)((e - 1) / P.nk);
        const long t0 = (long)tile * P.nk;
        const int ke = (int)(e - t0), kb = u0 > t0 ? (int)(u0 - t0) : 0;
        e = t0 + kb;
        int tm, tn;
        t256_tile_of(P.q, tile, tm, tn);
        const int m0 = tm * BT2, n0 = tn * BT2;
        f32x16 acc[4][2];
        t256q_run<TN>(P.q, m0, n0, kb, ke, acc, As0, Bs0, As1, Bs1, l, w);
        bool done = kb == 0 && ke == P.nk;
        if (!done && !(LV_SK_ABL & 1)) {
            // ---- a partial tile: handed over inside the launch.  The tile's value is DEFINED as the sum of its pieces in descending K
            // order, (((p_c1 + p_c1-1) + ...) + p_c0), whoever computes it.
            const int c0 = sk_owner(P, nwg, t0), c1 = sk_owner(P, nwg, t0 + P.nk - 1);
            // (1) The contributor that closes the K range, on the last segment it runs, looks (for a bounded while: it has nothing
            // else left to do) whether everybody else has arrived; if so it adds their slabs to its registers in that order and its own
            // piece never travels.
            if (t == 0) {
                int others = -1;                           // contributors besides this one: the workgroups of [c0, c1] whose range is not empty
                for (int c = c0; c <= c1; ++c) others += sk_start(P, nwg, c) < sk_start(P, nwg, c + 1);
                unsigned role = 0;                         // 0: hand the piece over, 1: sum the others' into the registers
                if (s == c1 && e == u0) {
                    for (int spin = 0; spin < LV_SK_SPIN; ++spin) {
                        if (lv_agent_load_u32(cnt + tile) == (unsigned)others) { role = 1; break; }
                        lv_sleep_short();
                    }
                    if (role) {
                        atomicExch(cnt + tile, 0u);        // nobody else touches it any more: ready for the next launch
                        lv_acquire_agent();
                    }
                }
                flag[0] = role; flag[1] = (unsigned)others;
            }
            __syncthreads();
            const bool closer = lv_wave_uniform((int)flag[0]) != 0;
            const int others = lv_wave_uniform((int)flag[1]);
            __syncthreads();
            bool last = false;
            if (!closer) {
                // (2) Everybody else: the piece leaves as a slab (element q of thread t at float4 slot (q / 4) * 512 + t: every store
                // instruction of the workgroup is 8 KB contiguous) of write-through stores, drained by every wave, then ONE ticket;
                // whoever draws the last one reads all slabs back, its own included, in the defining order.
                float4* const mine = reinterpret_cast<float4*>(P.slabs + (2L * s + (tile != first_tile)) * (BT2 * BT2));
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int a = 0; a < 4; ++a)
                            lv_store_wt_f4(mine + ((i2 * 2 + j) * 4 + a) * 512 + t,
                                           make_float4(acc[i2][j][4 * a], acc[i2][j][4 * a + 1], acc[i2][j][4 * a + 2], acc[i2][j][4 * a + 3]));
                LV_WAIT_VMEM();
                __syncthreads();
                if (t == 0) {
                    const bool l0 = atomicAdd(cnt + tile, 1u) == (unsigned)others;
                    if (l0) {
                        atomicExch(cnt + tile, 0u);
                        lv_acquire_agent();
                    }
                    flag[0] = l0 ? 1u : 0u;
                }
                __syncthreads();
                last = lv_wave_uniform((int)flag[0]) != 0;
                __syncthreads();                           // the flag's LDS words belong to the next segment's first K tile again
                if (last) {
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int q = 0; q < 16; ++q) acc[i2][j][q] = 0.f;
                }
            }
            const bool summed = closer || last;
            if (summed) {                                  // p_c1 (in the registers, or read back first into zeroed ones), then c1 - 1 ... c0
                for (int c = closer ? c1 - 1 : c1; c >= c0; --c) {
                    const long cu0 = sk_start(P, nwg, c);
                    if (cu0 == sk_start(P, nwg, c + 1)) continue;                // (an empty range: fewer units than workgroups)
                    sk_add_slab(acc, reinterpret_cast<const float4*>(P.slabs + (2L * c + (tile != (int)(cu0 / P.nk))) * (BT2 * BT2)) + t);
                }
            }
            done = summed;
        }
#if LV_SK_ABL & 2
        if (done && acc[0][0][0] == 1.2345e-30f) sk_store_tile(P.q, acc, m0, n0, l, w);
#else
        if (done) sk_store_tile(P.q, acc, m0, n0, l, w);
#endif
    }
}

// diagnostic: how many arrival counters are not back at zero (between launches: none)
__global__ __launch_bounds__(256) void sk_pending_kernel(int* out) {
    int n = 0;
    for (int i = (int)threadIdx.x; i < SK_SLOTS * SK_TILES; i += 256) n += lv_sk_arrivals[i] != 0u;
    n = lv_wave_sum(n);
    __shared__ int part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) *out = part[0] + part[1] + part[2] + part[3];
}

template <bool TN0, bool TN1>
__global__ __launch_bounds__(512) void lv_gemm_b16_t256g_kernel(GemmG g) {
    __shared__ __attribute__((aligned(1024))) LdsTile2 As0, Bs0;
    __shared__ __attribute__((aligned(1024))) LdsTile2 As1, Bs1;
    const int t = (int)threadIdx.x;
    const int l = t & 63, w = lv_wave_uniform(t >> 6);
    const int bid = (int)blockIdx.x, xcd = bid % 8, row = bid / 8;           // (workgroup b runs on XCD b % 8: for L2 locality only)
    if (row < g.pr[0].rows) sk_work<TN0>(g.pr[0], xcd * g.pr[0].rows + row, As0, Bs0, As1, Bs1, t, l, w);
    else sk_work<TN1>(g.pr[1], xcd * g.pr[1].rows + (row - g.pr[0].rows), As0, Bs0, As1, Bs1, t, l, w);
}

// the K pieces of the tail tiles, added in piece order, + the epilogue; one workgroup per (tail tile, TR_ROWS rows).  TR_ROWS = 8 since
// round 6 (32 before): a thread's row iterations are dependent memory round trips (4 pieces in flight each), and with 8 of them on
// 480 workgroups the launch ran at 2 TB/s (dW_pred's: 31 us for 63 MB)
#ifndef LV_TR_ROWS
#define LV_TR_ROWS 8
#endif
constexpr int TR_ROWS = LV_TR_ROWS, TR_WGS = BT2 / TR_ROWS;
__global__ __launch_bounds__(256) void tail_reduce_t256_kernel(GemmQ p, Tail256 q) {
    const int tile = q.full + (int)blockIdx.x / TR_WGS;
    int tm, tn;
    t256_tile_of(p, tile, tm, tn);
    const float* slab = p.ws + (long)(tile - q.full) * q.tail_s * (BT2 * BT2);
    const int t = (int)threadIdx.x;
    const int c4 = (t & 63) * 4;
    float ss = 0.f;
    const bool fast = !p.add1 && !p.add2 && !p.accumulate && p.ldc % 4 == 0 && p.N % 4 == 0 && (((uintptr_t)p.C) & 15) == 0 &&
                      (((uintptr_t)p.keep) & 3) == 0;
#pragma unroll
    for (int it = 0; it < TR_ROWS / 4; ++it) {
        const int rr = ((int)blockIdx.x % TR_WGS) * TR_ROWS + 4 * it + (t >> 6);
        const int row = tm * BT2 + rr;
        if (row >= p.M) continue;
        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
        // the pieces four at a time (clamped: the surplus loads are never added): one memory round trip per batch instead of one per
        // piece (dO: five pieces; 48 -> 3x us)
        for (int k0 = 0; k0 < q.tail_s; k0 += 4) {
            float4 pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                pv[u] = *reinterpret_cast<const float4*>(slab + (long)(k0 + u < q.tail_s ? k0 + u : k0) * (BT2 * BT2) + rr * BT2 + c4);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < q.tail_s) { sacc.x += pv[u].x; sacc.y += pv[u].y; sacc.z += pv[u].z; sacc.w += pv[u].w; }
        }
        const float sv[4] = {sacc.x, sacc.y, sacc.z, sacc.w};
        if (fast && tn * BT2 + c4 + 3 < p.N) {
            // whole float4 inside C, no addends: one mask word, one 16-byte store
            const int col = tn * BT2 + c4;
            float4 o = make_float4(p.alpha * sv[0], p.alpha * sv[1], p.alpha * sv[2], p.alpha * sv[3]);
            if (p.keep) {
                const uint32_t mk = *reinterpret_cast<const uint32_t*>(p.keep + ((long)(row % p.Bsz) * p.keepT + row / p.Bsz) * p.N + col);
                o.x = (mk & 0xFFu) ? o.x * p.kscale : 0.f;
                o.y = (mk & 0xFF00u) ? o.y * p.kscale : 0.f;
                o.z = (mk & 0xFF0000u) ? o.z * p.kscale : 0.f;
                o.w = (mk & 0xFF000000u) ? o.w * p.kscale : 0.f;
            }
            ss += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
            if (!p.sq_only) *reinterpret_cast<float4*>(p.C + (long)row * p.ldc + col) = o;
            continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = tn * BT2 + c4 + e;
            if (col >= p.N) continue;
            float v = p.alpha * sv[e];
            if (p.add1) v += p.add1[(long)(row % p.mod1) * p.ld1 + col];
            if (p.add2) v += p.add2[(long)(row % p.mod2) * p.ld2 + col];
            float* c = p.C + (long)row * p.ldc + col;
            if (p.accumulate) v += *c;
            if (p.keep) v *= p.keep[((long)(row % p.Bsz) * p.keepT + row / p.Bsz) * p.N + col] ? p.kscale : 0.f;
            ss += v * v;
            if (!p.sq_only) *c = v;
        }
    }
    if (p.sq) {
        ss = lv_wave_sum(ss);
        if ((t & 63) == 0) p.sq[(long)p.tilesM * p.tilesN * 8 + (long)blockIdx.x * 4 + (t >> 6)] = ss;
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

// four consecutive outputs per thread (N % 4 == 0, no addends: the LSTM-sized products of the step): 16-byte loads of the pieces, four
// pieces in flight, one mask word, one 16-byte store; every output is still the sum of its pieces in piece order
__global__ __launch_bounds__(256) void splitk_reduce_b16_v4_kernel(GemmQ p) {
    const long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long MN = (long)p.M * p.N;
    if (idx >= MN) return;
    const int row = (int)(idx / p.N), col = (int)(idx % p.N);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < p.splits; k0 += 4) {
        float4 pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[u] = *reinterpret_cast<const float4*>(p.ws + (long)(k0 + u < p.splits ? k0 + u : k0) * MN + idx);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < p.splits) { s.x += pv[u].x; s.y += pv[u].y; s.z += pv[u].z; s.w += pv[u].w; }
    }
    float4 o = make_float4(p.alpha * s.x, p.alpha * s.y, p.alpha * s.z, p.alpha * s.w);
    if (p.keep) {
        const uint32_t mk = *reinterpret_cast<const uint32_t*>(p.keep + ((long)(row % p.Bsz) * p.keepT + row / p.Bsz) * p.N + col);
        o.x = (mk & 0xFFu) ? o.x * p.kscale : 0.f;
        o.y = (mk & 0xFF00u) ? o.y * p.kscale : 0.f;
        o.z = (mk & 0xFF0000u) ? o.z * p.kscale : 0.f;
        o.w = (mk & 0xFF000000u) ? o.w * p.kscale : 0.f;
    }
    if (p.nsplit > 0 && col >= p.nsplit) *reinterpret_cast<float4*>(p.C2 + (long)row * p.ldc2 + (col - p.nsplit)) = o;
    else *reinterpret_cast<float4*>(p.C + (long)row * p.ldc + col) = o;
}

__global__ __launch_bounds__(256) void splitk_reduce_b16_kernel(GemmQ p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long MN = (long)p.M * p.N;
    if (idx >= MN) return;
    const int row = (int)(idx / p.N), col = (int)(idx % p.N);
    // the pieces in flight together (2, 4 or 8 at a time by the split count), added in piece order: one load -> add per iteration
    // was one memory round trip per piece
    float s = 0.f;
    if (p.splits <= 2) s = sum_pieces<2>(p.ws + idx, MN, p.splits);
    else if (p.splits <= 4) s = sum_pieces<4>(p.ws + idx, MN, p.splits);
    else s = sum_pieces<8>(p.ws + idx, MN, p.splits);
    float v = p.alpha * s;
    if (p.add1) v += p.add1[(long)(row % p.mod1) * p.ld1 + col];
    if (p.add2) v += p.add2[(long)(row % p.mod2) * p.ld2 + col];
    float* c = (p.nsplit > 0 && col >= p.nsplit) ? p.C2 + (long)row * p.ldc2 + (col - p.nsplit) : p.C + (long)row * p.ldc + col;
    if (p.accumulate) v += *c;
    if (p.keep) v *= p.keep[((long)(row % p.Bsz) * p.keepT + row / p.Bsz) * p.N + col] ? p.kscale : 0.f;
    *c = v;
}

// f32 [R][C] -> bf16 [R][C] (dst) and/or bf16 [C][R] (dstT), RNE; 64x64 tiles, every global access a full row
// segment (256 B reads, 128 B writes); the transposed copy goes through a pitch-66 LDS tile (bank stride 33).
// gate_H > 0: the rows of src are the 4H gate rows of an LSTM weight (g*H + u); dst gets them in unit-major order
// (row u*4 + g) so that the GEMM it feeds emits each unit's four gate pre-activations side by side; dstT is unaffected.
// keep != nullptr: src is a time-major activation [T*Bsz][C] (row t*Bsz + b) and keep the reference-layout dropout mask
// [Bsz][T][C]: the image is taken of src * (keep ? kscale : 0) -- nn.Dropout applied while converting, with mask loads that run
// along C (the persistent recurrence would read the same bytes 8 per row and step: +0.3 us per timestep measured).
// gids != nullptr: the source rows are GATHERED -- row r = t*Bsz + b of the image is row clamp(gids[b * gstride + t], 0, gV - 1) of
// src (nn.Embedding lookup fused with the conversion: the f32 activation matrix is never written).
__global__ __launch_bounds__(256) void cvt_b16_kernel(const float* __restrict__ src, long lds_, int R, int C,
                                                      uint16_t* __restrict__ dst, long ldd, uint16_t* __restrict__ dstT, long ldt,
                                                      int gate_H, const uint8_t* __restrict__ keep, float kscale, int Bsz,
                                                      const int64_t* __restrict__ gids, long gstride, int gV, int lo) {
    // lo == 1: the image of the RESIDUAL x - bf16(x) (itself rounded to bf16, RNE): the low half of a split-bf16 operand
    // (x = hi + lo up to 2^-17 |x|), see lv_cvt_bf16_lo_f32
    // lo == 2: dst as IEEE binary16 (RNE), dstT as bf16: the operand of a FORWARD product that takes the binary16 matrix pipe
    // (lv_gemm_h16) next to the transposed bf16 image a gradient product reads, see lv_cvt_h16_f32
    __shared__ uint16_t tile[64][66];
    const int t = (int)threadIdx.x;
    const int r0 = (int)blockIdx.y * 64, c0 = (int)blockIdx.x * 64;
    const int lane = t & 63, q = t >> 6;
    // (time step, batch row) of the thread's current row, advanced by 4 rows per iteration: no division per element (there is no
    // integer divide instruction; two 64-bit ones per element made the masked conversion 31 us instead of 20)
    const bool tb = keep || gids;
    const int Tt = tb ? R / Bsz : 1;
    int bb = tb ? (r0 + q) % Bsz : 0, tt = tb ? (r0 + q) / Bsz : 0;
    // Eight rows at a time, every stage of their loads issued together (row index of a gathered source, the source element, the
    // keep byte), from addresses CLAMPED into the matrix; the stores are predicated afterwards.  The first form loaded inside
    // `if (gr < R && gc < C)`: a thread's 16 rows were 16 (gathered: 32) dependent memory round trips, and with one round of
    // workgroups on the chip that chain WAS the kernel's duration (17 us for 52 MB).
    const int gc = c0 + lane, gcc = gc < C ? gc : C - 1;
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 8) {
        int grs[8], bbs[8], tts[8];
        long sr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            grs[u] = r0 + q + 4 * (i0 + u);
            bbs[u] = bb; tts[u] = tt;
            if (tb) {
                bb += 4;
                while (bb >= Bsz) { bb -= Bsz; ++tt; }
            }
            sr[u] = grs[u] < R ? grs[u] : R - 1;
        }
        if (gids) {
#pragma unroll
            for (int u = 0; u < 8; ++u) sr[u] = gids[(long)bbs[u] * gstride + (grs[u] < R ? tts[u] : 0)];
#pragma unroll
            for (int u = 0; u < 8; ++u) sr[u] = sr[u] < 0 ? 0 : (sr[u] >= gV ? gV - 1 : sr[u]);
        }
        float v[8];
        uint8_t kp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[sr[u] * lds_ + gcc];
        if (keep) {
#pragma unroll
            for (int u = 0; u < 8; ++u) kp[u] = keep[((long)bbs[u] * Tt + (grs[u] < R ? tts[u] : 0)) * C + gcc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int gr = grs[u];
            uint16_t b = 0;
            if (gr < R && gc < C) {
                float x = v[u];
                if (keep) {
                    if (gids) x = kp[u] ? x * kscale : 0.f;      // as lv_embed_gather_f32 writes it (+0 for a dropped element)
                    else x *= kp[u] ? kscale : 0.f;              // as h * (keep * scale) rounds (a dropped negative element is -0)
                }
                b = (uint16_t)lv_f32_to_bf16_bits(x);
                if (lo == 1) b = (uint16_t)lv_f32_to_bf16_bits(x - lv_bf16_bits_to_f32(b));
                if (dst) {
                    const long dr = gate_H > 0 ? (long)(gr % gate_H) * 4 + gr / gate_H : gr;
                    dst[dr * ldd + gc] = lo == 2 ? lv_f32_to_f16_bits(lv_sat_f16(x)) : b;      // binary16 saturates, never inf; NaN stays NaN
                }
            }
            tile[q + 4 * (i0 + u)][lane] = b;
        }
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int c = q + 4 * i;
        const int gc = c0 + c, gr = r0 + lane;
        if (gc < C && gr < R) dstT[(long)gc * ldt + gr] = tile[lane][c];
    }
}


// The same conversion with 16-byte loads and 8-byte stores: lane (row-in-pass t >> 4, column quad t & 15) takes four CONSECUTIVE
// columns of its rows (one float4 load, one 4-byte mask load, one 8-byte store of the row image), and the transposed image leaves as
// 8-byte stores of four consecutive rows of a column.  A vector-memory instruction costs the address unit a quad of lanes per cycle
// whatever its width (the lesson of the persistent recurrences' hand-off): the 4-byte loads / 2-byte stores
// of cvt_b16_kernel kept a CU at 8-16 bytes per cycle -- 48 instructions per thread and tile here become 12.  Same arguments, same
// arithmetic, bit-identical images.  Needs C % 4 == 0 and 16- / 8-byte aligned rows (cvt_launch falls back to cvt_b16_kernel).
__global__ __launch_bounds__(256) void cvt_b16_v4_kernel(const float* __restrict__ src, long lds_, int R, int C,
                                                         uint16_t* __restrict__ dst, long ldd, uint16_t* __restrict__ dstT, long ldt,
                                                         int gate_H, const uint8_t* __restrict__ keep, float kscale, int Bsz,
                                                         const int64_t* __restrict__ gids, long gstride, int gV, int lo) {
    __shared__ __attribute__((aligned(8))) uint16_t tile[64][68];      // pitch 136 bytes: 8-byte aligned rows
    const int t = (int)threadIdx.x;
    const int r0 = (int)blockIdx.y * 64, c0 = (int)blockIdx.x * 64;
    const int q = t >> 4, cl = t & 15;
    const bool tb = keep || gids;
    const int Tt = tb ? R / Bsz : 1;
    int bb = tb ? (r0 + q) % Bsz : 0, tt = tb ? (r0 + q) / Bsz : 0;
    const int gc = c0 + 4 * cl;
    const int gcc = gc < C ? gc : C - 4;                  // clamped column quad (C % 4 == 0, C >= 4)
    int grs[4], bbs[4], tts[4];
    long sr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        grs[u] = r0 + q + 16 * u;
        bbs[u] = bb; tts[u] = tt;
        if (tb) {
            bb += 16;
            while (bb >= Bsz) { bb -= Bsz; ++tt; }
        }
        sr[u] = grs[u] < R ? grs[u] : R - 1;
    }
    if (gids) {
#pragma unroll
        for (int u = 0; u < 4; ++u) sr[u] = gids[(long)bbs[u] * gstride + (grs[u] < R ? tts[u] : 0)];
#pragma unroll
        for (int u = 0; u < 4; ++u) sr[u] = sr[u] < 0 ? 0 : (sr[u] >= gV ? gV - 1 : sr[u]);
    }
    float4 v[4];
    uint32_t kp[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + sr[u] * lds_ + gcc);
    if (keep) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            kp[u] = *reinterpret_cast<const uint32_t*>(keep + ((long)bbs[u] * Tt + (grs[u] < R ? tts[u] : 0)) * C + gcc);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int gr = grs[u];
        uint16_t b[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
        const bool in = gr < R && gc < C;
        if (in) {
            const float xs[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = xs[e];
                if (keep) {
                    const bool k = (kp[u] >> (8 * e)) & 0xFFu;
                    if (gids) x = k ? x * kscale : 0.f;
                    else x *= k ? kscale : 0.f;
                }
                b[e] = (uint16_t)lv_f32_to_bf16_bits(x);
                if (lo == 1) b[e] = (uint16_t)lv_f32_to_bf16_bits(x - lv_bf16_bits_to_f32(b[e]));
                d[e] = lo == 2 ? lv_f32_to_f16_bits(lv_sat_f16(x)) : b[e];
            }
            if (dst) {
                const long dr = gate_H > 0 ? (long)(gr % gate_H) * 4 + gr / gate_H : gr;
                *reinterpret_cast<uint2*>(dst + dr * ldd + gc) = make_uint2((uint32_t)d[0] | ((uint32_t)d[1] << 16), (uint32_t)d[2] | ((uint32_t)d[3] << 16));
            }
        }
        *reinterpret_cast<uint2*>(&tile[q + 16 * u][4 * cl]) = make_uint2((uint32_t)b[0] | ((uint32_t)b[1] << 16), (uint32_t)b[2] | ((uint32_t)b[3] << 16));
    }
    if (!dstT) return;
    __syncthreads();
    // transposed image: lane (column-in-pass t >> 4, row quad t & 15) stores rows 4 rl .. 4 rl + 3 of column c
    const int rl = t & 15, gr = r0 + 4 * rl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = q + 16 * i, gct = c0 + c;
        const uint16_t e0 = tile[4 * rl][c], e1 = tile[4 * rl + 1][c], e2 = tile[4 * rl + 2][c], e3 = tile[4 * rl + 3][c];
        if (gct < C) {
            uint16_t* o = dstT + (long)gct * ldt + gr;
            if (gr + 3 < R) *reinterpret_cast<uint2*>(o) = make_uint2((uint32_t)e0 | ((uint32_t)e1 << 16), (uint32_t)e2 | ((uint32_t)e3 << 16));
            else {
                if (gr < R) o[0] = e0;
                if (gr + 1 < R) o[1] = e1;
                if (gr + 2 < R) o[2] = e2;
            }
        }
    }
}

// picks the 16-byte form where the operands allow it
static void cvt_launch(void* stream, const float* src, long lds, int R, int C, uint16_t* dst, long ldd, uint16_t* dstT, long ldt, int gate_H,
                       const uint8_t* keep, float kscale, int Bsz, const int64_t* gids, long gstride, int gV, int lo) {
    const dim3 grid((unsigned)lv_cdiv(C, 64), (unsigned)lv_cdiv(R, 64)), block(256);
    const bool v4 = C >= 4 && C % 4 == 0 && lds % 4 == 0 && (((uintptr_t)src) & 15) == 0 &&
                    (!dst || (ldd % 4 == 0 && (((uintptr_t)dst) & 7) == 0)) && (!dstT || (ldt % 4 == 0 && (((uintptr_t)dstT) & 7) == 0)) &&
                    (!keep || (((uintptr_t)keep) & 3) == 0);
    if (v4) LV_LAUNCH(cvt_b16_v4_kernel, grid, block, 0, stream, src, lds, R, C, dst, ldd, dstT, ldt, gate_H, keep, kscale, Bsz, gids, gstride, gV, lo);
    else LV_LAUNCH(cvt_b16_kernel, grid, block, 0, stream, src, lds, R, C, dst, ldd, dstT, ldt, gate_H, keep, kscale, Bsz, gids, gstride, gV, lo);
}

}  // namespace

// ---- tile selection ------------------------------------------------------------------------------------------------------
// the 256 x 256 kernel pays on the vocabulary-sized products (2.6e11 flop at the Yahoo shape); below ~1e11 the 128 x 128 kernel's
// finer grid and shorter prologue win (measured: profiles/microbench/gemm_b16_shapes.py).  tile: 0 = by shape, 128 / 256 = the
// caller names the tile edge (lv_gemm_b16_tile / lv_gemm_b16_nll_tile: tests and microbenchmarks; no process-wide state).
static bool t256_wanted(int tile, int M, int N, int K) {
    if (tile) return tile >= 256;
    return 2.0 * M * N * K >= 1.0e11 && M >= 1024 && N >= 1024 && K >= 1024;
}
#ifndef LV_B16_SCHED_DEFAULT
#define LV_B16_SCHED_DEFAULT -1   // schedule of the 256 x 256 kernel when the caller does not name it: -1 = by K tiles per workgroup, 0 = lockstep, 1 = ping-pong by k-steps, 2 = ping-pong by quadrants with a continuous DMA stream
#endif
// tile argument of the *_tile entries: 0 = by shape, 128, 256 / 257 / 258 = the 256 x 256 tile on schedule 0 / 1 / 2
// By shape (measured, profiles/r04f_gemm_schedules.txt): the quadrant schedule wins where a workgroup walks many K tiles (dO: 63 per
// piece, 278 -> 270 us; dW_pred: 100 / 25, 274 -> 258 us; 8192^3 1234 -> 1264 TF), the k-step ping-pong where it walks few and the
// prologue (seven half-tiles before the first MFMA) shows (logits, K = 1024: 271 vs 280 us; fused NLL 308 vs 322; 4096^3).
static int t256_sched(int tile, int kt_per_wg) {
    if (tile >= 256) return tile - 256;
    return LV_B16_SCHED_DEFAULT >= 0 ? LV_B16_SCHED_DEFAULT : (kt_per_wg >= 24 ? 2 : 1);
}

// How the tail (tiles % 256) of a 256 x 256 launch is cut along K: estimated microseconds for s pieces per tail tile =
// rounds x K tiles per piece x ~2 us per tile step + the slab traffic of the reduce ((s + 1) passes over tail x 256 KB at ~4.5 TB/s)
static Tail256 t256_plan(long tiles, int nk, long ws_floats) {
    Tail256 q;
    q.full = (int)(tiles / 256 * 256);
    q.tail_s = 1;
    q.kt_per_piece = nk;
    const long tail = tiles - q.full;
    q.tail = (int)tail;
    if (tail == 0) return q;
    double best = (double)nk * 2.0;
    for (int s = 2; s <= 16; ++s) {
        if (nk / s < 4 || tail * s * (long)(BT2 * BT2) > ws_floats) break;
        const double cost = (double)lv_cdiv(tail * s, 256) * lv_cdiv(nk, s) * 2.0 + (s + 1) * tail * 0.058;
        if (cost < best) { best = cost; q.tail_s = s; }
    }
    q.kt_per_piece = lv_cdiv(nk, q.tail_s);
    q.tail_s = lv_cdiv(nk, q.kt_per_piece);
    return q;
}

// C[M,N] = alpha * op(A) . B^T (+ add1 + add2 (+ C)), bf16 operands, f32 accumulate/output.
// A: transA == 0 -> stored [M][K] (lda >= K); transA == 1 -> stored [K][M] (lda >= M).  B: stored [N][K] (ldb >= K).
// All leading dimensions % 8 == 0 and bases 16 B-aligned (else LV_ERR_ALIGN).  Rows may be read up to the next
// multiple of 8 elements past their logical end (never past ld); what lies there never reaches C.
extern "C" int lv_keep_scale_f32(float* x, const uint8_t* keep, float kscale, int T, int Bsz, int C, void* stream);

static int gemm_b16_launch(int tile, int transA, int M, int N, int K, float alpha,
                           const uint16_t* A, long lda, const uint16_t* B, long ldb,
                           float* C, long ldc, int accumulate,
                           const float* add1, long ld1, int mod1,
                           const float* add2, long ld2, int mod2,
                           float* ws, long ws_floats, void* stream, const uint8_t* keep, float kscale, int Bsz,
                           float* sq = nullptr, int sq_only = 0, int f16 = 0, float* C2 = nullptr, long ldc2 = 0, int nsplit = 0) {
    if (tile != 0 && tile != 128 && (tile < 256 || tile > 258)) return LV_ERR_ARG;
    if (f16) {                       // binary16 operands: the K-contiguous 128-tile LDS-DMA kernel only (forward products)
        if (transA || keep || sq || !LV_B16_GLDS) return LV_ERR_UNSUPPORTED;
        tile = 128;
    }
    if (M < 0 || N < 0 || K < 0) return LV_ERR_SHAPE;
    if (M == 0 || N == 0) return LV_OK;
    if (!A || !B || !C) return LV_ERR_ARG;
    if ((add1 && mod1 <= 0) || (add2 && mod2 <= 0)) return LV_ERR_ARG;
    if (lda < (transA ? M : K) || ldb < K || ldc < (C2 && nsplit > 0 ? nsplit : N)) return LV_ERR_SHAPE;
    if (ldb % 8 != 0 || (((uintptr_t)B) & 15) != 0) return LV_ERR_ALIGN;
    if (lda % 8 != 0 || (((uintptr_t)A) & 15) != 0) return LV_ERR_ALIGN;
    GemmQ p{};
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.alpha = alpha; p.accumulate = accumulate;
    p.add1 = add1; p.ld1 = ld1; p.mod1 = mod1 > 0 ? mod1 : 1;
    p.add2 = add2; p.ld2 = ld2; p.mod2 = mod2 > 0 ? mod2 : 1;
    p.ws = ws;
    p.keep = nullptr; p.kscale = 1.f; p.keepT = 1; p.Bsz = Bsz > 0 ? Bsz : 1;
    p.sq = sq; p.sq_only = sq ? sq_only : 0;
    p.C2 = C2; p.ldc2 = ldc2; p.nsplit = C2 ? nsplit : 0;
    if (p.nsplit > 0) {              // two destinations: written by the split-K reduction stage only (see lv_gemm_b16_dual)
        if (p.nsplit >= N || ldc < p.nsplit || ldc2 < N - p.nsplit || add1 || add2 || accumulate || keep || sq || f16) return LV_ERR_ARG;
        if (t256_wanted(tile, M, N, K) || !ws) return LV_ERR_UNSUPPORTED;
    }
    if (sq && (!t256_wanted(tile, M, N, K) || accumulate || add1 || add2 || keep)) return LV_ERR_ARG;     // see lv_gemm_b16_sumsq_parts
    // a keep-mask rides in the reduction kernel when every output element passes through one; else a pass of its own follows
    bool keep_pending = keep != nullptr;
    const int nk = lv_cdiv(K > 0 ? K : 1, BK);
    if (t256_wanted(tile, M, N, K)) {
        p.tilesM = lv_cdiv(M, BT2); p.tilesN = lv_cdiv(N, BT2);
        p.splits = 1; p.kt_per_split = nk;
        const Tail256 q = t256_plan((long)p.tilesM * p.tilesN, nk, ws ? ws_floats : 0);
        const long tail = (long)p.tilesM * p.tilesN - q.full;
        if (keep && q.full == 0 && q.tail_s > 1) { p.keep = keep; p.kscale = kscale; p.keepT = M / p.Bsz; keep_pending = false; }
        dim3 grid((unsigned)(q.full + tail * q.tail_s)), block(512);
        const int sched = t256_sched(tile, q.tail_s > 1 && q.full == 0 ? q.kt_per_piece : nk);
        if (sched == 2) {
            if (transA) LV_LAUNCH((lv_gemm_b16_t256q_kernel<false, true>), grid, block, 0, stream, p, q);
            else LV_LAUNCH((lv_gemm_b16_t256q_kernel<false, false>), grid, block, 0, stream, p, q);
        } else if (sched == 1) {
            if (transA) LV_LAUNCH((lv_gemm_b16_t256_kernel<false, true, true>), grid, block, 0, stream, p, q);
            else LV_LAUNCH((lv_gemm_b16_t256_kernel<false, false, true>), grid, block, 0, stream, p, q);
        } else if (transA) LV_LAUNCH((lv_gemm_b16_t256_kernel<false, true>), grid, block, 0, stream, p, q);
        else LV_LAUNCH((lv_gemm_b16_t256_kernel<false, false>), grid, block, 0, stream, p, q);
        if (q.tail_s > 1) LV_LAUNCH(tail_reduce_t256_kernel, dim3((unsigned)(tail * TR_WGS)), dim3(256), 0, stream, p, q);
        LV_CHECK_LAUNCH();
        if (keep_pending) return lv_keep_scale_f32(C, keep, kscale, M / p.Bsz, p.Bsz, N, stream);
        return LV_OK;
    }
    p.tilesM = lv_cdiv(M, BT); p.tilesN = lv_cdiv(N, BT);
    const long tiles = (long)p.tilesM * p.tilesN;
    // split-K (deterministic: partial slabs + ordered reduce) when the tile count alone cannot fill 256 CUs x 2
    int splits = 1;
    if (ws && tiles < 512 && nk >= 8) {
        long sp = lv_cdiv(LV_B16_SPLIT_TARGET, tiles);
        if (sp > nk / 4) sp = nk / 4;
        if (sp > 64) sp = 64;
        const long cap = ws_floats / ((long)M * N);
        if (sp > cap) sp = cap;
        if (sp > 1) splits = (int)sp;
    }
    if (p.nsplit > 0 && splits < 2) {
        if (nk < 2 || ws_floats < 2L * M * N) return LV_ERR_UNSUPPORTED;
        splits = 2;
    }
    if (p.nsplit > 0) {
        // ... and into pieces of <= 32 K tiles: those take the single-buffer instantiation (3 workgroups per CU).  Measured on the
        // two LSTM weight gradients as one product (M = 4096, N = 1536, K = 6400: 384 tiles): 2 pieces of 50 K tiles = 768
        // workgroups on 512 slots (one and a half rounds) lost 30 us against the two separate products; 4 pieces of 25 = 1536
        // workgroups on 768 slots, two whole rounds (profiles/r05j_gemm_dual.txt)
        while (lv_cdiv(nk, splits) > 32 && splits < 64 && ws_floats >= (long)(splits + 1) * M * N) ++splits;
    }
    p.kt_per_split = lv_cdiv(nk, splits);
    splits = lv_cdiv(nk, p.kt_per_split);
    p.splits = splits;
    if (keep && splits > 1) { p.keep = keep; p.kscale = kscale; p.keepT = M / p.Bsz; keep_pending = false; }
    dim3 grid((unsigned)tiles, (unsigned)splits), block(256);
    if (transA && LV_B16_GLDS == 1 && p.kt_per_split > 32) LV_LAUNCH((lv_gemm_b16_nt_glds_kernel<false, false, true>), grid, block, 0, stream, p);
    else if (transA && LV_B16_GLDS) LV_LAUNCH((lv_gemm_b16_nt_glds_kernel<true, false, true>), grid, block, 0, stream, p);
    else if (transA) LV_LAUNCH((lv_gemm_b16_kernel<false>), grid, block, 0, stream, p);
    else if (f16 && p.kt_per_split > 32) LV_LAUNCH((lv_gemm_b16_nt_glds_kernel<false, false, false, true>), grid, block, 0, stream, p);
    else if (f16) LV_LAUNCH((lv_gemm_b16_nt_glds_kernel<true, false, false, true>), grid, block, 0, stream, p);
    else if (LV_B16_GLDS == 1 && p.kt_per_split > 32) LV_LAUNCH(lv_gemm_b16_nt_glds_kernel<false>, grid, block, 0, stream, p);
    else if (LV_B16_GLDS) LV_LAUNCH(lv_gemm_b16_nt_glds_kernel<true>, grid, block, 0, stream, p);
    else LV_LAUNCH((lv_gemm_b16_kernel<true>), grid, block, 0, stream, p);
    if (splits > 1) {
        const bool v4 = !add1 && !add2 && !accumulate && N % 4 == 0 && ldc % 4 == 0 && (((uintptr_t)C | (uintptr_t)ws) & 15) == 0 &&
                        (((uintptr_t)p.keep) & 3) == 0 &&
                        (p.nsplit == 0 || (p.nsplit % 4 == 0 && ldc2 % 4 == 0 && (((uintptr_t)C2) & 15) == 0));
        if (v4) LV_LAUNCH(splitk_reduce_b16_v4_kernel, dim3((unsigned)lv_cdiv((long)M * N / 4, 256)), dim3(256), 0, stream, p);
        else LV_LAUNCH(splitk_reduce_b16_kernel, dim3((unsigned)lv_cdiv((long)M * N, 256)), dim3(256), 0, stream, p);
    }
    LV_CHECK_LAUNCH();
    if (keep_pending) return lv_keep_scale_f32(C, keep, kscale, M / p.Bsz, p.Bsz, N, stream);
    return LV_OK;
}

// ---- grouped stream-K launch (lv_gemm_b16_t256g_kernel): up to two independent products on one workgroup per CU ----------------
// How the 32 workgroup rows are divided: the split that minimises the longer of the two unit runs per workgroup, with a 10 % handicap
// on a first product whose runs are not whole aligned pieces of its tiles (its workgroups then share less in L2).
static int sk_rows0(long U0, int nk0, long U1) {
    if (U1 <= 0) return 32;
    int best = 0;
    double bestc = 0.;
    for (int r0 = 1; r0 < 32; ++r0) {
        const long n0 = 8L * r0, n1 = 8L * (32 - r0);
        const long L0 = lv_cdiv(U0, n0), L1 = lv_cdiv(U1, n1);
        double c = (double)(L0 > L1 ? L0 : L1);
        if (U0 % n0 != 0 || nk0 % L0 != 0) c *= 1.1;
        if (!best || c < bestc) { best = r0; bestc = c; }
    }
    return best;
}
static bool sk_fits32(int transA, int M, int N, int K, long lda, long ldb) {
    const long ea = (long)(transA ? K : M) * lda + 64, eb = (long)N * ldb + 64;      // element offsets the K loop forms from A / B
    return ea < (1L << 32) && eb < (1L << 32) && (transA ? K : M) < (1 << 30);
}
static long sk_ws_floats() { return 2L * 256 * BT2 * BT2; }

// Possible: operands as lv_gemm_b16 asks, the second product (if any) in NT form, the workspace holds the slabs, the tiles fit the
// arrival counters.  lv_gemm_b16_pair_supported adds "worth it": two products, big enough that a workgroup walks >= 12 K tiles.
static bool sk_possible(int transA1, int M0, int N0, int K0, int M1, int N1, int K1, long ws_floats) {
    if (!LV_B16_GLDS) return false;
    if (M0 <= 0 || N0 <= 0 || K0 <= 0 || M1 < 0 || N1 < 0 || K1 < 0) return false;
    const bool two = M1 > 0 && N1 > 0 && K1 > 0;
    if (transA1 != 0 && two) return false;                                // instantiated: (TN | NT) + NT
    const long t0 = (long)lv_cdiv(M0, BT2) * lv_cdiv(N0, BT2), t1 = two ? (long)lv_cdiv(M1, BT2) * lv_cdiv(N1, BT2) : 0;
    return t0 + t1 <= SK_TILES && ws_floats >= sk_ws_floats();
}
extern "C" int lv_gemm_b16_pair_supported(int transA0, int M0, int N0, int K0, int transA1, int M1, int N1, int K1, long ws_floats) {
    (void)transA0;
    if (!sk_possible(transA1, M0, N0, K0, M1, N1, K1, ws_floats) || M1 <= 0 || N1 <= 0 || K1 <= 0) return 0;
    // measured (profiles/r06o_gemm_pair_probe.txt; H = 1024, ni = 512; pair / separate launches): T*B = 6400 -> 110 / 156 us, 3200 -> 73 / 91,
    // 25600 -> 385 / 506, 1600 -> 58 / 65 (13 K tiles per workgroup); below that the hand-off (~20 us) is no longer paid for
    const long U = (long)lv_cdiv(M0, BT2) * lv_cdiv(N0, BT2) * lv_cdiv(K0, BK) + (long)lv_cdiv(M1, BT2) * lv_cdiv(N1, BT2) * lv_cdiv(K1, BK);
    return U >= 256L * 12;
}

// C0 (| C0b) = op(A0) . B0^T and C1 = A1 . B1^T in ONE launch (operands, layouts and alignment as lv_gemm_b16; plain outputs, no
// addends; nsplit0 > 0: columns >= nsplit0 of the first product go to C0b [M0][ldc0b] as in lv_gemm_b16_dual).  M1 == 0: one product.
extern "C" int lv_gemm_b16_pair(int transA0, int M0, int N0, int K0, const uint16_t* A0, long lda0, const uint16_t* B0, long ldb0,
                                float* C0, long ldc0, int nsplit0, float* C0b, long ldc0b,
                                int transA1, int M1, int N1, int K1, const uint16_t* A1, long lda1, const uint16_t* B1, long ldb1,
                                float* C1, long ldc1, float* ws, long ws_floats, void* stream) {
    if (!sk_possible(transA1, M0, N0, K0, M1, N1, K1, ws_floats)) return LV_ERR_UNSUPPORTED;
    const bool two = M1 > 0 && N1 > 0 && K1 > 0;
    if (!A0 || !B0 || !C0 || !ws || (two && (!A1 || !B1 || !C1))) return LV_ERR_ARG;
    if (nsplit0 < 0 || (nsplit0 > 0 && (!C0b || nsplit0 >= N0 || ldc0b < N0 - nsplit0))) return LV_ERR_ARG;
    if (lda0 < (transA0 ? M0 : K0) || ldb0 < K0 || ldc0 < (nsplit0 > 0 ? nsplit0 : N0)) return LV_ERR_SHAPE;
    if (two && (lda1 < K1 || ldb1 < K1 || ldc1 < N1)) return LV_ERR_SHAPE;
    if (lda0 % 8 != 0 || ldb0 % 8 != 0 || ((((uintptr_t)A0) | ((uintptr_t)B0)) & 15) != 0) return LV_ERR_ALIGN;
    if (two && (lda1 % 8 != 0 || ldb1 % 8 != 0 || ((((uintptr_t)A1) | ((uintptr_t)B1)) & 15) != 0)) return LV_ERR_ALIGN;
    if ((((uintptr_t)ws) & 15) != 0) return LV_ERR_ALIGN;
    if (!sk_fits32(transA0, M0, N0, K0, lda0, ldb0) || (two && !sk_fits32(0, M1, N1, K1, lda1, ldb1))) return LV_ERR_UNSUPPORTED;
    static unsigned next_slot = 0;
    const unsigned slot = __atomic_fetch_add(&next_slot, 1u, __ATOMIC_RELAXED) % SK_SLOTS;
    GemmG g{};
    auto fill = [&](SkProb& P, int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb, float* C, long ldc) {
        P.q.A = A; P.q.B = B; P.q.C = C; P.q.M = M; P.q.N = N; P.q.K = K;
        P.q.lda = lda; P.q.ldb = ldb; P.q.ldc = ldc; P.q.alpha = 1.f;
        P.q.tilesM = lv_cdiv(M, BT2); P.q.tilesN = lv_cdiv(N, BT2);
        P.nk = lv_cdiv(K, BK);
        P.units = (long)P.q.tilesM * P.q.tilesN * P.nk;
    };
    fill(g.pr[0], M0, N0, K0, A0, lda0, B0, ldb0, C0, ldc0);
    g.pr[0].q.C2 = nsplit0 > 0 ? C0b : nullptr; g.pr[0].q.ldc2 = ldc0b; g.pr[0].q.nsplit = nsplit0 > 0 ? nsplit0 : 0;
    if (two) fill(g.pr[1], M1, N1, K1, A1, lda1, B1, ldb1, C1, ldc1);
    g.pr[0].rows = (LV_SK_ROWS0 > 0 && two) ? LV_SK_ROWS0 : sk_rows0(g.pr[0].units, g.pr[0].nk, two ? g.pr[1].units : 0);
    g.pr[1].rows = 32 - g.pr[0].rows;
    for (int j = 0; j < 2; ++j) {
        SkProb& P = g.pr[j];
        const long n = 8L * P.rows;
        if (LV_SK_SKEW <= 0 || n == 0 || P.units % n != 0) continue;
        const long L = P.units / n;                        // units per workgroup
        if (L < 8 || P.nk % L != 0 || P.nk / L < 2) continue;
        P.pieces = (int)(P.nk / L); P.plen = (int)L;
        P.skew = LV_SK_SKEW < L / 8 ? LV_SK_SKEW : (int)(L / 8);
    }
    g.pr[0].slabs = ws;
    g.pr[1].slabs = ws + 2L * 8 * g.pr[0].rows * (BT2 * BT2);
    g.pr[0].cnt0 = (int)(slot * SK_TILES);
    g.pr[1].cnt0 = g.pr[0].cnt0 + g.pr[0].q.tilesM * g.pr[0].q.tilesN;
    if (transA0) LV_LAUNCH((lv_gemm_b16_t256g_kernel<true, false>), dim3(256), dim3(512), 0, stream, g);
    else LV_LAUNCH((lv_gemm_b16_t256g_kernel<false, false>), dim3(256), dim3(512), 0, stream, g);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// *pending (device int) = the number of arrival counters of the grouped launches that are not zero: 0 whenever no such launch is in
// flight on the device (every launch returns its counters to zero) -- tests and soaks check exactly that.
extern "C" int lv_gemm_b16_pair_pending(int* pending, void* stream) {
    if (!pending) return LV_ERR_ARG;
    LV_LAUNCH(sk_pending_kernel, dim3(1), dim3(256), 0, stream, pending);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_gemm_b16_tile(int tile, int transA, int M, int N, int K, float alpha,
                                const uint16_t* A, long lda, const uint16_t* B, long ldb,
                                float* C, long ldc, int accumulate,
                                const float* add1, long ld1, int mod1,
                                const float* add2, long ld2, int mod2,
                                float* ws, long ws_floats, void* stream) {
    return gemm_b16_launch(tile, transA, M, N, K, alpha, A, lda, B, ldb, C, ldc, accumulate, add1, ld1, mod1, add2, ld2, mod2, ws, ws_floats,
                           stream, nullptr, 1.f, 1);
}

// ONE product, TWO destinations: C1 [M][nsplit] (ldc1) = columns [0, nsplit) of op(A) . B^T, C2 [M][N - nsplit] (ldc2) the rest.  For
// the two weight gradients of an LSTM layer that share their A operand, dW_ih = dG^T X and dW_hh = dG^T h_prev (dec_lstm.py:104 /
// enc_lstm.py:55 backward): with X^T and h_prev^T stored as one [ni + H][T*B] image they are ONE product of N = ni + H columns --
// one launch that fills the chip better than a 27- and a 54-GFLOP one, and one reduction stage instead of two.  The split of the
// columns happens in the split-K reduction stage, so the product must take one (forced to two pieces if the shape alone would not
// split); LV_ERR_UNSUPPORTED where it would take the 256 x 256 tile (lv_gemm_b16_dual_supported says so beforehand).
extern "C" int lv_gemm_b16_dual_supported(int M, int N, int K, long ws_floats) {
    if (M <= 0 || N <= 0 || K <= 0 || t256_wanted(0, M, N, K)) return 0;
    return lv_cdiv(K, BK) >= 2 && ws_floats >= 2L * M * N;
}
extern "C" int lv_gemm_b16_dual(int transA, int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                                float* C1, long ldc1, int nsplit, float* C2, long ldc2, float* ws, long ws_floats, void* stream) {
    if (!C2 || nsplit <= 0) return LV_ERR_ARG;
    return gemm_b16_launch(0, transA, M, N, K, 1.f, A, lda, B, ldb, C1, ldc1, 0, nullptr, 0, 1, nullptr, 0, 1, ws, ws_floats, stream,
                           nullptr, 1.f, 1, nullptr, 0, 0, C2, ldc2, nsplit);
}

// C [M][N] (ldc = N) = (A . B^T) * (keep ? kscale : 0): the product of lv_gemm_b16 (transA = 0) with the backward of nn.Dropout
// (dec_lstm.py:106 on the LSTM output: dO = dlogits . W_pred, masked) folded into the reduction stage of the product -- rows are
// time-major (r = t * Bsz + b), keep is the reference-layout mask [Bsz][M / Bsz][N] (uint8).  Where the product has no reduction
// stage (it fits whole tiles) the mask is applied by a pass of its own (lv_keep_scale_f32): same result either way.
extern "C" int lv_gemm_b16_keep(int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb, float* C,
                                const uint8_t* keep, float kscale, int Bsz, float* ws, long ws_floats, void* stream) {
    if (!keep || Bsz <= 0 || M % Bsz != 0) return LV_ERR_ARG;
    return gemm_b16_launch(0, 0, M, N, K, 1.f, A, lda, B, ldb, C, N, 0, nullptr, 0, 1, nullptr, 0, 1, ws, ws_floats, stream, keep, kscale, Bsz);
}

// C = op(A) . B^T as lv_gemm_b16 computes it, and the product's sum of squares in the same pass: sq[0 .. parts) receives one
// partial per wave of the kernels that hold C's final values (their sum in any fixed order is |C|^2; parts =
// lv_gemm_b16_sumsq_parts(M, N, K, ws_floats), 0 = this shape does not take the 256 x 256 tile and the entry refuses it).
// sq_only != 0: C is NOT written (a gradient that is only needed for the norm of clip_grad_norm_: the aggressive inner loop,
// text.py:383-387, clips over all parameters and steps the encoder alone).
extern "C" int lv_gemm_b16_sumsq_parts(int M, int N, int K, long ws_floats) {
    if (M <= 0 || N <= 0 || K <= 0 || !t256_wanted(0, M, N, K)) return 0;
    const long tiles = (long)lv_cdiv(M, BT2) * lv_cdiv(N, BT2);
    const Tail256 q = t256_plan(tiles, lv_cdiv(K, BK), ws_floats);
    return (int)(tiles * 8 + (q.tail_s > 1 ? (long)q.tail * TR_WGS * 4 : 0));
}

extern "C" int lv_gemm_b16_sumsq(int transA, int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                                 float* C, long ldc, float* ws, long ws_floats, float* sq, int sq_only, void* stream) {
    if (!sq || (!ws && ws_floats > 0)) return LV_ERR_ARG;      // (the partial count follows the tile plan, and the plan the workspace size)
    return gemm_b16_launch(0, transA, M, N, K, 1.f, A, lda, B, ldb, C, ldc, 0, nullptr, 0, 1, nullptr, 0, 1, ws, ws_floats, stream,
                           nullptr, 1.f, 1, sq, sq_only);
}

// The same with the tile edge chosen by shape (the product's entry).
extern "C" int lv_gemm_b16(int transA, int M, int N, int K, float alpha,
                           const uint16_t* A, long lda, const uint16_t* B, long ldb,
                           float* C, long ldc, int accumulate,
                           const float* add1, long ld1, int mod1,
                           const float* add2, long ld2, int mod2,
                           float* ws, long ws_floats, void* stream) {
    return lv_gemm_b16_tile(0, transA, M, N, K, alpha, A, lda, B, ldb, C, ldc, accumulate, add1, ld1, mod1, add2, ld2, mod2, ws,
                            ws_floats, stream);
}

// C = A . B^T (+ addends / accumulate) as lv_gemm_b16 with transA = 0, on operands that are IEEE BINARY16 images (lv_cvt_h16_f32):
// v_mfma_f32_32x32x16_f16, f32 accumulation -- the forward input projection of the encoder (enc_lstm.py:50-55), whose operands'
// rounding is a third of what moves the KL in the bf16 configuration (profiles/r05a_kl_ablation.txt)
extern "C" int lv_gemm_h16(int M, int N, int K, float alpha, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                           float* C, long ldc, int accumulate, const float* add1, long ld1, int mod1,
                           const float* add2, long ld2, int mod2, float* ws, long ws_floats, void* stream) {
    return gemm_b16_launch(128, 0, M, N, K, alpha, A, lda, B, ldb, C, ldc, accumulate, add1, ld1, mod1, add2, ld2, mod2, ws, ws_floats,
                           stream, nullptr, 1.f, 1, nullptr, 0, 1);
}

// src f32 [R][C] (lds) -> dst bf16 [R][C] (ldd) and/or dstT bf16 [C][R] (ldt); either destination may be null.
extern "C" int lv_cvt_bf16_f32(const float* src, long lds, int R, int C, uint16_t* dst, long ldd, uint16_t* dstT, long ldt,
                               void* stream) {
    if (!src || (!dst && !dstT)) return LV_ERR_ARG;
    if (R < 0 || C < 0 || lds < C || (dst && ldd < C) || (dstT && ldt < R)) return LV_ERR_SHAPE;
    if (R == 0 || C == 0) return LV_OK;
    cvt_launch(stream, src, lds, R, C,
              dst, ldd, dstT, ldt, 0, (const uint8_t*)nullptr, 1.f, 1, (const int64_t*)nullptr, 0L, 0, 0);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// The same conversion with nn.Dropout folded in (dec_lstm.py:106: dropout_out on the LSTM output): src = h [T*Bsz][C] time-major
// (row t*Bsz + b), keep = the reference-layout mask [Bsz][T][C] (uint8), images of h * (keep ? kscale : 0).
extern "C" int lv_cvt_bf16_keep_f32(const float* src, long lds, int T, int Bsz, int C, const uint8_t* keep, float kscale,
                                    uint16_t* dst, long ldd, uint16_t* dstT, long ldt, void* stream) {
    if (!src || !keep || (!dst && !dstT)) return LV_ERR_ARG;
    const long R = (long)T * Bsz;
    if (T < 0 || Bsz <= 0 || C < 0 || lds < C || (dst && ldd < C) || (dstT && ldt < R)) return LV_ERR_SHAPE;
    if (R == 0 || C == 0) return LV_OK;
    cvt_launch(stream, src, lds, (int)R, C,
              dst, ldd, dstT, ldt, 0, keep, kscale, Bsz, (const int64_t*)nullptr, 0L, 0, 0);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// nn.Embedding lookup (+ nn.Dropout) straight into the bf16 operand images of the input projection (enc_lstm.py:50-52,
// dec_lstm.py:86-93): row r = t*Bsz + b of dst [T*Bsz][C] / column r of dstT [C][T*Bsz] = emb[clamp(ids[b*ids_stride + t])] *
// (keep ? (keep[b][t][c] ? kscale : 0) : 1), RNE -- bit-identical to lv_embed_gather_f32 followed by lv_cvt_bf16_f32, without
// the f32 matrix in between.  keep may be NULL (no dropout).
extern "C" int lv_embed_gather_b16(const float* emb, const int64_t* ids, long ids_stride, const uint8_t* keep, float kscale,
                                   int T, int Bsz, int C, int V, uint16_t* dst, long ldd, uint16_t* dstT, long ldt, void* stream) {
    if (!emb || !ids || (!dst && !dstT)) return LV_ERR_ARG;
    const long R = (long)T * Bsz;
    if (T < 0 || Bsz <= 0 || C < 0 || V <= 0 || (dst && ldd < C) || (dstT && ldt < R)) return LV_ERR_SHAPE;
    if (R == 0 || C == 0) return LV_OK;
    cvt_launch(stream, emb, (long)C, (int)R, C,
              dst, ldd, dstT, ldt, 0, keep, kscale, Bsz, ids, ids_stride, V, 0);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// LSTM gate weight W [4H][C] (rows g*H + u, gate order i|f|g|o): dst = bf16 image with the rows in unit-major order
// (u*4 + g), dstT = bf16 image of W^T [C][4H] in the standard order; either may be null.
extern "C" int lv_cvt_bf16_gates_f32(const float* src, long lds, int H, int C, uint16_t* dst, long ldd, uint16_t* dstT, long ldt,
                                     void* stream) {
    if (!src || (!dst && !dstT)) return LV_ERR_ARG;
    if (H <= 0 || C < 0 || lds < C || (dst && ldd < C) || (dstT && ldt < 4 * H)) return LV_ERR_SHAPE;
    if (C == 0) return LV_OK;
    cvt_launch(stream, src, lds, 4 * H, C,
              dst, ldd, dstT, ldt, H, (const uint8_t*)nullptr, 1.f, 1, (const int64_t*)nullptr, 0L, 0, 0);
    LV_CHECK_LAUNCH();
    return LV_OK;
}


// The LOW half of a split-bf16 operand: image of bf16(x - bf16(x)) for the same sources and layouts as the conversions above
// (x = hi + lo up to 2^-17 |x|; a product of two split operands, hi.hi' + hi.lo' + lo.hi' on the bf16 pipe with f32 accumulation,
// is f32-like: the dropped lo.lo' is 2^-16 relative).  gate_H > 0: src = an LSTM gate weight [4 gate_H][C], dst rows unit-major as
// lv_cvt_bf16_gates_f32; ids != NULL: src = an embedding table [V][C], row r = t*Bsz + b of the image is row ids[b*ids_stride + t]
// (R = T*Bsz) as lv_embed_gather_b16; otherwise a plain [R][C] matrix as lv_cvt_bf16_f32.  Used where weight rounding -- a
// perturbation that acts coherently over all timesteps -- would otherwise own the error of the encoder's last state
// (profiles/r05a_kl_ablation.txt).
extern "C" int lv_cvt_bf16_lo_f32(const float* src, long lds, int R, int C, int gate_H, const int64_t* ids, long ids_stride, int Bsz,
                                  int V, uint16_t* dst, long ldd, uint16_t* dstT, long ldt, void* stream) {
    if (!src || (!dst && !dstT)) return LV_ERR_ARG;
    if (R < 0 || C < 0 || lds < C || (dst && ldd < C) || (dstT && ldt < R)) return LV_ERR_SHAPE;
    if (gate_H > 0 && (R != 4 * gate_H || ids)) return LV_ERR_ARG;
    if (ids && (Bsz <= 0 || V <= 0 || R % Bsz != 0)) return LV_ERR_SHAPE;
    if (R == 0 || C == 0) return LV_OK;
    cvt_launch(stream, src, lds, R, C,
              dst, ldd, dstT, ldt, gate_H > 0 ? gate_H : 0, (const uint8_t*)nullptr, 1.f, ids ? Bsz : 1, ids, ids_stride, ids ? V : 0, 1);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// Operand images for a forward product on the BINARY16 matrix pipe (lv_gemm_h16): dst = IEEE half (RNE) in the layouts of the bf16
// conversions (plain / gate_H > 0: unit-major LSTM gate rows / ids != NULL: gathered embedding rows), dstT = the transposed BF16
// image (what the gradient products of the same operand read: gradients need bf16's exponent range, forward operands of an LSTM --
// weights U(-0.01, 0.01)-ish, embeddings, h in (-1, 1) -- do not, and binary16's 11-bit significand rounds them 8 x finer).  Values
// beyond +-65504 saturate (a weight of that size has no business in an LSTM; the bf16 image of the same operand keeps its range).
extern "C" int lv_cvt_h16_f32(const float* src, long lds, int R, int C, int gate_H, const int64_t* ids, long ids_stride, int Bsz,
                              int V, uint16_t* dst, long ldd, uint16_t* dstT, long ldt, void* stream) {
    if (!src || (!dst && !dstT)) return LV_ERR_ARG;
    if (R < 0 || C < 0 || lds < C || (dst && ldd < C) || (dstT && ldt < R)) return LV_ERR_SHAPE;
    if (gate_H > 0 && (R != 4 * gate_H || ids)) return LV_ERR_ARG;
    if (ids && (Bsz <= 0 || V <= 0 || R % Bsz != 0)) return LV_ERR_SHAPE;
    if (R == 0 || C == 0) return LV_OK;
    cvt_launch(stream, src, lds, R, C,
              dst, ldd, dstT, ldt, gate_H > 0 ? gate_H : 0, (const uint8_t*)nullptr, 1.f, ids ? Bsz : 1, ids, ids_stride, ids ? V : 0, 2);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// LSTMDecoder's vocabulary projection fused with the statistics of nn.CrossEntropyLoss (modules/decoders/dec_lstm.py:117,
// 140-146): logits = A . B^T (bf16 operands as lv_gemm_b16, transA = 0) are written ONCE, as binary16 [M][ldl16] (RNE of the
// f32 accumulators; the f32 logits image of the unfused route -- 510 MB at the Yahoo shape -- is never produced), together
// with per-row partial statistics over 64-column pieces, part [M][nparts] (max, sum exp(x - max)), nparts =
// lv_gemm_b16_nll_parts(N), and tgt_logit [M], the logit of row r's target token ids[(r % Bsz) * ids_stride + r / Bsz +
// tgt_off].  lv_softmax_nll_merge_f32 turns them into lse / nll; lv_softmax_nll_bwd_h16 reads the binary16 image back.
// 64-column pieces of a row, counted in whole 256-column tiles (both tile sizes fill all of them)
extern "C" int lv_gemm_b16_nll_parts(int N) { return 4 * lv_cdiv(N, BT2); }

extern "C" int lv_gemm_b16_nll_tile(int tile, int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                                    uint16_t* logits16, long ldl16, const int64_t* ids, long ids_stride, int tgt_off, int Bsz,
                                    float* part, float* tgt_logit, void* stream) {
    if (tile != 0 && tile != 128 && (tile < 256 || tile > 258)) return LV_ERR_ARG;
    if (M < 0 || N <= 0 || K <= 0 || Bsz <= 0) return LV_ERR_SHAPE;
    if (M == 0) return LV_OK;
    if (!A || !B || !logits16 || !ids || !part || !tgt_logit) return LV_ERR_ARG;
    if (lda < K || ldb < K || ldl16 < N) return LV_ERR_SHAPE;
    if (lda % 8 != 0 || ldb % 8 != 0 || ldl16 % 8 != 0 || (((uintptr_t)A) & 15) != 0 || (((uintptr_t)B) & 15) != 0 ||
        (((uintptr_t)logits16) & 15) != 0 || (((uintptr_t)part) & 7) != 0)
        return LV_ERR_ALIGN;
    GemmQ p{};
    p.A = A; p.B = B; p.C = nullptr; p.M = M; p.N = N; p.K = K;
    p.lda = lda; p.ldb = ldb; p.ldc = 0; p.alpha = 1.f; p.accumulate = 0;
    p.add1 = nullptr; p.add2 = nullptr; p.mod1 = p.mod2 = 1; p.ws = nullptr;
    p.splits = 1; p.kt_per_split = lv_cdiv(K, BK);
    p.C16 = logits16; p.ldc16 = ldl16; p.ids = ids; p.ids_stride = ids_stride; p.tgt_off = tgt_off; p.Bsz = Bsz;
    p.part = reinterpret_cast<float2*>(part); p.nparts = lv_gemm_b16_nll_parts(N); p.tgt = tgt_logit;
    if (t256_wanted(tile, M, N, K)) {
        p.tilesM = lv_cdiv(M, BT2); p.tilesN = lv_cdiv(N, BT2);
        Tail256 q;
        const long tiles = (long)p.tilesM * p.tilesN;
        q.full = (int)(tiles / 256 * 256); q.tail = (int)(tiles - q.full); q.tail_s = 1; q.kt_per_piece = p.kt_per_split;
        const int sched = t256_sched(tile, p.kt_per_split);
        // (a persistent grid of 256 workgroups walking the 1975 tiles -- epilogue stores draining under the next tile's first loads --
        // measured the same for this launch, 292 vs 293 us, and cost the plain-store variants of the kernel ~1 KB of scratch per lane)
        if (sched == 2) LV_LAUNCH((lv_gemm_b16_t256q_kernel<true, false>), dim3((unsigned)tiles), dim3(512), 0, stream, p, q);
        else if (sched == 1) LV_LAUNCH((lv_gemm_b16_t256_kernel<true, false, true>), dim3((unsigned)tiles), dim3(512), 0, stream, p, q);
        else LV_LAUNCH((lv_gemm_b16_t256_kernel<true, false>), dim3((unsigned)tiles), dim3(512), 0, stream, p, q);
        LV_CHECK_LAUNCH();
        return LV_OK;
    }
    p.tilesM = lv_cdiv(M, BT); p.tilesN = lv_cdiv(N, BT);
    dim3 grid((unsigned)((long)p.tilesM * p.tilesN), 1), block(256);
    LV_LAUNCH((lv_gemm_b16_nt_glds_kernel<true, true>), grid, block, 0, stream, p);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_gemm_b16_nll(int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                               uint16_t* logits16, long ldl16, const int64_t* ids, long ids_stride, int tgt_off, int Bsz,
                               float* part, float* tgt_logit, void* stream) {
    return lv_gemm_b16_nll_tile(0, M, N, K, A, lda, B, ldb, logits16, ldl16, ids, ids_stride, tgt_off, Bsz, part, tgt_logit, stream);
}
