// lv_lstm.hip -- the LSTM time recurrence (forward and BPTT) of the encoder/decoder, exact f32.
//
// Replaces nn.LSTM forward/backward on the hot path (modules/encoders/enc_lstm.py:60,
// modules/decoders/dec_lstm.py:104; autograd backward from text.py:384).  Math restated in
// oracle/text_vae_oracle.py (PyTorch gate order i|f|g|o).
//
// HBM layout: everything time-major so one timestep is one contiguous [B][*] slab:
//   gx    [T][B][4H]  x_t W_ih^T + b_ih + b_hh (+ z W_z^T for the decoder)  -- produced by lv_gemm_*
//   hs,cs [T+1][B][H] hs[0]/cs[0] = initial state, step t writes index t+1
//   gates [T][B][H][4] activated (i,f,g,o) of one unit as one 16-byte record, saved for BPTT (private to these
//                      kernels);  dG [T][B][4H] grads wrt pre-activations (gate-major, as the GEMMs consume them)
//
// One launch per timestep: the kernel boundary is the grid-wide h_t hand-off (cheaper on gfx950 than an in-kernel
// grid barrier -- MI355X_MICROARCH.md price list).  The step kernels are latency-bound: one workgroup per CU, 4
// waves, each wave streams its slice of both operands of a 32 x 16 x K product straight into
// v_mfma_f32_16x16x4_f32.  Measured on MI355X (profiles/r01_microbench_*): a wave64 float4 load that touches 16 rows
// x 64 B costs twice one that touches whole 128 B lines, whatever the row pitch.  So both recurrent operands are kept
// in an MFMA-fragment-major packed layout in which every wave-level load is 1 KB of consecutive bytes:
//   activations  Ap[k/4][mb][16 rows][4 k]      (k-quad major; mb = 16-row batch block)
//   weights      Wp[nb][k/4][16 cols][4 k]      (nb = 16-column block owned by one workgroup)
// Lane (i = l&15, kq = l>>4) loads the float4 (k = 4*(4*it+kq) .. +3) of row/column i and feeds element j of it to
// MFMA j -- the same permutation of K on both operands.  W_hh is re-packed once per call (16.8 MB, ~1% of a step);
// h_t is written by the step epilogue both in the standard layout (for the GEMMs that consume hs) and packed (for
// step t+1): a workgroup owns 4 hidden units = exactly one k-quad, so its packed store is 512 contiguous bytes.
// K is zero-padded to a multiple of 16 and rows/columns beyond B / H are zero, so the inner loop has no predicates.
//
// Forward step: grid.x = ceil(H/4) workgroups, each owns 4 hidden units = 16 gate columns (i,f,g,o x 4 units) for
//   all batch rows, so sigma/tanh, the c/h update and the decoder's output dropout fuse into the epilogue.
// Backward step t = two launches: an elementwise kernel (gate grads, dc chain, dG[t], running sum of dG) and the
//   recurrent matmul dh_{t-1} = dG[t] . W_hh as split-K partial slabs (same core, weights packed transposed); the
//   next step's elementwise kernel sums the slabs in a fixed order (deterministic, no atomics).
//
// lv_lstm_persist.hip holds the one-launch persistent forms of the bf16 recurrences (H = 1024 on a full MI355X); the step
// kernels here are the general path (any H / B, f32 parity path, partitioned devices) and the reference they are tested against.
#include "lv_device.h"

namespace {

static inline int h_round_up(int x, int m) { return (x + m - 1) / m * m; }

template <int MB> struct RecNit { static constexpr int value = MB <= 2 ? 16 : (MB == 4 ? 8 : 4); };

// acc[mb] += sum over iterations [it0, it1) of (16 k each): A-block(mb) . W-block^T.
// ap  -> Ap + mbase*64: packed activations, quad pitch = a_qpitch floats
// wp  -> Wp + nb*Kq*64: this workgroup's packed weight block, quad pitch = 64 floats
template <int MB, int NIT>
__device__ __forceinline__ void rec_mm_core(const float* __restrict__ ap, long a_qpitch,
                                            const float* __restrict__ wp, int it0, int it1, int l, f32x4 (&acc)[MB]) {
    const int i = l & 15, kq = l >> 4;
    const float* a_lane = ap + i * 4;
    const float* w_lane = wp + i * 4;
    for (int itb = it0; itb < it1; itb += NIT) {
        float4 wv[NIT];
        float4 av[NIT][MB];
        if (itb + NIT <= it1) {
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const long q = 4 * (itb + u) + kq;
                wv[u] = *reinterpret_cast<const float4*>(w_lane + q * 64);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[u][mb] = *reinterpret_cast<const float4*>(a_lane + q * a_qpitch + mb * 64);
                LV_SCHED_BARRIER();   // keep the loads in consumption order: MFMA group u then waits only for loads <= u
            }
        } else {
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const bool ok = itb + u < it1;
                const long q = ok ? 4 * (itb + u) + kq : 4 * it0 + kq;     // clamped: load something valid, use zeros
                const float4 t = *reinterpret_cast<const float4*>(w_lane + q * 64);
                wv[u] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[u][mb] = *reinterpret_cast<const float4*>(a_lane + q * a_qpitch + mb * 64);
            }
        }
        LV_SCHED_BARRIER();   // every load of the macro-chunk is in flight before the first MFMA waits on one
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = lv_mfma_16x16x4(av[u][mb].x, wv[u].x, acc[mb]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = lv_mfma_16x16x4(av[u][mb].y, wv[u].y, acc[mb]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = lv_mfma_16x16x4(av[u][mb].z, wv[u].z, acc[mb]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = lv_mfma_16x16x4(av[u][mb].w, wv[u].w, acc[mb]);
        }
    }
}

// bf16 recurrent operands (throughput configuration): same structure on v_mfma_f32_16x16x32_bf16 -- one 16-byte load
// per lane carries 8 consecutive k of its row/column, K is covered 32 at a time, operands packed
//   Ap16[k/8][mb][16 rows][8 k],  Wp16[nb][k/8][16 cols][8 k]   (uint4 = one lane's 8 bf16)
template <int MB, int NIT>
__device__ __forceinline__ void rec_mm_core_bf16(const uint4* __restrict__ ap, long a_opitch,
                                                 const uint4* __restrict__ wp, int it0, int it1, int l, f32x4 (&acc)[MB]) {
    const int i = l & 15, kq = l >> 4;
    for (int itb = it0; itb < it1; itb += NIT) {
        uint4 wv[NIT];
        uint4 av[NIT][MB];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const bool ok = itb + u < it1;
            const long o = 4L * (ok ? itb + u : it0) + kq;
            const uint4 t = wp[o * 16 + i];
            wv[u] = ok ? t : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[u][mb] = ap[o * a_opitch + mb * 16 + i];
        }
        LV_SCHED_BARRIER();
#pragma unroll
        for (int u = 0; u < NIT; ++u)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = lv_mfma_16x16x32_bf16(av[u][mb], wv[u], acc[mb]);
    }
}

template <int MB> struct RecNit16 { static constexpr int value = MB <= 2 ? 8 : (MB == 4 ? 4 : 2); };

// ---- packing ---------------------------------------------------------------------------------------------------
// forward weights: Wp[nb][kq][j][e] = whh[(j>>2)*H + 4*nb + (j&3)][4*kq + e]   (0 outside)
__global__ __launch_bounds__(256) void pack_w_fwd_kernel(const float* __restrict__ whh, float* __restrict__ wp, int H, int Kq) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;            // one float4 of Wp
    const long total = (long)((H + 3) / 4) * Kq * 16;
    if (idx >= total) return;
    const int j = (int)(idx % 16);
    const int kq = (int)((idx / 16) % Kq);
    const int nb = (int)(idx / (16L * Kq));
    const int unit = 4 * nb + (j & 3);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (unit < H) {
        const float* row = whh + ((long)(j >> 2) * H + unit) * H;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int k = 4 * kq + e; if (k < H) v[e] = row[k]; }
    }
    *reinterpret_cast<float4*>(wp + idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

// backward weights: WpT[nb][nq][j][e] = whh[4*nq + e][16*nb + j]   (contraction over the 4H gate rows)
__global__ __launch_bounds__(256) void pack_w_bwd_kernel(const float* __restrict__ whh, float* __restrict__ wpT, int H, int Kq4) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)((H + 15) / 16) * Kq4 * 16;
    if (idx >= total) return;
    const int j = (int)(idx % 16);
    const int nq = (int)((idx / 16) % Kq4);
    const int nb = (int)(idx / (16L * Kq4));
    const int unit = 16 * nb + j;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (unit < H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int n = 4 * nq + e; if (n < 4 * H) v[e] = whh[(long)n * H + unit]; }
    }
    *reinterpret_cast<float4*>(wpT + idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

// activations: Ap[k/4][b/16][b%16][k%4] = src[b][k]  (buffer pre-zeroed; only valid entries written)
__global__ __launch_bounds__(256) void pack_act_kernel(const float* __restrict__ src, float* __restrict__ ap, int B, int K, int MBTp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * K) return;
    const int b = (int)(idx / K), k = (int)(idx % K);
    ap[((long)(k >> 2) * MBTp + (b >> 4)) * 64 + (b & 15) * 4 + (k & 3)] = src[idx];
}

// bf16 variants of the three packers (Kq8 = octs of K)
__global__ __launch_bounds__(256) void pack_w_fwd_bf16_kernel(const float* __restrict__ whh, uint4* __restrict__ wp, int H, int Kq8) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;            // one uint4 (8 bf16) of Wp16
    const long total = (long)((H + 3) / 4) * Kq8 * 16;
    if (idx >= total) return;
    const int j = (int)(idx % 16);
    const int oc = (int)((idx / 16) % Kq8);
    const int nb = (int)(idx / (16L * Kq8));
    const int unit = 4 * nb + (j & 3);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (unit < H) {
        const float* row = whh + ((long)(j >> 2) * H + unit) * H;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int k = 8 * oc + e; if (k < H) v[e] = row[k]; }
    }
    wp[idx] = make_uint4(lv_pack_bf16x2(v[0], v[1]), lv_pack_bf16x2(v[2], v[3]), lv_pack_bf16x2(v[4], v[5]), lv_pack_bf16x2(v[6], v[7]));
}

__global__ __launch_bounds__(256) void pack_w_bwd_bf16_kernel(const float* __restrict__ whh, uint4* __restrict__ wpT, int H, int Kq8) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)((H + 15) / 16) * Kq8 * 16;
    if (idx >= total) return;
    const int j = (int)(idx % 16);
    const int oc = (int)((idx / 16) % Kq8);
    const int nb = (int)(idx / (16L * Kq8));
    const int unit = 16 * nb + j;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (unit < H) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int n = 8 * oc + e; if (n < 4 * H) v[e] = whh[(long)n * H + unit]; }
    }
    wpT[idx] = make_uint4(lv_pack_bf16x2(v[0], v[1]), lv_pack_bf16x2(v[2], v[3]), lv_pack_bf16x2(v[4], v[5]), lv_pack_bf16x2(v[6], v[7]));
}

__global__ __launch_bounds__(256) void pack_act_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ ap, int B, int K, int MBTp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * K) return;
    const int b = (int)(idx / K), k = (int)(idx % K);
    ap[(((long)(k >> 3) * MBTp + (b >> 4)) * 16 + (b & 15)) * 8 + (k & 7)] = (unsigned short)lv_f32_to_bf16_bits(src[idx]);
}

struct LstmFwdP {
    const float* gx; const float* wp; float* hs; float* cs; float* gates; float* hp;   // hp: 2 packed buffers
    const uint8_t* dmask; float dscale; float* hdrop;
    int T, B, H, Kq, MBTp;
    int gx_unit_major;     // gx rows hold (i,f,g,o) of unit u at columns 4u..4u+3 instead of g*H + u
};

// ABL: ablation switches for profiles/microbench/lstm_step_probe.hip only (product launches use ABL = 0):
//      1 = skip the recurrent matmul, 2 = skip the gate math and all stores except h, 4 = skip the epilogue operand loads
template <int MB, int ABL = 0, bool BF = false>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(LstmFwdP p, int t) {
    __shared__ float red[4][MB][16][17];
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int B = p.B, H = p.H;
    const long BH = (long)B * H;
    const int nb = (int)blockIdx.x;
    const int u0 = nb * 4;
    const int mbase = (int)blockIdx.y * MB;
    const int rb = mbase * 16;
    const long hp_sz = (long)p.Kq * p.MBTp * 64;     // floats per packed buffer (f32: Kq quads x 64; bf16: Kq octs x 16 uint4)
    const float* gx_t = p.gx + (long)t * B * 4 * H;
    const float* hp_in = p.hp + (long)(t & 1) * hp_sz;
    float* hp_out = p.hp + (long)((t + 1) & 1) * hp_sz;
    const float* c_prev = p.cs + (long)t * BH;
    float* h_out = p.hs + (long)(t + 1) * BH;
    float* c_out = p.cs + (long)(t + 1) * BH;
    float* g_out = p.gates + (long)t * B * 4 * H;

    // epilogue operands for this thread's (batch row, unit) pairs: issue the loads before the matmul
    constexpr int NP = (64 * MB + 255) / 256;
    float pre[NP][4], cp[NP], keep[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pi = tid + 256 * q;
        const int bb = pi >> 2, uu = pi & 3;
        const int b = rb + bb, u = u0 + uu;
        const bool ok = (pi < 64 * MB) && b < B && u < H && !(ABL & 4);
        if (p.gx_unit_major) {       // one 16-byte load: the producing GEMM ran on gate-interleaved weight rows
            const float4 gq = ok ? *reinterpret_cast<const float4*>(gx_t + ((long)b * H + u) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            pre[q][0] = gq.x; pre[q][1] = gq.y; pre[q][2] = gq.z; pre[q][3] = gq.w;
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[q][g] = ok ? gx_t[(long)b * 4 * H + (long)g * H + u] : 0.f;
        }
        cp[q] = ok ? c_prev[(long)b * H + u] : 0.f;
        keep[q] = 1.f;                               // dropout keep-mask x scale of the output copy, fetched up front
        if (ok && p.hdrop && p.dmask) keep[q] = p.dmask[((long)b * p.T + t) * H + u] ? p.dscale : 0.f;
    }

    // recurrent matmul: this workgroup's 16 gate columns, K split over the 4 waves in units of 16
    const int nit = p.Kq / 4;
    const int ipw = (nit + 3) / 4;
    const int it0 = w * ipw;
    const int it1 = (it0 + ipw) < nit ? (it0 + ipw) : nit;
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (it0 < it1 && !(ABL & 1)) {
        if (BF)
            rec_mm_core_bf16<MB, RecNit16<MB>::value>(reinterpret_cast<const uint4*>(hp_in) + (long)mbase * 16, (long)p.MBTp * 16,
                                                      reinterpret_cast<const uint4*>(p.wp) + (long)nb * p.Kq * 16, it0, it1, l, acc);
        else
            rec_mm_core<MB, RecNit<MB>::value>(hp_in + (long)mbase * 64, (long)p.MBTp * 64, p.wp + (long)nb * p.Kq * 64,
                                               it0, it1, l, acc);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][mb][(l >> 4) * 4 + r][l & 15] = acc[mb][r];
    __syncthreads();

#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pi = tid + 256 * q;
        const int bb = pi >> 2, uu = pi & 3;
        const int b = rb + bb, u = u0 + uu;
        if (pi < 64 * MB && b < B && u < H) {
            float a[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = g * 4 + uu;
                const float s = (red[0][bb >> 4][bb & 15][col] + red[1][bb >> 4][bb & 15][col]) +
                                (red[2][bb >> 4][bb & 15][col] + red[3][bb >> 4][bb & 15][col]);
                a[g] = pre[q][g] + s;
            }
            if (ABL & 2) { h_out[(long)b * H + u] = a[0] + a[1] + a[2] + a[3]; continue; }
            float ig, fg, gg, og, h;
            float c;
            if (BF) {
                ig = lv_sigmoid_fast(a[0]); fg = lv_sigmoid_fast(a[1]); gg = lv_tanh_fast(a[2]); og = lv_sigmoid_fast(a[3]);
                c = fg * cp[q] + ig * gg;
                h = og * lv_tanh_fast(c);
            } else {
                ig = lv_sigmoid(a[0]); fg = lv_sigmoid(a[1]); gg = tanhf(a[2]); og = lv_sigmoid(a[3]);
                c = fg * cp[q] + ig * gg;
                h = og * tanhf(c);
            }
            *reinterpret_cast<float4*>(g_out + ((long)b * H + u) * 4) = make_float4(ig, fg, gg, og);   // unit-major record
            c_out[(long)b * H + u] = c;
            h_out[(long)b * H + u] = h;
            if (BF)     // packed bf16 copy for step t+1: oct = u/8, element u%8
                reinterpret_cast<unsigned short*>(hp_out)[((((long)(u >> 3)) * p.MBTp + (b >> 4)) * 16 + (b & 15)) * 8 + (u & 7)] =
                    (unsigned short)lv_f32_to_bf16_bits(h);
            else
                hp_out[((long)nb * p.MBTp + (b >> 4)) * 64 + (b & 15) * 4 + uu] = h;    // packed copy for step t+1
            if (p.hdrop) p.hdrop[(long)t * BH + (long)b * H + u] = h * keep[q];
        }
    }
}

struct LstmBwdP {
    const float* dh_ext;   // [T][B][H] or null
    const float* dh_last;  // [B][H] or null (added at t = T-1)
    const uint8_t* dmask; float dscale;   // dropout on dh_ext (mask in reference [B][T][H] layout)
    const float* wpT;      // packed transposed recurrent weights
    const float* gates; const float* cs;
    float* dG; float* dGsum; float* dGp; float* dh_part; float* dc_rec;
    int T, B, H, KS, Kq4, MBTp;
    uint16_t* dG16;        // optional bf16 image of dG for lv_gemm_b16 (bf16 path; dG may then be null)
};

// elementwise part of BPTT step t (KS = number of split-K slabs of the previous step's matmul, compile-time so
// that all slab loads are issued together)
template <int KS, bool BF = false>
__global__ __launch_bounds__(256) void lstm_step_bwd_elem_kernel(LstmBwdP p, int t) {
    const int B = p.B, H = p.H;
    const long BH = (long)B * H;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= BH) return;
    const int b = (int)(idx / H), u = (int)(idx % H);
    const bool first = (t == p.T - 1);
    float dh = 0.f;
    if (p.dh_ext) {
        float m = 1.f;
        if (p.dmask) m = p.dmask[((long)b * p.T + t) * H + u] ? p.dscale : 0.f;
        dh = p.dh_ext[(long)t * BH + idx] * m;
    }
    if (first && p.dh_last) dh += p.dh_last[idx];
    float parts[KS];
    float dcr = 0.f;
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};          // running sum of dG: read up front so the update below is not a second round trip
    const long si = (long)b * 4 * H + u;
    if (!first) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) parts[ks] = p.dh_part[(long)ks * BH + idx];
        dcr = p.dc_rec[idx];
#pragma unroll
        for (int g = 0; g < 4; ++g) gsum[g] = p.dGsum[si + (long)g * H];
    }
    const long gi = (long)t * B * 4 * H + (long)b * 4 * H + u;
    const float4 gq = *reinterpret_cast<const float4*>(p.gates + ((long)t * BH + idx) * 4);
    const float ig = gq.x, fg = gq.y, gg = gq.z, og = gq.w;
    const float c = p.cs[(long)(t + 1) * BH + idx];
    const float cprev = p.cs[(long)t * BH + idx];
    if (!first) {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s += parts[ks];
        dh += s;
    }
    const float tc = BF ? lv_tanh_fast(c) : tanhf(c);
    float dc = dh * og * (1.f - tc * tc);
    if (!first) dc += dcr;
    const float d_o = dh * tc;
    const float d_i = dc * gg, d_g = dc * ig, d_f = dc * cprev;
    float da[4];
    da[0] = d_i * ig * (1.f - ig);
    da[1] = d_f * fg * (1.f - fg);
    da[2] = d_g * (1.f - gg * gg);
    da[3] = d_o * og * (1.f - og);
    p.dc_rec[idx] = dc * fg;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (p.dG) p.dG[gi + (long)g * H] = da[g];
        p.dGsum[si + (long)g * H] = gsum[g] + da[g];
        const int n = g * H + u;                                   // packed copy: A operand of this step's matmul
        if (BF) {
            const unsigned short h16 = (unsigned short)lv_f32_to_bf16_bits(da[g]);
            if (p.dG16) p.dG16[gi + (long)g * H] = h16;
            reinterpret_cast<unsigned short*>(p.dGp)[(((long)(n >> 3) * p.MBTp + (b >> 4)) * 16 + (b & 15)) * 8 + (n & 7)] = h16;
        } else
            p.dGp[((long)(n >> 2) * p.MBTp + (b >> 4)) * 64 + (b & 15) * 4 + (n & 3)] = da[g];
    }
}

// recurrent matmul of BPTT step t: dh_part[ks][b][j] = sum_{n in slice ks} dG[t][b][n] * whh[n][j]
template <int MB, bool BF = false>
__global__ __launch_bounds__(256) void lstm_step_bwd_mm_kernel(LstmBwdP p, int t) {
    __shared__ float red[4][MB][16][17];
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int B = p.B, H = p.H;
    const int nb = (int)blockIdx.x / p.KS, ks = (int)blockIdx.x % p.KS;
    const int mbase = (int)blockIdx.y * MB;
    const int rb = mbase * 16;
    const int nit = p.Kq4 / 4;
    const int ips = (nit + p.KS - 1) / p.KS;        // iterations per split-K slice
    const int s0 = ks * ips;
    const int s1 = (s0 + ips) < nit ? (s0 + ips) : nit;
    const int ipw = (ips + 3) / 4;
    const int it0 = s0 + w * ipw;
    const int it1 = (it0 + ipw) < s1 ? (it0 + ipw) : s1;
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (it0 < it1) {
        if (BF)
            rec_mm_core_bf16<MB, RecNit16<MB>::value>(reinterpret_cast<const uint4*>(p.dGp) + (long)mbase * 16, (long)p.MBTp * 16,
                                                      reinterpret_cast<const uint4*>(p.wpT) + (long)nb * p.Kq4 * 16, it0, it1, l, acc);
        else
            rec_mm_core<MB, RecNit<MB>::value>(p.dGp + (long)mbase * 64, (long)p.MBTp * 64, p.wpT + (long)nb * p.Kq4 * 64,
                                               it0, it1, l, acc);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[w][mb][(l >> 4) * 4 + r][l & 15] = acc[mb][r];
    __syncthreads();
    for (int o = tid; o < 256 * MB; o += 256) {
        const int bb = o >> 4, col = o & 15;
        const int b = rb + bb, jj = nb * 16 + col;
        if (b < B && jj < H) {
            const float s = (red[0][bb >> 4][bb & 15][col] + red[1][bb >> 4][bb & 15][col]) +
                            (red[2][bb >> 4][bb & 15][col] + red[3][bb >> 4][bb & 15][col]);
            p.dh_part[(long)ks * B * H + (long)b * H + jj] = s;
        }
    }
}

// after the last BPTT step: dh0 = sum of slabs; dc0 = dc_rec (+ dh0 * (1 - h0^2) when h0 = tanh(c0))
__global__ __launch_bounds__(256) void lstm_bwd_finish_kernel(const float* dh_part, int KS, const float* dc_rec,
                                                              const float* h0, int tanh_init,
                                                              float* dh0, float* dc0, long BH) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= BH) return;
    float s = 0.f;
    for (int ks = 0; ks < KS; ++ks) s += dh_part[(long)ks * BH + idx];
    if (dh0) dh0[idx] = s;
    float dc = dc_rec[idx];
    if (tanh_init) { const float h = h0[idx]; dc += s * (1.f - h * h); }
    if (dc0) dc0[idx] = dc;
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, long in_ld, float* __restrict__ out,
                                                        long out_ld, int rows, int cols) {
    __shared__ float tile[32][33];
    const int tx = (int)threadIdx.x & 31, ty = (int)threadIdx.x >> 5;   // 32 x 8
    const int c0 = (int)blockIdx.x * 32, r0 = (int)blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(long)r * in_ld + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) out[(long)c * out_ld + r] = tile[tx][i];
    }
}

inline int pick_mb(int B) { return B <= 16 ? 1 : (B <= 32 ? 2 : (B <= 64 ? 4 : 8)); }

struct Geo { int MB, MBT, MBTp, Kq, Kq4, NBf, NBb, KS; long wp, hp, wpT, dGp, part, dcrec; };

inline int ksplit_for(int H) {
    int nb = (H + 15) / 16;
    int want = 256 / (nb > 0 ? nb : 1);
    int ks = 1;                       // power of two in {1,2,4,8}: the elementwise kernel is templated on it
    while (ks * 2 <= want && ks < 8) ks *= 2;
    return ks;
}

inline Geo geo(int B, int H, bool bf = false) {
    Geo g;
    g.MB = pick_mb(B);
    g.MBT = (B + 15) / 16;
    g.MBTp = h_round_up(g.MBT, g.MB);
    // f32: K in quads (16 k per MFMA group of 4);  bf16: K in octs (32 k per MFMA)
    g.Kq = bf ? h_round_up(H, 32) / 8 : h_round_up(H, 16) / 4;
    g.Kq4 = bf ? h_round_up(4 * H, 32) / 8 : h_round_up(4 * H, 16) / 4;
    g.NBf = (H + 3) / 4;
    g.NBb = (H + 15) / 16;
    g.KS = ksplit_for(H);
    g.wp = (long)g.NBf * g.Kq * 64;
    g.hp = 2L * g.Kq * g.MBTp * 64;
    g.wpT = (long)g.NBb * g.Kq4 * 64;
    g.dGp = (long)g.Kq4 * g.MBTp * 64;
    g.part = ((long)g.KS * B * H + 3) / 4 * 4;
    g.dcrec = ((long)B * H + 3) / 4 * 4;
    return g;
}

template <int MB, bool BF>
int launch_fwd_steps(const LstmFwdP& p, void* stream) {
    dim3 grid((unsigned)lv_cdiv(p.H, 4), (unsigned)(p.MBTp / MB)), block(256);
    for (int t = 0; t < p.T; ++t) LV_LAUNCH((lstm_step_fwd_kernel<MB, 0, BF>), grid, block, 0, stream, p, t);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

template <int MB, bool BF>
void launch_bwd_mm(const LstmBwdP& p, int t, void* stream) {
    dim3 grid((unsigned)(lv_cdiv(p.H, 16) * p.KS), (unsigned)(p.MBTp / MB)), block(256);
    LV_LAUNCH((lstm_step_bwd_mm_kernel<MB, BF>), grid, block, 0, stream, p, t);
}

template <bool BF>
void launch_bwd_elem(const LstmBwdP& p, int t, dim3 egrid, void* stream) {
    dim3 block(256);
    switch (p.KS) {
        case 1: LV_LAUNCH((lstm_step_bwd_elem_kernel<1, BF>), egrid, block, 0, stream, p, t); break;
        case 2: LV_LAUNCH((lstm_step_bwd_elem_kernel<2, BF>), egrid, block, 0, stream, p, t); break;
        case 4: LV_LAUNCH((lstm_step_bwd_elem_kernel<4, BF>), egrid, block, 0, stream, p, t); break;
        default: LV_LAUNCH((lstm_step_bwd_elem_kernel<8, BF>), egrid, block, 0, stream, p, t); break;
    }
}

template <bool BF>
int lstm_fwd_impl(const float* gx, const float* whh, float* hs, float* cs, float* gates,
                  const uint8_t* dmask, float dscale, float* hdrop, float* ws, int T, int B, int H, void* stream,
                  int gx_unit_major = 0) {
    if (!gx || !whh || !hs || !cs || !gates || !ws) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || H <= 0) return LV_ERR_SHAPE;
    if (dmask && !hdrop) return LV_ERR_ARG;
    if ((((uintptr_t)ws) & 15) != 0 || (((uintptr_t)gates) & 15) != 0) return LV_ERR_ALIGN;
    if (gx_unit_major && (((uintptr_t)gx) & 15) != 0) return LV_ERR_ALIGN;
    const Geo g = geo(B, H, BF);
    float* wp = ws;
    float* hp = ws + g.wp;
    // the packed state buffers carry zero padding for batch rows >= B and k >= H; with no padding every entry is written
    // (pack_act for step 0, the step epilogues afterwards) before it is read, and the fill is skipped
    if (g.MBTp * 16 != B || g.Kq * (BF ? 8 : 4) != H) hipMemsetAsync(hp, 0, (size_t)g.hp * sizeof(float), (hipStream_t)stream);
    if (BF) {
        LV_LAUNCH(pack_w_fwd_bf16_kernel, dim3((unsigned)lv_cdiv((long)g.NBf * g.Kq * 16, 256)), dim3(256), 0, stream, whh,
                  reinterpret_cast<uint4*>(wp), H, g.Kq);
        LV_LAUNCH(pack_act_bf16_kernel, dim3((unsigned)lv_cdiv((long)B * H, 256)), dim3(256), 0, stream, (const float*)hs,
                  reinterpret_cast<unsigned short*>(hp), B, H, g.MBTp);
    } else {
        LV_LAUNCH(pack_w_fwd_kernel, dim3((unsigned)lv_cdiv((long)g.NBf * g.Kq * 16, 256)), dim3(256), 0, stream, whh, wp, H, g.Kq);
        LV_LAUNCH(pack_act_kernel, dim3((unsigned)lv_cdiv((long)B * H, 256)), dim3(256), 0, stream, (const float*)hs, hp, B, H, g.MBTp);
    }
    LstmFwdP p{gx, wp, hs, cs, gates, hp, dmask, dscale, hdrop, T, B, H, g.Kq, g.MBTp, gx_unit_major};
    switch (g.MB) {
        case 1: return launch_fwd_steps<1, BF>(p, stream);
        case 2: return launch_fwd_steps<2, BF>(p, stream);
        case 4: return launch_fwd_steps<4, BF>(p, stream);
        default: return launch_fwd_steps<8, BF>(p, stream);
    }
}

template <bool BF>
int lstm_bwd_impl(const float* dh_ext, const float* dh_last, const uint8_t* dmask, float dscale,
                  const float* whh, const float* gates, const float* hs, const float* cs,
                  float* dG, float* dGsum, float* ws, float* dh0, float* dc0, int tanh_init,
                  int T, int B, int H, void* stream, uint16_t* dG16 = nullptr) {
    if (!whh || !gates || !cs || (!dG && !dG16) || !dGsum || !ws) return LV_ERR_ARG;
    if (T <= 0 || B <= 0 || H <= 0) return LV_ERR_SHAPE;
    if (tanh_init && !hs) return LV_ERR_ARG;
    if ((((uintptr_t)ws) & 15) != 0 || (((uintptr_t)gates) & 15) != 0) return LV_ERR_ALIGN;
    const Geo g = geo(B, H, BF);
    float* wpT = ws;
    float* dGp = wpT + g.wpT;
    float* part = dGp + g.dGp;
    float* dcrec = part + g.part;
    if (BF)
        LV_LAUNCH(pack_w_bwd_bf16_kernel, dim3((unsigned)lv_cdiv((long)g.NBb * g.Kq4 * 16, 256)), dim3(256), 0, stream, whh,
                  reinterpret_cast<uint4*>(wpT), H, g.Kq4);
    else
        LV_LAUNCH(pack_w_bwd_kernel, dim3((unsigned)lv_cdiv((long)g.NBb * g.Kq4 * 16, 256)), dim3(256), 0, stream, whh, wpT, H, g.Kq4);
    if (g.MBTp * 16 != B || g.Kq4 * (BF ? 8 : 4) != 4 * H) hipMemsetAsync(dGp, 0, (size_t)g.dGp * sizeof(float), (hipStream_t)stream);
    LstmBwdP p{dh_ext, dh_last, dmask, dscale, wpT, gates, cs, dG, dGsum, dGp, part, dcrec, T, B, H, g.KS, g.Kq4, g.MBTp, dG16};
    const long BH = (long)B * H;
    const bool need_h0 = (dh0 != nullptr) || tanh_init;
    dim3 egrid((unsigned)lv_cdiv(BH, 256)), block(256);
    for (int t = T - 1; t >= 0; --t) {
        launch_bwd_elem<BF>(p, t, egrid, stream);
        if (t > 0 || need_h0) {
            switch (g.MB) {
                case 1: launch_bwd_mm<1, BF>(p, t, stream); break;
                case 2: launch_bwd_mm<2, BF>(p, t, stream); break;
                case 4: launch_bwd_mm<4, BF>(p, t, stream); break;
                default: launch_bwd_mm<8, BF>(p, t, stream); break;
            }
        }
    }
    if (need_h0 || dc0) {
        if (!need_h0) {
            LV_LAUNCH(lstm_bwd_finish_kernel, egrid, block, 0, stream, (const float*)part, 0,
                      (const float*)dcrec, (const float*)nullptr, 0, (float*)nullptr, dc0, BH);
        } else {
            LV_LAUNCH(lstm_bwd_finish_kernel, egrid, block, 0, stream, (const float*)part, p.KS,
                      (const float*)dcrec, hs, tanh_init, dh0, dc0, BH);
        }
    }
    LV_CHECK_LAUNCH();
    return LV_OK;
}

}  // namespace

extern "C" int lv_lstm_bwd_ksplit(int H) { return ksplit_for(H); }

// floats of scratch lv_lstm_fwd_f32 / lv_lstm_bwd_f32 need (packed weights + packed state + split-K slabs)
extern "C" long lv_lstm_ws_floats(int B, int H) {
    if (B <= 0 || H <= 0) return 0;
    const Geo g = geo(B, H);
    const long fwd = g.wp + g.hp;
    const long bwd = g.wpT + g.dGp + g.part + g.dcrec;
    return (fwd > bwd ? fwd : bwd) + 64;
}

extern "C" int lv_transpose_f32(const float* in, float* out, int rows, int cols, void* stream) {
    if (!in || !out || rows < 0 || cols < 0) return LV_ERR_ARG;
    if (rows == 0 || cols == 0) return LV_OK;
    dim3 grid((unsigned)lv_cdiv(cols, 32), (unsigned)lv_cdiv(rows, 32)), block(256);
    LV_LAUNCH(transpose_kernel, grid, block, 0, stream, in, (long)cols, out, (long)rows, rows, cols);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// out[c][r] (pitch out_ld) = in[r][c] (pitch in_ld): strided-view transpose (weight-gradient operands of the bf16 path)
extern "C" int lv_transpose_ld_f32(const float* in, long in_ld, float* out, long out_ld, int rows, int cols, void* stream) {
    if (!in || !out || rows < 0 || cols < 0 || in_ld < cols || out_ld < rows) return LV_ERR_ARG;
    if (rows == 0 || cols == 0) return LV_OK;
    dim3 grid((unsigned)lv_cdiv(cols, 32), (unsigned)lv_cdiv(rows, 32)), block(256);
    LV_LAUNCH(transpose_kernel, grid, block, 0, stream, in, in_ld, out, out_ld, rows, cols);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_lstm_fwd_f32(const float* gx, const float* whh, float* hs, float* cs, float* gates,
                               const uint8_t* dmask, float dscale, float* hdrop, float* ws,
                               int T, int B, int H, void* stream) {
    return lstm_fwd_impl<false>(gx, whh, hs, cs, gates, dmask, dscale, hdrop, ws, T, B, H, stream);
}

// Throughput configuration: the recurrent product h_{t-1} W_hh^T runs on the bf16 matrix pipe (operands rounded to bf16,
// f32 accumulate); gates, cell state, outputs and everything stored stay f32.  Same arguments as lv_lstm_fwd_f32.
extern "C" int lv_lstm_fwd_bf16(const float* gx, const float* whh, float* hs, float* cs, float* gates,
                                const uint8_t* dmask, float dscale, float* hdrop, float* ws,
                                int T, int B, int H, void* stream) {
    return lstm_fwd_impl<true>(gx, whh, hs, cs, gates, dmask, dscale, hdrop, ws, T, B, H, stream);
}

// BPTT.  ws: lv_lstm_ws_floats(B, H) floats of scratch.  dh0/dc0 may be null.
extern "C" int lv_lstm_bwd_f32(const float* dh_ext, const float* dh_last, const uint8_t* dmask, float dscale,
                               const float* whh, const float* gates, const float* hs, const float* cs,
                               float* dG, float* dGsum, float* ws, float* dh0, float* dc0, int tanh_init,
                               int T, int B, int H, void* stream) {
    return lstm_bwd_impl<false>(dh_ext, dh_last, dmask, dscale, whh, gates, hs, cs, dG, dGsum, ws, dh0, dc0, tanh_init,
                                T, B, H, stream);
}

extern "C" int lv_lstm_bwd_bf16(const float* dh_ext, const float* dh_last, const uint8_t* dmask, float dscale,
                                const float* whh, const float* gates, const float* hs, const float* cs,
                                float* dG, float* dGsum, float* ws, float* dh0, float* dc0, int tanh_init,
                                int T, int B, int H, void* stream) {
    return lstm_bwd_impl<true>(dh_ext, dh_last, dmask, dscale, whh, gates, hs, cs, dG, dGsum, ws, dh0, dc0, tanh_init,
                               T, B, H, stream);
}

// lv_lstm_bwd_bf16 that also (or only) emits the bf16 image of dG the input-side GEMMs consume (lv_gemm_b16): dG16
// [T][B][4H] holds exactly what lv_cvt_bf16_f32 would make of dG; either of dG / dG16 may be null, not both.
extern "C" int lv_lstm_bwd_bf16_img(const float* dh_ext, const float* dh_last, const uint8_t* dmask, float dscale,
                                    const float* whh, const float* gates, const float* hs, const float* cs,
                                    float* dG, uint16_t* dG16, float* dGsum, float* ws, float* dh0, float* dc0,
                                    int tanh_init, int T, int B, int H, void* stream) {
    return lstm_bwd_impl<true>(dh_ext, dh_last, dmask, dscale, whh, gates, hs, cs, dG, dGsum, ws, dh0, dc0, tanh_init,
                               T, B, H, stream, dG16);
}

// lv_lstm_fwd_bf16 for a gx whose 4H columns are unit-major (column 4u + g): what lv_gemm_b16 produces from the
// lv_cvt_bf16_gates_f32 image of W_ih with lv_gate_interleave_f32-ed addends.  Everything else as lv_lstm_fwd_bf16.
extern "C" int lv_lstm_fwd_bf16_ug(const float* gx, const float* whh, float* hs, float* cs, float* gates,
                                   const uint8_t* dmask, float dscale, float* hdrop, float* ws,
                                   int T, int B, int H, void* stream) {
    return lstm_fwd_impl<true>(gx, whh, hs, cs, gates, dmask, dscale, hdrop, ws, T, B, H, stream, 1);
}

// lv_lstm_fwd_f32 for a unit-major gx (column 4u + g): the exact-f32 recurrence behind an input projection that was produced
// in the bf16-image path's column order (the encoder's exact forward, engine.LSTMEncoderEngine.exact_forward).
extern "C" int lv_lstm_fwd_f32_ug(const float* gx, const float* whh, float* hs, float* cs, float* gates,
                                  const uint8_t* dmask, float dscale, float* hdrop, float* ws,
                                  int T, int B, int H, void* stream) {
    return lstm_fwd_impl<false>(gx, whh, hs, cs, gates, dmask, dscale, hdrop, ws, T, B, H, stream, 1);
}

namespace {
// out[r][4u + g] = a[r][g*H + u] (+ b[r][g*H + u])
__global__ __launch_bounds__(256) void gate_interleave_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              long n, int H, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // index into out
    if (idx >= n) return;
    const long r = idx / (4L * H);
    const int c = (int)(idx % (4L * H));
    const long src = r * 4L * H + (long)(c & 3) * H + (c >> 2);
    out[idx] = a[src] + (b ? b[src] : 0.f);
}
}  // namespace

// Rows of 4H gate values (biases b_ih + b_hh, or the decoder's per-sequence z-projection) from gate-major to the
// unit-major column order of lv_lstm_fwd_bf16_ug's gx.  b may be null.
extern "C" int lv_gate_interleave_f32(const float* a, const float* b, int R, int H, float* out, void* stream) {
    if (!a || !out || R < 0 || H <= 0) return LV_ERR_ARG;
    if (R == 0) return LV_OK;
    const long n = (long)R * 4 * H;
    LV_LAUNCH(gate_interleave_kernel, dim3((unsigned)lv_cdiv(n, 256)), dim3(256), 0, stream, a, b, n, H, out);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
