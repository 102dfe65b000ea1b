// lv_lstm.hip -- the LSTM time recurrence (forward and BPTT) of the encoder/decoder, exact f32.
//
// Replaces nn.LSTM forward/backward on the hot path (modules/encoders/enc_lstm.py:60,
// modules/decoders/dec_lstm.py:104; autograd backward from text.py:384).  Math restated in
// oracle/text_vae_oracle.py (PyTorch gate order i|f|g|o).
//
// HBM layout: everything time-major so one timestep is one contiguous [B][*] slab:
//   gx    [T][B][4H]  x_t W_ih^T + b_ih + b_hh (+ z W_z^T for the decoder)  -- produced by lv_gemm_f32
//   hs,cs [T+1][B][H] hs[0]/cs[0] = initial state, step t writes index t+1
//   gates [T][B][4H]  activated i,f,g,o saved for BPTT;  dG [T][B][4H] grads wrt pre-activations
//
// Forward step (one launch per timestep; the kernel boundary is the grid-wide h_t hand-off, ~1.5 us on
// gfx950, cheaper than an in-kernel grid barrier -- MI355X_MICROARCH.md price list):
//   grid.x = ceil(H/4) workgroups; each owns 4 hidden units = 16 gate columns (i,f,g,o x 4 units) for all
//   batch rows, so the gate nonlinearity and the c/h update fuse into the epilogue.  The 4 waves split the
//   K=H contraction; each wave streams float4s of h_{t-1} (A) and of its W_hh rows (B) straight from L2
//   into v_mfma_f32_16x16x4_f32 (lane (i,kq) feeds element j of its float4 to MFMA j, the same
//   permutation of K on both operands), partial 16x16 tiles are combined through LDS.
//   block b lands on XCD b%8, so one XCD's 32 workgroups re-read the same 1/8 of W_hh every step and
//   it stays resident in that XCD's 4 MiB L2 (2.1 MB of 16.8 MB at H=1024).
// Backward step t = two launches: an elementwise kernel (gate grads, dc chain, dG[t]) and the recurrent
//   matmul dh_{t-1} = dG[t] . W_hh as split-K partial slabs against a pre-transposed W_hh^T (same core as
//   forward); the next step's elementwise kernel sums the slabs (deterministic, no atomics).
#include "lv_device.h"

namespace {

struct Ld4 { float v[4]; };

template <bool ALIGNED>
__device__ __forceinline__ Ld4 ld4(const float* __restrict__ row, bool rvalid, int k, int klim) {
    Ld4 o;
    const bool ok = rvalid && k < klim;
    if (ALIGNED) {
        // klim % 4 == 0 and k % 4 == 0 in this instantiation, so a float4 at k < klim is fully inside.  The load is
        // unconditional (clamped to offset 0 of a valid row) and zeroed by select: no exec-mask branch per load.
        const float4 t = *reinterpret_cast<const float4*>(row + (ok ? k : 0));
        o.v[0] = ok ? t.x : 0.f; o.v[1] = ok ? t.y : 0.f; o.v[2] = ok ? t.z : 0.f; o.v[3] = ok ? t.w : 0.f;
    } else {
        o.v[0] = o.v[1] = o.v[2] = o.v[3] = 0.f;
        if (ok) {
            o.v[0] = row[k];
            if (k + 1 < klim) o.v[1] = row[k + 1];
            if (k + 2 < klim) o.v[2] = row[k + 2];
            if (k + 3 < klim) o.v[3] = row[k + 3];
        }
    }
    return o;
}

// acc[mb] += A[rb + mb*16 + (l&15)][kbeg:kend] . W_lane_row[kbeg:kend]^T  (one 16x16 output tile per mb).
// The step kernels are latency-bound (one workgroup per CU, one wave per SIMD), so the loads of a whole
// macro-chunk (NIT x 16 floats of K per operand row) are issued back-to-back into registers before the first
// MFMA consumes them: at H=1024 a wave's entire K slice (16 iterations, 48 float4 at MB=2) is in flight at once.
template <int MB, bool ALIGNED, int NIT>
__device__ __forceinline__ void rec_mm_core(const float* __restrict__ A, long lda, int nrowsA, int rb,
                                            const float* __restrict__ wrow, bool wvalid,
                                            int kbeg, int kend, int l, f32x4 (&acc)[MB]) {
    const int i = l & 15, kq = l >> 4;
    const float* arow[MB];
    bool avalid[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int b = rb + mb * 16 + i;
        avalid[mb] = b < nrowsA;
        arow[mb] = A + (long)(avalid[mb] ? b : 0) * lda;
    }
    // Rows beyond the batch / beyond H read row 0 (a valid address) and their products land in output rows /
    // columns the epilogue discards, so only the K range needs zero-fill -- and a full macro-chunk needs none.
    for (int k0 = kbeg; k0 < kend; k0 += 16 * NIT) {
        Ld4 wv[NIT];
        Ld4 av[NIT][MB];
        if (ALIGNED && k0 + 16 * NIT <= kend) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int kk = k0 + 16 * it + 4 * kq;
                const float4 t = *reinterpret_cast<const float4*>(wrow + kk);
                wv[it].v[0] = t.x; wv[it].v[1] = t.y; wv[it].v[2] = t.z; wv[it].v[3] = t.w;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const float4 a = *reinterpret_cast<const float4*>(arow[mb] + kk);
                    av[it][mb].v[0] = a.x; av[it][mb].v[1] = a.y; av[it][mb].v[2] = a.z; av[it][mb].v[3] = a.w;
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int kk = k0 + 16 * it + 4 * kq;
                wv[it] = ld4<ALIGNED>(wrow, wvalid, kk, kend);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[it][mb] = ld4<ALIGNED>(arow[mb], avalid[mb], kk, kend);
            }
        }
        LV_SCHED_BARRIER();   // all loads of the macro-chunk are in flight before the first MFMA waits on one
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = lv_mfma_16x16x4(av[it][mb].v[j], wv[it].v[j], acc[mb]);
    }
}

template <int MB> struct RecNit { static constexpr int value = MB <= 2 ? 16 : (MB == 4 ? 8 : 4); };

__device__ __forceinline__ int round_up16(int x) { return (x + 15) & ~15; }

struct LstmFwdP {
    const float* gx; const float* whh; float* hs; float* cs; float* gates;
    const uint8_t* dmask; float dscale; float* hdrop;
    int T, B, H;
};

template <int MB, bool ALIGNED>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(LstmFwdP p, int t) {
    __shared__ float red[4][MB][16][17];
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int B = p.B, H = p.H;
    const long BH = (long)B * H;
    const int u0 = (int)blockIdx.x * 4;
    const int rb = (int)blockIdx.y * 16 * MB;
    const float* gx_t = p.gx + (long)t * B * 4 * H;
    const float* h_prev = p.hs + (long)t * BH;
    const float* c_prev = p.cs + (long)t * BH;
    float* h_out = p.hs + (long)(t + 1) * BH;
    float* c_out = p.cs + (long)(t + 1) * BH;
    float* g_out = p.gates + (long)t * B * 4 * H;

    // epilogue operands for this thread's (batch row, unit) pairs: issue the loads before the matmul
    constexpr int NP = (64 * MB + 255) / 256;
    float pre[NP][4], cp[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pi = tid + 256 * q;
        const int bb = pi >> 2, uu = pi & 3;
        const int b = rb + bb, u = u0 + uu;
        const bool ok = (pi < 64 * MB) && b < B && u < H;
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[q][g] = ok ? gx_t[(long)b * 4 * H + (long)g * H + u] : 0.f;
        cp[q] = ok ? c_prev[(long)b * H + u] : 0.f;
    }

    // recurrent matmul: this workgroup's 16 gate columns, K = H split over the 4 waves
    const int n = l & 15;
    const int unit = u0 + (n & 3);
    const bool wvalid = unit < H;
    const float* wrow = p.whh + ((long)(n >> 2) * H + (wvalid ? unit : 0)) * H;
    const int chunk = round_up16((H + 3) / 4);
    const int kbeg = w * chunk;
    const int kend = (kbeg + chunk) < H ? (kbeg + chunk) : H;
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    rec_mm_core<MB, ALIGNED, RecNit<MB>::value>(h_prev, H, B, rb, wrow, wvalid, kbeg, kend, l, acc);
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
            const float ig = lv_sigmoid(a[0]), fg = lv_sigmoid(a[1]), gg = tanhf(a[2]), og = lv_sigmoid(a[3]);
            const float c = fg * cp[q] + ig * gg;
            const float h = og * tanhf(c);
            const long gi = (long)b * 4 * H + u;
            g_out[gi] = ig; g_out[gi + H] = fg; g_out[gi + 2L * H] = gg; g_out[gi + 3L * H] = og;
            c_out[(long)b * H + u] = c;
            h_out[(long)b * H + u] = h;
            if (p.hdrop) {
                float m = 1.f;
                if (p.dmask) m = p.dmask[((long)b * p.T + t) * H + u] ? p.dscale : 0.f;
                p.hdrop[(long)t * BH + (long)b * H + u] = h * m;
            }
        }
    }
}

struct LstmBwdP {
    const float* dh_ext;   // [T][B][H] or null
    const float* dh_last;  // [B][H] or null (added at t = T-1)
    const uint8_t* dmask; float dscale;   // dropout on dh_ext (mask in reference [B][T][H] layout)
    const float* whhT;     // [H][4H]
    const float* gates; const float* cs;
    float* dG; float* dGsum; float* dh_part; float* dc_rec;
    int T, B, H, KS;
};

// elementwise part of BPTT step t (KS = number of split-K slabs of the previous step's matmul, compile-time so
// that all slab loads are issued together)
template <int KS>
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
    if (!first) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) parts[ks] = p.dh_part[(long)ks * BH + idx];
        dcr = p.dc_rec[idx];
    }
    const long gi = (long)t * B * 4 * H + (long)b * 4 * H + u;
    const float ig = p.gates[gi], fg = p.gates[gi + H], gg = p.gates[gi + 2L * H], og = p.gates[gi + 3L * H];
    const float c = p.cs[(long)(t + 1) * BH + idx];
    const float cprev = p.cs[(long)t * BH + idx];
    if (!first) {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s += parts[ks];
        dh += s;
    }
    const float tc = tanhf(c);
    float dc = dh * og * (1.f - tc * tc);
    if (!first) dc += dcr;
    const float d_o = dh * tc;
    const float d_i = dc * gg, d_g = dc * ig, d_f = dc * cprev;
    const float da_i = d_i * ig * (1.f - ig);
    const float da_f = d_f * fg * (1.f - fg);
    const float da_g = d_g * (1.f - gg * gg);
    const float da_o = d_o * og * (1.f - og);
    p.dc_rec[idx] = dc * fg;
    p.dG[gi] = da_i; p.dG[gi + H] = da_f; p.dG[gi + 2L * H] = da_g; p.dG[gi + 3L * H] = da_o;
    const long si = (long)b * 4 * H + u;
    if (first) {
        p.dGsum[si] = da_i; p.dGsum[si + H] = da_f; p.dGsum[si + 2L * H] = da_g; p.dGsum[si + 3L * H] = da_o;
    } else {
        p.dGsum[si] += da_i; p.dGsum[si + H] += da_f; p.dGsum[si + 2L * H] += da_g; p.dGsum[si + 3L * H] += da_o;
    }
}

// recurrent matmul of BPTT step t: dh_part[ks][b][j] = sum_{n in slice ks} dG[t][b][n] * whhT[j][n]
template <int MB, bool ALIGNED>
__global__ __launch_bounds__(256) void lstm_step_bwd_mm_kernel(LstmBwdP p, int t) {
    __shared__ float red[4][MB][16][17];
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int B = p.B, H = p.H, K = 4 * p.H;
    const int nb = (int)blockIdx.x / p.KS, ks = (int)blockIdx.x % p.KS;
    const int rb = (int)blockIdx.y * 16 * MB;
    const float* A = p.dG + (long)t * B * K;
    const int j = nb * 16 + (l & 15);
    const bool wvalid = j < H;
    const float* wrow = p.whhT + (long)(wvalid ? j : 0) * K;
    const int kc = round_up16((K + p.KS - 1) / p.KS);
    const int sbeg = ks * kc;
    const int send = (sbeg + kc) < K ? (sbeg + kc) : K;
    const int chunk = round_up16((kc + 3) / 4);
    int kbeg = sbeg + w * chunk;
    int kend = (kbeg + chunk) < send ? (kbeg + chunk) : send;
    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kbeg < kend) rec_mm_core<MB, ALIGNED, RecNit<MB>::value>(A, K, B, rb, wrow, wvalid, kbeg, kend, l, acc);
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

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int rows, int cols) {
    __shared__ float tile[32][33];
    const int tx = (int)threadIdx.x & 31, ty = (int)threadIdx.x >> 5;   // 32 x 8
    const int c0 = (int)blockIdx.x * 32, r0 = (int)blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) out[(long)c * rows + r] = tile[tx][i];
    }
}

template <int MB>
int launch_fwd_steps(const LstmFwdP& p, bool aligned, void* stream) {
    dim3 grid((unsigned)lv_cdiv(p.H, 4), (unsigned)lv_cdiv(p.B, 16 * MB)), block(256);
    for (int t = 0; t < p.T; ++t) {
        if (aligned) LV_LAUNCH((lstm_step_fwd_kernel<MB, true>), grid, block, 0, stream, p, t);
        else LV_LAUNCH((lstm_step_fwd_kernel<MB, false>), grid, block, 0, stream, p, t);
    }
    LV_CHECK_LAUNCH();
    return LV_OK;
}

template <int MB>
void launch_bwd_mm(const LstmBwdP& p, int t, bool aligned, void* stream) {
    dim3 grid((unsigned)(lv_cdiv(p.H, 16) * p.KS), (unsigned)lv_cdiv(p.B, 16 * MB)), block(256);
    if (aligned) LV_LAUNCH((lstm_step_bwd_mm_kernel<MB, true>), grid, block, 0, stream, p, t);
    else LV_LAUNCH((lstm_step_bwd_mm_kernel<MB, false>), grid, block, 0, stream, p, t);
}

inline int pick_mb(int B) { return B <= 16 ? 1 : (B <= 32 ? 2 : (B <= 64 ? 4 : 8)); }

}  // namespace

extern "C" int lv_lstm_bwd_ksplit(int H) {
    int nb = (H + 15) / 16;
    int want = 256 / (nb > 0 ? nb : 1);
    int ks = 1;                       // power of two in {1,2,4,8}: the elementwise kernel is templated on it
    while (ks * 2 <= want && ks < 8) ks *= 2;
    return ks;
}

extern "C" int lv_transpose_f32(const float* in, float* out, int rows, int cols, void* stream) {
    if (!in || !out || rows < 0 || cols < 0) return LV_ERR_ARG;
    if (rows == 0 || cols == 0) return LV_OK;
    dim3 grid((unsigned)lv_cdiv(cols, 32), (unsigned)lv_cdiv(rows, 32)), block(256);
    LV_LAUNCH(transpose_kernel, grid, block, 0, stream, in, out, rows, cols);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_lstm_fwd_f32(const float* gx, const float* whh, float* hs, float* cs, float* gates,
                               const uint8_t* dmask, float dscale, float* hdrop,
                               int T, int B, int H, void* stream) {
    if (!gx || !whh || !hs || !cs || !gates) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || H <= 0) return LV_ERR_SHAPE;
    if (dmask && !hdrop) return LV_ERR_ARG;
    LstmFwdP p{gx, whh, hs, cs, gates, dmask, dscale, hdrop, T, B, H};
    const bool aligned = (H % 4 == 0) && ((((uintptr_t)whh) | ((uintptr_t)hs)) & 15) == 0;
    switch (pick_mb(B)) {
        case 1: return launch_fwd_steps<1>(p, aligned, stream);
        case 2: return launch_fwd_steps<2>(p, aligned, stream);
        case 4: return launch_fwd_steps<4>(p, aligned, stream);
        default: return launch_fwd_steps<8>(p, aligned, stream);
    }
}

// BPTT.  dh_part must hold lv_lstm_bwd_ksplit(H) * B * H floats; dc_rec B*H floats (returns dc wrt c_0's
// successor chain, i.e. dL/dc_0 before the optional tanh-init term).  dh0/dc0 may be null.
extern "C" int lv_lstm_bwd_f32(const float* dh_ext, const float* dh_last, const uint8_t* dmask, float dscale,
                               const float* whhT, const float* gates, const float* hs, const float* cs,
                               float* dG, float* dGsum, float* dh_part, float* dc_rec,
                               float* dh0, float* dc0, int tanh_init,
                               int T, int B, int H, void* stream) {
    if (!whhT || !gates || !cs || !dG || !dGsum || !dh_part || !dc_rec) return LV_ERR_ARG;
    if (T <= 0 || B <= 0 || H <= 0) return LV_ERR_SHAPE;
    if (tanh_init && !hs) return LV_ERR_ARG;
    LstmBwdP p{dh_ext, dh_last, dmask, dscale, whhT, gates, cs, dG, dGsum, dh_part, dc_rec, T, B, H,
               lv_lstm_bwd_ksplit(H)};
    const bool aligned = ((((uintptr_t)whhT) | ((uintptr_t)dG)) & 15) == 0;
    const long BH = (long)B * H;
    const bool need_h0 = (dh0 != nullptr) || tanh_init;
    dim3 egrid((unsigned)lv_cdiv(BH, 256)), block(256);
    const int mb = pick_mb(B);
    for (int t = T - 1; t >= 0; --t) {
        switch (p.KS) {
            case 1: LV_LAUNCH((lstm_step_bwd_elem_kernel<1>), egrid, block, 0, stream, p, t); break;
            case 2: LV_LAUNCH((lstm_step_bwd_elem_kernel<2>), egrid, block, 0, stream, p, t); break;
            case 4: LV_LAUNCH((lstm_step_bwd_elem_kernel<4>), egrid, block, 0, stream, p, t); break;
            default: LV_LAUNCH((lstm_step_bwd_elem_kernel<8>), egrid, block, 0, stream, p, t); break;
        }
        if (t > 0 || need_h0) {
            switch (mb) {
                case 1: launch_bwd_mm<1>(p, t, aligned, stream); break;
                case 2: launch_bwd_mm<2>(p, t, aligned, stream); break;
                case 4: launch_bwd_mm<4>(p, t, aligned, stream); break;
                default: launch_bwd_mm<8>(p, t, aligned, stream); break;
            }
        }
    }
    if (need_h0 || dc0) {
        if (!need_h0) {
            // no recurrent term wanted: dc0 = dc_rec
            LV_LAUNCH(lstm_bwd_finish_kernel, egrid, block, 0, stream, (const float*)dh_part, 0,
                      (const float*)dc_rec, (const float*)nullptr, 0, (float*)nullptr, dc0, BH);
        } else {
            LV_LAUNCH(lstm_bwd_finish_kernel, egrid, block, 0, stream, (const float*)dh_part, p.KS,
                      (const float*)dc_rec, hs, tanh_init, dh0, dc0, BH);
        }
    }
    LV_CHECK_LAUNCH();
    return LV_OK;
}
