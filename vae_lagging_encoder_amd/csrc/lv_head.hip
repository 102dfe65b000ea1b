// lv_head.hip -- the batch-sized ends of the two LSTM networks, each as ONE launch instead of a chain of tiny GEMMs.
//
// Around the two recurrences the reference runs a handful of contractions whose "long" dimension is the batch or the
// latent size: the encoder head h_T W_lin^T + reparameterise + KL (modules/encoders/enc_lstm.py:62, encoder.py:53-79),
// the decoder's initial state c0 = trans_linear(z), h0 = tanh(c0) and the z-part of its input projection
// (modules/decoders/dec_lstm.py:95-101), and their backward (text.py:384).  At B = 32, nz = 32 each of them is a few
// MFLOP: as separate tile GEMMs they cost 6-36 us apiece (a 128x128 tile kernel on a 32 x 32 x 4096 product) plus a
// launch boundary each.  Here every output element is one short dot product computed by one thread (or one wave), with
// the small operand staged in LDS.  Arithmetic is f32 FMA chains in a fixed order: deterministic, within 1e-6 of the
// MFMA GEMM route.
#include "lv_device.h"

namespace {

// ---- encoder head forward: mulv = h_T . W_lin^T ; z = mu + eps*exp(lv/2) ; KL = 0.5*sum(mu^2 + exp(lv) - lv - 1) -------
// one workgroup per batch row; a wave owns outputs o = w, w+4, ... and keeps 16 of them in flight (independent loads)
__global__ __launch_bounds__(256) void enc_head_fwd_kernel(const float* __restrict__ hT, const float* __restrict__ wlin,
                                                           const float* __restrict__ eps, float* __restrict__ mulv,
                                                           float* __restrict__ z, float* __restrict__ kl,
                                                           int H, int ns, int nz) {
    LV_DYN_SHARED(smem);
    float* sh = reinterpret_cast<float*>(smem);          // [H] the row of h_T
    float* sm = sh + H;                                   // [2nz] mu | logvar
    float* sk = sm + 2 * nz;                              // [nz] KL terms
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int nz2 = 2 * nz;
    for (int h = tid; h < H; h += 256) sh[h] = hT[(long)b * H + h];
    __syncthreads();
    for (int o0 = 0; o0 < nz2; o0 += 64) {
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int h = l; h < H; h += 64) {
            const float x = sh[h];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int o = o0 + w + 4 * i;
                acc[i] = fmaf(x, wlin[(long)(o < nz2 ? o : 0) * H + h], acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = o0 + w + 4 * i;
            const float s = lv_wave_sum(acc[i]);
            if (l == 0 && o < nz2) { sm[o] = s; mulv[(long)b * nz2 + o] = s; }
        }
    }
    __syncthreads();
    for (int j = tid; j < nz; j += 256) {
        const float m = sm[j], lv = sm[nz + j];
        const float sd = expf(0.5f * lv);
        sk[j] = (m * m + expf(lv)) - lv - 1.f;
        for (int s = 0; s < ns; ++s) {
            const long zi = ((long)b * ns + s) * nz + j;
            z[zi] = m + eps[zi] * sd;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int j = 0; j < nz; ++j) s += sk[j];
        kl[b] = 0.5f * s;
    }
}

// ---- encoder head backward: dmulv from (dz, dKL); dh_T = dmulv . W_lin ; dW_lin = dmulv^T . h_T --------------------------
// grid (ceil(H/64), SLICES): a workgroup covers 64 columns h; the B + 2nz output rows (dh_T rows, then dW_lin rows) are
// dealt round-robin to SLICES x 4 workers.  dz may arrive as `parts` partial sums [parts][B][ns][nz] (lv_dec_tail_bwd).
constexpr int HB_SLICES = 8;
__global__ __launch_bounds__(256) void enc_head_bwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps,
                                                           const float* __restrict__ dz, int parts, const float* __restrict__ dkl,
                                                           const float* __restrict__ hT, const float* __restrict__ wlin,
                                                           float* __restrict__ dmulv, float* __restrict__ dhT,
                                                           float* __restrict__ gwlin, int B, int H, int ns, int nz) {
    LV_DYN_SHARED(smem);
    float* sd = reinterpret_cast<float*>(smem);           // [B][2nz] dmulv
    const int tid = (int)threadIdx.x;
    const int nz2 = 2 * nz;
    const long pstride = (long)B * ns * nz;
    for (int i = tid; i < B * nz; i += 256) {
        const int b = i / nz, j = i % nz;
        const float m = mulv[(long)b * nz2 + j], lv = mulv[(long)b * nz2 + nz + j];
        const float sdv = expf(0.5f * lv);
        float gz = 0.f, gze = 0.f;
        for (int s = 0; s < ns; ++s) {
            const long zi = ((long)b * ns + s) * nz + j;
            float g = 0.f;
            for (int q = 0; q < parts; ++q) g += dz[q * pstride + zi];
            gz += g;
            gze += g * eps[zi];
        }
        const float gk = dkl[b];
        const float dm = gz + gk * m, dl = gze * (0.5f * sdv) + gk * (0.5f * (expf(lv) - 1.f));
        sd[b * nz2 + j] = dm;
        sd[b * nz2 + nz + j] = dl;
        if (blockIdx.x == 0 && blockIdx.y == 0) { dmulv[(long)b * nz2 + j] = dm; dmulv[(long)b * nz2 + nz + j] = dl; }
    }
    __syncthreads();
    const int h = (int)blockIdx.x * 64 + (tid & 63);
    if (h >= H) return;
    const int worker = (int)blockIdx.y * 4 + (tid >> 6);
    // Loads are issued in batches of 8 with a compile-time trip count: a thread that walks a runtime-length loop of
    // load -> fma pairs pays one memory round trip per iteration (the first version of this kernel took 66 us that way).
    for (int r = worker; r < B + nz2; r += HB_SLICES * 4) {
        float s0 = 0.f, s1 = 0.f;
        if (r < B) {                                      // dh_T[r][h] = sum_j dmulv[r][j] W_lin[j][h]
            for (int j0 = 0; j0 < nz2; j0 += 8) {
                float wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = wlin[(long)(j0 + u < nz2 ? j0 + u : 0) * H + h];
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    s0 = fmaf(j0 + u < nz2 ? sd[r * nz2 + j0 + u] : 0.f, wv[u], s0);
                    s1 = fmaf(j0 + u + 1 < nz2 ? sd[r * nz2 + j0 + u + 1] : 0.f, wv[u + 1], s1);
                }
            }
            dhT[(long)r * H + h] = s0 + s1;
        } else {                                          // dW_lin[j][h] = sum_b dmulv[b][j] h_T[b][h]
            const int j = r - B;
            for (int b0 = 0; b0 < B; b0 += 8) {
                float hv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) hv[u] = hT[(long)(b0 + u < B ? b0 + u : 0) * H + h];
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    s0 = fmaf(b0 + u < B ? sd[(b0 + u) * nz2 + j] : 0.f, hv[u], s0);
                    s1 = fmaf(b0 + u + 1 < B ? sd[(b0 + u + 1) * nz2 + j] : 0.f, hv[u + 1], s1);
                }
            }
            gwlin[(long)j * H + h] = s0 + s1;
        }
    }
}

// ---- decoder initial state and the z-part of its input projection ------------------------------------------------------
// thread n < H: c0[b][n] = z[b] . W_trans[n], h0 = tanh(c0); thread H + n', n' < 4H: Zp[b][n'] = z[b] . W_ih[n'][col0:] + b_ih + b_hh
// (written gate-major, or unit-major 4u+g when unit_major != 0).  z staged in LDS; a thread keeps (a 32-wide chunk of) its
// weight row in registers and walks the batch.
constexpr int NZC = 32;      // latent chunk held in registers
__global__ __launch_bounds__(256) void dec_init_kernel(const float* __restrict__ z, const float* __restrict__ wtr,
                                                       const float* __restrict__ wih, long ld_wih, int col0,
                                                       const float* __restrict__ bih, const float* __restrict__ bhh,
                                                       float* __restrict__ c0, float* __restrict__ h0, float* __restrict__ zp,
                                                       int unit_major, int B, int H, int nz) {
    LV_DYN_SHARED(smem);
    float* sz = reinterpret_cast<float*>(smem);           // [B][nz]
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < B * nz; i += 256) sz[i] = z[i];
    __syncthreads();
    const int n = (int)blockIdx.x * 256 + tid;
    if (n >= 5 * H) return;
    const bool init = n < H;
    const int r = init ? n : n - H;
    const float* wrow = init ? wtr + (long)r * nz : wih + (long)r * ld_wih + col0;
    const float bias = init ? 0.f : bih[r] + bhh[r];
    long ocol = r;
    if (!init && unit_major) ocol = 4L * (r % H) + r / H;
    float* orow = init ? c0 + r : zp + ocol;
    const long ostride = init ? (long)H : 4L * H;
    const bool one_chunk = nz <= NZC;
    for (int k0 = 0; k0 < nz; k0 += NZC) {
        const int kn = nz - k0 < NZC ? nz - k0 : NZC;
        float wr[NZC];
#pragma unroll
        for (int k = 0; k < NZC; ++k) wr[k] = wrow[k0 + (k < kn ? k : 0)];
        for (int b = 0; b < B; ++b) {
            const float* zb = sz + b * nz + k0;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < NZC; k += 2) {
                s0 = fmaf(k < kn ? zb[k] : 0.f, wr[k], s0);
                s1 = fmaf(k + 1 < kn ? zb[k + 1] : 0.f, wr[k + 1], s1);
            }
            const float s = s0 + s1;
            float* o = orow + (long)b * ostride;
            if (k0 == 0) *o = s + bias;
            else *o += s;
            if (init && one_chunk) h0[(long)b * H + r] = tanhf(s);
        }
    }
    if (init && !one_chunk)
        for (int b = 0; b < B; ++b) h0[(long)b * H + r] = tanhf(c0[(long)b * H + r]);
}

// ---- decoder tail of the backward ---------------------------------------------------------------------------------------
// blocks [0, nA): thread n < 4H: gW_ih[n][col0 + k] = sum_b dGsum[b][n] z[b][k], g_bih[n] = g_bhh[n] = sum_b dGsum[b][n];
//                 thread 4H + j, j < H: gW_trans[j][k] = sum_b dc0[b][j] z[b][k]
// blocks [nA, nA + parts): partial dz over a slice of DZ_ROWS contraction rows of the stacked [W_ih[:, col0:] ; W_trans]:
//                 dzp[part][b][k] = sum_{rows of the slice} d[b][row] W[row][k]   (the consumer sums the parts in order)
constexpr int DZ_ROWS = 128;
constexpr int DZ_BC = 32;       // batch rows staged per pass
__global__ __launch_bounds__(256) void dec_tail_bwd_kernel(const float* __restrict__ dGsum, const float* __restrict__ dc0,
                                                           const float* __restrict__ z, const float* __restrict__ wih,
                                                           long ld_wih, int col0, const float* __restrict__ wtr,
                                                           float* __restrict__ gwih, long ld_gwih, float* __restrict__ gwtr,
                                                           float* __restrict__ gbih, float* __restrict__ gbhh,
                                                           float* __restrict__ dzp, int nA, int B, int H, int nz) {
    LV_DYN_SHARED(smem);
    float* sm = reinterpret_cast<float*>(smem);
    const int tid = (int)threadIdx.x;
    if ((int)blockIdx.x < nA) {
        float* sz = sm;                                   // [B][nz]
        for (int i = tid; i < B * nz; i += 256) sz[i] = z[i];
        __syncthreads();
        const int n = (int)blockIdx.x * 256 + tid;
        if (n >= 5 * H) return;
        const bool gate = n < 4 * H;
        const int r = gate ? n : n - 4 * H;
        const float* col = gate ? dGsum + r : dc0 + r;
        const long cs = gate ? 4L * H : (long)H;
        float* orow = gate ? gwih + (long)r * ld_gwih + col0 : gwtr + (long)r * nz;
        float tot = 0.f;
        for (int k0 = 0; k0 < nz; k0 += NZC) {
            const int kn = nz - k0 < NZC ? nz - k0 : NZC;
            float acc[NZC];
#pragma unroll
            for (int k = 0; k < NZC; ++k) acc[k] = 0.f;
            tot = 0.f;
            for (int b0 = 0; b0 < B; b0 += 8) {           // 8 column loads in flight (see enc_head_bwd_kernel)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = col[(long)(b0 + u < B ? b0 + u : 0) * cs];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (b0 + u < B) {
                        tot += v[u];
                        const float* zb = sz + (b0 + u) * nz + k0;
#pragma unroll
                        for (int k = 0; k < NZC; ++k) acc[k] = fmaf(v[u], k < kn ? zb[k] : 0.f, acc[k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NZC; ++k)
                if (k < kn) orow[k0 + k] = acc[k];
        }
        if (gate) { gbih[r] = tot; gbhh[r] = tot; }
        return;
    }
    // ---- partial dz: this block's slice of the stacked contraction rows; thread = (latent k = tid & 31, batch lane tid >> 5);
    // the batch is walked in chunks of DZ_BC rows staged in LDS
    const int part = (int)blockIdx.x - nA;
    const int row0 = part * DZ_ROWS;
    const int nrows = 5 * H - row0 < DZ_ROWS ? 5 * H - row0 : DZ_ROWS;
    float* sdv = sm;                                      // [DZ_BC][DZ_ROWS] a chunk of the slice of d = [dGsum | dc0]
    const int kl = tid & 31, bg = tid >> 5;
    for (int bc = 0; bc < B; bc += DZ_BC) {
        const int nb = B - bc < DZ_BC ? B - bc : DZ_BC;
        __syncthreads();
        for (int i = tid; i < DZ_BC * DZ_ROWS; i += 256) {
            const int bl = i / DZ_ROWS, rr = i % DZ_ROWS;
            const int row = row0 + rr, b = bc + bl;
            float v = 0.f;
            if (rr < nrows && bl < nb) v = row < 4 * H ? dGsum[(long)b * 4 * H + row] : dc0[(long)b * H + (row - 4 * H)];
            sdv[i] = v;
        }
        __syncthreads();
        for (int k0 = 0; k0 < nz; k0 += 32) {
            const int k = k0 + kl;
            if (k >= nz) continue;
            float acc[DZ_BC / 8];
#pragma unroll
            for (int q = 0; q < DZ_BC / 8; ++q) acc[q] = 0.f;
            for (int r0 = 0; r0 < nrows; r0 += 8) {
                float wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = row0 + (r0 + u < nrows ? r0 + u : 0);
                    wv[u] = row < 4 * H ? wih[(long)row * ld_wih + col0 + k] : wtr[(long)(row - 4 * H) * nz + k];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float w1 = r0 + u < nrows ? wv[u] : 0.f;
#pragma unroll
                    for (int q = 0; q < DZ_BC / 8; ++q) acc[q] = fmaf(sdv[(bg + 8 * q) * DZ_ROWS + (r0 + u < nrows ? r0 + u : 0)], w1, acc[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < DZ_BC / 8; ++q) {
                const int bl = bg + 8 * q;
                if (bl < nb) dzp[((long)part * B + bc + bl) * nz + k] = acc[q];
            }
        }
    }
}

}  // namespace

// LSTMEncoder head + GaussianEncoderBase.encode (enc_lstm.py:62-64, encoder.py:40-57) in one launch.
// hT [B][H] (last hidden state), W_lin [2nz][H], eps [B][ns][nz] -> mulv [B][2nz], z [B][ns][nz], kl [B]
extern "C" int lv_enc_head_fwd_f32(const float* hT, const float* wlin, const float* eps, float* mulv, float* z, float* kl,
                                   int B, int H, int ns, int nz, void* stream) {
    if (!hT || !wlin || !eps || !mulv || !z || !kl) return LV_ERR_ARG;
    if (B <= 0 || H <= 0 || ns <= 0 || nz <= 0) return LV_ERR_SHAPE;
    const size_t sh = (size_t)(H + 3 * nz) * sizeof(float);
    if (sh > 60000) return LV_ERR_UNSUPPORTED;
    LV_LAUNCH(enc_head_fwd_kernel, dim3((unsigned)B), dim3(256), sh, stream, hT, wlin, eps, mulv, z, kl, H, ns, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// Backward of the same: (dz, dkl [B]) -> dmulv [B][2nz], dhT [B][H] (gradient entering the BPTT at the last step),
// gW_lin [2nz][H] ('=' semantics).  dz: [dz_parts][B][ns][nz], summed over the leading index in order (dz_parts = 1 for a
// plain gradient; lv_dec_tail_bwd_f32 hands over lv_dec_tail_parts(H) partial sums).
extern "C" int lv_enc_head_bwd_f32(const float* mulv, const float* eps, const float* dz, int dz_parts, const float* dkl,
                                   const float* hT, const float* wlin, float* dmulv, float* dhT, float* gwlin, int B, int H,
                                   int ns, int nz, void* stream) {
    if (!mulv || !eps || !dz || !dkl || !hT || !wlin || !dmulv || !dhT || !gwlin) return LV_ERR_ARG;
    if (B <= 0 || H <= 0 || ns <= 0 || nz <= 0 || dz_parts <= 0) return LV_ERR_SHAPE;
    const size_t sh = (size_t)B * 2 * nz * sizeof(float);
    if (sh > 60000) return LV_ERR_UNSUPPORTED;
    LV_LAUNCH(enc_head_bwd_kernel, dim3((unsigned)lv_cdiv(H, 64), HB_SLICES), dim3(256), sh, stream, mulv, eps, dz, dz_parts, dkl,
              hT, wlin, dmulv, dhT, gwlin, B, H, ns, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// LSTMDecoder.decode's z-dependent prologue (dec_lstm.py:95-101): c0 = z W_trans^T, h0 = tanh(c0), and
// Zp = z W_ih[:, col0:]^T + b_ih + b_hh (the part of the input projection that cat((word_embed, z_)) contributes),
// gate-major [B][4H] or unit-major (column 4u+g) for the unit-major Gx epilogue.
extern "C" int lv_dec_init_f32(const float* z, const float* wtr, const float* wih, long ld_wih, int col0, const float* bih,
                               const float* bhh, float* c0, float* h0, float* zp, int unit_major, int B, int H, int nz,
                               void* stream) {
    if (!z || !wtr || !wih || !bih || !bhh || !c0 || !h0 || !zp) return LV_ERR_ARG;
    if (B <= 0 || H <= 0 || nz <= 0 || ld_wih < col0 + nz) return LV_ERR_SHAPE;
    const size_t sh = (size_t)B * nz * sizeof(float);
    if (sh > 60000) return LV_ERR_UNSUPPORTED;
    LV_LAUNCH(dec_init_kernel, dim3((unsigned)lv_cdiv(5L * H, 256)), dim3(256), sh, stream, z, wtr, wih, ld_wih, col0, bih, bhh,
              c0, h0, zp, unit_major, B, H, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// number of partial sums lv_dec_tail_bwd_f32 writes for dz
extern "C" int lv_dec_tail_parts(int H) { return lv_cdiv(5L * H, DZ_ROWS); }

// Backward of the same, fed by the BPTT's sums: dGsum [B][4H] (gate-major, sum over time of the gate gradients) and
// dc0 [B][H] -> gW_ih[:, col0:col0+nz], g_b_ih, g_b_hh, gW_trans ('=' semantics) and dz = dGsum . W_ih[:, col0:] +
// dc0 . W_trans as lv_dec_tail_parts(H) partial sums dz_parts [parts][B][nz] over slices of the contraction (summed in
// order by the consumer: lv_enc_head_bwd_f32, or lv_colsum_f32 over the leading index).
extern "C" int lv_dec_tail_bwd_f32(const float* dGsum, const float* dc0, const float* z, const float* wih, long ld_wih, int col0,
                                   const float* wtr, float* gwih, long ld_gwih, float* gwtr, float* gbih, float* gbhh,
                                   float* dz_parts, int B, int H, int nz, void* stream) {
    if (!dGsum || !dc0 || !z || !wih || !wtr || !gwih || !gwtr || !gbih || !gbhh || !dz_parts) return LV_ERR_ARG;
    if (B <= 0 || H <= 0 || nz <= 0 || ld_wih < col0 + nz || ld_gwih < col0 + nz) return LV_ERR_SHAPE;
    size_t sh = (size_t)B * nz * sizeof(float);
    if (sh > 60000) return LV_ERR_UNSUPPORTED;
    if (sh < (size_t)DZ_BC * DZ_ROWS * sizeof(float)) sh = (size_t)DZ_BC * DZ_ROWS * sizeof(float);
    const int nA = lv_cdiv(5L * H, 256);
    const int parts = lv_cdiv(5L * H, DZ_ROWS);
    LV_LAUNCH(dec_tail_bwd_kernel, dim3((unsigned)(nA + parts)), dim3(256), sh, stream, dGsum, dc0, z, wih, ld_wih, col0, wtr, gwih,
              ld_gwih, gwtr, gbih, gbhh, dz_parts, nA, B, H, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
