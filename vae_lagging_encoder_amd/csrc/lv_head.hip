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
// one workgroup per batch row
__global__ __launch_bounds__(256) void enc_head_fwd_kernel(const float* __restrict__ hT, const float* __restrict__ wlin,
                                                           const float* __restrict__ eps, float* __restrict__ mulv,
                                                           float* __restrict__ z, float* __restrict__ kl,
                                                           int H, int ns, int nz) {
    LV_DYN_SHARED(smem);
    float* sh = reinterpret_cast<float*>(smem);          // [H] the row of h_T
    float* sm = sh + H;                                   // [2nz] mu | logvar
    float* sk = sm + 2 * nz;                              // [nz] KL terms
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    for (int h = tid; h < H; h += 256) sh[h] = hT[(long)b * H + h];
    __syncthreads();
    for (int o = w; o < 2 * nz; o += 4) {                 // one wave per output, lanes stride the contraction
        const float* wr = wlin + (long)o * H;
        float s = 0.f;
        for (int h = l; h < H; h += 64) s = fmaf(sh[h], wr[h], s);
        s = lv_wave_sum(s);
        if (l == 0) { sm[o] = s; mulv[(long)b * 2 * nz + o] = s; }
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
// one workgroup per 64 columns h; 4 sub-groups split the batch rows (dh_T) and the 2nz rows (dW_lin)
__global__ __launch_bounds__(256) void enc_head_bwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps,
                                                           const float* __restrict__ dz, const float* __restrict__ dkl,
                                                           const float* __restrict__ hT, const float* __restrict__ wlin,
                                                           float* __restrict__ dmulv, float* __restrict__ dhT,
                                                           float* __restrict__ gwlin, int B, int H, int ns, int nz) {
    LV_DYN_SHARED(smem);
    float* sd = reinterpret_cast<float*>(smem);           // [B][2nz] dmulv
    const int tid = (int)threadIdx.x;
    const int nz2 = 2 * nz;
    for (int i = tid; i < B * nz; i += 256) {
        const int b = i / nz, j = i % nz;
        const float m = mulv[(long)b * nz2 + j], lv = mulv[(long)b * nz2 + nz + j];
        const float sdv = expf(0.5f * lv);
        float gz = 0.f, gze = 0.f;
        for (int s = 0; s < ns; ++s) {
            const long zi = ((long)b * ns + s) * nz + j;
            const float g = dz[zi];
            gz += g;
            gze += g * eps[zi];
        }
        const float gk = dkl[b];
        const float dm = gz + gk * m, dl = gze * (0.5f * sdv) + gk * (0.5f * (expf(lv) - 1.f));
        sd[b * nz2 + j] = dm;
        sd[b * nz2 + nz + j] = dl;
        if (blockIdx.x == 0) { dmulv[(long)b * nz2 + j] = dm; dmulv[(long)b * nz2 + nz + j] = dl; }
    }
    __syncthreads();
    const int h = (int)blockIdx.x * 64 + (tid & 63), g = tid >> 6;
    if (h >= H) return;
    for (int b = g; b < B; b += 4) {                      // dh_T[b][h] = sum_j dmulv[b][j] W_lin[j][h]
        float s = 0.f;
        for (int j = 0; j < nz2; ++j) s = fmaf(sd[b * nz2 + j], wlin[(long)j * H + h], s);
        dhT[(long)b * H + h] = s;
    }
    for (int j = g; j < nz2; j += 4) {                    // dW_lin[j][h] = sum_b dmulv[b][j] h_T[b][h]
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(sd[b * nz2 + j], hT[(long)b * H + h], s);
        gwlin[(long)j * H + h] = s;
    }
}

// ---- decoder initial state and the z-part of its input projection ------------------------------------------------------
// thread n < H: c0[b][n] = z[b] . W_trans[n], h0 = tanh(c0); thread H + n', n' < 4H: Zp[b][n'] = z[b] . W_ih[n'][col0:] + b_ih + b_hh
// (written gate-major, or unit-major 4u+g when unit_major != 0).  z staged in LDS; a thread keeps its weight row in
// registers and walks the batch.
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
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
        for (int k0 = 0; k0 < nz; k0 += NZC) {
            const int kn = nz - k0 < NZC ? nz - k0 : NZC;
            for (int k = 0; k < kn; ++k) s = fmaf(sz[b * nz + k0 + k], wrow[k0 + k], s);
        }
        if (init) {
            c0[(long)b * H + r] = s;
            h0[(long)b * H + r] = tanhf(s);
        } else {
            zp[(long)b * 4 * H + ocol] = s + bias;
        }
    }
}

// ---- decoder tail of the backward ---------------------------------------------------------------------------------------
// blocks [0, nA): thread n < 4H: gW_ih[n][col0 + k] = sum_b dGsum[b][n] z[b][k], g_bih[n] = g_bhh[n] = sum_b dGsum[b][n];
//                 thread 4H + j, j < H: gW_trans[j][k] = sum_b dc0[b][j] z[b][k]
// blocks [nA, nA + B): dz[b][k] = sum_n dGsum[b][n] W_ih[n][col0 + k] + sum_j dc0[b][j] W_trans[j][k]
__global__ __launch_bounds__(256) void dec_tail_bwd_kernel(const float* __restrict__ dGsum, const float* __restrict__ dc0,
                                                           const float* __restrict__ z, const float* __restrict__ wih,
                                                           long ld_wih, int col0, const float* __restrict__ wtr,
                                                           float* __restrict__ gwih, long ld_gwih, float* __restrict__ gwtr,
                                                           float* __restrict__ gbih, float* __restrict__ gbhh,
                                                           float* __restrict__ dz, int nA, int B, int H, int nz) {
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
            for (int b = 0; b < B; ++b) {
                const float v = col[(long)b * cs];
                tot += v;
#pragma unroll
                for (int k = 0; k < NZC; ++k)
                    if (k < kn) acc[k] = fmaf(v, sz[b * nz + k0 + k], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < NZC; ++k)
                if (k < kn) orow[k0 + k] = acc[k];
        }
        if (nz == 0) for (int b = 0; b < B; ++b) tot += col[(long)b * cs];
        if (gate) { gbih[r] = tot; gbhh[r] = tot; }
        return;
    }
    // ---- dz row b: 8 parts x 32 latent lanes; part p walks rows n = p, p + 8, ...
    const int b = (int)blockIdx.x - nA;
    const int kl = tid & 31, part = tid >> 5;
    float* red = sm;                                      // [8][32]
    for (int k0 = 0; k0 < nz; k0 += 32) {
        const int k = k0 + kl;
        float s = 0.f;
        if (k < nz) {
            for (int n = part; n < 4 * H; n += 8) s = fmaf(dGsum[(long)b * 4 * H + n], wih[(long)n * ld_wih + col0 + k], s);
            for (int j = part; j < H; j += 8) s = fmaf(dc0[(long)b * H + j], wtr[(long)j * nz + k], s);
        }
        red[part * 32 + kl] = s;
        __syncthreads();
        if (part == 0 && k < nz) {
            float t = 0.f;
            for (int p2 = 0; p2 < 8; ++p2) t += red[p2 * 32 + kl];
            dz[(long)b * nz + k] = t;
        }
        __syncthreads();
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

// Backward of the same: (dz [B][ns][nz], dkl [B]) -> dmulv [B][2nz], dhT [B][H] (gradient entering the BPTT at the last
// step), gW_lin [2nz][H] ('=' semantics).
extern "C" int lv_enc_head_bwd_f32(const float* mulv, const float* eps, const float* dz, const float* dkl, const float* hT,
                                   const float* wlin, float* dmulv, float* dhT, float* gwlin, int B, int H, int ns, int nz,
                                   void* stream) {
    if (!mulv || !eps || !dz || !dkl || !hT || !wlin || !dmulv || !dhT || !gwlin) return LV_ERR_ARG;
    if (B <= 0 || H <= 0 || ns <= 0 || nz <= 0) return LV_ERR_SHAPE;
    const size_t sh = (size_t)B * 2 * nz * sizeof(float);
    if (sh > 60000) return LV_ERR_UNSUPPORTED;
    LV_LAUNCH(enc_head_bwd_kernel, dim3((unsigned)lv_cdiv(H, 64)), dim3(256), sh, stream, mulv, eps, dz, dkl, hT, wlin, dmulv,
              dhT, gwlin, B, H, ns, nz);
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

// Backward of the same, fed by the BPTT's sums: dGsum [B][4H] (gate-major, sum over time of the gate gradients) and
// dc0 [B][H] -> gW_ih[:, col0:col0+nz], g_b_ih, g_b_hh, gW_trans ('=' semantics) and dz [B][nz].
extern "C" int lv_dec_tail_bwd_f32(const float* dGsum, const float* dc0, const float* z, const float* wih, long ld_wih, int col0,
                                   const float* wtr, float* gwih, long ld_gwih, float* gwtr, float* gbih, float* gbhh, float* dz,
                                   int B, int H, int nz, void* stream) {
    if (!dGsum || !dc0 || !z || !wih || !wtr || !gwih || !gwtr || !gbih || !gbhh || !dz) return LV_ERR_ARG;
    if (B <= 0 || H <= 0 || nz <= 0 || ld_wih < col0 + nz || ld_gwih < col0 + nz) return LV_ERR_SHAPE;
    size_t sh = (size_t)B * nz * sizeof(float);
    if (sh < 8 * 32 * sizeof(float)) sh = 8 * 32 * sizeof(float);
    if (sh > 60000) return LV_ERR_UNSUPPORTED;
    const int nA = lv_cdiv(5L * H, 256);
    LV_LAUNCH(dec_tail_bwd_kernel, dim3((unsigned)(nA + B)), dim3(256), sh, stream, dGsum, dc0, z, wih, ld_wih, col0, wtr, gwih,
              ld_gwih, gwtr, gbih, gbhh, dz, nA, B, H, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
