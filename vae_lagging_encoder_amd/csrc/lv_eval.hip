// lv_eval.hip -- the evaluation statistics that reuse the hot path's forward.
//
// During aggressive training the reference evaluates, every `log_niter` iterations over the whole validation set
// (text.py:431-436) and at each epoch end (text.py:457): the mutual information I(x; z) under q (calc_mi, text.py:186-198 ->
// GaussianEncoderBase.calc_mi, modules/encoders/encoder.py:111-145), the number of active units (calc_au, text.py:200-227)
// and the importance-weighted NLL (VAE.nll_iw, modules/vae.py:100-129, with log_sum_exp of modules/utils.py:3-16 and
// eval_inference_dist, encoder.py:81-109).  The encoder / decoder passes they sit on are the hot path's HIP forward; the
// pieces below are what the reference computes on top of them with broadcasting tensor algebra -- here one launch each,
// no (z_batch, x_batch, nz) or (batch, nsamples, nz) temporaries.
#include "lv_device.h"

namespace {

__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { m = mn; s = 0.f; return; }
    s = s * expf(m - mn) + s2 * expf(m2 - mn);
    m = mn;
}

constexpr float LOG_2PI = 1.8378770664093453f;

// log N(z[b][s]; mu[b], diag exp(logvar[b])) for every sample: one thread per (b, s).  mu == NULL: standard normal prior.
__global__ __launch_bounds__(256) void gauss_logpdf_kernel(const float* __restrict__ z, const float* __restrict__ mu,
                                                           const float* __restrict__ logvar, float* __restrict__ out,
                                                           int B, int ns, int nz) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * ns) return;
    const int b = (int)(i / ns);
    const float* zr = z + i * nz;
    float q = 0.f, sl = 0.f;
    if (mu) {
        const float* m = mu + (long)b * nz;
        const float* lv = logvar + (long)b * nz;
        for (int k = 0; k < nz; ++k) {
            const float d = zr[k] - m[k];
            q += d * d / expf(lv[k]);
            sl += lv[k];
        }
    } else {
        for (int k = 0; k < nz; ++k) q += zr[k] * zr[k];
    }
    out[i] = -0.5f * q - 0.5f * ((float)nz * LOG_2PI + sl);
}

// out[r] = log(sum_c exp(in[r][c])) + add: one wave per row, online (max, sum)
__global__ __launch_bounds__(256) void logsumexp_rows_kernel(const float* __restrict__ in, long ld, int R, int C, float add,
                                                             float* __restrict__ out) {
    const int r = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int l = (int)threadIdx.x & 63;
    if (r >= R) return;
    float m = -INFINITY, s = 0.f;
    for (int c = l; c < C; c += 64) lse_merge(m, s, in[(long)r * ld + c], 1.f);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float m2 = __shfl_xor(m, d, 64), s2 = __shfl_xor(s, d, 64);
        lse_merge(m, s, m2, s2);
    }
    if (l == 0) out[r] = m + logf(s) + add;
}

// calc_mi, stage 1: log q(z_i) = logsumexp_j log N(z_i; mu_j, var_j) - log(Bx): one wave per z row i, lanes stride j
__global__ __launch_bounds__(256) void mi_logqz_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                       const float* __restrict__ z, float* __restrict__ log_qz,
                                                       int Bx, int Bz, int nz) {
    const int i = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int l = (int)threadIdx.x & 63;
    if (i >= Bz) return;
    const float* zr = z + (long)i * nz;
    float m = -INFINITY, s = 0.f;
    for (int j = l; j < Bx; j += 64) {
        const float* mj = mu + (long)j * nz;
        const float* lj = logvar + (long)j * nz;
        float q = 0.f, sl = 0.f;
        for (int k = 0; k < nz; ++k) {
            const float d = zr[k] - mj[k];
            q += d * d / expf(lj[k]);
            sl += lj[k];
        }
        lse_merge(m, s, -0.5f * q - 0.5f * ((float)nz * LOG_2PI + sl), 1.f);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float m2 = __shfl_xor(m, d, 64), s2 = __shfl_xor(s, d, 64);
        lse_merge(m, s, m2, s2);
    }
    if (l == 0) log_qz[i] = m + logf(s) - logf((float)Bx);
}

// calc_mi, stage 2 (single workgroup): mi = mean_b(-0.5 nz log 2pi - 0.5 sum_k (1 + logvar[b][k])) - mean_i log_qz[i]
__global__ __launch_bounds__(256) void mi_finish_kernel(const float* __restrict__ logvar, const float* __restrict__ log_qz,
                                                        float* __restrict__ out, int Bx, int Bz, int nz) {
    __shared__ double red[2][4];
    const int tid = (int)threadIdx.x;
    double ne = 0.0, lq = 0.0;
    for (int b = tid; b < Bx; b += 256) {
        float sl = 0.f;
        for (int k = 0; k < nz; ++k) sl += 1.f + logvar[(long)b * nz + k];
        ne += (double)(-0.5f * (float)nz * LOG_2PI - 0.5f * sl);
    }
    for (int i = tid; i < Bz; i += 256) lq += (double)log_qz[i];
    ne = lv_wave_sum(ne);
    lq = lv_wave_sum(lq);
    if ((tid & 63) == 0) { red[0][tid >> 6] = ne; red[1][tid >> 6] = lq; }
    __syncthreads();
    if (tid == 0) {
        const double a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const double c = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        out[0] = (float)(a / Bx - c / Bz);
        out[1] = (float)(a / Bx);
        out[2] = (float)(c / Bz);
    }
}

// calc_au accumulators: acc[k] += sum_b mu[b][k]  (mean == NULL)  or  sum_b (mu[b][k] - mean[k])^2
__global__ __launch_bounds__(256) void au_accum_kernel(const float* __restrict__ mu, const float* __restrict__ mean,
                                                       float* __restrict__ acc, int B, int nz) {
    const int k = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (k >= nz) return;
    float s = 0.f;
    const float mk = mean ? mean[k] : 0.f;
    for (int b = 0; b < B; ++b) {
        const float v = mu[(long)b * nz + k];
        s += mean ? (v - mk) * (v - mk) : v;
    }
    acc[k] += s;
}

}  // namespace

// log q(z|x) of GaussianEncoderBase.eval_inference_dist (encoder.py:81-109) for z [B][ns][nz] -> out [B][ns]; with
// mu == NULL the standard normal prior (VAE.eval_prior_dist, vae.py:135-145).
extern "C" int lv_gauss_logpdf_f32(const float* z, const float* mu, const float* logvar, float* out, int B, int ns, int nz,
                                   void* stream) {
    if (!z || !out || (mu && !logvar)) return LV_ERR_ARG;
    if (B <= 0 || ns <= 0 || nz <= 0) return LV_ERR_SHAPE;
    LV_LAUNCH(gauss_logpdf_kernel, dim3((unsigned)lv_cdiv((long)B * ns, 256)), dim3(256), 0, stream, z, mu, logvar, out, B, ns, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// log_sum_exp(value, dim=-1) (+ add) of modules/utils.py:3-16 over the rows of in [R][C] (ld)
extern "C" int lv_logsumexp_rows_f32(const float* in, long ld, int R, int C, float add, float* out, void* stream) {
    if (!in || !out) return LV_ERR_ARG;
    if (R <= 0 || C <= 0 || ld < C) return LV_ERR_SHAPE;
    LV_LAUNCH(logsumexp_rows_kernel, dim3((unsigned)lv_cdiv(R, 4)), dim3(256), 0, stream, in, ld, R, C, add, out);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// GaussianEncoderBase.calc_mi (encoder.py:111-145) given the encoder's (mu, logvar) [Bx][nz] and the samples z [Bz][nz]:
// out_dev[0] = I(x;z) estimate, out_dev[1] = E log q(z|x) (negative entropy), out_dev[2] = E log q(z); ws: Bz floats.
extern "C" int lv_calc_mi_f32(const float* mu, const float* logvar, const float* z, float* ws, float* out_dev, int Bx, int Bz,
                              int nz, void* stream) {
    if (!mu || !logvar || !z || !ws || !out_dev) return LV_ERR_ARG;
    if (Bx <= 0 || Bz <= 0 || nz <= 0) return LV_ERR_SHAPE;
    LV_LAUNCH(mi_logqz_kernel, dim3((unsigned)lv_cdiv(Bz, 4)), dim3(256), 0, stream, mu, logvar, z, ws, Bx, Bz, nz);
    LV_LAUNCH(mi_finish_kernel, dim3(1), dim3(256), 0, stream, logvar, (const float*)ws, out_dev, Bx, Bz, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// calc_au (text.py:200-227) accumulators over a batch of posterior means mu [B][nz]: acc_dev[k] += sum_b mu[b][k] (mean ==
// NULL, first pass) or sum_b (mu[b][k] - mean[k])^2 (second pass)
extern "C" int lv_au_accum_f32(const float* mu, const float* mean, float* acc_dev, int B, int nz, void* stream) {
    if (!mu || !acc_dev) return LV_ERR_ARG;
    if (B <= 0 || nz <= 0) return LV_ERR_SHAPE;
    LV_LAUNCH(au_accum_kernel, dim3((unsigned)lv_cdiv(nz, 256)), dim3(256), 0, stream, mu, mean, acc_dev, B, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// ---- generation (greedy / sample / beam decoding, modules/decoders/dec_lstm.py:163-367) -----------------
namespace {

// one wave per row: idx[r] = argmax_c in[r][c] (lowest index among equal maxima, as torch.argmax on CPU)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ in, long ld, int R, int C,
                                                          int64_t* __restrict__ idx) {
    const int r = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int l = (int)threadIdx.x & 63;
    if (r >= R) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = l; c < C; c += 64) {
        const float v = in[(long)r * ld + c];
        if (v > best || (v == best && c < bi)) { best = v; bi = c; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(best, d, 64);
        const int oi = __shfl_xor(bi, d, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (l == 0) idx[r] = bi == 0x7fffffff ? 0 : bi;
}

// out[r][c] = in[r][c] - logsumexp_c in[r][:]  (+ addrow[r]): F.log_softmax (+ the hypothesis' running log-probability)
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ in, long ld, int R, int C,
                                                               const float* __restrict__ addrow, float* __restrict__ out, long ldo) {
    __shared__ float sm[4], ss[4];
    const int r = (int)blockIdx.x, tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    float m = -INFINITY, s = 0.f;
    for (int c = tid; c < C; c += 256) lse_merge(m, s, in[(long)r * ld + c], 1.f);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float m2 = __shfl_xor(m, d, 64), s2 = __shfl_xor(s, d, 64);
        lse_merge(m, s, m2, s2);
    }
    if (l == 0) { sm[w] = m; ss[w] = s; }
    __syncthreads();
    float M = sm[0], S = ss[0];
    for (int i = 1; i < 4; ++i) lse_merge(M, S, sm[i], ss[i]);
    const float L = M + logf(S) - (addrow ? addrow[r] : 0.f);
    for (int c = tid; c < C; c += 256) out[(long)r * ldo + c] = in[(long)r * ld + c] - L;
}

// categorical draw from softmax(in[r][:]) by inverse CDF with the uniform u[r] in [0,1): single wave per row, two passes
// (logsumexp, then a blocked running sum); idx[r] = first c with cumsum_c softmax >= u
__global__ __launch_bounds__(64) void sample_rows_kernel(const float* __restrict__ in, long ld, int R, int C,
                                                         const float* __restrict__ u, int64_t* __restrict__ idx) {
    const int r = (int)blockIdx.x, l = (int)threadIdx.x;
    float m = -INFINITY, s = 0.f;
    for (int c = l; c < C; c += 64) lse_merge(m, s, in[(long)r * ld + c], 1.f);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float m2 = __shfl_xor(m, d, 64), s2 = __shfl_xor(s, d, 64);
        lse_merge(m, s, m2, s2);
    }
    const float target = u[r] * s;                     // in units of exp(x - m)
    float run = 0.f;
    int found = C - 1;
    bool done = false;
    for (int c0 = 0; c0 < C && !done; c0 += 64) {      // blocks of 64 consecutive columns, inclusive scan inside the wave
        const int c = c0 + l;
        float p = c < C ? expf(in[(long)r * ld + c] - m) : 0.f;
        float sc = p;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_up(sc, d, 64);
            if (l >= d) sc += o;
        }
        const bool hit = c < C && run + sc >= target;
        // lowest lane that hit
        int first = hit ? l : 64;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(first, d, 64);
            first = o < first ? o : first;
        }
        if (first < 64) { found = c0 + first; done = true; }
        run += __shfl(sc, 63, 64);
    }
    if (l == 0) idx[r] = found;
}

}  // namespace

// torch.argmax(logits, dim=1) of LSTMDecoder.greedy_decode (dec_lstm.py:304)
extern "C" int lv_argmax_rows_f32(const float* in, long ld, int R, int C, int64_t* idx, void* stream) {
    if (!in || !idx) return LV_ERR_ARG;
    if (R <= 0 || C <= 0 || ld < C) return LV_ERR_SHAPE;
    LV_LAUNCH(argmax_rows_kernel, dim3((unsigned)lv_cdiv(R, 4)), dim3(256), 0, stream, in, ld, R, C, idx);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// F.log_softmax(logits, dim=-1) + the live hypotheses' running log-probabilities (beam_search_decode, dec_lstm.py:214-218)
extern "C" int lv_log_softmax_rows_f32(const float* in, long ld, int R, int C, const float* addrow, float* out, long ldo,
                                       void* stream) {
    if (!in || !out) return LV_ERR_ARG;
    if (R <= 0 || C <= 0 || ld < C || ldo < C) return LV_ERR_SHAPE;
    LV_LAUNCH(log_softmax_rows_kernel, dim3((unsigned)R), dim3(256), 0, stream, in, ld, R, C, addrow, out, ldo);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// torch.multinomial(F.softmax(logits, dim=1), 1) of LSTMDecoder.sample_decode (dec_lstm.py:351-352) as an inverse-CDF draw
// from the caller's uniforms u [R] (lv_rng_* or torch): idx[r] = first c with cumsum softmax(logits[r])[c] >= u[r]
extern "C" int lv_sample_rows_f32(const float* in, long ld, int R, int C, const float* u, int64_t* idx, void* stream) {
    if (!in || !u || !idx) return LV_ERR_ARG;
    if (R <= 0 || C <= 0 || ld < C) return LV_ERR_SHAPE;
    LV_LAUNCH(sample_rows_kernel, dim3((unsigned)R), dim3(64), 0, stream, in, ld, R, C, u, idx);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
