// lv_loss.hip -- the non-GEMM pieces of VAE.loss: reparameterise + analytic KL, token NLL, loss assembly.
//
// Replaces GaussianEncoderBase.encode / .reparameterize (modules/encoders/encoder.py:40-79: z = mu + eps*exp(0.5 lv),
// KL = 0.5*sum(mu^2 + exp(lv) - lv - 1)), nn.CrossEntropyLoss(reduce=False) + .sum(-1) of
// LSTMDecoder.reconstruct_error (modules/decoders/dec_lstm.py:143-148), the loss assembly of VAE.loss
// (modules/vae.py:95-98), and their autograd backward.  eps is an INPUT (host torch RNG in parity mode,
// lv_rng.hip Philox in throughput mode) -- SURVEY.md App. B.
#include "lv_device.h"

namespace {

// one sub-wave group of G lanes per batch row; G = 32 when nz <= 32 (two rows per wave64), else 64
template <int G>
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps,
                                                             float* __restrict__ z, float* __restrict__ kl,
                                                             int B, int ns, int nz) {
    const int tid = (int)threadIdx.x;
    const int row = ((int)blockIdx.x * 256 + tid) / G;
    const int j0 = tid % G;
    const bool ok = row < B;
    const float* mu = mulv + (long)(ok ? row : 0) * 2 * nz;
    const float* lv = mu + nz;
    float part = 0.f;
    for (int j = j0; j < nz; j += G) {
        if (!ok) break;
        const float m = mu[j], l = lv[j];
        const float sd = expf(0.5f * l);
        part += (m * m + expf(l)) - l - 1.f;
        for (int s = 0; s < ns; ++s) {
            const long zi = ((long)row * ns + s) * nz + j;
            z[zi] = m + eps[zi] * sd;
        }
    }
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    if (ok && j0 == 0) kl[row] = 0.5f * part;
}

__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps,
                                                             const float* __restrict__ dz, const float* __restrict__ dkl,
                                                             float* __restrict__ dmulv, int B, int ns, int nz) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * nz) return;
    const int b = (int)(idx / nz), j = (int)(idx % nz);
    const float m = mulv[(long)b * 2 * nz + j], l = mulv[(long)b * 2 * nz + nz + j];
    const float sd = expf(0.5f * l);
    float gz = 0.f, gze = 0.f;
    for (int s = 0; s < ns; ++s) {
        const long zi = ((long)b * ns + s) * nz + j;
        const float g = dz[zi];
        gz += g;
        gze += g * eps[zi];
    }
    const float gk = dkl[b];
    dmulv[(long)b * 2 * nz + j] = gz + gk * m;
    dmulv[(long)b * 2 * nz + nz + j] = gze * (0.5f * sd) + gk * (0.5f * (expf(l) - 1.f));
}

// online (max, sum-exp) merge
__device__ __forceinline__ void ms_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { m = mn; s = 0.f; return; }
    s = s * expf(m - mn) + s2 * expf(m2 - mn);
    m = mn;
}

// one workgroup per logits row r = t*B + b
__global__ __launch_bounds__(256) void softmax_nll_fwd_kernel(const float* __restrict__ logits, long ldl,
                                                              const int64_t* __restrict__ ids, long ids_stride, int tgt_off,
                                                              float* __restrict__ lse, float* __restrict__ nll,
                                                              int T, int B, int V) {
    __shared__ float sm[4], ss[4];
    const int r = (int)blockIdx.x;
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const float* row = logits + (long)r * ldl;
    float m = -INFINITY, s = 0.f;
    const bool vec = (ldl % 4 == 0) && ((((uintptr_t)logits) & 15) == 0);
    if (vec) {
        const int V4 = V & ~3;
        for (int k = tid * 4; k < V4; k += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(row + k);
            const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
            const float mn = fmaxf(m, mx);
            s = s * expf(m - mn) + ((expf(v.x - mn) + expf(v.y - mn)) + (expf(v.z - mn) + expf(v.w - mn)));
            m = mn;
        }
        for (int k = V4 + tid; k < V; k += 256) ms_merge(m, s, row[k], 1.f);
    } else {
        for (int k = tid; k < V; k += 256) ms_merge(m, s, row[k], 1.f);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float m2 = __shfl_xor(m, d, 64), s2 = __shfl_xor(s, d, 64);
        ms_merge(m, s, m2, s2);
    }
    if (l == 0) { sm[w] = m; ss[w] = s; }
    __syncthreads();
    if (tid == 0) {
        float M = sm[0], S = ss[0];
        for (int i = 1; i < 4; ++i) ms_merge(M, S, sm[i], ss[i]);
        const float L = M + logf(S);
        const int t = r / B, b = r % B;
        long tg = ids[(long)b * ids_stride + t + tgt_off];
        if (tg < 0) tg = 0;
        if (tg >= V) tg = V - 1;
        lse[r] = L;
        nll[r] = L - row[tg];
    }
}

// in place: logits[r][c] <- (exp(logits[r][c] - lse[r]) - [c == target_r]) * rowscale[b]
__global__ __launch_bounds__(256) void softmax_nll_bwd_kernel(float* __restrict__ logits, long ldl, const float* __restrict__ lse,
                                                              const int64_t* __restrict__ ids, long ids_stride, int tgt_off,
                                                              const float* __restrict__ rowscale, int T, int B, int V) {
    const int r = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    float* row = logits + (long)r * ldl;
    const int t = r / B, b = r % B;
    long tg = ids[(long)b * ids_stride + t + tgt_off];
    if (tg < 0) tg = 0;
    if (tg >= V) tg = V - 1;
    const float L = lse[r], sc = rowscale[b];
    const bool vec = (ldl % 4 == 0) && ((((uintptr_t)logits) & 15) == 0);
    const int itg = (int)tg;
    if (vec) {
        const int V4 = V & ~3;
        for (int k = tid * 4; k < V4; k += 1024) {
            float4 v = *reinterpret_cast<float4*>(row + k);
            v.x = (expf(v.x - L) - (k == itg ? 1.f : 0.f)) * sc;
            v.y = (expf(v.y - L) - (k + 1 == itg ? 1.f : 0.f)) * sc;
            v.z = (expf(v.z - L) - (k + 2 == itg ? 1.f : 0.f)) * sc;
            v.w = (expf(v.w - L) - (k + 3 == itg ? 1.f : 0.f)) * sc;
            *reinterpret_cast<float4*>(row + k) = v;
        }
        for (int k = V4 + tid; k < V; k += 256) row[k] = (expf(row[k] - L) - (k == itg ? 1.f : 0.f)) * sc;
    } else {
        for (int k = tid; k < V; k += 256) row[k] = (expf(row[k] - L) - (k == itg ? 1.f : 0.f)) * sc;
    }
}

// Same gradient, written as bf16 (RNE) into a separate image for lv_gemm_b16; the f32 logits are left untouched.
// Rounding here is the rounding lv_gemm_bf16 applies to the f32 dlogits on the fly, so both routes agree bit for bit.
__global__ __launch_bounds__(256) void softmax_nll_bwd_b16_kernel(const float* __restrict__ logits, long ldl,
                                                                  const float* __restrict__ lse, const int64_t* __restrict__ ids,
                                                                  long ids_stride, int tgt_off, const float* __restrict__ rowscale,
                                                                  uint16_t* __restrict__ out, long ldo, int T, int B, int V) {
    const int r = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    const float* row = logits + (long)r * ldl;
    uint16_t* orow = out + (long)r * ldo;
    const int t = r / B, b = r % B;
    long tg = ids[(long)b * ids_stride + t + tgt_off];
    if (tg < 0) tg = 0;
    if (tg >= V) tg = V - 1;
    const float L = lse[r], sc = rowscale[b];
    const bool vec = (ldl % 4 == 0) && ((((uintptr_t)logits) & 15) == 0) && (ldo % 4 == 0) && ((((uintptr_t)out) & 7) == 0);
    const int itg = (int)tg;
    int done = 0;
    if (vec) {
        const int V4 = V & ~3;
        for (int k = tid * 4; k < V4; k += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(row + k);
            const float g0 = (expf(v.x - L) - (k == itg ? 1.f : 0.f)) * sc;
            const float g1 = (expf(v.y - L) - (k + 1 == itg ? 1.f : 0.f)) * sc;
            const float g2 = (expf(v.z - L) - (k + 2 == itg ? 1.f : 0.f)) * sc;
            const float g3 = (expf(v.w - L) - (k + 3 == itg ? 1.f : 0.f)) * sc;
            *reinterpret_cast<uint2*>(orow + k) = make_uint2(lv_pack_bf16x2(g0, g1), lv_pack_bf16x2(g2, g3));
        }
        done = V4;
    }
    for (int k = done + tid; k < V; k += 256)
        orow[k] = (uint16_t)lv_f32_to_bf16_bits((expf(row[k] - L) - (k == itg ? 1.f : 0.f)) * sc);
    for (long k = V + tid; k < ldo; k += 256) orow[k] = 0;      // keep the row padding finite (zero)
}

// Fused route (lv_gemm_b16_nll): merge the per-piece (max, sum exp) statistics of every logits row -- one wave per row,
// fixed merge order -- into lse[r], and nll[r] = lse[r] - target logit.
__global__ __launch_bounds__(256) void softmax_nll_merge_kernel(const float2* __restrict__ part, int nparts,
                                                                const float* __restrict__ tgt, float* __restrict__ lse,
                                                                float* __restrict__ nll, int R) {
    const int r = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int l = (int)threadIdx.x & 63;
    if (r >= R) return;
    float m = -INFINITY, s = 0.f;
    // eight pieces per lane in flight (512 per round: one round at any vocabulary below 32 768), from clamped addresses, merged in
    // the same order as one by one -- load -> merge per piece was a chain of nparts / 64 memory round trips (9.6 us at V = 20001)
    for (int i0 = l; i0 < nparts; i0 += 512) {
        float2 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = part[(long)r * nparts + (i0 + 64 * u < nparts ? i0 + 64 * u : nparts - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + 64 * u < nparts) ms_merge(m, s, q[u].x, q[u].y);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float m2 = __shfl_xor(m, d, 64), s2 = __shfl_xor(s, d, 64);
        ms_merge(m, s, m2, s2);
    }
    if (l == 0) {
        const float L = m + logf(s);
        lse[r] = L;
        nll[r] = L - tgt[r];
    }
}

// dlogits as a bf16 image from the binary16 logits image of the fused route: (exp(x - lse) - [c == target]) * rowscale[b]
__global__ __launch_bounds__(256) void softmax_nll_bwd_h16_kernel(const uint16_t* __restrict__ logits16, long ldl,
                                                                  const float* __restrict__ lse, const int64_t* __restrict__ ids,
                                                                  long ids_stride, int tgt_off, const float* __restrict__ rowscale,
                                                                  uint16_t* __restrict__ out, long ldo, int T, int B, int V) {
    const int r = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    const uint16_t* row = logits16 + (long)r * ldl;
    uint16_t* orow = out + (long)r * ldo;
    const int t = r / B, b = r % B;
    long tg = ids[(long)b * ids_stride + t + tgt_off];
    if (tg < 0) tg = 0;
    if (tg >= V) tg = V - 1;
    const float L = lse[r], sc = rowscale[b];
    const int itg = (int)tg;
    const bool vec = (ldl % 8 == 0) && ((((uintptr_t)logits16) & 15) == 0) && (ldo % 8 == 0) && ((((uintptr_t)out) & 15) == 0);
    int done = 0;
    if (vec) {
        const int V8 = V & ~7;
        for (int k = tid * 8; k < V8; k += 2048) {
            const uint4 q = *reinterpret_cast<const uint4*>(row + k);
            const uint32_t wv[4] = {q.x, q.y, q.z, q.w};
            uint32_t o[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float x0 = lv_f16_bits_to_f32((uint16_t)(wv[h] & 0xFFFFu)), x1 = lv_f16_bits_to_f32((uint16_t)(wv[h] >> 16));
                const float g0 = (lv_exp_fast(x0 - L) - (k + 2 * h == itg ? 1.f : 0.f)) * sc;
                const float g1 = (lv_exp_fast(x1 - L) - (k + 2 * h + 1 == itg ? 1.f : 0.f)) * sc;
                o[h] = lv_pack_bf16x2(g0, g1);
            }
            *reinterpret_cast<uint4*>(orow + k) = make_uint4(o[0], o[1], o[2], o[3]);
        }
        done = V8;
    }
    for (int k = done + tid; k < V; k += 256)
        orow[k] = (uint16_t)lv_f32_to_bf16_bits((lv_exp_fast(lv_f16_bits_to_f32(row[k]) - L) - (k == itg ? 1.f : 0.f)) * sc);
    for (long k = V + tid; k < ldo; k += 256) orow[k] = 0;      // keep the row padding finite (zero)
}

// rec[b] = sum_t nll[t][b]; loss[b] = rec[b] + kl_weight * kl[b].  One wave per sequence: lanes stride over t (each
// keeps a sequential partial), then a fixed-shape butterfly -- deterministic, and one memory round trip instead of T.
__global__ __launch_bounds__(256) void vae_loss_kernel(const float* __restrict__ nll, const float* __restrict__ kl,
                                                       const float* __restrict__ klw, float* __restrict__ loss,
                                                       float* __restrict__ rec, int T, int B) {
    const int b = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int l = (int)threadIdx.x & 63;
    if (b >= B) return;
    float s = 0.f;
    for (int t = l; t < T; t += 64) s += nll[(long)t * B + b];
    s = lv_wave_sum(s);
    if (l == 0) {
        rec[b] = s;
        loss[b] = s + klw[0] * kl[b];
    }
}

// Loss assembly of one VAE.loss call + everything the trainer derives from it, in one single-workgroup launch:
// rec[b] = sum_t nll[t][b] (one wave per sequence, as vae_loss_kernel), loss[b] = rec[b] + w*kl[b],
// acc[0..2] += sum_b (loss, rec, kl) (the running sums text.py:381,426-427 read), and the backward seeds of
// mean_b(loss_b): rowscale[b] = g_loss[b], dkl[b] = w*g_loss[b].
constexpr int LA_WAVES = 16;
__global__ __launch_bounds__(64 * LA_WAVES) void loss_assemble_kernel(const float* __restrict__ nll, const float* __restrict__ kl,
                                                            const float* __restrict__ klw, const float* __restrict__ g_loss,
                                                            float* __restrict__ loss, float* __restrict__ rec,
                                                            float* __restrict__ rowscale, float* __restrict__ dkl,
                                                            float* __restrict__ acc, int T, int B,
                                                            unsigned long long* rng_state, unsigned long long rng_inc) {
    __shared__ float red[3][LA_WAVES];
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const float kw = klw[0];
    float sl = 0.f, sr = 0.f, sk = 0.f;
    for (int b = w; b < B; b += LA_WAVES) {
        float s = 0.f;
        for (int t = l; t < T; t += 64) s += nll[(long)t * B + b];
        s = lv_wave_sum(s);
        if (l == 0) {
            const float k = kl[b], lo = s + kw * k, g = g_loss[b];
            rec[b] = s; loss[b] = lo;
            rowscale[b] = g; dkl[b] = kw * g;
            sl += lo; sr += s; sk += k;
        }
    }
    if (l == 0) { red[0][w] = sl; red[1][w] = sr; red[2][w] = sk; }
    __syncthreads();
    if (tid < 3) {
        float t = 0.f;
        for (int i = 0; i < LA_WAVES; ++i) t += red[tid][i];
        acc[tid] += t;
    }
    // the Philox offset of the step's noise (lv_rng_noise_step with inc = 0 drew from it at the head of the step): advanced here, in
    // a launch the step has anyway, instead of by a launch of its own
    if (tid == 0 && rng_state) rng_state[1] += rng_inc;
}

// upstream grads (each may be null) -> per-row scales used by the backward kernels
__global__ __launch_bounds__(256) void loss_bwd_scales_kernel(const float* g_loss, const float* g_rec, const float* g_kl,
                                                              const float* klw, float* rowscale, float* dkl, int B) {
    const int b = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (b >= B) return;
    const float gl = g_loss ? g_loss[b] : 0.f;
    rowscale[b] = gl + (g_rec ? g_rec[b] : 0.f);
    dkl[b] = klw[0] * gl + (g_kl ? g_kl[b] : 0.f);
}

__global__ __launch_bounds__(256) void tanh_kernel(const float* __restrict__ in, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = tanhf(in[i]);
}

// out[c] = sum_r in[r][c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, long ld, int R, int C,
                                                     float* __restrict__ out, float* __restrict__ out2) {
    const int c = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int r0 = 0; r0 < R; r0 += 16) {                 // 16 independent loads per round trip, added in row order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = in[(long)(r0 + u < R ? r0 + u : 0) * ld + c];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += r0 + u < R ? v[u] : 0.f;
    }
    out[c] = s;
    if (out2) out2[c] = s;
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// `a` and `out` may alias (no __restrict__): the in-place gradient accumulation of the image tape
__global__ __launch_bounds__(256) void add_v4_kernel(const float* a, const float* b, float* out, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 x = *reinterpret_cast<const float4*>(a + 4 * i), y = *reinterpret_cast<const float4*>(b + 4 * i);
    *reinterpret_cast<float4*>(out + 4 * i) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}

}  // namespace

extern "C" int lv_reparam_kl_fwd_f32(const float* mulv, const float* eps, float* z, float* kl,
                                     int B, int ns, int nz, void* stream) {
    if (!mulv || !eps || !z || !kl) return LV_ERR_ARG;
    if (B <= 0 || ns <= 0 || nz <= 0) return LV_ERR_SHAPE;
    if (nz <= 32) {
        LV_LAUNCH((reparam_kl_fwd_kernel<32>), dim3((unsigned)lv_cdiv((long)B * 32, 256)), dim3(256), 0, stream,
                  mulv, eps, z, kl, B, ns, nz);
    } else {
        LV_LAUNCH((reparam_kl_fwd_kernel<64>), dim3((unsigned)lv_cdiv((long)B * 64, 256)), dim3(256), 0, stream,
                  mulv, eps, z, kl, B, ns, nz);
    }
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_reparam_kl_bwd_f32(const float* mulv, const float* eps, const float* dz, const float* dkl,
                                     float* dmulv, int B, int ns, int nz, void* stream) {
    if (!mulv || !eps || !dz || !dkl || !dmulv) return LV_ERR_ARG;
    if (B <= 0 || ns <= 0 || nz <= 0) return LV_ERR_SHAPE;
    LV_LAUNCH(reparam_kl_bwd_kernel, dim3((unsigned)lv_cdiv((long)B * nz, 256)), dim3(256), 0, stream,
              mulv, eps, dz, dkl, dmulv, B, ns, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_softmax_nll_fwd_f32(const float* logits, long ldl, const int64_t* ids, long ids_stride, int tgt_off,
                                      float* lse, float* nll, int T, int B, int V, void* stream) {
    if (!logits || !ids || !lse || !nll) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || V <= 0 || ldl < V) return LV_ERR_SHAPE;
    if (T == 0) return LV_OK;
    LV_LAUNCH(softmax_nll_fwd_kernel, dim3((unsigned)(T * B)), dim3(256), 0, stream, logits, ldl, ids, ids_stride,
              tgt_off, lse, nll, T, B, V);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_softmax_nll_bwd_f32(float* logits, long ldl, const float* lse, const int64_t* ids, long ids_stride,
                                      int tgt_off, const float* rowscale, int T, int B, int V, void* stream) {
    if (!logits || !ids || !lse || !rowscale) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || V <= 0 || ldl < V) return LV_ERR_SHAPE;
    if (T == 0) return LV_OK;
    LV_LAUNCH(softmax_nll_bwd_kernel, dim3((unsigned)(T * B)), dim3(256), 0, stream, logits, ldl, lse, ids, ids_stride,
              tgt_off, rowscale, T, B, V);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_softmax_nll_bwd_b16(const float* logits, long ldl, const float* lse, const int64_t* ids, long ids_stride,
                                      int tgt_off, const float* rowscale, uint16_t* dlogits, long ldo, int T, int B, int V,
                                      void* stream) {
    if (!logits || !ids || !lse || !rowscale || !dlogits) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || V <= 0 || ldl < V || ldo < V) return LV_ERR_SHAPE;
    if (T == 0) return LV_OK;
    LV_LAUNCH(softmax_nll_bwd_b16_kernel, dim3((unsigned)(T * B)), dim3(256), 0, stream, logits, ldl, lse, ids, ids_stride,
              tgt_off, rowscale, dlogits, ldo, T, B, V);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_vae_loss_f32(const float* nll, const float* kl, const float* kl_weight_dev,
                               float* loss, float* rec, int T, int B, void* stream) {
    if (!nll || !kl || !kl_weight_dev || !loss || !rec) return LV_ERR_ARG;
    if (T < 0 || B <= 0) return LV_ERR_SHAPE;
    LV_LAUNCH(vae_loss_kernel, dim3((unsigned)lv_cdiv(B, 4)), dim3(256), 0, stream, nll, kl, kl_weight_dev, loss, rec, T, B);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_loss_bwd_scales_f32(const float* g_loss, const float* g_rec, const float* g_kl,
                                      const float* kl_weight_dev, float* rowscale, float* dkl, int B, void* stream) {
    if (!kl_weight_dev || !rowscale || !dkl) return LV_ERR_ARG;
    if (B <= 0) return LV_ERR_SHAPE;
    LV_LAUNCH(loss_bwd_scales_kernel, dim3((unsigned)lv_cdiv(B, 256)), dim3(256), 0, stream, g_loss, g_rec, g_kl,
              kl_weight_dev, rowscale, dkl, B);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_tanh_f32(const float* in, float* out, long n, void* stream) {
    if (!in || !out || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    LV_LAUNCH(tanh_kernel, dim3((unsigned)lv_cdiv(n, 256)), dim3(256), 0, stream, in, out, n);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_colsum_f32(const float* in, long ld, int R, int C, float* out, float* out2, void* stream) {
    if (!in || !out || R < 0 || C <= 0 || ld < C) return LV_ERR_ARG;
    LV_LAUNCH(colsum_kernel, dim3((unsigned)lv_cdiv(C, 256)), dim3(256), 0, stream, in, ld, R, C, out, out2);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_add_f32(const float* a, const float* b, float* out, long n, void* stream) {
    if (!a || !b || !out || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    if ((n & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) == 0) {
        LV_LAUNCH(add_v4_kernel, dim3((unsigned)lv_cdiv(n / 4, 256)), dim3(256), 0, stream, a, b, out, n / 4);
        LV_CHECK_LAUNCH();
        return LV_OK;
    }
    LV_LAUNCH(add_kernel, dim3((unsigned)lv_cdiv(n, 256)), dim3(256), 0, stream, a, b, out, n);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// VAE.loss assembly (modules/vae.py:95-98) + the running report sums (text.py:381,426-427) + the backward seeds of
// loss.mean().backward() (text.py:382-384) in one launch.  nll [T][B]; kl, g_loss [B]; acc: device float[3] (+=).
extern "C" int lv_loss_assemble_f32(const float* nll, const float* kl, const float* kl_weight_dev, const float* g_loss,
                                    float* loss, float* rec, float* rowscale, float* dkl, float* acc, int T, int B,
                                    void* stream) {
    if (!nll || !kl || !kl_weight_dev || !g_loss || !loss || !rec || !rowscale || !dkl || !acc) return LV_ERR_ARG;
    if (T < 0 || B <= 0) return LV_ERR_SHAPE;
    LV_LAUNCH(loss_assemble_kernel, dim3(1), dim3(64 * LA_WAVES), 0, stream, nll, kl, kl_weight_dev, g_loss, loss, rec, rowscale,
              dkl, acc, T, B, (unsigned long long*)nullptr, 0ULL);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// The same, and rng_state[1] += rng_inc (the {seed, offset} state of lv_rng_*): the fused step draws its noise with
// lv_rng_noise_step(..., inc = 0) and advances the offset here -- one launch less per step.
extern "C" int lv_loss_assemble_rng_f32(const float* nll, const float* kl, const float* kl_weight_dev, const float* g_loss,
                                        float* loss, float* rec, float* rowscale, float* dkl, float* acc, int T, int B,
                                        uint64_t* rng_state, uint64_t rng_inc, void* stream) {
    if (!nll || !kl || !kl_weight_dev || !g_loss || !loss || !rec || !rowscale || !dkl || !acc || !rng_state) return LV_ERR_ARG;
    if (T < 0 || B <= 0) return LV_ERR_SHAPE;
    LV_LAUNCH(loss_assemble_kernel, dim3(1), dim3(64 * LA_WAVES), 0, stream, nll, kl, kl_weight_dev, g_loss, loss, rec, rowscale,
              dkl, acc, T, B, reinterpret_cast<unsigned long long*>(rng_state), (unsigned long long)rng_inc);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// Second half of the fused vocabulary projection + NLL (lv_gemm_b16_nll): part [R][nparts] (max, sum exp) pairs and the target
// logits -> lse [R], nll [R].
extern "C" int lv_softmax_nll_merge_f32(const float* part, int nparts, const float* tgt_logit, float* lse, float* nll, int R,
                                        void* stream) {
    if (!part || !tgt_logit || !lse || !nll) return LV_ERR_ARG;
    if (R < 0 || nparts <= 0) return LV_ERR_SHAPE;
    if (R == 0) return LV_OK;
    LV_LAUNCH(softmax_nll_merge_kernel, dim3((unsigned)lv_cdiv(R, 4)), dim3(256), 0, stream, reinterpret_cast<const float2*>(part),
              nparts, tgt_logit, lse, nll, R);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// lv_softmax_nll_bwd_b16 reading the binary16 logits image of the fused route.
extern "C" int lv_softmax_nll_bwd_h16(const uint16_t* logits16, long ldl, const float* lse, const int64_t* ids, long ids_stride,
                                      int tgt_off, const float* rowscale, uint16_t* dlogits, long ldo, int T, int B, int V,
                                      void* stream) {
    if (!logits16 || !ids || !lse || !rowscale || !dlogits) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || V <= 0 || ldl < V || ldo < V) return LV_ERR_SHAPE;
    if (T == 0) return LV_OK;
    LV_LAUNCH(softmax_nll_bwd_h16_kernel, dim3((unsigned)(T * B)), dim3(256), 0, stream, logits16, ldl, lse, ids, ids_stride,
              tgt_off, rowscale, dlogits, ldo, T, B, V);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
