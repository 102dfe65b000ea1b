// lv_optim.hip -- global grad-norm, clip coefficient, fused clip+SGD / clip+Adam over flat parameter buffers.
//
// Replaces torch.nn.utils.clip_grad_norm_(vae.parameters(), 5.0) (text.py:385, image.py:312; the norm spans
// encoder AND decoder grads -- SURVEY.md G1), optim.SGD(lr=1.0, momentum=0).step (text.py:325,387) and
// optim.Adam(lr=1e-3).step (image.py:267,314).  Parameters and grads live in flat HBM buffers (one segment per
// nn.Parameter, each 16-byte aligned) so the norm is one streaming reduction and the update one streaming pass.
// Scalars that change between graph replays (lr, step count) are read from device memory.
#include "lv_device.h"

namespace {

constexpr int NORM_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_stage1_kernel(const float* __restrict__ x, long n, float* __restrict__ partial, int nblk) {
    __shared__ float red[4];
    const int tid = (int)threadIdx.x;
    // contiguous chunk per block, 16-byte vector loads on the aligned interior
    const long per = (((n + nblk - 1) / nblk) + 3) & ~3L;
    const long beg = (long)blockIdx.x * per;
    long end = beg + per;
    if (end > n) end = n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (beg < end) {
        const bool vec = ((((uintptr_t)x) & 15) == 0);
        if (vec) {
            const long nv = (end - beg) / 4;
            const float4* x4 = reinterpret_cast<const float4*>(x + beg);
            for (long i = tid; i < nv; i += 256) {
                const float4 v = x4[i];
                s0 += v.x * v.x; s1 += v.y * v.y; s2 += v.z * v.z; s3 += v.w * v.w;
            }
            for (long i = beg + nv * 4 + tid; i < end; i += 256) s0 += x[i] * x[i];
        } else {
            for (long i = beg + tid; i < end; i += 256) s0 += x[i] * x[i];
        }
    }
    float s = (s0 + s1) + (s2 + s3);
    s = lv_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] (=|+=) sum(partial[0..NORM_BLOCKS))
__global__ __launch_bounds__(256) void sumsq_stage2_kernel(const float* __restrict__ partial, float* __restrict__ out, int accumulate, int nblk) {
    __shared__ double red[4];
    const int tid = (int)threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < nblk; i += 256) s += (double)partial[i];
    s = lv_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        const double tot = (red[0] + red[1]) + (red[2] + red[3]);
        out[0] = (float)(tot + (accumulate ? (double)out[0] : 0.0));
    }
}

// stage 1 over TWO buffers in one launch: blocks [0, nb1) reduce x1, blocks [nb1, nb1 + nb2) reduce x2
__global__ __launch_bounds__(256) void sumsq2_stage1_kernel(const float* __restrict__ x1, long n1, int nb1,
                                                            const float* __restrict__ x2, long n2, int nb2,
                                                            float* __restrict__ partial) {
    __shared__ float red[4];
    const int tid = (int)threadIdx.x;
    const bool first = (int)blockIdx.x < nb1;
    const float* x = first ? x1 : x2;
    const long n = first ? n1 : n2;
    const int nblk = first ? nb1 : nb2;
    const int blk = first ? (int)blockIdx.x : (int)blockIdx.x - nb1;
    const long per = (((n + nblk - 1) / nblk) + 3) & ~3L;
    const long beg = (long)blk * per;
    long end = beg + per;
    if (end > n) end = n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (beg < end) {
        if ((((uintptr_t)x) & 15) == 0) {
            const long nv = (end - beg) / 4;
            const float4* x4 = reinterpret_cast<const float4*>(x + beg);
            for (long i = tid; i < nv; i += 256) {
                const float4 v = x4[i];
                s0 += v.x * v.x; s1 += v.y * v.y; s2 += v.z * v.z; s3 += v.w * v.w;
            }
            for (long i = beg + nv * 4 + tid; i < end; i += 256) s0 += x[i] * x[i];
        } else {
            for (long i = beg + tid; i < end; i += 256) s0 += x[i] * x[i];
        }
    }
    float s = (s0 + s1) + (s2 + s3);
    s = lv_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The transaction gate of a fused training step (lv_clip_*_txn_f32).  A persistent LSTM recurrence that ran into its bounded
// hand-off spin returns early and leaves a non-zero status word; everything queued behind it then works on a half-finished
// recurrence.  The gate sits in the one single-thread spot every step passes before anything is applied -- the clip coefficient --
// and decides on the DEVICE whether the step counts: status words zero (and, data parallel, the exchanged guard element zero: no
// rank saw a timeout) -> the step's pending report sums are committed and the step counter advances; otherwise the sticky void
// flag goes up, and the update kernels (lv_sgd_step_txn_f32 / lv_scale_txn_f32) leave weights and gradients alone for this and
// every later step until the host has noticed (one read per loss window), moved the engines down the fallback ladder and replayed
// the voided steps.  txn = {void flag, steps committed, pending loss / rec / kl sums}; acc = the three running report sums.
struct TxnGate {
    const int* status1; const int* status2;      // the two engines' persistent-launch status words (either may be null)
    const float* guard;                          // data parallel: element of the exchanged gradient buffer that is non-zero iff a rank was void
    float* txn;                                  // null: no gate (the plain entries)
    float* acc;
};

__device__ __forceinline__ bool txn_gate(const TxnGate& g) {
    if (!g.txn) return true;
    const bool bad = (g.status1 && g.status1[0] != 0) || (g.status2 && g.status2[0] != 0) || (g.guard && g.guard[0] != 0.f) ||
                     g.txn[0] != 0.f;
    if (bad) {
        g.txn[0] = 1.f;
    } else {
        if (g.acc) { g.acc[0] += g.txn[2]; g.acc[1] += g.txn[3]; g.acc[2] += g.txn[4]; }
        g.txn[1] += 1.f;
    }
    g.txn[2] = 0.f; g.txn[3] = 0.f; g.txn[4] = 0.f;
    return !bad;
}

// stage 2 + clip coefficient: sumsq = sum(partial) in double; norm = sqrt; coef = min(1, max_norm / (norm + 1e-6))
// (extra: partial sums of squares the gradients' PRODUCERS emitted -- lv_gemm_b16_sumsq, lv_embed_scatter_full_sumsq_f32 -- for the
// tensors stage 1 was not run over; 256 or 1024 threads)
__global__ __launch_bounds__(1024) void clip_finish_kernel(const float* __restrict__ partial, int nblk, const float* __restrict__ extra,
                                                           int n_extra, float max_norm, float* sumsq, float* coef, float* norm_out,
                                                           TxnGate gate) {
    __shared__ double red[16];
    const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
    double s = 0.0;
    for (int i = tid; i < nblk; i += nt) s += (double)partial[i];
    if (n_extra > 0 && (((uintptr_t)extra) & 15) == 0) {
        // ~30k producer partials at the Yahoo shape, written on other XCDs (every load is a trip to memory): 16-byte loads, four in
        // flight per thread -- two round trips instead of thirty
        const float4* e4 = reinterpret_cast<const float4*>(extra);
        const int n4 = n_extra / 4;
        for (int i0 = tid; i0 < n4; i0 += 4 * nt) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = e4[i0 + u * nt < n4 ? i0 + u * nt : i0];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * nt < n4) s += ((double)v[u].x + (double)v[u].y) + ((double)v[u].z + (double)v[u].w);
        }
        for (int i = n4 * 4 + tid; i < n_extra; i += nt) s += (double)extra[i];
    } else {
        for (int i = tid; i < n_extra; i += nt) s += (double)extra[i];
    }
    s = lv_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < nt / 64; ++w) tot += red[w];
        const float ss = (float)tot;
        const float nrm = sqrtf(ss);
        float c = max_norm / (nrm + 1e-6f);
        if (c > 1.f) c = 1.f;
        if (sumsq) sumsq[0] = ss;
        coef[0] = c;
        if (norm_out) norm_out[0] = nrm;
        txn_gate(gate);
    }
}

// norm = sqrt(sumsq); coef = min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float* coef, float* norm_out, TxnGate gate) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float nrm = sqrtf(sumsq[0]);
        float c = max_norm / (nrm + 1e-6f);
        if (c > 1.f) c = 1.f;
        coef[0] = c;
        if (norm_out) norm_out[0] = nrm;
        txn_gate(gate);
    }
}

// data parallel: guard[0] = 1 when one of this rank's persistent launches reported a timeout, else 0 -- written into the tail
// padding of the gradient buffer that is about to be mean-all-reduced, so that afterwards it is non-zero on EVERY rank iff any
// rank was void (all ranks then void the same step and replay it together)
__global__ void txn_guard_kernel(const int* status1, const int* status2, float* guard) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        guard[0] = ((status1 && status1[0] != 0) || (status2 && status2[0] != 0)) ? 1.f : 0.f;
}

// g <- g*coef (optional write-back); p <- p - lr * g
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g, long n,
                                                  const float* __restrict__ lr, const float* __restrict__ coef,
                                                  int write_back, const float* __restrict__ void_flag,
                                                  float* __restrict__ x2, long n2) {
    if (void_flag && void_flag[0] != 0.f) return;      // the step was voided by the transaction gate: nothing is applied
    const float c = coef ? coef[0] : 1.f;
    if (x2 && c != 1.0f) {
        // clip_grad_norm_ scales EVERY gradient, the buffer that is not stepped too: in this launch instead of one of its own
        // (16-byte accesses where the buffer allows: 150 MB of decoder gradient at the Yahoo shape whenever the clip is active)
        const long stride2 = (long)gridDim.x * 256;
        const long m4 = (((uintptr_t)x2) & 15) == 0 ? n2 / 4 : 0;
        float4* x4 = reinterpret_cast<float4*>(x2);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < m4; i += stride2) {
            float4 v = x4[i];
            v.x *= c; v.y *= c; v.z *= c; v.w *= c;
            x4[i] = v;
        }
        for (long i = m4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride2) x2[i] *= c;
    }
    const float a = lr[0];
    const bool wb = write_back && c != 1.0f;      // g * 1 is g: the clipped-gradient write-back is skipped when the clip is inactive
    const long stride = (long)gridDim.x * 256;
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g)) & 15) == 0;
    if (vec) {
        const long n4 = n / 4;
        float4* p4 = reinterpret_cast<float4*>(p);
        float4* g4 = reinterpret_cast<float4*>(g);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            float4 gv = g4[i], pv = p4[i];
            gv.x *= c; gv.y *= c; gv.z *= c; gv.w *= c;
            pv.x -= a * gv.x; pv.y -= a * gv.y; pv.z -= a * gv.z; pv.w -= a * gv.w;
            p4[i] = pv;
            if (wb) g4[i] = gv;
        }
        for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const float gv = g[i] * c;
            p[i] -= a * gv;
            if (wb) g[i] = gv;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const float gv = g[i] * c;
            p[i] -= a * gv;
            if (wb) g[i] = gv;
        }
    }
}

__device__ __forceinline__ float bf16_bits_hi_to_f32(uint32_t bits_in_high_half) {      // bf16 = the high half of an f32
    float f;
    memcpy(&f, &bits_in_high_half, 4);
    return f;
}

// dst[i] = bf16(src[i]) * scale: the way back from a bf16 gradient payload (data-parallel exchange), with the 1/world mean folded in
__global__ __launch_bounds__(256) void cvt_f32_bf16_scaled_kernel(const uint16_t* __restrict__ src, long n, float scale,
                                                                  float* __restrict__ dst) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const uint2 q = *reinterpret_cast<const uint2*>(src + i);
        float4 o;
        o.x = bf16_bits_hi_to_f32(q.x << 16) * scale; o.y = bf16_bits_hi_to_f32(q.x & 0xFFFF0000u) * scale;
        o.z = bf16_bits_hi_to_f32(q.y << 16) * scale; o.w = bf16_bits_hi_to_f32(q.y & 0xFFFF0000u) * scale;
        *reinterpret_cast<float4*>(dst + i) = o;
    } else {
        for (long j = i; j < n; ++j) dst[j] = bf16_bits_hi_to_f32((uint32_t)src[j] << 16) * scale;
    }
}

__global__ __launch_bounds__(256) void keep_scale_kernel(float* __restrict__ x, const uint8_t* __restrict__ keep, float kscale, int T,
                                                         int Bsz, int C, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long r = i / C;
    const int c = (int)(i % C);
    x[i] *= keep[((r % Bsz) * (long)T + r / Bsz) * C + c] ? kscale : 0.f;
}

// four consecutive columns per thread (C % 4 == 0, 16-byte aligned x, 4-byte aligned keep): one float4, four keep bytes in one
// load, one row division per four elements -- the scalar form above spends its time on a 64-bit division per element
__global__ __launch_bounds__(256) void keep_scale_v4_kernel(float* __restrict__ x, const uint8_t* __restrict__ keep, float kscale, int T,
                                                            int Bsz, int C4, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int r = (int)(i / C4), c4 = (int)(i - (long)r * C4);
    const int t = r / Bsz, b = r - t * Bsz;
    const uint32_t k = *reinterpret_cast<const uint32_t*>(keep + ((long)b * T + t) * (4L * C4) + 4 * c4);
    float4 v = reinterpret_cast<float4*>(x)[i];
    v.x *= (k & 0xFFu) ? kscale : 0.f;
    v.y *= (k & 0xFF00u) ? kscale : 0.f;
    v.z *= (k & 0xFF0000u) ? kscale : 0.f;
    v.w *= (k & 0xFF000000u) ? kscale : 0.f;
    reinterpret_cast<float4*>(x)[i] = v;
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, const float* __restrict__ coef,
                                                    const float* __restrict__ void_flag) {
    if (void_flag && void_flag[0] != 0.f) return;
    const float c = coef[0];
    if (c == 1.0f) return;            // clip inactive (coef is exactly 1): x * 1 is x bit for bit, skip the pass
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] *= c;
}

// torch.optim.Adam (no amsgrad, no weight decay): state = {m, v}; step count t read from device (float)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, const float* __restrict__ lr,
                                                   const float* __restrict__ coef, const float* __restrict__ step,
                                                   float beta1, float beta2, float eps, int write_back) {
    const float c = coef ? coef[0] : 1.f;
    const float a = lr[0];
    const float t = step[0];
    const float bc1 = 1.f - powf(beta1, t);
    const float bc2 = 1.f - powf(beta2, t);
    const float step_size = a / bc1;
    const float bc2s = sqrtf(bc2);
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gv = g[i] * c;
        const float mi = m[i] + (gv - m[i]) * (1.f - beta1);        // lerp form, as torch's _single_tensor_adam
        const float vi = v[i] * beta2 + (1.f - beta2) * gv * gv;
        const float denom = sqrtf(vi) / bc2s + eps;
        p[i] -= step_size * (mi / denom);
        m[i] = mi; v[i] = vi;
        if (write_back) g[i] = gv;
    }
}

// out[0] += sum(x[0..n))  (single workgroup; n is a batch size)
__global__ __launch_bounds__(256) void sum_accum_kernel(const float* __restrict__ x, long n, float* out) {
    __shared__ float red[4];
    const int tid = (int)threadIdx.x;
    float s = 0.f;
    for (long i = tid; i < n; i += 256) s += x[i];
    s = lv_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[0] += (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void add_scalar_kernel(float* x, float v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += v;
}

}  // namespace

extern "C" int lv_sumsq_workspace_floats() { return 2 * NORM_BLOCKS; }

// out[0] (=|+=) sum(x[i]^2); ws: lv_sumsq_workspace_floats() floats.  Deterministic two-stage reduction.
extern "C" int lv_sumsq_f32(const float* x, long n, float* ws, float* out, int accumulate, void* stream) {
    if (!x || !ws || !out || n < 0) return LV_ERR_ARG;
    long nb = (n + 8191) / 8192;
    if (nb < 1) nb = 1;
    if (nb > NORM_BLOCKS) nb = NORM_BLOCKS;
    const int nblk = (int)nb;
    LV_LAUNCH(sumsq_stage1_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, x, n, ws, nblk);
    LV_LAUNCH(sumsq_stage2_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, out, accumulate, nblk);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_clip_coef_f32(const float* sumsq, float max_norm, float* coef, float* norm_out, void* stream) {
    if (!sumsq || !coef) return LV_ERR_ARG;
    LV_LAUNCH(clip_coef_kernel, dim3(1), dim3(64), 0, stream, sumsq, max_norm, coef, norm_out, TxnGate{nullptr, nullptr, nullptr, nullptr, nullptr});
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// lv_clip_coef_f32 + the transaction gate (see TxnGate): status1 / status2 / guard may be null; txn = device float[5] {void flag,
// steps committed, pending loss, rec, kl sums}; acc = device float[3] the committed running sums (may be null)
extern "C" int lv_clip_coef_txn_f32(const float* sumsq, float max_norm, float* coef, float* norm_out, const int* status1,
                                    const int* status2, const float* guard, float* txn, float* acc, void* stream) {
    if (!sumsq || !coef || !txn) return LV_ERR_ARG;
    LV_LAUNCH(clip_coef_kernel, dim3(1), dim3(64), 0, stream, sumsq, max_norm, coef, norm_out, TxnGate{status1, status2, guard, txn, acc});
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_txn_guard_f32(const int* status1, const int* status2, float* guard, void* stream) {
    if (!guard) return LV_ERR_ARG;
    LV_LAUNCH(txn_guard_kernel, dim3(1), dim3(64), 0, stream, status1, status2, guard);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

static inline unsigned lv_stream_grid(long n) {
    long b = (n + 1023) / 1024;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (unsigned)b;
}

extern "C" int lv_sgd_step_f32(float* p, float* g, long n, const float* lr_dev, const float* coef_dev,
                               int write_back_clipped, void* stream) {
    if (!p || !g || !lr_dev || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    LV_LAUNCH(sgd_kernel, dim3(lv_stream_grid(n)), dim3(256), 0, stream, p, g, n, lr_dev, coef_dev, write_back_clipped, (const float*)nullptr, (float*)nullptr, 0L);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// lv_sgd_step_f32 behind the transaction gate: a no-op while void_flag_dev[0] != 0 (txn[0] of lv_clip_*_txn_f32)
extern "C" int lv_sgd_step_txn_f32(float* p, float* g, long n, const float* lr_dev, const float* coef_dev,
                                   int write_back_clipped, const float* void_flag_dev, void* stream) {
    if (!p || !g || !lr_dev || !void_flag_dev || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    LV_LAUNCH(sgd_kernel, dim3(lv_stream_grid(n)), dim3(256), 0, stream, p, g, n, lr_dev, coef_dev, write_back_clipped, void_flag_dev, (float*)nullptr, 0L);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// x [T*Bsz][C] time-major (row t*Bsz + b) *= keep [Bsz][T][C] ? kscale : 0: the backward of nn.Dropout on the decoder LSTM's
// output (dec_lstm.py:106), applied to dO once with loads along C instead of inside the persistent BPTT
extern "C" int lv_keep_scale_f32(float* x, const uint8_t* keep, float kscale, int T, int Bsz, int C, void* stream) {
    if (!x || !keep) return LV_ERR_ARG;
    if (T < 0 || Bsz <= 0 || C <= 0) return LV_ERR_SHAPE;
    const long n = (long)T * Bsz * C;
    if (n == 0) return LV_OK;
    if (C % 4 == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)keep) & 3) == 0 && (long)T * Bsz < (1L << 31))
        LV_LAUNCH(keep_scale_v4_kernel, dim3((unsigned)lv_cdiv(n / 4, 256)), dim3(256), 0, stream, x, keep, kscale, T, Bsz, C / 4, n / 4);
    else
        LV_LAUNCH(keep_scale_kernel, dim3((unsigned)lv_cdiv(n, 256)), dim3(256), 0, stream, x, keep, kscale, T, Bsz, C, n);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// dst f32 [n] = bf16 src [n] * scale (src 8-byte aligned, dst 16-byte aligned): unpacks a bf16 gradient payload
extern "C" int lv_cvt_f32_bf16_scaled(const uint16_t* src, long n, float scale, float* dst, void* stream) {
    if (!src || !dst || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    if ((((uintptr_t)src) & 7) != 0 || (((uintptr_t)dst) & 15) != 0) return LV_ERR_ALIGN;
    LV_LAUNCH(cvt_f32_bf16_scaled_kernel, dim3((unsigned)lv_cdiv(lv_cdiv(n, 4), 256)), dim3(256), 0, stream, src, n, scale, dst);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_scale_f32(float* x, long n, const float* coef_dev, void* stream) {
    if (!x || !coef_dev || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    LV_LAUNCH(scale_kernel, dim3(lv_stream_grid(n)), dim3(256), 0, stream, x, n, coef_dev, (const float*)nullptr);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// lv_sgd_step_txn_f32 on (p, g) + lv_scale_txn_f32 on a second gradient buffer x2 (the one that is not stepped) in one launch
extern "C" int lv_sgd_step_scale_txn_f32(float* p, float* g, long n, const float* lr_dev, const float* coef_dev,
                                         int write_back_clipped, float* x2, long n2, const float* void_flag_dev, void* stream) {
    if (!p || !g || !lr_dev || !coef_dev || !void_flag_dev || !x2 || n < 0 || n2 < 0) return LV_ERR_ARG;
    if (n == 0 && n2 == 0) return LV_OK;
    LV_LAUNCH(sgd_kernel, dim3(lv_stream_grid(n > n2 ? n : n2)), dim3(256), 0, stream, p, g, n, lr_dev, coef_dev, write_back_clipped,
              void_flag_dev, x2, n2);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_scale_txn_f32(float* x, long n, const float* coef_dev, const float* void_flag_dev, void* stream) {
    if (!x || !coef_dev || !void_flag_dev || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    LV_LAUNCH(scale_kernel, dim3(lv_stream_grid(n)), dim3(256), 0, stream, x, n, coef_dev, void_flag_dev);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_adam_step_f32(float* p, float* g, float* m, float* v, long n, const float* lr_dev,
                                const float* coef_dev, const float* step_dev, float beta1, float beta2, float eps,
                                int write_back_clipped, void* stream) {
    if (!p || !g || !m || !v || !lr_dev || !step_dev || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    LV_LAUNCH(adam_kernel, dim3(lv_stream_grid(n)), dim3(256), 0, stream, p, g, m, v, n, lr_dev, coef_dev, step_dev,
              beta1, beta2, eps, write_back_clipped);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_sum_accum_f32(const float* x, long n, float* out_dev, void* stream) {
    if (!x || !out_dev || n < 0) return LV_ERR_ARG;
    LV_LAUNCH(sum_accum_kernel, dim3(1), dim3(256), 0, stream, x, n, out_dev);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_add_scalar_f32(float* x_dev, float v, void* stream) {
    if (!x_dev) return LV_ERR_ARG;
    LV_LAUNCH(add_scalar_kernel, dim3(1), dim3(64), 0, stream, x_dev, v);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// clip_grad_norm_ over two flat gradient buffers (encoder, decoder: the norm spans both) in two launches:
// one streaming pass over both buffers, then sum + norm + coefficient.  ws: lv_sumsq_workspace_floats() floats.
extern "C" int lv_clip_norm2_f32(const float* g1, long n1, const float* g2, long n2, float* ws, float max_norm,
                                 float* sumsq_dev, float* coef_dev, float* norm_dev, void* stream) {
    if (!g1 || !g2 || !ws || !coef_dev || n1 < 0 || n2 < 0) return LV_ERR_ARG;
    long nb1 = (n1 + 8191) / 8192, nb2 = (n2 + 8191) / 8192;
    if (nb1 < 1) nb1 = 1;
    if (nb1 > NORM_BLOCKS) nb1 = NORM_BLOCKS;
    if (nb2 < 1) nb2 = 1;
    if (nb2 > NORM_BLOCKS) nb2 = NORM_BLOCKS;
    LV_LAUNCH(sumsq2_stage1_kernel, dim3((unsigned)(nb1 + nb2)), dim3(256), 0, stream, g1, n1, (int)nb1, g2, n2, (int)nb2, ws);
    LV_LAUNCH(clip_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, (int)(nb1 + nb2), (const float*)nullptr, 0, max_norm,
              sumsq_dev, coef_dev, norm_dev, TxnGate{nullptr, nullptr, nullptr, nullptr, nullptr});
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// lv_clip_norm2_f32 + the transaction gate (see TxnGate / lv_clip_coef_txn_f32)
static int clip_norm2_txn(const float* g1, long n1, const float* g2, long n2, float* ws, const float* extra, int n_extra, float max_norm,
                          float* sumsq_dev, float* coef_dev, float* norm_dev, const int* status1, const int* status2,
                          const float* guard, float* txn, float* acc, void* stream) {
    if (!g1 || !g2 || !ws || !coef_dev || !txn || n1 < 0 || n2 < 0 || n_extra < 0 || (n_extra > 0 && !extra)) return LV_ERR_ARG;
    long nb1 = (n1 + 8191) / 8192, nb2 = (n2 + 8191) / 8192;
    if (nb1 < 1) nb1 = 1;
    if (nb1 > NORM_BLOCKS) nb1 = NORM_BLOCKS;
    if (nb2 < 1) nb2 = 1;
    if (nb2 > NORM_BLOCKS) nb2 = NORM_BLOCKS;
    LV_LAUNCH(sumsq2_stage1_kernel, dim3((unsigned)(nb1 + nb2)), dim3(256), 0, stream, g1, n1, (int)nb1, g2, n2, (int)nb2, ws);
    LV_LAUNCH(clip_finish_kernel, dim3(1), dim3(n_extra > 0 ? 1024 : 256), 0, stream, (const float*)ws, (int)(nb1 + nb2), extra, n_extra,
              max_norm, sumsq_dev, coef_dev, norm_dev, TxnGate{status1, status2, guard, txn, acc});
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_clip_norm2_txn_f32(const float* g1, long n1, const float* g2, long n2, float* ws, float max_norm,
                                     float* sumsq_dev, float* coef_dev, float* norm_dev, const int* status1, const int* status2,
                                     const float* guard, float* txn, float* acc, void* stream) {
    return clip_norm2_txn(g1, n1, g2, n2, ws, nullptr, 0, max_norm, sumsq_dev, coef_dev, norm_dev, status1, status2, guard, txn, acc, stream);
}

// The same where part of the gradient arrives as partial sums of squares its producers emitted (extra[0 .. n_extra): the
// vocabulary-sized tensors, see lv_gemm_b16_sumsq / lv_embed_scatter_full_sumsq_f32): g1 / g2 are then the REST of the two flat
// gradients, and the 164 MB of the three big tensors are not read a second time for the norm.
extern "C" int lv_clip_norm2_fold_txn_f32(const float* g1, long n1, const float* g2, long n2, float* ws, const float* extra, int n_extra,
                                          float max_norm, float* sumsq_dev, float* coef_dev, float* norm_dev, const int* status1,
                                          const int* status2, const float* guard, float* txn, float* acc, void* stream) {
    return clip_norm2_txn(g1, n1, g2, n2, ws, extra, n_extra, max_norm, sumsq_dev, coef_dev, norm_dev, status1, status2, guard, txn, acc, stream);
}
