// lv_conv.hip -- convolution / batch-norm / ELU / BCE pieces of the Omniglot VAE (ResNet encoder, PixelCNN decoder).
//
// Replaces nn.Conv2d / MaskedConv2d / nn.BatchNorm2d(train) / nn.ELU / nn.Sigmoid + BCE on the image hot path:
// ResNetEncoderV2.forward (modules/encoders/enc_resnet_v2.py:27-126), PixelCNNDecoderV2.forward /
// reconstruct_error (modules/decoders/dec_pixelcnn_v2.py:12-195), driven by image.py:300-327.
//
// Layout: activations are NHWC ([N*H*W pixels][C channels], channel fastest) so that a convolution is a GEMM over
// pixels: col[p][tap*Cin + c] (lv_im2col_f32) times a [Cout][taps*Cin] weight panel (lv_conv_pack_w_f32) on the MFMA
// GEMM (lv_gemm_*), 1x1 convolutions are plain GEMMs with no im2col at all.  Masked convolutions skip taps: the taps
// of a type-B k x k mask are a PREFIX of the raster order (rows above the centre, then the centre row up to and
// including the centre: 25 of 49 for 7x7), so the forward / data-gradient GEMMs simply use K = ntaps*Cin on the same
// im2col rows, while the weight gradient uses the full K (the reference keeps non-zero grads on masked taps and they
// enter clip_grad_norm_: SURVEY.md G5/G1).
// BatchNorm (train mode): deterministic two-stage per-channel statistics (f32 partials, f64 combine), normalise +
// residual add + ELU fused in one pass; backward = one reduction (dbeta, dgamma) + one apply pass, ELU' from the
// saved output.
#include "lv_device.h"

namespace {

// col[p][t*C + c] = x[n][ho*s + dh][wo*s + dw][c]  (0 outside); taps t = 0..nt-1 in raster order of the kh x kw window
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, long ldcol,
                                                     int N, int H, int W, int C, int Ho, int Wo, int kw, int pad,
                                                     int stride, int nt) {
    const long total = (long)N * Ho * Wo * nt * C;
    const long gs = (long)gridDim.x * 256;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += gs) {
        const int c = (int)(idx % C);
        const int t = (int)((idx / C) % nt);
        const long p = idx / ((long)C * nt);
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), n = (int)(p / ((long)Wo * Ho));
        const int h = ho * stride + t / kw - pad, w = wo * stride + t % kw - pad;
        float v = 0.f;
        if (h >= 0 && h < H && w >= 0 && w < W) v = x[(((long)n * H + h) * W + w) * C + c];
        col[p * ldcol + (long)t * C + c] = v;
    }
}

// same, four channels per thread (C % 4 == 0, 16-byte aligned rows): one index computation per float4
__global__ __launch_bounds__(256) void im2col_vec4_kernel(const float* __restrict__ x, float* __restrict__ col, long ldcol,
                                                          int N, int H, int W, int C, int Ho, int Wo, int kw, int pad,
                                                          int stride, int nt) {
    const int C4 = C >> 2;
    const long total = (long)N * Ho * Wo * nt * C4;
    const long gs = (long)gridDim.x * 256;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += gs) {
        const int c4 = (int)(idx % C4);
        const int t = (int)((idx / C4) % nt);
        const long p = idx / ((long)C4 * nt);
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), n = (int)(p / ((long)Wo * Ho));
        const int h = ho * stride + t / kw - pad, w = wo * stride + t % kw - pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h >= 0 && h < H && w >= 0 && w < W)
            v = *reinterpret_cast<const float4*>(x + (((long)n * H + h) * W + w) * C + 4 * c4);
        *reinterpret_cast<float4*>(col + p * ldcol + (long)t * C + 4 * c4) = v;
    }
}

// dx[n][h][w][c] (=|+=) sum_t dcol[p(h,w,t)][t*C + c] over the taps/outputs that read (h,w)
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, long ldcol, float* __restrict__ dx,
                                                     int N, int H, int W, int C, int Ho, int Wo, int kw, int pad,
                                                     int stride, int nt, int accumulate) {
    const long total = (long)N * H * W * C;
    const long gs = (long)gridDim.x * 256;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += gs) {
        const int c = (int)(idx % C);
        const int w = (int)((idx / C) % W), h = (int)((idx / ((long)C * W)) % H), n = (int)(idx / ((long)C * W * H));
        float s = 0.f;
        for (int t = 0; t < nt; ++t) {
            const int hn = h + pad - t / kw, wn = w + pad - t % kw;
            if (hn < 0 || wn < 0 || hn % stride != 0 || wn % stride != 0) continue;
            const int ho = hn / stride, wo = wn / stride;
            if (ho >= Ho || wo >= Wo) continue;
            s += dcol[(((long)n * Ho + ho) * Wo + wo) * ldcol + (long)t * C + c];
        }
        dx[idx] = accumulate ? dx[idx] + s : s;
    }
}

// wg[co][t][ci] = w[co][ci][t]   (reference weights are [Cout][Cin][kh][kw])
__global__ __launch_bounds__(256) void conv_pack_w_kernel(const float* __restrict__ w, float* __restrict__ wg,
                                                          int Cout, int Cin, int KK) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Cout * Cin * KK) return;
    const int ci = (int)(idx % Cin), t = (int)((idx / Cin) % KK), co = (int)(idx / ((long)Cin * KK));
    wg[idx] = w[((long)co * Cin + ci) * KK + t];
}

// dw[co][ci][t] = dwg[co][t][ci]
__global__ __launch_bounds__(256) void conv_unpack_dw_kernel(const float* __restrict__ dwg, float* __restrict__ dw,
                                                             int Cout, int Cin, int KK, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Cout * Cin * KK) return;
    const int t = (int)(idx % KK), ci = (int)((idx / KK) % Cin), co = (int)(idx / ((long)Cin * KK));
    const float v = dwg[((long)co * KK + t) * Cin + ci];
    dw[idx] = accumulate ? dw[idx] + v : v;
}

// in-place weight masking of MaskedConv2d.forward: w *= mask (dec_pixelcnn_v2.py:29)
__global__ __launch_bounds__(256) void mul_inplace_kernel(float* __restrict__ w, const float* __restrict__ m, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) w[i] *= m[i];
}

constexpr int BN_BLOCKS = 1024;

// Per-channel column reductions over a [P][C] matrix, stage 1: block blk reduces rows blk, blk+nblk, ... (in groups
// of 256/CT rows, CT = channels handled per pass) and writes partial[blk][q][c], q = 0,1.
// MODE 0 (forward stats):  q0 = sum x,            q1 = sum x*x
// MODE 1 (backward):       dv = dy * elu'(y) (act) written to dv_out; q0 = sum dv, q1 = sum dv * xhat
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ dy2, const float* __restrict__ dy3,
                                                        const float* __restrict__ dy4,
                                                        const float* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, int act,
                                                        float* __restrict__ dv_out, float* __restrict__ partial,
                                                        long P, int C, int nblk) {
    __shared__ float s0[256], s1[256];
    const int tid = (int)threadIdx.x;
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int CT = (C - c0) < 256 ? (C - c0) : 256;      // channels in this pass
        const int RPB = 256 / CT;                             // rows per block iteration (>= 1)
        const int c = c0 + tid % CT, rsub = tid / CT;
        float a0 = 0.f, a1 = 0.f;
        if (rsub < RPB) {
            float mu = 0.f, is = 0.f;
            if (MODE == 1) { mu = mean[c]; is = invstd[c]; }
            for (long r = (long)blockIdx.x * RPB + rsub; r < P; r += (long)nblk * RPB) {
                const long i = r * C + c;
                if (MODE == 0) {
                    const float v = x[i];
                    a0 += v; a1 += v * v;
                } else {
                    float g = dy[i];
                    if (dy2) g += dy2[i];
                    if (dy3) g += dy3[i];
                    if (dy4) g += dy4[i];
                    if (act) { const float yy = y[i]; g = yy > 0.f ? g : g * (yy + 1.f); }
                    dv_out[i] = g;
                    a0 += g; a1 += g * ((x[i] - mu) * is);
                }
            }
        }
        s0[tid] = a0; s1[tid] = a1;
        __syncthreads();
        if (tid < CT) {
            float t0 = 0.f, t1 = 0.f;
            for (int k = 0; k < RPB; ++k) { t0 += s0[tid + k * CT]; t1 += s1[tid + k * CT]; }
            partial[((long)blockIdx.x * 2 + 0) * C + c0 + tid] = t0;
            partial[((long)blockIdx.x * 2 + 1) * C + c0 + tid] = t1;
        }
        __syncthreads();
    }
}

// stage 2: one wave per channel sums the per-block partials (lanes stride over blocks, f64 shuffle reduction in a
// fixed order -> deterministic); 4 channels per workgroup
__device__ __forceinline__ void bn_partial_sums(const float* __restrict__ partial, int nblk, int C, int c, int lane,
                                                double& s, double& q) {
    s = 0.0; q = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        s += (double)partial[((long)b * 2 + 0) * C + c];
        q += (double)partial[((long)b * 2 + 1) * C + c];
    }
    s = lv_wave_sum(s);
    q = lv_wave_sum(q);
}

// forward: mean, invstd = 1/sqrt(var_biased + eps); running stats with momentum (unbiased var)
__global__ __launch_bounds__(256) void bn_finish_fwd_kernel(const float* __restrict__ partial, int nblk, long P, int C,
                                                            float eps, float momentum, float* __restrict__ mean,
                                                            float* __restrict__ invstd, float* __restrict__ run_mean,
                                                            float* __restrict__ run_var) {
    const int lane = (int)threadIdx.x & 63;
    const int c = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int cc = c < C ? c : C - 1;          // keep every wave in the shuffles
    double s, q;
    bn_partial_sums(partial, nblk, C, cc, lane, s, q);
    if (c >= C || lane != 0) return;
    const double m = s / (double)P;
    double var = q / (double)P - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
        run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * m);
        run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * unb);
    }
}

// backward: this layer's sums into (dgl, dbl) and (=|+=) into the parameter gradients
__global__ __launch_bounds__(256) void bn_finish_bwd_kernel(const float* __restrict__ partial, int nblk, int C,
                                                            float* __restrict__ dgl, float* __restrict__ dbl,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int accumulate) {
    const int lane = (int)threadIdx.x & 63;
    const int c = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int cc = c < C ? c : C - 1;
    double s, q;
    bn_partial_sums(partial, nblk, C, cc, lane, s, q);
    if (c >= C || lane != 0) return;
    dbl[c] = (float)s;
    dgl[c] = (float)q;
    dbeta[c] = (float)(s + (accumulate ? (double)dbeta[c] : 0.0));
    dgamma[c] = (float)(q + (accumulate ? (double)dgamma[c] : 0.0));
}

// y = act((x - mean) * invstd * gamma + beta (+ res))
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ res,
                                                           int act, float* __restrict__ y, long n, int C) {
    const long gs = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) {
        const int c = (int)(i % C);
        float v = (x[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
        if (res) v += res[i];
        if (act) v = v > 0.f ? v : expm1f(v);
        y[i] = v;
    }
}

// dx = gamma*invstd * (dv - dbeta/P - xhat * dgamma/P); dbeta/dgamma are THIS layer's sums (dbl, dgl), not accumulated
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dv,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ dgl,
                                                           const float* __restrict__ dbl, float* __restrict__ dx,
                                                           long n, int C, float invP) {
    const long gs = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) {
        const int c = (int)(i % C);
        const float is = invstd[c];
        const float xh = (x[i] - mean[c]) * is;
        dx[i] = gamma[c] * is * (dv[i] - dbl[c] * invP - xh * dgl[c] * invP);
    }
}

// ---- vector path (C a power of two in [16, 256]; every BatchNorm of the two networks) -------------------------------------
// Two launches per direction instead of three: the reduction writes <= BN_V4_BLOCKS per-block partials with float4 loads, and
// every workgroup of the apply kernel re-derives the per-channel totals from them (f64, fixed order: <= 64 KB of L2 reads per
// workgroup) instead of waiting for a separate one-workgroup "finish" launch (5-6 us each, 162 of them per Omniglot step).
constexpr int BN_V4_BLOCKS = 256;

template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_v4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ dy2, const float* __restrict__ dy3,
                                                           const float* __restrict__ dy4,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, int act,
                                                           float* __restrict__ dv_out, float* __restrict__ partial,
                                                           long P, int C, int nblk) {
    __shared__ __attribute__((aligned(16))) float sred[4][2][256];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int CT4 = C >> 2, RPB = 256 / CT4;
    const int c4 = tid & (CT4 - 1), rsub = tid / CT4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, mu = a0, is = a0;
    if (MODE == 1) {
        mu = *reinterpret_cast<const float4*>(mean + 4 * c4);
        is = *reinterpret_cast<const float4*>(invstd + 4 * c4);
    }
    // 4 rows per trip, their loads issued together (the trip count is data-dependent: the compiler will not batch on its own)
    const long stride = (long)nblk * RPB;
    for (long r = (long)blockIdx.x * RPB + rsub; r < P; r += 4 * stride) {
        float4 xv[4], gv[4], yv[4], g2[4], g3[4], g4[4];
        long idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long rr = r + u * stride;
            idx[u] = (rr < P ? rr : r) * C + 4 * c4;
            xv[u] = *reinterpret_cast<const float4*>(x + idx[u]);
            if (MODE == 1) gv[u] = *reinterpret_cast<const float4*>(dy + idx[u]);
        }
        if (MODE == 1) {
            // The incoming gradient arrives as up to four summands (an activation with several consumers), added in the order
            // given.  ALL loads of the trip are issued before the first addition: one uniform branch per summand around its four
            // loads (a load and its `+=` inside a per-row `if (dy2)` made every summand of every row its own memory round trip).
            if (dy2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) g2[u] = *reinterpret_cast<const float4*>(dy2 + idx[u]);
            }
            if (dy3) {
#pragma unroll
                for (int u = 0; u < 4; ++u) g3[u] = *reinterpret_cast<const float4*>(dy3 + idx[u]);
            }
            if (dy4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) g4[u] = *reinterpret_cast<const float4*>(dy4 + idx[u]);
            }
            if (act) {
#pragma unroll
                for (int u = 0; u < 4; ++u) yv[u] = *reinterpret_cast<const float4*>(y + idx[u]);
            }
            if (dy2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { gv[u].x += g2[u].x; gv[u].y += g2[u].y; gv[u].z += g2[u].z; gv[u].w += g2[u].w; }
            }
            if (dy3) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { gv[u].x += g3[u].x; gv[u].y += g3[u].y; gv[u].z += g3[u].z; gv[u].w += g3[u].w; }
            }
            if (dy4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { gv[u].x += g4[u].x; gv[u].y += g4[u].y; gv[u].z += g4[u].z; gv[u].w += g4[u].w; }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long rr = r + u * stride;
            if (rr >= P) break;
            const long i = rr * C + 4 * c4;
            if (MODE == 0) {
                const float4 v = xv[u];
                a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
                a1.x += v.x * v.x; a1.y += v.y * v.y; a1.z += v.z * v.z; a1.w += v.w * v.w;
            } else {
                float4 g = gv[u];
                if (act) {
                    const float4 yy = yv[u];
                    g.x = yy.x > 0.f ? g.x : g.x * (yy.x + 1.f);
                    g.y = yy.y > 0.f ? g.y : g.y * (yy.y + 1.f);
                    g.z = yy.z > 0.f ? g.z : g.z * (yy.z + 1.f);
                    g.w = yy.w > 0.f ? g.w : g.w * (yy.w + 1.f);
                }
                *reinterpret_cast<float4*>(dv_out + i) = g;
                a0.x += g.x; a0.y += g.y; a0.z += g.z; a0.w += g.w;
                a1.x += g.x * ((xv[u].x - mu.x) * is.x); a1.y += g.y * ((xv[u].y - mu.y) * is.y);
                a1.z += g.z * ((xv[u].z - mu.z) * is.z); a1.w += g.w * ((xv[u].w - mu.w) * is.w);
            }
        }
    }
    // lanes of a wave with equal c4 (lane bits >= log2 CT4), then the 4 waves through LDS: a fixed order
    for (int off = CT4; off < 64; off <<= 1) {
        a0.x += __shfl_xor(a0.x, off, 64); a0.y += __shfl_xor(a0.y, off, 64);
        a0.z += __shfl_xor(a0.z, off, 64); a0.w += __shfl_xor(a0.w, off, 64);
        a1.x += __shfl_xor(a1.x, off, 64); a1.y += __shfl_xor(a1.y, off, 64);
        a1.z += __shfl_xor(a1.z, off, 64); a1.w += __shfl_xor(a1.w, off, 64);
    }
    if (lane < CT4) {
        *reinterpret_cast<float4*>(&sred[wv][0][4 * lane]) = a0;
        *reinterpret_cast<float4*>(&sred[wv][1][4 * lane]) = a1;
    }
    __syncthreads();
    // CT4 == 64: each wave's 64 lanes are 64 distinct channel groups and rsub == wv; CT4 < 64: lanes < CT4 hold wave totals
    for (int t = tid; t < 2 * C; t += 256) {
        const int q = t / C, c = t % C;
        partial[((long)blockIdx.x * 2 + q) * C + c] = ((sred[0][q][c] + sred[1][q][c]) + sred[2][q][c]) + sred[3][q][c];
    }
}

// per-(q, c) totals of the per-block partials, all 256 threads cooperating: thread t sums float4 group t % (C/2) over blocks
// t / (C/2), + 256/(C/2), ... (16 independent loads in flight per batch: a runtime-length load -> add loop would pay one L2
// round trip per block), f64, fixed order; tot: 2*C doubles, scratch: 1024 doubles (LDS)
#ifndef LV_BN_TOT_BATCH
#define LV_BN_TOT_BATCH 8               // partial rows a thread has in flight per round trip.  Round 6 A/B on the Omniglot step (hipGraph,
                                        // profiles/r06i_omniglot_bn_prologue_ab.txt): 4: 9 450, 6: 9 380, 8: 9 640, 12: 9 480, 16 (rounds 3-5): 9 500,
                                        // 32: 9 250 img/s -- every workgroup of every apply launch re-reads ALL partial rows (27 MB per launch at
                                        // 350 rows x 306 workgroups, more than the activation itself), and the tail batch's clamped dummy loads are
                                        // part of that traffic: the batch length trades round trips against wasted loads.  Same sums, same order
#endif
#ifndef LV_BN_TOT_CLAMP_OWN
#define LV_BN_TOT_CLAMP_OWN 0           // measurement knob: rows beyond nblk re-read the thread's OWN first row instead of row 0
#endif
__device__ __forceinline__ void bn_block_totals(const float* __restrict__ partial, int nblk, int C, double* tot, double* scratch) {
    constexpr int NB = LV_BN_TOT_BATCH;
    const int tid = (int)threadIdx.x, npair = 2 * C, NF4 = C >> 1, nsub = 256 / NF4;
    const int pg = tid & (NF4 - 1), bsub = tid / NF4;
    const float4* p4 = reinterpret_cast<const float4*>(partial) + pg;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = bsub;
#ifdef LV_BN_TOT_WHATIF               // measurement knob: no statistics prologue at all (garbage results; the time is the point)
    if (tid < npair) tot[tid] = tid < C ? 0.0 : (double)nblk;
    __syncthreads();
    return;
#endif
    for (; b + (NB - 1) * nsub < nblk; b += NB * nsub) {
        float4 v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) v[u] = p4[(long)(b + u * nsub) * NF4];
#pragma unroll
        for (int u = 0; u < NB; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
    }
    {
        float4 v[NB];
        const int spare = LV_BN_TOT_CLAMP_OWN ? (bsub < nblk ? bsub : 0) : 0;
#pragma unroll
        for (int u = 0; u < NB; ++u) {                 // unconditional loads from clamped indices (a load behind a condition is
            const int bb = b + u * nsub;                // waited for right behind its issue: NB round trips instead of one)
            v[u] = p4[(long)(bb < nblk ? bb : spare) * NF4];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (b + u * nsub < nblk) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
        }
    }
    double* sc = scratch + (long)bsub * npair + 4 * pg;
    sc[0] = s0; sc[1] = s1; sc[2] = s2; sc[3] = s3;
    __syncthreads();
    for (int t = tid; t < npair; t += 256) {
        double acc = 0.0;
#ifdef LV_BN_TOT_L2_UNROLL            // measurement knob: the second level reads its nsub values before it adds them (same order)
        if (nsub == 16) {
            double v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = scratch[(long)k * npair + t];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += v[k];
        } else if (nsub == 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = scratch[(long)k * npair + t];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        } else
#endif
        for (int k = 0; k < nsub; ++k) acc += scratch[(long)k * npair + t];
        tot[t] = acc;
    }
    __syncthreads();
}

template <int BN_V4_ITEMS>
__global__ __launch_bounds__(256) void bn_apply_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ partial, int nblk,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ res, int act, float* __restrict__ y,
                                                              float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                              float* __restrict__ run_mean, float* __restrict__ run_var,
                                                              long P, int C, float eps, float momentum) {
    __shared__ double tot[512], scratch[1024];
    __shared__ __attribute__((aligned(16))) float smu[256], sis[256], sga[256], sbe[256];
    const int tid = (int)threadIdx.x;
    // this thread's BN_V4_ITEMS float4 first: they do not depend on the statistics and stay in flight under the totals
    const long n4 = P * (C >> 2);
    const int CT4 = C >> 2;
    const long i0 = (long)blockIdx.x * (256 * BN_V4_ITEMS) + tid;
    float4 xs[BN_V4_ITEMS], rs[BN_V4_ITEMS];
#pragma unroll
    for (int u = 0; u < BN_V4_ITEMS; ++u) {
        const long i = i0 + 256 * u;
        const long ii = i < n4 ? i : 0;
        xs[u] = *reinterpret_cast<const float4*>(x + 4 * ii);
        if (res) rs[u] = *reinterpret_cast<const float4*>(res + 4 * ii);
    }
    bn_block_totals(partial, nblk, C, tot, scratch);
    if (tid < C) {
        const double m = tot[tid] / (double)P;
        double var = tot[C + tid] / (double)P - m * m;
        if (var < 0.0) var = 0.0;
        const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)eps));
        smu[tid] = mf; sis[tid] = isf; sga[tid] = gamma[tid]; sbe[tid] = beta[tid];
        if (blockIdx.x == 0) {
            mean_out[tid] = mf;
            invstd_out[tid] = isf;
            if (run_mean) {
                const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
                run_mean[tid] = (float)((1.0 - momentum) * (double)run_mean[tid] + momentum * m);
                run_var[tid] = (float)((1.0 - momentum) * (double)run_var[tid] + momentum * unb);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BN_V4_ITEMS; ++u) {
        const long i = i0 + 256 * u;
        if (i >= n4) break;
        const int c = 4 * (int)(i & (CT4 - 1));
        const float4 xv = xs[u];
        const float4 m4 = *reinterpret_cast<const float4*>(&smu[c]), i4 = *reinterpret_cast<const float4*>(&sis[c]);
        const float4 g4 = *reinterpret_cast<const float4*>(&sga[c]), b4 = *reinterpret_cast<const float4*>(&sbe[c]);
        float4 v;
        v.x = (xv.x - m4.x) * i4.x * g4.x + b4.x; v.y = (xv.y - m4.y) * i4.y * g4.y + b4.y;
        v.z = (xv.z - m4.z) * i4.z * g4.z + b4.z; v.w = (xv.w - m4.w) * i4.w * g4.w + b4.w;
        if (res) { v.x += rs[u].x; v.y += rs[u].y; v.z += rs[u].z; v.w += rs[u].w; }
        if (act) {
            v.x = v.x > 0.f ? v.x : expm1f(v.x); v.y = v.y > 0.f ? v.y : expm1f(v.y);
            v.z = v.z > 0.f ? v.z : expm1f(v.z); v.w = v.w > 0.f ? v.w : expm1f(v.w);
        }
        *reinterpret_cast<float4*>(y + 4 * i) = v;
    }
}

template <int BN_V4_ITEMS>
__global__ __launch_bounds__(256) void bn_apply_bwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ dv,
                                                              const float* __restrict__ partial, int nblk,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate, float* __restrict__ dx,
                                                              long P, int C, float invP) {
    __shared__ double tot[512], scratch[1024];
    __shared__ __attribute__((aligned(16))) float smu[256], sis[256], sga[256], sdg[256], sdb[256];
    const int tid = (int)threadIdx.x;
    const long n4 = P * (C >> 2);
    const int CT4 = C >> 2;
    const long i0 = (long)blockIdx.x * (256 * BN_V4_ITEMS) + tid;
    float4 xs[BN_V4_ITEMS], ds[BN_V4_ITEMS];
#pragma unroll
    for (int u = 0; u < BN_V4_ITEMS; ++u) {
        const long i = i0 + 256 * u;
        const long ii = i < n4 ? i : 0;
        xs[u] = *reinterpret_cast<const float4*>(x + 4 * ii);
        ds[u] = *reinterpret_cast<const float4*>(dv + 4 * ii);
    }
    bn_block_totals(partial, nblk, C, tot, scratch);
    if (tid < C) {
        const double s = tot[tid], q = tot[C + tid];
        smu[tid] = mean[tid]; sis[tid] = invstd[tid]; sga[tid] = gamma[tid];
        sdb[tid] = (float)s; sdg[tid] = (float)q;
        if (blockIdx.x == 0) {
            dbeta[tid] = (float)(s + (accumulate ? (double)dbeta[tid] : 0.0));
            dgamma[tid] = (float)(q + (accumulate ? (double)dgamma[tid] : 0.0));
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BN_V4_ITEMS; ++u) {
        const long i = i0 + 256 * u;
        if (i >= n4) break;
        const int c = 4 * (int)(i & (CT4 - 1));
        const float4 xv = xs[u], d4 = ds[u];
        const float4 m4 = *reinterpret_cast<const float4*>(&smu[c]), i4 = *reinterpret_cast<const float4*>(&sis[c]);
        const float4 g4 = *reinterpret_cast<const float4*>(&sga[c]);
        const float4 dg = *reinterpret_cast<const float4*>(&sdg[c]), db = *reinterpret_cast<const float4*>(&sdb[c]);
        float4 o;
        o.x = g4.x * i4.x * (d4.x - db.x * invP - ((xv.x - m4.x) * i4.x) * dg.x * invP);
        o.y = g4.y * i4.y * (d4.y - db.y * invP - ((xv.y - m4.y) * i4.y) * dg.y * invP);
        o.z = g4.z * i4.z * (d4.z - db.z * invP - ((xv.z - m4.z) * i4.z) * dg.z * invP);
        o.w = g4.w * i4.w * (d4.w - db.w * invP - ((xv.w - m4.w) * i4.w) * dg.w * invP);
        *reinterpret_cast<float4*>(dx + 4 * i) = o;
    }
}

static inline bool bn_v4_ok(int C) { return C >= 16 && C <= 256 && (C & (C - 1)) == 0; }
static inline int bn_v4_blocks(long P, int C) {
    const int rpb = 256 / (C >> 2);
    long nb = (P + 4L * rpb - 1) / (4L * rpb);          // >= 4 rows per thread
    return (int)(nb < 1 ? 1 : nb > BN_V4_BLOCKS ? BN_V4_BLOCKS : nb);
}
// float4 per thread of the apply kernels: every workgroup re-reads all the partials (nblk * 2C floats), so the wider layers
// use fewer, fatter workgroups
#ifndef LV_BN_ITEMS_NARROW
#define LV_BN_ITEMS_NARROW 4            // float4 per thread of the apply kernels at C < 64 / C >= 64 (2, 4, 8 or 16): measurement knobs
#endif
#ifndef LV_BN_ITEMS_WIDE
#define LV_BN_ITEMS_WIDE 8
#endif
static inline int bn_v4_items(int C) { return C >= 64 ? LV_BN_ITEMS_WIDE : LV_BN_ITEMS_NARROW; }
// launch an apply kernel template on its item count
#define BN_APPLY_LAUNCH(KERN, n4, C, stream, ...)                                                                                     \
    do {                                                                                                                              \
        const int it_ = bn_v4_items(C);                                                                                               \
        if (it_ == 16) LV_LAUNCH(KERN<16>, dim3(bn_v4_apply_grid((n4), 16)), dim3(256), 0, stream, __VA_ARGS__);                      \
        else if (it_ == 8) LV_LAUNCH(KERN<8>, dim3(bn_v4_apply_grid((n4), 8)), dim3(256), 0, stream, __VA_ARGS__);                    \
        else if (it_ == 2) LV_LAUNCH(KERN<2>, dim3(bn_v4_apply_grid((n4), 2)), dim3(256), 0, stream, __VA_ARGS__);                    \
        else LV_LAUNCH(KERN<4>, dim3(bn_v4_apply_grid((n4), 4)), dim3(256), 0, stream, __VA_ARGS__);                                  \
    } while (0)
static inline unsigned bn_v4_apply_grid(long n4, int items) { return (unsigned)lv_cdiv(n4, 256L * items); }

// rec[b] = -sum_pix x*log(p+eps) + (1-x)*log(1-p+eps), p = sigmoid(logit); one workgroup per image
__global__ __launch_bounds__(256) void sigmoid_bce_fwd_kernel(const float* __restrict__ logit, const float* __restrict__ x,
                                                              float* __restrict__ rec, int npix, float eps) {
    __shared__ float red[4];
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < npix; i += 256) {
        const float p = 1.f / (1.f + expf(-logit[(long)b * npix + i]));
        const float xv = x[(long)b * npix + i];
        s += logf(p + eps) * xv + logf(1.f - p + eps) * (1.f - xv);
    }
    s = lv_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) rec[b] = -((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void sigmoid_bce_bwd_kernel(const float* __restrict__ logit, const float* __restrict__ x,
                                                              const float* __restrict__ drec, float* __restrict__ dlogit,
                                                              int npix, long n, float eps) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = (int)(i / npix);
    const float p = 1.f / (1.f + expf(-logit[i]));
    const float xv = x[i];
    const float dp = -(xv / (p + eps) - (1.f - xv) / (1.f - p + eps));
    dlogit[i] = drec[b] * dp * p * (1.f - p);
}

// decoder input: in5[b][pix][0] = x[b][pix]; in5[b][pix][1+c] = zt[b][c*npix + pix]   (dec_pixelcnn_v2.py:178-186)
__global__ __launch_bounds__(256) void dec_input_fwd_kernel(const float* __restrict__ x, const float* __restrict__ zt,
                                                            float* __restrict__ in5, int npix, int fm, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*npix*(1+fm)
    if (i >= n) return;
    const int C = 1 + fm;
    const int c = (int)(i % C);
    const long p = i / C;
    const int pix = (int)(p % npix);
    const long b = p / npix;
    in5[i] = c == 0 ? x[p] : zt[(b * fm + (c - 1)) * npix + pix];
}

__global__ __launch_bounds__(256) void dec_input_bwd_kernel(const float* __restrict__ din5, float* __restrict__ dzt,
                                                            int npix, int fm, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;     // over B*fm*npix
    if (i >= n) return;
    const int pix = (int)(i % npix);
    const int c = (int)((i / npix) % fm);
    const long b = i / ((long)npix * fm);
    dzt[i] = din5[(b * npix + pix) * (1 + fm) + 1 + c];
}

static inline unsigned conv_grid(long n) {
    long b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (unsigned)b;
}

// nn.BatchNorm2d in EVAL mode (+ residual add) (+ nn.ELU): y = act((x - running_mean) / sqrt(running_var + eps) * gamma + beta (+ res)),
// one streaming pass; four channels per thread when C % 4 == 0 and the buffers are 16-byte aligned.  mean_out / invstd_out (may be
// NULL) receive the statistics that were applied, in the layout the backward kernels read.
template <bool V4>
__global__ __launch_bounds__(256) void bn_eval_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ rmean,
                                                      const float* __restrict__ rvar, float eps, const float* __restrict__ res, int act,
                                                      float* __restrict__ y, float* __restrict__ mean_out,
                                                      float* __restrict__ invstd_out, long n, int C) {
    const long gs = (long)gridDim.x * 256;
    if (blockIdx.x == 0 && mean_out && invstd_out)
        for (int c = (int)threadIdx.x; c < C; c += 256) { mean_out[c] = rmean[c]; invstd_out[c] = 1.0f / sqrtf(rvar[c] + eps); }
    if (V4) {
        const int C4 = C >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += gs) {
            const int c = 4 * (int)(i % C4);
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            float v[4] = {xv.x, xv.y, xv.z, xv.w};
            float r[4] = {0.f, 0.f, 0.f, 0.f};
            if (res) { const float4 rv = reinterpret_cast<const float4*>(res)[i]; r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float o = (v[e] - rmean[c + e]) * (1.0f / sqrtf(rvar[c + e] + eps)) * gamma[c + e] + beta[c + e];
                if (res) o += r[e];
                if (act) o = o > 0.f ? o : expm1f(o);
                v[e] = o;
            }
            reinterpret_cast<float4*>(y)[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += gs) {
            const int c = (int)(i % C);
            float o = (x[i] - rmean[c]) * (1.0f / sqrtf(rvar[c] + eps)) * gamma[c] + beta[c];
            if (res) o += res[i];
            if (act) o = o > 0.f ? o : expm1f(o);
            y[i] = o;
        }
    }
}

}  // namespace

// col[N*Ho*Wo][ldcol] <- the first `ntaps` raster-order taps of a kh x kw window (stride, pad) of NHWC x
extern "C" int lv_im2col_f32(const float* x, float* col, long ldcol, int N, int H, int W, int C, int Ho, int Wo,
                             int kh, int kw, int pad, int stride, int ntaps, void* stream) {
    if (!x || !col) return LV_ERR_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return LV_ERR_SHAPE;
    if (ntaps <= 0 || ntaps > kh * kw || ldcol < (long)ntaps * C) return LV_ERR_SHAPE;
    if (C % 4 == 0 && ldcol % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)col)) & 15) == 0)
        LV_LAUNCH(im2col_vec4_kernel, dim3(conv_grid((long)N * Ho * Wo * ntaps * (C / 4))), dim3(256), 0, stream, x, col, ldcol, N, H, W,
                  C, Ho, Wo, kw, pad, stride, ntaps);
    else
        LV_LAUNCH(im2col_kernel, dim3(conv_grid((long)N * Ho * Wo * ntaps * C)), dim3(256), 0, stream, x, col, ldcol, N, H, W, C, Ho, Wo,
                  kw, pad, stride, ntaps);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_col2im_f32(const float* dcol, long ldcol, float* dx, int N, int H, int W, int C, int Ho, int Wo,
                             int kh, int kw, int pad, int stride, int ntaps, int accumulate, void* stream) {
    if (!dcol || !dx) return LV_ERR_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return LV_ERR_SHAPE;
    if (ntaps <= 0 || ntaps > kh * kw || ldcol < (long)ntaps * C) return LV_ERR_SHAPE;
    LV_LAUNCH(col2im_kernel, dim3(conv_grid((long)N * H * W * C)), dim3(256), 0, stream, dcol, ldcol, dx, N, H, W, C, Ho, Wo,
              kw, pad, stride, ntaps, accumulate);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_conv_pack_w_f32(const float* w, float* wg, int Cout, int Cin, int KK, void* stream) {
    if (!w || !wg || Cout <= 0 || Cin <= 0 || KK <= 0) return LV_ERR_ARG;
    LV_LAUNCH(conv_pack_w_kernel, dim3((unsigned)lv_cdiv((long)Cout * Cin * KK, 256)), dim3(256), 0, stream, w, wg, Cout, Cin, KK);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_conv_unpack_dw_f32(const float* dwg, float* dw, int Cout, int Cin, int KK, int accumulate, void* stream) {
    if (!dwg || !dw || Cout <= 0 || Cin <= 0 || KK <= 0) return LV_ERR_ARG;
    LV_LAUNCH(conv_unpack_dw_kernel, dim3((unsigned)lv_cdiv((long)Cout * Cin * KK, 256)), dim3(256), 0, stream, dwg, dw, Cout, Cin, KK,
              accumulate);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_mul_inplace_f32(float* w, const float* m, long n, void* stream) {
    if (!w || !m || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    LV_LAUNCH(mul_inplace_kernel, dim3((unsigned)lv_cdiv(n, 256)), dim3(256), 0, stream, w, m, n);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_bn_workspace_floats(int C) { return BN_BLOCKS * 2 * (C > 0 ? C : 1); }

// nn.BatchNorm2d with the module in eval mode (the evaluation passes of image.py:96-187 and PixelCNN sampling,
// dec_pixelcnn_v2.py:201-232): running statistics, nothing is updated.
extern "C" int lv_bn_eval_f32(const float* x, const float* gamma, const float* beta, const float* run_mean, const float* run_var,
                              float eps, const float* res, int act_elu, float* y, float* mean_out, float* invstd_out, long P, int C,
                              void* stream) {
    if (!x || !gamma || !beta || !run_mean || !run_var || !y) return LV_ERR_ARG;
    if (P <= 0 || C <= 0) return LV_ERR_SHAPE;
    const long n = P * C;
    const bool v4 = (C % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0);
    if (v4) LV_LAUNCH(bn_eval_kernel<true>, dim3(conv_grid(n >> 2)), dim3(256), 0, stream, x, gamma, beta, run_mean, run_var, eps, res,
                      act_elu, y, mean_out, invstd_out, n, C);
    else LV_LAUNCH(bn_eval_kernel<false>, dim3(conv_grid(n)), dim3(256), 0, stream, x, gamma, beta, run_mean, run_var, eps, res, act_elu,
                   y, mean_out, invstd_out, n, C);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// BatchNorm2d (train) forward over x [P][C]: batch stats -> mean/invstd (saved), running stats (momentum, unbiased var),
// y = act(xhat*gamma + beta (+ res)).  ws: lv_bn_workspace_floats(C).
extern "C" int lv_bn_fwd_f32(const float* x, const float* gamma, const float* beta, const float* res, int act_elu,
                             float* y, float* mean, float* invstd, float* run_mean, float* run_var,
                             float eps, float momentum, float* ws, long P, int C, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !invstd || !ws) return LV_ERR_ARG;
    if (P <= 0 || C <= 0) return LV_ERR_SHAPE;
    if (bn_v4_ok(C) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0) {
        const int nb = bn_v4_blocks(P, C);
        LV_LAUNCH((bn_reduce_v4_kernel<0>), dim3((unsigned)nb), dim3(256), 0, stream, x, (const float*)nullptr, (const float*)nullptr,
                  (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0,
                  (float*)nullptr, ws, P, C, nb);
        BN_APPLY_LAUNCH(bn_apply_fwd_v4_kernel, P * (C >> 2), C, stream, x, (const float*)ws, nb, gamma, beta, res, act_elu, y, mean, invstd, run_mean, run_var, P, C, eps, momentum);
        LV_CHECK_LAUNCH();
        return LV_OK;
    }
    int nblk = (int)((P + 3) / 4);          // few rows per block: the rows of a block are read one after the other
    if (nblk > BN_BLOCKS) nblk = BN_BLOCKS;
    LV_LAUNCH((bn_reduce_kernel<0>), dim3((unsigned)nblk), dim3(256), 0, stream, x, (const float*)nullptr, (const float*)nullptr,
              (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0,
              (float*)nullptr, ws, P, C, nblk);
    LV_LAUNCH(bn_finish_fwd_kernel, dim3((unsigned)lv_cdiv(C, 4)), dim3(256), 0, stream, (const float*)ws, nblk, P, C, eps, momentum,
              mean, invstd, run_mean, run_var);
    LV_LAUNCH(bn_apply_fwd_kernel, dim3(conv_grid(P * C)), dim3(256), 0, stream, x, (const float*)mean, (const float*)invstd, gamma,
              beta, res, act_elu, y, P * C, C);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// The same forward with the stage-1 partials already written by the producing convolution (lv_conv32_bnstat_f32 /
// lv_conv1x1_bnstat_f32): partial [nblk][2][C], nblk <= 512.  C must take the vector path (power of two in [16, 256]).
extern "C" int lv_bn_fwd_partials_f32(const float* x, const float* gamma, const float* beta, const float* res, int act_elu,
                                      float* y, float* mean, float* invstd, float* run_mean, float* run_var, float eps,
                                      float momentum, const float* partial, int nblk, long P, int C, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !invstd || !partial) return LV_ERR_ARG;
    if (P <= 0 || C <= 0 || nblk <= 0 || nblk > BN_BLOCKS) return LV_ERR_SHAPE;
    if (!bn_v4_ok(C)) return LV_ERR_UNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res | (uintptr_t)partial) & 15) != 0) return LV_ERR_ALIGN;
    BN_APPLY_LAUNCH(bn_apply_fwd_v4_kernel, P * (C >> 2), C, stream, x, partial, nblk, gamma, beta, res, act_elu, y, mean, invstd, run_mean, run_var, P, C, eps, momentum);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// BatchNorm2d (train) backward.  dy = grad wrt y (post-activation); y = saved output (for ELU').  Writes
// dv = dy*elu'(y) (also the gradient of the residual input), dgamma/dbeta (=|+=), dx.
// ws: lv_bn_workspace_floats(C) + 2*C floats.
extern "C" int lv_bn_bwd4_f32(const float* x, const float* dy, const float* dy2, const float* dy3, const float* dy4, const float* y, const float* mean, const float* invstd,
                             const float* gamma, int act_elu, float* dv, float* dx, float* dgamma, float* dbeta,
                             int accumulate_param_grads, float* ws, long P, int C, void* stream) {
    if (!x || !dy || !mean || !invstd || !gamma || !dv || !dx || !dgamma || !dbeta || !ws) return LV_ERR_ARG;
    if (act_elu && !y) return LV_ERR_ARG;
    if ((dy3 && !dy2) || (dy4 && !dy3)) return LV_ERR_ARG;       // summands are given in order
    if (P <= 0 || C <= 0) return LV_ERR_SHAPE;
    if (bn_v4_ok(C) && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dy2 | (uintptr_t)dy3 | (uintptr_t)dy4 | (uintptr_t)y | (uintptr_t)dv | (uintptr_t)dx) & 15) == 0) {
        const int nb = bn_v4_blocks(P, C);
        LV_LAUNCH((bn_reduce_v4_kernel<1>), dim3((unsigned)nb), dim3(256), 0, stream, x, dy, dy2, dy3, dy4, y, mean, invstd, act_elu, dv, ws, P, C, nb);
        BN_APPLY_LAUNCH(bn_apply_bwd_v4_kernel, P * (C >> 2), C, stream, x, (const float*)dv, (const float*)ws, nb, mean, invstd, gamma, dgamma, dbeta, accumulate_param_grads, dx, P, C, 1.0f / (float)P);
        LV_CHECK_LAUNCH();
        return LV_OK;
    }
    int nblk = (int)((P + 3) / 4);
    if (nblk > BN_BLOCKS) nblk = BN_BLOCKS;
    float* dgl = ws + (long)BN_BLOCKS * 2 * C;
    float* dbl = dgl + C;
    LV_LAUNCH((bn_reduce_kernel<1>), dim3((unsigned)nblk), dim3(256), 0, stream, x, dy, dy2, dy3, dy4, y, mean, invstd, act_elu, dv, ws, P, C, nblk);
    LV_LAUNCH(bn_finish_bwd_kernel, dim3((unsigned)lv_cdiv(C, 4)), dim3(256), 0, stream, (const float*)ws, nblk, C, dgl, dbl,
              dgamma, dbeta, accumulate_param_grads);
    LV_LAUNCH(bn_apply_bwd_kernel, dim3(conv_grid(P * C)), dim3(256), 0, stream, x, (const float*)dv, mean, invstd, gamma,
              (const float*)dgl, (const float*)dbl, dx, P * C, C, 1.0f / (float)P);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// The BatchNorm backward whose stage 1 ran in the epilogue of the data-gradient convolution in front of it (lv_conv32_bnbwd /
// lv_conv1x1_bnbwd_f32): dv and partial [nblk][2][C] (sum dv, sum dv * xhat per workgroup) are given; writes dx and dgamma / dbeta.
extern "C" int lv_bn_bwd_apply_partials_f32(const float* x, const float* dv, const float* partial, int nblk, const float* mean,
                                            const float* invstd, const float* gamma, float* dx, float* dgamma, float* dbeta,
                                            int accumulate_param_grads, long P, int C, void* stream) {
    if (!x || !dv || !partial || !mean || !invstd || !gamma || !dx || !dgamma || !dbeta) return LV_ERR_ARG;
    if (P <= 0 || C <= 0 || nblk <= 0 || nblk > BN_BLOCKS) return LV_ERR_SHAPE;
    if (!bn_v4_ok(C)) return LV_ERR_UNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)dv | (uintptr_t)dx | (uintptr_t)partial) & 15) != 0) return LV_ERR_ALIGN;
    BN_APPLY_LAUNCH(bn_apply_bwd_v4_kernel, P * (C >> 2), C, stream, x, dv, partial, nblk, mean, invstd, gamma, dgamma, dbeta, accumulate_param_grads, dx, P, C, 1.0f / (float)P);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_bn_bwd2_f32(const float* x, const float* dy, const float* dy2, const float* y, const float* mean, const float* invstd,
                              const float* gamma, int act_elu, float* dv, float* dx, float* dgamma, float* dbeta,
                              int accumulate_param_grads, float* ws, long P, int C, void* stream) {
    return lv_bn_bwd4_f32(x, dy, dy2, nullptr, nullptr, y, mean, invstd, gamma, act_elu, dv, dx, dgamma, dbeta, accumulate_param_grads, ws, P, C,
                          stream);
}

// dy2 == NULL
extern "C" int lv_bn_bwd_f32(const float* x, const float* dy, const float* y, const float* mean, const float* invstd,
                             const float* gamma, int act_elu, float* dv, float* dx, float* dgamma, float* dbeta,
                             int accumulate_param_grads, float* ws, long P, int C, void* stream) {
    return lv_bn_bwd4_f32(x, dy, nullptr, nullptr, nullptr, y, mean, invstd, gamma, act_elu, dv, dx, dgamma, dbeta, accumulate_param_grads, ws, P, C,
                          stream);
}

extern "C" int lv_sigmoid_bce_fwd_f32(const float* logit, const float* x, float* rec, int B, int npix, float eps, void* stream) {
    if (!logit || !x || !rec || B <= 0 || npix <= 0) return LV_ERR_ARG;
    LV_LAUNCH(sigmoid_bce_fwd_kernel, dim3((unsigned)B), dim3(256), 0, stream, logit, x, rec, npix, eps);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_sigmoid_bce_bwd_f32(const float* logit, const float* x, const float* drec, float* dlogit, int B, int npix,
                                      float eps, void* stream) {
    if (!logit || !x || !drec || !dlogit || B <= 0 || npix <= 0) return LV_ERR_ARG;
    const long n = (long)B * npix;
    LV_LAUNCH(sigmoid_bce_bwd_kernel, dim3((unsigned)lv_cdiv(n, 256)), dim3(256), 0, stream, logit, x, drec, dlogit, npix, n, eps);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_dec_input_fwd_f32(const float* x, const float* zt, float* in5, int B, int npix, int fm, void* stream) {
    if (!x || !zt || !in5 || B <= 0 || npix <= 0 || fm < 0) return LV_ERR_ARG;
    const long n = (long)B * npix * (1 + fm);
    LV_LAUNCH(dec_input_fwd_kernel, dim3((unsigned)lv_cdiv(n, 256)), dim3(256), 0, stream, x, zt, in5, npix, fm, n);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_dec_input_bwd_f32(const float* din5, float* dzt, int B, int npix, int fm, void* stream) {
    if (!din5 || !dzt || B <= 0 || npix <= 0 || fm <= 0) return LV_ERR_ARG;
    const long n = (long)B * npix * fm;
    LV_LAUNCH(dec_input_bwd_kernel, dim3((unsigned)lv_cdiv(n, 256)), dim3(256), 0, stream, din5, dzt, npix, fm, n);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
