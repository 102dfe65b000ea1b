// lv_bn_totals.h -- train-mode BatchNorm statistics from per-workgroup partial sums, shared by the BatchNorm kernels (lv_conv.hip)
// and by the direct convolutions that apply BatchNorm + ELU to their INPUT while staging it (lv_conv_direct.hip).
#pragma once
#include "lv_device.h"

// BatchNorm2d (train) + ELU of a convolution's input, applied while the input is staged: partial = the stage-1 partial sums
// [nblk][2][C] the PRODUCER of x left behind; mean / invstd / running statistics are written by workgroup 0; y receives the
// activated input (the backward pass reads it: ELU' and the weight gradient), each element exactly once.
struct LvBnIn {
    const float* partial; int nblk;
    const float* gamma; const float* beta;
    float* mean; float* invstd; float* run_mean; float* run_var;
    float* y;
    long P;
    float eps, momentum;
};

// per-(q, c) totals of the per-block partials, the first 256 threads of the workgroup cooperating (every thread must call: two
// barriers inside): thread t sums float4 group t % (C/2) over blocks t / (C/2), + 256/(C/2), ... (16 independent loads in flight
// per batch: a runtime-length load -> add loop would pay one L2 round trip per block), f64, fixed order; tot: 2*C doubles,
// scratch: 1024 doubles (LDS).  C a power of two in [16, 256].
__device__ __forceinline__ void lv_bn_block_totals(const float* __restrict__ partial, int nblk, int C, double* tot, double* scratch) {
    const int tid = (int)threadIdx.x, npair = 2 * C, NF4 = C >> 1, nsub = 256 / NF4;
    if (tid < 256) {
        const int pg = tid & (NF4 - 1), bsub = tid / NF4;
        const float4* p4 = reinterpret_cast<const float4*>(partial) + pg;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int b = bsub;
        for (; b + 15 * nsub < nblk; b += 16 * nsub) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p4[(long)(b + u * nsub) * NF4];
#pragma unroll
            for (int u = 0; u < 16; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
        }
        {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int bb = b + u * nsub;
                v[u] = bb < nblk ? p4[(long)bb * NF4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; s2 += (double)v[u].z; s3 += (double)v[u].w; }
        }
        double* sc = scratch + (long)bsub * npair + 4 * pg;
        sc[0] = s0; sc[1] = s1; sc[2] = s2; sc[3] = s3;
    }
    __syncthreads();
    for (int t = tid; t < npair; t += (int)blockDim.x) {
        double acc = 0.0;
        for (int k = 0; k < nsub; ++k) acc += scratch[(long)k * npair + t];
        tot[t] = acc;
    }
    __syncthreads();
}

// channel c's (mean, invstd) from the totals; `publish`: also the saved statistics and the running-statistics update
__device__ __forceinline__ void lv_bn_channel_stats(const double* tot, int C, int c, long P, float eps, float momentum, bool publish,
                                                    float* mean_out, float* invstd_out, float* run_mean, float* run_var, float& mf,
                                                    float& isf) {
    const double m = tot[c] / (double)P;
    double var = tot[C + c] / (double)P - m * m;
    if (var < 0.0) var = 0.0;
    mf = (float)m;
    isf = (float)(1.0 / sqrt(var + (double)eps));
    if (publish) {
        mean_out[c] = mf;
        invstd_out[c] = isf;
        if (run_mean) {
            const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
            run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * m);
            run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * unb);
        }
    }
}

// y = ELU((x - mean) * invstd * gamma + beta): the expression of bn_apply_fwd_v4_kernel, term for term (bit-identical results)
__device__ __forceinline__ float lv_bn_elu(float x, float mu, float is, float ga, float be) {
    const float v = (x - mu) * is * ga + be;
    return v > 0.f ? v : expm1f(v);
}
