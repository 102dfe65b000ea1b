// lv_conv_direct.hip -- direct (implicit-GEMM) 32 -> 32 channel convolutions on 28 x 28 maps: the masked k x k convolutions
// of the PixelCNN decoder's residual blocks (modules/decoders/dec_pixelcnn_v2.py:12-62), forward, data gradient and weight
// gradient, without the im2col buffer.
//
// These 23 convolutions (k = 7 / 5 / 3, type-B masks) carry ~3/4 of the decoder's FLOPs.  Through im2col + the generic
// GEMM each one cost an im2col pass (up to 123 MB written and re-read), a GEMM with N = 32 on a 128-wide tile, a split-K
// reduction and a col2im pass.  Here a workgroup owns 4 image rows (4 waves x one row of 28 pixels, padded to the MFMA's
// 32) of one image: the rows plus their halo are staged ONCE in LDS (NHWC, pixel pitch 36 floats -> conflict-free
// ds_read_b128 of a pixel's 16-channel half), and the K loop runs over the taps that the mask keeps -- the type-B taps are a
// raster-order PREFIX (dec_pixelcnn_v2.py:17-20), so masked taps are skipped, not multiplied by zero -- with
// v_mfma_f32_32x32x2_f32 (exact f32): A = a row's 32 pixels x 2 channels (channel 16k + s for k-step s: both halves of a
// pixel are contiguous in LDS), B = that tap's 32 x 32 weight slice streamed from L2 in MFMA-fragment order (4 KB per tap,
// packed once per step by conv32_pack_kernel).  The data gradient is the same kernel on dy with the taps mirrored and the
// weight slice transposed.  The weight gradient keeps ALL k*k taps (the reference's masked taps carry non-zero gradients
// that enter the clip norm): a workgroup takes a slab of pixel rows and a quarter of the taps, each wave
// accumulates its taps' 32 x 32 blocks in registers over the whole slab (K = pixels), and a second kernel sums the slabs'
// partials in a fixed order straight into the reference's [Cout][Cin][kh][kw] gradient layout.
#include "lv_device.h"

namespace {

// Stage 1 of a BatchNorm BACKWARD inside the epilogue of the data-gradient convolution that produces its incoming gradient (round 4:
// as the forward statistics come from the convolutions' epilogues).  The kernel's result g = dL/d(BN output) becomes
// dv = g * ELU'(y) (y: the BatchNorm's saved output; ELU'(.) = 1 for y > 0, y + 1 otherwise) on its way out, and the workgroup
// leaves the per-channel (sum dv, sum dv * xhat) of what it stored, xhat = (x - mean) * invstd, in partial[block][2][C] -- the layout
// of bn_reduce_v4_kernel<1>, so that lv_bn_bwd_apply_partials_f32 (lv_conv.hip) finishes the BatchNorm with no reduction launch.
struct BnBwdFuse {
    const float* y;            // saved BatchNorm output [P][C] (null with act == 0)
    const float* x;            // BatchNorm input [P][C]
    const float* mean;         // [C]
    const float* invstd;       // [C]
    int act;                   // ELU behind the BatchNorm
};
// (all of a lane's y / x loads are issued before the first store: taken one element at a time -- load, compute, store -- every
// element was its own memory round trip behind the previous element's store, and the fused epilogue cost 7 us per launch)
__device__ __forceinline__ void bn_bwd_fuse16(const BnBwdFuse& f, const f32x16& acc, const long (&idx)[16], uint32_t valid, int c,
                                              float* __restrict__ out, float& st0, float& st1) {
    float yv[16], xv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) xv[e] = f.x[idx[e]];
    if (f.act) {
#pragma unroll
        for (int e = 0; e < 16; ++e) yv[e] = f.y[idx[e]];
    }
    const float mu = f.mean[c], is = f.invstd[c];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float g = acc[e];
        if (f.act) g = yv[e] > 0.f ? g : g * (yv[e] + 1.f);
        if ((valid >> e) & 1u) {
            out[idx[e]] = g;
            st0 += g;
            st1 += g * ((xv[e] - mu) * is);
        }
    }
}

constexpr int CC = 32;                 // channels in and out
constexpr int IW = 28, IH = 28;        // feature map
constexpr int TR = 4;                  // image rows per workgroup (one per wave)
constexpr int PP = 36;                 // LDS pixel pitch in floats (16-byte aligned, 9 sixteen-byte slots: odd -> no conflicts)
constexpr int KMAX = 7;
constexpr int HALO_F4 = ((TR + KMAX - 1) * (IW + KMAX - 1) * (CC / 4) + 255) / 256;      // float4 per thread: the k = 7 halo

// wp[t][s][lane]: lane (j = l & 31, k = l >> 5) holds the B element of k-step s: weight of (ci = 16k + s) -> (co = j) at tap t.
// transpose != 0 (data gradient): roles of ci / co swapped (the kernel negates the tap offsets itself).
__global__ __launch_bounds__(256) void conv32_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int KK, int ntaps,
                                                          int transpose) {
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (idx >= ntaps * 16 * 64) return;
    const int l = idx & 63, s = (idx >> 6) & 15, t = idx >> 10;
    const int j = l & 31, c = 16 * (l >> 5) + s;
    // reference layout w[co][ci][tap]
    wp[idx] = transpose ? w[((long)c * CC + j) * KK + t] : w[((long)j * CC + c) * KK + t];
}

// y[n][r][x][co] = sum_{t < ntaps} sum_ci in[n][r + dy_t][x + dx_t][ci] * W_t[ci][co],  (dy_t, dx_t) = (t / k - p, t % k - p),
// negated when `mirror` (data gradient: the prefix of the mirrored tap order).
// bn_partial != nullptr: also writes this workgroup's per-channel (sum, sum of squares) of its outputs as
// bn_partial[block][2][32] -- the stage-1 partials of the BatchNorm that follows (lv_bn_fwd_partials_f32), saving its pass over y.
// KS = 1: a workgroup owns 4 image rows, one per wave.  KS = 2: 2 rows, and two waves share a row by splitting the taps (their
// accumulators meet in LDS): twice the workgroups of half the MFMA time each.  The work is MFMA-bound and a wave-row is its
// indivisible unit; at the benchmark's 50 images 1400 units on 1024 SIMDs take 2 rounds, 2800 half units take 3 half rounds.
template <int KS>
__global__ __launch_bounds__(256) void conv32_direct_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                            float* __restrict__ out, float* __restrict__ bn_partial, int k, int ntaps,
                                                            int mirror, int accumulate, BnBwdFuse fuse) {
    constexpr int TRW = TR / KS;         // image rows per workgroup
    __shared__ __attribute__((aligned(16))) float halo[(TRW + KMAX - 1) * (IW + KMAX - 1) * PP];
    __shared__ float sstat[TRW][2][CC];
    __shared__ float red[KS == 2 ? TRW * 16 * 64 : 1];
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wr = w % TRW, wk = w / TRW;                    // this wave's row of the tile and its share of the taps
    const int n = (int)blockIdx.x / (IH / TRW), r0 = ((int)blockIdx.x % (IH / TRW)) * TRW;
    const int p = k / 2, HW = IW + 2 * p, HR = TRW + 2 * p;
    // stage rows r0-p .. r0+TRW-1+p, columns -p .. IW-1+p (zero outside the image): all of a thread's loads first (in flight
    // together), then the LDS writes -- a load -> store loop would pay one memory round trip per iteration.  The loads are
    // UNCONDITIONAL, from coordinates clamped into the image, and what lies outside is zeroed by a select at the LDS write: a
    // float4 loaded behind an `if` meets the zero in a phi, stops being one register tuple and is copied out right behind the
    // load, i.e. `s_waitcnt vmcnt(0)` after every load (the first form of this loop: 17 such waits per workgroup).
    {
        float4 hv[HALO_F4];
        uint32_t inside = 0u;
#pragma unroll
        for (int u = 0; u < HALO_F4; ++u) {
            const int i = tid + 256 * u;
            const int c4 = i % (CC / 4), hx = (i / (CC / 4)) % HW, hy = i / ((CC / 4) * HW);
            const int gy = r0 - p + hy, gx = hx - p;
            if (hy < HR && gy >= 0 && gy < IH && gx >= 0 && gx < IW) inside |= 1u << u;
            const int cy = gy < 0 ? 0 : (gy < IH ? gy : IH - 1), cx = gx < 0 ? 0 : (gx < IW ? gx : IW - 1);
            hv[u] = *reinterpret_cast<const float4*>(in + (((long)n * IH + cy) * IW + cx) * CC + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < HALO_F4; ++u) {
            const int i = tid + 256 * u;
            const bool ok = (inside >> u) & 1u;
            const float4 v = make_float4(ok ? hv[u].x : 0.f, ok ? hv[u].y : 0.f, ok ? hv[u].z : 0.f, ok ? hv[u].w : 0.f);
            if (i < HR * HW * (CC / 4)) *reinterpret_cast<float4*>(&halo[(i / (CC / 4)) * PP + 4 * (i % (CC / 4))]) = v;
        }
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int px = (l & 31) < IW ? (l & 31) : IW - 1;        // lanes 28..31 shadow pixel 27: their rows of the result are dropped
    const int kh = l >> 5;
    const int t_lo = wk == 0 ? 0 : (ntaps + 1) / 2;          // KS = 1: all taps
    const int t_hi = (KS == 1 || wk == 1) ? ntaps : (ntaps + 1) / 2;
    // the tap's weight fragment comes from L2 (4 KB per tap, shared by every workgroup): two register sets, the fetch of tap
    // t + 1 issued (and pinned there: the scheduler would sink it below the MFMAs) before tap t multiplies
    auto fetch_b = [&](float (&b)[16], int t) {
        const float* bw = wp + (long)(t < ntaps ? t : ntaps - 1) * 16 * 64 + l;
#pragma unroll
        for (int s = 0; s < 16; ++s) b[s] = bw[s * 64];
    };
    auto tap = [&](const float (&b)[16], int t) {
        int dy = t / k - p, dx = t % k - p;
        if (mirror) { dy = -dy; dx = -dx; }
        const float* a = &halo[((wr + p + dy) * HW + (px + p + dx)) * PP + 16 * kh];
        const float4 a0 = *reinterpret_cast<const float4*>(a), a1 = *reinterpret_cast<const float4*>(a + 4);
        const float4 a2 = *reinterpret_cast<const float4*>(a + 8), a3 = *reinterpret_cast<const float4*>(a + 12);
        const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = lv_mfma_32x32x2(av[s], b[s], acc);
    };
    float b0[16], b1[16];
    fetch_b(b0, t_lo);
    int t = t_lo;
    for (; t + 1 < t_hi; t += 2) {
        fetch_b(b1, t + 1);
        LV_SCHED_BARRIER();
        tap(b0, t);
        LV_SCHED_BARRIER();
        fetch_b(b0, t + 2);
        LV_SCHED_BARRIER();
        tap(b1, t + 1);
        LV_SCHED_BARRIER();
    }
    if (t < t_hi) tap(b0, t);
    if constexpr (KS == 2) {             // the two tap halves of a row meet in LDS
        if (wk == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(wr * 16 + e) * 64 + l] = acc[e];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += red[(wr * 16 + e) * 64 + l];
        }
    }
    const int r = r0 + wr;
    const int col = l & 31;
    float st0 = 0.f, st1 = 0.f;
    if (wk == 0 && fuse.x) {                                    // data gradient feeding a BatchNorm backward (see BnBwdFuse)
        long idx[16];
        uint32_t valid = 0u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int x = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
            if (x < IW) valid |= 1u << e;
            idx[e] = (((long)n * IH + r) * IW + (x < IW ? x : IW - 1)) * CC + col;      // clamped: loaded, never stored
        }
        bn_bwd_fuse16(fuse, acc, idx, valid, col, out, st0, st1);
    } else if (wk == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int x = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
            if (x < IW) {
                float* o = out + (((long)n * IH + r) * IW + x) * CC + col;
                const float v = accumulate ? *o + acc[e] : acc[e];
                *o = v;
                st0 += v; st1 += v * v;
            }
        }
    }
    if (bn_partial) {
        st0 += __shfl_xor(st0, 32, 64);
        st1 += __shfl_xor(st1, 32, 64);
        if (wk == 0 && l < 32) { sstat[wr][0][l] = st0; sstat[wr][1][l] = st1; }
        __syncthreads();
        if (tid < 2 * CC) {
            const int q = tid / CC, c = tid % CC;
            float tsum = 0.f;
#pragma unroll
            for (int i = 0; i < TRW; ++i) tsum += sstat[i][q][c];
            bn_partial[((long)blockIdx.x * 2 + q) * CC + c] = tsum;
        }
    }
}

// ---- split-bf16 forms of the same convolution (round 4) -------------------------------------------------------------------------
// The exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32: 64 cycles for 4096 multiply-adds) is what bounds the 5 x 5 and 7 x 7 masked
// convolutions (~1 us per tap at B = 50, profiles/microbench/conv32_probe.py).  Here every operand is written as the sum of two
// bf16 numbers, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (|x - hi - lo| <= 2^-17 |x|), and a product as
// hi*hi' + hi*lo' + lo*hi' on v_mfma_f32_32x32x16_bf16 with f32 accumulation (TERMS = 3: the dropped lo*lo' term and the
// representation error are ~2^-16 relative per product, i.e. f32-like sums; TERMS = 1: plain bf16 operands, hi*hi' only): 6 (2)
// instructions of 32 cycles per tap instead of 16 of 64.  The split happens ONCE per element -- activations when the halo is staged
// into LDS (two bf16 images, pixel pitch 80 bytes = 5 sixteen-byte slots: conflict-free ds_read_b128 of a pixel's 8-channel
// group), weights when they are packed -- so the tap loop is LDS reads, weight-fragment loads and MFMAs only.
// A operand: lane (m = l & 31: pixel, kh = l >> 5) holds channels 16 ks + 8 kh + [0, 8) of k-step ks; B: lane (n = l & 31: output
// channel, kh) the same channels of the tap's weight slice; D as in the f32 kernel.
constexpr int PPB = 80;                // bytes per halo pixel of a bf16 image

__device__ __forceinline__ void split_bf16(float x, uint32_t& hi, uint32_t& lo) {
    hi = lv_f32_to_bf16_bits(x);
    const uint32_t hb = hi << 16;                      // bf16 = the high half of an f32
    float hf;
    memcpy(&hf, &hb, 4);
    lo = lv_f32_to_bf16_bits(x - hf);
}

// wp16[t][ks (2)][part (hi, lo)][lane (64)] uint4: lane (j = l & 31, kh = l >> 5) holds, for output channel j, the weights of the
// input channels c = 16 ks + 8 kh + e, e < 8, at tap t (transpose: roles of the channel indices swapped, as conv32_pack_kernel)
__global__ __launch_bounds__(256) void conv32_pack_b16_kernel(const float* __restrict__ w, uint4* __restrict__ wp, int KK, int ntaps,
                                                              int transpose) {
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (idx >= ntaps * 4 * 64) return;
    const int l = idx & 63, part = (idx >> 6) & 1, ks = (idx >> 7) & 1, t = idx >> 8;
    const int j = l & 31, c0 = 16 * ks + 8 * (l >> 5);
    uint32_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        const float v = transpose ? w[((long)c * CC + j) * KK + t] : w[((long)j * CC + c) * KK + t];
        uint32_t hi, lo;
        split_bf16(v, hi, lo);
        h[e] = part ? lo : hi;
    }
    wp[idx] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}

template <int KS, int TERMS>
__global__ __launch_bounds__(256) void conv32_direct_b16_kernel(const float* __restrict__ in, const uint4* __restrict__ wp,
                                                                float* __restrict__ out, float* __restrict__ bn_partial, int k, int ntaps,
                                                                int mirror, int accumulate, BnBwdFuse fuse) {
    constexpr int TRW = TR / KS;         // image rows per workgroup
    constexpr int HALO_PIX = (TRW + KMAX - 1) * (IW + KMAX - 1);
    __shared__ __attribute__((aligned(16))) unsigned char halo_hi[HALO_PIX * PPB];
    __shared__ __attribute__((aligned(16))) unsigned char halo_lo[TERMS == 3 ? HALO_PIX * PPB : 16];
    __shared__ float sstat[TRW][2][CC];
    __shared__ float red[KS == 2 ? TRW * 16 * 64 : 1];
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wr = w % TRW, wk = w / TRW;                    // this wave's row of the tile and its share of the taps
    const int n = (int)blockIdx.x / (IH / TRW), r0 = ((int)blockIdx.x % (IH / TRW)) * TRW;
    const int p = k / 2, HW = IW + 2 * p, HR = TRW + 2 * p;
    {
        // staging as in conv32_direct_kernel (all loads first, unconditional from clamped coordinates, zero outside the image by a
        // select), each float4 split into four (hi, lo) pairs on its way into LDS
        float4 hv[HALO_F4];
        uint32_t inside = 0u;
#pragma unroll
        for (int u = 0; u < HALO_F4; ++u) {
            const int i = tid + 256 * u;
            const int c4 = i % (CC / 4), hx = (i / (CC / 4)) % HW, hy = i / ((CC / 4) * HW);
            const int gy = r0 - p + hy, gx = hx - p;
            if (hy < HR && gy >= 0 && gy < IH && gx >= 0 && gx < IW) inside |= 1u << u;
            const int cy = gy < 0 ? 0 : (gy < IH ? gy : IH - 1), cx = gx < 0 ? 0 : (gx < IW ? gx : IW - 1);
            hv[u] = *reinterpret_cast<const float4*>(in + (((long)n * IH + cy) * IW + cx) * CC + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < HALO_F4; ++u) {
            const int i = tid + 256 * u;
            const bool ok = (inside >> u) & 1u;
            const float v[4] = {ok ? hv[u].x : 0.f, ok ? hv[u].y : 0.f, ok ? hv[u].z : 0.f, ok ? hv[u].w : 0.f};
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split_bf16(v[e], hi[e], lo[e]);
            if (i < HR * HW * (CC / 4)) {
                const int off = (i / (CC / 4)) * PPB + 8 * (i % (CC / 4));
                *reinterpret_cast<uint2*>(&halo_hi[off]) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                if constexpr (TERMS == 3) *reinterpret_cast<uint2*>(&halo_lo[off]) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
            }
        }
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int px = (l & 31) < IW ? (l & 31) : IW - 1;        // lanes 28..31 shadow pixel 27: their rows of the result are dropped
    const int kh = l >> 5;
    const int t_lo = wk == 0 ? 0 : (ntaps + 1) / 2;          // KS = 1: all taps
    const int t_hi = (KS == 1 || wk == 1) ? ntaps : (ntaps + 1) / 2;
    constexpr int NB = TERMS == 3 ? 4 : 2;                   // weight fragments per tap: [ks][part]
    auto fetch_b = [&](uint4 (&b)[4], int t) {
        const uint4* bw = wp + (long)(t < ntaps ? t : ntaps - 1) * 4 * 64 + l;
        b[0] = bw[0];                                        // ks 0, hi
        b[2] = bw[2 * 64];                                   // ks 1, hi
        if constexpr (TERMS == 3) { b[1] = bw[64]; b[3] = bw[3 * 64]; }
    };
    auto tap = [&](const uint4 (&b)[4], int t) {
        int dy = t / k - p, dx = t % k - p;
        if (mirror) { dy = -dy; dx = -dx; }
        const int off = ((wr + p + dy) * HW + (px + p + dx)) * PPB + 16 * kh;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 ah = *reinterpret_cast<const uint4*>(&halo_hi[off + 32 * ks]);
            if constexpr (TERMS == 3) {
                const uint4 al = *reinterpret_cast<const uint4*>(&halo_lo[off + 32 * ks]);
                acc = lv_mfma_32x32x16_bf16(al, b[2 * ks], acc);          // the small terms first
                acc = lv_mfma_32x32x16_bf16(ah, b[2 * ks + 1], acc);
            }
            acc = lv_mfma_32x32x16_bf16(ah, b[2 * ks], acc);
        }
    };
    (void)NB;
    uint4 b0[4], b1[4];
    fetch_b(b0, t_lo);
    int t = t_lo;
    for (; t + 1 < t_hi; t += 2) {
        fetch_b(b1, t + 1);
        LV_SCHED_BARRIER();
        tap(b0, t);
        LV_SCHED_BARRIER();
        fetch_b(b0, t + 2);
        LV_SCHED_BARRIER();
        tap(b1, t + 1);
        LV_SCHED_BARRIER();
    }
    if (t < t_hi) tap(b0, t);
    if constexpr (KS == 2) {             // the two tap halves of a row meet in LDS
        if (wk == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) red[(wr * 16 + e) * 64 + l] = acc[e];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += red[(wr * 16 + e) * 64 + l];
        }
    }
    const int r = r0 + wr;
    const int col = l & 31;
    float st0 = 0.f, st1 = 0.f;
    if (wk == 0 && fuse.x) {                                    // data gradient feeding a BatchNorm backward (see BnBwdFuse)
        long idx[16];
        uint32_t valid = 0u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int x = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
            if (x < IW) valid |= 1u << e;
            idx[e] = (((long)n * IH + r) * IW + (x < IW ? x : IW - 1)) * CC + col;      // clamped: loaded, never stored
        }
        bn_bwd_fuse16(fuse, acc, idx, valid, col, out, st0, st1);
    } else if (wk == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int x = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
            if (x < IW) {
                float* o = out + (((long)n * IH + r) * IW + x) * CC + col;
                const float v = accumulate ? *o + acc[e] : acc[e];
                *o = v;
                st0 += v; st1 += v * v;
            }
        }
    }
    if (bn_partial) {
        st0 += __shfl_xor(st0, 32, 64);
        st1 += __shfl_xor(st1, 32, 64);
        if (wk == 0 && l < 32) { sstat[wr][0][l] = st0; sstat[wr][1][l] = st1; }
        __syncthreads();
        if (tid < 2 * CC) {
            const int q = tid / CC, c = tid % CC;
            float tsum = 0.f;
#pragma unroll
            for (int i = 0; i < TRW; ++i) tsum += sstat[i][q][c];
            bn_partial[((long)blockIdx.x * 2 + q) * CC + c] = tsum;
        }
    }
}

// tap split (KS = 2) only where it shortens the schedule: wave-rows on 1024 SIMDs, rounds x time per unit
static inline int conv32_ks(int N) {
    const long units = (long)N * IH;                         // wave-rows
    const long r1 = (units + 1023) / 1024 * 2, r2 = (2 * units + 1023) / 1024;       // in half units of time
    return r2 < r1 ? 2 : 1;
}

// weight gradient, stage 1.  grid (slabs, G tap groups), G = 4 / 2 / 1 for k = 7 / 5 / 3; slab = a contiguous range of 4-row
// tiles; wave w of group g owns taps t = (4g + w) + 4G*q, q < 4: dwp[slab][t][ci][co] = sum over the slab's pixels of
// x[pixel + off_t][ci] * dy[pixel][co], two pixels per v_mfma_f32_32x32x2_f32 (k = the pixel pair).  The dy operand of a pixel
// pair is read from LDS once and reused by the wave's taps; LDS pixel pitch 32 floats puts the two pixels of a pair on disjoint
// bank halves (conflict-free ds_read_b32).  The next tile's rows are fetched into registers while the current tile is
// multiplied, so the staging cost is the LDS write only.
constexpr int WG_TAPS = 7;             // taps per wave at most (each on HALF of a tile's pixels)
constexpr int WPP = 32;                // LDS pixel pitch of the weight-gradient kernel
constexpr int WG_HALO_F4 = HALO_F4;
constexpr int WG_GY_F4 = (TR * IW * (CC / 4) + 255) / 256;

__device__ __forceinline__ void wgrad_fetch(const float* __restrict__ x, const float* __restrict__ dy, int tile, int p, int HW,
                                            int HR, int tid, float4 (&hv)[WG_HALO_F4], float4 (&gv)[WG_GY_F4]) {
    const int n = tile / (IH / TR), r0 = (tile % (IH / TR)) * TR;
#pragma unroll
    for (int u = 0; u < WG_HALO_F4; ++u) {
        const int i = tid + 256 * u;
        const int c4 = i % (CC / 4), hx = (i / (CC / 4)) % HW, hy = i / ((CC / 4) * HW);
        const int yy = r0 - p + hy, xx = hx - p;
        hv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hy < HR && yy >= 0 && yy < IH && xx >= 0 && xx < IW)
            hv[u] = *reinterpret_cast<const float4*>(x + (((long)n * IH + yy) * IW + xx) * CC + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < WG_GY_F4; ++u) {
        const int i = tid + 256 * u;
        gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < TR * IW * (CC / 4)) gv[u] = *reinterpret_cast<const float4*>(dy + ((long)n * IH + r0) * IW * CC + 4L * i);
    }
}

// one staged tile: D_q[ci][co] += x[pixel + off_q][ci] * dy[pixel][co] over rows py0, py0 + 1 of the tile (56 pixels, two per
// MFMA), NQ taps
template <int NQ>
__device__ __forceinline__ void wgrad_tile(const float* halo, const float* gy, int HW, int kp, int ci, int py0,
                                           const int (&base)[WG_TAPS], f32x16 (&acc)[WG_TAPS]) {
    for (int py = py0; py < py0 + TR / 2; ++py) {
        const float* hrow = halo + py * HW * WPP;
        const float* grow = gy + (py * IW + kp) * WPP + ci;
#pragma unroll 7
        for (int xx = 0; xx < IW / 2; ++xx) {                // pixels 2xx + kp of row py
            const float b = grow[2 * xx * WPP];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = lv_mfma_32x32x2(hrow[base[q] + 2 * xx * WPP], b, acc[q]);
        }
    }
}

__global__ __launch_bounds__(256) void conv32_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dwp, int N, int k, int tiles_per_slab) {
    constexpr int HALO_N = (TR + KMAX - 1) * (IW + KMAX - 1) * WPP, GY_N = TR * IW * WPP;
    static_assert(HALO_N + GY_N >= 2 * WG_TAPS * 16 * 64, "the end-of-slab exchange reuses the staging buffers");
    __shared__ __attribute__((aligned(16))) float lds[HALO_N + GY_N];
    float* const halo = lds;
    float* const gy = lds + HALO_N;
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int KK = k * k, p = k / 2, HW = IW + 2 * p, HR = TR + 2 * p;
    // Waves 2j and 2j + 1 of a workgroup share a TAP LANE and split every tile's pixels (rows 0-1 / rows 2-3): a lane's share is
    // ceil(k*k / lanes) taps on half the pixels, i.e. the tap granularity is halved (k = 7: 7 x 1/2 = 3.5 tap-tiles per wave
    // against 4 when every wave owns whole taps: 49 taps do not divide by 16 waves).  The two halves meet in LDS at the end.
    const int nw = 2 * (int)gridDim.y;                       // tap lanes that share the taps
    const int gw = (int)blockIdx.y * 2 + (w >> 1);
    const int py0 = (TR / 2) * (w & 1);
    f32x16 acc[WG_TAPS];
#pragma unroll
    for (int q = 0; q < WG_TAPS; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const int ntiles = N * (IH / TR);
    const int t0 = (int)blockIdx.x * tiles_per_slab;
    const int t1 = t0 + tiles_per_slab < ntiles ? t0 + tiles_per_slab : ntiles;
    const int ci = l & 31, kp = l >> 5;
    // LDS offset of this lane's x operand for pixel pair 0 of row 0, per tap; taps beyond k*k alias tap 0 and are not stored
    int base[WG_TAPS];
    int nq = 0;
#pragma unroll
    for (int q = 0; q < WG_TAPS; ++q) {
        const int t = gw + nw * q;
        if (t < KK) nq = q + 1;
        const int tt = t < KK ? t : 0;
        base[q] = ((tt / k) * HW + (kp + tt % k)) * WPP + ci;
    }
    nq = lv_wave_uniform(nq);
    float4 hv[WG_HALO_F4], gv[WG_GY_F4];
    if (t0 < t1) wgrad_fetch(x, dy, t0, p, HW, HR, tid, hv, gv);
    for (int tile = t0; tile < t1; ++tile) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < WG_HALO_F4; ++u) {
            const int i = tid + 256 * u;
            if (i < HR * HW * (CC / 4)) *reinterpret_cast<float4*>(&halo[(i / (CC / 4)) * WPP + 4 * (i % (CC / 4))]) = hv[u];
        }
#pragma unroll
        for (int u = 0; u < WG_GY_F4; ++u) {
            const int i = tid + 256 * u;
            if (i < TR * IW * (CC / 4)) *reinterpret_cast<float4*>(&gy[4 * i]) = gv[u];
        }
        __syncthreads();
        if (tile + 1 < t1) wgrad_fetch(x, dy, tile + 1, p, HW, HR, tid, hv, gv);
        switch (nq) {          // wave-uniform: the tap loop below is unrolled with no predication
            case 1: wgrad_tile<1>(halo, gy, HW, kp, ci, py0, base, acc); break;
            case 2: wgrad_tile<2>(halo, gy, HW, kp, ci, py0, base, acc); break;
            case 3: wgrad_tile<3>(halo, gy, HW, kp, ci, py0, base, acc); break;
            case 4: wgrad_tile<4>(halo, gy, HW, kp, ci, py0, base, acc); break;
            case 5: wgrad_tile<5>(halo, gy, HW, kp, ci, py0, base, acc); break;
            case 6: wgrad_tile<6>(halo, gy, HW, kp, ci, py0, base, acc); break;
            case 7: wgrad_tile<7>(halo, gy, HW, kp, ci, py0, base, acc); break;
            default: break;
        }
    }
    // the two pixel halves of a tap lane meet in LDS (the staging buffers are free now), then the even wave writes the partial
    __syncthreads();
    float* const xch = lds + (w >> 1) * (WG_TAPS * 16 * 64);             // 2 tap lanes x 28 KB inside the 56.5 KB of staging buffers
    if (w & 1) {
#pragma unroll
        for (int q = 0; q < WG_TAPS; ++q)
            if (q < nq) {
#pragma unroll
                for (int e = 0; e < 16; ++e) xch[(q * 16 + e) * 64 + l] = acc[q][e];
            }
    }
    __syncthreads();
    if (w & 1) return;
    float* outp = dwp + (long)blockIdx.x * KK * CC * CC;
#pragma unroll
    for (int q = 0; q < WG_TAPS; ++q) {
        const int t = gw + nw * q;
        if (t >= KK) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);      // ci
            outp[((long)t * CC + row) * CC + (l & 31)] = acc[q][e] + xch[(q * 16 + e) * 64 + l];
        }
    }
}

// ---- split-bf16 weight gradient (round 4) -------------------------------------------------------------------------------------
// D_t[ci][co] += sum over the tile's 112 pixels of x[pixel + off_t][ci] dy[pixel][co] on v_mfma_f32_32x32x16_bf16: M = ci, N = co,
// K = 16 PIXELS per instruction (seven k-steps cover a 4-row tile exactly; a k-step's pixels may straddle an image row: every lane
// computes the address of its own pixel).  Both operands are K-contiguous fragments of images that are CHANNEL-contiguous in LDS
// (NHWC), i.e. transposed reads: ds_read_b64_tr_b16 (lv_ds_read_tr16_b64) hands lane (m = l & 31, kh = l >> 5) the four pixels
// 8 kh + 4 h + [0, 4) of its channel from four pixel rows of the image, two reads per fragment.  x and dy are split into (hi, lo)
// bf16 images when a tile is staged (pixel pitch 80 bytes); a wave owns whole taps (t = lane + lanes q, q < 4, lanes = 4 waves x
// gridDim.y) on ALL pixels of the tile: the dy fragments of a k-step are read once and reused by the wave's taps.  Partials go to
// the same [slab][t][ci][co] scratch as conv32_wgrad_kernel's, so the reduction stage is shared.
constexpr int WGB_TAPS = 4;

template <int NQ, int TERMS>
__device__ __forceinline__ void wgrad_b16_tile(const unsigned char* xh, const unsigned char* xl, const unsigned char* gh,
                                               const unsigned char* gl, const int (&offx)[14], const int (&offg)[14],
                                               const int (&toff)[WGB_TAPS], f32x16 (&acc)[WGB_TAPS]) {
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        const uint2 b0 = lv_ds_read_tr16_b64(gh + offg[2 * ks]), b1 = lv_ds_read_tr16_b64(gh + offg[2 * ks + 1]);
        const uint4 bh = make_uint4(b0.x, b0.y, b1.x, b1.y);
        uint4 bl = bh;
        if constexpr (TERMS == 3) {
            const uint2 c0 = lv_ds_read_tr16_b64(gl + offg[2 * ks]), c1 = lv_ds_read_tr16_b64(gl + offg[2 * ks + 1]);
            bl = make_uint4(c0.x, c0.y, c1.x, c1.y);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint2 a0 = lv_ds_read_tr16_b64(xh + offx[2 * ks] + toff[q]), a1 = lv_ds_read_tr16_b64(xh + offx[2 * ks + 1] + toff[q]);
            const uint4 ah = make_uint4(a0.x, a0.y, a1.x, a1.y);
            if constexpr (TERMS == 3) {
                const uint2 d0 = lv_ds_read_tr16_b64(xl + offx[2 * ks] + toff[q]), d1 = lv_ds_read_tr16_b64(xl + offx[2 * ks + 1] + toff[q]);
                const uint4 al = make_uint4(d0.x, d0.y, d1.x, d1.y);
                acc[q] = lv_mfma_32x32x16_bf16(al, bh, acc[q]);
                acc[q] = lv_mfma_32x32x16_bf16(ah, bl, acc[q]);
            }
            acc[q] = lv_mfma_32x32x16_bf16(ah, bh, acc[q]);
        }
    }
}

template <int TERMS>
__global__ __launch_bounds__(256) void conv32_wgrad_b16_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dwp, int N, int k, int tiles_per_slab) {
    constexpr int HALO_PIX = (TR + KMAX - 1) * (IW + KMAX - 1), GY_PIX = TR * IW;
    __shared__ __attribute__((aligned(16))) unsigned char xh[HALO_PIX * PPB];
    __shared__ __attribute__((aligned(16))) unsigned char xl[TERMS == 3 ? HALO_PIX * PPB : 16];
    __shared__ __attribute__((aligned(16))) unsigned char gh[GY_PIX * PPB];
    __shared__ __attribute__((aligned(16))) unsigned char gl[TERMS == 3 ? GY_PIX * PPB : 16];
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int KK = k * k, p = k / 2, HW = IW + 2 * p, HR = TR + 2 * p;
    const int nw = 4 * (int)gridDim.y, gw = (int)blockIdx.y * 4 + w;      // tap lanes: every wave owns whole taps on all pixels
    f32x16 acc[WGB_TAPS];
#pragma unroll
    for (int q = 0; q < WGB_TAPS; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const int ntiles = N * (IH / TR);
    const int t0 = (int)blockIdx.x * tiles_per_slab;
    const int t1 = t0 + tiles_per_slab < ntiles ? t0 + tiles_per_slab : ntiles;
    // this lane's part in the transposing reads: r = lane within its group of 16 points at pixel row (r >> 2) of the four and at
    // channels m0 + 4 (r & 3) + [0, 4); the group (l >> 4) fixes m0 = 16 (g & 1) and the k half 8 (g >> 1)
    const int r = l & 15, g16 = l >> 4;
    const int chb = (16 * (g16 & 1) + 4 * (r & 3)) * 2, kb = 8 * (g16 >> 1) + (r >> 2);
    int offx[14], offg[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
        const int q = 16 * (i >> 1) + kb + 4 * (i & 1);                  // pixel of the tile, row-major over its 4 x 28 pixels
        offx[i] = ((q / IW) * HW + q % IW) * PPB + chb;
        offg[i] = q * PPB + chb;
    }
    int toff[WGB_TAPS];
    int nq = 0;
#pragma unroll
    for (int q = 0; q < WGB_TAPS; ++q) {
        const int t = gw + nw * q;
        if (t < KK) nq = q + 1;
        const int tt = t < KK ? t : 0;
        toff[q] = ((tt / k) * HW + tt % k) * PPB;
    }
    nq = lv_wave_uniform(nq);
    float4 hv[WG_HALO_F4], gv[WG_GY_F4];
    if (t0 < t1) wgrad_fetch(x, dy, t0, p, HW, HR, tid, hv, gv);
    for (int tile = t0; tile < t1; ++tile) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < WG_HALO_F4; ++u) {
            const int i = tid + 256 * u;
            const float v[4] = {hv[u].x, hv[u].y, hv[u].z, hv[u].w};
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split_bf16(v[e], hi[e], lo[e]);
            if (i < HR * HW * (CC / 4)) {
                const int off = (i / (CC / 4)) * PPB + 8 * (i % (CC / 4));
                *reinterpret_cast<uint2*>(&xh[off]) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                if constexpr (TERMS == 3) *reinterpret_cast<uint2*>(&xl[off]) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
            }
        }
#pragma unroll
        for (int u = 0; u < WG_GY_F4; ++u) {
            const int i = tid + 256 * u;
            const float v[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split_bf16(v[e], hi[e], lo[e]);
            if (i < TR * IW * (CC / 4)) {
                const int off = (i / (CC / 4)) * PPB + 8 * (i % (CC / 4));
                *reinterpret_cast<uint2*>(&gh[off]) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                if constexpr (TERMS == 3) *reinterpret_cast<uint2*>(&gl[off]) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
            }
        }
        __syncthreads();
        if (tile + 1 < t1) wgrad_fetch(x, dy, tile + 1, p, HW, HR, tid, hv, gv);
        switch (nq) {          // wave-uniform: the tap loop is unrolled with no predication
            case 1: wgrad_b16_tile<1, TERMS>(xh, xl, gh, gl, offx, offg, toff, acc); break;
            case 2: wgrad_b16_tile<2, TERMS>(xh, xl, gh, gl, offx, offg, toff, acc); break;
            case 3: wgrad_b16_tile<3, TERMS>(xh, xl, gh, gl, offx, offg, toff, acc); break;
            case 4: wgrad_b16_tile<4, TERMS>(xh, xl, gh, gl, offx, offg, toff, acc); break;
            default: break;
        }
    }
    float* outp = dwp + (long)blockIdx.x * KK * CC * CC;
#pragma unroll
    for (int q = 0; q < WGB_TAPS; ++q) {
        const int t = gw + nw * q;
        if (t >= KK) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);      // ci
            outp[((long)t * CC + row) * CC + (l & 31)] = acc[q][e];
        }
    }
}

// stage 2: dw[co][ci][t] (=|+=) sum_slab dwp[slab][t][ci][co]   (fixed order; writes the reference's parameter layout)
__global__ __launch_bounds__(256) void conv32_wgrad_reduce_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int KK,
                                                                  int slabs, int accumulate) {
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (idx >= KK * CC * CC) return;
    const int co = idx % CC, ci = (idx / CC) % CC, t = idx / (CC * CC);
    float s = 0.f;
    for (int b0 = 0; b0 < slabs; b0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = dwp[(long)(b0 + u < slabs ? b0 + u : 0) * KK * CC * CC + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += b0 + u < slabs ? v[u] : 0.f;
    }
    float* o = dw + ((long)co * CC + ci) * KK + t;
    *o = accumulate ? *o + s : s;
}

}  // namespace

// floats of the packed weight image for `ntaps` taps / of the weight-gradient scratch for N images and a k x k kernel
extern "C" long lv_conv32_wpack_floats(int ntaps) { return (long)ntaps * 16 * 64; }
static inline int wgrad_groups(int k) { return k >= 7 ? 4 : k >= 5 ? 2 : 1; }     // x 2 tap lanes per workgroup
extern "C" int lv_conv32_wgrad_slabs(int N, int k) {
    const int ntiles = N * (IH / TR), cap = 256 / wgrad_groups(k);
    return ntiles < cap ? ntiles : cap;
}
extern "C" long lv_conv32_wgrad_ws_floats(int N, int k) { return (long)lv_conv32_wgrad_slabs(N, k) * k * k * CC * CC; }

// pack the first `ntaps` (raster order) taps of w [32][32][k*k] (the reference's nn.Conv2d layout) for lv_conv32_f32;
// transpose != 0: the data-gradient image (roles of the channel indices swapped)
extern "C" int lv_conv32_pack_f32(const float* w, float* wp, int k, int ntaps, int transpose, void* stream) {
    if (!w || !wp) return LV_ERR_ARG;
    if (k <= 0 || k > KMAX || !(k & 1) || ntaps <= 0 || ntaps > k * k) return LV_ERR_SHAPE;
    LV_LAUNCH(conv32_pack_kernel, dim3((unsigned)lv_cdiv((long)ntaps * 16 * 64, 256)), dim3(256), 0, stream, w, wp, k * k, ntaps,
              transpose);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// MaskedConv2d / nn.Conv2d 32 -> 32, k x k, stride 1, padding k/2 on NHWC [N][28][28][32] (dec_pixelcnn_v2.py:27-30, 44-48):
// out (=|+=) conv(in) over the first `ntaps` raster-order taps (k*k = an ordinary convolution; (k/2)*k + k/2 + 1 = type-B mask).
// mirror = 0: forward with wp = lv_conv32_pack_f32(w, ., transpose = 0); mirror = 1: data gradient (in = dy, out = dx) with the
// transposed image.
extern "C" int lv_conv32_f32(const float* in, const float* wp, float* out, int N, int k, int ntaps, int mirror, int accumulate,
                             void* stream) {
    if (!in || !wp || !out) return LV_ERR_ARG;
    if (N <= 0 || k <= 0 || k > KMAX || !(k & 1) || ntaps <= 0 || ntaps > k * k) return LV_ERR_SHAPE;
    if ((((uintptr_t)in) & 15) != 0) return LV_ERR_ALIGN;
    if (conv32_ks(N) == 2)
        LV_LAUNCH(conv32_direct_kernel<2>, dim3((unsigned)(N * (IH / 2))), dim3(256), 0, stream, in, wp, out, (float*)nullptr, k, ntaps,
                  mirror, accumulate, BnBwdFuse{});
    else
        LV_LAUNCH(conv32_direct_kernel<1>, dim3((unsigned)(N * (IH / TR))), dim3(256), 0, stream, in, wp, out, (float*)nullptr, k, ntaps,
                  mirror, accumulate, BnBwdFuse{});
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// forward convolution that also leaves the following BatchNorm's stage-1 partials: bn_partial [lv_conv32_blocks(N)][2][32]
extern "C" int lv_conv32_blocks(int N) { return N * (IH / (TR / conv32_ks(N))); }
// the tap-split form (1 or 2) lv_conv32_f32 uses at batch size N: lv_pixelcnn_pixel_step_f32 follows the same summation order
extern "C" int lv_conv32_tap_split(int N) { return conv32_ks(N); }
extern "C" int lv_conv32_bnstat_f32(const float* in, const float* wp, float* out, float* bn_partial, int N, int k, int ntaps, void* stream) {
    if (!in || !wp || !out || !bn_partial) return LV_ERR_ARG;
    if (N <= 0 || k <= 0 || k > KMAX || !(k & 1) || ntaps <= 0 || ntaps > k * k) return LV_ERR_SHAPE;
    if ((((uintptr_t)in) & 15) != 0) return LV_ERR_ALIGN;
    if (conv32_ks(N) == 2)
        LV_LAUNCH(conv32_direct_kernel<2>, dim3((unsigned)(N * (IH / 2))), dim3(256), 0, stream, in, wp, out, bn_partial, k, ntaps, 0, 0, BnBwdFuse{});
    else
        LV_LAUNCH(conv32_direct_kernel<1>, dim3((unsigned)(N * (IH / TR))), dim3(256), 0, stream, in, wp, out, bn_partial, k, ntaps, 0, 0, BnBwdFuse{});
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// The split-bf16 forms (see conv32_direct_b16_kernel): wp16 = lv_conv32_pack_b16(w, ., transpose) holds lv_conv32_wpack_floats(ntaps)
// floats' worth of bytes (the same buffer size as the f32 image).  terms: 3 = hi*hi' + hi*lo' + lo*hi' (f32-like results), 1 = plain
// bf16 operands.  bn_partial may be NULL (then as lv_conv32_f32), else as lv_conv32_bnstat_f32 (forward only).
extern "C" int lv_conv32_pack_b16(const float* w, void* wp16, int k, int ntaps, int transpose, void* stream) {
    if (!w || !wp16) return LV_ERR_ARG;
    if (k <= 0 || k > KMAX || !(k & 1) || ntaps <= 0 || ntaps > k * k) return LV_ERR_SHAPE;
    LV_LAUNCH(conv32_pack_b16_kernel, dim3((unsigned)lv_cdiv((long)ntaps * 4 * 64, 256)), dim3(256), 0, stream, w,
              reinterpret_cast<uint4*>(wp16), k * k, ntaps, transpose);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

static int conv32_b16_launch(const float* in, const void* wp16, float* out, float* bn_partial, int N, int k, int ntaps, int mirror,
                             int accumulate, int terms, BnBwdFuse fuse, void* stream) {
    if (!in || !wp16 || !out) return LV_ERR_ARG;
    if (N <= 0 || k <= 0 || k > KMAX || !(k & 1) || ntaps <= 0 || ntaps > k * k) return LV_ERR_SHAPE;
    if (terms != 1 && terms != 3) return LV_ERR_ARG;
    if (((((uintptr_t)in) | ((uintptr_t)wp16)) & 15) != 0) return LV_ERR_ALIGN;
    const uint4* wp = reinterpret_cast<const uint4*>(wp16);
    const bool two = conv32_ks(N) == 2;
    const dim3 grid((unsigned)(N * (IH / (two ? 2 : TR))));
    if (two && terms == 3) LV_LAUNCH((conv32_direct_b16_kernel<2, 3>), grid, dim3(256), 0, stream, in, wp, out, bn_partial, k, ntaps, mirror, accumulate, fuse);
    else if (two) LV_LAUNCH((conv32_direct_b16_kernel<2, 1>), grid, dim3(256), 0, stream, in, wp, out, bn_partial, k, ntaps, mirror, accumulate, fuse);
    else if (terms == 3) LV_LAUNCH((conv32_direct_b16_kernel<1, 3>), grid, dim3(256), 0, stream, in, wp, out, bn_partial, k, ntaps, mirror, accumulate, fuse);
    else LV_LAUNCH((conv32_direct_b16_kernel<1, 1>), grid, dim3(256), 0, stream, in, wp, out, bn_partial, k, ntaps, mirror, accumulate, fuse);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_conv32_b16(const float* in, const void* wp16, float* out, float* bn_partial, int N, int k, int ntaps, int mirror,
                             int accumulate, int terms, void* stream) {
    if (bn_partial && (mirror || accumulate)) return LV_ERR_ARG;
    return conv32_b16_launch(in, wp16, out, bn_partial, N, k, ntaps, mirror, accumulate, terms, BnBwdFuse{}, stream);
}

// The DATA GRADIENT of a masked convolution whose input came out of a BatchNorm (+ ELU), with stage 1 of that BatchNorm's backward in
// its epilogue (BnBwdFuse): in = dL/d(conv output), wp = the transposed image; dv [N*784][32] = dL/d(BN output) * ELU'(y) is written
// instead of the raw data gradient and partial [lv_conv32_blocks(N)][2][32] receives the per-channel (sum dv, sum dv * xhat) of every
// workgroup; lv_bn_bwd_apply_partials_f32 finishes the BatchNorm.  y: the BatchNorm's saved output (= the convolution's input),
// x: its input; act_elu = 0: no activation behind the BatchNorm (y unused).  terms = 0: the exact-f32 kernel (wp from
// lv_conv32_pack_f32), 1 / 3: the split-bf16 kernels (lv_conv32_pack_b16).
static int conv32_f32_launch(const float* in, const float* wp, float* out, float* bn_partial, int N, int k, int ntaps, int mirror,
                             int accumulate, BnBwdFuse fuse, void* stream) {
    if (!in || !wp || !out) return LV_ERR_ARG;
    if (N <= 0 || k <= 0 || k > KMAX || !(k & 1) || ntaps <= 0 || ntaps > k * k) return LV_ERR_SHAPE;
    if ((((uintptr_t)in) & 15) != 0) return LV_ERR_ALIGN;
    if (conv32_ks(N) == 2)
        LV_LAUNCH(conv32_direct_kernel<2>, dim3((unsigned)(N * (IH / 2))), dim3(256), 0, stream, in, wp, out, bn_partial, k, ntaps, mirror,
                  accumulate, fuse);
    else
        LV_LAUNCH(conv32_direct_kernel<1>, dim3((unsigned)(N * (IH / TR))), dim3(256), 0, stream, in, wp, out, bn_partial, k, ntaps, mirror,
                  accumulate, fuse);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_conv32_bnbwd(const float* in, const void* wp, float* dv, float* partial, int N, int k, int ntaps, const float* y,
                               const float* x, const float* mean, const float* invstd, int act_elu, int terms, void* stream) {
    if (!partial || !x || !mean || !invstd || (act_elu && !y)) return LV_ERR_ARG;
    const BnBwdFuse fuse{y, x, mean, invstd, act_elu};
    if (terms == 0) return conv32_f32_launch(in, reinterpret_cast<const float*>(wp), dv, partial, N, k, ntaps, 1, 0, fuse, stream);
    return conv32_b16_launch(in, wp, dv, partial, N, k, ntaps, 1, 0, terms, fuse, stream);
}

// weight gradient over ALL k*k taps: dw [32][32][k*k] (=|+=) sum_pixels dy[p][co] x[p + off_t][ci]; ws: lv_conv32_wgrad_ws_floats
// partial blocks lv_conv32_wgrad_f32 leaves in ws ([parts][k*k][32 ci][32 co]) -- what lv_wgrad_reduce_batched needs to know
extern "C" int lv_conv32_wgrad_parts(int N, int k) {
    const int ntiles = N * (IH / TR);
    return lv_cdiv(ntiles, lv_cdiv(ntiles, lv_conv32_wgrad_slabs(N, k)));
}
// dw == NULL: stage 1 only -- the partials stay in ws for a later lv_wgrad_reduce_batched
extern "C" int lv_conv32_wgrad_f32(const float* x, const float* dy, float* dw, float* ws, int N, int k, int accumulate, void* stream) {
    if (!x || !dy || !ws) return LV_ERR_ARG;
    if (N <= 0 || k <= 0 || k > KMAX || !(k & 1)) return LV_ERR_SHAPE;
    if (((((uintptr_t)x) | ((uintptr_t)dy)) & 15) != 0) return LV_ERR_ALIGN;
    const int ntiles = N * (IH / TR);
    const int slabs = lv_conv32_wgrad_slabs(N, k);
    const int tps = lv_cdiv(ntiles, slabs);
    const int used = lv_cdiv(ntiles, tps);
    LV_LAUNCH(conv32_wgrad_kernel, dim3((unsigned)used, (unsigned)wgrad_groups(k)), dim3(256), 0, stream, x, dy, ws, N, k, tps);
    if (dw)
        LV_LAUNCH(conv32_wgrad_reduce_kernel, dim3((unsigned)lv_cdiv((long)k * k * CC * CC, 256)), dim3(256), 0, stream, (const float*)ws,
                  dw, k * k, used, accumulate);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// the split-bf16 form of lv_conv32_wgrad_f32 (same scratch, same partial layout, same reduction stage; terms as lv_conv32_b16)
extern "C" int lv_conv32_wgrad_b16(const float* x, const float* dy, float* dw, float* ws, int N, int k, int accumulate, int terms,
                                   void* stream) {
    if (!x || !dy || !ws) return LV_ERR_ARG;
    if (N <= 0 || k <= 0 || k > KMAX || !(k & 1)) return LV_ERR_SHAPE;
    if (terms != 1 && terms != 3) return LV_ERR_ARG;
    if (((((uintptr_t)x) | ((uintptr_t)dy)) & 15) != 0) return LV_ERR_ALIGN;
    const int ntiles = N * (IH / TR);
    const int slabs = lv_conv32_wgrad_slabs(N, k);
    const int tps = lv_cdiv(ntiles, slabs);
    const int used = lv_cdiv(ntiles, tps);
    const dim3 grid((unsigned)used, (unsigned)wgrad_groups(k));
    if (terms == 3) LV_LAUNCH(conv32_wgrad_b16_kernel<3>, grid, dim3(256), 0, stream, x, dy, ws, N, k, tps);
    else LV_LAUNCH(conv32_wgrad_b16_kernel<1>, grid, dim3(256), 0, stream, x, dy, ws, N, k, tps);
    if (dw)
        LV_LAUNCH(conv32_wgrad_reduce_kernel, dim3((unsigned)lv_cdiv((long)k * k * CC * CC, 256)), dim3(256), 0, stream, (const float*)ws,
                  dw, k * k, used, accumulate);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// =====================================================================================================================
// Pointwise (1 x 1) convolutions between 32 and 64 channels (the bottleneck / expansion convolutions of PixelCNNBlock,
// dec_pixelcnn_v2.py:41-43,49-51, and the decoder head): out[p][co] = sum_ci in[p][ci] W[co][ci] over P = N*28*28 pixels.
// As tile GEMMs these are M = 39200, N <= 64, K <= 64 problems on 128 x 128 x 16 tiles (32 us each, plus split-K for the
// gradients); here a workgroup takes 128 pixels, stages them and the whole weight matrix (<= 16 KB) in LDS with coalesced
// loads and runs K/2 v_mfma_f32_32x32x2_f32 per 32-pixel wave and 32-channel output block.  The data gradient is the same
// kernel with the weight matrix read transposed; the weight gradient accumulates dy^T x over a slab of pixels in registers
// (one 32 x 32 block per wave) and a fixed-order reduction sums the slabs.
namespace {

constexpr int PWP = 160;               // pixels per workgroup (5 waves x 32): 39200 pixels -> 245 workgroups, one round on 256 CUs
constexpr int PWT = 2 * PWP;           // threads

template <int CIN, int COUT>
__global__ __launch_bounds__(PWT) void conv1x1_kernel(const float* __restrict__ in, const float* __restrict__ w, float* __restrict__ out,
                                                      float* __restrict__ bn_partial, long P, int w_transposed, int accumulate, BnBwdFuse fuse) {
    constexpr int PA = CIN + 4;        // LDS pitches (floats): 16-byte slots per row odd -> conflict-free ds_read_b128
    __shared__ __attribute__((aligned(16))) float sa[PWP * PA];
    __shared__ __attribute__((aligned(16))) float sw[COUT * PA];
    __shared__ float sstat[PWP / 32][2][COUT];
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const long p0 = (long)blockIdx.x * PWP;
    constexpr int NF4 = PWP * (CIN / 4) / PWT;           // float4 per thread (8 or 4): loads first, then the LDS writes
    {
        float4 v[NF4];
#pragma unroll
        for (int u = 0; u < NF4; ++u) {                   // unconditional loads from a clamped pixel (see conv32_direct_kernel)
            const int i = tid + PWT * u;
            const int c4 = i % (CIN / 4), pp = i / (CIN / 4);
            v[u] = *reinterpret_cast<const float4*>(in + (p0 + pp < P ? p0 + pp : P - 1) * CIN + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < NF4; ++u) {
            const int i = tid + PWT * u;
            const bool ok = p0 + i / (CIN / 4) < P;
            *reinterpret_cast<float4*>(&sa[(i / (CIN / 4)) * PA + 4 * (i % (CIN / 4))]) =
                make_float4(ok ? v[u].x : 0.f, ok ? v[u].y : 0.f, ok ? v[u].z : 0.f, ok ? v[u].w : 0.f);
        }
    }
    // sw[co][ci] = W[co][ci]  (w stored [COUT][CIN]), or W^T when the caller hands the [CIN][COUT] matrix of the forward
    for (int i = tid; i < COUT * CIN; i += PWT) {
        const int ci = i % CIN, co = i / CIN;
        sw[co * PA + ci] = w_transposed ? w[(long)ci * COUT + co] : w[i];
    }
    __syncthreads();
    const int kh = l >> 5;
    const float* a = &sa[(wv * 32 + (l & 31)) * PA + (CIN / 2) * kh];
    float av[CIN / 2];
#pragma unroll
    for (int s = 0; s < CIN / 2; s += 4) {
        const float4 q = *reinterpret_cast<const float4*>(a + s);
        av[s] = q.x; av[s + 1] = q.y; av[s + 2] = q.z; av[s + 3] = q.w;
    }
#pragma unroll
    for (int nb = 0; nb < COUT / 32; ++nb) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const float* b = &sw[(nb * 32 + (l & 31)) * PA + (CIN / 2) * kh];
#pragma unroll
        for (int s = 0; s < CIN / 2; s += 4) {
            const float4 q = *reinterpret_cast<const float4*>(b + s);
            acc = lv_mfma_32x32x2(av[s], q.x, acc);
            acc = lv_mfma_32x32x2(av[s + 1], q.y, acc);
            acc = lv_mfma_32x32x2(av[s + 2], q.z, acc);
            acc = lv_mfma_32x32x2(av[s + 3], q.w, acc);
        }
        float st0 = 0.f, st1 = 0.f;
        if (fuse.x) {                                           // data gradient feeding a BatchNorm backward (see BnBwdFuse)
            long idx[16];
            uint32_t valid = 0u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long pp = p0 + wv * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
                if (pp < P) valid |= 1u << e;
                idx[e] = (pp < P ? pp : P - 1) * COUT + nb * 32 + (l & 31);
            }
            bn_bwd_fuse16(fuse, acc, idx, valid, nb * 32 + (l & 31), out, st0, st1);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long pp = p0 + wv * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
                if (pp < P) {
                    float* o = out + pp * COUT + nb * 32 + (l & 31);
                    const float v = accumulate ? *o + acc[e] : acc[e];
                    *o = v;
                    st0 += v; st1 += v * v;
                }
            }
        }
        if (bn_partial) {
            st0 += __shfl_xor(st0, 32, 64);
            st1 += __shfl_xor(st1, 32, 64);
            if (l < 32) { sstat[wv][0][nb * 32 + l] = st0; sstat[wv][1][nb * 32 + l] = st1; }
        }
    }
    if (bn_partial) {
        __syncthreads();
        if (tid < 2 * COUT) {
            const int q = tid / COUT, c = tid % COUT;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < PWP / 32; ++k) t += sstat[k][q][c];
            bn_partial[((long)blockIdx.x * 2 + q) * COUT + c] = t;
        }
    }
}

// stage 1 of the weight gradient: every wave takes a contiguous run of pixels and accumulates the whole dy^T x matrix
// ((COUT/32) x (CIN/32) MFMA blocks) with operands loaded straight from global memory -- for v_mfma_f32_32x32x2_f32 with
// k = pixel both operands are channel-contiguous per pixel, i.e. each load is two coalesced 128-byte rows -- in batches of
// PW_U pixel pairs so that all of a batch's loads are in flight together.  The 8 waves of a workgroup are summed through
// LDS in a fixed tree and the workgroup writes one partial: dwp[wg][co][ci].
constexpr int PW_U = 10;
template <int CIN, int COUT>
__global__ __launch_bounds__(512) void conv1x1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dwp, long P, long pix_per_wave) {
    constexpr int NA = COUT / 32, NB = CIN / 32, NACC = NA * NB;
    __shared__ float red[4][NACC * 16 * 64];
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const int kp = l >> 5, cl = l & 31;
    f32x16 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const long p0 = ((long)blockIdx.x * 8 + wv) * pix_per_wave;
    const long p1 = p0 + pix_per_wave < P ? p0 + pix_per_wave : P;
    for (long p = p0; p < p1; p += 2 * PW_U) {
        float a[PW_U][NA], b[PW_U][NB];
#pragma unroll
        for (int u = 0; u < PW_U; ++u) {
            const long pp = p + 2 * u + kp;
            const long pc = pp < p1 ? pp : p1 - 1;         // unconditional loads from a clamped pixel; the tail is zeroed below
#pragma unroll
            for (int i = 0; i < NA; ++i) a[u][i] = dy[pc * COUT + 32 * i + cl];
#pragma unroll
            for (int j = 0; j < NB; ++j) b[u][j] = x[pc * CIN + 32 * j + cl];
        }
        if (p + 2 * PW_U > p1) {                           // ragged last batch only (uniform): pixels beyond the run count as zero
#pragma unroll
            for (int u = 0; u < PW_U; ++u) {
                if (p + 2 * u + kp >= p1) {
#pragma unroll
                    for (int i = 0; i < NA; ++i) a[u][i] = 0.f;
#pragma unroll
                    for (int j = 0; j < NB; ++j) b[u][j] = 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PW_U; ++u)
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i * NB + j] = lv_mfma_32x32x2(a[u][i], b[u][j], acc[i * NB + j]);
    }
    // 8 -> 4 -> 2 -> 1 waves
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
        if (wv >= half && wv < 2 * half) {
#pragma unroll
            for (int q = 0; q < NACC; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[wv - half][(q * 16 + e) * 64 + l] = acc[q][e];
        }
        __syncthreads();
        if (wv < half) {
#pragma unroll
            for (int q = 0; q < NACC; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[q][e] += red[wv][(q * 16 + e) * 64 + l];
        }
        __syncthreads();
    }
    if (wv == 0) {
        float* o = dwp + (long)blockIdx.x * COUT * CIN;
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);      // co within the block
                    o[(long)(32 * i + row) * CIN + 32 * j + cl] = acc[i * NB + j][e];
                }
    }
}

// stage 2: dw[idx] (=|+=) sum_part dwp[part][idx]; a workgroup owns 32 outputs (8 float4 lanes) and deals the partials to its
// 32 thread groups (all loads of a thread in flight together), then sums the groups in a fixed order through LDS
constexpr int PW_PARTS_MAX = 256;
__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int n,
                                                                   int parts, int accumulate) {
    __shared__ __attribute__((aligned(16))) float sred[32][32];
    const int tid = (int)threadIdx.x, f4 = tid & 7, sub = tid >> 3;
    const long base = (long)blockIdx.x * 32 + 4 * f4;
    float4 v[PW_PARTS_MAX / 32];
#pragma unroll
    for (int u = 0; u < PW_PARTS_MAX / 32; ++u) {
        const int part = sub + 32 * u;
        v[u] = *reinterpret_cast<const float4*>(dwp + (long)(part < parts ? part : 0) * n + base);
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < PW_PARTS_MAX / 32; ++u)
        if (sub + 32 * u < parts) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    *reinterpret_cast<float4*>(&sred[sub][4 * f4]) = s;
    __syncthreads();
    if (tid < 32) {
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += sred[k][tid];
        const long idx = (long)blockIdx.x * 32 + tid;
        dw[idx] = accumulate ? dw[idx] + t : t;
    }
}

}  // namespace

static int conv1x1_launch(const float* in, const float* w, float* out, float* bn_partial, long P, int Cin, int Cout, int w_transposed,
                          int accumulate, void* stream, BnBwdFuse fuse = BnBwdFuse{}) {
    const dim3 grid((unsigned)lv_cdiv(P, PWP)), block(PWT);
    if (Cin == 64 && Cout == 32) LV_LAUNCH((conv1x1_kernel<64, 32>), grid, block, 0, stream, in, w, out, bn_partial, P, w_transposed, accumulate, fuse);
    else if (Cin == 32 && Cout == 64) LV_LAUNCH((conv1x1_kernel<32, 64>), grid, block, 0, stream, in, w, out, bn_partial, P, w_transposed, accumulate, fuse);
    else if (Cin == 64 && Cout == 64) LV_LAUNCH((conv1x1_kernel<64, 64>), grid, block, 0, stream, in, w, out, bn_partial, P, w_transposed, accumulate, fuse);
    else if (Cin == 32 && Cout == 32) LV_LAUNCH((conv1x1_kernel<32, 32>), grid, block, 0, stream, in, w, out, bn_partial, P, w_transposed, accumulate, fuse);
    else return LV_ERR_UNSUPPORTED;
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// out [P][Cout] (=|+=) in [P][Cin] . W^T with W [Cout][Cin] (w_transposed = 0) or given as [Cin][Cout] (w_transposed = 1: the
// data gradient of the forward convolution whose weight this is).  (Cin, Cout) in {32, 64}^2; else LV_ERR_UNSUPPORTED.
extern "C" int lv_conv1x1_f32(const float* in, const float* w, float* out, long P, int Cin, int Cout, int w_transposed,
                              int accumulate, void* stream) {
    if (!in || !w || !out) return LV_ERR_ARG;
    if (P <= 0) return LV_ERR_SHAPE;
    if ((((uintptr_t)in) & 15) != 0) return LV_ERR_ALIGN;
    return conv1x1_launch(in, w, out, nullptr, P, Cin, Cout, w_transposed, accumulate, stream);
}

// forward that also leaves the following BatchNorm's stage-1 partials: bn_partial [lv_conv1x1_blocks(P)][2][Cout]
extern "C" long lv_conv1x1_blocks(long P) { return lv_cdiv(P, PWP); }
extern "C" int lv_conv1x1_bnstat_f32(const float* in, const float* w, float* out, float* bn_partial, long P, int Cin, int Cout,
                                     void* stream) {
    if (!in || !w || !out || !bn_partial) return LV_ERR_ARG;
    if (P <= 0) return LV_ERR_SHAPE;
    if ((((uintptr_t)in) & 15) != 0) return LV_ERR_ALIGN;
    return conv1x1_launch(in, w, out, bn_partial, P, Cin, Cout, 0, 0, stream);
}

// the pointwise convolution's DATA GRADIENT with stage 1 of the preceding BatchNorm's backward in its epilogue (see lv_conv32_bnbwd):
// in = dL/d(conv output) [P][Cin], w = the forward's weight as stored ([Cin][Cout] from this call's point of view), dv [P][Cout],
// partial [lv_conv1x1_blocks(P)][2][Cout]
extern "C" int lv_conv1x1_bnbwd_f32(const float* in, const float* w, float* dv, float* partial, long P, int Cin, int Cout, const float* y,
                                    const float* x, const float* mean, const float* invstd, int act_elu, void* stream) {
    if (!in || !w || !dv || !partial || !x || !mean || !invstd || (act_elu && !y)) return LV_ERR_ARG;
    if (P <= 0) return LV_ERR_SHAPE;
    if ((((uintptr_t)in) & 15) != 0) return LV_ERR_ALIGN;
    return conv1x1_launch(in, w, dv, partial, P, Cin, Cout, 1, 0, stream, BnBwdFuse{y, x, mean, invstd, act_elu});
}

extern "C" long lv_conv1x1_wgrad_ws_floats(int Cin, int Cout) { return (long)PW_PARTS_MAX * Cin * Cout; }

// dw [Cout][Cin] (=|+=) dy^T . x over P pixels; ws: lv_conv1x1_wgrad_ws_floats floats
static inline long conv1x1_wgrad_ppw(long P) {
    // an even number of pixels per wave (a pixel pair per MFMA), at least one batch each, at most PW_PARTS_MAX workgroups
    long ppw = lv_cdiv(P, (long)PW_PARTS_MAX * 8);
    if (ppw < 2 * PW_U) ppw = 2 * PW_U;
    return (ppw + 1) / 2 * 2;
}
// partial blocks lv_conv1x1_wgrad_f32 leaves in ws ([parts][Cout][Cin])
extern "C" int lv_conv1x1_wgrad_parts(long P) { return P > 0 ? (int)lv_cdiv(P, conv1x1_wgrad_ppw(P) * 8) : 0; }
// dw == NULL: stage 1 only -- the partials stay in ws for a later lv_wgrad_reduce_batched
extern "C" int lv_conv1x1_wgrad_f32(const float* x, const float* dy, float* dw, float* ws, long P, int Cin, int Cout, int accumulate,
                                    void* stream) {
    if (!x || !dy || !ws) return LV_ERR_ARG;
    if (P <= 0) return LV_ERR_SHAPE;
    const long ppw = conv1x1_wgrad_ppw(P);
    const int wgs = (int)lv_cdiv(P, ppw * 8);
    const dim3 grid((unsigned)wgs), block(512);
    if (Cin == 64 && Cout == 32) LV_LAUNCH((conv1x1_wgrad_kernel<64, 32>), grid, block, 0, stream, x, dy, ws, P, ppw);
    else if (Cin == 32 && Cout == 64) LV_LAUNCH((conv1x1_wgrad_kernel<32, 64>), grid, block, 0, stream, x, dy, ws, P, ppw);
    else if (Cin == 64 && Cout == 64) LV_LAUNCH((conv1x1_wgrad_kernel<64, 64>), grid, block, 0, stream, x, dy, ws, P, ppw);
    else if (Cin == 32 && Cout == 32) LV_LAUNCH((conv1x1_wgrad_kernel<32, 32>), grid, block, 0, stream, x, dy, ws, P, ppw);
    else return LV_ERR_UNSUPPORTED;
    if (dw)
        LV_LAUNCH(conv1x1_wgrad_reduce_kernel, dim3((unsigned)(Cin * Cout / 32)), dim3(256), 0, stream, (const float*)ws, dw, Cin * Cout,
                  wgs, accumulate);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// ---- all weight-gradient reductions of a backward pass in ONE launch ------------------------------------------------------
// The decoder's 23 + 47 stage-2 reductions are 5-9 us launches of a few dozen workgroups each (0.43 ms per Omniglot step);
// nothing reads the reduced gradients before the clip norm at the end of the backward, so the stage-1 kernels leave their
// partials in per-layer scratch and one kernel sums them all.  Descriptors travel as kernel arguments (no table in memory, no
// host-to-device copy: capturable in a hipGraph).  Same sums in the same order as the per-layer kernels.
namespace {
constexpr int WB_MAX = 96;
struct WgradDesc { const float* part; float* dst; int n; int parts; int kk; int pad; };       // kk = 0: pointwise layout; else k*k
struct WgradBatch { int ndesc; int blk0[WB_MAX + 1]; WgradDesc d[WB_MAX]; };

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(WgradBatch bt) {
    __shared__ __attribute__((aligned(16))) float sred[32][32];
    const int b = (int)blockIdx.x;
    int lo = 0, hi = bt.ndesc;                     // blk0[lo] <= b < blk0[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (b >= bt.blk0[mid]) lo = mid; else hi = mid;
    }
    const WgradDesc d = bt.d[lo];
    const int tid = (int)threadIdx.x, f4 = tid & 7, sub = tid >> 3;
    const long base = (long)(b - bt.blk0[lo]) * 32 + 4 * f4;
    float4 v[PW_PARTS_MAX / 32];
#pragma unroll
    for (int u = 0; u < PW_PARTS_MAX / 32; ++u) {
        const int part = sub + 32 * u;
        v[u] = *reinterpret_cast<const float4*>(d.part + (long)(part < d.parts ? part : 0) * d.n + base);
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < PW_PARTS_MAX / 32; ++u)
        if (sub + 32 * u < d.parts) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    *reinterpret_cast<float4*>(&sred[sub][4 * f4]) = s;
    __syncthreads();
    if (tid < 32) {
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += sred[k][tid];
        const long idx = (long)(b - bt.blk0[lo]) * 32 + tid;
        if (d.kk == 0) d.dst[idx] = t;
        else {                                     // partial layout [t][ci][co] -> the reference's [co][ci][t]
            const int co = (int)(idx % CC), ci = (int)((idx / CC) % CC), tt = (int)(idx / (CC * CC));
            d.dst[((long)co * CC + ci) * d.kk + tt] = t;
        }
    }
}
}  // namespace

// desc: 4 x int64 per entry on the HOST: {partials pointer, destination pointer, n (outputs, a multiple of 32) | parts << 32,
// k*k for a 32 -> 32 convolution (partials [parts][k*k][ci][co], destination [co][ci][k*k]) or 0 (pointwise: both [.][n])}
extern "C" int lv_wgrad_reduce_batched(const long long* desc, int ndesc, void* stream) {
    if (ndesc < 0 || (ndesc > 0 && !desc)) return LV_ERR_ARG;
    for (int i0 = 0; i0 < ndesc; i0 += WB_MAX) {
        WgradBatch bt;
        const int m = ndesc - i0 < WB_MAX ? ndesc - i0 : WB_MAX;
        bt.ndesc = m;
        int blocks = 0;
        for (int i = 0; i < m; ++i) {
            const long long* e = desc + 4L * (i0 + i);
            WgradDesc& d = bt.d[i];
            d.part = reinterpret_cast<const float*>((uintptr_t)e[0]);
            d.dst = reinterpret_cast<float*>((uintptr_t)e[1]);
            d.n = (int)(e[2] & 0xFFFFFFFFll);
            d.parts = (int)(e[2] >> 32);
            d.kk = (int)e[3];
            d.pad = 0;
            if (!d.part || !d.dst) return LV_ERR_ARG;
            if (d.n <= 0 || d.n % 32 != 0 || d.parts <= 0 || d.parts > PW_PARTS_MAX || d.kk < 0 || (d.kk > 0 && d.n != d.kk * CC * CC))
                return LV_ERR_SHAPE;
            if ((((uintptr_t)d.part) & 15) != 0) return LV_ERR_ALIGN;
            bt.blk0[i] = blocks;
            blocks += d.n / 32;
        }
        bt.blk0[m] = blocks;
        LV_LAUNCH(wgrad_reduce_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, bt);
    }
    LV_CHECK_LAUNCH();
    return LV_OK;
}
