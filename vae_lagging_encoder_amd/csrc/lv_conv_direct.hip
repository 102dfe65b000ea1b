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

constexpr int CC = 32;                 // channels in and out
constexpr int IW = 28, IH = 28;        // feature map
constexpr int TR = 4;                  // image rows per workgroup (one per wave)
constexpr int PP = 36;                 // LDS pixel pitch in floats (16-byte aligned, 9 sixteen-byte slots: odd -> no conflicts)
constexpr int KMAX = 7;

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
__global__ __launch_bounds__(256) void conv32_direct_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                            float* __restrict__ out, int k, int ntaps, int mirror, int accumulate) {
    __shared__ __attribute__((aligned(16))) float halo[(TR + KMAX - 1) * (IW + KMAX - 1) * PP];
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int n = (int)blockIdx.x / (IH / TR), r0 = ((int)blockIdx.x % (IH / TR)) * TR;
    const int p = k / 2, HW = IW + 2 * p, HR = TR + 2 * p;
    // stage rows r0-p .. r0+TR-1+p, columns -p .. IW-1+p (zero outside the image)
    for (int i = tid; i < HR * HW * (CC / 4); i += 256) {
        const int c4 = i % (CC / 4), hx = (i / (CC / 4)) % HW, hy = i / ((CC / 4) * HW);
        const int gy = r0 - p + hy, gx = hx - p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < IH && gx >= 0 && gx < IW)
            v = *reinterpret_cast<const float4*>(in + (((long)n * IH + gy) * IW + gx) * CC + 4 * c4);
        *reinterpret_cast<float4*>(&halo[(hy * HW + hx) * PP + 4 * c4]) = v;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int px = (l & 31) < IW ? (l & 31) : IW - 1;        // lanes 28..31 shadow pixel 27: their rows of the result are dropped
    const int kh = l >> 5;
    for (int t = 0; t < ntaps; ++t) {
        int dy = t / k - p, dx = t % k - p;
        if (mirror) { dy = -dy; dx = -dx; }
        const float* a = &halo[((w + p + dy) * HW + (px + p + dx)) * PP + 16 * kh];
        const float4 a0 = *reinterpret_cast<const float4*>(a), a1 = *reinterpret_cast<const float4*>(a + 4);
        const float4 a2 = *reinterpret_cast<const float4*>(a + 8), a3 = *reinterpret_cast<const float4*>(a + 12);
        const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
        const float* bw = wp + (long)t * 16 * 64 + l;
        float bv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) bv[s] = bw[s * 64];
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = lv_mfma_32x32x2(av[s], bv[s], acc);
    }
    const int r = r0 + w;
    const int col = l & 31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int x = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
        if (x < IW) {
            float* o = out + (((long)n * IH + r) * IW + x) * CC + col;
            *o = accumulate ? *o + acc[e] : acc[e];
        }
    }
}

// weight gradient, stage 1.  grid (slabs, 4 tap groups); slab = a contiguous range of 4-row tiles; wave w of group g owns
// taps t = 4 * j + w ... (taps dealt round-robin over the 16 waves of the 4 groups): dwp[slab][t][ci][co] = sum over the slab's
// pixels of x[pixel + off_t][ci] * dy[pixel][co].
constexpr int WG_TAPS = 4;             // taps per wave at most (k = 7: 49 taps over 16 waves -> 4)
__global__ __launch_bounds__(256) void conv32_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dwp, int N, int k, int tiles_per_slab) {
    __shared__ __attribute__((aligned(16))) float halo[(TR + KMAX - 1) * (IW + KMAX - 1) * PP];
    __shared__ __attribute__((aligned(16))) float gy[TR * IW * PP];
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int KK = k * k, p = k / 2, HW = IW + 2 * p, HR = TR + 2 * p;
    const int gw = (int)blockIdx.y * 4 + w;                  // wave index among the 16 that share the taps
    f32x16 acc[WG_TAPS];
#pragma unroll
    for (int q = 0; q < WG_TAPS; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    const int ntiles = N * (IH / TR);
    const int t0 = (int)blockIdx.x * tiles_per_slab;
    const int t1 = t0 + tiles_per_slab < ntiles ? t0 + tiles_per_slab : ntiles;
    const int ci = l & 31, kp = l >> 5;
    for (int tile = t0; tile < t1; ++tile) {
        const int n = tile / (IH / TR), r0 = (tile % (IH / TR)) * TR;
        __syncthreads();
        for (int i = tid; i < HR * HW * (CC / 4); i += 256) {
            const int c4 = i % (CC / 4), hx = (i / (CC / 4)) % HW, hy = i / ((CC / 4) * HW);
            const int yy = r0 - p + hy, xx = hx - p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < IH && xx >= 0 && xx < IW)
                v = *reinterpret_cast<const float4*>(x + (((long)n * IH + yy) * IW + xx) * CC + 4 * c4);
            *reinterpret_cast<float4*>(&halo[(hy * HW + hx) * PP + 4 * c4]) = v;
        }
        for (int i = tid; i < TR * IW * (CC / 4); i += 256) {
            const int c4 = i % (CC / 4), pix = i / (CC / 4);
            *reinterpret_cast<float4*>(&gy[pix * PP + 4 * c4]) =
                *reinterpret_cast<const float4*>(dy + (((long)n * IH + r0) * IW + pix) * CC + 4 * c4);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < WG_TAPS; ++q) {
            const int t = gw + 16 * q;
            if (t >= KK) continue;
            const int dyo = t / k - p, dxo = t % k - p;
            for (int m = 0; m < TR * IW / 2; ++m) {          // two pixels per MFMA: pixel 2m + kp
                const int pix = 2 * m + kp, py = pix / IW, pxx = pix % IW;
                const float a = halo[((py + p + dyo) * HW + (pxx + p + dxo)) * PP + ci];
                const float b = gy[pix * PP + ci];
                acc[q] = lv_mfma_32x32x2(a, b, acc[q]);      // D[ci][co] += x[pixel + off][ci] * dy[pixel][co]
            }
        }
    }
    float* outp = dwp + (long)blockIdx.x * KK * CC * CC;
#pragma unroll
    for (int q = 0; q < WG_TAPS; ++q) {
        const int t = gw + 16 * q;
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
extern "C" int lv_conv32_wgrad_slabs(int N) {
    const int ntiles = N * (IH / TR);
    return ntiles < 64 ? ntiles : 64;
}
extern "C" long lv_conv32_wgrad_ws_floats(int N, int k) { return (long)lv_conv32_wgrad_slabs(N) * k * k * CC * CC; }

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
    LV_LAUNCH(conv32_direct_kernel, dim3((unsigned)(N * (IH / TR))), dim3(256), 0, stream, in, wp, out, k, ntaps, mirror, accumulate);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// weight gradient over ALL k*k taps: dw [32][32][k*k] (=|+=) sum_pixels dy[p][co] x[p + off_t][ci]; ws: lv_conv32_wgrad_ws_floats
extern "C" int lv_conv32_wgrad_f32(const float* x, const float* dy, float* dw, float* ws, int N, int k, int accumulate, void* stream) {
    if (!x || !dy || !dw || !ws) return LV_ERR_ARG;
    if (N <= 0 || k <= 0 || k > KMAX || !(k & 1)) return LV_ERR_SHAPE;
    if (((((uintptr_t)x) | ((uintptr_t)dy)) & 15) != 0) return LV_ERR_ALIGN;
    const int ntiles = N * (IH / TR);
    const int slabs = lv_conv32_wgrad_slabs(N);
    const int tps = lv_cdiv(ntiles, slabs);
    const int used = lv_cdiv(ntiles, tps);
    LV_LAUNCH(conv32_wgrad_kernel, dim3((unsigned)used, 4), dim3(256), 0, stream, x, dy, ws, N, k, tps);
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

constexpr int PWP = 128;               // pixels per workgroup

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv1x1_kernel(const float* __restrict__ in, const float* __restrict__ w, float* __restrict__ out,
                                                      long P, int w_transposed, int accumulate) {
    constexpr int PA = CIN + 4;        // LDS pitches (floats): 16-byte slots per row odd -> conflict-free ds_read_b128
    __shared__ __attribute__((aligned(16))) float sa[PWP * PA];
    __shared__ __attribute__((aligned(16))) float sw[COUT * PA];
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    const long p0 = (long)blockIdx.x * PWP;
    for (int i = tid; i < PWP * (CIN / 4); i += 256) {
        const int c4 = i % (CIN / 4), pp = i / (CIN / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + pp < P) v = *reinterpret_cast<const float4*>(in + (p0 + pp) * CIN + 4 * c4);
        *reinterpret_cast<float4*>(&sa[pp * PA + 4 * c4]) = v;
    }
    // sw[co][ci] = W[co][ci]  (w stored [COUT][CIN]), or W^T when the caller hands the [CIN][COUT] matrix of the forward
    for (int i = tid; i < COUT * CIN; i += 256) {
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
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long pp = p0 + wv * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
            if (pp < P) {
                float* o = out + pp * COUT + nb * 32 + (l & 31);
                *o = accumulate ? *o + acc[e] : acc[e];
            }
        }
    }
}

// stage 1 of the weight gradient: dwp[slab][co][ci] = sum over the slab's pixels of dy[p][co] x[p][ci]; the (COUT/32) x (CIN/32)
// output blocks are dealt to the 4 waves (at most 4 blocks)
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dwp, long P, long pix_per_slab) {
    constexpr int PX = CIN + 1, PY = COUT + 1;          // scalar LDS reads along the channel index: odd pitch
    __shared__ float sx[PWP * PX];
    __shared__ float sy[PWP * PY];
    const int tid = (int)threadIdx.x, l = tid & 63, wv = tid >> 6;
    constexpr int NBLK = (COUT / 32) * (CIN / 32);
    const int blk = wv % NBLK, ksplit = wv / NBLK;       // waves beyond NBLK split the pixel range of a block
    constexpr int KS = 4 / NBLK > 0 ? 4 / NBLK : 1;
    const int cob = blk / (CIN / 32), cib = blk % (CIN / 32);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const long s0 = (long)blockIdx.x * pix_per_slab;
    const long s1 = s0 + pix_per_slab < P ? s0 + pix_per_slab : P;
    const int kp = l >> 5, cl = l & 31;
    for (long p0 = s0; p0 < s1; p0 += PWP) {
        __syncthreads();
        for (int i = tid; i < PWP * CIN; i += 256) {
            const int c = i % CIN, pp = i / CIN;
            sx[pp * PX + c] = p0 + pp < s1 ? x[(p0 + pp) * CIN + c] : 0.f;
        }
        for (int i = tid; i < PWP * COUT; i += 256) {
            const int c = i % COUT, pp = i / COUT;
            sy[pp * PY + c] = p0 + pp < s1 ? dy[(p0 + pp) * COUT + c] : 0.f;
        }
        __syncthreads();
        if (wv < NBLK * KS) {
            // D[co][ci] += dy[p][co] * x[p][ci], two pixels per MFMA; this wave's share of the 128 staged pixels
            const int m0 = ksplit * (PWP / 2 / KS), m1 = m0 + PWP / 2 / KS;
            for (int m = m0; m < m1; ++m) {
                const int pp = 2 * m + kp;
                acc = lv_mfma_32x32x2(sy[pp * PY + cob * 32 + cl], sx[pp * PX + cib * 32 + cl], acc);
            }
        }
    }
    // partial layout [slab][KS][COUT][CIN]
    if (wv < NBLK * KS) {
        float* o = dwp + ((long)blockIdx.x * KS + ksplit) * COUT * CIN;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);      // co within the block
            o[(long)(cob * 32 + row) * CIN + cib * 32 + cl] = acc[e];
        }
    }
}

__global__ __launch_bounds__(256) void conv1x1_wgrad_reduce_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int n,
                                                                   int parts, int accumulate) {
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (idx >= n) return;
    float s = 0.f;
    for (int b0 = 0; b0 < parts; b0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = dwp[(long)(b0 + u < parts ? b0 + u : 0) * n + idx];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += b0 + u < parts ? v[u] : 0.f;
    }
    dw[idx] = accumulate ? dw[idx] + s : s;
}

constexpr int PW_SLABS = 128;

}  // namespace

// out [P][Cout] (=|+=) in [P][Cin] . W^T with W [Cout][Cin] (w_transposed = 0) or given as [Cin][Cout] (w_transposed = 1: the
// data gradient of the forward convolution whose weight this is).  (Cin, Cout) in {32, 64}^2; else LV_ERR_UNSUPPORTED.
extern "C" int lv_conv1x1_f32(const float* in, const float* w, float* out, long P, int Cin, int Cout, int w_transposed,
                              int accumulate, void* stream) {
    if (!in || !w || !out) return LV_ERR_ARG;
    if (P <= 0) return LV_ERR_SHAPE;
    if ((((uintptr_t)in) & 15) != 0) return LV_ERR_ALIGN;
    const dim3 grid((unsigned)lv_cdiv(P, PWP)), block(256);
    if (Cin == 64 && Cout == 32) LV_LAUNCH((conv1x1_kernel<64, 32>), grid, block, 0, stream, in, w, out, P, w_transposed, accumulate);
    else if (Cin == 32 && Cout == 64) LV_LAUNCH((conv1x1_kernel<32, 64>), grid, block, 0, stream, in, w, out, P, w_transposed, accumulate);
    else if (Cin == 64 && Cout == 64) LV_LAUNCH((conv1x1_kernel<64, 64>), grid, block, 0, stream, in, w, out, P, w_transposed, accumulate);
    else if (Cin == 32 && Cout == 32) LV_LAUNCH((conv1x1_kernel<32, 32>), grid, block, 0, stream, in, w, out, P, w_transposed, accumulate);
    else return LV_ERR_UNSUPPORTED;
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" long lv_conv1x1_wgrad_ws_floats(int Cin, int Cout) { return (long)PW_SLABS * 4 * Cin * Cout; }

// dw [Cout][Cin] (=|+=) dy^T . x over P pixels; ws: lv_conv1x1_wgrad_ws_floats floats
extern "C" int lv_conv1x1_wgrad_f32(const float* x, const float* dy, float* dw, float* ws, long P, int Cin, int Cout, int accumulate,
                                    void* stream) {
    if (!x || !dy || !dw || !ws) return LV_ERR_ARG;
    if (P <= 0) return LV_ERR_SHAPE;
    long per = lv_cdiv(P, PW_SLABS);
    per = (per + PWP - 1) / PWP * PWP;
    const int slabs = lv_cdiv(P, per);
    const dim3 grid((unsigned)slabs), block(256);
    int ks;
    if (Cin == 64 && Cout == 32) { ks = 2; LV_LAUNCH((conv1x1_wgrad_kernel<64, 32>), grid, block, 0, stream, x, dy, ws, P, per); }
    else if (Cin == 32 && Cout == 64) { ks = 2; LV_LAUNCH((conv1x1_wgrad_kernel<32, 64>), grid, block, 0, stream, x, dy, ws, P, per); }
    else if (Cin == 64 && Cout == 64) { ks = 1; LV_LAUNCH((conv1x1_wgrad_kernel<64, 64>), grid, block, 0, stream, x, dy, ws, P, per); }
    else if (Cin == 32 && Cout == 32) { ks = 4; LV_LAUNCH((conv1x1_wgrad_kernel<32, 32>), grid, block, 0, stream, x, dy, ws, P, per); }
    else return LV_ERR_UNSUPPORTED;
    LV_LAUNCH(conv1x1_wgrad_reduce_kernel, dim3((unsigned)lv_cdiv((long)Cin * Cout, 256)), dim3(256), 0, stream, (const float*)ws, dw,
              Cin * Cout, slabs * ks, accumulate);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
