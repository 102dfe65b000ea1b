// lv_pixelcnn_sample.hip -- ancestral sampling from the PixelCNN decoder one PIXEL at a time (SURVEY.md 8f row 4;
// reference modules/decoders/dec_pixelcnn_v2.py:201-232: 784 sequential full forwards of the 82-convolution network).
//
// The network is causal: the activations at raster positions <= p never depend on the image at positions >= p.  So after
// pixel p - 1 has been drawn, the probability of pixel p needs every layer's output at ONE position -- p -- given the cached
// outputs of the positions before it: the only layers that look at neighbours are the MaskA convolution (the image + latent
// maps, which are the input itself) and the 23 masked 32 -> 32 convolutions of the residual blocks, whose INPUT maps (after
// BatchNorm + ELU) are kept in HBM and extended by one position per step.  One launch per pixel (the kernel boundary orders
// a step's reads behind the previous step's writes), one single-wave workgroup per image, lane = channel.
//
// BIT-EQUAL to the full forward of the evaluation path (image_engine.decoder_forward in eval mode): every output element is
// produced by the same f32 fma chain, in the same order, as the kernel that computes it there --
//   * generic GEMM convolutions (MaskA 5 -> 64 over 49 taps; head 64 -> 1): lv_gemm_f32's chain, k = tap * Cin + c ascending
//     (v_mfma_f32_32x32x2_f32 = two chained fmas, k then k + 1), cut into the split-K pieces lv_gemm_f32 would cut at this
//     batch size and summed slab by slab;
//   * pointwise convolutions (conv1x1_kernel<CIN, COUT>): for s < CIN / 2: channel s, then channel CIN / 2 + s;
//   * masked 32 -> 32 convolutions (conv32_direct_kernel<KS>): taps in raster order, per tap for s < 16: channel s, then
//     channel 16 + s; with the tap-split form (KS = 2, chosen by batch size) the two tap halves are separate chains, summed;
//   * eval-mode BatchNorm (+ residual) (+ ELU) with the expression of bn_eval_kernel, the residual adds of the direct
//     connections as lv_add_f32 (a + b).
// tests: the probabilities of every pixel and the sampled images equal the 784-full-forward path bit for bit.
#include "lv_device.h"

namespace {

struct PixBn { const float* gamma; const float* beta; const float* rmean; const float* rvar; double eps; };
struct PixBlock {
    const float* w1t;       // [64][32]: W1[co][ci] stored ci-major (lane = co reads coalesced)
    PixBn bn1;
    const float* wp;        // conv32_pack_kernel image of the masked convolution (forward, taps < ntaps)
    long long k, ntaps;
    PixBn bn2;
    const float* w3t;       // [32][64]
    PixBn bn3;
    float* a1;              // [B][28][28][32]: the masked convolution's input map, extended by one position per step
};
struct PixNet {
    const float* in5;       // [B][784][5]: image channel (filled in as the pixels are drawn) + the 4 latent maps
    const float* wAt;       // [245][64]: MaskA weights, k = tap * 5 + c major (already multiplied by the mask)
    PixBn bnA;
    long long splitA_k;     // first k of the second split-K piece of the MaskA GEMM at this batch size (0: one piece)
    const PixBlock* main;   // 12 residual blocks
    const PixBlock* dc;     // 11 direct-connection blocks
    const float* c1t;       // [64][64] head 1x1, ci-major
    PixBn bnH;
    const float* c2;        // [64] head 64 -> 1
    float* logit;           // [B][784]
    long long ks;           // tap-split form of the masked convolutions at this batch size (conv32_ks)
    long long nmain, ndc;
};

constexpr int IWS = 28;

__device__ __forceinline__ float bn_eval(float v, const PixBn& bn, int c, bool has_res, float res, bool act) {
    float o = (v - bn.rmean[c]) * (1.0f / sqrtf(bn.rvar[c] + (float)bn.eps)) * bn.gamma[c] + bn.beta[c];
    if (has_res) o += res;
    if (act) o = o > 0.f ? o : expm1f(o);
    return o;
}

// one residual block at position (i, j) of image b: x (64 channels, in LDS) -> block output (returned per lane = channel)
__device__ float pix_block(const PixBlock& bk, const float* xin, float* s_a, float* s_win, int b, int i, int j, int ks, int l) {
    const int co32 = l & 31, half = l >> 5;
    // 1 x 1, 64 -> 32 (conv1x1_kernel<64, 32>), BatchNorm + ELU
    float acc = 0.f;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
        acc = fmaf(xin[s], bk.w1t[s * 32 + co32], acc);
        acc = fmaf(xin[32 + s], bk.w1t[(32 + s) * 32 + co32], acc);
    }
    const float a1 = bn_eval(acc, bk.bn1, co32, false, 0.f, true);
    const int k = (int)bk.k, ntaps = (int)bk.ntaps, p = k / 2;
    float* a1map = bk.a1 + (long)b * IWS * IWS * 32;
    __syncthreads();
    if (half == 0) { a1map[(i * IWS + j) * 32 + co32] = a1; s_a[co32] = a1; }
    __syncthreads();
    // the taps' inputs: cached positions (zero outside the image), the centre (the last kept tap) from this step
    for (int e = l; e < ntaps * 32; e += 64) {
        const int t = e >> 5, c = e & 31;
        const int y = i + t / k - p, x = j + t % k - p;
        float v = 0.f;
        if (t == ntaps - 1) v = s_a[c];
        else if (y >= 0 && y < IWS && x >= 0 && x < IWS) v = a1map[(y * IWS + x) * 32 + c];
        s_win[e] = v;
    }
    __syncthreads();
    // masked k x k, 32 -> 32 (conv32_direct_kernel<KS>)
    const int hsplit = (ntaps + 1) / 2;
    const int t_lo = (ks == 2 && half == 1) ? hsplit : 0;
    const int t_hi = (ks == 2 && half == 0) ? hsplit : ntaps;
    acc = 0.f;
    for (int t = t_lo; t < t_hi; ++t) {
        const float* xw = s_win + t * 32;
        const float* ww = bk.wp + (long)t * 16 * 64 + co32;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            acc = fmaf(xw[s], ww[s * 64], acc);
            acc = fmaf(xw[16 + s], ww[s * 64 + 32], acc);
        }
    }
    if (ks == 2) acc = __shfl(acc, co32, 64) + __shfl(acc, co32 + 32, 64);     // tap half 0 + tap half 1
    const float a2 = bn_eval(acc, bk.bn2, co32, false, 0.f, true);
    __syncthreads();
    if (half == 0) s_a[co32] = a2;
    __syncthreads();
    // 1 x 1, 32 -> 64 (conv1x1_kernel<32, 64>), BatchNorm + residual (the block's input) + ELU
    acc = 0.f;
#pragma unroll 8
    for (int s = 0; s < 16; ++s) {
        acc = fmaf(s_a[s], bk.w3t[s * 64 + l], acc);
        acc = fmaf(s_a[16 + s], bk.w3t[(16 + s) * 64 + l], acc);
    }
    const float out = bn_eval(acc, bk.bn3, l, true, xin[l], true);
    __syncthreads();
    return out;
}

__global__ __launch_bounds__(64) void pixelcnn_pixel_step_kernel(PixNet net, int i, int j) {
    __shared__ float s_x[4][64];        // inp and the three pending direct-connection inputs
    __shared__ float s_a[64];
    __shared__ float s_win[49 * 32];
    __shared__ float s_in5[245 + 11];
    const int l = (int)threadIdx.x, b = (int)blockIdx.x;
    // ---- MaskA: 7 x 7 over (image, 4 latent maps), 64 outputs; lv_gemm_f32's chain over k = tap * 5 + c -------------------------
    const float* in5 = net.in5 + (long)b * IWS * IWS * 5;
    for (int e = l; e < 256; e += 64) {
        float v = 0.f;
        if (e < 245) {
            const int t = e / 5, c = e % 5;
            const int y = i + t / 7 - 3, x = j + t % 7 - 3;
            if (y >= 0 && y < IWS && x >= 0 && x < IWS) v = in5[(y * IWS + x) * 5 + c];
        }
        s_in5[e] = v;
    }
    __syncthreads();
    float acc;
    {
        const int kb = (int)net.splitA_k;
        float p0 = 0.f, p1 = 0.f;
        const int k0_end = kb > 0 ? kb : 245;
        for (int k = 0; k < k0_end; ++k) p0 = fmaf(s_in5[k], net.wAt[k * 64 + l], p0);
        if (kb > 0) {
            for (int k = kb; k < 245; ++k) p1 = fmaf(s_in5[k], net.wAt[k * 64 + l], p1);
            float s = 0.f;                   // splitk_reduce_kernel: slab by slab
            s += p0;
            s += p1;
            acc = s;
        } else {
            acc = p0;
        }
    }
    float inp = bn_eval(acc, net.bnA, l, false, 0.f, true);
    // ---- the residual stack with its direct connections (image_engine.decoder_forward) ------------------------------------------
    // slots: s_x[0] = inp; the queue of direct-connection inputs cycles through s_x[1..3]
    float qv[3];                         // the queue's values for this lane's channel (oldest first)
    int qn = 0;
    qv[0] = inp; qn = 1;
    const int nmain = (int)net.nmain;
    for (int m = 0; m < nmain; ++m) {    // main[m + 1] of the reference's ModuleList
        if (m + 1 > 2) {
            const float di = qv[0];
            qv[0] = qv[1]; qv[1] = qv[2]; --qn;
            s_x[1][l] = di;
            __syncthreads();
            const float d = pix_block(net.dc[m - 2], s_x[1], s_a, s_win, b, i, j, (int)net.ks, l);
            inp = inp + d;               // lv_add_f32(inp, block)
        }
        s_x[0][l] = inp;
        __syncthreads();
        inp = pix_block(net.main[m], s_x[0], s_a, s_win, b, i, j, (int)net.ks, l);
        qv[qn++] = inp;
    }
    {
        s_x[1][l] = qv[0];
        __syncthreads();
        const float d = pix_block(net.dc[net.ndc - 1], s_x[1], s_a, s_win, b, i, j, (int)net.ks, l);
        inp = inp + d;
    }
    // ---- head: 1 x 1 64 -> 64, BatchNorm + ELU, 1 x 1 64 -> 1 ----------------------------------------------------------------------
    s_x[0][l] = inp;
    __syncthreads();
    acc = 0.f;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
        acc = fmaf(s_x[0][s], net.c1t[s * 64 + l], acc);
        acc = fmaf(s_x[0][32 + s], net.c1t[(32 + s) * 64 + l], acc);
    }
    const float hh = bn_eval(acc, net.bnH, l, false, 0.f, true);
    __syncthreads();
    s_a[l] = hh;
    __syncthreads();
    if (l == 0) {
        float o = 0.f;
        for (int k = 0; k < 64; ++k) o = fmaf(s_a[k], net.c2[k], o);
        net.logit[(long)b * IWS * IWS + i * IWS + j] = o;
    }
}

}  // namespace

// One position (i, j) of PixelCNNDecoderV2's forward for every image of the batch, given the cached maps of the earlier
// positions (see the file header).  net: HOST pointer to the PixNet words (sizeof(PixNet) / 8 64-bit words: device pointers,
// int64 and double fields in declaration order); the block tables it points to live in device memory.
extern "C" int lv_pixelcnn_net_words(void) { return (int)(sizeof(PixNet) / 8); }
extern "C" int lv_pixelcnn_block_words(void) { return (int)(sizeof(PixBlock) / 8); }
extern "C" int lv_pixelcnn_pixel_step_f32(const long long* net, int B, int i, int j, void* stream) {
    if (!net || B <= 0 || i < 0 || i >= IWS || j < 0 || j >= IWS) return LV_ERR_ARG;
    PixNet n;
    memcpy(&n, net, sizeof(PixNet));
    if (!n.in5 || !n.wAt || !n.main || !n.dc || !n.c1t || !n.c2 || !n.logit || n.nmain < 3 || n.ndc != n.nmain - 1) return LV_ERR_ARG;
    LV_LAUNCH(pixelcnn_pixel_step_kernel, dim3((unsigned)B), dim3(64), 0, stream, n, i, j);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
