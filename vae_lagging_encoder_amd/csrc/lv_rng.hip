// lv_rng.hip -- counter-based Philox4x32-10 generators for throughput mode.
//
// The reference draws eps with zeros_like(std).normal_() (modules/encoders/encoder.py:77) and dropout masks
// with nn.Dropout (modules/decoders/dec_lstm.py:81,106) from torch's global generator.  A GPU cannot replay
// torch's CPU mt19937 stream, so parity tests feed eps/masks as INPUTS and throughput
// mode generates statistically equivalent draws on device here.  The (seed, offset) state lives in device
// memory and is advanced by a kernel so a captured hipGraph draws fresh numbers on every replay.
#include "lv_device.h"

namespace {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    const uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c.x, hi0, lo0);
        mulhilo(0xCD9E8D57u, c.z, hi1, lo1);
        U4 n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

__device__ __forceinline__ U4 draw(const uint64_t* state, uint64_t substream, uint64_t i) {
    const uint64_t seed = state[0], off = state[1];
    U4 c;
    c.x = (uint32_t)i;
    c.y = (uint32_t)(i >> 32);
    c.z = (uint32_t)off ^ (uint32_t)(substream << 20);
    c.w = (uint32_t)(off >> 32) ^ (uint32_t)(substream >> 12);
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__global__ __launch_bounds__(256) void rng_normal_kernel(float* __restrict__ out, long n, const uint64_t* __restrict__ state,
                                                         uint64_t substream) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // 4 outputs per thread
    if (i * 4 >= n) return;
    const U4 r = draw(state, substream, (uint64_t)i);
    const float u1 = u01(r.x), u2 = u01(r.y), u3 = u01(r.z), u4 = u01(r.w);
    const float r1 = sqrtf(-2.f * logf(u1)), r2 = sqrtf(-2.f * logf(u3));
    const float t1 = 6.283185307179586f * u2, t2 = 6.283185307179586f * u4;
    const float v[4] = {r1 * cosf(t1), r1 * sinf(t1), r2 * cosf(t2), r2 * sinf(t2)};
    for (int j = 0; j < 4; ++j)
        if (i * 4 + j < n) out[i * 4 + j] = v[j];
}

__global__ __launch_bounds__(256) void rng_keepmask_kernel(uint8_t* __restrict__ out, long n, float keep,
                                                           const uint64_t* __restrict__ state, uint64_t substream) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // 8 outputs per thread
    if (i * 8 >= n) return;
    const U4 r = draw(state, substream, (uint64_t)i);
    const uint32_t thr = (uint32_t)(keep * 65536.0f);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    for (int j = 0; j < 8; ++j) {
        const uint32_t h = (w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
        if (i * 8 + j < n) out[i * 8 + j] = (h < thr) ? 1 : 0;
    }
}

// out[i] = 1 with probability p[i] (torch.bernoulli(batch) of image.py:287,318: dynamic binarisation)
__global__ __launch_bounds__(256) void rng_bernoulli_kernel(const float* __restrict__ p, float* __restrict__ out, long n,
                                                            const uint64_t* __restrict__ state, uint64_t substream) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // 4 outputs per thread
    if (i * 4 >= n) return;
    const U4 r = draw(state, substream, (uint64_t)i);
    const float u[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
    for (int j = 0; j < 4; ++j)
        if (i * 4 + j < n) out[i * 4 + j] = (u[j] < p[i * 4 + j]) ? 1.f : 0.f;
}

__global__ void rng_advance_kernel(uint64_t* state, uint64_t inc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += inc;
}

// All the noise of one VAE.loss call in ONE launch (same draws as rng_normal substream 0 + rng_keepmask substreams 1, 2
// followed by rng_advance): blocks [0, nb0) fill eps, [nb0, nb0+nb1) the dropout_in keep-mask, the rest the dropout_out
// keep-mask.  The offset is advanced by rng_advance_kernel, queued right behind it by the same entry point (a last-block
// ticket inside this kernel was measured at 59 us: ~2000 blocks serialising on one atomic).
struct NoiseP {
    float* eps; long n_eps;
    uint8_t* m1; long n1; float keep1;
    uint8_t* m2; long n2; float keep2;
    uint64_t* state; uint64_t inc;
    unsigned nb0, nb1;
};

__global__ __launch_bounds__(256) void rng_noise_step_kernel(NoiseP p) {
    const unsigned bid = blockIdx.x;
    const int tid = (int)threadIdx.x;
    if (bid < p.nb0) {
        const long i = (long)bid * 256 + tid;
        if (i * 4 < p.n_eps) {
            const U4 r = draw(p.state, 0, (uint64_t)i);
            const float u1 = u01(r.x), u2 = u01(r.y), u3 = u01(r.z), u4 = u01(r.w);
            const float r1 = sqrtf(-2.f * logf(u1)), r2 = sqrtf(-2.f * logf(u3));
            const float t1 = 6.283185307179586f * u2, t2 = 6.283185307179586f * u4;
            const float v[4] = {r1 * cosf(t1), r1 * sinf(t1), r2 * cosf(t2), r2 * sinf(t2)};
            for (int j = 0; j < 4; ++j)
                if (i * 4 + j < p.n_eps) p.eps[i * 4 + j] = v[j];
        }
    } else {
        const bool first = bid < p.nb0 + p.nb1;
        uint8_t* out = first ? p.m1 : p.m2;
        const long n = first ? p.n1 : p.n2;
        const long i = (long)(bid - p.nb0 - (first ? 0u : p.nb1)) * 256 + tid;
        if (i * 8 < n) {
            const U4 r = draw(p.state, first ? 1 : 2, (uint64_t)i);
            const uint32_t thr = (uint32_t)((first ? p.keep1 : p.keep2) * 65536.0f);
            const uint32_t w[4] = {r.x, r.y, r.z, r.w};
            if (i * 8 + 8 <= n && ((((uintptr_t)out) & 7) == 0)) {
                uint32_t lo = 0, hi = 0;
                for (int j = 0; j < 4; ++j) {
                    lo |= (((w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu) < thr ? 1u : 0u) << (8 * j);
                    hi |= (((w[2 + (j >> 1)] >> ((j & 1) * 16)) & 0xFFFFu) < thr ? 1u : 0u) << (8 * j);
                }
                *reinterpret_cast<uint2*>(out + i * 8) = make_uint2(lo, hi);
            } else {
                for (int j = 0; j < 8; ++j) {
                    const uint32_t h = (w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                    if (i * 8 + j < n) out[i * 8 + j] = (h < thr) ? 1 : 0;
                }
            }
        }
    }
}

}  // namespace

// state: device uint64[2] = {seed, offset}
extern "C" int lv_rng_normal_f32(float* out, long n, const uint64_t* state, uint64_t substream, void* stream) {
    if (!out || !state || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    LV_LAUNCH(rng_normal_kernel, dim3((unsigned)lv_cdiv((n + 3) / 4, 256)), dim3(256), 0, stream, out, n, state, substream);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_rng_keepmask_u8(uint8_t* out, long n, float keep_prob, const uint64_t* state, uint64_t substream,
                                  void* stream) {
    if (!out || !state || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    LV_LAUNCH(rng_keepmask_kernel, dim3((unsigned)lv_cdiv((n + 7) / 8, 256)), dim3(256), 0, stream, out, n, keep_prob, state,
              substream);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_rng_bernoulli_f32(const float* p, float* out, long n, const uint64_t* state, uint64_t substream,
                                    void* stream) {
    if (!p || !out || !state || n < 0) return LV_ERR_ARG;
    if (n == 0) return LV_OK;
    LV_LAUNCH(rng_bernoulli_kernel, dim3((unsigned)lv_cdiv((n + 3) / 4, 256)), dim3(256), 0, stream, p, out, n, state, substream);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_rng_advance(uint64_t* state, uint64_t inc, void* stream) {
    if (!state) return LV_ERR_ARG;
    LV_LAUNCH(rng_advance_kernel, dim3(1), dim3(64), 0, stream, state, inc);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// eps (normal, substream 0), the dropout_in keep-mask (substream 1) and the dropout_out keep-mask (substream 2) of one
// VAE.loss call in one launch, then offset += inc.  state: device uint64[2] = {seed, offset}; either mask may be NULL
// (eval mode).
extern "C" int lv_rng_noise_step(float* eps, long n_eps, uint8_t* mask_in, long n_in, float keep_in, uint8_t* mask_out,
                                 long n_out, float keep_out, uint64_t* state, uint64_t inc, void* stream) {
    if (!eps || !state || n_eps < 0 || n_in < 0 || n_out < 0) return LV_ERR_ARG;
    if (!mask_in) n_in = 0;
    if (!mask_out) n_out = 0;
    NoiseP p{eps, n_eps, mask_in, n_in, keep_in, mask_out, n_out, keep_out, state, inc,
             (unsigned)lv_cdiv((n_eps + 3) / 4, 256), (unsigned)lv_cdiv((n_in + 7) / 8, 256)};
    const unsigned nb = p.nb0 + p.nb1 + (unsigned)lv_cdiv((n_out + 7) / 8, 256);
    if (nb > 0) LV_LAUNCH(rng_noise_step_kernel, dim3(nb), dim3(256), 0, stream, p);
    if (inc != 0) LV_LAUNCH(rng_advance_kernel, dim3(1), dim3(64), 0, stream, state, inc);      // inc = 0: the caller advances (lv_loss_assemble_rng_f32)
    LV_CHECK_LAUNCH();
    return LV_OK;
}
