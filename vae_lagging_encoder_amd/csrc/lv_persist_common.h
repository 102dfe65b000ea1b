// lv_persist_common.h -- the XCD-group geometry and the tagged 8-byte hand-off granules of the persistent LSTM kernels
// (lv_lstm_persist16.hip).
#pragma once
#include "lv_device.h"

namespace lvp {

constexpr int PH = 1024;            // hidden size the persistent kernels are built for
constexpr int PGROUPS = 8;          // XCD-sized groups (blockIdx % 8)
constexpr int PMEMBERS = 32;        // workgroups per group
constexpr int SPIN_LIMIT = LV_SPIN_LIMIT;

typedef unsigned long long gran_t;  // a hand-off granule: payload + tag in one 64-bit word, written and read with agent-scope accesses

__device__ __forceinline__ gran_t gran_load(const gran_t* p) { return lv_agent_load_u64(p); }
__device__ __forceinline__ void gran_store(gran_t* p, gran_t v) { lv_agent_store_u64(p, v); }

// reduce-scatter granule: two partial sums as 28-bit floats (sign, exponent, 19 mantissa bits: 1e-6 relative, three orders below
// the bf16 rounding of the operands) beside an 8-bit phase tag (phase k >= 1 -> 1..255: a zeroed buffer never matches)
__device__ __forceinline__ uint32_t rs_tag(int k) { return 1u + (uint32_t)(k - 1) % 255u; }
__device__ __forceinline__ gran_t rs_pack(float a, float b, uint32_t tag) {
    uint32_t ua, ub;
    memcpy(&ua, &a, 4);
    memcpy(&ub, &b, 4);
    return (gran_t)((ua + 8u) >> 4) | ((gran_t)((ub + 8u) >> 4) << 28) | ((gran_t)tag << 56);
}
__device__ __forceinline__ float rs_lo(gran_t g) {
    const uint32_t u = ((uint32_t)g & 0x0FFFFFFFu) << 4;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__device__ __forceinline__ float rs_hi(gran_t g) {
    const uint32_t u = ((uint32_t)(g >> 28) & 0x0FFFFFFFu) << 4;
    float f;
    memcpy(&f, &u, 4);
    return f;
}


// reduce-scatter granule of the 4-row BPTT: FOUR partial sums in 16 bytes, each dword a 30-bit float (sign, exponent, 21 mantissa
// bits) under a 2-bit phase tag of its own -- a granule is valid when all four dwords carry the wanted tag, so a torn 16-byte access
// is detected like a late one.  Phases cycle 1, 2, 3: a slot of the parity-double-buffered exchange last held phase k - 2, whose tag
// differs from phase k's, and a zeroed buffer matches no phase.
__device__ __forceinline__ uint32_t rs4_tag(int k) { return 1u + (uint32_t)(k - 1) % 3u; }
__device__ __forceinline__ uint32_t rs4_pack(float a, uint32_t tag) {
    uint32_t u;
    memcpy(&u, &a, 4);
    return ((u + 2u) >> 2) | (tag << 30);
}
__device__ __forceinline__ float rs4_val(uint32_t g) {
    const uint32_t u = g << 2;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// all dwords of N granules carry tag `want`
template <int N> __device__ __forceinline__ bool rs4_all_tagged(const uint4 (&v)[N], uint32_t want) {
    const uint32_t w = want << 30;
    uint32_t x = 0u;
#pragma unroll
    for (int j = 0; j < N; ++j) x |= (v[j].x ^ w) | (v[j].y ^ w) | (v[j].z ^ w) | (v[j].w ^ w);
    return (x >> 30) == 0u;
}

}  // namespace lvp
