// lv_device.h -- device-side vocabulary shared by every kernel file (gfx950 / CDNA4, wave64).
//
// The product build is hipcc --offload-arch=gfx950.  The single LV_EMU switch below exists only so
// that the GPU-less CI can compile the same kernel sources with g++ against tests/emu/hip_emu.h
// (a thread-level emulator used by the `-m "not gpu"` tests): no kernel file switches on the PLATFORM.  What the kernel files do
// contain are `#ifndef LV_<KNOB>` defaults of measurement knobs (tile schedules, block lengths, the what-if builds of
// profiles/microbench/): the product build defines none of them, the probes build variant libraries with -D.
#pragma once

#ifdef LV_EMU
#include "hip_emu.h"
#define LV_LAUNCH(kern, grid, block, shmem, stream, ...) \
    lv_emu::launch((grid), (block), (shmem), [=]() { kern(__VA_ARGS__); })
#define LV_DYN_SHARED(name) char* name = lv_emu::dyn_smem()
static inline f32x4 lv_mfma_16x16x4(float a, float b, f32x4 c) { return lv_emu_mfma_16x16x4(a, b, c); }
static inline f32x16 lv_mfma_32x32x2(float a, float b, f32x16 c) { return lv_emu_mfma_32x32x2(a, b, c); }
#define LV_SCHED_BARRIER() do { } while (0)
static inline int lv_wave_uniform(int v) { return v; }
// persistent (spin-synchronised) kernels: every workgroup of the grid live at once, block-private LDS objects, polls yield
#define LV_LAUNCH_RESIDENT(kern, grid, block, shmem, stream, ...) \
    lv_emu::launch_concurrent((grid), (block), (shmem), [=]() { kern(__VA_ARGS__); })
#define LV_BLOCK_SHARED(T, name) T& name = *reinterpret_cast<T*>(lv_emu::block_shared(sizeof(T)))
static inline unsigned long long lv_agent_load_u64(const unsigned long long* p) {
    lv_emu::yield_all();                     // somebody else has to run for the value to change
    return __atomic_load_n(p, __ATOMIC_RELAXED);
}
static inline void lv_agent_store_u64(unsigned long long* p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
static inline void lv_xcd_store_u64(unsigned long long* p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
static inline void lv_poll_backoff() { }
#define LV_WAIT_LDS() lv_emu::wave_sync()      // lanes are fibers here: 'the wave's own LDS writes are visible' needs a rendezvous
template <class T> static inline void lv_store_nt(T v, T* p) { *p = v; }
static inline int lv_device_cus() { return 1 << 20; }
#define LV_SPIN_LIMIT (1 << 15)             // polls per wait before a hand-off is reported lost (a poll = one scheduling round here)
static inline f32x16 lv_mfma_32x32x16_bf16(uint4 a, uint4 b, f32x16 c) { return lv_emu_mfma_32x32x16_bf16(a, b, c); }
static inline f32x4 lv_mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c) { return lv_emu_mfma_16x16x32_bf16(a, b, c); }
static inline f32x4 lv_mfma_16x16x32_bf16_areg(uint4 a, uint4 b, f32x4 c) { return lv_emu_mfma_16x16x32_bf16(a, b, c); }
static inline f32x4 lv_mfma_16x16x32_bf16_areg_first(uint4 a, uint4 b) { return lv_emu_mfma_16x16x32_bf16(a, b, f32x4{0.f, 0.f, 0.f, 0.f}); }
#define LV_MFMA_DRAIN() do { } while (0)
static inline void LV_MFMA_RESULT(f32x4&) { }
static inline f32x4 lv_mfma_4x4x4_16b_bf16(uint2 a, uint2 b, f32x4 c) { return lv_emu_mfma_4x4x4_16b_bf16(a, b, c); }
// LDS-DMA: lane l's 16 bytes at g land at lds_wave_base + 16*l (the base is wave-uniform)
static inline void lv_glds16(const void* g, void* lds_wave_base) { memcpy((char*)lds_wave_base + 16 * lv_emu::lane(), g, 16); }
static inline void lv_glds16_uncounted(const void* g, void* lds_wave_base) { lv_glds16(g, lds_wave_base); }
#define LV_WAIT_VMEM() do { } while (0)
#define LV_WAIT_VMEM_N(n) do { } while (0)
// in-launch hand-off of a slab to whichever workgroup arrives last (grouped stream-K GEMM): write-through payload store, agent acquire
static inline void lv_store_wt_f4(float4* p, float4 v) { *p = v; }
static inline void lv_acquire_agent() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
static inline unsigned lv_agent_load_u32(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
static inline void lv_sleep_short() { }
#define LV_ARRIVAL_POLLS 2                  // bounded look for other workgroups' arrivals: they run one after the other here -- what has not arrived will not
static inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
#define LV_S_BARRIER() __syncthreads()      // a bare workgroup barrier (no counter waits attached); fibers: the same rendezvous
#define LV_SETPRIO(n) do { } while (0)
static inline uint2 lv_ds_read_tr16_b64(const void* lds_ptr) {            // lane map: see the HIP definition below
    auto& w = lv_emu::my_wave();
    const int l = lv_emu::lane();
    uint64_t blk;
    memcpy(&blk, lds_ptr, 8);
    w.u[l] = blk;
    lv_emu::wave_sync();
    uint16_t o[4];
    for (int j = 0; j < 4; ++j) {
        const uint64_t b = w.u[(l & ~15) + 4 * j + ((l & 15) >> 2)];
        o[j] = (uint16_t)(b >> (16 * (l & 3)));
    }
    lv_emu::wave_sync();
    uint2 r;
    memcpy(&r, o, 8);
    return r;
}
// DPP row_shr with a bank mask: lanes of bank BANK (lanes 4 BANK .. 4 BANK + 3 of every 16-lane row) take src from the lane SHIFT
// positions lower in their row, every other lane keeps old
template <int SHIFT, int BANK> static inline float lv_row_shr_into(float old, float src) {
    auto& w = lv_emu::my_wave();
    const int l = lv_emu::lane();
    uint32_t b;
    memcpy(&b, &src, 4);
    w.u[l] = b;
    lv_emu::wave_sync();
    float out = old;
    if (((l & 15) >> 2) == BANK && (l & 15) >= SHIFT) { const uint32_t v = (uint32_t)w.u[l - SHIFT]; memcpy(&out, &v, 4); }
    lv_emu::wave_sync();
    return out;
}
// saturation in front of a binary16 conversion: finite values beyond +-65504 clamp (never inf), NaN stays NaN (fmaxf / fminf
// would turn it into -65504: a poisoned weight must poison the forward as it poisons the bf16 images of the backward)
static inline float lv_sat_f16(float x) { return x != x ? x : fminf(fmaxf(x, -65504.f), 65504.f); }
// IEEE binary16 <-> f32 in software (round-to-nearest-even, as v_cvt_f16_f32 does): g++ 11 has no _Float16
static inline uint16_t lv_f32_to_f16_bits(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7FFFFFFFu;
    if (u >= 0x7F800000u) return (uint16_t)(sign | (u > 0x7F800000u ? 0x7E00u : 0x7C00u));       // nan / inf
    if (u >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                       // rounds to inf
    if (u < 0x33000001u) return (uint16_t)sign;                                                    // rounds to zero
    if (u < 0x38800000u) {                                                                         // subnormal half
        const int shift = 126 - (int)(u >> 23);                                                    // 14 .. 24 bits to drop
        const uint32_t mant = (u & 0x7FFFFFu) | 0x800000u;
        uint32_t h = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((u - 0x38000000u) >> 13);
    const uint32_t rem = u & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (uint16_t)(sign | h);
}
static inline float lv_f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 0x400u)) { mm <<= 1; ++sh; }
            u = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3FFu) << 13);
        }
    } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// the value held by lane (l ^ 1)
static inline float lv_lane_xor1(float v) { return __shfl_xor(v, 1, 64); }
static inline uint32_t lv_lane_xor1_u32(uint32_t v) { return (uint32_t)__shfl_xor((int)v, 1, 64); }
// the value held by lane I of this lane's quad (lanes 4q .. 4q + 3)
template <int I> static inline uint32_t lv_quad_bcast_u32(uint32_t v) { return (uint32_t)lv_emu_xchg((int)v, (lv_emu::lane() & ~3) + I); }
// x + (the value of lane l ^ 16) / (l ^ 32): one step of a butterfly sum across a wave's 16-lane rows / its two halves
static inline float lv_add_xor16(float x) { return x + __shfl_xor(x, 16, 64); }
static inline float lv_add_xor32(float x) { return x + __shfl_xor(x, 32, 64); }
// x + (the value of lane l ^ 8)
static inline float lv_add_xor8(float x) { return x + __shfl_xor(x, 8, 64); }
// TRANSPOSING butterfly steps: two registers in, one out.  lv_fold8: a lane with bit 3 clear gets x(l) + x(l + 8), one with bit 3 set
// y(l - 8) + y(l); lv_fold16: a lane of an even 16-lane row gets x(l) + x(l + 16), a lane of an
// odd row y(l - 16) + y(l); lv_fold32: a lane of the lower half gets x(l) + x(l + 32), a lane of the upper half y(l - 32) + y(l) --
// each step halves the number of live registers instead of leaving every lane with every sum.
static inline float lv_fold8(float x, float y) {
    const float xs = __shfl_xor(x, 8, 64), ys = __shfl_xor(y, 8, 64);
    return (lv_emu::lane() & 8) ? ys + y : x + xs;
}
static inline float lv_fold16(float x, float y) {
    const float xs = __shfl_xor(x, 16, 64), ys = __shfl_xor(y, 16, 64);
    return (lv_emu::lane() & 16) ? ys + y : x + xs;
}
static inline float lv_fold32(float x, float y) {
    const float xs = __shfl_xor(x, 32, 64), ys = __shfl_xor(y, 32, 64);
    return (lv_emu::lane() & 32) ? ys + y : x + xs;
}
// 16-byte hand-off granules: four dwords that each carry their own tag bits -- 16-byte atomicity is assumed NOWHERE, and here the
// stores are torn on purpose (another fiber runs between the two halves).  Four granules are polled by one call (one wait for all).
static inline void lv_agent_load_q4x4(const void* p0, const void* p1, const void* p2, const void* p3, uint4 (&v)[4]) {
    lv_emu::yield_all();
    const void* ps[4] = {p0, p1, p2, p3};
    for (int i = 0; i < 4; ++i) {
        const unsigned* q = static_cast<const unsigned*>(ps[i]);
        v[i] = make_uint4(__atomic_load_n(q, __ATOMIC_RELAXED), __atomic_load_n(q + 1, __ATOMIC_RELAXED),
                          __atomic_load_n(q + 2, __ATOMIC_RELAXED), __atomic_load_n(q + 3, __ATOMIC_RELAXED));
    }
}
static inline void lv_agent_load_q4x8(const void* const (&ps)[8], uint4 (&v)[8]) {
    lv_emu::yield_all();
    for (int i = 0; i < 8; ++i) {
        const unsigned* q = static_cast<const unsigned*>(ps[i]);
        v[i] = make_uint4(__atomic_load_n(q, __ATOMIC_RELAXED), __atomic_load_n(q + 1, __ATOMIC_RELAXED),
                          __atomic_load_n(q + 2, __ATOMIC_RELAXED), __atomic_load_n(q + 3, __ATOMIC_RELAXED));
    }
}
// eight granules 4 KB apart (one per row of the forward's h exchange)
static inline void lv_agent_load_q4x8_rows(const char* p, uint4 (&v)[8]) {
    const void* ps[8];
    for (int i = 0; i < 8; ++i) ps[i] = p + i * 4096;
    lv_agent_load_q4x8(ps, v);
}
// eight granules at p + (i & 3) S + (i >> 2) 128 (two polling rounds of the BPTT's reduce-scatter: four senders S bytes apart, rounds 128 apart)
template <int S> static inline void lv_agent_load_q4x8_rs(const char* p, uint4 (&v)[8]) {
    const void* ps[8];
    for (int i = 0; i < 8; ++i) ps[i] = p + (i & 3) * S + (i >> 2) * 128;
    lv_agent_load_q4x8(ps, v);
}
static inline void lv_agent_store_q4(void* p, uint4 v) {
    unsigned* q = static_cast<unsigned*>(p);
    __atomic_store_n(q + 2, v.z, __ATOMIC_RELAXED); __atomic_store_n(q + 3, v.w, __ATOMIC_RELAXED);
    lv_emu::yield_all();
    __atomic_store_n(q, v.x, __ATOMIC_RELAXED); __atomic_store_n(q + 1, v.y, __ATOMIC_RELAXED);
}
static inline void lv_xcd_store_q4(void* p, uint4 v) { lv_agent_store_q4(p, v); }
// two f32 -> packed binary16 (RNE), lo in bits 0..15
static inline uint32_t lv_pack_f16x2(float lo, float hi) { return (uint32_t)lv_f32_to_f16_bits(lo) | ((uint32_t)lv_f32_to_f16_bits(hi) << 16); }
// binary16 forms of the two 16-bit-operand MFMAs (v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16): the lane maps of the bf16
// forms, elements decoded as IEEE half
static inline f32x16 lv_mfma_32x32x16_f16(uint4 a, uint4 b, f32x16 c) {
    auto& w = lv_emu::my_wave();
    const int l = lv_emu::lane();
    w.qa[l] = a; w.qb[l] = b;
    lv_emu::wave_sync();
    f32x16 d = c;
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int h = 0; h < 2; ++h) {
            unsigned short ea[8], eb[8];
            memcpy(ea, &w.qa[row + 32 * h], 16);
            memcpy(eb, &w.qb[col + 32 * h], 16);
            for (int e = 0; e < 8; ++e) acc = fmaf(lv_f16_bits_to_f32(ea[e]), lv_f16_bits_to_f32(eb[e]), acc);
        }
        d[r] = acc;
    }
    lv_emu::wave_sync();
    return d;
}
static inline f32x4 lv_mfma_16x16x32_f16_areg(uint4 a, uint4 b, f32x4 c) {
    auto& w = lv_emu::my_wave();
    const int l = lv_emu::lane();
    w.qa[l] = a; w.qb[l] = b;
    lv_emu::wave_sync();
    f32x4 d = c;
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            unsigned short ea[8], eb[8];
            memcpy(ea, &w.qa[row + 16 * g], 16);
            memcpy(eb, &w.qb[col + 16 * g], 16);
            for (int e = 0; e < 8; ++e) acc = fmaf(lv_f16_bits_to_f32(ea[e]), lv_f16_bits_to_f32(eb[e]), acc);
        }
        d[r] = acc;
    }
    lv_emu::wave_sync();
    return d;
}
static inline f32x4 lv_mfma_16x16x32_f16_areg_first(uint4 a, uint4 b) { return lv_mfma_16x16x32_f16_areg(a, b, f32x4{0.f, 0.f, 0.f, 0.f}); }
#else
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LV_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), (shmem), (hipStream_t)(stream), __VA_ARGS__)
#define LV_DYN_SHARED(name) extern __shared__ __attribute__((aligned(16))) char name[]
// pin instruction order across this point (keeps a block of independent loads issued ahead of their consumers)
#define LV_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// a value every lane of the wave agrees on, moved to a scalar register (branches on it are scalar branches, not exec masks)
__device__ __forceinline__ int lv_wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// persistent (spin-synchronised) kernels: an ordinary launch whose grid the caller sized to be resident at once
#define LV_LAUNCH_RESIDENT(kern, grid, block, shmem, stream, ...) LV_LAUNCH(kern, grid, block, shmem, stream, __VA_ARGS__)
#define LV_BLOCK_SHARED(T, name) __shared__ T name
// agent-scope relaxed 64-bit accesses (sc1: served by L2, never by another CU's stale L1) for tagged hand-off granules
__device__ __forceinline__ unsigned long long lv_agent_load_u64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lv_agent_store_u64(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same 64-bit store WITHOUT the agent-scope write-through (no sc1): it reaches the XCD's L2 through the write-through vector
// L1 and stays there, visible to sc1 loads of the CUs of THAT XCD only.  For hand-offs inside an XCD-local group: the granules
// then never travel to the fabric (the agent-scope form writes every granule through to memory).  A reader on another XCD would
// never see it -- callers must be able to detect that (bounded spins) and fall back.
__device__ __forceinline__ void lv_xcd_store_u64(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#define LV_SPIN_LIMIT (1 << 22)             // polls per wait before a hand-off is reported lost (~1 s)
// lgkmcnt(0): a wave's own LDS accesses are ordered; enough when the data is wave-private
#define LV_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xc07f)
// streaming store: consumed by later kernels only (no write-allocate fetch of a partially written line)
template <class T> __device__ __forceinline__ void lv_store_nt(T v, T* p) { __builtin_nontemporal_store(v, p); }
// compute units of the current device (cached per ordinal; the persistent launches need their whole grid resident)
static inline int lv_device_cus() {
    static int cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int c = __atomic_load_n(&cache[dev], __ATOMIC_RELAXED);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        __atomic_store_n(&cache[dev], c, __ATOMIC_RELAXED);
    }
    return c;
}
// v_mfma_f32_16x16x4_f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)*4+r][col=l&15]
__device__ __forceinline__ f32x4 lv_mfma_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x2_f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//                          D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
__device__ __forceinline__ f32x16 lv_mfma_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_32x32x16_bf16: a/b = 8 bf16 (raw bits in a uint4) of A row / B column (l&31), k = 8*(l>>5)+e
typedef __bf16 lv_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 lv_mfma_32x32x16_bf16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<lv_bf16x8*>(&a), *reinterpret_cast<lv_bf16x8*>(&b),
                                                   c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16: a/b = 8 bf16 of A row / B column (l&15), k = 8*(l>>4)+e; D[row=(l>>4)*4+r][col=l&15]
__device__ __forceinline__ f32x4 lv_mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<lv_bf16x8*>(&a), *reinterpret_cast<lv_bf16x8*>(&b),
                                                   c, 0, 0, 0);
}
// The same instruction with its A operand READ FROM THE ACCUMULATION REGISTERS (gfx90a+: srcA / srcB of an MFMA may be AGPRs).  For
// operands that stay resident for a whole kernel and fill half of the 512 registers (W_hh of the persistent LSTM kernels): the
// register allocator otherwise keeps such values in AGPRs as SPILL slots and copies them back (4 x v_accvgpr_read per MFMA, as
// many VALU cycles as the MFMA's own pipe time).  Inline assembly, so the compiler's MFMA hazard handling does not see it: the
// caller puts LV_MFMA_DRAIN() between the last of these and the first instruction that reads an accumulator (an 8-pass XDL
// write needs 11 wait states before a VALU / LDS / VMEM read of its destination).
typedef unsigned int lv_u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 lv_mfma_16x16x32_bf16_areg(uint4 a, uint4 b, f32x4 c) {
    const lv_u32x4v av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(av), "v"(bv));
    return c;
}
// first product of an accumulation chain: C = 0 as an inline constant (no zeroing v_mov in front of an instruction the compiler
// does not know to be an MFMA)
__device__ __forceinline__ f32x4 lv_mfma_16x16x32_bf16_areg_first(uint4 a, uint4 b) {
    const lv_u32x4v av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    f32x4 c;
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(c) : "a"(av), "v"(bv));
    return c;
}
// the binary16 forms (same shapes, lane maps and rates; 11 bits of significand instead of 8 -- for operands whose range permits:
// the ENCODER's forward, whose weight rounding is what moves the KL, profiles/r05a_kl_ablation.txt)
typedef _Float16 lv_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 lv_mfma_32x32x16_f16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<lv_f16x8*>(&a), *reinterpret_cast<lv_f16x8*>(&b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 lv_mfma_16x16x32_f16_areg(uint4 a, uint4 b, f32x4 c) {
    const lv_u32x4v av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(av), "v"(bv));
    return c;
}
__device__ __forceinline__ f32x4 lv_mfma_16x16x32_f16_areg_first(uint4 a, uint4 b) {
    const lv_u32x4v av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
    f32x4 c;
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "a"(av), "v"(bv));
    return c;
}
#define LV_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 3" ::: "memory")
// ... and passes every accumulator it is about to read through LV_MFMA_RESULT() BEHIND that point in program order: volatile asm
// statements keep their order, ordinary instructions that merely depend on an asm's output do not -- without this the scheduler is
// free to hoist a read of acc right behind the MFMA that writes it (no interlock: the read returns a half-written accumulator;
// the first build of the BPTT did exactly that, 2 % rms error after 200 timesteps).
__device__ __forceinline__ void LV_MFMA_RESULT(f32x4& c) { asm volatile("" : "+v"(c)); }
// v_mfma_f32_4x4x4_16b_bf16: 16 independent 4 x 4 x 4 products.  Lane l belongs to block l >> 2; a = 4 bf16 (k = 0..3) of A row
// (l & 3), b = 4 bf16 of B column (l & 3); D[row r][col l & 3] of the block in register r.  2 passes: for a 4-row A it does the
// useful work of a 16x16x32 MFMA (whose other 12 rows would be padding) in half the matrix-pipe time.
typedef short lv_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 lv_mfma_4x4x4_16b_bf16(uint2 a, uint2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(*reinterpret_cast<lv_s16x4*>(&a), *reinterpret_cast<lv_s16x4*>(&b), c, 0, 0, 0);
}
// global -> LDS without passing through registers (global_load_lds_dwordx4): lane l's 16 bytes at g land at
// lds_wave_base + 16*l; lds_wave_base must be wave-uniform.  The data is ordered for LDS readers by the issuing wave's
// vmcnt(0) (LV_WAIT_VMEM) followed by a workgroup barrier.
__device__ __forceinline__ void lv_glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same transfer as an asm statement, i.e. OUTSIDE hipcc's bookkeeping: the compiler neither counts it for its s_waitcnt
// insertion nor sees an LDS write it would have to order its own ds_reads of the same array behind (it puts s_waitcnt vmcnt(0) in
// front of every read of an LDS object that has a counted LDS-DMA in flight -- which is exactly what a schedule that keeps the DMA
// stream running while it reads OTHER parts of the same buffer must not have).  The caller owns the ordering: a counted
// s_waitcnt vmcnt(N) of its own, then a barrier, then the reads.  M0 is compiler-reserved: saved and restored inside the statement
// (cdna_hip_programming.md 5.7).
#ifndef LV_GLDS_POLICY
#define LV_GLDS_POLICY ""          // cache-policy bits of the uncounted LDS-DMA (" nt", " sc0", " sc1", ...: A/B knob of profiles/microbench)
#endif
__device__ __forceinline__ void lv_glds16_uncounted(const void* g, void* lds_wave_base) {
    unsigned keep;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" LV_GLDS_POLICY "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
#define LV_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// In-launch hand-off of a slab between workgroups (cdna_hip_programming.md Guideline 16, form R1): the payload leaves as 16-byte
// WRITE-THROUGH stores (sc1: the bytes reach memory, no release fence needed; the s_nop keeps hipcc from reusing the data registers
// before the store has read them), every storing wave drains them (LV_WAIT_VMEM: the statement is outside hipcc's counters), then a
// workgroup barrier and ONE lane's agent-scope atomic; the consumer's one lane issues lv_acquire_agent() (drops this CU's stale L1
// lines) behind its atomic, a workgroup barrier, then plain loads.
typedef float lv_f32x4_wt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lv_store_wt_f4(float4* p, float4 v) {
    const lv_f32x4_wt x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void lv_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ unsigned lv_agent_load_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lv_sleep_short() { __builtin_amdgcn_s_sleep(8); }
#define LV_ARRIVAL_POLLS 64                 // bounded look for other workgroups' arrivals (~0.5-1 us per poll): never a condition of progress
// counted form: all but the newest n vector-memory operations of this wave have completed (n a literal)
#define LV_WAIT_VMEM_N(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// s_barrier alone: __syncthreads() puts s_waitcnt vmcnt(0) lgkmcnt(0) in front of it whenever anything is in flight -- for
// schedules that keep LDS-DMA in flight across a barrier and order it by counted waits of their own
#define LV_S_BARRIER() __builtin_amdgcn_s_barrier()
// issue priority of this wave against the other waves of its SIMD (0..3): the wave that multiplies goes first
#define LV_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
// ds_read_b64_tr_b16 (LDS transpose read), lane map measured with profiles/microbench/ds_read_tr_probe.hip: inside each group
// of 16 lanes every lane r supplies the address of FOUR consecutive 16-bit elements, a 16 x 4 matrix Mx[r][c]; lane i receives
// out[j] = Mx[4j + (i >> 2)][i & 3].  Pointing lane r at T[k0 + (r >> 2)][m0 + 4 (r & 3)] of a row-major [k][m] tile therefore
// gives lane i the four elements T[k0 + j][m0 + i]: a K-contiguous MFMA fragment out of an M-contiguous image.
typedef short lv_s16x4_lds __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lv_ds_read_tr16_b64(const void* lds_ptr) {
    lv_s16x4_lds v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lv_s16x4_lds*)lds_ptr);
    return *reinterpret_cast<uint2*>(&v);
}
// saturation in front of a binary16 conversion: finite values beyond +-65504 clamp (never inf), NaN stays NaN (fmaxf / fminf
// would turn it into -65504: a poisoned weight must poison the forward as it poisons the bf16 images of the backward)
__device__ __forceinline__ float lv_sat_f16(float x) { return x != x ? x : fminf(fmaxf(x, -65504.f), 65504.f); }
// IEEE binary16 <-> f32 (v_cvt_f16_f32 / v_cvt_f32_f16, round-to-nearest-even)
__device__ __forceinline__ uint16_t lv_f32_to_f16_bits(float x) {
    const _Float16 h = (_Float16)x;
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
__device__ __forceinline__ float lv_f16_bits_to_f32(uint16_t b) {
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}
// the value held by lane (l ^ 1): DPP quad_perm [1,0,3,2], no LDS crossbar
__device__ __forceinline__ float lv_lane_xor1(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ uint32_t lv_lane_xor1_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
// the value held by lane I of this lane's quad (DPP quad_perm:[I, I, I, I])
template <int I> __device__ __forceinline__ uint32_t lv_quad_bcast_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, I * 0x55, 0xF, 0xF, true);
}
// x + (the value of lane l ^ 16) / (l ^ 32): one step of a butterfly sum across a wave's 16-lane rows / its two halves, on gfx950's
// lane-swap instructions (v_permlane16_swap / v_permlane32_swap: VALU, a few cycles) instead of __shfl_xor's ds_bpermute (an LDS
// crossbar round trip, ~120 cycles on a timestep's critical path).  With both operands the same register the swap leaves the even
// rows' (lower half's) values in one result and the odd rows' (upper half's) in the other, in EVERY lane: their sum is self + partner
// in the partner-less order -- the same bits as the shuffle form, since the one add is commutative.
__device__ __forceinline__ float lv_add_xor16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float lv_add_xor32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// x + (the value of lane l ^ 8): DPP row_ror:8 (inside a 16-lane row a rotation by 8 IS the xor), folded into the add by the compiler
__device__ __forceinline__ float lv_add_xor8(float x) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true);
    return x + __builtin_bit_cast(float, r);
}
// TRANSPOSING butterfly steps on the lane-swap instructions: two registers in, one out.  v_permlane16_swap exchanges the ODD rows of
// its first operand with the EVEN rows of its second: afterwards the first holds (x.row0, y.row0, x.row2, y.row2) and the second
// (x.row1, y.row1, x.row3, y.row3), so their sum is x(l) + x(l + 16) in the even rows and y(l - 16) + y(l) in the odd rows
// (lv_fold16); v_permlane32_swap does the same with the two halves of the wave (lv_fold32).  One swap + one add reduces TWO
// registers by one butterfly level and leaves every lane with a sum nobody else holds.
// the same step between the two halves of a 16-lane row: DPP row_ror:8 of both registers, the lane keeps the sum it is to own
__device__ __forceinline__ float lv_fold8(float x, float y) {
    const float xs = lv_add_xor8(x), ys = lv_add_xor8(y);
    return (threadIdx.x & 8) ? ys : xs;
}
__device__ __forceinline__ float lv_fold16(float x, float y) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float lv_fold32(float x, float y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// 16-byte hand-off granules (four dwords that each carry their own tag bits: 16-byte atomicity is assumed nowhere).  The loads are
// L2-served (sc1, like lv_agent_load_u64); four granules are polled by ONE statement that also holds the wait -- the compiler does not
// know when an inline-assembly load lands and would otherwise reuse its destination registers.  Stores: the agent-scope form writes
// through (sc1), the XCD form stays in the XCD's L2 (see lv_xcd_store_u64).  s_nop: a store of more than 8 bytes must not be followed
// directly by a write of its data registers (the compiler's hazard recogniser does not see into the statement).
__device__ __forceinline__ void lv_agent_load_q4x4(const void* p0, const void* p1, const void* p2, const void* p3, uint4 (&v)[4]) {
    lv_u32x4v r0, r1, r2, r3;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
    v[0] = make_uint4(r0.x, r0.y, r0.z, r0.w); v[1] = make_uint4(r1.x, r1.y, r1.z, r1.w);
    v[2] = make_uint4(r2.x, r2.y, r2.z, r2.w); v[3] = make_uint4(r3.x, r3.y, r3.z, r3.w);
}
__device__ __forceinline__ void lv_agent_load_q4x8(const void* const (&ps)[8], uint4 (&v)[8]) {
    lv_u32x4v r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                 : "v"(ps[0]), "v"(ps[1]), "v"(ps[2]), "v"(ps[3]), "v"(ps[4]), "v"(ps[5]), "v"(ps[6]), "v"(ps[7]) : "memory");
    v[0] = make_uint4(r0.x, r0.y, r0.z, r0.w); v[1] = make_uint4(r1.x, r1.y, r1.z, r1.w);
    v[2] = make_uint4(r2.x, r2.y, r2.z, r2.w); v[3] = make_uint4(r3.x, r3.y, r3.z, r3.w);
    v[4] = make_uint4(r4.x, r4.y, r4.z, r4.w); v[5] = make_uint4(r5.x, r5.y, r5.z, r5.w);
    v[6] = make_uint4(r6.x, r6.y, r6.z, r6.w); v[7] = make_uint4(r7.x, r7.y, r7.z, r7.w);
}
// eight granules 4 KB apart (one per row of the forward's h exchange): four address register pairs, each reaching its row and the
// one below it through the signed 13-bit immediate
__device__ __forceinline__ void lv_agent_load_q4x8_rows(const char* p, uint4 (&v)[8]) {
    const char *p1 = p + 4096, *p3 = p + 3 * 4096, *p5 = p + 5 * 4096, *p7 = p + 7 * 4096;
    lv_u32x4v r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("global_load_dwordx4 %0, %8, off offset:-4096 sc1\n\tglobal_load_dwordx4 %1, %8, off sc1\n\t"
                 "global_load_dwordx4 %2, %9, off offset:-4096 sc1\n\tglobal_load_dwordx4 %3, %9, off sc1\n\t"
                 "global_load_dwordx4 %4, %10, off offset:-4096 sc1\n\tglobal_load_dwordx4 %5, %10, off sc1\n\t"
                 "global_load_dwordx4 %6, %11, off offset:-4096 sc1\n\tglobal_load_dwordx4 %7, %11, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                 : "v"(p1), "v"(p3), "v"(p5), "v"(p7) : "memory");
    v[0] = make_uint4(r0.x, r0.y, r0.z, r0.w); v[1] = make_uint4(r1.x, r1.y, r1.z, r1.w);
    v[2] = make_uint4(r2.x, r2.y, r2.z, r2.w); v[3] = make_uint4(r3.x, r3.y, r3.z, r3.w);
    v[4] = make_uint4(r4.x, r4.y, r4.z, r4.w); v[5] = make_uint4(r5.x, r5.y, r5.z, r5.w);
    v[6] = make_uint4(r6.x, r6.y, r6.z, r6.w); v[7] = make_uint4(r7.x, r7.y, r7.z, r7.w);
}
// eight granules at p + (i & 3) S + (i >> 2) 128 (two polling rounds of the BPTT's reduce-scatter: four senders S bytes apart, rounds
// 128 apart): ONE or TWO address registers pairs and immediate offsets (12 bits) instead of eight pointers
template <int S> __device__ __forceinline__ void lv_agent_load_q4x8_rs(const char* p, uint4 (&v)[8]) {
    static_assert(3 * S + 128 < 8192, "two address registers reach 8 KB");
    const char* const ph = p + 4096;
#define LV_RS_O(i) (((i) & 3) * S + ((i) >> 2) * 128)
#define LV_RS_P(i) (LV_RS_O(i) < 4096 ? p : ph)
    lv_u32x4v r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("global_load_dwordx4 %0, %8, off offset:%16 sc1\n\tglobal_load_dwordx4 %1, %9, off offset:%17 sc1\n\t"
                 "global_load_dwordx4 %2, %10, off offset:%18 sc1\n\tglobal_load_dwordx4 %3, %11, off offset:%19 sc1\n\t"
                 "global_load_dwordx4 %4, %12, off offset:%20 sc1\n\tglobal_load_dwordx4 %5, %13, off offset:%21 sc1\n\t"
                 "global_load_dwordx4 %6, %14, off offset:%22 sc1\n\tglobal_load_dwordx4 %7, %15, off offset:%23 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                 : "v"(LV_RS_P(0)), "v"(LV_RS_P(1)), "v"(LV_RS_P(2)), "v"(LV_RS_P(3)), "v"(LV_RS_P(4)), "v"(LV_RS_P(5)), "v"(LV_RS_P(6)), "v"(LV_RS_P(7)),
                   "n"(LV_RS_O(0) & 4095), "n"(LV_RS_O(1) & 4095), "n"(LV_RS_O(2) & 4095), "n"(LV_RS_O(3) & 4095),
                   "n"(LV_RS_O(4) & 4095), "n"(LV_RS_O(5) & 4095), "n"(LV_RS_O(6) & 4095), "n"(LV_RS_O(7) & 4095)
                 : "memory");
#undef LV_RS_O
#undef LV_RS_P
    v[0] = make_uint4(r0.x, r0.y, r0.z, r0.w); v[1] = make_uint4(r1.x, r1.y, r1.z, r1.w);
    v[2] = make_uint4(r2.x, r2.y, r2.z, r2.w); v[3] = make_uint4(r3.x, r3.y, r3.z, r3.w);
    v[4] = make_uint4(r4.x, r4.y, r4.z, r4.w); v[5] = make_uint4(r5.x, r5.y, r5.z, r5.w);
    v[6] = make_uint4(r6.x, r6.y, r6.z, r6.w); v[7] = make_uint4(r7.x, r7.y, r7.z, r7.w);
}
__device__ __forceinline__ void lv_agent_store_q4(void* p, uint4 v) {
    const lv_u32x4v d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
}
#ifndef LV_POLL_SLEEP
#define LV_POLL_SLEEP 0                     // measurement knob: s_sleep units (64 cycles) after a FAILED poll of a hand-off, before the next one
#endif
__device__ __forceinline__ void lv_poll_backoff() { if (LV_POLL_SLEEP > 0) __builtin_amdgcn_s_sleep(LV_POLL_SLEEP); }
#ifndef LV_XCD_ST_MODS
#define LV_XCD_ST_MODS ""                   // measurement knob: cache-policy bits of the XCD-local granule store (" nt", " sc0", ...)
#endif
__device__ __forceinline__ void lv_xcd_store_q4(void* p, uint4 v) {
    const lv_u32x4v d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off" LV_XCD_ST_MODS "\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
}
// DPP row_shr with a bank mask (v_mov_b32_dpp row_shr:SHIFT bank_mask:1 << BANK): lanes of bank BANK (lanes 4 BANK .. 4 BANK + 3 of
// every 16-lane row) take src from the lane SHIFT positions lower in their row, every other lane keeps old.  Merges the valid
// quarter-rows of four MFMA result registers into one fully populated register (one store instruction instead of four).
template <int SHIFT, int BANK> __device__ __forceinline__ float lv_row_shr_into(float old, float src) {
    const int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x110 + SHIFT, 0xF, 1 << BANK, false);
    return __builtin_bit_cast(float, r);
}
typedef _Float16 lv_h2 __attribute__((ext_vector_type(2)));
typedef float lv_f2 __attribute__((ext_vector_type(2)));
// two f32 -> packed binary16 (RNE), lo in bits 0..15
__device__ __forceinline__ uint32_t lv_pack_f16x2(float lo, float hi) {
    lv_f2 f; f.x = lo; f.y = hi;
    const lv_h2 h = __builtin_convertvector(f, lv_h2);
    return __builtin_bit_cast(uint32_t, h);
}
#endif

#include <stdint.h>
#include <string.h>

// Measurement tooling (profiles/microbench/build_trace.sh builds a SEPARATE library with -DLV_TRACE; the product build has no
// trace code): LV_TRACE_MARK(step, i) stores the shader clock of one chosen lane into a caller-provided buffer, for phase
// breakdowns inside the persistent kernels.
#if defined(LV_TRACE) && !defined(LV_EMU)
extern __device__ unsigned long long* lv_trace_buf;
#define LV_TRACE_MARK(step, i)                                                                                              \
    do {                                                                                                                    \
        if (lv_trace_buf && blockIdx.x == 8 && threadIdx.x == 0) lv_trace_buf[(long)(step) * 8 + (i)] = clock64();         \
    } while (0)
#define LV_TRACE_VAL(step, i, v)                                                                                           \
    do {                                                                                                                    \
        if (lv_trace_buf && blockIdx.x == 8 && threadIdx.x == 0) lv_trace_buf[(long)(step) * 8 + (i)] = (unsigned long long)(v); \
    } while (0)
#define LV_TRACE_ONLY(...) __VA_ARGS__
#else
#define LV_TRACE_MARK(step, i) do { } while (0)
#define LV_TRACE_VAL(step, i, v) do { } while (0)
#define LV_TRACE_ONLY(...)
#endif

#define LV_WAVE 64

// (q0 + ro) mod m for 0 <= q0 < m, 0 <= ro < 32, without a division per element: the GEMM epilogue addends are indexed by
// row % mod, and one (hardware-less) 32-bit modulo per output element cost the K = 512 input projection 55 -> 82 us.
__device__ __forceinline__ int lv_wrap_row(int q0, int ro, int m) {
    if (m == 1) return 0;
    const int q = q0 + ro;
    if (m >= 32) return q >= m ? q - m : q;
    return q % m;
}

// ---- status codes returned through the C ABI (0 = ok, >0 = hipError_t, <0 = argument check) ----
#define LV_OK 0
#define LV_ERR_ARG (-1)
#define LV_ERR_SHAPE (-2)
#define LV_ERR_ALIGN (-3)
#define LV_ERR_UNSUPPORTED (-4)

#define LV_CHECK_LAUNCH()                        \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

// f32 -> bf16 bits, round-to-nearest-even on the integer pipe (finite inputs).  A/B on MI355X
// (profiles/r01_microbench_gemm_shapes.txt): this is 1.6-1.7x faster in the bf16 GEMM's staging path than the
// (__bf16) cast route hipcc lowers through v_cvt_pk_bf16_f32 (logits GEMM 813 us vs 1389 us).
__device__ __forceinline__ uint32_t lv_f32_to_bf16_bits(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float lv_bf16_bits_to_f32(uint32_t b) {
    const uint32_t u = b << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
__device__ __forceinline__ uint32_t lv_pack_bf16x2(float lo, float hi) {
    return lv_f32_to_bf16_bits(lo) | (lv_f32_to_bf16_bits(hi) << 16);
}

__device__ __forceinline__ float lv_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// Throughput-path activations (bf16 recurrent kernels only): hardware exp2 / reciprocal, ~1e-6 absolute error --
// three orders below the bf16 rounding of the operands they sit next to.  The f32 parity path keeps expf / tanhf.
#ifdef LV_EMU
__device__ __forceinline__ float lv_sigmoid_fast(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float lv_exp_fast(float x) { return expf(x); }
#else
// (v_rcp_f32 through the builtin: __frcp_rn is the correctly rounded reciprocal and expands to the full division sequence --
//  v_div_scale / v_rcp / 4 x fma / v_div_fmas / v_div_fixup, ~10 VALU instructions per activation in the recurrences' epilogue)
__device__ __forceinline__ float lv_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// hardware exp2 (v_exp_f32), ~2 ulp: for values that are rounded to bf16 right afterwards
__device__ __forceinline__ float lv_exp_fast(float x) { return __expf(x); }
#endif
__device__ __forceinline__ float lv_tanh_fast(float x) { return 2.0f * lv_sigmoid_fast(2.0f * x) - 1.0f; }

template <class T>
__device__ __forceinline__ T lv_wave_sum(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ float lv_wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}

__device__ __forceinline__ bool lv_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static inline int lv_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
