// hip_emu.h -- TEST-ONLY thread-level emulator for the gfx950 kernel sources.
//
// The product is the hipcc build of vae_lagging_encoder_amd/csrc/*.hip.  The GPU-less CI box
// cannot run those kernels, so the `-m "not gpu"` tests compile the SAME .hip files with g++
// (-DLV_EMU) against this header to check index math, MFMA fragment layouts, LDS staging and
// host-side launch sequencing.  Nothing in the package imports or loads the emulator build;
// see tests/emu/README.md.
//
// Model: one workgroup at a time; every HIP thread of the workgroup is a pooled OS thread;
// __syncthreads() is a pthread barrier over the workgroup; wave64 cross-lane ops (shuffles,
// MFMA) rendezvous on a per-wave barrier and exchange operands through per-wave buffers.
// MFMA lane->element maps follow /opt/skills/guides/cdna_hip_programming.md section 3:
//   16x16x4 f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)*4+r][col=l&15]
//   32x32x2 f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
// and the arithmetic is the k-ordered fmaf chain the hardware is documented to produce.
#pragma once
#include <pthread.h>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct lv_emu_idx { unsigned x, y, z; };

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

typedef float f32x4 __attribute__((vector_size(16)));
typedef float f32x16 __attribute__((vector_size(64)));

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

namespace lv_emu {

constexpr int kMaxThreads = 1024;
constexpr int kWave = 64;

struct WaveX {
    pthread_barrier_t bar;
    int count = 0;
    float fa[kWave], fb[kWave];
    uint64_t u[kWave];
    uint4 qa[kWave], qb[kWave];
};

struct State {
    std::mutex launch_mu;
    pthread_barrier_t wg_bar;       // __syncthreads / end-of-block
    pthread_barrier_t start_bar;    // pool start/finish (nthreads+1)
    int nthreads = 0;               // threads in current config
    std::vector<pthread_t> pool;
    std::function<void()> job;
    dim3 grid, block;
    std::vector<WaveX*> waves;
    std::vector<char> dyn;
    bool shutdown = false;
};

inline State& st() { static State s; return s; }

extern thread_local lv_emu_idx t_threadIdx;
extern thread_local lv_emu_idx t_blockIdx;
extern thread_local int t_lin;     // linear thread id in block

#ifdef LV_EMU_IMPL
thread_local lv_emu_idx t_threadIdx{0, 0, 0};
thread_local lv_emu_idx t_blockIdx{0, 0, 0};
thread_local int t_lin = 0;
#endif

inline void* worker(void* arg) {
    State& s = st();
    int tid = (int)(intptr_t)arg;
    for (;;) {
        pthread_barrier_wait(&s.start_bar);           // wait for a launch
        if (s.shutdown) return nullptr;
        unsigned bx = s.block.x, by = s.block.y;
        t_lin = tid;
        t_threadIdx.x = tid % bx;
        t_threadIdx.y = (tid / bx) % by;
        t_threadIdx.z = tid / (bx * by);
        for (unsigned gz = 0; gz < s.grid.z; ++gz)
            for (unsigned gy = 0; gy < s.grid.y; ++gy)
                for (unsigned gx = 0; gx < s.grid.x; ++gx) {
                    t_blockIdx.x = gx; t_blockIdx.y = gy; t_blockIdx.z = gz;
                    s.job();
                    pthread_barrier_wait(&s.wg_bar);  // block done before statics are reused
                }
        pthread_barrier_wait(&s.start_bar);           // signal completion
    }
}

inline void configure(int nthreads) {
    State& s = st();
    if (s.nthreads == nthreads) return;
    // tear down the old pool
    if (s.nthreads > 0) {
        s.shutdown = true;
        pthread_barrier_wait(&s.start_bar);
        for (auto& p : s.pool) pthread_join(p, nullptr);
        s.pool.clear();
        pthread_barrier_destroy(&s.start_bar);
        pthread_barrier_destroy(&s.wg_bar);
        for (auto* w : s.waves) { pthread_barrier_destroy(&w->bar); delete w; }
        s.waves.clear();
        s.shutdown = false;
    }
    s.nthreads = nthreads;
    pthread_barrier_init(&s.start_bar, nullptr, nthreads + 1);
    pthread_barrier_init(&s.wg_bar, nullptr, nthreads);
    int nw = (nthreads + kWave - 1) / kWave;
    for (int w = 0; w < nw; ++w) {
        WaveX* wx = new WaveX();
        wx->count = std::min(kWave, nthreads - w * kWave);
        pthread_barrier_init(&wx->bar, nullptr, wx->count);
        s.waves.push_back(wx);
    }
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 1 << 20);
    s.pool.resize(nthreads);
    for (int t = 0; t < nthreads; ++t)
        pthread_create(&s.pool[t], &attr, worker, (void*)(intptr_t)t);
    pthread_attr_destroy(&attr);
}

template <class F>
inline void launch(dim3 grid, dim3 block, size_t shmem, F&& f) {
    State& s = st();
    std::lock_guard<std::mutex> lk(s.launch_mu);
    int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > kMaxThreads) { fprintf(stderr, "lv_emu: bad block size %d\n", nthreads); abort(); }
    if (grid.x * grid.y * grid.z == 0) return;
    configure(nthreads);
    s.grid = grid; s.block = block;
    s.dyn.assign(shmem + 64, 0);
    s.job = std::function<void()>(f);
    pthread_barrier_wait(&s.start_bar);   // release workers
    pthread_barrier_wait(&s.start_bar);   // wait for completion
}

inline char* dyn_smem() {
    State& s = st();
    uintptr_t p = (uintptr_t)s.dyn.data();
    return (char*)((p + 15) & ~(uintptr_t)15);
}

inline WaveX& my_wave() { return *st().waves[t_lin / kWave]; }
inline int lane() { return t_lin % kWave; }

}  // namespace lv_emu

#define threadIdx (lv_emu::t_threadIdx)
#define blockIdx (lv_emu::t_blockIdx)
#define blockDim (lv_emu::st().block)
#define gridDim (lv_emu::st().grid)

static inline void __syncthreads() { pthread_barrier_wait(&lv_emu::st().wg_bar); }

// ---- wave64 cross-lane -------------------------------------------------------------------
template <class T>
static inline T lv_emu_xchg(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    w.u[l] = bits;
    pthread_barrier_wait(&w.bar);
    T out = v;
    if (src_lane >= 0 && src_lane < w.count) { uint64_t b = w.u[src_lane]; memcpy(&out, &b, sizeof(T)); }
    pthread_barrier_wait(&w.bar);
    return out;
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int l = lv_emu::lane();
    int src = l ^ mask;
    if ((src / width) != (l / width)) src = l;
    return lv_emu_xchg(v, src);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = lv_emu::lane();
    int src = l + (int)d;
    if ((src / width) != (l / width)) src = l;
    return lv_emu_xchg(v, src);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = lv_emu::lane();
    int src = l - (int)d;
    if (src < 0 || (src / width) != (l / width)) src = l;
    return lv_emu_xchg(v, src);
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = lv_emu::lane();
    int s = (l / width) * width + (src % width);
    return lv_emu_xchg(v, s);
}

// ---- MFMA (f32 in / f32 accumulate), bit-for-bit the k-ordered fmaf chain -------------------
static inline f32x4 lv_emu_mfma_16x16x4(float a, float b, f32x4 c) {
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    w.fa[l] = a; w.fb[l] = b;
    pthread_barrier_wait(&w.bar);
    f32x4 d = c;
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.fa[row + 16 * k], w.fb[col + 16 * k], acc);
        d[r] = acc;
    }
    pthread_barrier_wait(&w.bar);
    return d;
}
static inline f32x16 lv_emu_mfma_32x32x2(float a, float b, f32x16 c) {
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    w.fa[l] = a; w.fb[l] = b;
    pthread_barrier_wait(&w.bar);
    f32x16 d = c;
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(w.fa[row + 32 * k], w.fb[col + 32 * k], acc);
        d[r] = acc;
    }
    pthread_barrier_wait(&w.bar);
    return d;
}

// v_mfma_f32_32x32x16_bf16: lane l holds 8 bf16 of A row (l&31) and of B column (l&31) for k = 8*(l>>5)+e
// (any k assignment that is the same for A and B gives the same product); D layout as the f32 32x32 form.
static inline float lv_emu_bf16_to_f32(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }
static inline f32x16 lv_emu_mfma_32x32x16_bf16(uint4 a, uint4 b, f32x16 c) {
    static thread_local int dummy = 0; (void)dummy;
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    w.qa[l] = a; w.qb[l] = b;
    pthread_barrier_wait(&w.bar);
    f32x16 d = c;
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int h = 0; h < 2; ++h) {
            unsigned short ea[8], eb[8];
            memcpy(ea, &w.qa[row + 32 * h], 16);
            memcpy(eb, &w.qb[col + 32 * h], 16);
            for (int e = 0; e < 8; ++e) acc = fmaf(lv_emu_bf16_to_f32(ea[e]), lv_emu_bf16_to_f32(eb[e]), acc);
        }
        d[r] = acc;
    }
    pthread_barrier_wait(&w.bar);
    return d;
}

// v_mfma_f32_16x16x32_bf16: lane l holds 8 bf16 of A row (l&15) / B column (l&15) for k = 8*(l>>4)+e; D as 16x16x4
static inline f32x4 lv_emu_mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c) {
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    w.qa[l] = a; w.qb[l] = b;
    pthread_barrier_wait(&w.bar);
    f32x4 d = c;
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            unsigned short ea[8], eb[8];
            memcpy(ea, &w.qa[row + 16 * g], 16);
            memcpy(eb, &w.qb[col + 16 * g], 16);
            for (int e = 0; e < 8; ++e) acc = fmaf(lv_emu_bf16_to_f32(ea[e]), lv_emu_bf16_to_f32(eb[e]), acc);
        }
        d[r] = acc;
    }
    pthread_barrier_wait(&w.bar);
    return d;
}

// ---- runtime API subset used by the host side of the C ABI -------------------------------
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum { hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memmove(d, s, n); return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
