// hip_emu.h -- TEST-ONLY thread-level emulator for the gfx950 kernel sources.
//
// The product is the hipcc build of vae_lagging_encoder_amd/csrc/*.hip.  The GPU-less CI box
// cannot run those kernels, so the `-m "not gpu"` tests compile the SAME .hip files with g++
// (-DLV_EMU) against this header to check index math, MFMA fragment layouts, LDS staging and
// host-side launch sequencing.  Nothing in the package imports or loads the emulator build;
// see tests/emu/README.md.
//
// Model: every HIP thread is a user-level FIBER (own stack, hand-rolled x86-64 context switch) scheduled round-robin on
// the ONE OS thread that called the launch; a barrier is "yield until the generation changes", so a __syncthreads()
// over 256 threads costs 256 context switches of a few ns instead of 256 futex round trips (the first version of this
// emulator used pooled OS threads and spent 80 % of its time in the kernel).  A plain launch runs one workgroup at a
// time; a CONCURRENT launch (lv_emu::launch_concurrent) keeps every workgroup of the grid live at once, which is what
// the spin-synchronised persistent kernels need (their polls yield).  wave64 cross-lane ops (shuffles, MFMA)
// rendezvous on a per-wave barrier and exchange operands through per-wave buffers.
// MFMA lane->element maps follow /opt/skills/guides/cdna_hip_programming.md section 3:
//   16x16x4 f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)*4+r][col=l&15]
//   32x32x2 f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
// and the arithmetic is the k-ordered fmaf chain the hardware is documented to produce.
#pragma once
#include <sys/mman.h>
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

// switch stacks: save the callee-saved registers on the current stack, park its pointer in *save_sp, adopt load_sp
extern "C" void lv_emu_switch(void** save_sp, void* load_sp);

namespace lv_emu {

constexpr int kMaxThreads = 1024;
constexpr int kWave = 64;
constexpr size_t kStack = 256 << 10;      // per fiber; mmap'd, only touched pages are ever backed

struct Bar {                               // generation barrier over `count` fibers
    int count = 0, arrived = 0;
    unsigned gen = 0;
};

struct WaveX {
    Bar bar;
    int base = 0;                          // first fiber of the wave (index into the live set)
    int size = 0;                          // lanes of the wave (bar.count drops as lanes exit)
    float fa[kWave], fb[kWave];
    uint64_t u[kWave];
    uint4 qa[kWave], qb[kWave];
};

struct Block {                             // one live workgroup
    lv_emu_idx idx{0, 0, 0};
    Bar bar;
    int base = 0;                          // first fiber of the block
    std::vector<WaveX> waves;
    std::vector<char> shared;              // `__shared__` storage of the block in a concurrent launch (see LV_SHARED)
};

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    lv_emu_idx tidx{0, 0, 0};
    int lin = 0;                           // linear thread id in its block
    Block* blk = nullptr;
    bool done = true;
};

struct State {
    std::recursive_mutex launch_mu;
    std::vector<Fiber> fibers;             // the live set (one block, or the whole grid in a concurrent launch)
    std::vector<int> ring_next, ring_prev; // circular list of the fibers that have not finished (global yields walk it in O(1))
    std::vector<Block> blocks;
    size_t nstacks = 0;
    char* stacks = nullptr;
    int cur = -1, nlive = 0, remaining = 0;
    bool concurrent = false;               // every workgroup live at once: all waits pass the turn along the global ring
    void* main_sp = nullptr;
    std::function<void()> job;
    dim3 grid, block;
    std::vector<char> dyn;
};

extern State g_state;
extern lv_emu_idx t_threadIdx;
extern lv_emu_idx t_blockIdx;
extern int t_lin;
inline State& st() { return g_state; }

void run_live(int nlive);                  // hip_emu_impl.cpp

inline void enter(Fiber& f) {              // make `f` the running fiber's identity
    t_threadIdx = f.tidx;
    t_lin = f.lin;
    t_blockIdx = f.blk->idx;
}

// hand the OS thread to fiber `next` and come back when somebody hands it back
inline void switch_to(int next) {
    State& s = st();
    Fiber* me = &s.fibers[s.cur];
    s.cur = next;
    lv_emu_switch(&me->sp, s.fibers[next].sp);
    enter(*me);
}

// next runnable fiber in [base, base+n) after the running one, cyclically; -1 when it is the only one
inline int next_in(int base, int n) {
    State& s = st();
    int i = s.cur - base;
    for (int k = 1; k < n; ++k) {
        int j = base + (i + k) % n;
        if (!s.fibers[j].done) return j;
    }
    return -1;
}

inline void yield_in(int base, int n) {
    int nx = next_in(base, n);
    if (nx >= 0) switch_to(nx);
}
inline void yield_all() {
    State& s = st();
    const int nx = s.ring_next[s.cur];
    if (nx != s.cur) switch_to(nx);
}

inline void bar_wait(Bar& b, int base, int span) {
    const unsigned g = b.gen;
    if (++b.arrived >= b.count) { b.arrived = 0; ++b.gen; return; }
    while (b.gen == g) {
        // In a concurrent launch a waiter hands the turn to its GLOBAL successor: switching inside the block / wave would trap
        // the scheduler there as soon as one of its fibers polls for another workgroup's data (the poller's ring successor is
        // a local waiter, which would hand the turn straight back).
        State& s = st();
        int nx = s.concurrent ? (s.ring_next[s.cur] != s.cur ? s.ring_next[s.cur] : -1) : next_in(base, span);
        if (nx < 0) { fprintf(stderr, "lv_emu: barrier can never complete (divergent barrier?)\n"); abort(); }
        switch_to(nx);
    }
}
// a thread that exits stops counting towards its block's and wave's barriers (as exited waves do on the hardware)
inline void bar_leave(Bar& b) {
    --b.count;
    if (b.count > 0 && b.arrived >= b.count) { b.arrived = 0; ++b.gen; }
}

template <class F>
inline void launch_impl(dim3 grid, dim3 block, size_t shmem, F&& f, bool concurrent) {
    State& s = st();
    std::lock_guard<std::recursive_mutex> lk(s.launch_mu);
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > kMaxThreads) { fprintf(stderr, "lv_emu: bad block size %d\n", nthreads); abort(); }
    const long nblocks = (long)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    s.grid = grid; s.block = block;
    s.dyn.assign(shmem + 64, 0);
    s.job = std::function<void()>(f);
    s.concurrent = concurrent;
    const int live_blocks = concurrent ? (int)nblocks : 1;
    const int nlive = live_blocks * nthreads;
    if ((size_t)nlive > s.nstacks) {
        if (s.stacks) munmap(s.stacks, s.nstacks * kStack);
        s.stacks = (char*)mmap(nullptr, (size_t)nlive * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (s.stacks == (char*)MAP_FAILED) { fprintf(stderr, "lv_emu: cannot map %d fiber stacks\n", nlive); abort(); }
        s.nstacks = (size_t)nlive;
    }
    s.fibers.assign((size_t)nlive, Fiber());
    s.blocks.assign((size_t)live_blocks, Block());
    const int nw = (nthreads + kWave - 1) / kWave;
    for (int b = 0; b < live_blocks; ++b) {
        Block& B = s.blocks[b];
        B.base = b * nthreads;
        B.bar.count = nthreads;
        B.waves.assign((size_t)nw, WaveX());
        for (int w = 0; w < nw; ++w) {
            B.waves[w].size = B.waves[w].bar.count = std::min(kWave, nthreads - w * kWave);
            B.waves[w].base = B.base + w * kWave;
        }
        for (int t = 0; t < nthreads; ++t) {
            Fiber& fb = s.fibers[B.base + t];
            fb.stack = s.stacks + (size_t)(B.base + t) * kStack;
            fb.lin = t;
            fb.tidx.x = t % block.x;
            fb.tidx.y = (t / block.x) % block.y;
            fb.tidx.z = t / (block.x * block.y);
            fb.blk = &B;
        }
    }
    if (concurrent) {
        long b = 0;
        for (unsigned gz = 0; gz < grid.z; ++gz)
            for (unsigned gy = 0; gy < grid.y; ++gy)
                for (unsigned gx = 0; gx < grid.x; ++gx, ++b) s.blocks[b].idx = lv_emu_idx{gx, gy, gz};
        run_live(nlive);
    } else {
        for (unsigned gz = 0; gz < grid.z; ++gz)
            for (unsigned gy = 0; gy < grid.y; ++gy)
                for (unsigned gx = 0; gx < grid.x; ++gx) {
                    Block& B = s.blocks[0];
                    B.idx = lv_emu_idx{gx, gy, gz};
                    B.bar = Bar(); B.bar.count = nthreads;
                    for (auto& w : B.waves) { w.bar = Bar(); w.bar.count = w.size; }
                    run_live(nlive);
                }
    }
}

template <class F> inline void launch(dim3 grid, dim3 block, size_t shmem, F&& f) { launch_impl(grid, block, shmem, f, false); }
template <class F> inline void launch_concurrent(dim3 grid, dim3 block, size_t shmem, F&& f) { launch_impl(grid, block, shmem, f, true); }

inline char* dyn_smem() {
    State& s = st();
    uintptr_t p = (uintptr_t)s.dyn.data();
    return (char*)((p + 15) & ~(uintptr_t)15);
}

inline Fiber& me() { State& s = st(); return s.fibers[s.cur]; }
inline WaveX& my_wave() { Fiber& f = me(); return f.blk->waves[f.lin / kWave]; }
inline int lane() { return t_lin % kWave; }
inline void wave_sync() { WaveX& w = my_wave(); bar_wait(w.bar, w.base, w.size); }

// per-block storage for a kernel's shared arrays in a concurrent launch (a function-local `static` would be one copy
// for the whole grid): the first caller of a block sizes it, every thread of the block gets the same pointer
inline void* block_shared(size_t bytes) {
    Block* b = me().blk;
    if (b->shared.size() < bytes + 64) b->shared.assign(bytes + 64, 0);
    uintptr_t p = (uintptr_t)b->shared.data();
    return (void*)((p + 63) & ~(uintptr_t)63);
}

}  // namespace lv_emu

// In a plain launch workgroups run one after another, so a function-local static IS the block's LDS.
#define __shared__ static

#define threadIdx (lv_emu::t_threadIdx)
#define blockIdx (lv_emu::t_blockIdx)
#define blockDim (lv_emu::st().block)
#define gridDim (lv_emu::st().grid)

static inline void __syncthreads() {
    lv_emu::Fiber& f = lv_emu::me();
    lv_emu::bar_wait(f.blk->bar, f.blk->base, (int)(lv_emu::st().block.x * lv_emu::st().block.y * lv_emu::st().block.z));
}

// ---- wave64 cross-lane -------------------------------------------------------------------
template <class T>
static inline T lv_emu_xchg(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    w.u[l] = bits;
    lv_emu::wave_sync();
    T out = v;
    if (src_lane >= 0 && src_lane < w.size) { uint64_t b = w.u[src_lane]; memcpy(&out, &b, sizeof(T)); }
    lv_emu::wave_sync();
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
// wave vote: true iff the predicate holds on every lane of the wave
static inline bool __all(bool p) {
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    w.u[l] = p ? 1u : 0u;
    lv_emu::wave_sync();
    bool r = true;
    for (int i = 0; i < w.size; ++i) r = r && (lv_emu::st().fibers[w.base + i].done || w.u[i] != 0);
    lv_emu::wave_sync();
    return r;
}

// ---- MFMA (f32 in / f32 accumulate), bit-for-bit the k-ordered fmaf chain -------------------
static inline f32x4 lv_emu_mfma_16x16x4(float a, float b, f32x4 c) {
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    w.fa[l] = a; w.fb[l] = b;
    lv_emu::wave_sync();
    f32x4 d = c;
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.fa[row + 16 * k], w.fb[col + 16 * k], acc);
        d[r] = acc;
    }
    lv_emu::wave_sync();
    return d;
}
static inline f32x16 lv_emu_mfma_32x32x2(float a, float b, f32x16 c) {
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    w.fa[l] = a; w.fb[l] = b;
    lv_emu::wave_sync();
    f32x16 d = c;
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(w.fa[row + 32 * k], w.fb[col + 32 * k], acc);
        d[r] = acc;
    }
    lv_emu::wave_sync();
    return d;
}

// v_mfma_f32_32x32x16_bf16: lane l holds 8 bf16 of A row (l&31) and of B column (l&31) for k = 8*(l>>5)+e
// (any k assignment that is the same for A and B gives the same product); D layout as the f32 32x32 form.
static inline float lv_emu_bf16_to_f32(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }
static inline f32x16 lv_emu_mfma_32x32x16_bf16(uint4 a, uint4 b, f32x16 c) {
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    w.qa[l] = a; w.qb[l] = b;
    lv_emu::wave_sync();
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
    lv_emu::wave_sync();
    return d;
}

// v_mfma_f32_16x16x32_bf16: lane l holds 8 bf16 of A row (l&15) / B column (l&15) for k = 8*(l>>4)+e; D as 16x16x4
static inline f32x4 lv_emu_mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c) {
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    w.qa[l] = a; w.qb[l] = b;
    lv_emu::wave_sync();
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
    lv_emu::wave_sync();
    return d;
}

// v_mfma_f32_4x4x4_16b_bf16: block = l >> 2; lane holds 4 bf16 (k) of A row (l & 3) / B column (l & 3); D[r][l & 3] in register r
static inline f32x4 lv_emu_mfma_4x4x4_16b_bf16(uint2 a, uint2 b, f32x4 c) {
    auto& w = lv_emu::my_wave();
    int l = lv_emu::lane();
    w.qa[l] = make_uint4(a.x, a.y, 0, 0); w.qb[l] = make_uint4(b.x, b.y, 0, 0);
    lv_emu::wave_sync();
    f32x4 d = c;
    const int blk = l & ~3;
    unsigned short eb[4];
    memcpy(eb, &w.qb[l], 8);
    for (int r = 0; r < 4; ++r) {
        unsigned short ea[4];
        memcpy(ea, &w.qa[blk + r], 8);
        float acc = c[r];
        for (int e = 0; e < 4; ++e) acc = fmaf(lv_emu_bf16_to_f32(ea[e]), lv_emu_bf16_to_f32(eb[e]), acc);
        d[r] = acc;
    }
    lv_emu::wave_sync();
    return d;
}

// ---- runtime API subset used by the host side of the C ABI -------------------------------
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum { hipMemcpyDeviceToDevice = 3 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memmove(d, s, n); return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
