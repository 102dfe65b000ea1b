// Microbenchmark (measurement tooling, not product): what would ONE in-kernel BatchNorm statistics exchange cost inside a fused,
// persistent PixelCNN residual-block kernel?  BatchNorm (train mode) needs per-channel (sum, sum of squares)
// over ALL N*H*W pixels before any pixel can be normalised: a grid-wide dependency, three times per residual block
// (dec_pixelcnn_v2.py:32-62).  Between kernels the launch boundary is that dependency (1.5-1.9 us on this chip); inside one
// persistent kernel it has to be a hand-off across all 8 XCDs, i.e. through memory (agent-scope, write-through granules):
//   1. every workgroup publishes its P = 2 C partial sums as tagged 8-byte granules {float, tag};
//   2. reducer workgroup p (p < P) gathers partial p of all G workgroups, adds them in a FIXED order (the results must be
//      bit-reproducible: no float atomics) and publishes the total;
//   3. every workgroup polls the P totals.
// Two dependent cross-XCD hand-offs per exchange.  Prints microseconds per exchange for G = 200 / 256 workgroups and C = 32 / 64.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/bn_exchange_probe.hip -o profiles/microbench/bn_exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ void put(u64* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 get(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// part[parity][G][P], tot[parity][P]
__global__ __launch_bounds__(256) void bn_exchange(u64* part, u64* tot, int G, int P, int rounds, int work_iters, int* err, float* sink) {
    __shared__ float red[4];
    __shared__ float totals[128];
    const int b = blockIdx.x, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    float carry = 1.0f + b * 1e-3f;
    for (int r = 1; r <= rounds; ++r) {
        // stand-in for the convolution between two exchanges (keeps the workgroups from arriving in lock step for free)
        for (int i = 0; i < work_iters; ++i) carry = carry * 1.0000001f + 1e-7f;
        u64* mypart = part + ((size_t)(r & 1) * G + b) * P;
        if (tid < P) put(mypart + tid, carry + tid, (unsigned)r);
        if (b < P) {                                   // reducer of partial b: G granules, G / 256 (<= 1) per thread, fixed-order tree
            const u64* col = part + (size_t)(r & 1) * G * P + b;
            float v = 0.f;
            int spins = 0;
            if (tid < G) {
                u64 g;
                do {
                    g = get(col + (size_t)tid * P);
                    if (++spins > (1 << 22)) { atomicExch(err, r); return; }
                } while ((unsigned)(g >> 32) != (unsigned)r);
                v = __uint_as_float((unsigned)g);
            }
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (l == 0) red[w] = v;
            __syncthreads();
            if (tid == 0) put(tot + (size_t)(r & 1) * P + b, (red[0] + red[1]) + (red[2] + red[3]), (unsigned)r);
        }
        if (tid < P) {                                 // everybody: the P totals
            const u64* tp = tot + (size_t)(r & 1) * P + tid;
            u64 g;
            int spins = 0;
            do {
                g = get(tp);
                if (++spins > (1 << 22)) { atomicExch(err, 1000000 + r); return; }
            } while ((unsigned)(g >> 32) != (unsigned)r);
            totals[tid] = __uint_as_float((unsigned)g);
        }
        __syncthreads();
        carry += totals[l % P] * 1e-9f;
    }
    if (carry == 12345.678f) sink[b * 256 + tid] = carry;
}

int main() {
    int dev = 0;
    CK(hipSetDevice(dev));
    u64 *part, *tot; int* err; float* sink;
    const int GMAX = 256, PMAX = 128;
    CK(hipMalloc(&part, 2ull * GMAX * PMAX * 8)); CK(hipMalloc(&tot, 2ull * PMAX * 8));
    CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, GMAX * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int work : {0, 2000}) {
        for (int G : {200, 256}) {
            for (int C : {32, 64}) {
                const int P = 2 * C, rounds = 2000;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(part, 0, 2ull * GMAX * PMAX * 8)); CK(hipMemset(tot, 0, 2ull * PMAX * 8)); CK(hipMemset(err, 0, 4));
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(bn_exchange, dim3(G), dim3(256), 0, 0, part, tot, G, P, rounds, work, err, sink);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                printf("work %4d iterations between exchanges, G = %3d workgroups, C = %2d channels (%3d partials): %.2f us per round%s\n",
                       work, G, C, P, 1e3f * best / rounds, herr ? "  [TIMEOUT]" : "");
            }
        }
    }
    printf("(a round = the stand-in work + one exchange; the difference between the two `work` settings at the same G, C is the work itself,\n"
           " so `work 0` rows are the exchange alone: two dependent cross-XCD hand-offs + the reducers' fixed-order sums)\n");
    return 0;
}
