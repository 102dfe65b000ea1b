// Microbenchmark (measurement tooling, not product): what would one timestep of a PERSISTENT LSTM recurrence cost in
// synchronisation on MI355X?  256 workgroups (one per CU) in 8 groups of 32 (group = blockIdx % 8, i.e. one XCD if the
// usual round-robin placement holds; correctness does not depend on it).  Per step every wave publishes its share of a
// [ROWS][H] bf16 state as 8-byte {data, tag} granules (agent-scope relaxed 64-bit atomic stores = sc1 write-through) and
// then gathers the WHOLE state of its group by polling the tags (agent-scope 64-bit loads).  All spins are bounded.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/xcd_gather_probe.hip -o profiles/microbench/xcd_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// state granules: buf[parity][group][ngran]; granule = (tag << 32) | payload(2 bf16)
template <int GRAN_PER_WAVE_OUT, int MODE>
__global__ __launch_bounds__(256) void gather_probe(unsigned long long* buf, int ngran, int steps, int* err, float* sink) {
    const int group = blockIdx.x % 8, member = blockIdx.x / 8;       // 32 members per group
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int wave_in_group = member * 4 + w;                         // 0..127
    float acc = 0.f;
    for (int t = 1; t <= steps; ++t) {
        unsigned long long* cur = buf + ((size_t)(t & 1) * 8 + group) * ngran;
        // publish this wave's share (GRAN_PER_WAVE_OUT granules, lanes 0..GRAN-1)
        if (l < GRAN_PER_WAVE_OUT) {
            const unsigned payload = (unsigned)(wave_in_group * 131 + l + t);
            const unsigned long long g = ((unsigned long long)(unsigned)t << 32) | payload;
            __hip_atomic_store(cur + wave_in_group * GRAN_PER_WAVE_OUT + l, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // gather the whole group state: ngran granules, 64 lanes x (ngran/64) each, poll until every tag == t
        // (ONE_WAVE: only wave 0 of the workgroup gathers -- into LDS in a real kernel -- the others wait at the barrier)
        unsigned sum = 0;
        // MODE 0: every wave gathers everything; 1: wave 0 gathers everything; 2: each of the 4 waves gathers a quarter
        const int gbeg = MODE == 2 ? w * (ngran / 4) : 0;
        const int gend = MODE == 2 ? gbeg + ngran / 4 : ngran;
        for (int base = gbeg; base < gend && (MODE != 1 || w == 0); base += 64 * 8) {
            unsigned long long v[8];
            int spins = 0;
            bool ok;
            do {
                ok = true;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int idx = base + j * 64 + l;
                    v[j] = idx < gend ? __hip_atomic_load(cur + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : ((unsigned long long)(unsigned)t << 32);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) ok = ok && ((unsigned)(v[j] >> 32) == (unsigned)t);
                ok = __all(ok);
                if (++spins > (1 << 20)) { if (l == 0) atomicExch(err, t); return; }
            } while (!ok);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += (unsigned)v[j];
        }
        acc += (float)(sum & 0xFF);
        if (MODE) __syncthreads();
    }
    if (acc == 12345.678f) sink[blockIdx.x * 256 + tid] = acc;
}

// where do consecutive blocks land?  (HW_REG_XCC_ID; on this machine block b runs on XCC (b + 7) % 8, so the
// kernels' groups = blockIdx % 8 are XCD-local; agent-scope accesses keep them correct wherever they land)
__global__ void xcc_probe(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg(6164) & 0xF; }

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    float* sink; CK(hipMalloc(&sink, 256 * 256 * 4));
    const int steps = 400;
    { int* xo; CK(hipMalloc(&xo, 256 * 4)); hipLaunchKernelGGL(xcc_probe, dim3(256), dim3(64), 0, s, xo); CK(hipStreamSynchronize(s)); int h[256]; CK(hipMemcpy(h, xo, 1024, hipMemcpyDeviceToHost)); int bad = 0; for (int i = 0; i < 256; ++i) if (h[i] != h[i % 8]) ++bad; printf("XCC_ID of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("  | blocks whose XCC differs from block (i %% 8): %d of 256\n", bad); }
    for (int variant = 0; variant < 9; ++variant) {
        // variant 0: 8 KB payload per group (4 rows x 1024 x bf16 = 2048 granules): forward-like
        // variant 1: 16 KB payload (4096 granules): reduce-scatter-like volume
        // variant 2: 32 KB payload (8192 granules): dG-gather-like volume
        const int ngran = 2048 << (variant % 3);
        unsigned long long* buf; CK(hipMalloc(&buf, (size_t)2 * 8 * ngran * 8)); CK(hipMemset(buf, 0, (size_t)2 * 8 * ngran * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(buf, 0, (size_t)2 * 8 * ngran * 8, s));
            CK(hipEventRecord(e0, s));
            if (variant == 0) hipLaunchKernelGGL((gather_probe<16, 0>), dim3(256), dim3(256), 0, s, buf, ngran, steps, err, sink);
            if (variant == 1) hipLaunchKernelGGL((gather_probe<32, 0>), dim3(256), dim3(256), 0, s, buf, ngran, steps, err, sink);
            if (variant == 2) hipLaunchKernelGGL((gather_probe<64, 0>), dim3(256), dim3(256), 0, s, buf, ngran, steps, err, sink);
            if (variant == 3) hipLaunchKernelGGL((gather_probe<16, 1>), dim3(256), dim3(256), 0, s, buf, ngran, steps, err, sink);
            if (variant == 4) hipLaunchKernelGGL((gather_probe<32, 1>), dim3(256), dim3(256), 0, s, buf, ngran, steps, err, sink);
            if (variant == 5) hipLaunchKernelGGL((gather_probe<64, 1>), dim3(256), dim3(256), 0, s, buf, ngran, steps, err, sink);
            if (variant == 6) hipLaunchKernelGGL((gather_probe<16, 2>), dim3(256), dim3(256), 0, s, buf, ngran, steps, err, sink);
            if (variant == 7) hipLaunchKernelGGL((gather_probe<32, 2>), dim3(256), dim3(256), 0, s, buf, ngran, steps, err, sink);
            if (variant == 8) hipLaunchKernelGGL((gather_probe<64, 2>), dim3(256), dim3(256), 0, s, buf, ngran, steps, err, sink);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int h_err; CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
            printf("payload %2d KB per group, %s gather, %d steps: %8.2f us/step  (err=%d)\n", ngran * 4 / 1024, variant >= 6 ? "quarter-per-wave" : (variant >= 3 ? "one-wave" : "every-wave"), steps, ms * 1e3f / steps, h_err);
        }
        CK(hipFree(buf));
    }
    return 0;
}
