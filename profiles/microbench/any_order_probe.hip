// any_order_probe.hip -- does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let a kernel start beside its predecessor on
// the SAME stream on gfx950 / ROCm 7.2?  hip_ext.h says the flag "is not supported on AMD GFX9xx boards" for the module-launch form;
// this measures it.  Two kinds of kernels: `spin` (one workgroup busy for a fixed number of shader-clock ticks: pure latency, what
// the small glue kernels of a training step are) and `stream` (a grid-filling copy: what a full-chip kernel is).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o any_order_probe any_order_probe.hip && ./any_order_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(long long ticks, int* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (out && threadIdx.x == 0) out[blockIdx.x] = 1;
}

__global__ void stream_kernel(const float4* __restrict__ a, float4* __restrict__ b, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}

// a dependent pair: the second kernel reads what the first wrote (must NOT be launched any-order; here to show what breaks)
__global__ void produce_kernel(int* buf, int n, int v, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    for (int i = threadIdx.x; i < n; i += blockDim.x) buf[i] = v;
}
__global__ void consume_kernel(const int* buf, int n, int v, int* bad) {
    int b = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) b += buf[i] != v;
    if (b) atomicAdd(bad, b);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class F> static float time_us(hipStream_t s, int reps, F body) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) body();
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) body();
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 1e3f * ms / reps;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    int* out; CK(hipMalloc(&out, 4096 * sizeof(int)));
    const long n4 = 64L << 20 >> 4;                     // 64 MB
    float4 *a, *b; CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16));
    CK(hipMemset(a, 0, n4 * 16));
    int wc_khz = 0; CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0));
    const long long ticks10 = (long long)wc_khz * 10 / 1000;      // 10 us
    printf("wall clock %d kHz\n", wc_khz);
    const int reps = 200;
    for (int k = 2; k <= 5; ++k) {
        float t_norm = time_us(s, reps, [&]() {
            for (int i = 0; i < k; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks10, out);
        });
        float t_any = time_us(s, reps, [&]() {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks10, out);
            for (int i = 1; i < k; ++i) hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks10, out);
        });
        float t_ext0 = time_us(s, reps, [&]() {
            for (int i = 0; i < k; ++i) hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, ticks10, out);
        });
        printf("%d x 10-us one-workgroup kernels on one stream: ordinary %.2f us | first ordinary, rest any-order %.2f us | ext launch, flags 0 %.2f us\n",
               k, t_norm, t_any, t_ext0);
    }
    {   // a grid-filling copy with a small latency kernel behind it
        float t_norm = time_us(s, reps, [&]() {
            hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s, a, b, n4);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks10, out);
        });
        float t_any = time_us(s, reps, [&]() {
            hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s, a, b, n4);
            hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks10, out);
        });
        float t_alone = time_us(s, reps, [&]() { hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s, a, b, n4); });
        printf("64 MB copy (2048 workgroups) + 10-us kernel: ordinary %.2f us | any-order %.2f us | the copy alone %.2f us\n", t_norm, t_any, t_alone);
    }
    {   // ordering that must still hold: an ORDINARY kernel behind an any-order one waits for everything in front of it
        int* buf; int* bad; CK(hipMalloc(&buf, 4096 * sizeof(int))); CK(hipMalloc(&bad, sizeof(int))); CK(hipMemset(bad, 0, sizeof(int)));
        for (int it = 0; it < 200; ++it) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks10, out);
            hipExtLaunchKernelGGL(produce_kernel, dim3(1), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, buf, 4096, it + 1, 2 * ticks10);
            hipLaunchKernelGGL(consume_kernel, dim3(1), dim3(256), 0, s, (const int*)buf, 4096, it + 1, bad);      // ordinary: behind both
        }
        CK(hipStreamSynchronize(s));
        int h = -1; CK(hipMemcpy(&h, bad, sizeof(int), hipMemcpyDeviceToHost));
        printf("ordinary consumer behind an any-order producer: %d stale reads in 200 rounds (must be 0)\n", h);
        // ... and what any-order does to a DEPENDENT pair (expected: stale reads -- the reason the flag is per launch)
        CK(hipMemset(bad, 0, sizeof(int)));
        for (int it = 0; it < 200; ++it) {
            hipLaunchKernelGGL(produce_kernel, dim3(1), dim3(256), 0, s, buf, 4096, 1000 + it, 2 * ticks10);
            hipExtLaunchKernelGGL(consume_kernel, dim3(1), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, (const int*)buf, 4096, 1000 + it, bad);
        }
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(&h, bad, sizeof(int), hipMemcpyDeviceToHost));
        printf("any-order consumer behind its producer: %d stale reads in 200 rounds (> 0 shows the flag is honoured)\n", h);
    }
    {   // under stream capture: is the launch recorded, and does the replay keep / drop the overlap?
        hipGraph_t g; hipGraphExec_t ge;
        hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks10, out);
            hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks10, out);
            hipError_t e2 = hipGetLastError(), e3 = hipSuccess;
            hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks10, out);
            e3 = hipStreamEndCapture(s, &g);
            printf("capture: ext launch -> %s, end capture -> %s\n", hipGetErrorString(e2), hipGetErrorString(e3));
            if (e3 == hipSuccess && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
                float t = time_us(s, reps, [&]() { hipGraphLaunch(ge, s); });
                printf("graph replay of (ordinary + 2 any-order) 10-us kernels: %.2f us per replay\n", t);
            }
        }
    }
    return 0;
}
