// Microbenchmark (measurement tooling): cost of a wave64 float4 load pattern "16 rows x 64 B" as a function of the row
// stride -- do power-of-two row strides camp on one L2 channel on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// every WG: 4 waves; wave w reads K range [w*256, w*256+256) of 16 W rows + 32 A rows, 48 float4 per lane (as the LSTM step)
__global__ __launch_bounds__(256) void loads_kernel(const float* __restrict__ A, const float* __restrict__ W, float* out,
                                                    long lda, long ldw, int wrows_per_wg) {
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int i = l & 15, kq = l >> 4;
    const float* wrow = W + ((long)blockIdx.x * wrows_per_wg + i) * ldw;
    const float* a0 = A + (long)i * lda;
    const float* a1 = A + (long)(16 + i) * lda;
    const int kbeg = w * 256;
    float4 v[48];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int kk = kbeg + 16 * it + 4 * kq;
        v[3 * it] = *reinterpret_cast<const float4*>(wrow + kk);
        v[3 * it + 1] = *reinterpret_cast<const float4*>(a0 + kk);
        v[3 * it + 2] = *reinterpret_cast<const float4*>(a1 + kk);
    }
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 48; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * 256 + tid] = acc.x;
}

// alternative mapping: one instruction = 4 rows x 256 B (lane = (row = l>>4, 16B chunk = l&15))
__global__ __launch_bounds__(256) void loads_kernel_wide(const float* __restrict__ A, const float* __restrict__ W, float* out,
                                                         long lda, long ldw, int wrows_per_wg) {
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int r4 = l >> 4, c = l & 15;
    const int kbeg = w * 256;
    float4 v[48];
#pragma unroll
    for (int q = 0; q < 48; ++q) {
        // 48 rows (16 W + 32 A) x 256 floats per wave = 12 row-groups of 4 rows x 4 column blocks of 64 floats
        const int rg = q >> 2, cb = q & 3;
        const int row = rg * 4 + r4;
        const float* base = row < 16 ? W + ((long)blockIdx.x * wrows_per_wg + row) * ldw : A + (long)(row - 16) * lda;
        v[q] = *reinterpret_cast<const float4*>(base + kbeg + cb * 64 + c * 4);
    }
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 48; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * 256 + tid] = acc.x;
}

template <class F>
float time_us(F&& f, int iters, hipStream_t s) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *A, *W, *out;
    CK(hipMalloc(&A, (size_t)64 << 20)); CK(hipMalloc(&W, (size_t)256 << 20)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(A, 0, (size_t)64 << 20)); CK(hipMemset(W, 0, (size_t)256 << 20));
    const int pads[] = {0, 4, 8, 16, 32, 64, 128, 256, 1024 + 64};
    for (int pa : pads) for (int pw : {0, 64}) {
        long lda = 1024 + pa, ldw = 1024 + pw;
        if (pa != 0 && pw == 0 && pa != 64) continue;
        float t1 = time_us([&] { hipLaunchKernelGGL(loads_kernel, dim3(256), dim3(256), 0, s, A, W, out, lda, ldw, 16); }, 1000, s);
        float t2 = time_us([&] { hipLaunchKernelGGL(loads_kernel_wide, dim3(256), dim3(256), 0, s, A, W, out, lda, ldw, 16); }, 1000, s);
        printf("lda=%5ld ldw=%5ld : 16rows x 64B mapping %6.2f us   4rows x 256B mapping %6.2f us\n", lda, ldw, t1, t2);
    }
    return 0;
}
