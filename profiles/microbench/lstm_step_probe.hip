// Microbenchmark (measurement tooling, not product): where do the ~11 us of one LSTM forward step go on MI355X?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vae_lagging_encoder_amd/csrc profiles/microbench/lstm_step_probe.hip -o /tmp/lstm_probe
#include "lv_lstm.hip"
#include <vector>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 1000) p[0] = 1.f; }

// loads only: same addresses as the fwd kernel's matmul, reduced into one value
template <int SAMEW>
__global__ __launch_bounds__(256) void loads_only_kernel(const float* __restrict__ h, const float* __restrict__ whh, float* out, int H, int B) {
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int i = l & 15, kq = l >> 4;
    const int u0 = (SAMEW ? 0 : (int)blockIdx.x * 4);
    const int n = l & 15;
    const float* wrow = whh + ((long)(n >> 2) * H + u0 + (n & 3)) * H;
    const float* a0 = h + (long)i * H;
    const float* a1 = h + (long)(16 + i) * H;
    const int kbeg = w * (H / 4);
    float4 acc = make_float4(0, 0, 0, 0);
    float4 v[48];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int kk = kbeg + 16 * it + 4 * kq;
        v[3 * it] = *reinterpret_cast<const float4*>(wrow + kk);
        v[3 * it + 1] = *reinterpret_cast<const float4*>(a0 + kk);
        v[3 * it + 2] = *reinterpret_cast<const float4*>(a1 + kk);
    }
#pragma unroll
    for (int q = 0; q < 48; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x * 256 + tid] = acc.x;
}

__global__ __launch_bounds__(256) void mfma_only_kernel(float* out, float seed) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = seed + threadIdx.x, y = seed * 0.5f;
#pragma unroll
    for (int it = 0; it < 64; ++it) {
        a0 = lv_mfma_16x16x4(x, y, a0);
        a1 = lv_mfma_16x16x4(y, x, a1);
    }
    if (a0[0] + a1[1] == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = a0[0];
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
    const int H = 1024, B = 32, T = 64;
    hipStream_t s; CK(hipStreamCreate(&s));
    float *gx, *whh, *hs, *cs, *gates, *out, *dG, *dGsum, *dhext, *ws;
    CK(hipMalloc(&gx, (size_t)T * B * 4 * H * 4)); CK(hipMalloc(&whh, (size_t)4 * H * H * 4));
    CK(hipMalloc(&hs, (size_t)(T + 1) * B * H * 4)); CK(hipMalloc(&cs, (size_t)(T + 1) * B * H * 4));
    CK(hipMalloc(&gates, (size_t)T * B * 4 * H * 4)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMalloc(&dG, (size_t)T * B * 4 * H * 4)); CK(hipMalloc(&dGsum, (size_t)B * 4 * H * 4));
    CK(hipMalloc(&dhext, (size_t)T * B * H * 4)); CK(hipMalloc(&ws, (size_t)lv_lstm_ws_floats(B, H) * 4));
    CK(hipMemset(gx, 0, (size_t)T * B * 4 * H * 4)); CK(hipMemset(whh, 0, (size_t)4 * H * H * 4));
    CK(hipMemset(hs, 0, (size_t)(T + 1) * B * H * 4)); CK(hipMemset(cs, 0, (size_t)(T + 1) * B * H * 4)); CK(hipMemset(dhext, 0, (size_t)T * B * H * 4));
    CK(hipMemset(gates, 0, (size_t)T * B * 4 * H * 4));

    printf("empty kernel 256x256, back-to-back          : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, out); }, 2000, s));
    printf("mfma only (128 x 16x16x4 per wave)           : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL(mfma_only_kernel, dim3(256), dim3(256), 0, s, out, 1.0f); }, 2000, s));
    printf("loads only, 16 rows x 64 B per wave load     : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL(loads_only_kernel<0>, dim3(256), dim3(256), 0, s, hs, whh, out, H, B); }, 2000, s));
    printf("lv_lstm_fwd_f32 (packed operands), T=%d       : %7.2f us/step\n", T, time_us([&] { lv_lstm_fwd_f32(gx, whh, hs, cs, gates, nullptr, 1.f, nullptr, ws, T, B, H, s); }, 20, s) / T);
    printf("lv_lstm_bwd_f32 (packed operands), T=%d       : %7.2f us/step\n", T, time_us([&] { lv_lstm_bwd_f32(dhext, nullptr, nullptr, 1.f, whh, gates, hs, cs, dG, dGsum, ws, nullptr, nullptr, 0, T, B, H, s); }, 20, s) / T);
    const Geo g = geo(B, H);
    LstmFwdP p{gx, ws, hs, cs, gates, ws + g.wp, nullptr, 1.f, nullptr, T, B, H, g.Kq, g.MBTp, 0};
    printf("fwd step kernel alone, back-to-back same t    : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL((lstm_step_fwd_kernel<2>), dim3(256, 1), dim3(256), 0, s, p, 3); }, 2000, s));
    printf("  fwd ablation: no matmul                      : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 1>), dim3(256, 1), dim3(256), 0, s, p, 3); }, 2000, s));
    printf("  fwd ablation: no gate math / gate stores     : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 2>), dim3(256, 1), dim3(256), 0, s, p, 3); }, 2000, s));
    printf("  fwd ablation: no epilogue operand loads      : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 4>), dim3(256, 1), dim3(256), 0, s, p, 3); }, 2000, s));
    printf("  fwd ablation: matmul only (2+4)              : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 6>), dim3(256, 1), dim3(256), 0, s, p, 3); }, 2000, s));
    printf("  fwd ablation: nothing (1+2+4)                : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 7>), dim3(256, 1), dim3(256), 0, s, p, 3); }, 2000, s));
    lv_lstm_fwd_bf16(gx, whh, hs, cs, gates, nullptr, 1.f, nullptr, ws, 1, B, H, s);   // packs bf16 operands into ws
    CK(hipStreamSynchronize(s));
    const Geo g16 = geo(B, H, true);
    LstmFwdP p16{gx, ws, hs, cs, gates, ws + g16.wp, nullptr, 1.f, nullptr, T, B, H, g16.Kq, g16.MBTp, 0};
    printf("fwd step kernel, bf16 recurrent operands      : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL((lstm_step_fwd_kernel<2, 0, true>), dim3(256, 1), dim3(256), 0, s, p16, 3); }, 2000, s));
    printf("lv_lstm_fwd_bf16, T=%d                         : %7.2f us/step\n", T, time_us([&] { lv_lstm_fwd_bf16(gx, whh, hs, cs, gates, nullptr, 1.f, nullptr, ws, T, B, H, s); }, 20, s) / T);
    printf("lv_lstm_bwd_bf16, T=%d                         : %7.2f us/step\n", T, time_us([&] { lv_lstm_bwd_bf16(dhext, nullptr, nullptr, 1.f, whh, gates, hs, cs, dG, dGsum, ws, nullptr, nullptr, 0, T, B, H, s); }, 20, s) / T);
    // the real dependent chain (t = 0..T-1, each step reads what the previous one wrote), with the ablation switches
    auto chain = [&](auto kern) { for (int t = 0; t < T; ++t) hipLaunchKernelGGL(kern, dim3(256, 1), dim3(256), 0, s, p16, t); };
    printf("bf16 fwd chain, T=%d (real dependencies)       : %7.2f us/step\n", T, time_us([&] { chain(lstm_step_fwd_kernel<2, 0, true>); }, 20, s) / T);
    printf("  chain ablation: no gate math / gate stores   : %7.2f us/step\n", time_us([&] { chain(lstm_step_fwd_kernel<2, 2, true>); }, 20, s) / T);
    printf("  chain ablation: no epilogue operand loads    : %7.2f us/step\n", time_us([&] { chain(lstm_step_fwd_kernel<2, 4, true>); }, 20, s) / T);
    printf("  chain ablation: matmul only (2+4)            : %7.2f us/step\n", time_us([&] { chain(lstm_step_fwd_kernel<2, 6, true>); }, 20, s) / T);
    printf("  chain ablation: no matmul                    : %7.2f us/step\n", time_us([&] { chain(lstm_step_fwd_kernel<2, 1, true>); }, 20, s) / T);
    lv_lstm_fwd_f32(gx, whh, hs, cs, gates, nullptr, 1.f, nullptr, ws, 1, B, H, s);
    CK(hipStreamSynchronize(s));
    LstmBwdP q{dhext, nullptr, nullptr, 1.f, ws, gates, cs, dG, dGsum, ws + g.wpT, ws + g.wpT + g.dGp, ws + g.wpT + g.dGp + g.part, T, B, H, g.KS, g.Kq4, g.MBTp};
    printf("bwd matmul kernel alone, back-to-back         : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL((lstm_step_bwd_mm_kernel<2>), dim3(64 * g.KS, 1), dim3(256), 0, s, q, 3); }, 2000, s));
    printf("bwd elementwise kernel alone, back-to-back    : %7.2f us/launch\n", time_us([&] { hipLaunchKernelGGL((lstm_step_bwd_elem_kernel<4>), dim3(128), dim3(256), 0, s, q, 3); }, 2000, s));
    return 0;
}
