// Probe (measurement tooling): lane -> element map of ds_read_b64_tr_b16 on gfx950.
// LDS holds u16 value i at element i; each lane passes an address; we dump the 4 returned u16 per lane for several address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int pattern) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    const int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int elem;                                     // element index this lane points at
    if (pattern == 0) elem = 0;                   // uniform
    else if (pattern == 1) elem = 4 * l;          // lane-linear 8 bytes
    else if (pattern == 2) elem = 64 * (l & 15) + 4 * (l >> 4);      // row (l&15) of a [16][64] matrix, 4-element group l>>4
    else elem = 16 * (l & 15) + 256 * (l >> 4);   // row pitch 16 elements
    auto p = (__attribute__((address_space(3))) s4*)(lds + elem);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int pat = 0; pat < 4; ++pat) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pat);
        uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d%s", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3], (l % 4 == 3) ? "\n" : " |");
    }
    return 0;
}
