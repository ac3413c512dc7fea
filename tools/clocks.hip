// what do the two clocks cost a wave?  s_memrealtime (100 MHz, what the resident grids time their requests and their idle time-out with)
// and s_memtime (shader clock), each read 1000 times back to back by one wave, timed with the other.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint64_t *out)
{
    uint64_t acc = 0;
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 1000; ++i) acc += __builtin_amdgcn_s_memrealtime();
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 1000; ++i) acc += __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    // an LDS round trip and a barrier, for scale
    __shared__ uint32_t lds[256];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const uint64_t c2 = __builtin_amdgcn_s_memtime();
    uint32_t x = threadIdx.x;
    for (int i = 0; i < 1000; ++i) x = lds[x & 255];
    const uint64_t c3 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 1000; ++i) __syncthreads();
    const uint64_t c4 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = acc; out[3] = c3 - c2; out[4] = c4 - c3; out[5] = x; }
}
int main()
{
    uint64_t *d, h[6];
    hipMalloc(&d, sizeof(h));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    }
    printf("s_memrealtime: %.1f shader cycles per read | s_memtime: %.1f ns per read | dependent LDS read: %.1f cycles | barrier of 4 waves: %.1f cycles\n",
           h[0] / 1000.0, h[1] * 10.0 / 1000.0, h[3] / 1000.0, h[4] / 1000.0);
    return 0;
}
