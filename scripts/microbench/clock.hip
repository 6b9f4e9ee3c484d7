// What is the shader clock under a latency-bound launch, and what does a dependent fp64 instruction cost?  One wavefront per CU
// runs a loop of s_nop 15 and, in a second kernel, a chain of dependent v_fma_f64 (measured: 2.62 ns = 6.3 cycles per link -- the
// floor of every single-wavefront recursion in the engines: 5.8 cycles per instruction in the factorisation sweep); wall time from s_memrealtime (100 MHz).  build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void nops(long long *out, int iters)
{
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        asm volatile("s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 15" ::: "memory");
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
__global__ void fmas(long long *out, double *sink, int iters, double a, double b)
{
    double x = threadIdx.x;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) x = __builtin_fma(x, a, b);
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main()
{
    long long *d, h[256];
    double *sink;
    hipMalloc(&d, 256 * 8);
    hipMalloc(&sink, 256 * 256 * 8);
    for (int rep = 0; rep < 3; rep++) {
        for (int blocks : {1, 256}) {
            const int it = 2000000;
            nops<<<blocks, 64>>>(d, it);
            hipDeviceSynchronize();
            hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
            // (s_nop 15 holds the wavefront for 16 QUAD cycles + its own issue: 68 cycles -- 28.5 ns here = 2.39 GHz, the clock
            // GRBM_GUI_ACTIVE / 8 XCDs / kernel time gives for the real kernels as well)
            printf("s_nop 15, %3d wavefronts: %.2f ns each\n", blocks, h[0] * 10.0 / ((double)it * 8));
            const int itf = 1000000;
            fmas<<<blocks, 64>>>(d, sink, itf, 0.999999, 1e-9);
            hipDeviceSynchronize();
            hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
            printf("dependent v_fma_f64 chain, %3d wavefronts: %.2f ns per fma\n", blocks, h[0] * 10.0 / ((double)itf * 16));
        }
    }
    return 0;
}
