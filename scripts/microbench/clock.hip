// What is the shader clock under a latency-bound launch?  One wavefront per CU runs a loop of s_nop 15 (16 cycles each) and, in a
// second kernel, a chain of dependent v_fma_f64; wall time from s_memrealtime (100 MHz).  build: hipcc --offload-arch=gfx950 -O2
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
            const double sec = h[0] * 1e-8, cyc = (double)it * 8 * 16;
            printf("s_nop loop, %3d wavefronts: %.3f s for %.3g nop cycles (+ loop overhead) -> >= %.2f GHz\n", blocks, sec, cyc, cyc / sec * 1e-9);
            const int itf = 1000000;
            fmas<<<blocks, 64>>>(d, sink, itf, 0.999999, 1e-9);
            hipDeviceSynchronize();
            hipMemcpy(h, d, blocks * 8, hipMemcpyDeviceToHost);
            printf("dependent v_fma_f64 chain, %3d wavefronts: %.2f ns per fma\n", blocks, h[0] * 10.0 / ((double)itf * 16));
        }
    }
    return 0;
}
