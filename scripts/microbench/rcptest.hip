// accuracy of v_rcp_f64 with 0, 1, 2 Newton steps (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *x, double *o0, double *o1, double *o2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i];
    double r = __builtin_amdgcn_rcp(a);
    o0[i] = r;
    r = fma(fma(-a, r, 1.0), r, r);
    o1[i] = r;
    r = fma(fma(-a, r, 1.0), r, r);
    o2[i] = r;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), r0(n), r1(n), r2(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-30, 30);
    for (auto &v : x) v = std::ldexp(1.0 + (g() >> 12) * 0x1p-52, (int)u(g));
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        long double t = 1.0L / (long double)x[i];
        e0 = fmax(e0, (double)fabsl((r0[i] - t) / t));
        e1 = fmax(e1, (double)fabsl((r1[i] - t) / t));
        e2 = fmax(e2, (double)fabsl((r2[i] - t) / t));
    }
    printf("max rel err: rcp %.3e  +1 Newton %.3e  +2 Newton %.3e  (eps %.3e)\n", e0, e1, e2, 0x1p-53);
    return 0;
}
