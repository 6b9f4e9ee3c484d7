// Device unit test of Engine::copy_rect (diagnostic): HBM -> LDS -> HBM round trip for the rectangle
// shapes the passes use, with 1 and 4 wavefronts per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../robotic_mpc_amd/csrc/mpc_core.h"
using namespace mpcb;
__shared__ __attribute__((aligned(16))) Smem g_sm;
extern __shared__ __attribute__((aligned(16))) double g_pool[];
template <int NWV>
struct DevExec {
    static constexpr int NT = WAVE * NWV;
    __device__ __forceinline__ static int lane_id() { return (int)threadIdx.x; }
    __device__ __forceinline__ Smem &smem() const { return g_sm; }
    __device__ __forceinline__ double *pool() const { return g_pool; }
    template <class T> struct PerLane { T v; __device__ __forceinline__ T &at(int) { return v; } };
    template <class F> __device__ __forceinline__ void par(F &&f) { f(lane_id()); if (NWV == 1) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); }
    template <class F> __device__ __forceinline__ void seq(F &&f) { if (NWV == 1 || threadIdx.x < WAVE) { f(lane_id()); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } }
    __device__ __forceinline__ void join() { if (NWV > 1) __syncthreads(); }
    __device__ __forceinline__ double reduce_sum(const double *r) { return r[0]; }
    __device__ __forceinline__ double reduce_max(const double *r) { return r[0]; }
    __device__ __forceinline__ double reduce_min(const double *r) { return r[0]; }
    __device__ __forceinline__ double clock() { return 0.0; }
};
template <int NWV, int W, int C0, int LDG, int LDL>
__global__ void k_copy(const double *src, double *dst, int k_lo, int k_hi)
{
    DevExec<NWV> ex;
    Problem pb{};
    Ctx c{&pb, Ws{}, 16384, 10};
    Engine<DevExec<NWV>> eng(ex, c);
    eng.template copy_rect<W, C0, LDG, LDL, true>(g_pool, const_cast<double *>(src), k_lo, k_hi);
    eng.template copy_rect<W, C0, LDG, LDL, false>(g_pool, dst, k_lo, k_hi);
}
template <int NWV, int W, int C0, int LDG, int LDL>
int run(int rows_total, int k_lo, int k_hi)
{
    std::vector<double> h((size_t)rows_total * LDG), out((size_t)rows_total * LDG, -1.0);
    for (size_t i = 0; i < h.size(); i++) h[i] = (double)i + 0.25;
    double *s, *d;
    hipMalloc(&s, h.size() * 8); hipMalloc(&d, h.size() * 8);
    hipMemcpy(s, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d, out.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k_copy<NWV, W, C0, LDG, LDL>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL((k_copy<NWV, W, C0, LDG, LDL>), dim3(1), dim3(64 * NWV), 131072, 0, s, d, k_lo, k_hi);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; }
    hipMemcpy(out.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < rows_total; r++)
        for (int cc = 0; cc < LDG; cc++) {
            const bool in = r >= k_lo && r <= k_hi && cc >= C0 && cc < C0 + W;
            const double want = in ? h[(size_t)r * LDG + cc] : -1.0;
            if (out[(size_t)r * LDG + cc] != want) bad++;
        }
    printf("NWV=%d W=%3d C0=%3d LDG=%3d LDL=%3d rows %d..%d: %s (%d bad)\n", NWV, W, C0, LDG, LDL, k_lo, k_hi, bad ? "FAIL" : "ok", bad);
    hipFree(s); hipFree(d);
    return bad != 0;
}
int main()
{
    int f = 0;
    f += run<1, 272, 0, 272, 272>(60, 3, 48);   f += run<4, 272, 0, 272, 272>(60, 3, 48);
    f += run<1, 78, 24, 112, 78>(60, 0, 45);    f += run<4, 78, 24, 112, 78>(60, 0, 45);
    f += run<1, 12, 90, 112, 12>(60, 5, 39);    f += run<4, 12, 90, 112, 12>(60, 5, 39);
    f += run<1, 48, 48, 96, 48>(60, 5, 39);     f += run<4, 48, 48, 96, 48>(60, 5, 39);
    f += run<1, 18, 0, 96, 96>(60, 1, 21);      f += run<4, 18, 0, 96, 96>(60, 1, 21);
    f += run<1, 10, 0, 112, 60>(60, 0, 20);     f += run<4, 10, 0, 112, 60>(60, 0, 20);
    f += run<1, 66, 0, 144, 66>(120, 0, 100);   f += run<4, 66, 0, 144, 66>(120, 0, 100);
    f += run<1, 96, 0, 96, 96>(120, 0, 101);    f += run<4, 96, 0, 96, 96>(120, 0, 101);
    f += run<4, 144, 0, 144, 144>(120, 0, 30);  f += run<4, 42, 60, 112, 42>(120, 7, 7);
    printf(f ? "SOME FAILED\n" : "ALL OK\n");
    return f;
}
