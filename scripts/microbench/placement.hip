// Where do the wavefronts of co-resident workgroups land?  512 workgroups of 4 wavefronts with half a CU's LDS each (the <4,2>
// geometry of the latency engine): per wavefront HW_ID (SIMD, CU, SE), XCC_ID and LDS_ALLOC.  build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
extern __shared__ double pool[];
__global__ __launch_bounds__(256, 2) void k(unsigned *out, int spin)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), lds = __builtin_amdgcn_s_getreg((31 << 11) | 6), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    pool[threadIdx.x] = 1.0;
    double a = pool[threadIdx.x];
    for (int i = 0; i < spin; i++) a = a * 1.0000001 + 1e-9;   // keep the first wave of workgroups resident while the second arrives
    if ((threadIdx.x & 63) == 0) {
        unsigned *o = out + (blockIdx.x * 4 + threadIdx.x / 64) * 4;
        o[0] = hw; o[1] = lds; o[2] = xcc; o[3] = a > 0;
    }
}
int main()
{
    const int B = 512;
    unsigned *d;
    hipMalloc(&d, B * 16 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 73232);
    k<<<B, 256, 73232>>>(d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;   // (xcc, se, cu) -> blocks
    int same_simd0 = 0, perm_identity = 0;
    for (int b = 0; b < B; b++) {
        unsigned simd[4];
        for (int w = 0; w < 4; w++) simd[w] = (h[(b * 4 + w) * 4] >> 4) & 3;
        const unsigned hw = h[b * 16], key = ((h[b * 16 + 2] & 15) << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15);
        cu[key].push_back(b);
        perm_identity += simd[0] == 0 && simd[1] == 1 && simd[2] == 2 && simd[3] == 3;
        if (b < 8 || (b >= 256 && b < 264))
            printf("block %3d: simd of waves %u %u %u %u  hw %08x lds_alloc %08x xcc %u key %06x\n", b, simd[0], simd[1], simd[2], simd[3], hw, h[b * 16 + 1], h[b * 16 + 2] & 15, key);
    }
    int pairs = 0;
    for (auto &kv : cu) {
        if (kv.second.size() == 2) {
            pairs++;
            const unsigned s0 = (h[kv.second[0] * 16] >> 4) & 3, s1 = (h[kv.second[1] * 16] >> 4) & 3;
            same_simd0 += s0 == s1;
            if (pairs <= 6) printf("CU %06x: blocks %d %d, wave-0 SIMDs %u %u, lds base %u %u\n", kv.first, kv.second[0], kv.second[1], s0, s1, h[kv.second[0] * 16 + 1] & 255, h[kv.second[1] * 16 + 1] & 255);
        }
    }
    printf("%zu CUs, %d with two workgroups, %d of them with both wave 0 on the same SIMD; %d of %d workgroups have wave w on SIMD w\n", cu.size(), pairs, same_simd0, perm_identity, B);
    return 0;
}
