// Where do the two waves of 128-thread workgroups (45 KB of LDS each: three per CU) land?  Prints, per CU, the SIMD of
// wave 0 and of wave 1 of every resident workgroup.   hipcc --offload-arch=gfx950 -O2 -o wave_place wave_place.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <string>

__global__ void __launch_bounds__(128) place_kernel(unsigned *out, int hold)
{
    extern __shared__ char smem[];
    smem[threadIdx.x] = 0;
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    for (int i = 0; i < hold; ++i) __builtin_amdgcn_s_sleep(64);
}

int main()
{
    const int G = 766;
    unsigned *d;
    hipMalloc(&d, G * 4 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 46080);
    hipLaunchKernelGGL(place_kernel, dim3(G), dim3(128), 46080, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(G * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::string> cu;
    int same = 0;
    std::map<std::string, int> pat;
    for (int b = 0; b < G; ++b) {
        const unsigned hw0 = h[b * 4], x0 = h[b * 4 + 1], hw1 = h[b * 4 + 2];
        const unsigned key = ((x0 & 0xf) << 8) | ((hw0 >> 8) & 0xff);
        char t[32];
        snprintf(t, sizeof t, "%u/%u ", (hw0 >> 4) & 3, (hw1 >> 4) & 3);
        cu[key] += t;
        if (((hw0 >> 8) & 0xff) != ((hw1 >> 8) & 0xff)) ++same;
    }
    for (auto &kv : cu) pat[kv.second]++;
    printf("CUs in use: %zu; workgroups whose waves sit on different CUs: %d\n", cu.size(), same);
    for (auto &kv : pat) printf("  %4d CUs: walker/prefetcher SIMD of their workgroups: %s\n", kv.second, kv.first.c_str());
    return 0;
}
