// Probe: where do the two waves of each 128-thread workgroup land (SIMD / wave slot / CU) when 1024 workgroups
// with ~36 KB of LDS each fill the chip (the sweep kernel's launch shape)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <string>
__global__ __launch_bounds__(128) void k(unsigned* out, unsigned long long* spin) {
    extern __shared__ unsigned lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + wave) * 2 + 0] = hw;
        out[(blockIdx.x * 2 + wave) * 2 + 1] = xcc;
    }
    // stay resident long enough for the whole grid to be placed
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000000ull) lds[threadIdx.x] += 1;  // ~20 ms at 100 MHz
    if (lds[threadIdx.x] == 12345) spin[0] = 1;
}
int main() {
    const int blocks = 1024;
    unsigned* d; unsigned long long* s;
    hipMalloc(&d, blocks * 4 * sizeof(unsigned)); hipMalloc(&s, 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(128), 36 * 1024, 0, d, s);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<std::string, std::string> per_cu;
    int same_simd = 0; int hist[4][2] = {{0}};
    for (int b = 0; b < blocks; ++b) {
        char key[64], val[64];
        unsigned hw0 = h[b * 4 + 0], x0 = h[b * 4 + 1] & 0xf, hw1 = h[b * 4 + 2];
        unsigned cu = (hw0 >> 8) & 0xf, sh = (hw0 >> 12) & 1, se = (hw0 >> 13) & 7;
        snprintf(key, sizeof key, "xcc%u se%u sh%u cu%02u", x0, se, sh, cu);
        snprintf(val, sizeof val, " [wg%d: s%u.w%u s%u.w%u]", b, (hw0 >> 4) & 3, hw0 & 0xf, (hw1 >> 4) & 3, hw1 & 0xf);
        per_cu[key] += val;
        if (((hw0 >> 4) & 3) == ((hw1 >> 4) & 3)) ++same_simd;
        hist[(hw0 >> 4) & 3][0]++; hist[(hw1 >> 4) & 3][1]++;
    }
    int shown = 0;
    for (auto& kv : per_cu) if (shown++ < 24) printf("%s:%s\n", kv.first.c_str(), kv.second.c_str());
    printf("CUs used: %zu; workgroups with both waves on one SIMD: %d\n", per_cu.size(), same_simd);
    for (int s2 = 0; s2 < 4; ++s2) printf("SIMD%d: wave0 count %d, wave1 count %d\n", s2, hist[s2][0], hist[s2][1]);
    return 0;
}
