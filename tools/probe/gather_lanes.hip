// Cost of an 8-byte gather as a function of the number of lanes that take part in it (execution mask) and of the
// number of distinct addresses, for a lone wave and for the sweep kernel's occupancy (four stepping waves per CU).
// Each lane follows its own dependent chain through a 1 MiB table (L2 resident); GROUP independent chains per lane
// give GROUP gathers in flight, like the table gathers of one step.
//   hipcc --offload-arch=gfx950 -O3 -o gather_lanes gather_lanes.hip && ./gather_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int GROUP>
__global__ void chase(const uint2* __restrict__ tab, unsigned mask, unsigned* out, int iters, int active, int same) {
    const unsigned lane = threadIdx.x;
    unsigned a[GROUP];
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
        // lanes >= `same` all start on one index and therefore stay on one address
        const unsigned id = (lane < (unsigned)same) ? (blockIdx.x * 64 + lane) * 4 + g + 1 : (unsigned)g + 1;
        a[g] = (id * 2654435761u) & mask;
    }
    if (lane < (unsigned)active) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint2 v[GROUP];
#pragma unroll
                for (int g = 0; g < GROUP; ++g) v[g] = tab[a[g]];
#pragma unroll
                for (int g = 0; g < GROUP; ++g) a[g] = v[g].x & mask;
            }
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int g = 0; g < GROUP; ++g) s += a[g];
    out[blockIdx.x * 64 + lane] = s;
}

template <int GROUP>
static void run(const uint2* tab, unsigned mask, unsigned* out, int blocks, int active, int same) {
    const int it = 500;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chase<GROUP>, dim3(blocks), dim3(64), 0, 0, tab, mask, out, it, active, same);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chase<GROUP>, dim3(blocks), dim3(64), 0, 0, tab, mask, out, it, active, same);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("group %d  blocks %4d  active lanes %2d  distinct addresses %2d : %7.1f ns per dependent group\n", GROUP, blocks,
           active, same < active ? same + 1 : active, ms * 1e6 / (it * 8.0));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    const unsigned n = 1u << 17;  // 1 MiB of uint2
    std::vector<uint2> t(n);
    unsigned long long s = 12345;
    for (unsigned i = 0; i < n; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        t[i].x = (unsigned)(s >> 33);
        t[i].y = i;
    }
    uint2* tab;
    unsigned* out;
    hipMalloc(&tab, n * sizeof(uint2));
    hipMalloc(&out, 1024 * 64 * 4);
    hipMemcpy(tab, t.data(), n * sizeof(uint2), hipMemcpyHostToDevice);
    for (int blocks : {1, 1024}) {
        for (int active : {64, 32, 16, 8, 1}) run<1>(tab, n - 1, out, blocks, active, 64);
        run<1>(tab, n - 1, out, blocks, 64, 8);  // 64 lanes active, 56 of them on one address
        for (int active : {64, 32, 8}) run<4>(tab, n - 1, out, blocks, active, 64);
        run<4>(tab, n - 1, out, blocks, 64, 8);
    }
    return 0;
}
