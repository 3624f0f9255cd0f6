// Probe: layout of global_load_lds_ubyte (LDS-DMA of single bytes) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char* g, unsigned* out) {
    __shared__ unsigned lds[128];
    lds[threadIdx.x] = 0xAAAAAAAAu;
    lds[64 + threadIdx.x] = 0xBBBBBBBBu;
    __syncthreads();
    __builtin_amdgcn_global_load_lds(g + (63 - threadIdx.x) * 3, lds, 1, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
    out[64 + threadIdx.x] = lds[64 + threadIdx.x];
}
int main() {
    std::vector<unsigned char> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (unsigned char)(i + 1);
    unsigned char* d; unsigned* o;
    hipMalloc(&d, 256); hipMalloc(&o, 512);
    hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    std::vector<unsigned> r(128);
    hipMemcpy(r.data(), o, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("lds[%d]=%08x ", i, r[i]);
    printf("\n"); for (int i = 60; i < 68; ++i) printf("lds[%d]=%08x ", i, r[i]);
    printf("\n");
    return 0;
}
