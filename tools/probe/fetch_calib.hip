// Known-bytes calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of the sweep kernel
// (MI355X_MICROARCH.md section HBM: FETCH_SIZE reports exactly half of a wide coalesced stream on gfx950,
// other widths are uncalibrated).  Buffer of 2 GiB (>> 256 MiB Infinity Cache, >> 32 MiB of L2), every kernel touches
// each byte / sector at most once, so the bytes that must come from HBM are known:
//   stream16   16 B per lane, coalesced               bytes = N
//   stream4     4 B per lane, coalesced               bytes = N
//   rows64     every lane reads its own 64-B row with four 16-B loads (the feeder's id loads)     bytes = N
//   gather1    one BYTE per lane, every lane in a different 128-B line, each line touched once    useful bytes = lanes,
//              HBM bytes = lanes x (sector the memory system really moves: this run tells)
//   gather1x64 one byte per lane, 64-B apart (two lanes per 128-B line)
// Run each under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE for the store kernel); the program prints the byte
// counts to compare with.   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip && ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void stream16(const uint4* __restrict__ p, size_t n16, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void stream4(const unsigned* __restrict__ p, size_t n4, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
// lane l of block-iteration j reads row (j * lanes + perm(l)) of 64 bytes as four uint4: rows contiguous per lane
__global__ void rows64(const uint4* __restrict__ p, size_t rows, unsigned* out) {
    unsigned acc = 0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
        const uint4* q = p + r * 4;
        const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// one byte per lane at byte offset i * stride (+ a scrambled offset inside the granule): every granule touched once
__global__ void gather1(const unsigned char* __restrict__ p, size_t n, size_t stride, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        // scatter the lanes of a wave over the buffer (neighbouring lanes are far apart, like label gathers)
        const size_t j = (i * 0x9E3779B97F4A7C15ull) % n;
        acc += p[j * stride + (j & (stride - 1) & 63)];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void store16(uint4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)2 << 30;
    const char* which = argc > 1 ? argv[1] : "all";
    unsigned char* buf;
    unsigned* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const dim3 grid(256 * 8), block(256);
    auto is = [&](const char* name) { return !strcmp(which, "all") || !strcmp(which, name); };
    if (is("stream16")) {
        hipLaunchKernelGGL(stream16, grid, block, 0, 0, (const uint4*)buf, bytes / 16, out);
        printf("stream16: %zu bytes read, 16 B per lane coalesced\n", bytes);
    }
    if (is("stream4")) {
        hipLaunchKernelGGL(stream4, grid, block, 0, 0, (const unsigned*)buf, bytes / 4, out);
        printf("stream4: %zu bytes read, 4 B per lane coalesced\n", bytes);
    }
    if (is("rows64")) {
        hipLaunchKernelGGL(rows64, grid, block, 0, 0, (const uint4*)buf, bytes / 64, out);
        printf("rows64: %zu bytes read, one 64-B row per lane (4 x 16 B)\n", bytes);
    }
    if (is("gather1")) {
        const size_t n = bytes / 128;
        hipLaunchKernelGGL(gather1, grid, block, 0, 0, buf, n, (size_t)128, out);
        printf("gather1: %zu single-byte loads, one per 128-B line (useful bytes %zu; 64-B sectors %zu B, 128-B lines %zu B)\n", n, n,
               n * 64, n * 128);
    }
    if (is("gather1x64")) {
        const size_t n = bytes / 64;
        hipLaunchKernelGGL(gather1, grid, block, 0, 0, buf, n, (size_t)64, out);
        printf("gather1x64: %zu single-byte loads, one per 64-B sector (useful bytes %zu; sectors %zu B)\n", n, n, n * 64);
    }
    if (is("store16")) {
        hipLaunchKernelGGL(store16, grid, block, 0, 0, (uint4*)buf, bytes / 16);
        printf("store16: %zu bytes written, 16 B per lane coalesced\n", bytes);
    }
    hipDeviceSynchronize();
    hipFree(buf);
    hipFree(out);
    return 0;
}
