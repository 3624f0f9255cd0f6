// Probe: cost of handing a result from one wave of a workgroup to another through LDS (write payload + sequence word,
// the other wave polls the sequence word), as a ping-pong between the two waves of every workgroup at the sweep
// kernel's launch shape (1024 workgroups of 128 threads, 4 per CU).  Prints s_memtime ticks per ROUND TRIP (two hand-offs), with
// the poll loop written the way a stepping wave would write it (ds_read -> readfirstlane -> branch).  On this chip s_memtime
// advances at about the shader clock (the sweep kernel's stage stamps add up to its measured time per step at ~2.3 GHz), so
// a tick is a cycle.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_handoff tools/probe/wave_handoff.hip && /tmp/wave_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(128) void pingpong(unsigned long long* out, uint32_t* simd, int rounds, int payload_words) {
    __shared__ uint32_t box[2][16];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    if (threadIdx.x < 32) ((uint32_t*)box)[threadIdx.x] = 0;
    __syncthreads();
    uint32_t hw_id;
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    if (lane == 0) simd[blockIdx.x * 2 + wave] = (hw_id >> 4) & 3u;
    unsigned long long t0, t1;
    __asm__ volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    uint32_t acc = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (wave == 0) {
            if (lane < (uint32_t)payload_words) box[0][1 + lane] = acc + lane;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_store(&box[0][0], (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (LDS ops of one wave complete in order)
            while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&box[1][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != (uint32_t)r) {
            }
            acc += __hip_atomic_load(&box[1][1 + (lane & 3u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&box[0][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != (uint32_t)r) {
            }
            acc += __hip_atomic_load(&box[0][1 + (lane & 3u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane < (uint32_t)payload_words) box[1][1 + lane] = acc + lane;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_store(&box[1][0], (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __asm__ volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) out[blockIdx.x * 2 + wave] = (t1 - t0) + (acc == 0xdeadbeefu ? 1 : 0);
}

int main() {
    const int wgs = 1024, rounds = 20000;
    unsigned long long* d_out;
    uint32_t* d_simd;
    hipMalloc(&d_out, wgs * 2 * sizeof(unsigned long long));
    hipMalloc(&d_simd, wgs * 2 * sizeof(uint32_t));
    for (int grid : {1, 256, 1024}) {
        for (int payload : {0, 8}) {
            hipLaunchKernelGGL(pingpong, dim3(grid), dim3(128), 0, 0, d_out, d_simd, rounds, payload);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(grid * 2);
            std::vector<uint32_t> s(grid * 2);
            hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(s.data(), d_simd, s.size() * 4, hipMemcpyDeviceToHost);
            std::vector<double> per(grid);
            int same_simd = 0;
            for (int i = 0; i < grid; ++i) {
                per[i] = (double)h[2 * i] / rounds;
                same_simd += s[2 * i] == s[2 * i + 1];
            }
            std::sort(per.begin(), per.end());
            printf("grid %4d payload %d words: round trip (two hand-offs) median %.1f cycles, min %.1f, max %.1f; workgroups with both"
                   " waves on one SIMD: %d\n",
                   grid, payload, per[grid / 2], per[0], per[grid - 1], same_simd);
        }
    }
    return 0;
}
