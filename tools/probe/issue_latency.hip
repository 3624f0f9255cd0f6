// Probe: per-instruction cost seen by ONE wave on a SIMD (the situation of the sweep kernel's stepping
// wave): dependent chains and independent streams of the instruction kinds the step is made of.
// Prints ns per instruction (HIP events) and the same in shader cycles (s_memtime is a fixed 100 MHz
// counter on gfx9; the clock is derived from a v_add_f32 stream whose cost is 4 cycles per instruction).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_latency tools/probe/issue_latency.hip && /tmp/issue_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)

#define KERNEL_V(name, body64, perloop)                                              \
    __global__ void name(double* io, int iters) {                                    \
        double x = io[threadIdx.x], y = io[64 + threadIdx.x], z = io[128 + threadIdx.x], \
               w = io[192 + threadIdx.x];                                            \
        float f = (float)x, g = (float)y, h = (float)z, e = (float)w;                \
        int a = (int)x + threadIdx.x, b = (int)y, c = 3, d = 5;                      \
        for (int i = 0; i < iters; ++i) {                                            \
            asm volatile(body64                                                      \
                         : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "+v"(f), "+v"(g), "+v"(h), "+v"(e), "+v"(a), \
                           "+v"(b), "+v"(c), "+v"(d)                                 \
                         :                                                           \
                         : "s20", "s21", "s22", "s23", "vcc", "scc", "memory");      \
        }                                                                            \
        io[256 + threadIdx.x] = x + y + z + w + f + g + h + e + a + b + c + d;       \
    }                                                                                \
    static const int name##_n = perloop;

// operands: %0..%3 f64 x y z w; %4..%7 f32 f g h e; %8..%11 i32 a b c d
KERNEL_V(k_add_f32_dep, R64("v_add_f32 %4, %4, %5\n"), 64)
KERNEL_V(k_add_f32_ind, R16("v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7\n"), 64)
KERNEL_V(k_add_f64_dep, R64("v_add_f64 %0, %0, %1\n"), 64)
KERNEL_V(k_add_f64_ind, R16("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3\n"), 64)
KERNEL_V(k_mul_f64_dep, R64("v_mul_f64 %0, %0, %1\n"), 64)
KERNEL_V(k_fma_f64_dep, R64("v_fma_f64 %0, %0, %1, %2\n"), 64)
KERNEL_V(k_fma_f64_ind, R16("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"), 64)
KERNEL_V(k_add_u32_dep, R64("v_add_u32 %8, %8, %9\n"), 64)
KERNEL_V(k_dpp_add_u32, R64("s_nop 1\n v_add_u32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"), 64)
KERNEL_V(k_cndmask_dep, R64("v_cndmask_b32 %8, %8, %9, vcc\n"), 64)
KERNEL_V(k_readlane_rt, R64("v_readlane_b32 s20, %8, 5\n v_add_u32 %8, s20, %9\n"), 64)
KERNEL_V(k_readfirst_salu_rt, R64("v_readfirstlane_b32 s20, %8\n s_add_u32 s20, s20, 1\n v_mov_b32 %8, s20\n"), 64)
KERNEL_V(k_cmp_branchless, R64("v_cmp_lt_i32 vcc, %8, %9\n s_and_b64 s[20:21], vcc, exec\n v_cndmask_b32 %8, %8, %9, s[20:21]\n"), 64)
KERNEL_V(k_saveexec_pair, R64("s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %8, %8, %9\n s_or_b64 exec, exec, s[20:21]\n"), 64)
KERNEL_V(k_rsq_f64_dep, R64("v_rsq_f64 %0, %0\n"), 64)
KERNEL_V(k_rcp_f64_dep, R64("v_rcp_f64 %0, %0\n"), 64)
KERNEL_V(k_exp_f32_dep, R64("v_exp_f32 %4, %4\n"), 64)
KERNEL_V(k_exp_f32_ind, R16("v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"), 64)
KERNEL_V(k_cvt_f64_i32_rt, R64("v_cvt_f64_i32 %0, %8\n v_cvt_i32_f64 %8, %0\n"), 64)
KERNEL_V(k_cvt_f32_f64_rt, R64("v_cvt_f32_f64 %4, %0\n v_cvt_f64_f32 %0, %4\n"), 64)
KERNEL_V(k_rndne_f64_dep, R64("v_rndne_f64 %0, %0\n"), 64)
KERNEL_V(k_ldexp_f64_dep, R64("v_ldexp_f64 %0, %0, %10\n"), 64)
KERNEL_V(k_mul_lo_u32_dep, R64("v_mul_lo_u32 %8, %8, %9\n"), 64)
KERNEL_V(k_mul_u24_dep, R64("v_mul_u32_u24 %8, %8, %9\n"), 64)
KERNEL_V(k_mad_u64_u32_dep, R64("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n"), 64)
KERNEL_V(k_salu_dep, R64("s_add_u32 s20, s20, 3\n"), 64)
KERNEL_V(k_salu_mul_dep, R64("s_mul_i32 s20, s20, 3\n"), 64)
KERNEL_V(k_snop0, R64("s_nop 0\n"), 64)
KERNEL_V(k_permlane32_swap, R64("s_nop 1\n v_permlane32_swap_b32 %8, %9\n"), 64)
KERNEL_V(k_mix_valu_salu, R64("v_add_u32 %8, %8, %9\n s_add_u32 s20, s20, 3\n"), 128)
KERNEL_V(k_ballot_ff1, R64("v_cmp_lt_u32 vcc, %8, %9\n s_ff1_i32_b64 s20, vcc\n v_add_u32 %8, s20, %8\n"), 64)

KERNEL_V(k_cndmask_vcc_init, "v_cmp_lt_u32 vcc, %8, %9\n" R64("v_cndmask_b32 %8, %8, %9, vcc\n"), 64)
KERNEL_V(k_cndmask_e64_sgpr, "v_cmp_lt_u32 s[20:21], %8, %9\n" R64("v_cndmask_b32_e64 %8, %8, %9, s[20:21]\n"), 64)
KERNEL_V(k_cndmask_e64_ind, "v_cmp_lt_u32 s[20:21], %8, %9\n" R16("v_cndmask_b32_e64 %8, %8, %9, s[20:21]\n v_cndmask_b32_e64 %9, %9, %10, s[20:21]\n v_cndmask_b32_e64 %10, %10, %11, s[20:21]\n v_cndmask_b32_e64 %11, %11, %8, s[20:21]\n"), 64)
KERNEL_V(k_cmp_u32_vcc, R64("v_cmp_lt_u32 vcc, %8, %9\n"), 64)
KERNEL_V(k_cmp_f64_sgpr, R64("v_cmp_lt_f64 s[20:21], %0, %1\n"), 64)
KERNEL_V(k_addc_co, R64("v_addc_co_u32 %8, vcc, %8, %9, vcc\n"), 64)
KERNEL_V(k_readlane_ind, R64("v_readlane_b32 s20, %8, 5\n"), 64)
KERNEL_V(k_readlane_sidx, "s_mov_b32 s22, 7\n" R64("v_readlane_b32 s20, %8, s22\n"), 64)
KERNEL_V(k_valu_sgpr_operand, "s_mov_b32 s20, 7\n" R64("v_add_u32 %8, s20, %8\n"), 64)
KERNEL_V(k_mul_f64_sgpr, "s_mov_b32 s20, 7\n s_mov_b32 s21, 0x3ff00000\n" R64("v_mul_f64 %0, s[20:21], %0\n"), 64)
KERNEL_V(k_add_f64_literal, R64("v_add_f64 %0, %0, 0.5\n"), 64)
KERNEL_V(k_mov_v_s, "s_mov_b32 s20, 7\n" R64("v_mov_b32 %8, s20\n"), 64)
KERNEL_V(k_waitcnt_idle, R64("s_waitcnt vmcnt(0) lgkmcnt(0)\n"), 64)
KERNEL_V(k_branch_not_taken, "s_cmp_eq_u32 s20, s20\n" R64("s_cbranch_scc0 1f\n v_add_u32 %8, %8, %9\n 1:\n"), 64)
KERNEL_V(k_branch_taken, "s_cmp_eq_u32 s20, s20\n" R64("s_cbranch_scc1 1f\n v_add_u32 %8, %8, %9\n 1:\n v_add_u32 %9, %8, %9\n"), 64)
KERNEL_V(k_vcc_branch, R64("v_cmp_lt_u32 vcc, %8, %9\n s_cbranch_vccz 1f\n v_add_u32 %8, 1, %8\n 1:\n"), 64)
KERNEL_V(k_s_and_exec, R64("s_and_b64 s[20:21], s[20:21], exec\n"), 64)
KERNEL_V(k_dpp_mov_nonop, R64("v_mov_b32_dpp %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"), 64)
KERNEL_V(k_dpp_add_dep_nop0, R64("s_nop 0\n v_add_u32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"), 64)
KERNEL_V(k_fmac_f64_dep, R64("v_fmac_f64 %0, %1, %2\n"), 64)
KERNEL_V(k_cvt_f64_u32, R64("v_cvt_f64_u32 %0, %8\n"), 64)
KERNEL_V(k_exp_f32_then_valu, R64("v_exp_f32 %4, %4\n v_add_u32 %8, %8, %9\n"), 64)
KERNEL_V(k_rsq_then_3valu, R64("v_rsq_f64 %0, %0\n v_add_u32 %8, %8, %9\n v_add_u32 %10, %10, %9\n v_add_u32 %11, %11, %9\n"), 64)
// one butterfly level on a double (two v_mov_b32_dpp + v_add_f64), 16 dependent levels per loop
__global__ void k_dpp_f64_level(double* io, int iters) {
    double x = io[threadIdx.x];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xF, 0xF, false);
            const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xF, 0xF, false);
            x = x * 0.5 + __hiloint2double(hi, lo);
        }
    }
    io[256 + threadIdx.x] = x;
}
static const int k_dpp_f64_level_n = 16;
// LDS dependent read chain: the address of the next read is the value of the previous one
__global__ void k_lds_chase(double* io, int iters) {
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = ((i + 64) & 1023) * 4;
    __syncthreads();
    unsigned a = threadIdx.x * 4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) a = *(volatile unsigned*)((char*)lds + a);
    }
    io[256 + threadIdx.x] = a;
}
// LDS read whose address comes from an SGPR each time and whose result goes back to an SGPR (the step's pattern)
__global__ void k_lds_sgpr_rt(double* io, int iters) {
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = ((i + 64) & 1023) * 4;
    __syncthreads();
    unsigned a = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const unsigned v = *(volatile unsigned*)((char*)lds + a + threadIdx.x * 4);
            a = __builtin_amdgcn_readfirstlane(v) & 0xfff;
        }
    }
    io[256 + threadIdx.x] = a;
}
// global dependent gather chain over a table of `span` bytes (L2 / MALL / HBM latency as seen by one wave)
__global__ void k_global_chase(const unsigned* tab, double* io, int iters) {
    unsigned a = threadIdx.x * 16;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) a = tab[a / 4];
    }
    io[256 + threadIdx.x] = a;
}

static double time_ms(void (*launch)(int, hipStream_t), int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(blocks, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    launch(blocks, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

static double* g_io;
static unsigned* g_tab;
static int g_iters = 20000;

#define RUNV(name)                                                                                  \
    {                                                                                               \
        auto l = [](int blocks, hipStream_t s) { hipLaunchKernelGGL(name, dim3(blocks), dim3(64), 0, s, g_io, g_iters); }; \
        for (int blocks : {1, 256, 1024}) {                                                              \
            const double ms = time_ms(l, blocks);                                                   \
            const double ns = ms * 1e6 / ((double)g_iters * name##_n);                              \
            printf("%-22s blocks=%-5d %8.3f ns/instr-group  %7.2f cycles@%.2fGHz\n", #name, blocks, ns, ns * ghz[blocks > 1], ghz[blocks > 1]); \
        }                                                                                           \
    }

int main() {
    hipMalloc(&g_io, 4096 * sizeof(double));
    std::vector<double> h(4096, 1.0000001);
    hipMemcpy(g_io, h.data(), 4096 * sizeof(double), hipMemcpyHostToDevice);
    double ghz[2] = {2.4, 2.4};
    // clock calibration: independent v_add_f32 stream = 4 cycles per instruction for one wave
    for (int full = 0; full < 2; ++full) {
        auto l = [](int blocks, hipStream_t s) { hipLaunchKernelGGL(k_add_f32_ind, dim3(blocks), dim3(64), 0, s, g_io, g_iters); };
        const double ms = time_ms(l, full ? 1024 : 1);
        const double ns = ms * 1e6 / ((double)g_iters * 64);
        ghz[full] = 4.0 / ns;
        printf("calibration (%s): v_add_f32 independent %.3f ns -> %.3f GHz if 4 cycles each\n", full ? "1024 waves" : "1 wave", ns, ghz[full]);
    }
    RUNV(k_cndmask_vcc_init) RUNV(k_cndmask_e64_sgpr) RUNV(k_cndmask_e64_ind) RUNV(k_cmp_u32_vcc) RUNV(k_cmp_f64_sgpr)
    RUNV(k_addc_co) RUNV(k_readlane_ind) RUNV(k_readlane_sidx) RUNV(k_valu_sgpr_operand) RUNV(k_mul_f64_sgpr)
    RUNV(k_add_f64_literal) RUNV(k_mov_v_s) RUNV(k_waitcnt_idle) RUNV(k_branch_not_taken) RUNV(k_branch_taken)
    RUNV(k_vcc_branch) RUNV(k_s_and_exec) RUNV(k_dpp_mov_nonop) RUNV(k_dpp_add_dep_nop0) RUNV(k_fmac_f64_dep)
    RUNV(k_cvt_f64_u32) RUNV(k_exp_f32_then_valu) RUNV(k_rsq_then_3valu)
    RUNV(k_add_f32_dep) RUNV(k_add_f32_ind) RUNV(k_add_f64_dep) RUNV(k_add_f64_ind) RUNV(k_mul_f64_dep)
    RUNV(k_fma_f64_dep) RUNV(k_fma_f64_ind) RUNV(k_add_u32_dep) RUNV(k_dpp_add_u32) RUNV(k_dpp_f64_level)
    RUNV(k_cndmask_dep) RUNV(k_readlane_rt) RUNV(k_readfirst_salu_rt) RUNV(k_cmp_branchless) RUNV(k_saveexec_pair)
    RUNV(k_rsq_f64_dep) RUNV(k_rcp_f64_dep) RUNV(k_exp_f32_dep) RUNV(k_exp_f32_ind) RUNV(k_cvt_f64_i32_rt)
    RUNV(k_cvt_f32_f64_rt) RUNV(k_rndne_f64_dep) RUNV(k_ldexp_f64_dep) RUNV(k_mul_lo_u32_dep) RUNV(k_mul_u24_dep)
    RUNV(k_mad_u64_u32_dep) RUNV(k_salu_dep) RUNV(k_salu_mul_dep) RUNV(k_snop0) RUNV(k_permlane32_swap)
    RUNV(k_mix_valu_salu) RUNV(k_ballot_ff1)
    {
        static const int k_lds_chase_n = 64, k_lds_sgpr_rt_n = 64;
        RUNV(k_lds_chase) RUNV(k_lds_sgpr_rt)
    }
    // global chase over tables of growing span: every lane follows its own chain
    for (size_t span_mb : {1, 3, 64, 1024}) {
        const size_t nwords = span_mb * (1 << 20) / 4;
        hipMalloc(&g_tab, nwords * 4);
        std::vector<unsigned> t(nwords);
        unsigned long long s = 12345;
        for (size_t i = 0; i < nwords; ++i) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            t[i] = (unsigned)((s >> 33) % nwords) * 4u & ~63u;  // byte offset of a 64 B aligned slot
        }
        hipMemcpy(g_tab, t.data(), nwords * 4, hipMemcpyHostToDevice);
        const int it = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_global_chase, dim3(1), dim3(64), 0, 0, g_tab, g_io, it);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_global_chase, dim3(1), dim3(64), 0, 0, g_tab, g_io, it);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("global gather chase, table %4zu MiB: %8.1f ns per dependent gather \n", span_mb, ms * 1e6 / (it * 16.0));
        hipFree(g_tab);
    }
    return 0;
}
