// bisbm_device.hpp -- device-side building blocks of the MH sweep engine (gfx950, wave64).
//
// One wavefront owns one Markov chain.  Everything here is wave-synchronous: control flow is
// wave-uniform, lanes parallelise the per-node work (lane j <-> neighbour j for the CSR walk,
// lane t <-> opposite-type block t for the dS / Hastings terms and the proposal CDF).
//
// Reference lines cited as <file>:<line> relative to /root/reference/src.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bisbm {

constexpr int kWave = 64;
constexpr int kQNmax = 10000;  // blockmodel.cc:48 init_q_cache(10000)

enum : int { RNG_PHILOX = 0, RNG_COMPAT = 1 };
enum : int { SCHED_EXPONENTIAL = 0, SCHED_LINEAR = 1, SCHED_LOGARITHMIC = 2, SCHED_CONSTANT = 3, SCHED_ABRUPT = 4 };
enum : uint32_t { PHX_STEP_A = 0, PHX_STEP_B = 1, PHX_SWEEP_KEY = 2, PHX_INIT_SHUFFLE = 3 };

// ------------------------------------------------------------------------------------------
// cross-lane helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ double bcast(double x, int src) { return __shfl(x, src, kWave); }
__device__ __forceinline__ int bcast(int x, int src) { return __shfl(x, src, kWave); }

// v_readlane with a wave-uniform lane index: the value lands in scalar registers, no LDS crossbar
__device__ __forceinline__ int readlane(int x, uint32_t src) { return __builtin_amdgcn_readlane(x, (int)src); }
__device__ __forceinline__ uint32_t readlane(uint32_t x, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)src);
}
__device__ __forceinline__ double readlane(double x, uint32_t src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), (int)src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), (int)src);
    return __hiloint2double(hi, lo);
}

// DPP controls (CDNA ISA): quad_perm, row_shr:n, row_mirror, row_half_mirror, row_bcast15/31
constexpr int kDppXor1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // lane i <-> 7-i inside each group of 8
constexpr int kDppMirror = 0x140;      // lane i <-> 15-i inside each row of 16
constexpr int kDppBcast15 = 0x142;
constexpr int kDppBcast31 = 0x143;

// One DPP move of a double.  The destination's previous content is left undefined (v_mov_b32_dpp with
// no separate `old` register to initialise): with a row mask, lanes of masked-off rows hold garbage
// afterwards, so callers must only consume lanes of enabled rows.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

// Fixed 64-leaf xor butterfly, levels 32, then 1,2,4,8,16: the summation tree of Philox mode (the CPU checker restates the
// same tree).  Level 32 comes first -- leaf l + leaf l ^ 32, one v_permlane32_swap pair: afterwards both halves of the wave hold
// the same 32 partial sums, which is what lets a variant that keeps TWO leaves per lane (two steps per pass with more than 32
// blocks) add its own pair first, and two sums share one butterfly (butterfly_pair64).  Levels 1 and 2 are quad permutes; after
// them every lane of a quad holds the quad's sum, so the mirror permutes pair the same partial sums as xor 4 / xor 8 would (FP
// add is commutative: same bits); level 16 is R1 += R0 (row_bcast15 into rows 1, 3).  With at most 32 / 16 / 8 leaves in use
// the others are +0.0 and the levels that only add them drop out (butterfly_sum_low32, butterfly_rows16, butterfly_groups8:
// same bits).  Rows 1 and 3 are complete at the end; lane 63 is returned wave-uniform.
__device__ __forceinline__ double butterfly_fold32(double x) {  // every lane: leaf l + leaf l ^ 32
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);  // [x.lo, x.lo] + [x.hi, x.hi]
}
__device__ __forceinline__ double butterfly_sum(double x) {
    x = butterfly_fold32(x);
    x = x + dpp_f64<kDppXor1>(x);
    x = x + dpp_f64<kDppXor2>(x);
    x = x + dpp_f64<kDppHalfMirror>(x);
    x = x + dpp_f64<kDppMirror>(x);
    x = x + dpp_f64<kDppBcast15, 0xA>(x);
    return readlane(x, 63u);
}

// The two Hastings sums (accu0, accu1) use the tree with level 16 SECOND -- levels 32, 16, then 1,2,4,8 -- so that a pass which
// holds one step per 32-lane half can fold BOTH sums with one v_permlane16_swap pair (rows 0 / 2 keep accu0's leaves l + l ^ 16,
// rows 1 / 3 accu1's) and run the four row levels once for both (butterfly_accu_rows32): 17 instructions instead of 30.  The
// dS sum keeps levels 32, 1,2,4,8,16 (butterfly_sum and its shortened forms).  The CPU checker restates both trees.
__device__ __forceinline__ double butterfly_fold16(double x) {  // every lane: leaf l + leaf l ^ 16 (rows 0 <-> 1, 2 <-> 3)
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);  // [r0, r0, r2, r2] + [r1, r1, r3, r3]
}
__device__ __forceinline__ double butterfly_rows_1248(double x) {
    x = x + dpp_f64<kDppXor1>(x);
    x = x + dpp_f64<kDppXor2>(x);
    x = x + dpp_f64<kDppHalfMirror>(x);
    x = x + dpp_f64<kDppMirror>(x);
    return x;
}
// one Hastings sum over 64 leaves (the generic kernel)
__device__ __forceinline__ double butterfly_sum_accu(double x) {
    return readlane(butterfly_rows_1248(butterfly_fold16(butterfly_fold32(x))), 63u);
}
// accu0, accu1 over 64 leaves each (general step of the production kernel with more than 32 blocks): the level-32 swap leaves
// the lower half for a's fold and the upper half for b's
__device__ __forceinline__ void butterfly_pair64(double a, double b, double& sum_a, double& sum_b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    double x = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);  // [a.lo, b.lo] + [a.hi, b.hi]
    x = butterfly_rows_1248(butterfly_fold16(x));
    sum_a = readlane(x, 31u);
    sum_b = readlane(x, 63u);
}
// accu0, accu1 of TWO steps at once, one step per 32-lane half (leaves = the half's lanes; leaves 32..63 of the tree are
// +0.0).  Afterwards every lane of rows 1 and 3 holds both sums of its half.
__device__ __forceinline__ void butterfly_accu_rows32(double a0, double a1, double& accu0, double& accu1) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a0), __double2loint(a1), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a0), __double2hiint(a1), false, false);
    // [a0.r0, a1.r0, a0.r2, a1.r2] + [a0.r1, a1.r1, a0.r3, a1.r3]: level 16 of both sums
    const double x = butterfly_rows_1248(__hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]));
    accu1 = x;                                // rows 1, 3
    accu0 = dpp_f64<kDppBcast15, 0xA>(x);     // rows 0, 2 -> rows 1, 3
}

// The tree of rounds 1 and 2 (levels 1,2,4,8,16,32 in that order), kept for entropy()'s block-state sum, whose recorded
// reference values are compared to 1e-13.
__device__ __forceinline__ double butterfly_sum_levels_up(double x) {
    x = x + dpp_f64<kDppXor1>(x);
    x = x + dpp_f64<kDppXor2>(x);
    x = x + dpp_f64<kDppHalfMirror>(x);
    x = x + dpp_f64<kDppMirror>(x);
    x = x + dpp_f64<kDppBcast15, 0xA>(x);
    x = x + dpp_f64<kDppBcast31, 0xC>(x);
    return readlane(x, 63u);
}

// Two butterfly sums at once when both inputs are zero in lanes 32..63 (at most 32 leaves in use): b is
// moved to the upper half with v_permlane32_swap and one 32-leaf butterfly per half gives both sums
// (rows 1 and 3 are the complete ones).  Same tree as butterfly_sum: its first level would only add the
// empty upper half (x + 0.0 = x).
__device__ __forceinline__ void butterfly_pair32(double a, double b, double& sum_a, double& sum_b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    double x = __hiloint2double(hi[0], lo[0]);  // lanes 0..31: a, lanes 32..63: b's lanes 0..31
    x = butterfly_rows_1248(butterfly_fold16(x));  // (the Hastings sums' tree: level 16 before the row levels)
    sum_a = readlane(x, 31u);
    sum_b = readlane(x, 63u);
}

// The sum of a 64-leaf butterfly whose leaves 32..63 are all +0.0: its level 32 (the first) adds x + 0.0 = x, so the
// five levels of the lower half give the same bits.
__device__ __forceinline__ double butterfly_sum_low32(double x) {
    x = x + dpp_f64<kDppXor1>(x);
    x = x + dpp_f64<kDppXor2>(x);
    x = x + dpp_f64<kDppHalfMirror>(x);
    x = x + dpp_f64<kDppMirror>(x);
    x = x + dpp_f64<kDppBcast15, 0xA>(x);
    return readlane(x, 31u);
}

// Two 32-leaf butterflies at once, one per half of the wave (leaves = lanes 0..31 and lanes 32..63): the five levels of
// butterfly_sum_low32 without the read-out.  Afterwards every lane of row 1 (lanes 16..31) holds the sum of the lower
// half and every lane of row 3 (lanes 48..63) the sum of the upper half; rows 0 and 2 hold garbage.
__device__ __forceinline__ double butterfly_rows32(double x) {
    x = x + dpp_f64<kDppXor1>(x);
    x = x + dpp_f64<kDppXor2>(x);
    x = x + dpp_f64<kDppHalfMirror>(x);
    x = x + dpp_f64<kDppMirror>(x);
    x = x + dpp_f64<kDppBcast15, 0xA>(x);
    return x;
}

// Four 16-leaf butterflies at once, one per row of the wave: the first four levels of butterfly_sum.  Every lane of a row ends
// up with its row's sum (levels 1, 2 are quad permutes, then the two mirror permutes pair whole quads / octets).  Same bits
// as butterfly_sum over 64 leaves whose leaves 16..63 are +0.0: the last two levels add x + 0.0.
__device__ __forceinline__ double butterfly_rows16(double x) {
    x = x + dpp_f64<kDppXor1>(x);
    x = x + dpp_f64<kDppXor2>(x);
    x = x + dpp_f64<kDppHalfMirror>(x);
    x = x + dpp_f64<kDppMirror>(x);
    return x;
}

// Eight 8-leaf butterflies at once, one per group of eight lanes: the first three levels of butterfly_sum (two quad permutes,
// then lane i <-> 7 - i inside the group).  Every lane of a group ends up with its group's sum; same bits as the 64-leaf tree
// whose leaves 8..63 are +0.0.
__device__ __forceinline__ double butterfly_groups8(double x) {
    x = x + dpp_f64<kDppXor1>(x);
    x = x + dpp_f64<kDppXor2>(x);
    x = x + dpp_f64<kDppHalfMirror>(x);
    return x;
}

// inclusive prefix sum inside every group of eight lanes (lb = lane & 7): row_shr pulls across the middle of a 16-lane row, so
// the lanes whose source lies in the neighbouring group add nothing
__device__ __forceinline__ int group_inclusive_scan8(int x, uint32_t lb) {
    const int s1 = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += lb >= 1u ? s1 : 0;
    const int s2 = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += lb >= 2u ? s2 : 0;
    const int s4 = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += lb >= 4u ? s4 : 0;
    return x;
}

// inclusive prefix sum inside every 16-lane row (row_shr never leaves its row)
__device__ __forceinline__ int row_inclusive_scan16(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
    return x;
}

// inclusive prefix sum of int32 over the wave: Kogge-Stone inside rows (row_shr), then the row
// totals are carried with row_bcast15 / row_bcast31
__device__ __forceinline__ int wave_inclusive_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, kDppBcast15, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, kDppBcast31, 0xC, 0xF, false);
    return x;
}

// the same scan when only lanes 0..31 are consumed (no carry into rows 2,3 from row 1 needed)
__device__ __forceinline__ int wave_inclusive_scan32(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, kDppBcast15, 0xA, 0xF, false);
    return x;
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) and the production draw definitions
// ------------------------------------------------------------------------------------------
struct U4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ U4 phx_draw(uint64_t seed, uint32_t chain, uint32_t purpose, uint64_t idx) {
    return philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), chain, purpose, (uint32_t)seed,
                         (uint32_t)(seed >> 32));
}

// 53-bit uniform in [0,1) from two words
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    const uint64_t b = ((uint64_t)hi << 32) | lo;
    return (double)(b >> 11) * 0x1.0p-53;
}

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// Keyed bijection of [0,n): 4-round alternating Feistel network on max(2, bitlen(n-1)) bits with
// cycle walking.  Gives the per-sweep visit order (and the label shuffle) without storing a
// permutation: position i is evaluated on the fly, any lane can evaluate any position.
struct Feistel {
    uint32_t k0, k1, k2, k3;
    uint32_t n, wa, wb;
    __device__ __forceinline__ void init(U4 keys, uint32_t n_) {
        k0 = keys.x;
        k1 = keys.y;
        k2 = keys.z;
        k3 = keys.w;
        n = n_;
        uint32_t b = 0;
        while (((uint64_t)1 << b) < (uint64_t)n_) ++b;
        if (b < 2) b = 2;
        wa = b / 2;
        wb = b - wa;
    }
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const {
        if (n <= 1) return 0;
        const uint32_t ma = (1u << wa) - 1u, mb = (1u << wb) - 1u;
        uint32_t x = i;
        do {
            uint32_t A = x >> wb, B = x & mb;
            // widths alternate (wa,wb) -> (wb,wa) -> (wa,wb) -> ...
            uint32_t nA, nB;
            nA = B; nB = A ^ (mix32(B ^ k0) & ma); A = nA; B = nB;  // A: wb bits, B: wa bits
            nA = B; nB = A ^ (mix32(B ^ k1) & mb); A = nA; B = nB;  // A: wa bits, B: wb bits
            nA = B; nB = A ^ (mix32(B ^ k2) & ma); A = nA; B = nB;
            nA = B; nB = A ^ (mix32(B ^ k3) & mb); A = nA; B = nB;
            x = (A << wb) | B;
        } while (x >= n);
        return x;
    }
};

// The visit order of a Philox-mode phase: a keyed bijection of [0, n) that is LOCAL in node ids -- tiles of 4096
// consecutive ids in a keyed order, inside a tile its 64 cells of 64 ids in a keyed order, inside a cell the ids in a
// keyed order.  A cell is one chunk of the production kernel: its 64 adjacency rows are adjacent in the CSR arrays.
// The labels a step gathers belong to the neighbours of its node; where ids follow the graph's structure (as they
// do in most data sets, and as a reordering pass can arrange) consecutive steps gather from the same few KB of
// labels, which stay in the L2.  Measured at config 3 against a structure-blind Feistel order over the whole class:
// HBM traffic 1449 -> 440 B per update, L2 hit rate 39 -> 67 %, 1.14 -> 1.35e9 updates/s.  Cycle walking over the
// padded domain keeps it a bijection when n is not a multiple of 4096.
struct TiledOrder {
    static constexpr uint32_t kTileBits = 12;  // tiles of 4096 ids ...
    static constexpr uint32_t kCellBits = 6;   // ... made of cells of 64 ids (one chunk of the production kernel)
    Feistel tiles, cells, inner;
    uint32_t n;
    __device__ __forceinline__ void init(U4 keys, uint32_t n_) {
        n = n_;
        const uint32_t n_tiles = (n_ + (1u << kTileBits) - 1u) >> kTileBits;
        tiles.init(keys, n_tiles);
        // a class that fits one tile has only the cells it needs: the padded domain is then less than 64 ids larger than
        // the class instead of 4096 (cycle walking over 4096 positions for 18 nodes was the cost of a tiny graph's step)
        cells.init(U4{keys.y, keys.z, keys.w, keys.x},
                   n_tiles == 1u ? (n_ + (1u << kCellBits) - 1u) >> kCellBits : 1u << (kTileBits - kCellBits));
        inner.init(U4{keys.z, keys.w, keys.x, keys.y}, 1u << kCellBits);
    }
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const {
        constexpr uint32_t cmask = (1u << (kTileBits - kCellBits)) - 1u, omask = (1u << kCellBits) - 1u;
        uint32_t x = i;
        do {
            const uint32_t t = tiles(x >> kTileBits);
            Feistel f = cells;
            f.k0 ^= t * 0x9E3779B9u;  // every tile its own order of cells
            const uint32_t c = f((x >> kCellBits) & cmask);
            f = inner;
            f.k0 ^= ((t << (kTileBits - kCellBits)) | c) * 0x85EBCA6Bu;  // every cell its own inner order
            x = (t << kTileBits) | (c << kCellBits) | f(x & omask);
        } while (x >= n);
        return x;
    }
};

// ------------------------------------------------------------------------------------------
// std::mt19937 + libstdc++-11 distributions, state in LDS (compat mode)
// ------------------------------------------------------------------------------------------
// Every member and mt_shuffle are force-inlined: one out-of-line call taking the engine by reference would pin the
// whole struct (pointers and position) in scratch memory for the kernel's lifetime -- a vector-memory round trip per
// draw on a lone wave (measured: 3.4 us per compat step, most of it those round trips).
struct Mt {
    uint32_t* mt;  // 624 words in LDS
    int idx;       // wave-uniform
    uint32_t* tm = nullptr;  // optional, 624 words in LDS: the tempered outputs of mt[], so that a draw is one LDS read
                             // (the sweep kernel draws ~10 words per step; the state that is saved stays mt[] alone)

    static __device__ __forceinline__ uint32_t temper(uint32_t y) {
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    // fill tm[] from mt[] (after loading a saved state)
    __device__ __forceinline__ void retemper() {
        for (int k = lane_id(); k < 624; k += kWave) tm[k] = temper(mt[k]);
        __syncthreads();
    }
    // regenerate all 624 words; chunks of 64 lanes, reads of a chunk complete before its writes
    __device__ __forceinline__ void twist() {
        const int lane = lane_id();
#pragma nounroll
        for (int c = 0; c < 624; c += kWave) {
            const int k = c + lane;
            uint32_t v = 0;
            if (k < 624) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                v = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            __syncthreads();
            if (k < 624) {
                mt[k] = v;
                if (tm) tm[k] = temper(v);
            }
            __syncthreads();
        }
        idx = 0;
    }
    __device__ __forceinline__ uint32_t next() {
        if (idx >= 624) twist();
        if (tm) return tm[idx++];
        return temper(mt[idx++]);
    }
    // uniform_real_distribution<double>(0,1) -> generate_canonical<double,53> (random.tcc:3348-3384)
    static __device__ __forceinline__ double canonical_of(uint32_t w0, uint32_t w1) {
        double r = ((double)w0 + (double)w1 * 4294967296.0) / 18446744073709551616.0;
        if (r >= 1.0) r = 0x1.fffffffffffffp-1;  // nextafter(1, 0)
        return r;
    }
    __device__ __forceinline__ double canonical() {
        if (tm && idx + 2 <= 624) {  // both words in one LDS round trip
            const uint32_t w0 = tm[idx], w1 = tm[idx + 1];
            idx += 2;
            return canonical_of(w0, w1);
        }
        const uint32_t w0 = next();
        const uint32_t w1 = next();
        return canonical_of(w0, w1);
    }
    // The next four canonical() values at once, value p in every lane with (lane & 3) == p: one LDS round trip for all
    // the uniforms a step can draw from this engine.  Only when no regeneration falls inside (idx + 8 <= 624); the
    // caller advances idx by two words per value it consumed.
    __device__ __forceinline__ bool window4_ok() const { return tm != nullptr && idx + 8 <= 624; }
    __device__ __forceinline__ double window4() const {
        const int p = lane_id() & 3;
        return canonical_of(tm[idx + 2 * p], tm[idx + 2 * p + 1]);
    }
    // Lemire downscale for a 32-bit URBG (uniform_int_dist.h:245-272)
    __device__ __forceinline__ uint32_t lemire(uint32_t range) {
        uint64_t product = (uint64_t)next() * (uint64_t)range;
        uint32_t low = (uint32_t)product;
        if (low < range) {
            const uint32_t threshold = (0u - range) % range;
            while (low < threshold) {
                product = (uint64_t)next() * (uint64_t)range;
                low = (uint32_t)product;
            }
        }
        return (uint32_t)(product >> 32);
    }
};

// std::shuffle (stl_algo.h:3729-3793) of v[0..n) with wave-uniform control flow.  T is uint32_t
// (vlist) or uint8_t (labels).  Every lane computes the same indices; lane 0 stores.
template <class T>
__device__ __forceinline__ void mt_shuffle(Mt& g, T* v, uint32_t n) {
    if (n == 0) return;
    const int lane = lane_id();
    auto swap_at = [&](uint32_t i, uint32_t j) {
        const T a = v[i], b = v[j];
        __syncthreads();
        if (lane == 0) {
            v[i] = b;
            v[j] = a;
        }
        __syncthreads();
    };
    if (0xFFFFFFFFull / n >= n) {
        uint32_t i = 1;
        if ((n % 2) == 0) {
            swap_at(i, g.lemire(2));
            ++i;
        }
        while (i < n) {
            const uint32_t s = i + 1;
            const uint32_t x = g.lemire(s * (s + 1));
            swap_at(i, x / (s + 1));
            ++i;
            swap_at(i, x % (s + 1));
            ++i;
        }
        return;
    }
    for (uint32_t i = 1; i < n; ++i) swap_at(i, g.lemire(i + 1));
}

// ------------------------------------------------------------------------------------------
// numerics: lgamma table, log_q table / approximation (support/cache.hh, int_part.{hh,cc}, spence.cc)
// ------------------------------------------------------------------------------------------
struct Tables {
    const double* lg;  // lg[i] = glibc lgamma(i), lg[0] = +inf (cache.cc:64-79), built on the host
    uint64_t lg_size;
    const double* q;    // q[n * q_stride + k], n <= 10000, k <= q_kcap (int_part.cc:34-51)
    uint32_t q_stride;  // q_kcap + 1
    const double* logtab;  // logtab[i] = glibc log(i) (the reference's __safelog_cache, cache.cc:25-37), lg_size entries
};

__device__ __forceinline__ double lgamma_fast(const Tables& t, long long x) {
#if defined(BISBM_ABLATE) && (BISBM_ABLATE & 8)
    return (double)x * 1e-3;  // diagnostic build: table gathers removed (wrong results)
#endif
    if ((unsigned long long)x < t.lg_size) return t.lg[x];
    return NAN;  // the host sizes the table to cover every index the kernels can form (bisbm_create)
}

__device__ __forceinline__ double lbinom_fast(const Tables& t, unsigned long long N, unsigned long long k) {
    if (N == 0 || k == 0 || k > N) return 0;  // util.hh:41-47
    return (lgamma_fast(t, (long long)(N + 1)) - lgamma_fast(t, (long long)(k + 1))) -
           lgamma_fast(t, (long long)(N - k + 1));
}

__device__ __forceinline__ double polevl8(double x, const double* c) {  // spence.cc:91-106, N = 7
    double ans = c[0];
#pragma unroll
    for (int i = 1; i <= 7; ++i) ans = ans * x + c[i];
    return ans;
}

__device__ inline double spence(double x) {  // spence.cc:108-154 (Cephes dilogarithm)
    const double A[8] = {4.65128586073990045278E-5, 7.31589045238094711071E-3, 1.33847639578309018650E-1,
                         8.79691311754530315341E-1, 2.71149851196553469920E0,  4.25697156008121755724E0,
                         3.29771340985225106936E0,  1.00000000000000000126E0};
    const double B[8] = {6.90990488912553276999E-4, 2.54043763932544379113E-2, 2.82974860602568089943E-1,
                         1.41172597751831069617E0,  3.63800533345137075418E0,  5.03278880143316990390E0,
                         3.54771340985225096217E0,  9.99999999999999998740E-1};
    const double kPi = 3.14159265358979323846;
    double w, y, z;
    int flag = 0;
    if (x < 0.0) return NAN;
    if (x == 1.0) return 0.0;
    if (x == 0.0) return kPi * kPi / 6.0;
    if (x > 2.0) {
        x = 1.0 / x;
        flag |= 2;
    }
    if (x > 1.5) {
        w = (1.0 / x) - 1.0;
        flag |= 2;
    } else if (x < 0.5) {
        w = -x;
        flag |= 1;
    } else
        w = x - 1.0;
    y = -w * polevl8(w, A) / polevl8(w, B);
    if (flag & 1) y = (kPi * kPi) / 6.0 - log(x) * log1p(-x) - y;
    if (flag & 2) {
        z = log(x);
        y = -0.5 * z * z - y;
    }
    return y;
}

__device__ inline double get_v(double u) {  // int_part.cc:77-87
    double v = u;
    double delta = 1;
    int guard = 0;
    while (delta > 1e-8 && guard < 1000) {  // the guard only bounds a wave that would never finish
        const double n_v = u * sqrt(spence(exp(-v)));
        delta = fabs(n_v - v);
        v = n_v;
        ++guard;
    }
    return v;
}

// exp(z) to ~2e-7 relative for any finite z: f64 range reduction, v_exp_f32 on the fraction (over/underflow
// through ldexp)
__device__ __forceinline__ double exp_lowprec(double z) {
    const double t = z * 0x1.71547652b82fep+0;
    const double ti = rint(t);
    const float f = (float)(t - ti);
    const int e = (int)fmax(fmin(ti, 2000.0), -2000.0);
    return ldexp((double)__builtin_amdgcn_exp2f(f), e);
}

// Metropolis test  lhs < rhs0 * exp(z)  with the exact exp only when a 2e-7-accurate one cannot decide:
// the outcome always equals the exact expression's (margin 1e-5 relative >> the estimate's error).
__device__ __forceinline__ bool less_than_scaled_exp(double lhs, double rhs0, double z) {
    const double est = rhs0 * exp_lowprec(z);
    if (__builtin_expect(fabs(lhs - est) > 1e-5 * est, 1)) return lhs < est;
    return lhs < rhs0 * exp(z);
}

// Constants of the closed-form tier of log_q_approx (u = k / sqrt(n) > 18), see log_q_closed.
// the closed-form tier of log_q_approx begins at u = k / sqrt(n) > 18 (k^2 > 324 n), see log_q_closed
constexpr double kDirectU2 = 324.0;
struct LogQConsts {
    double nc0l2e;  // -(pi/sqrt 6) log2(e)
    double c1c0;    // (3/pi^2)(pi/sqrt 6)
    double c1;      // 3/pi^2
    double c2c0;    // 2 pi/sqrt 6
    double lfc;     // log(pi/sqrt 6) - 1.5 log 2 - log pi
};
__device__ __forceinline__ LogQConsts log_q_consts() {
    return {-0x1.d9af1d38092ecp+0, 0x1.37423899a1558p-2 * 0x1.48552f88091a8p+0, 0x1.37423899a1558p-2,
            2 * 0x1.48552f88091a8p+0, -0x1.ef8383c50bb74p+0};
}

// 2^t for t <= 0 of moderate size to ~1e-7 relative: f64 range reduction, v_exp_f32 on the fraction.  The
// integer part goes through v_cvt_i32_f64 itself (saturating, NaN -> 0), so any t is safe.
__device__ __forceinline__ double exp2_filter(double t) {
    const double ti = rint(t);
    const float f = (float)(t - ti);  // |f| <= 0.5
    int e;
    __asm__("v_cvt_i32_f64 %0, %1" : "=v"(e) : "v"(ti));
    return ldexp((double)__builtin_amdgcn_exp2f(f), e);
}

// sqrt(n) and 1/sqrt(n) without a division: hardware reciprocal square root, one Newton step on it, then one
// residual step on the root itself (sq within 1 ulp; r good to ~1e-15)
__device__ __forceinline__ void sqrt_rsqrt(double nd, double& sq, double& r) {
    const double r0 = __builtin_amdgcn_rsq(nd);
    r = __builtin_fma(0.5 * r0, __builtin_fma(-(nd * r0), r0, 1.0), r0);
    const double s0 = nd * r;
    sq = __builtin_fma(__builtin_fma(-s0, s0, nd), 0.5 * r, s0);
}

// log_q_approx for u = k / sqrt(n) > 18 (Philox mode; > 24 until late round 4).  There the get_v iteration's limit can be written down
// directly: v = C0 u (1 - eps), eps = C1 (C0 u + 1) x, x = exp(-C0 u), C0 = pi/sqrt 6, C1 = 3/pi^2 (the first
// iterate's own correction changes x by < 1e-8 relative, i.e. the result by < 1e-17), and the closing
// formula becomes
//   (LFC - log n + 2 C0 sq) + x (k + (1 + u^2/2)/2) - eps (2 C0 sq + 1).
// log(n) comes from the host-built table; the x terms are < 1.7e-9 of the result and are evaluated with fused
// multiply-adds and a 1e-7-accurate exponential (against the converged evaluation for u in [17, 30], n up to 6e7, with x off by
// 1.5e-7 either way: within 5e-16 relative -- the second-order term of the tier below is < 1e-17 from u = 18 on).  Against the literal evaluation for u in [24, 70], n up to
// 6e7: within 6e-16 relative.
// (log_q_closed_x: the formula for a given x -- the tier 13 <= u <= 18 evaluates it with a more accurate x and adds its
// second-order term, log_q_closed2)
__device__ __forceinline__ double log_q_closed_x(double kd, double u, double x, double sq, double logn, const LogQConsts& c) {
    const double eps = __builtin_fma(c.c1c0, u, c.c1) * x;
    const double a = __builtin_fma(u * u, 0.25, kd + 0.5);
    const double t2 = c.c2c0 * sq;
    const double corr = __builtin_fma(x, a, -__builtin_fma(eps, t2, eps));
#ifndef BISBM_EXP_NO_LOGN_LATE
    // (scheduling only: log n comes out of a table gather that is still in flight when the evaluation starts -- tying it to
    // `corr` here keeps the compiler from placing its first use, and with it the wait for the gather, at the top)
    __asm__ volatile("" : "+v"(logn) : "v"(corr));
#endif
    return ((c.lfc - logn) + t2) + corr;
}
// (written out rather than through log_q_closed_x: the same operations in the same order -- the same bits --, but the compiler's
// schedule of the all-lanes-far-out path of the hot steps, the bench line's path, then stays the one of round 3)
__device__ __forceinline__ double log_q_closed(double kd, double sq, double r, double logn, const LogQConsts& c) {
    const double u = kd * r;
    const double x = exp2_filter(u * c.nc0l2e);
    const double eps = __builtin_fma(c.c1c0, u, c.c1) * x;
    const double a = __builtin_fma(u * u, 0.25, kd + 0.5);
    const double t2 = c.c2c0 * sq;
    const double corr = __builtin_fma(x, a, -__builtin_fma(eps, t2, eps));
#ifndef BISBM_EXP_NO_LOGN_LATE
    __asm__ volatile("" : "+v"(logn) : "v"(corr));
#endif
    return ((c.lfc - logn) + t2) + corr;
}

// exp(-t) for 0 <= t < 700 to ~2e-16 relative: ln 2 in two parts, degree-12 Taylor polynomial on |r| <= ln(2)/2
__device__ __forceinline__ double exp_neg(double t) {
    const double kf = rint(t * -0x1.71547652b82fep+0);              // -t log2(e), rounded
    double r = __builtin_fma(kf, -0x1.62e42fee00000p-1, -t);         // -t - kf ln2_hi (exact)
    r = __builtin_fma(kf, -0x1.a39ef35793c76p-33, r);               //        - kf ln2_lo
    double p = 0x1.1eed8eff8d898p-29;                               // 1/12!
    p = __builtin_fma(p, r, 0x1.ae64567f544e4p-26);                 // 1/11!
    p = __builtin_fma(p, r, 0x1.27e4fb7789f5cp-22);                 // 1/10!
    p = __builtin_fma(p, r, 0x1.71de3a556c734p-19);                 // 1/9!
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-16);                 // 1/8!
    p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-13);                 // 1/7!
    p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);                 // 1/6!
    p = __builtin_fma(p, r, 0x1.1111111111111p-7);                  // 1/5!
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);                  // 1/4!
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);                  // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    int e;
    __asm__("v_cvt_i32_f64 %0, %1" : "=v"(e) : "v"(kf));
    return ldexp(p, e);
}
// exp(-t) for 0 <= t < 700 to 7e-9 relative: one ln 2, degree-7 polynomial (the x0 of log_q_closed2: what it multiplies
// is < 2.5e-8 of the result)
__device__ __forceinline__ double exp_neg7(double t) {
    const double kf = rint(t * -0x1.71547652b82fep+0);
    const double r = __builtin_fma(kf, -0x1.62e42fefa39efp-1, -t);   // -t - kf ln 2
    double p = 0x1.a01a01a01a01ap-13;                               // 1/7!
    p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);                 // 1/6!
    p = __builtin_fma(p, r, 0x1.1111111111111p-7);                  // 1/5!
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);                  // 1/4!
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);                  // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    int e;
    __asm__("v_cvt_i32_f64 %0, %1" : "=v"(e) : "v"(kf));
    return ldexp(p, e);
}

// log_q_approx for 13 <= u = k / sqrt(n) <= 18 (Philox mode; <= 24 until late round 4): the fixed point of get_v to SECOND order in x0 = exp(-C0 u).
// With a = C0 u, v = a (1 - delta) and spence(x) = pi^2/6 - T, T = sum_k (v/k + 1/k^2) x^k, the fixed point v^2 = u^2
// spence(exp(-v)) reads (1 - delta)^2 = 1 - T / (pi^2/6), i.e. delta = w/2 + w^2/8 + ..., w = T/(pi^2/6).  First order:
// delta1 = e1 x0, e1 = C1 (a + 1): that is log_q_closed's formula.  Feeding v = a (1 - delta1), x = x0 (1 + a delta1)
// back into T and into the closing formulas (int_part.cc:94-97, their logarithms as two-term series) and collecting what is of
// order x0^2 -- the terms k a e1 and 2 C0 sq C1 a^2 e1 cancel, because 2 C0^2 C1 = 1 -- leaves
//   log_q = closed(x0) + x0^2 [ h (a e1 / 2 + h / 4) - e1^2 (1 + C0 sq) - C1 (a^2 e1 + a / 2 + 1/4 + C0 sq / 2) ],  h = 1 + u^2 / 2,
// up to terms of relative size a^2 x0 <= 1.6e-5 of a correction that is itself <= 5e-14 of the result (round 4; rounds 1-3
// evaluated delta, x and the closing formulas one after the other: ~55 instructions where this takes ~45 -- and a pass
// whose arguments straddle the tier boundary shares the closed form's arithmetic between the two tiers instead of evaluating two
// formulas).  x0 from a 7e-9-accurate exponential (it multiplies < 2.5e-8 of the result).  Against the converged
// evaluation (the CPU checker's Philox-mode log_q): <= 1.3e-15 relative for u >= 13.
__device__ __forceinline__ double log_q_delta2(double u, double x0, double sq, const LogQConsts& c) {
    const double a = 0x1.48552f88091a8p+0 * u;  // (pi / sqrt 6) u
    const double e1 = __builtin_fma(c.c1c0, u, c.c1);
    const double ae = a * e1;
    const double h = __builtin_fma(0.5 * u, u, 1.0);
    const double t2 = c.c2c0 * sq;              // 2 C0 sq
    const double p1 = h * __builtin_fma(0.25, h, 0.5 * ae);
    const double p2 = (e1 * e1) * __builtin_fma(0.5, t2, 1.0);
    const double p3 = c.c1 * (__builtin_fma(ae, a, __builtin_fma(0.5, a, 0.25)) + 0.25 * t2);
    return (x0 * x0) * ((p1 - p2) - p3);
}
__device__ __forceinline__ double log_q_closed2(double kd, double sq, double r, double logn, const LogQConsts& c) {
    const double u = kd * r;
    const double x0 = exp_neg7(0x1.48552f88091a8p+0 * u);
    return log_q_closed_x(kd, u, x0, sq, logn, c) + log_q_delta2(u, x0, sq, c);
}

// log_q_approx for 8 <= u = k / sqrt(n) <= 24 (Philox mode): the reference's formulas (int_part.cc:77-98) with the
// fixed point v = u sqrt(spence(exp(-v))) of get_v taken to convergence instead of to its |dv| <= 1e-8 stop, and no
// library calls but one exponential.  With x = exp(-v) <= 3.6e-5 here,
//   spence(x) = pi^2/6 + v log1p(-x) - Li2(x) = pi^2/6 - sum_k (v/k + 1/k^2) x^k        (three terms: x^4 v/4 < 5e-18),
//   d spence / dv = v x / (1 - x),
// so two Newton steps from v0 = (pi/sqrt 6) u land on the root to ~1e-15 (the plain iteration contracts by
// rho = u^2 x / 2 <= 1.2e-3 per step; Newton squares the 1.2e-3 starting error twice); x follows v through
// exp(-dv) as a polynomial (|dv| <= 1.3e-3).  At the root v / u = sqrt(spence), so
//   log v - log u = log(pi/sqrt 6) + log1p(-T / (pi^2/6)) / 2,      T = pi^2/6 - spence,
// and the two remaining log1p arguments are <= 1.2e-3: five and six series terms.
// Against the literal evaluation as restated for the tests: <= 5e-16 relative for u >= 13, <= 1.5e-14 for
// 10 <= u < 13, <= 1.1e-12 for 8 <= u < 10 -- the differences are the literal's stopping tolerance: iterated to
// |dv| <= 1e-14 it returns this function's value to the last digit.
__device__ __forceinline__ double log_q_mid(double kd, double sq, double r, double logn, const LogQConsts& c) {
    const double kC6 = 0x1.a51a6625307d3p+0;  // pi^2 / 6
    const double kThird = 0x1.5555555555555p-2, kNinth = 0x1.c71c71c71c71cp-4;
    const double u = kd * r;
    double v = 0x1.48552f88091a8p+0 * u;  // (pi / sqrt 6) u
    double x = exp_neg(v);
    double T, s, rs;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        T = x * ((v + 1.0) + x * (__builtin_fma(0.5, v, 0.25) + x * __builtin_fma(kThird, v, kNinth)));
        sqrt_rsqrt(kC6 - T, s, rs);
        const double rho = (0.5 * u) * rs * (v * x) * (1.0 + x);
        const double d = __builtin_fma(u, s, -v);
        const double dv = __builtin_fma(d, __builtin_fma(rho, rho, rho), d);  // d / (1 - rho) to O(rho^3)
        v = v + dv;
        const double t = -dv;  // x <- x exp(-dv)
        double e = __builtin_fma(0.2, t, 1.0);
        e = __builtin_fma(0.25 * t, e, 1.0);
        e = __builtin_fma(kThird * t, e, 1.0);
        e = __builtin_fma(0.5 * t, e, 1.0);
        e = __builtin_fma(t, e, 1.0);
        x = x * e;
    }
    T = x * ((v + 1.0) + x * (__builtin_fma(0.5, v, 0.25) + x * __builtin_fma(kThird, v, kNinth)));
    sqrt_rsqrt(kC6 - T, s, rs);
    const double t = T * 0x1.37423899a1558p-1;  // T / (pi^2/6)
    double l1 = __builtin_fma(0.2, t, 0.25);    // log1p(-t) = -t (1 + t/2 + t^2/3 + t^3/4 + t^4/5)
    l1 = __builtin_fma(l1, t, kThird);
    l1 = __builtin_fma(l1, t, 0.5);
    l1 = __builtin_fma(l1, t, 1.0);
    l1 = -t * l1;
    const double y = x * __builtin_fma(0.5 * u, u, 1.0);
    double l2 = __builtin_fma(0x1.5555555555555p-3, y, 0.2);  // log1p(-y), one term more
    l2 = __builtin_fma(l2, y, 0.25);
    l2 = __builtin_fma(l2, y, kThird);
    l2 = __builtin_fma(l2, y, 0.5);
    l2 = __builtin_fma(l2, y, 1.0);
    l2 = -y * l2;
    const double lf = __builtin_fma(0.5, l1 - l2, c.lfc);                                  // :94-95
    const double g = __builtin_fma(u * x, __builtin_fma(x, __builtin_fma(kThird, x, 0.5), 1.0), 2.0 * s);  // :96
    return (lf - logn) + sq * g;                                                           // :97
}

// log1p(-a) for 0 <= a <= 0.25 as -2 atanh(a / (2 - a)): nine odd powers of z <= 0.143 (z^19 / 19 < 1e-17)
__device__ __forceinline__ double log1p_neg(double a) {
    const double z = a / (2.0 - a), z2 = z * z;
    double p = 1.0 / 17.0;
    p = __builtin_fma(p, z2, 1.0 / 15.0);
    p = __builtin_fma(p, z2, 1.0 / 13.0);
    p = __builtin_fma(p, z2, 1.0 / 11.0);
    p = __builtin_fma(p, z2, 1.0 / 9.0);
    p = __builtin_fma(p, z2, 1.0 / 7.0);
    p = __builtin_fma(p, z2, 0.2);
    p = __builtin_fma(p, z2, 0x1.5555555555555p-2);
    p = __builtin_fma(p, z2, 1.0);
    return -2.0 * z * p;
}

// log_q_approx for 2.5 <= u < 8 (Philox mode): dense graphs, where a block of n_r nodes has u = sqrt(n_r / mean
// degree).  Same scheme as log_q_mid -- the root of v = u sqrt(spence(exp(-v))) by Newton steps on the series of
// spence, then the reference's closing formulas -- with everything sized for x = exp(-v) <= 0.041: twelve series
// terms, four Newton steps, the exponential evaluated afresh for every step, log1p through atanh series.  Straight-line
// code (~400 instructions) instead of the literal loop over exp / spence / log (~1700 for u = 5).  It returns what the
// literal formulas return when get_v iterates to |dv| <= 1e-14 (<= 7e-16 relative); the reference's |dv| <= 1e-8 stop
// leaves up to 4e-10 relative in log_q at u = 2.5 (8e-11 at u = 5, 1e-11 at u = 7).
__device__ inline double log_q_low(double kd, double sq, double r, double logn, const LogQConsts& c) {
    const double kC6 = 0x1.a51a6625307d3p+0;  // pi^2 / 6
    const double u = kd * r;
    double v = 0x1.48552f88091a8p+0 * u;  // (pi / sqrt 6) u
    double x, T, s, rs;
    auto series = [&]() {  // T = pi^2/6 - spence(x) = sum_k (v/k + 1/k^2) x^k
        double p = __builtin_fma(1.0 / 12.0, v, 1.0 / 144.0);
        p = __builtin_fma(p, x, __builtin_fma(1.0 / 11.0, v, 1.0 / 121.0));
        p = __builtin_fma(p, x, __builtin_fma(0.1, v, 0.01));
        p = __builtin_fma(p, x, __builtin_fma(1.0 / 9.0, v, 1.0 / 81.0));
        p = __builtin_fma(p, x, __builtin_fma(0.125, v, 1.0 / 64.0));
        p = __builtin_fma(p, x, __builtin_fma(1.0 / 7.0, v, 1.0 / 49.0));
        p = __builtin_fma(p, x, __builtin_fma(1.0 / 6.0, v, 1.0 / 36.0));
        p = __builtin_fma(p, x, __builtin_fma(0.2, v, 0.04));
        p = __builtin_fma(p, x, __builtin_fma(0.25, v, 0.0625));
        p = __builtin_fma(p, x, __builtin_fma(0x1.5555555555555p-2, v, 1.0 / 9.0));
        p = __builtin_fma(p, x, __builtin_fma(0.5, v, 0.25));
        p = __builtin_fma(p, x, v + 1.0);
        T = p * x;
    };
    for (int it = 0; it < 4; ++it) {
        x = exp_neg(v);
        series();
        sqrt_rsqrt(kC6 - T, s, rs);
        const double rho = (0.5 * u) * rs * (v * x) * __builtin_fma(x, __builtin_fma(x, x + 1.0, 1.0), 1.0);  // u spence' / (2 sqrt spence)
        const double d = __builtin_fma(u, s, -v);
        const double w = __builtin_fma(__builtin_fma(rho, rho, rho), rho, rho);  // rho + rho^2 + rho^3
        v = __builtin_fma(d, w, v + d);
    }
    x = exp_neg(v);
    series();
    sqrt_rsqrt(kC6 - T, s, rs);
    const double l1 = log1p_neg(T * 0x1.37423899a1558p-1);           // log(spence / (pi^2/6)) = 2 (log v - log u - log(pi/sqrt 6))
    const double l2 = log1p_neg(x * __builtin_fma(0.5 * u, u, 1.0));
    const double lf = __builtin_fma(0.5, l1 - l2, c.lfc);            // :94-95
    const double g = __builtin_fma(-u, log1p_neg(x), 2.0 * s);      // :96
    return (lf - logn) + sq * g;                                     // :97
}

// log_q_approx, int_part.cc:89-98.
//
// Branch test `k < pow(n, 1/4.)` (:90) is evaluated as k^4 < n in integers: for n < 2^32 the
// correctly rounded pow can equal an integer only when n is a perfect fourth power, and the distance
// of n^(1/4) to the nearest integer is otherwise >= 1/(4 j^3) >> ulp, so the two tests agree.
//
// FAST (Philox mode only): u > 18 is log_q_closed, 13 <= u <= 18 log_q_closed2, 8 <= u < 13 log_q_mid, 2.5 <= u < 8
// log_q_low (all above); smaller
// u take the literal path.
template <bool FAST>
__device__ inline double log_q_approx(const Tables& t, unsigned long long n, unsigned long long k, double logn_pre) {
    const double kPi = 3.14159265358979323846;
    bool small;  // int_part.cc:90
    if (FAST) {  // n < 2^31 here: k^4 < n needs k < 216, so k^2 and k^4 fit 24-bit multiplies
        const uint32_t k2 = __umul24((uint32_t)k & 255u, (uint32_t)k & 255u);
        small = (uint32_t)k < 256u && __umul24(k2 & 0xffffu, k2 & 0xffffu) < (uint32_t)n;
    } else {
        small = k < 65536ull && (k * k) * (k * k) < n;
    }
    if (__builtin_expect(small, 0)) return lbinom_fast(t, n - 1, k - 1) - lgamma_fast(t, (long long)(k + 1));  // :73-75
    double sq, u;
    if (FAST) {
        // tiers by u^2 = k^2 / n against 18^2, 13^2, 8^2 and 2.5^2, in exact double arithmetic (k^2 < 2^52): the production kernel's
        // hot step makes the very same tests (logn_pre = logtab[n], loaded by the caller with the other gathers)
        const double kd = (double)(uint32_t)k, nd = (double)(uint32_t)n, k2 = kd * kd;
        double r;
        sqrt_rsqrt(nd, sq, r);
        if (__builtin_expect(k2 > kDirectU2 * nd, 1)) return log_q_closed(kd, sq, r, logn_pre, log_q_consts());
        if (k2 >= 169.0 * nd) return log_q_closed2(kd, sq, r, logn_pre, log_q_consts());  // u >= 13
        if (k2 >= ldexp(nd, 6)) return log_q_mid(kd, sq, r, logn_pre, log_q_consts());
        if (4.0 * k2 >= 25.0 * nd) return log_q_low(kd, sq, r, logn_pre, log_q_consts());  // u >= 2.5
        u = kd * r;
    } else {
        sq = sqrt((double)n);
        u = (double)k / sq;  // :92
    }
    const double v = get_v(u);
    const double lf = log(v) - log1p(-exp(-v) * (1 + u * u / 2)) / 2 - log(2.) * 3 / 2. - log(u) - log(kPi);
    const double g = 2 * v / u - u * log1p(-exp(-v));
    return lf - log((double)n) + sq * g;
}

// log_q for n <= 10000 straight from the table (int_part.hh:27-37 with k already min(k, n)): the same entries log_q()
// returns, without the tier ladder around them -- the only tier a graph with at most 10^4 edges ever uses
__device__ __forceinline__ double log_q_table(const Tables& t, int n, int k_min) {
    if (n <= 0 || k_min < 1) return 0;
    if ((uint32_t)k_min >= t.q_stride) return NAN;  // outside the uploaded columns (cannot happen on the sweep path)
    return t.q[(size_t)n * t.q_stride + (size_t)k_min];
}

// logn_pre: log(n) from the host table when the caller has it already (FAST only), else unused
template <bool FAST>
__device__ inline double log_q(const Tables& t, int n, int k, double logn_pre = 0.) {  // int_part.hh:27-37
    if (n <= 0 || k < 1) return 0;
    if (k > n) k = n;
    if (__builtin_expect(n < kQNmax + 1, 0)) {
        if ((uint32_t)k >= t.q_stride) return NAN;  // outside the uploaded columns (cannot happen on the sweep path)
        return t.q[(size_t)n * t.q_stride + (size_t)k];
    }
    return log_q_approx<FAST>(t, (unsigned long long)n, (unsigned long long)k, logn_pre);
}

}  // namespace bisbm
