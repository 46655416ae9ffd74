// Shared pieces of the one-wave-per-SIMD attention kernels (attn_bwd1w.hip: dK / dV; a dQ kernel and a forward kernel of the same build are kept, retired, under tools/exp): the 32-row slice image and its swizzle,
// MFMAs on literal accumulator registers, AGPR helpers.  Both kernels stream 32-row slices of two [rows][64] bf16 operands (Q | dO, resp. K | V) through an LDS ring
// and keep their per-wave operands and accumulators in AGPRs named literally in inline asm (the compiler must touch no AGPR itself: tests/test_build_cpu.py).
#pragma once
#include "common.h"
#include <utility>

namespace {

constexpr float LOG2E_1W = 1.4426950408889634f;
constexpr float LN2_1W = 0.6931471805599453f;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(3))) const char* lds_cptr;

enum { IMG_BYTES = 4096, STAGE_BYTES = 2 * IMG_BYTES };

// ---- the slice image: [32 rows][128 B], 16-byte chunk c of row r at r * 128 + ((c ^ swz(r)) << 4) -------------------------------------------------------------
// Row reads (ds_read_b128, lane = row, chunk 2 ks + hi): the instruction's 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) need 16 distinct
// (row & 1, chunk ^ swz) pairs -> swz must be distinct over the four rows of a group with equal (row & 1, (row >> 1) & 1): (row >> 2) & 3 is.
// Transposed reads (ds_read_b64_tr_b16, 32 lanes = rows r0 .. r0 + 3 (r0 % 4 == 0) x four consecutive chunks x two halves): rows r0 and r0 + 2 share a bank half and
// must take disjoint chunk sets -> bit 2 of swz = (row >> 1) & 1.  Both kinds of read are conflict-free (measured: SQ_LDS_BANK_CONFLICT = 0).
__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int img_off(int row, int c) { return row * 128 + ((c ^ swz(row)) << 4); }

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const bf16x2 t = {(bf16)a, (bf16)b};
    return *reinterpret_cast<const unsigned*>(&t);
}

// ---- MFMAs on literal accumulator registers: the compiler sees neither the AGPRs nor the instruction (no hazard recognizer, no register allocation for them) ----
// D (VGPRs) = A (VGPRs) x B (AGPRs a[BREG .. BREG + 3]) + C (VGPRs): the first k-step of a score / dP tile, started from the row constants
template <int BREG> __device__ __forceinline__ void mfma_vab_init(f32x16& d, const bf16x8& a, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c3:%c4], %2" : "=&v"(d) : "v"(a), "v"(c), "i"(BREG), "i"(BREG + 3));
}
// ... and its further k-steps (D = C)
template <int BREG> __device__ __forceinline__ void mfma_vab_more(f32x16& d, const bf16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(BREG), "i"(BREG + 3));
}
// D (AGPRs a[DREG .. DREG + 15]) += A (VGPRs) x B (VGPRs): a gradient tile
template <int DREG> __device__ __forceinline__ void mfma_agpr(const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(DREG), "i"(DREG + 15));
}
#define DEVIAS_A10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
// (the clobber lists are what makes the kernel descriptor allocate the registers)
__device__ __forceinline__ void agpr_claim128() {
    asm volatile("" ::: DEVIAS_A10(), DEVIAS_A10(1), DEVIAS_A10(2), DEVIAS_A10(3), DEVIAS_A10(4), DEVIAS_A10(5), DEVIAS_A10(6), DEVIAS_A10(7), DEVIAS_A10(8), DEVIAS_A10(9),
                 DEVIAS_A10(10), DEVIAS_A10(11), "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
}
__device__ __forceinline__ void agpr_claim96() {      // (the retired forward kernel's 64 + 32)
    asm volatile("" ::: DEVIAS_A10(), DEVIAS_A10(1), DEVIAS_A10(2), DEVIAS_A10(3), DEVIAS_A10(4), DEVIAS_A10(5), DEVIAS_A10(6), DEVIAS_A10(7), DEVIAS_A10(8),
                 "a90", "a91", "a92", "a93", "a94", "a95");
}
__device__ __forceinline__ void agpr_claim192() {
    agpr_claim128();
    asm volatile("" ::: "a128", "a129", DEVIAS_A10(13), DEVIAS_A10(14), DEVIAS_A10(15), DEVIAS_A10(16), DEVIAS_A10(17), DEVIAS_A10(18), "a190", "a191");
}
template <int I> __device__ __forceinline__ void agpr_zero1() { asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"i"(I)); }
template <int I> __device__ __forceinline__ void agpr_write1(unsigned v) { asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(v), "i"(I)); }
template <int I> __device__ __forceinline__ float agpr_read1() { float x; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(I)); return x; }
template <typename F, int... N> __device__ __forceinline__ void sfor_seq(F&& f, std::integer_sequence<int, N...>) { (f(std::integral_constant<int, N>{}), ...); }
template <int COUNT, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_seq(f, std::make_integer_sequence<int, COUNT>{}); }

template <int OFF> __device__ __forceinline__ u32x2 lds_tr_off(unsigned addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(r) : "v"(addr), "i"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ bf16x8 frag_of(const u32x2& lo, const u32x2& hi2) {
    const u32x4 w = {lo[0], lo[1], hi2[0], hi2[1]};
    return *reinterpret_cast<const bf16x8*>(&w);
}

}  // namespace

#define SB __builtin_amdgcn_sched_barrier(0);
