// rows_common.h -- device helpers shared by the LDS-staged row kernels (conv_rows.hip: stride 1; conv_rows_s2.hip: stride 2): 16-byte
// buffer-addressed LDS DMA, the in-place accumulating 16x16x4 fp32 MFMA as inline assembly, and the filter-image job both prepare with.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // a 16-byte access at any float address
typedef __attribute__((address_space(3))) void* lds_void_ptr;

constexpr unsigned kOob = 0x80000000u;
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)lds, 16, (int)voff, (int)soff, 0, 0);
}

// The accumulating MFMA as inline assembly, destination = addend, both in AGPRs.  Through the builtin the register allocator is free to
// give the result another register than the addend; across the unrolled stage loop that ended in a rotation of the whole accumulator file
// at every back edge (~100 v_accvgpr_mov / read / write per stage in the round-5 ISA, 10 - 15 % of a stage).  What the compiler's hazard
// recogniser would have done for a builtin is written out: s_nop 1 in front (VALU / v_accvgpr_write result -> MFMA operand: 2 wait
// states; hidden behind the previous MFMA's 8 passes), and acc_settle() before anything else reads the accumulators.
__device__ __forceinline__ void mfma16(f32x4& c, float a, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// every accumulator of a[0..N) has left the MFMA pipeline (8 passes + write-back < 32 cycles): the wait sits in the first statement, the
// others only tie their accumulators behind it (asm volatile statements keep their order)
template <int N>
__device__ __forceinline__ void acc_settle(f32x4* a) {
    static_assert(N >= 7, "at least one group of seven");
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a[0]), "+a"(a[1]), "+a"(a[2]), "+a"(a[3]), "+a"(a[4]), "+a"(a[5]), "+a"(a[6]));
#pragma unroll
    for (int i = 7; i + 7 <= N; i += 7)
        asm volatile("" : "+a"(a[i]), "+a"(a[i + 1]), "+a"(a[i + 2]), "+a"(a[i + 3]), "+a"(a[i + 4]), "+a"(a[i + 5]), "+a"(a[i + 6]));
#pragma unroll
    for (int i = N - N % 7; i < N; ++i) asm volatile("" : "+a"(a[i]));
}

// the same for any count (the stride-2 kernels hold 3 .. 52 accumulators per wave)
template <int N>
__device__ __forceinline__ void acc_settle_n(f32x4* a) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a[0]));
#pragma unroll
    for (int i = 1; i < N; ++i) asm volatile("" : "+a"(a[i]));
}

// compile-time loops (the body gets its index as a std::integral_constant: usable as an immediate operand of inline assembly)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>());
}
// one LDS read with an immediate offset, in program order; the value is NOT there until an lgkm_wait that covers it (the compiler does
// not know about the pending write: pass the register through an empty asm volatile behind the wait before anything uses it)
template <int OFF>
__device__ __forceinline__ void lds_rd(float& dst, unsigned addr_bytes) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit immediate");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr_bytes), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is four bits");
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
}

constexpr int stride16(int len) {  // smallest stride >= len that is 16 (mod 32)
    return len <= 16 ? 16 : (len - 16 + 31) / 32 * 32 + 16;
}

}  // namespace
