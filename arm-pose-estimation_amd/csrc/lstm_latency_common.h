// Helpers shared by the two latency kernels (lstm_cluster_small.hip: 1-4 windows without dropout; lstm_cluster_mc.hip: one
// window, up to 64 Monte-Carlo dropout samples): lane sums without the LDS crossbar, the polled 16-byte granule loads, the
// register-resident GEMV spans.
#pragma once
#include "ape_internal.h"

namespace {

typedef unsigned u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
typedef unsigned u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned))));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr int MR = 4;                      // rows per cluster of the small-batch kernel (row = lane group)

// NI polling loads (16 bytes per lane, L1-bypassing) and the wait for them in ONE statement: the compiler knows nothing of the
// asynchronous return, so no use (or copy) of a result may be scheduled between a load and the wait.
// The leading `s_nop 4`: a VALU instruction that writes a scalar register needs 5 wait states before a vector-memory instruction
// reads it, and hipcc pads only its own loads -- when it reloads a spilled descriptor with v_readlane_b32 right in front of this
// statement (the diagnostic build of lstm_mc_small.hip did: every launch aborted on a stale descriptor) nothing else provides them
template <int NI>
__device__ __forceinline__ void poll_granules(u32x4 (&v)[NI], const unsigned (&off)[NI], u32x4 rsrc) {
    static_assert(NI >= 1 && NI <= 4, "pairs per thread");
    if constexpr (NI == 1)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]) : "v"(off[0]), "s"(rsrc) : "memory");
    else if constexpr (NI == 2)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %4, 0 offen sc1\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]) : "v"(off[0]), "v"(off[1]), "s"(rsrc) : "memory");
    else if constexpr (NI == 3)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %3, %6, 0 offen sc1\n\tbuffer_load_dwordx4 %1, %4, %6, 0 offen sc1\n\t"
                     "buffer_load_dwordx4 %2, %5, %6, 0 offen sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "s"(rsrc) : "memory");
    else
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %4, %8, 0 offen sc1\n\tbuffer_load_dwordx4 %1, %5, %8, 0 offen sc1\n\t"
                     "buffer_load_dwordx4 %2, %6, %8, 0 offen sc1\n\tbuffer_load_dwordx4 %3, %7, %8, 0 offen sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                     : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(rsrc) : "memory");
}

__device__ __forceinline__ float gate_act(float v, bool is_tanh) {
    const float e = __builtin_amdgcn_exp2f((is_tanh ? -2.885390081777927f : -1.4426950408889634f) * v);
    const float s = __builtin_amdgcn_rcpf(1.0f + e);
    return is_tanh ? 2.0f * s - 1.0f : s;
}

// value of lane (quad base + K) for every lane of the quad (DPP quad_perm broadcast)
template <int K>
__device__ __forceinline__ float quad_bcast(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x),
                                                                 K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, false));
}

// x + (x of lane ^ 16), x + (x of lane ^ 32): gfx950's row / half swaps instead of a trip through the LDS crossbar
// (ds_bpermute, ~100+ cycles each, and the k-group sums of a phase are a chain of them).  v_permlane16_swap exchanges the odd
// 16-lane rows of its first operand with the even rows of its second: with both = x the operands become [x0 x0 x2 x2] and
// [x1 x1 x3 x3], whose sum is the xor-16 all-reduce; v_permlane32_swap does the same with the wave's halves.  (Inline asm:
// the builtin's two results are folded into one by this compiler when both inputs are the same value.)
__device__ __forceinline__ float sum_xor16(float x) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float sum_xor32(float x) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// x + (x of the lane 8 further within its 16-lane row): one DPP row rotation
__device__ __forceinline__ float sum_ror8(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, false));
}

// part[m] += sum over this lane's k-slice of w * act[m]; NQ KB-deep blocks (KB = 4 x the k-groups of a wave) starting at
// weight register w0.
// The activation fragments of up to 16 k-blocks are fetched up front, then multiplied: one dependent LDS read per k-block
// costs its whole latency at one wave per SIMD (measured: 1.13 us of a 3.3 us phase was this loop).
template <int NR, int NQ, int KB, int NW>
__device__ __forceinline__ void gemv_span(f32x4 (&part)[NR], const float* __restrict__ src, int row_stride,
                                          const float (&w)[NW], int w0) {
    // k-blocks per chunk: CH * NR fragments (4 registers each) in flight.  The weights of a VALU GEMV must sit in
    // ARCHITECTURAL registers (200 of the 256): a larger chunk pushes some of them into the accumulator file and every
    // phase pays a v_accvgpr_read per weight
    constexpr int CH = (NR == 1) ? 8 : ((NR == 2) ? 4 : 2);
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += CH) {
        f32x4 a[CH][NR];
#pragma unroll
        for (int q = 0; q < CH; ++q)
#pragma unroll
            for (int m = 0; m < NR; ++m)
                if (q0 + q < NQ) a[q][m] = *reinterpret_cast<const f32x4*>(src + m * row_stride + KB * (q0 + q));
        // four independent accumulation chains per row (one per k of a quad): a single chain of 200+ dependent FMAs runs at
        // the FMA's dependent latency, not at its issue rate
#pragma unroll
        for (int q = 0; q < CH; ++q)
#pragma unroll
            for (int m = 0; m < NR; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (q0 + q < NQ) part[m][j] = fmaf(a[q][m][j], w[w0 + 4 * (q0 + q) + j], part[m][j]);
    }
}

// The same in two halves for the one-row instantiations: every fragment of a span requested at once (NQ ds_read_b128 in
// flight), multiplied later -- the caller orders requests and multiplications of different spans by hand (left alone the
// scheduler keeps two fragments in flight and pays the LDS latency per pair)
template <int NQ, int KB>
__device__ __forceinline__ void span_load(f32x4 (&a)[NQ], const float* __restrict__ src) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = *reinterpret_cast<const f32x4*>(src + KB * q);
}
template <int NQ, int NW>
__device__ __forceinline__ void span_fma(f32x4& part, const f32x4 (&a)[NQ], const float (&w)[NW], int w0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[j] = fmaf(a[q][j], w[w0 + 4 * q + j], part[j]);
}

// Diagnostic build (make diag: -DAPE_CLUSTER_STAMPS): shader-cycle sums per part of member 0's wave 0, written to the
// model's debug words (memory nothing else reads).  The shipped library has none of this code.
#ifdef APE_CLUSTER_STAMPS
#define SM_STAMP(k)                                                       \
    do {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
        st_acc[k] += now_ - st_t0;                                        \
        st_t0 = now_;                                                     \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
#else
#define SM_STAMP(k) do {} while (0)
#endif

}  // namespace
