// Device helpers shared by the f32 cluster kernels (lstm_cluster.hip, lstm_cluster_duo.hip): the AGPR-pinned
// v_mfma_f32_16x16x4_f32 wrapper, the double-buffered layer-step MFMA loop with its per-k-block hook, the gate
// non-linearity.  Included inside each translation unit's anonymous namespace.
#pragma once

// Diagnostic build (make diag: -DAPE_CLUSTER_STAMPS): per-section shader-cycle sums of workgroup 0,
// wave 0, written behind the status word (memory nothing else reads).  The shipped library has none
// of this code.  Shares, not absolute time, are what such a build is for.
#ifdef APE_CLUSTER_STAMPS
#define STAMP_DECL unsigned long long st_t0 = 0, st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP_BEGIN()                                        \
    do {                                                     \
        __builtin_amdgcn_sched_barrier(0);                   \
        st_t0 = __builtin_amdgcn_s_memtime();                \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)
#define STAMP_END(k)                                         \
    do {                                                     \
        __builtin_amdgcn_sched_barrier(0);                   \
        const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(); \
        st_acc[k] += st_t1 - st_t0;                          \
        st_t0 = st_t1;                                       \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)
#else
#define STAMP_DECL
#define STAMP_BEGIN() do {} while (0)
#define STAMP_END(k) do {} while (0)
#endif


constexpr unsigned SPIN_LIMIT = 1u << 22;   // bounded polls (~seconds) before giving up

__device__ __forceinline__ float gate_act(float v, bool is_tanh) {
    // sigmoid(v), or tanh(v) = 2*sigmoid(2v) - 1 on the g-gate lanes: one branch-free formula so
    // all 64 lanes (four different gates per 16-lane row) stay converged
    // hardware v_exp_f32 (2^x) and v_rcp_f32, both ~1 ulp: absolute error of the activation ~1e-7
    const float e = __builtin_amdgcn_exp2f((is_tanh ? -2.885390081777927f : -1.4426950408889634f) * v);
    const float s = __builtin_amdgcn_rcpf(1.0f + e);
    return is_tanh ? 2.0f * s - 1.0f : s;
}

// value of the lane D columns up inside the same 16-lane row (wrapping): DPP row rotate right by 16-D
// (row_ror:n -- lane i reads lane (i - n) mod 16), no LDS round trip
template <int D>
__device__ __forceinline__ float row_rot_up(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (16 - D), 0xF, 0xF, false));
}

// v_mfma_f32_16x16x4_f32 with the B operand (a weight that lives in an AGPR for the whole launch) and the
// accumulator pinned to the accumulator register file: no v_accvgpr copies around the matrix pipe, and the
// 256 architectural VGPRs stay free for activations, gathers and the cell update.  hipcc does not model an
// asm MFMA's hazards: the accumulators are read only after `mfma_drain()` (>= 18 wait states after the
// last 8-pass MFMA, CDNA4 ISA data-hazard table).
// Operand roles: A = weights (row i = lane&15 = this wave's gate column  unit*4 + gate), B = activations
// (column j = lane&15 = batch row).  The result tile D[gate column][batch row] then puts, on every lane, the
// four gates i,f,g,o of ONE unit (lane>>4) for ONE batch row (lane&15) into its four accumulator registers:
// the cell update needs no cross-lane traffic at all.
// ACC_V: the accumulator in the ARCHITECTURAL file instead -- for an instantiation whose weights alone fill the 256
// accumulator registers (ImuPoseLSTM's 128 + 128): with "+a" the compiler would have to rotate weights through
// v_accvgpr_write in front of an MFMA whose hazards it does not model.
template <bool ACC_V = false>
__device__ __forceinline__ void mfma_aw(f32x4& acc, float a, float w) {
    if constexpr (ACC_V) asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %1, %0" : "+v"(acc) : "v"(a), "a"(w));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %1, %0" : "+a"(acc) : "v"(a), "a"(w));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }

// One layer-step of MFMAs: acc[mt] += [in | rec] activations (LDS) x this wave's weight registers.
// k-blocks 0..QIN-1 read `in_src`, QIN..QTOT-1 read `rec_src` (skipped when !do_rec: h_{-1} = 0).
// The A fragments of block q+1 are fetched BEFORE the 4*NMT MFMAs of block q (explicit double buffer,
// pinned with sched_barrier), so the matrix pipe never waits on a just-issued ds_read.
// `hook(q)` runs in front of k-block q -- before the A fragments of block q+1 are fetched, so a barrier placed in
// hook(QIN-1) precedes every read of `rec_src` -- and q is a constant after unrolling: the caller uses it to put
// exchange traffic, LDS commits and x staging under this section's matrix work.
template <int NMT, int QIN, int QTOT, int NW, bool ACC_V = false, typename Hook>
__device__ __forceinline__ void layer_mfma(f32x4 (&acc)[NMT], const float* __restrict__ in_src, int in_stride,
                                           const float* __restrict__ rec_src, int rec_stride,
                                           const float (&w)[NW], bool do_rec, Hook&& hook) {
    f32x4 a_cur[NMT], a_nxt[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        a_cur[mt] = *reinterpret_cast<const f32x4*>(in_src + mt * 16 * in_stride);
        a_nxt[mt] = a_cur[mt];
    }
    if constexpr (ACC_V) asm volatile("s_nop 4" ::: "memory");     // the start values were just written by the VALU
    // input span (both spans fully unrolled: every weight-register index is a compile-time constant)
#pragma unroll
    for (int q = 0; q < QIN; ++q) {
        hook(q);
        if (q + 1 < QIN) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
                a_nxt[mt] = *reinterpret_cast<const f32x4*>(in_src + mt * 16 * in_stride + 16 * (q + 1));
        } else if (do_rec) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
                a_nxt[mt] = *reinterpret_cast<const f32x4*>(rec_src + mt * 16 * rec_stride);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
                mfma_aw<ACC_V>(acc[mt], a_cur[mt][j], w[4 * q + j]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) a_cur[mt] = a_nxt[mt];
    }
    if (do_rec) {                                            // recurrent span (uniform)
#pragma unroll
        for (int q = QIN; q < QTOT; ++q) {
            hook(q);
            if (q + 1 < QTOT) {
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
                    a_nxt[mt] = *reinterpret_cast<const f32x4*>(rec_src + mt * 16 * rec_stride + 16 * (q + 1 - QIN));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
                    mfma_aw<ACC_V>(acc[mt], a_cur[mt][j], w[4 * q + j]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) a_cur[mt] = a_nxt[mt];
        }
    }
}

