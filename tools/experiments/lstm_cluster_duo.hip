// Two-role variant of the weight-stationary cluster LSTM kernel (f32, big batches: 64 windows per cluster).
//
// Same decomposition, arithmetic, exchange buffers and hand-off protocol as lstm_cluster.hip (see there) -- GH = H/16
// workgroups per cluster, member m owns hidden units [16m,16m+16) of every layer, weights resident in AGPRs, slices
// exchanged with sc1 write-through stores + per-wave epoch flags, layers software-pipelined, ticketed clusters,
// self-cleaning flags -- but the workgroup has EIGHT waves, two per SIMD, with different jobs:
//
//   * waves 0-3, the MATRIX waves (one per SIMD, wave w owns units 4w..4w+3 as before), do the arithmetic: wait
//     until the LDS holds the section's inputs, run the input and recurrent spans, apply the gate non-linearities
//     and the cell update (lane-local, full VALU rate), leave their 4 x 64 fresh h values in a wave-private LDS
//     staging area, go on with the next section;
//   * waves 4-7, the HELPER waves (wave 4+w shares a SIMD with matrix wave w), do the exchange for the same units:
//     publishing the slice, draining and raising the flag, polling the peers' flags, gathering their slices,
//     committing them to LDS once the last reader is through, and the f64 z-score of the next input row -- few
//     instructions and long waits, which is what a wave beside a saturated matrix pipe is good for (it gets about
//     one VALU issue per MFMA: a first version that also gave it the gate math was helper-bound at 1.37 ms).
//
// Why: a measurement (tools/ubench/mfma_valu_coissue.hip) shows that VALU work of ANOTHER wave on the SIMD does not
// slow an MFMA wave down at all (32.4 cycles per v_mfma_f32_16x16x4_f32 with or without a co-resident wave running
// transcendentals; that wave gets ~1/3 of its stand-alone rate), while the same instructions inside the MFMA wave's
// own stream cost 25+ cycles each.  In the one-role kernel everything but the MFMAs adds ~7K cycles to the 25.6K of
// a phase; here it runs beside them.
//
// Synchronisation inside the workgroup is by monotonic counters in LDS (no s_barrier in the loop -- a barrier would
// couple the two roles):
//   m_in[w]       matrix wave w has finished the INPUT span of its n-th section (last read of the layer below)
//   m_done[w]     ... the whole n-th section (last read of its own recurrent buffer and of xin)
//   commit[l]     helper waves that have committed layer l's slices, summed over steps (4 per step)
//   x_cnt         helper waves that have staged x, summed over steps (4 per step)
// Sections are numbered in program order over the ACTIVE (phase, layer) pairs, identically in every wave.
#include "ape_internal.h"
#include "../../include/ape_hip.h"

namespace {

#include "lstm_cluster_common.h"

__device__ __forceinline__ int lds_load(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(int* p, int v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's LDS traffic before the signal is complete
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// one count per WAVE: every lane's LDS traffic is complete first, then lane 0 adds
__device__ __forceinline__ void lds_count_wave(int* p) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) (void)__hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Two waves per SIMD leave each wave 128 architectural + 128 accumulator registers (the compiler splits the 256 evenly
// once AGPRs are used), so the 216 registers of weights + accumulators of the one-role kernel do not fit the AGPR file:
// here layer 0's weights and the accumulators live in VGPRs, the upper layers' weights in AGPRs (128 for pocket).
__device__ __forceinline__ void mfma_vv(f32x4& acc, float a, float w) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %1, %0" : "+v"(acc) : "v"(a), "v"(w));
}
__device__ __forceinline__ void mfma_va(f32x4& acc, float a, float w) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %1, %0" : "+v"(acc) : "v"(a), "a"(w));
}

// One layer-step of MFMAs with the accumulators in VGPRs and the weights in VGPRs (WA = false) or AGPRs.  Only ONE set
// of activation fragments (16 registers) is kept: a k-block is worked off in two halves of 8 MFMAs, row tiles 0-1 then
// row tiles 2-3 (two accumulator chains 64 cycles apart: no dependent-issue stall), and the fragments of the finished
// pair are refilled IN PLACE for block q+1 (ds_read_b128, conflict-free as before) while the other pair computes -- the
// LDS latency hides behind 8 MFMAs (256 cycles) and the matrix wave fits 128 VGPRs next to 72 weight registers, 16
// accumulators and the cell state.
template <int NMT, int QIN, int QTOT, int NW, bool WA, typename Hook>
__device__ __forceinline__ void layer_mfma_duo(f32x4 (&acc)[NMT], const float* __restrict__ in_src, int in_stride,
                                               const float* __restrict__ rec_src, int rec_stride,
                                               const float (&w)[NW], bool do_rec, Hook&& hook) {
    static_assert(NMT == 4, "tile pairs");
    f32x4 a[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(in_src + mt * 16 * in_stride);
    auto half = [&](int q, int h2) {                      // the 8 MFMAs of block q on row tiles 2*h2, 2*h2+1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mt = 2 * h2; mt < 2 * h2 + 2; ++mt) {
                if constexpr (WA) mfma_va(acc[mt], a[mt][j], w[4 * q + j]);
                else mfma_vv(acc[mt], a[mt][j], w[4 * q + j]);
            }
        }
    };
    auto refill = [&](const float* src, int stride, int qs, int h2) {      // fragments of block qs for tiles 2*h2, 2*h2+1
#pragma unroll
        for (int mt = 2 * h2; mt < 2 * h2 + 2; ++mt)
            a[mt] = *reinterpret_cast<const f32x4*>(src + mt * 16 * stride + 16 * qs);
    };
    // both spans fully unrolled: every weight-register index is a compile-time constant
#pragma unroll
    for (int q = 0; q < QIN; ++q) {
        hook(q);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            __builtin_amdgcn_sched_barrier(0);
            half(q, h2);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < QIN) refill(in_src, in_stride, q + 1, h2);
            else if (do_rec) refill(rec_src, rec_stride, 0, h2);
        }
    }
    if (do_rec) {
#pragma unroll
        for (int q = QIN; q < QTOT; ++q) {
            hook(q);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                __builtin_amdgcn_sched_barrier(0);
                half(q, h2);
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 < QTOT) refill(rec_src, rec_stride, q + 1 - QIN, h2);
            }
        }
    }
}

#ifdef APE_CLUSTER_STAMPS
#define EV(cond, slot) do { if ((cond) && blockIdx.x == 0 && lane == 0 && p.dbg_wg) p.dbg_wg[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define EV(cond, slot) do {} while (0)
#endif

template <int H, int L, int KX>
__global__ __launch_bounds__(512, 1) void ape_lstm_cluster_duo(const ClusterParams p) {
    constexpr int NMT = 4;
    constexpr int GH = H / 16;
    constexpr int MR = 16 * NMT;
    constexpr int SH = H + 8, SX = KX + 8, SO = 16;    // SO: row stride of a matrix wave's slice staging area (64 rows x 16 floats, 4 used)
    constexpr int QX = KX / 16, QH = H / 16;
    constexpr int NW0 = (KX + H) / 4, NW1 = (2 * H) / 4;
    constexpr int NFL = 4 * GH;                   // flags per (cluster, layer): one per member helper wave
    constexpr int NGV = GH;                       // 16-byte pieces each helper thread moves per gather (256 threads)
    constexpr int NE = (MR * KX) / 256;           // x elements per helper thread and step
    constexpr int RPE = 256 / KX;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool helper = wave >= 4;
    const int w = wave & 3;                       // the unit quad this wave works for
    const int ht = tid & 255;                     // thread index within its role
    const int r = lane & 15, g = lane >> 4;
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool bcast_x = (p.flags & APE_FLAG_BROADCAST_X) != 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hbuf = smem;                           // [L][MR][SH]  gathered h of every layer
    float* xin = hbuf + L * MR * SH;              // [MR][SX]
    float* stage = xin + MR * SX;                 // [4][MR][SO]  fresh slice of each matrix wave (wave-private columns)
    float* bias_s = stage + 4 * MR * SO;          // [L][16 units][4 gates] of this member
    int* sync = reinterpret_cast<int*>(bias_s + APE_MAX_LAYERS * 64);
    int* m_in = sync + 8, *m_done = sync + 12;
    int* commit = sync + 16;                      // [L]
    int* x_cnt = sync + 16 + APE_MAX_LAYERS;
    int* ctl = x_cnt + 1;                         // [0] abort flag, [1] arrival ticket, [2] last-out
    if (tid < 32) sync[tid] = 0;
    if (tid == 0) ctl[1] = (int)__hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(ctl[1]);
    const int cluster = ticket / GH, member = ticket % GH;
    const int row0 = cluster * MR;
    if (tid < L * 64) {                           // bias: [l][unit][gate], a lane reads its four gates with one ds_read_b128
        const int l = tid / 64, u = (tid % 64) / 4, gt = tid % 4;
        bias_s[tid] = p.bias[l][gt * H + member * 16 + u];
    }
    __syncthreads();

    // bounded wait for an LDS counter; false on abort (a helper's poll of the peers expired, or this one did)
    auto wait_ge = [&](const int* ptr, int want) -> bool {
        unsigned spins = 0;
        while (lds_load(ptr) < want) {
            if (lds_load(ctl) != 0) return false;
            if (++spins > (SPIN_LIMIT << 2)) {
                if (lane == 0) {
                    ctl[0] = 1;
                    __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return false;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        return true;
    };
    const int P = T + L - 1;
    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    unsigned* const myflags = p.xflags + (size_t)cluster * L * NFL;
    constexpr unsigned SLICE_SET = GH * MR * 16 * sizeof(float);
    auto hx_base = [&](int l, int par) -> unsigned { return (unsigned)((((size_t)cluster * L + l) * 2 + par) * SLICE_SET); };

    if (!helper) {
        // =========================== matrix waves =========================================================================
        float w0[NW0];
        float w1[L > 1 ? NW1 : 1];
        float w2[L > 2 ? NW1 : 1];
        {
            const f32x4* s0 = reinterpret_cast<const f32x4*>(p.wcl[0]) + ((size_t)(member * 4 + w) * (NW0 / 4)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NW0 / 4; ++i) {
                const f32x4 v = s0[i * 64];
                w0[4 * i] = v[0]; w0[4 * i + 1] = v[1]; w0[4 * i + 2] = v[2]; w0[4 * i + 3] = v[3];
            }
            if constexpr (L > 1) {
                const f32x4* s1 = reinterpret_cast<const f32x4*>(p.wcl[1]) + ((size_t)(member * 4 + w) * (NW1 / 4)) * 64 + lane;
#pragma unroll
                for (int i = 0; i < NW1 / 4; ++i) {
                    const f32x4 v = s1[i * 64];
                    w1[4 * i] = v[0]; w1[4 * i + 1] = v[1]; w1[4 * i + 2] = v[2]; w1[4 * i + 3] = v[3];
                }
            }
            if constexpr (L > 2) {
                const f32x4* s2 = reinterpret_cast<const f32x4*>(p.wcl[2]) + ((size_t)(member * 4 + w) * (NW1 / 4)) * 64 + lane;
#pragma unroll
                for (int i = 0; i < NW1 / 4; ++i) {
                    const f32x4 v = s2[i * 64];
                    w2[4 * i] = v[0]; w2[4 * i + 1] = v[1]; w2[4 * i + 2] = v[2]; w2[4 * i + 3] = v[3];
                }
            }
        }
        float* const my_stage = stage + w * (MR * SO);
        float cst[L][NMT];
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) cst[l][mt] = 0.0f;
        int sidx = 0;
        // the flag owed for the slice stored at the end of the last section: raised, once those stores have drained, one
        // k-block into the next section (or at the very end)
        int pend_idx = -1;
        unsigned pend_epoch = 0u;
        auto raise_pending = [&]() {
            if (pend_idx < 0) return;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(myflags + pend_idx, pend_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pend_idx = -1;
        };
        STAMP_DECL
#pragma unroll 1
        for (int ph = 0; ph < P; ++ph) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = ph - l;
                if (t < 0 || t >= T) continue;
                STAMP_BEGIN();
                // inputs of the INPUT span in LDS?  x_t (layer 0) or h^{l-1}_t.  The recurrent input h^l_{t-1} is waited for
                // only in front of the recurrent span (hook below): its exchange, which started at the end of this
                // layer's previous section, then has that section's successor PLUS this input span to finish in
                // (a wave never blocks on the peers while it owes them a flag: with T = 1 the layer below finished in the
                //  section right before this one and its flag would otherwise go up only inside this section)
                if (l > 0 && lds_load(commit + l - 1) < 4 * (t + 1)) raise_pending();
                bool ok = true;
                if (l == 0) ok = wait_ge(x_cnt, 4 * (t + 1));
                if (ok && l > 0) ok = wait_ge(commit + l - 1, 4 * (t + 1));
                if (!ok) goto done;
                EV(w == 0 && sidx >= 40 && sidx < 44, (sidx - 40) * 10 + 0);      // section started (inputs of the input span there)
                STAMP_END(1);                            // 1: matrix wave waits for its inputs
                f32x4 acc[NMT];
                {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + (l * 16 + w * 4 + g) * 4);
#pragma unroll
                    for (int mt = 0; mt < NMT; ++mt) acc[mt] = bv;       // bias = initial accumulator, as everywhere
                }
                const float* rec_src = hbuf + (l * MR + r) * SH + 4 * g;
                const int QIN = (l == 0) ? QX : QH;
                auto hook = [&](int q) {
                    if (q == ((QIN - 1 < 6) ? QIN - 1 : 6)) {           // store acknowledged by now (~1.5 us): no stall in the drain;
                        raise_pending();                                  // and BEFORE the wait that depends on the peers
                        EV(w == 0 && sidx >= 41 && sidx < 45, (sidx - 41) * 10 + 3);   // flag of the previous section raised
                    }
                    if (q == QIN - 1 && t > 0) {
                        EV(w == 0 && sidx >= 40 && sidx < 44, (sidx - 40) * 10 + 8);   // reached the recurrent-input wait
                        (void)wait_ge(commit + l, 4 * t);                 // (an abort surfaces at the next checked wait)
                        EV(w == 0 && sidx >= 40 && sidx < 44, (sidx - 40) * 10 + 9);   // ... passed it
                    }
                    if (q == QIN) lds_store(m_in + w, sidx + 1);          // the layer below has no reader left in this wave
                };
                if (l == 0) {
                    layer_mfma_duo<NMT, QX, QX + QH, NW0, false>(acc, xin + r * SX + 4 * g, SX, rec_src, SH, w0, t > 0, hook);
                } else {
                    const float* in_src = hbuf + ((l - 1) * MR + r) * SH + 4 * g;
                    if (l == 1) {
                        if constexpr (L > 1) layer_mfma_duo<NMT, QH, 2 * QH, NW1, true>(acc, in_src, SH, rec_src, SH, w1, t > 0, hook);
                    } else {
                        if constexpr (L > 2) layer_mfma_duo<NMT, QH, 2 * QH, NW1, true>(acc, in_src, SH, rec_src, SH, w2, t > 0, hook);
                    }
                }
                mfma_drain();
                EV(w == 0 && sidx >= 40 && sidx < 44, (sidx - 40) * 10 + 1);      // MFMAs of the section done
                STAMP_END(2);                            // 2: MFMAs
                if (t == 0) lds_store(m_in + w, sidx + 1);
                lds_store(m_done + w, sidx + 1);                           // last LDS read of this section is behind us
                raise_pending();                                           // (a section too short to reach block 1)
                // gates + cell update, lane-local: registers 0..3 = i,f,g,o of (unit g, batch row 16*mt + r)
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    const float iv = gate_act(acc[mt][0], false), fv = gate_act(acc[mt][1], false);
                    const float gg = gate_act(acc[mt][2], true), ov = gate_act(acc[mt][3], false);
                    const float c = fv * cst[l][mt] + iv * gg;
                    cst[l][mt] = c;
                    my_stage[(16 * mt + r) * SO + g] = ov * gate_act(c, true);
                }
                {   // publish: lane = row, one 16-byte piece (this wave's four units) per row, write-through; the staging
                    // area is this wave's own, LDS operations of one wave are ordered
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(my_stage + lane * SO);
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, hv), hx_rsrc,
                        (unsigned)(((member * MR + lane) * 16 + 4 * w) * sizeof(float)), hx_base(l, t & 1), 16 /* sc1 */);
                    pend_idx = l * NFL + member * 4 + w;
                    pend_epoch = (unsigned)(t + 1);
                }
                EV(w == 0 && sidx >= 40 && sidx < 44, (sidx - 40) * 10 + 2);      // slice stored
                STAMP_END(3);                            // 3: gate math + publish
                ++sidx;
            }
        }
        raise_pending();
#ifdef APE_CLUSTER_STAMPS
        if (blockIdx.x == 0 && tid == 0) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.status + 8);
            for (int k = 0; k < 4; ++k) dbg[k] = st_acc[k];
        }
#endif
    } else {
        // =========================== helper waves =========================================================================
        // few instructions, all of them on somebody's critical path: let them win the issue arbitration against the
        // matrix wave of their SIMD (measured: without this a helper instruction waits ~90 cycles for a slot)
        __builtin_amdgcn_s_setprio(3);
        const int g_row = ht >> 2, g_quad = ht & 3;                          // gather: piece (row, quad) of every member
        const unsigned g_thread_off = (unsigned)((g_row * 16 + 4 * g_quad) * sizeof(float));

        // every helper wave polls for itself: all member helper waves published epoch `want` of layer l?
        auto wait_flags = [&](int l, unsigned want) -> bool {
            unsigned spins = 0;
            while (true) {
                unsigned v = want;
                if (lane < NFL) v = __hip_atomic_load(myflags + l * NFL + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((int)(v >= want))) return true;
                if (lds_load(ctl) != 0) return false;
                if (++spins > SPIN_LIMIT || __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    if (lane == 0) {
                        ctl[0] = 1;
                        __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return false;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        };

        // ---- x staging (as lstm_cluster.hip: buffer descriptor over the cluster's rows, f64 z-score) -------------------
        const int xk = ht % KX, xrow = ht / KX;
        const int rows_here = bcast_x ? MR : min(MR, p.B - row0);
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x) + (bcast_x ? (size_t)0 : (size_t)row0 * T * I), 0,
            (int)((size_t)(bcast_x ? 1 : rows_here) * T * I * sizeof(float)), 0x00020000);
        const unsigned x_off0 = (xk < I) ? (unsigned)(((bcast_x ? 0 : xrow) * T * I + xk) * sizeof(float)) : 0x80000000u;
        const unsigned x_estride = bcast_x ? 0u : (unsigned)(RPE * T * I * sizeof(float));
        float xr[NE];
        auto fetch_x = [&](int t) {
            const int slot = (t + p.x_ring >= T) ? t + p.x_ring - T : t + p.x_ring;
#pragma unroll
            for (int e = 0; e < NE; ++e)
                xr[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            x_rsrc, x_off0 + (unsigned)e * x_estride, (unsigned)(slot * I * sizeof(float)), 0));
        };
        const double x_mean = (normalize && xk < I) ? p.xx_m[xk] : 0.0;
        const double x_std = (normalize && xk < I) ? p.xx_s[xk] : 1.0;
        const double x_rstd = (normalize && xk < I) ? p.xx_r[xk] : 1.0;
        auto stage_x = [&]() {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const double d = (double)xr[e] - x_mean;
                const double q0 = d * x_rstd;
                const double rr = fma(-q0, x_std, d);
                const double q1 = fma(rr, x_rstd, q0);
                xin[(xrow + e * RPE) * SX + xk] = (float)((rr == rr) ? q1 : q0);
            }
        };
        fetch_x(0);
        stage_x();
        if (T > 1) fetch_x(1);
        lds_count_wave(x_cnt);

        int sidx = 0;
        STAMP_DECL
#pragma unroll 1
        for (int ph = 0; ph < P; ++ph) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = ph - l;
                if (t < 0 || t >= T) continue;
                STAMP_BEGIN();
                // ---- x_{t+1}: xin has no reader left once every matrix wave is through this layer-0 section ----------------------
                if (l == 0 && t + 1 < T) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) ok = ok && wait_ge(m_done + k, sidx + 1);
                    if (!ok) goto done;
                    stage_x();
                    lds_count_wave(x_cnt);
                    if (t + 2 < T) fetch_x(t + 2);
                }
                // ---- the peers' slices of this layer-step: gather, wait for the last reader of the old ones, commit ------------------
                if (!wait_flags(l, (unsigned)(t + 1))) goto done;
                EV(w == 0 && sidx >= 40 && sidx < 44, (sidx - 40) * 10 + 4);      // helper saw every flag of the section
                STAMP_END(8);                            // 8: waiting for the peers' flags
                f32x4 gv[NGV];
#pragma unroll
                for (int m = 0; m < NGV; ++m)
                    gv[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        hx_rsrc, g_thread_off, hx_base(l, t & 1) + (unsigned)(m * MR * 16 * sizeof(float)), 16 /* sc1 */));
                STAMP_END(7);                            // 7: gather issue + x staging
                {
                    // readers of hbuf[l] = h^l_{t-1}: this section (recurrent span) and, one section later in program order,
                    // layer l+1 on step t-1 (input span)
                    const bool next_reads = (l + 1 < L) && (t >= 1);
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        ok = ok && wait_ge(m_done + k, sidx + 1);
                        if (next_reads) ok = ok && wait_ge(m_in + k, sidx + 2);
                    }
                    if (!ok) goto done;
                }
                EV(w == 0 && sidx >= 40 && sidx < 44, (sidx - 40) * 10 + 5);      // gather issued, last reader through
                STAMP_END(9);                            // 9: gather issue + waiting for the last reader
#pragma unroll
                for (int m = 0; m < NGV; ++m)
                    *reinterpret_cast<f32x4*>(hbuf + (l * MR + g_row) * SH + m * 16 + 4 * g_quad) = gv[m];
                lds_count_wave(commit + l);
                EV(w == 0 && sidx >= 40 && sidx < 44, (sidx - 40) * 10 + 6);      // committed
                STAMP_END(10);                           // 10: commit
                ++sidx;
            }
        }
#ifdef APE_CLUSTER_STAMPS
        if (blockIdx.x == 0 && tid == 256) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.status + 8);
            for (int k = 4; k < 12; ++k) dbg[k] = st_acc[k];
        }
#endif
    }
done:
    __syncthreads();
    if (ctl[0] != 0) return;
    // ---- head: each member finishes RPM = MR/GH of the cluster's windows (rows of h^{L-1}_{T-1}, committed above) ------------
    {
        constexpr int RPM = (MR + GH - 1) / GH;
        const int n_out = RPM * O;
        constexpr int PL = (RPM * APE_MAX_OUTPUT * 4 <= 512) ? 4 : ((RPM * APE_MAX_OUTPUT * 2 <= 512) ? 2 : 1);
        const int oi = tid / PL, part = tid % PL;
        float s_acc = 0.0f;
        const bool live = oi < n_out;
        const int rr = live ? oi / O : 0, o = live ? oi - rr * O : 0;
        const int row = member * RPM + rr;
        if (live && row < MR) {
            const float* hv = hbuf + ((L - 1) * MR + row) * SH;
            const float* wv = p.w_out + (size_t)o * H;
            for (int k = 4 * part; k < H; k += 4 * PL) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(hv + k);
                const f32x4 ww = *reinterpret_cast<const f32x4*>(wv + k);
                s_acc = fmaf(a[0], ww[0], s_acc); s_acc = fmaf(a[1], ww[1], s_acc);
                s_acc = fmaf(a[2], ww[2], s_acc); s_acc = fmaf(a[3], ww[3], s_acc);
            }
        }
        if (PL >= 2) s_acc += __shfl_xor(s_acc, 1, 64);
        if (PL >= 4) s_acc += __shfl_xor(s_acc, 2, 64);
        const int b = row0 + row;
        if (live && part == 0 && row < MR && b < p.B) p.y[(size_t)b * O + o] = s_acc + p.b_out[o];
    }
    // ---- self-cleaning: the LAST workgroup out re-zeroes every polled word for the next launch -------------------------------
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        const int n_words = (int)(gridDim.x / GH) * L * NFL;
        for (int i = tid; i < n_words; i += 512)
            __hip_atomic_store(p.xflags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int H, int L, int KX>
size_t smem_bytes() {
    return ((size_t)L * 64 * (H + 8) + (size_t)64 * (KX + 8) + (size_t)4 * 64 * 16 + APE_MAX_LAYERS * 64) * sizeof(float) + 32 * sizeof(int);
}

template <int H, int L, int KX>
hipError_t launch(const ClusterParams& p, int clusters, hipStream_t stream) {
    const size_t smem = smem_bytes<H, L, KX>();
    hipLaunchKernelGGL((ape_lstm_cluster_duo<H, L, KX>), dim3(clusters * (H / 16)), dim3(512), smem, stream, p);
    return hipGetLastError();
}

template <int H, int L, int KX>
hipError_t prepare() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster_duo<H, L, KX>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes<H, L, KX>());
}

}  // namespace

hipError_t ape_prepare_lstm_cluster_duo(int H, int L, int KX) {
    if (H == 256 && L == 2 && KX == 32) return prepare<256, 2, 32>();
    if (H == 128 && L == 3 && KX == 64) return prepare<128, 3, 64>();
    return hipErrorInvalidValue;
}

hipError_t ape_launch_lstm_cluster_duo(int H, int L, int KX, int clusters, const ClusterParams& p, hipStream_t stream) {
    if (H == 256 && L == 2 && KX == 32) return launch<256, 2, 32>(p, clusters, stream);
    if (H == 128 && L == 3 && KX == 64) return launch<128, 3, 64>(p, clusters, stream);
    return hipErrorInvalidValue;
}
