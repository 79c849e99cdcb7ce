// Two-role variant of the weight-stationary cluster LSTM kernel (f32, big batches: 64 windows per cluster).
//
// Same decomposition, arithmetic, exchange buffers and hand-off protocol as lstm_cluster.hip (see there) -- GH = H/16
// workgroups per cluster, member m owns hidden units [16m,16m+16) of every layer, weights resident in AGPRs, slices
// exchanged with sc1 write-through stores + per-wave epoch flags, layers software-pipelined, ticketed clusters,
// self-cleaning flags -- but the workgroup has EIGHT waves, two per SIMD, with different jobs:
//
//   * waves 0-3, the MATRIX waves (one per SIMD, wave w owns units 4w..4w+3 as before), do nothing but the MFMAs:
//     wait until the LDS holds the section's inputs, run the input and recurrent spans, hand the 16 accumulator
//     registers to their partner through LDS, go on with the next section;
//   * waves 4-7, the HELPER waves (wave 4+w shares a SIMD with matrix wave w), do everything else for the same
//     units: gate non-linearities and cell update, publishing the slice, draining and raising the flag, polling
//     the peers' flags, gathering their slices, committing them to LDS once the last reader is through, the f64
//     z-score of the next input row.
//
// Why: a measurement (tools/ubench/mfma_valu_coissue.hip) shows that VALU work of ANOTHER wave on the SIMD does not
// slow an MFMA wave down at all (32.4 cycles per v_mfma_f32_16x16x4_f32 with or without a co-resident wave running
// transcendentals; that wave gets ~1/3 of its stand-alone rate), while the same instructions inside the MFMA wave's
// own stream cost 25+ cycles each.  In the one-role kernel everything but the MFMAs adds ~7K cycles to the 25.6K of
// a phase; here it runs beside them.
//
// Synchronisation inside the workgroup is by monotonic counters in LDS (no s_barrier in the loop -- a barrier would
// couple the two roles):
//   acc_seq[w]    matrix wave w has put the accumulators of its n-th section into accb[w]
//   acc_free[w]   its helper has taken them (accb[w] is single-buffered; the helper also stages its slice there)
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

// layer_mfma of lstm_cluster_common.h with the accumulators in VGPRs and the weights in VGPRs (WA = false) or AGPRs
template <int NMT, int QIN, int QTOT, int NW, bool WA, typename Hook>
__device__ __forceinline__ void layer_mfma_duo(f32x4 (&acc)[NMT], const float* __restrict__ in_src, int in_stride,
                                               const float* __restrict__ rec_src, int rec_stride,
                                               const float (&w)[NW], bool do_rec, Hook&& hook) {
    f32x4 a_cur[NMT], a_nxt[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        a_cur[mt] = *reinterpret_cast<const f32x4*>(in_src + mt * 16 * in_stride);
        a_nxt[mt] = a_cur[mt];
    }
    auto block = [&](int q) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                if constexpr (WA) mfma_va(acc[mt], a_cur[mt][j], w[4 * q + j]);
                else mfma_vv(acc[mt], a_cur[mt][j], w[4 * q + j]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) a_cur[mt] = a_nxt[mt];
    };
    // both spans fully unrolled: every weight-register index is a compile-time constant
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
        block(q);
    }
    if (do_rec) {
#pragma unroll
        for (int q = QIN; q < QTOT; ++q) {
            hook(q);
            if (q + 1 < QTOT) {
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
                    a_nxt[mt] = *reinterpret_cast<const f32x4*>(rec_src + mt * 16 * rec_stride + 16 * (q + 1 - QIN));
            }
            block(q);
        }
    }
}

template <int H, int L, int KX>
__global__ __launch_bounds__(512, 1) void ape_lstm_cluster_duo(const ClusterParams p) {
    constexpr int NMT = 4;
    constexpr int GH = H / 16;
    constexpr int MR = 16 * NMT;
    constexpr int SH = H + 8, SX = KX + 8, SO = 16;    // SO: slice staging row stride inside accb[w] (64 rows x 16 floats)
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
    float* accb = xin + MR * SX;                  // [4][MR*16]   accumulators matrix wave -> helper; then slice staging
    int* sync = reinterpret_cast<int*>(accb + 4 * MR * 16);
    int* acc_seq = sync, *acc_free = sync + 4, *m_in = sync + 8, *m_done = sync + 12;
    int* commit = sync + 16;                      // [L]
    int* x_cnt = sync + 16 + APE_MAX_LAYERS;
    int* ctl = x_cnt + 1;                         // [0] abort flag, [1] arrival ticket, [2] last-out
    if (tid < 32) sync[tid] = 0;
    if (tid == 0) ctl[1] = (int)__hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(ctl[1]);
    const int cluster = ticket / GH, member = ticket % GH;
    const int row0 = cluster * MR;

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
        float* const my_acc = accb + w * (MR * 16);
        int sidx = 0;
        STAMP_DECL
#pragma unroll 1
        for (int ph = 0; ph < P; ++ph) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = ph - l;
                if (t < 0 || t >= T) continue;
                STAMP_BEGIN();
                // inputs of this section in LDS?  x_t (layer 0), h^l_{t-1}, h^{l-1}_t
                bool ok = true;
                if (l == 0) ok = wait_ge(x_cnt, 4 * (t + 1));
                if (ok && t > 0) ok = wait_ge(commit + l, 4 * t);
                if (ok && l > 0) ok = wait_ge(commit + l - 1, 4 * (t + 1));
                if (!ok) goto done;
                STAMP_END(1);                            // 1: matrix wave waits for its inputs
                f32x4 acc[NMT];
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // the helper adds the bias
                const float* rec_src = hbuf + (l * MR + r) * SH + 4 * g;
                const int QIN = (l == 0) ? QX : QH;
                auto hook = [&](int q) {
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
                STAMP_END(2);                            // 2: MFMAs
                if (t == 0) lds_store(m_in + w, sidx + 1);
                if (!wait_ge(acc_free + w, sidx)) goto done;               // the helper took the last section's accumulators
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) *reinterpret_cast<f32x4*>(my_acc + (mt * 64 + lane) * 4) = acc[mt];
                lds_store(m_done + w, sidx + 1);
                lds_store(acc_seq + w, sidx + 1);
                STAMP_END(3);                            // 3: accumulator hand-off
                ++sidx;
            }
        }
#ifdef APE_CLUSTER_STAMPS
        if (blockIdx.x == 0 && tid == 0) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.status + 8);
            for (int k = 0; k < 4; ++k) dbg[k] = st_acc[k];
        }
#endif
    } else {
        // =========================== helper waves =========================================================================
        f32x4 bias_r[L];
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int k = 0; k < 4; ++k) bias_r[l][k] = p.bias[l][k * H + member * 16 + w * 4 + g];
        float cst[L][NMT];
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) cst[l][mt] = 0.0f;

        const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
        unsigned* const myflags = p.xflags + (size_t)cluster * L * NFL;
        constexpr unsigned SLICE_SET = GH * MR * 16 * sizeof(float);
        auto hx_base = [&](int l, int par) -> unsigned { return (unsigned)((((size_t)cluster * L + l) * 2 + par) * SLICE_SET); };
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
                __builtin_amdgcn_s_sleep(2);
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

        float* const my_acc = accb + w * (MR * 16);
        int sidx = 0;
        STAMP_DECL
#pragma unroll 1
        for (int ph = 0; ph < P; ++ph) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = ph - l;
                if (t < 0 || t >= T) continue;
                STAMP_BEGIN();
                // ---- this quad's accumulators -> gates, cell update, slice ---------------------------------------------------
                if (!wait_ge(acc_seq + w, sidx + 1)) goto done;
                STAMP_END(4);                            // 4: helper waits for the accumulators
                f32x4 acc[NMT];
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) acc[mt] = *reinterpret_cast<const f32x4*>(my_acc + (mt * 64 + lane) * 4);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    const float iv = gate_act(acc[mt][0] + bias_r[l][0], false), fv = gate_act(acc[mt][1] + bias_r[l][1], false);
                    const float gg = gate_act(acc[mt][2] + bias_r[l][2], true), ov = gate_act(acc[mt][3] + bias_r[l][3], false);
                    const float c = fv * cst[l][mt] + iv * gg;
                    cst[l][mt] = c;
                    my_acc[(16 * mt + r) * SO + g] = ov * gate_act(c, true);      // staging: row-major, this wave's 4 units
                }
                {   // publish: lane = row, one 16-byte piece (this quad's units) per row, write-through
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(my_acc + lane * SO);
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, hv), hx_rsrc,
                        (unsigned)(((member * MR + lane) * 16 + 4 * w) * sizeof(float)), hx_base(l, t & 1), 16 /* sc1 */);
                }
                lds_store(acc_free + w, sidx + 1);                       // accb[w] may take the next section's accumulators
                STAMP_END(5);                            // 5: gates + cell + staging + store issue
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // slice stores complete (and the x fetch, issued long ago)
                if (lane == 0)
                    __hip_atomic_store(myflags + l * NFL + member * 4 + w, (unsigned)(t + 1), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                STAMP_END(6);                            // 6: drain + flag
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
                STAMP_END(7);                            // 7: x staging (incl. waiting for the matrix waves)
                if (!wait_flags(l, (unsigned)(t + 1))) goto done;
                STAMP_END(8);                            // 8: waiting for the peers' flags
                f32x4 gv[NGV];
#pragma unroll
                for (int m = 0; m < NGV; ++m)
                    gv[m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        hx_rsrc, g_thread_off, hx_base(l, t & 1) + (unsigned)(m * MR * 16 * sizeof(float)), 16 /* sc1 */));
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
                STAMP_END(9);                            // 9: gather issue + waiting for the last reader
#pragma unroll
                for (int m = 0; m < NGV; ++m)
                    *reinterpret_cast<f32x4*>(hbuf + (l * MR + g_row) * SH + m * 16 + 4 * g_quad) = gv[m];
                lds_count_wave(commit + l);
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
    return ((size_t)L * 64 * (H + 8) + (size_t)64 * (KX + 8) + (size_t)4 * 64 * 16) * sizeof(float) + 32 * sizeof(int);
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
