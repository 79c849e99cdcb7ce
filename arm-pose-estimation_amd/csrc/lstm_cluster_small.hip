// Small-batch (1..4 windows) variant of the weight-stationary cluster LSTM kernel: the latency path
// (BASELINE.json configs[1]: batch=1 streaming at 50 Hz).
//
// At 1-4 windows a 16-row MFMA tile is 75-94 % padding, so this variant keeps everything else of
// lstm_cluster.hip (GH = H/16 workgroups per cluster, member m owns hidden units [16m,16m+16) of every layer,
// its weights resident in registers for the whole launch in the same fragment layout, h slices exchanged with
// sc1 write-through stores + epoch flags, self-cleaning, arrival tickets) but
//   * the stacked-gate product is a register-resident GEMV on the VALU: lane (column c = unit*4+gate, k-group
//     g) multiplies its 4 weights of every 16-deep k-block with the matching activations (one broadcast
//     ds_read_b128 per block and row) and the four k-groups are summed with two wave shuffles;
//   * lane group g then owns batch row g: one activation per lane, the four gates of a unit meet by DPP
//     quad broadcasts, the cell update is a handful of VALU ops;
//   * ALL layers of a phase are computed back to back and published together, so there is ONE exchange
//     round trip per phase (layer l works on step p - l, as in the big kernel) instead of one per layer.
// Same arithmetic as the other kernels up to float32 summation order.
#include "ape_internal.h"
#include "../../include/ape_hip.h"

namespace {

constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr int MR = 4;                      // rows per cluster (row = lane group)

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

// part[m] += sum over this lane's k-slice of w * act[m]; NQ 16-deep blocks starting at weight register w0
template <int NR, int NQ, int NW>
__device__ __forceinline__ void gemv_span(float (&part)[NR], const float* __restrict__ src, int row_stride,
                                          const float (&w)[NW], int w0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + m * row_stride + 16 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) part[m] = fmaf(a[j], w[w0 + 4 * q + j], part[m]);
        }
    }
}

template <int H, int L, int KX, int NR>
__global__ __launch_bounds__(256, 1) void ape_lstm_cluster_small(const ClusterParams p) {
    constexpr int GH = H / 16;
    constexpr int SH = H + 8, SX = KX + 8, SO = 20;
    constexpr int QX = KX / 16, QH = H / 16;
    constexpr int NW0 = (KX + H) / 4, NW1 = (2 * H) / 4;
    constexpr int PIECES = GH * MR * 4;            // 16-byte pieces of one layer's gathered slices (<= 256)
    static_assert(PIECES <= 256 && NR <= MR, "one gather piece per thread");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;        // column = unit*4 + gate; k-group, later batch row
    const int gate = c & 3, u = c >> 2;
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool bcast_x = (p.flags & APE_FLAG_BROADCAST_X) != 0;   // monte_carlo_predictions: one window, B rows

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hbuf = smem;                            // [L][MR][SH]
    float* xin = hbuf + L * MR * SH;               // [2][MR][SX]  double-buffered by step parity
    float* own = xin + 2 * MR * SX;                // [L][MR][SO]
    int* ctl = reinterpret_cast<int*>(own + L * MR * SO);
    // Membership is fixed by the block index: the launch has 8 x GH workgroups and only every eighth one takes part
    // (the others leave at once).  Under the placement observed on this hardware -- blocks are dealt round-robin over
    // the 8 XCDs -- those GH workgroups share ONE XCD, hence one L2, and the exchange can stay inside it: plain stores
    // (acknowledged by the L2, not written through to memory) and L1-bypassing loads, ~0.8 us less per phase.  HIP
    // guarantees no placement, so nothing is assumed: every member publishes the XCD it really runs on (a non-zero
    // word; the last member out zeroes the words again, like the flags), all members read all GH words once the
    // weights are in, and only if they agree is the in-L2 form used; otherwise the write-through form that is valid for any placement.  One cluster, one launch: a member
    // that is dispatched late delays the launch, it cannot deadlock it.
    if (blockIdx.x % 8 != 0) return;
    const int cluster = 0, member = blockIdx.x / 8;
    const int row0 = 0;
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc &= 0xFu;
    // a launch that finds the sticky status word set (an earlier launch on this model aborted and left stale epochs
    // behind) leaves without touching anything; ape_model_check reports and resets
    if (tid == 0) {
        ctl[0] = (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
        if (ctl[0] == 0)
            __hip_atomic_store(p.xcc_slots + member, 0x10u | my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = tid; i < L * MR * SH; i += 256) hbuf[i] = 0.0f;       // h_{-1} = 0: the first step reads zeros
    __syncthreads();
    if (ctl[0] != 0) return;

    // ---- weights: registers for the whole launch (same fragment layout as the MFMA cluster kernel) ----------------
    static_assert(NW0 % 4 == 0 && NW1 % 4 == 0, "weight registers are loaded four at a time");
    float w0[NW0];
    float w1[L > 1 ? NW1 : 1];
    float w2[L > 2 ? NW1 : 1];
    {
        // 16 bytes per lane and load (host order [k-quad][lane][4]): a quarter of the instructions of a dword walk
        const f32x4* s0 = reinterpret_cast<const f32x4*>(p.wcl[0]) + ((size_t)(member * 4 + wave) * (NW0 / 4)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NW0 / 4; ++i) {
            const f32x4 v = s0[i * 64];
            w0[4 * i] = v[0]; w0[4 * i + 1] = v[1]; w0[4 * i + 2] = v[2]; w0[4 * i + 3] = v[3];
        }
        if constexpr (L > 1) {
            const f32x4* s1 = reinterpret_cast<const f32x4*>(p.wcl[1]) + ((size_t)(member * 4 + wave) * (NW1 / 4)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NW1 / 4; ++i) {
                const f32x4 v = s1[i * 64];
                w1[4 * i] = v[0]; w1[4 * i + 1] = v[1]; w1[4 * i + 2] = v[2]; w1[4 * i + 3] = v[3];
            }
        }
        if constexpr (L > 2) {
            const f32x4* s2 = reinterpret_cast<const f32x4*>(p.wcl[2]) + ((size_t)(member * 4 + wave) * (NW1 / 4)) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NW1 / 4; ++i) {
                const f32x4 v = s2[i * 64];
                w2[4 * i] = v[0]; w2[4 * i + 1] = v[1]; w2[4 * i + 2] = v[2]; w2[4 * i + 3] = v[3];
            }
        }
    }
    float bias_r[L], cst[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        bias_r[l] = p.bias[l][gate * H + member * 16 + wave * 4 + u];
        cst[l] = 0.0f;
    }

    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    unsigned* const myflag = p.xflags + (size_t)cluster * L * GH;       // one epoch word per member (all layers at once)
    constexpr unsigned SET_BYTES = GH * MR * 16 * sizeof(float);        // one (layer, parity)
    auto hx_base = [&](int l, int par) -> unsigned { return (unsigned)((((size_t)cluster * L + l) * 2 + par) * SET_BYTES); };

    auto stage_x = [&](int t) {                    // x_t: f64 z-score, cast f32 (estimator.py:103-104)
        if (tid < MR * KX) {
            const int row = tid / KX, k = tid - row * KX;
            const int b = row0 + row;
            float v = 0.0f;
            if (k < I && b < p.B) {
                v = p.x[((size_t)(bcast_x ? 0 : b) * T + (t + p.x_ring >= T ? t + p.x_ring - T : t + p.x_ring)) * I + k];
                if (normalize) v = (float)(((double)v - p.xx_m[k]) / p.xx_s[k]);
            }
            xin[((t & 1) * MR + row) * SX + k] = v;
        }
    };
    stage_x(0);
    // ---- do all members really share an XCD?  (their words have been on the way since the weights were requested)
    if (wave == 0) {
        unsigned spins = 0, v = 0u;
        while (true) {
            v = 0x10u | my_xcc;
            if (lane < GH) v = __hip_atomic_load(p.xcc_slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(v != 0u))) break;
            if (++spins > SPIN_LIMIT || __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                if (lane == 0) {
                    ctl[0] = 1;
                    __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        const int same = __all((int)((v & 0xFu) == my_xcc));
        if (lane == 0) ctl[3] = same;
    }
    __syncthreads();
    if (ctl[0] != 0) return;
    // uniform over the cluster: every member read the same GH words
    const bool in_l2 = ctl[3] != 0 && (p.flags & APE_DIAG_WRITE_THROUGH) == 0;

    const int P = T + L - 1;
#pragma unroll 1
    for (int ph = 0; ph < P; ++ph) {
        // ---- every layer of this phase, back to back (layer l works on step t = ph - l) ----------------------------
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int t = ph - l;
            if (t < 0 || t >= T) continue;         // uniform
            float part[NR];
#pragma unroll
            for (int m = 0; m < NR; ++m) part[m] = 0.0f;
            const float* rec_src = hbuf + l * MR * SH + 4 * g;
            if (l == 0) {
                gemv_span<NR, QX, NW0>(part, xin + (t & 1) * MR * SX + 4 * g, SX, w0, 0);
                gemv_span<NR, QH, NW0>(part, rec_src, SH, w0, 4 * QX);
            } else {
                const float* in_src = hbuf + (l - 1) * MR * SH + 4 * g;
                if (l == 1) {
                    if constexpr (L > 1) {
                        gemv_span<NR, QH, NW1>(part, in_src, SH, w1, 0);
                        gemv_span<NR, QH, NW1>(part, rec_src, SH, w1, 4 * QH);
                    }
                } else {
                    if constexpr (L > 2) {
                        gemv_span<NR, QH, NW1>(part, in_src, SH, w2, 0);
                        gemv_span<NR, QH, NW1>(part, rec_src, SH, w2, 4 * QH);
                    }
                }
            }
            // sum the four k-groups (lanes c, c+16, c+32, c+48): two wave shuffles per row
            float pre = 0.0f;
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                float v = part[m];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (g == m) pre = v;               // lane group g owns batch row g from here on
            }
            const float a = gate_act(pre + bias_r[l], gate == 2);
            const float iv = quad_bcast<0>(a), fv = quad_bcast<1>(a), gv = quad_bcast<2>(a), ov = quad_bcast<3>(a);
            const float cn = fv * cst[l] + iv * gv;
            cst[l] = cn;
            if (gate == 0 && g < NR) own[(l * MR + g) * SO + wave * 4 + u] = ov * gate_act(cn, true);
        }
        if (ph + 1 < T) stage_x(ph + 1);           // into the other xin buffer (its readers finished a phase ago)
        __syncthreads();                            // own slices of every active layer complete
        // ---- publish all active layers' slices, drain, barrier, ONE flag ----------------------------------------------
        {
            const int l = tid / (MR * 4), idx = tid - l * (MR * 4);      // 16 pieces per layer slice
            const int t = ph - l;
            if (l < L && t >= 0 && t < T) {
                const int row = idx >> 2, quad = idx & 3;
                const f32x4 hv = *reinterpret_cast<const f32x4*>(own + (l * MR + row) * SO + 4 * quad);
                const auto hvu = __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, hv);
                const unsigned off = hx_base(l, t & 1) + (unsigned)(((member * MR + row) * 16 + 4 * quad) * sizeof(float));
                if (in_l2) __builtin_amdgcn_raw_buffer_store_b128(hvu, hx_rsrc, off, 0, 0);          // stays in the XCD's L2
                else __builtin_amdgcn_raw_buffer_store_b128(hvu, hx_rsrc, off, 0, 16 /* sc1: write-through */);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (in_l2) *reinterpret_cast<volatile unsigned*>(myflag + member) = (unsigned)(ph + 1);      // plain: in L2
            else __hip_atomic_store(myflag + member, (unsigned)(ph + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- wait for every member's phase-ph flag, gather all layers' slices (one piece per thread and layer) -------
        if (wave == 0) {
            unsigned spins = 0;
            while (true) {
                unsigned v = (unsigned)(ph + 1);
                if (lane < GH) v = __hip_atomic_load(myflag + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((int)(v >= (unsigned)(ph + 1)))) break;
                if (++spins > SPIN_LIMIT || __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    if (lane == 0) {
                        ctl[0] = 1;
                        __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (ctl[0] != 0) return;
        if (tid < PIECES) {
            const int m = tid / (MR * 4), idx = tid - m * (MR * 4);
            const int row = idx >> 2, quad = idx & 3;
            f32x4 gv[L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = ph - l;
                if (t >= 0 && t < T)
                    gv[l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        hx_rsrc, hx_base(l, t & 1) + (unsigned)(((m * MR + row) * 16 + 4 * quad) * sizeof(float)), 0, 16 /* sc1 */));
            }
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = ph - l;
                if (t >= 0 && t < T) *reinterpret_cast<f32x4*>(hbuf + (l * MR + row) * SH + m * 16 + 4 * quad) = gv[l];
            }
        }
        __syncthreads();
    }

    // ---- head: member m finishes row m (rows < NR <= 4 <= GH) ------------------------------------------------------------
    if (member < MR && tid < O) {
        const int b = row0 + member;
        if (b < p.B) {
            const float* hv = hbuf + ((L - 1) * MR + member) * SH;
            const float* wv = p.w_out + (size_t)tid * H;
            float s = 0.0f;
            for (int k = 0; k < H; ++k) s = fmaf(hv[k], wv[k], s);
            p.y[(size_t)b * O + tid] = s + p.b_out[tid];
        }
    }
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GH - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        const int n_words = L * GH;
        for (int i = tid; i < n_words; i += 256)
            __hip_atomic_store(p.xflags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < GH) __hip_atomic_store(p.xcc_slots + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int H, int L, int KX, int NR>
hipError_t launch_small(const ClusterParams& p, hipStream_t stream) {
    constexpr size_t smem = ((size_t)L * MR * (H + 8) + 2 * MR * (KX + 8) + (size_t)L * MR * 20 + 4) * sizeof(float);
    hipLaunchKernelGGL((ape_lstm_cluster_small<H, L, KX, NR>), dim3(8 * (H / 16)), dim3(256), smem, stream, p);
    return hipGetLastError();
}

}  // namespace

// one cluster, B <= 4 windows (nr = 1, 2 or 4 rows computed)
hipError_t ape_launch_lstm_cluster_small(int H, int L, int KX, int nr, const ClusterParams& p, hipStream_t stream) {
    if (H == 256 && L == 2 && KX == 32) {
        if (nr == 1) return launch_small<256, 2, 32, 1>(p, stream);
        if (nr == 2) return launch_small<256, 2, 32, 2>(p, stream);
        if (nr == 4) return launch_small<256, 2, 32, 4>(p, stream);
    } else if (H == 128 && L == 3 && KX == 64) {
        if (nr == 1) return launch_small<128, 3, 64, 1>(p, stream);
        if (nr == 2) return launch_small<128, 3, 64, 2>(p, stream);
        if (nr == 4) return launch_small<128, 3, 64, 4>(p, stream);
    }
    return hipErrorInvalidValue;
}
