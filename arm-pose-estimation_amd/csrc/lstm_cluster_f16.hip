// fp16 variant of the weight-stationary cluster LSTM kernel (BASELINE.json configs[4]: "fp16 hidden state
// with fp32 accumulate").
//
// Same decomposition and hand-off protocol as lstm_cluster.hip (see there): GH = H/16 workgroups
// per cluster, member m owns hidden units [16m,16m+16) of every layer, wave w one 16-column tile
// (column = unit*4 + gate; weights are the MFMA's A operand, so each lane gets i,f,g,o of one unit and batch
// row), weights resident in registers for the whole launch, h slices exchanged with sc1
// write-through stores + epoch flags, layers software-pipelined (phase p: layer l on step p - l), self-cleaning
// flags, ticketed clusters.
// What differs:
//   * the f16 MFMAs of a phase take ~1.6K cycles (the f32 kernel's: 25.6K), far too little to hide an exchange
//     behind, so the phase is SYNCHRONOUS like the small-batch kernel's: all active layers are computed back to
//     back from the LDS state, their slices published together under ONE flag per member, and one gather brings
//     in every layer's slices -- one fabric round trip per phase instead of one per layer-step;
//   * weights, the inputs x and the hidden state h are IEEE binary16; the stacked-gate product runs on
//     v_mfma_f32_16x16x32_f16 (K = 32 per instruction, 16 cycles: 16x the f32 MFMA rate) with f32
//     accumulators; gate pre-activations, the cell state c, the cell update and the linear head stay f32;
//   * half the registers (100 per lane for pocket/watch-only), half the LDS and half the exchange bytes.
// Reference semantics are unchanged (estimate/nn_models.py:169-174,180-189); only the storage precision of
// W, x and h is reduced, so parity is to a STATED tolerance (tests: <= 5e-3 abs on the NN targets).
#include "ape_internal.h"
#include "../../include/ape_hip.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ float gate_act(float v, bool is_tanh) {
    const float e = __builtin_amdgcn_exp2f((is_tanh ? -2.885390081777927f : -1.4426950408889634f) * v);
    const float s = __builtin_amdgcn_rcpf(1.0f + e);
    return is_tanh ? 2.0f * s - 1.0f : s;
}

template <int D>
__device__ __forceinline__ float row_rot_up(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (16 - D), 0xF, 0xF, false));
}

// acc[mt] += A (LDS, halves) x W (registers, halves) over NQ 32-deep k-blocks; A of block q+1 is fetched
// before the MFMAs of block q
template <int NMT, int NQ, int NW>
__device__ __forceinline__ void span_f16(f32x4 (&acc)[NMT], const _Float16* __restrict__ src, int row_stride,
                                         const half8 (&w)[NW], int w_off) {
    half8 a_cur[NMT], a_nxt[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        a_cur[mt] = *reinterpret_cast<const half8*>(src + mt * 16 * row_stride);
        a_nxt[mt] = a_cur[mt];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + 1 < NQ) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
                a_nxt[mt] = *reinterpret_cast<const half8*>(src + mt * 16 * row_stride + 32 * (q + 1));
        }
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[w_off + q], a_cur[mt], acc[mt], 0, 0, 0);   // A = weights, B = activations
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) a_cur[mt] = a_nxt[mt];
    }
}

template <int H, int L, int KX, int NMT>
__global__ __launch_bounds__(256, 1) void ape_lstm_cluster_f16(const ClusterParams p) {
    constexpr int GH = H / 16;
    constexpr int MR = 16 * NMT;
    constexpr int SH = H + 16;            // LDS row strides in HALVES (row = 16-byte multiple, conflict-free b128)
    constexpr int SX = KX + 16;
    constexpr int SO = 24;                // own-slice staging row stride (halves): 48 B rows
    constexpr int QX = KX / 32, QH = H / 32;
    constexpr int NB0 = QX + QH;          // 32-deep k-blocks of layer 0 / layers >= 1
    constexpr int NB1 = 2 * QH;
    constexpr int TPS = 2 * MR;           // 16-byte pieces per member slice (a row of 16 halves = 2 pieces)
    constexpr int SPP = 256 / TPS;        // slices per gather pass
    constexpr int NGV = (GH + SPP - 1) / SPP;
    static_assert(KX % 32 == 0 && H % 32 == 0, "fp16 variant needs 32-deep k-blocks");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool bcast_x = (p.flags & APE_FLAG_BROADCAST_X) != 0;   // monte_carlo_predictions: one window, B rows

    extern __shared__ __attribute__((aligned(16))) _Float16 smem16[];
    _Float16* hbuf = smem16;                          // [L][MR][SH]
    _Float16* xin = hbuf + L * MR * SH;               // [2][MR][SX]  double-buffered by step parity
    _Float16* own = xin + 2 * MR * SX;                // [L][MR][SO]
    int* ctl = reinterpret_cast<int*>(own + L * MR * SO); // [0] abort, [1] ticket, [2] last-out
    // A launch that finds the sticky status word set -- an earlier launch on this model aborted and skipped its
    // self-cleaning, so tickets, flags and counters are stale -- leaves without touching anything (status 2 tells
    // ape_model_check that later launches ran into it), and so does a workgroup whose ticket lies outside the grid.
    if (threadIdx.x == 0) {
        ctl[0] = 0;
        ctl[1] = -1;
        if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            const unsigned tk = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk < gridDim.x) ctl[1] = (int)tk;
            else __hip_atomic_store(p.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (ctl[1] < 0) return;
    const int ticket = __builtin_amdgcn_readfirstlane(ctl[1]);
    const int cluster = ticket / GH, member = ticket % GH;
    const int row0 = cluster * MR;

    // ---- weights: registers (halves), for the whole launch ---------------------------------------------
    half8 w0[NB0];
    half8 w1[L > 1 ? NB1 : 1];
    half8 w2[L > 2 ? NB1 : 1];
    {
        const half8* s0 = reinterpret_cast<const half8*>(p.wcl[0]) + ((size_t)(member * 4 + wave) * NB0) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NB0; ++i) w0[i] = s0[i * 64];
        if constexpr (L > 1) {
            const half8* s1 = reinterpret_cast<const half8*>(p.wcl[1]) + ((size_t)(member * 4 + wave) * NB1) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NB1; ++i) w1[i] = s1[i * 64];
        }
        if constexpr (L > 2) {
            const half8* s2 = reinterpret_cast<const half8*>(p.wcl[2]) + ((size_t)(member * 4 + wave) * NB1) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NB1; ++i) w2[i] = s2[i * 64];
        }
    }
    f32x4 bias_r[L];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int k = 0; k < 4; ++k) bias_r[l][k] = p.bias[l][k * H + member * 16 + wave * 4 + g];
    float cst[L][NMT];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) cst[l][mt] = 0.0f;

    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    unsigned* const myflags = p.xflags + (size_t)cluster * L * GH;      // word [member]: phases published (all layers at once)
    constexpr unsigned SLICE_SET = GH * MR * 16 * sizeof(_Float16);
    auto hx_base = [&](int l, int par) -> unsigned { return (unsigned)((((size_t)cluster * L + l) * 2 + par) * SLICE_SET); };

    const int g_sl = tid / TPS, g_idx = tid - g_sl * TPS;
    const int g_row = g_idx >> 1, g_hq = g_idx & 1;
    // every wave polls for itself: have all members published phase `want`?  bounded; on give-up raises the
    // sticky status word and the workgroup abort flag
    auto wait_flags = [&](unsigned want) {
        unsigned spins = 0;
        while (true) {
            unsigned v = want;
            if (lane < GH) v = __hip_atomic_load(myflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(v >= want))) return;
            if (++spins > SPIN_LIMIT || __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                if (lane == 0) {
                    ctl[0] = 1;
                    __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto issue_gather = [&](int l, int par, f32x4 (&gv)[NGV]) {
        const unsigned base = hx_base(l, par);
#pragma unroll
        for (int k = 0; k < NGV; ++k) {
            const int m = k * SPP + g_sl;
            if (m < GH)
                gv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    hx_rsrc, base + (unsigned)(((m * MR + g_row) * 16 + 8 * g_hq) * sizeof(_Float16)), 0, 16 /* sc1 */));
        }
    };
    auto commit_gather = [&](int l, const f32x4 (&gv)[NGV]) {
#pragma unroll
        for (int k = 0; k < NGV; ++k) {
            const int m = k * SPP + g_sl;
            if (m < GH) *reinterpret_cast<f32x4*>(hbuf + (l * MR + g_row) * SH + m * 16 + 8 * g_hq) = gv[k];
        }
    };
    // ---- x staging (f64 z-score, then binary16) -----------------------------------------------------------
    constexpr int NE = (MR * KX) / 256;
    const int xk = tid % KX;
    float xr[NE];
    auto fetch_x = [&](int t) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int b = row0 + (tid + 256 * e) / KX;
            xr[e] = (xk < I && b < p.B) ? p.x[((size_t)(bcast_x ? 0 : b) * T + (t + p.x_ring >= T ? t + p.x_ring - T : t + p.x_ring)) * I + xk] : 0.0f;
        }
    };
    auto stage_x = [&](int t) {
        const double x_mean = (normalize && xk < I) ? p.xx_m[xk] : 0.0;
        const double x_std = (normalize && xk < I) ? p.xx_s[xk] : 1.0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = (tid + 256 * e) / KX;
            float v = xr[e];
            if (normalize && xk < I && row0 + row < p.B) v = (float)(((double)v - x_mean) / x_std);
            xin[((t & 1) * MR + row) * SX + xk] = (_Float16)v;
        }
    };
    fetch_x(0);
    stage_x(0);
    if (T > 1) fetch_x(1);
    __syncthreads();

    const int P = T + L - 1;
#pragma unroll 1
    for (int ph = 0; ph < P; ++ph) {
        // x_{ph+1} into the other xin buffer (its readers finished a phase ago), x_{ph+2} into flight under the MFMAs and
        // the gate math, so that the store drain below does not wait on it
        if (ph + 1 < T) {
            stage_x(ph + 1);
            if (ph + 2 < T) fetch_x(ph + 2);
        }
        // ---- every active layer of this phase from the LDS state of the last one (layer l works on step ph - l) ----
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int t = ph - l;
            if (t < 0 || t >= T) continue;              // uniform over the grid
            f32x4 acc[NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) acc[mt] = bias_r[l];
            const _Float16* rec_src = hbuf + (l * MR + r) * SH + 8 * g;
            if (l == 0) {
                span_f16<NMT, QX, NB0>(acc, xin + ((t & 1) * MR + r) * SX + 8 * g, SX, w0, 0);
                if (t > 0) span_f16<NMT, QH, NB0>(acc, rec_src, SH, w0, QX);
            } else {
                const _Float16* in_src = hbuf + ((l - 1) * MR + r) * SH + 8 * g;
                if (l == 1) {
                    if constexpr (L > 1) {
                        span_f16<NMT, QH, NB1>(acc, in_src, SH, w1, 0);
                        if (t > 0) span_f16<NMT, QH, NB1>(acc, rec_src, SH, w1, QH);
                    }
                } else {
                    if constexpr (L > 2) {
                        span_f16<NMT, QH, NB1>(acc, in_src, SH, w2, 0);
                        if (t > 0) span_f16<NMT, QH, NB1>(acc, rec_src, SH, w2, QH);
                    }
                }
            }
            // gates + cell update, lane-local: registers 0..3 = i,f,g,o of (unit g, batch row 16*mt + r)
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const float iv = gate_act(acc[mt][0], false), fv = gate_act(acc[mt][1], false);
                const float gg = gate_act(acc[mt][2], true), ov = gate_act(acc[mt][3], false);
                const float c = fv * cst[l][mt] + iv * gg;               // cell state stays f32
                cst[l][mt] = c;
                own[(l * MR + 16 * mt + r) * SO + wave * 4 + g] = (_Float16)(ov * gate_act(c, true));
            }
        }
        __syncthreads();                                // own slices of every active layer complete
        // ---- publish all active layers' slices (16-byte write-through stores), drain, barrier, ONE flag --------------
        for (int idx = tid; idx < L * TPS; idx += 256) {
            const int l = idx / TPS, pc = idx - l * TPS;
            const int t = ph - l;
            if (t >= 0 && t < T) {
                const int row = pc >> 1, hq = pc & 1;
                const f32x4 hv = *reinterpret_cast<const f32x4*>(own + (l * MR + row) * SO + 8 * hq);
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, hv), hx_rsrc,
                    hx_base(l, ph & 1) + (unsigned)(((member * MR + row) * 16 + 8 * hq) * sizeof(_Float16)), 0, 16 /* sc1 */);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every storing wave drains before the flag (incl. the x fetch)
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(myflags + member, (unsigned)(ph + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- wait for every member's flag of this phase, gather every active layer's slices ------------------------------
        wait_flags((unsigned)(ph + 1));
        f32x4 gv[L][NGV];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int t = ph - l;
            if (t >= 0 && t < T) issue_gather(l, ph & 1, gv[l]);
        }
        // (no wave reads hbuf between the barrier above and the one below: the commit cannot race a reader)
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int t = ph - l;
            if (t >= 0 && t < T) commit_gather(l, gv[l]);
        }
        __syncthreads();                                // gathered h and x_{ph+1} visible
        if (ctl[0] != 0) return;
    }

    // ---- head (f32 weights, f16 h) --------------------------------------------------------------------------
    {
        constexpr int RPM = (MR + GH - 1) / GH;
        if (tid < RPM * O) {
            const int rr = tid / O, o = tid - rr * O;
            const int row = member * RPM + rr;
            const int b = row0 + row;
            if (row < MR && b < p.B) {
                const _Float16* hv = hbuf + ((L - 1) * MR + row) * SH;
                const float* wv = p.w_out + (size_t)o * H;
                float s = 0.0f;
                for (int k = 0; k < H; ++k) s = fmaf((float)hv[k], wv[k], s);
                p.y[(size_t)b * O + o] = s + p.b_out[o];
            }
        }
    }
    __syncthreads();
    if (tid == 0)
        ctl[2] = (__hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (ctl[2] != 0) {
        const int n_words = (int)(gridDim.x / GH) * L * GH;
        for (int i = tid; i < n_words; i += 256)
            __hip_atomic_store(p.xflags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int H, int L, int KX, int NMT>
size_t smem_bytes() {
    constexpr int MR = 16 * NMT;
    return ((size_t)L * MR * (H + 16) + (size_t)2 * MR * (KX + 16) + (size_t)L * MR * 24) * sizeof(_Float16) + 16;
}

template <int H, int L, int KX, int NMT>
hipError_t launch(const ClusterParams& p, int clusters, hipStream_t stream) {
    const size_t smem = smem_bytes<H, L, KX, NMT>();
    hipLaunchKernelGGL((ape_lstm_cluster_f16<H, L, KX, NMT>), dim3(clusters * (H / 16)), dim3(256), smem, stream, p);
    return hipGetLastError();
}

template <int H, int L, int KX, int NMT>
hipError_t prepare() {
    if (smem_bytes<H, L, KX, NMT>() > APE_LDS_BYTES) return hipErrorInvalidValue;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster_f16<H, L, KX, NMT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
}

}  // namespace

#define APE_CL16_DISPATCH(FN, ...)                                           \
    if (H == 256 && L == 2 && KX == 32) {                                    \
        if (nmt == 1) return FN<256, 2, 32, 1>(__VA_ARGS__);                 \
        if (nmt == 2) return FN<256, 2, 32, 2>(__VA_ARGS__);                 \
        if (nmt == 4) return FN<256, 2, 32, 4>(__VA_ARGS__);                 \
    } else if (H == 128 && L == 3 && KX == 64) {                             \
        if (nmt == 1) return FN<128, 3, 64, 1>(__VA_ARGS__);                 \
        if (nmt == 2) return FN<128, 3, 64, 2>(__VA_ARGS__);                 \
        if (nmt == 4) return FN<128, 3, 64, 4>(__VA_ARGS__);                 \
    }                                                                        \
    return hipErrorInvalidValue;

hipError_t ape_prepare_lstm_cluster_f16(int H, int L, int KX) {
    for (int nmt : {1, 2, 4}) {
        hipError_t e = [&]() -> hipError_t { APE_CL16_DISPATCH(prepare) }();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t ape_launch_lstm_cluster_f16(int H, int L, int KX, int nmt, int clusters, const ClusterParams& p,
                                       hipStream_t stream) {
    APE_CL16_DISPATCH(launch, p, clusters, stream)
}
