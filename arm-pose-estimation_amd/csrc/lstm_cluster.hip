// Weight-stationary "cluster" LSTM kernel for gfx950 (MI355X).
//
// Same arithmetic as lstm_tile16.hip (reference estimate/nn_models.py:169-174,180-189: L stacked
// LSTM layers, gates i,f,g,o, zero initial state, Linear head on the last step) but mapped so that
// the 3.3 MB of weights never move after the prologue:
//
//   * GH = H/16 workgroups (one per CU) form a CLUSTER that owns MR = 16*NMT windows for all T
//     steps.  Member m owns hidden units [16m, 16m+16) of EVERY layer with all four gates; wave w
//     of it owns 4 of those units = one 16-column MFMA tile (column = gate*4 + unit).
//   * each wave keeps its slice of [W_ih | W_hh] of every layer in REGISTERS for the whole launch,
//     already in v_mfma_f32_16x16x4_f32 B-fragment order (200 VGPR/AGPR per lane for the pocket
//     model): the MFMA B operand needs no load at all.  At B=1024 all 256 CUs are busy
//     (16 clusters x 16 CUs) instead of 64, and no CU re-streams weights every step.
//   * the A operand (activations of all H units of the cluster's windows) lives in LDS; every
//     layer-step each member publishes its 16-unit slice of h to a small exchange buffer and
//     gathers the other members' slices.  Hand-off protocol (placement-independent, MI355X guide
//     G16 / visibility table row 1): payload by 16-byte sc1 (write-through) buffer stores, every
//     storing wave drains vmcnt(0), workgroup barrier, ONE lane stores the epoch flag with an
//     agent-scope relaxed atomic; the consumer's wave 0 polls the members' flags with agent-scope
//     relaxed loads, workgroup barrier, then EVERY load of the payload is a 16-byte sc1 buffer
//     load.  Exchange buffers are double-buffered by step parity; flags are zeroed by a
//     hipMemsetAsync node in front of every launch; every spin is bounded and raises a status word.
//   * the layers are software-pipelined: in phase p layer l works on step t = p - l, so all the
//     layer computations of a phase depend only on the previous phase and the exchange of one
//     layer's slice flies under the other layers' MFMAs.
//   * gate activations are evaluated where the accumulators are (every lane: its own gate's
//     column), then a 4x4 lane transpose (cross-lane shuffles inside 16-lane rows) gives each lane
//     i,f,g,o of ONE (unit, row-tile), so the c/h update is spread over all 64 lanes.
#include "ape_internal.h"
#include "../../include/ape_hip.h"

namespace {

constexpr unsigned SPIN_LIMIT = 1u << 22;   // bounded polls (~seconds) before giving up

__device__ __forceinline__ float gate_act(float v, bool is_tanh) {
    // sigmoid(v), or tanh(v) = 2*sigmoid(2v) - 1 on the g-gate lanes: one branch-free formula so
    // all 64 lanes (four different gates per 16-lane row) stay converged
    const float s = 1.0f / (1.0f + expf(is_tanh ? -2.0f * v : -v));
    return is_tanh ? 2.0f * s - 1.0f : s;
}

// acc[mt] += A[16 rows of tile mt][16 k-values of block q] * W, weights from registers
template <int NMT, int NQ, int NW>
__device__ __forceinline__ void mfma_span(f32x4 (&acc)[NMT], const float* __restrict__ src, int row_stride,
                                          const float (&w)[NW], int w0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        f32x4 a[NMT];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
            a[mt] = *reinterpret_cast<const f32x4*>(src + mt * 16 * row_stride + 16 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], w[w0 + 4 * q + j], acc[mt], 0, 0, 0);
        }
    }
}

template <int H, int L, int KX, int NMT>
__global__ __launch_bounds__(256, 1) void ape_lstm_cluster(const ClusterParams p) {
    constexpr int GH = H / 16;            // workgroups (CUs) per cluster
    constexpr int MR = 16 * NMT;          // windows per cluster
    constexpr int SH = H + 8;             // LDS row strides (floats): conflict-free ds_read_b128
    constexpr int SX = KX + 8;
    constexpr int SO = 20;                // own-slice staging row stride
    constexpr int QX = KX / 16, QH = H / 16;
    constexpr int NW0 = (KX + H) / 4;     // weight registers per lane, layer 0
    constexpr int NW1 = (2 * H) / 4;      //                            layers >= 1
    constexpr int TPS = 4 * MR;           // threads that move one member slice (16 B each)
    constexpr int SPP = 256 / TPS;        // slices per gather pass

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int gate = r >> 2, u = r & 3;           // C-operand column = gate*4 + unit
    const int cluster = blockIdx.x / GH, member = blockIdx.x % GH;
    const int row0 = cluster * MR;
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* hbuf = smem;                           // [L][MR][SH]  gathered h of every layer
    float* xin = hbuf + L * MR * SH;              // [MR][SX]
    float* own = xin + MR * SX;                   // [MR][SO]     this member's fresh slice
    int* ctl = reinterpret_cast<int*>(own + MR * SO);   // [0] abort flag

    // ---- weights: registers, for the whole launch ---------------------------------------------
    float w0[NW0];
    float w1[L > 1 ? NW1 : 1];
    float w2[L > 2 ? NW1 : 1];
    {
        const float* s0 = p.wcl[0] + ((size_t)(member * 4 + wave) * NW0) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NW0; ++i) w0[i] = s0[i * 64];
        if constexpr (L > 1) {
            const float* s1 = p.wcl[1] + ((size_t)(member * 4 + wave) * NW1) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NW1; ++i) w1[i] = s1[i * 64];
        }
        if constexpr (L > 2) {
            const float* s2 = p.wcl[2] + ((size_t)(member * 4 + wave) * NW1) * 64 + lane;
#pragma unroll
            for (int i = 0; i < NW1; ++i) w2[i] = s2[i * 64];
        }
    }
    float bias_r[L];
#pragma unroll
    for (int l = 0; l < L; ++l) bias_r[l] = p.bias[l][gate * H + member * 16 + wave * 4 + u];

    float cst[L][4];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int i = 0; i < 4; ++i) cst[l][i] = 0.0f;

    // exchange buffer descriptors (wave-uniform: kernel arguments only)
    const __amdgpu_buffer_rsrc_t hx_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)p.hx_bytes, 0x00020000);
    unsigned* const myflags = p.xflags + (size_t)cluster * L * GH;
    if (tid == 0) ctl[0] = 0;
    __syncthreads();

    // wait until every member of this cluster has published epoch `want` of layer l, then gather the
    // GH slices (parity `par`) into hbuf[l]
    auto gather = [&](int l, unsigned want, int par) -> bool {
        if (wave == 0) {
            unsigned spins = 0;
            bool bad = false;
            while (true) {
                unsigned v = want;
                if (lane < GH)
                    v = __hip_atomic_load(myflags + l * GH + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((int)(v >= want))) break;
                if (++spins > SPIN_LIMIT ||
                    __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    bad = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (bad && lane == 0) {
                ctl[0] = 1;
                __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        if (ctl[0] != 0) return false;
        const unsigned base = (unsigned)((((size_t)cluster * L + l) * 2 + par) * GH * MR * 16 * sizeof(float));
        const int sl = tid / TPS, idx = tid - sl * TPS;      // slice within the pass, 16-byte piece within the slice
        const int row = idx >> 2, quad = idx & 3;
        f32x4 v[GH / SPP];
#pragma unroll
        for (int ps = 0; ps < GH / SPP; ++ps) {
            const int m = ps * SPP + sl;
            v[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                hx_rsrc, base + (unsigned)(((m * MR + row) * 16 + 4 * quad) * sizeof(float)), 0, 16 /* sc1 */));
        }
#pragma unroll
        for (int ps = 0; ps < GH / SPP; ++ps) {
            const int m = ps * SPP + sl;
            *reinterpret_cast<f32x4*>(hbuf + (l * MR + row) * SH + m * 16 + 4 * quad) = v[ps];
        }
        __syncthreads();
        return true;
    };

    const int P = T + L - 1;
#pragma unroll 1
    for (int ph = 0; ph < P; ++ph) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int t = ph - l;                      // the step layer l works on in this phase
            // h^l_{t-1} (published one phase ago) feeds layer l at step t AND layer l+1 at step t-1, so it
            // is gathered whenever it exists, also in the pipeline tail where layer l itself is done
            const bool have_prev = (t >= 1 && t <= T) && !(l == L - 1 && t == T);
            const bool active = (t >= 0 && t < T);      // both uniform over the whole grid
            if (active && l == 0) {
                // x_t: f64 z-score fused into the load (estimator.py:103-104, watch_phone_pocket_nn.py:100)
#pragma unroll
                for (int e = 0; e < (MR * KX) / 256; ++e) {
                    const int idx = tid + 256 * e;
                    const int row = idx / KX, k = idx - row * KX;
                    const int b = row0 + row;
                    float v = 0.0f;
                    if (k < I && b < p.B) {
                        v = p.x[((size_t)b * T + t) * I + k];
                        if (normalize) v = (float)(((double)v - p.xx_m[k]) / p.xx_s[k]);
                    }
                    xin[row * SX + k] = v;
                }
            }
            if (have_prev) {
                if (!gather(l, (unsigned)t, (t - 1) & 1)) return;
            } else if (active) {
                __syncthreads();                        // xin (l == 0) visible
            }
            if (!active) continue;
            // ---- stacked-gate product on the matrix cores ------------------------------------------
            f32x4 acc[NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) acc[mt] = f32x4{bias_r[l], bias_r[l], bias_r[l], bias_r[l]};
            const float* rec_src = hbuf + (l * MR + r) * SH + 4 * g;
            if (l == 0) {
                mfma_span<NMT, QX, NW0>(acc, xin + r * SX + 4 * g, SX, w0, 0);
                if (t > 0) mfma_span<NMT, QH, NW0>(acc, rec_src, SH, w0, 4 * QX);
            } else {
                const float* in_src = hbuf + ((l - 1) * MR + r) * SH + 4 * g;
                if (l == 1) {
                    if constexpr (L > 1) {
                        mfma_span<NMT, QH, NW1>(acc, in_src, SH, w1, 0);
                        if (t > 0) mfma_span<NMT, QH, NW1>(acc, rec_src, SH, w1, 4 * QH);
                    }
                } else {
                    if constexpr (L > 2) {
                        mfma_span<NMT, QH, NW1>(acc, in_src, SH, w2, 0);
                        if (t > 0) mfma_span<NMT, QH, NW1>(acc, rec_src, SH, w2, 4 * QH);
                    }
                }
            }
            // ---- gate non-linearities on the accumulators (lane = one gate of one unit) ---------------
            float act[NMT][4];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) act[mt][i] = gate_act(acc[mt][i], gate == 2);
            // ---- 4x4 lane transpose: this lane collects i,f,g,o of (unit u, row tile `gate`) -----------
            // NMT < 4: tiles beyond NMT do not exist and their lanes idle through the update
            float gi[4], gf[4], gg[4], go[4];
            const int rowbase = lane & 48;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float vi = 0.f, vf = 0.f, vg = 0.f, vo = 0.f;
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    const float ti = __shfl(act[mt][i], rowbase + 0 + u, 64);
                    const float tf = __shfl(act[mt][i], rowbase + 4 + u, 64);
                    const float tg = __shfl(act[mt][i], rowbase + 8 + u, 64);
                    const float to = __shfl(act[mt][i], rowbase + 12 + u, 64);
                    if (gate == mt) { vi = ti; vf = tf; vg = tg; vo = to; }
                }
                gi[i] = vi; gf[i] = vf; gg[i] = vg; go[i] = vo;
            }
            // ---- cell update for (unit u, rows 16*gate + 4g + i), h into the own-slice staging ---------
            if (gate < NMT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float c = gf[i] * cst[l][i] + gi[i] * gg[i];
                    cst[l][i] = c;
                    own[(16 * gate + 4 * g + i) * SO + wave * 4 + u] = go[i] * tanhf(c);
                }
            }
            __syncthreads();
            // ---- publish the slice: 16-byte write-through stores, drain, barrier, one flag -----------
            if (tid < TPS) {
                const int row = tid >> 2, quad = tid & 3;
                const f32x4 hv = *reinterpret_cast<const f32x4*>(own + row * SO + 4 * quad);
                const unsigned base = (unsigned)((((size_t)cluster * L + l) * 2 + (t & 1)) * GH * MR * 16 * sizeof(float));
                __builtin_amdgcn_raw_buffer_store_b128(
                    __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, hv), hx_rsrc,
                    base + (unsigned)(((member * MR + row) * 16 + 4 * quad) * sizeof(float)), 0, 16 /* sc1 */);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(myflags + l * GH + member, (unsigned)(t + 1), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- head: gather h^{L-1}_{T-1}, each member finishes MR/GH (>= 1) of the cluster's windows ------
    if (!gather(L - 1, (unsigned)T, (T - 1) & 1)) return;
    {
        constexpr int RPM = (MR + GH - 1) / GH;          // rows per member
        const int n_out = RPM * O;
        if (tid < n_out) {
            const int rr = tid / O, o = tid - rr * O;
            const int row = member * RPM + rr;
            const int b = row0 + row;
            if (row < MR && b < p.B) {
                const float* hv = hbuf + ((L - 1) * MR + row) * SH;
                const float* wv = p.w_out + (size_t)o * H;
                float s = 0.0f;
                for (int k = 0; k < H; ++k) s = fmaf(hv[k], wv[k], s);
                p.y[(size_t)b * O + o] = s + p.b_out[o];
            }
        }
    }
}

template <int H, int L, int KX, int NMT>
size_t smem_bytes() {
    constexpr int MR = 16 * NMT;
    return ((size_t)L * MR * (H + 8) + (size_t)MR * (KX + 8) + (size_t)MR * 20 + 4) * sizeof(float);
}

template <int H, int L, int KX, int NMT>
hipError_t launch(const ClusterParams& p, int clusters, hipStream_t stream) {
    const size_t smem = smem_bytes<H, L, KX, NMT>();
    hipLaunchKernelGGL((ape_lstm_cluster<H, L, KX, NMT>), dim3(clusters * (H / 16)), dim3(256), smem, stream, p);
    return hipGetLastError();
}

template <int H, int L, int KX, int NMT>
hipError_t prepare() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_cluster<H, L, KX, NMT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes<H, L, KX, NMT>());
}

}  // namespace

// shapes the cluster kernel is built for: the deployed pocket / watch-only (H=256, L=2, I<=32) and
// upper-arm (H=128, L=3, 32<I<=64) regressors
bool ape_cluster_supported(int H, int L, int KX) { return (H == 256 && L == 2 && KX == 32) || (H == 128 && L == 3 && KX == 64); }

#define APE_CL_DISPATCH(FN, ...)                                             \
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

hipError_t ape_prepare_lstm_cluster(int H, int L, int KX) {
    for (int nmt : {1, 2, 4}) {
        hipError_t e = [&]() -> hipError_t { APE_CL_DISPATCH(prepare) }();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t ape_launch_lstm_cluster(int H, int L, int KX, int nmt, int clusters, const ClusterParams& p,
                                   hipStream_t stream) {
    APE_CL_DISPATCH(launch, p, clusters, stream)
}
