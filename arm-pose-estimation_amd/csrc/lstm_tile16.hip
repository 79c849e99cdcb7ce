// Batch-tile LSTM kernel for gfx950 (MI355X): one workgroup = 16 IMU windows for ALL T steps.
//
// Replaces torch.nn.LSTM + torch.nn.Linear as used by DropoutLSTM.forward
// (reference estimate/nn_models.py:169-174,180-189): L stacked layers, gate order i,f,g,o,
// gates = (x W_ih^T + b_ih) + (h W_hh^T + b_hh), c = f*c + i*g, h = o*tanh(c), zero initial
// state per window, linear head on the last layer's output.
//
// Mapping onto CDNA4
//   * 256 threads = 4 wave64, one per SIMD (1 wave/SIMD -> the whole 512-VGPR file).
//   * wave w owns hidden units [w*H/4, (w+1)*H/4) and, for each of them, all four gates, so the
//     gate non-linearities and the cell update are lane-local on the MFMA accumulators; the
//     cell state c lives in registers for the whole window, h lives in LDS (double buffered)
//     and never touches HBM.
//   * per layer-step the stacked-gate product [16 x (in+H)] x [(in+H) x 4H] runs on
//     v_mfma_f32_16x16x4_f32 (exact f32, one rounding per product, f32 accumulate).  The A
//     operand (activations) is read from LDS as one ds_read_b128 per 16 k-values and reused by
//     16 (H=256) accumulator tiles; the B operand (weights) is streamed from L2 straight into
//     VGPRs as global_load_dwordx4 in a host-prepacked fragment order (1 KiB contiguous per
//     wave-instruction), double-buffered one 16-deep k-block ahead, including across
//     layer-step boundaries so the first block of the next layer-step flies under the
//     transcendental cell update.
//   * the only HBM traffic is x (read once, f64 z-score fused into the load) and y.
//
// Work per window: sum_l 2*4H*(in_l+H) FLOP per step.
#include "ape_internal.h"
#include "../../include/ape_hip.h"

namespace {

// sigmoid(v), or tanh(v) = 2*sigmoid(2v) - 1: hardware v_exp_f32 (2^x) and v_rcp_f32, both ~1 ulp -- the same formula
// as the cluster kernel's (lstm_cluster_common.h), absolute error of the activation ~1e-7; the libm forms cost about
// 10x the instructions, a fifth of a step's time beside the MFMAs
__device__ __forceinline__ float gate_act(float v, bool is_tanh) {
    const float e = __builtin_amdgcn_exp2f((is_tanh ? -2.885390081777927f : -1.4426950408889634f) * v);
    const float s = __builtin_amdgcn_rcpf(1.0f + e);
    return is_tanh ? 2.0f * s - 1.0f : s;
}

template <int NT>
__device__ __forceinline__ void load_b(f32x4 (&b)[NT], const f32x4* __restrict__ p) {
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = p[n * 64];
}

// 4*NT MFMAs: k-index outer so consecutive MFMAs hit different accumulators (40-cycle dependent
// latency of v_mfma_f32_16x16x4_f32 vs its 32-cycle issue).
template <int NT>
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[NT], const f32x4 a, const f32x4 (&b)[NT]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[n][j], acc[n], 0, 0, 0);
    }
}

template <int H, int L, int XE>
__global__ __launch_bounds__(256, 1) void ape_lstm_tile16(const LstmParams p) {
    constexpr int UB = H / 64;        // 16-unit blocks per wave
    constexpr int NT = 4 * UB;        // accumulator tiles per wave (4 gates x UB)
    constexpr int SH = H + 8;         // LDS row stride of the h buffers (floats): b128 reads conflict-free
    constexpr int QH = H / 16;        // k-blocks of the recurrent part

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15;          // A operand: batch row; C operand: unit column
    const int g = lane >> 4;          // A/B operand: k sub-index; C operand: row group
    const int row0 = blockIdx.x * APE_TILE_ROWS;
    const int KX = p.KX;
    const int SX = KX + 8;
    const int QX = KX / 16;
    const int T = p.T, I = p.I, O = p.O;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool bcast_x = (p.flags & APE_FLAG_BROADCAST_X) != 0;   // monte_carlo_predictions: one window, B rows
    const bool all_steps = (p.flags & APE_FLAG_ALL_STEPS) != 0;
    const bool drop_masks = (p.flags & APE_FLAG_DROPOUT_MASKS) != 0;
    const bool drop_philox = (p.flags & APE_FLAG_DROPOUT_PHILOX) != 0;
    const bool drop = (drop_masks || drop_philox) && L > 1;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                                  // [2][16][SX]
    float* hbuf = xin + 2 * APE_TILE_ROWS * SX;         // [L][2][16][SH]
    float* dbuf = hbuf + L * 2 * APE_TILE_ROWS * SH;    // [L-1][16][SH]   (carved only when dropout is on)
    float* wout_s = dbuf + (drop ? (L - 1) : 0) * APE_TILE_ROWS * SH;    // [O][H+1]
    float* bias_s = wout_s + O * (H + 1);                                // [L][4H]: b_ih + b_hh, read at every layer-step

    // ---- stage the biases and the head weights once ------------------------------------------
#pragma unroll
    for (int l = 0; l < L; ++l)
        for (int idx = tid; idx < 4 * H; idx += 256) bias_s[l * 4 * H + idx] = p.bias[l][idx];
    for (int idx = tid; idx < O * H; idx += 256) {
        const int o = idx / H, k = idx - o * H;
        wout_s[o * (H + 1) + k] = p.w_out[idx];
    }

    // ---- x staging: thread owns up to 4 (row,k) elements of the [16][KX] step slab -----------
    const int n_el = KX / 16;         // 2 (KX=32), 4 (KX=64) or 16 (KX=256: XE=16 instantiation, ImuPoseLSTM)
    // (the wide instantiation copies its 16 elements per thread straight to LDS in store_x -- one exposed
    //  round trip per step, ~4 % of a step -- instead of holding 16 more registers across the MFMAs)
    //  The one-layer wide instantiation (layer 1 over a shared layer-0 sequence) has the registers to spare: its 16
    //  values per thread are fetched a step ahead like the narrow inputs, so nothing of them is exposed.)
    constexpr bool XPRE = (XE > 4 && L == 1);            // grouped wide input prefetched into registers
    constexpr int XR = XPRE ? APE_TILE_ROWS : ((XE > 4) ? 1 : XE);
    const size_t x_rows = p.x_row_stride ? p.x_row_stride : (size_t)T * I;
    float xr[XR];
    int x_step = 0;
    // source window of each of the tile's rows (grouped input: row b reads window b / x_group)
    int xsrc[XPRE ? APE_TILE_ROWS : 1];
    if (XPRE && p.x_group > 0) {
#pragma unroll
        for (int e = 0; e < APE_TILE_ROWS; ++e) xsrc[XPRE ? e : 0] = (row0 + e < p.B) ? (row0 + e) / p.x_group : -1;
    }
    auto fetch_x = [&](int t) {
        x_step = t;
        if (XPRE && p.x_group > 0) {
            const int slot = (t + p.x_ring >= T) ? t + p.x_ring - T : t + p.x_ring;
#pragma unroll
            for (int e = 0; e < APE_TILE_ROWS; ++e)
                xr[XPRE ? e : 0] = (xsrc[XPRE ? e : 0] >= 0) ? p.x[((size_t)xsrc[XPRE ? e : 0] * T + slot) * I + tid] : 0.0f;
            return;
        }
        if (XE > 4) return;
#pragma unroll
        for (int e = 0; e < XR; ++e) {
            xr[e] = 0.0f;
            if (e < n_el) {
                const int idx = tid + 256 * e;
                const int row = idx / KX, k = idx - row * KX;
                const int b = row0 + row;
                if (k < I && b < p.B) {
                    const float v = p.x[(size_t)(bcast_x ? 0 : b) * x_rows + (size_t)(t + p.x_ring >= T ? t + p.x_ring - T : t + p.x_ring) * I + k];
                    // f64 z-score then round to f32: estimator.py:103-104 + watch_phone_pocket_nn.py:100
                    xr[e] = normalize ? (float)(((double)v - p.xx_m[k]) / p.xx_s[k]) : v;
                }
            }
        }
    };
    auto store_x = [&](int buf) {
        if (XE > 4 && p.x_group > 0) {
            // Monte-Carlo samples of shared windows (stream bank): the input is the layer below's output sequence
            // [B / x_group, T, H], computed ONCE per stream, and row b is sample b % x_group of stream b / x_group.
            // The inter-layer dropout mask is drawn here with the counters the fused kernel uses for its layer 0
            // (rows b & ~3, step, unit, layer 0; value index b & 3), so the samples are the ones a fused launch over
            // the same rows draws.  Thread = unit (256 / KX row groups side by side), one Philox call per four rows.
            const int slot = (x_step + p.x_ring >= T) ? x_step + p.x_ring - T : x_step + p.x_ring;
            const float keep = 1.0f / (1.0f - p.dropout_p);
            constexpr int KXC = 16 * XE;                 // input width = units of the layer below (256 or 128)
            constexpr int NP = 256 / KXC;                // row groups staged side by side
            const int unit = tid % KXC, part = tid / KXC;
            const uint32_t below = (uint32_t)(p.layer_base > 0 ? p.layer_base - 1 : 0);     // the layer whose output this is
#pragma unroll
            for (int gi = 0; gi < APE_TILE_ROWS / 4 / NP; ++gi) {
                const int grp = gi * NP + part;
                uint32_t rnd[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                if (drop_philox)
                    philox4x32((uint32_t)(row0 + 4 * grp), (uint32_t)x_step, (uint32_t)unit, below, (uint32_t)p.seed,
                               (uint32_t)(p.seed >> 32), rnd);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * grp + i, b = row0 + row;
                    float v = 0.0f;
                    if (b < p.B) {
                        v = XPRE ? xr[XPRE ? row : 0] : p.x[((size_t)(b / p.x_group) * T + slot) * I + unit];
                        if (drop_philox) {
                            const float uf = (float)(rnd[i] >> 8) * (1.0f / 16777216.0f);
                            v = (uf >= p.dropout_p) ? v * keep : 0.0f;
                        }
                    }
                    xin[(buf * APE_TILE_ROWS + row) * SX + unit] = v;
                }
            }
            return;
        }
        if (XE > 4) {                 // wide: rows are full (I == KX), already normalised by the layer in front
            const int slot = (x_step + p.x_ring >= T) ? x_step + p.x_ring - T : x_step + p.x_ring;
#pragma unroll
            for (int e = 0; e < XE / 4; ++e) {
                const int idx = (tid + 256 * e) * 4;
                const int row = idx / KX, k = idx - row * KX;
                const int b = row0 + row;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (b < p.B) v = *reinterpret_cast<const f32x4*>(p.x + ((size_t)(bcast_x ? 0 : b) * T + slot) * I + k);
                *reinterpret_cast<f32x4*>(xin + (buf * APE_TILE_ROWS + row) * SX + k) = v;
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < XR; ++e) {
            if (e < n_el) {
                const int idx = tid + 256 * e;
                const int row = idx / KX, k = idx - row * KX;
                xin[(buf * APE_TILE_ROWS + row) * SX + k] = xr[e];
            }
        }
    };
    fetch_x(0);
    store_x(0);
    if (XPRE && T > 1) fetch_x(1);

    // ---- per-wave weight stream bases ------------------------------------------------------------
    const f32x4* wbase[L];
    int qtot[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        qtot[l] = (l == 0 ? QX : QH) + QH;
        wbase[l] = p.wpack[l] + (size_t)wave * qtot[l] * NT * 64 + lane;
    }

    float cst[L][UB][4];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) cst[l][u][i] = 0.0f;

    // caller-given initial state (DropoutLSTM.forward(x, hs=(h0, c0)), nn_models.py:180-189): h0 into the buffer step 0
    // reads as h_{-1}, c0 into the cell registers
    const bool have_hs = p.h0 != nullptr;
    if (have_hs) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
            for (int idx = tid; idx < APE_TILE_ROWS * H; idx += 256) {
                const int row = idx / H, unit = idx - row * H, b = row0 + row;
                hbuf[((l * 2 + 1) * APE_TILE_ROWS + row) * SH + unit] = (b < p.B) ? p.h0[((size_t)l * p.hs_rows + b) * H + unit] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int b = row0 + 4 * g + i;
                    if (b < p.B) cst[l][u][i] = p.c0[((size_t)l * p.hs_rows + b) * H + wave * (H / 4) + u * 16 + r];
                }
        }
    }

    f32x4 b0[NT], b1[NT];
    load_b<NT>(b0, wbase[0]);     // first k-block of (layer 0, t = 0)
    __syncthreads();              // xin[0], wout_s (and h0) visible

#ifdef APE_T16_STAMPS
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#define TS(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tk[i] += now_ - tlast; tlast = now_; }
    unsigned long long tlast = t_begin;
#else
#define TS(i)
#endif
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1, prv = cur ^ 1;
        // global loads fly under this step's MFMAs.  (XPRE: the 16 loads per thread come from HBM and would sit in
        // front of the step's first weight loads in the in-order return queue; they are issued behind the k-loop
        // instead, a whole step ahead of their use)
        if (!XPRE && t + 1 < T) fetch_x(t + 1);

#pragma unroll
        for (int l = 0; l < L; ++l) {
            // ---- accumulators start at b_ih + b_hh -------------------------------------------
            f32x4 acc[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int gate = n / UB, u = n % UB;
                const float bv = bias_s[l * 4 * H + gate * H + wave * (H / 4) + u * 16 + r];
                acc[n] = f32x4{bv, bv, bv, bv};
            }
            // ---- A sources -------------------------------------------------------------------
            const int qin = (l == 0) ? QX : QH;
            const float* in_src;
            if (l == 0) in_src = xin + (cur * APE_TILE_ROWS + r) * SX + 4 * g;
            else if (drop) in_src = dbuf + ((l - 1) * APE_TILE_ROWS + r) * SH + 4 * g;
            else in_src = hbuf + (((l - 1) * 2 + cur) * APE_TILE_ROWS + r) * SH + 4 * g;
            const float* rec_src = hbuf + ((l * 2 + prv) * APE_TILE_ROWS + r) * SH + 4 * g;
            TS(0)                                        // 0: step/layer set-up (bias, pointers)
            // t == 0 without an initial state: h_{-1} = 0, the recurrent k-blocks contribute nothing and are skipped
            const int nq = (t == 0 && !have_hs) ? qin : qtot[l];
            // first k-block of the NEXT layer-step, prefetched under this one's tail
            const f32x4* wnext = wbase[(l + 1) % L];
            const f32x4* wl = wbase[l];

#pragma unroll 1
            for (int q = 0; q < nq; q += 2) {       // nq is even by construction (KX % 32 == 0, H % 64 == 0)
                load_b<NT>(b1, wl + (size_t)(q + 1) * NT * 64);
                {
                    const float* src = (q < qin) ? in_src + 16 * q : rec_src + 16 * (q - qin);
                    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
                    mfma_block<NT>(acc, a, b0);
                }
                load_b<NT>(b0, (q + 2 < nq) ? wl + (size_t)(q + 2) * NT * 64 : wnext);
                {
                    const int q1 = q + 1;
                    const float* src = (q1 < qin) ? in_src + 16 * q1 : rec_src + 16 * (q1 - qin);
                    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
                    mfma_block<NT>(acc, a, b1);
                }
            }

            TS(1)                                        // 1: k-loop
            if (XPRE) {                                  // (L == 1) x_{t+1} -> the other xin buffer, x_{t+2} into flight
                if (t + 1 < T) store_x(prv);
                if (t + 2 < T) fetch_x(t + 2);
            }
            TS(2)                                        // 2: x staging (XPRE)
            // ---- gates + cell update, lane-local: lane holds rows 4g..4g+3 of unit column r ----
            float* hdst = hbuf + ((l * 2 + cur) * APE_TILE_ROWS) * SH;
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int unit = wave * (H / 4) + u * 16 + r;
                uint32_t rnd[4] = {0, 0, 0, 0};
                if (drop_philox && l < L - 1)
                    philox4x32((uint32_t)(row0 + 4 * g), (uint32_t)t, (uint32_t)unit, (uint32_t)(l + p.layer_base),
                               (uint32_t)p.seed, (uint32_t)(p.seed >> 32), rnd);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float iv = gate_act(acc[0 * UB + u][i], false);
                    const float fv = gate_act(acc[1 * UB + u][i], false);
                    const float gv = gate_act(acc[2 * UB + u][i], true);
                    const float ov = gate_act(acc[3 * UB + u][i], false);
                    const float c = fv * cst[l][u][i] + iv * gv;
                    cst[l][u][i] = c;
                    const float h = ov * gate_act(c, true);
                    const int row = 4 * g + i;
                    hdst[row * SH + unit] = h;
                    if (l == L - 1 && p.hseq != nullptr && row0 + row < p.B)
                        p.hseq[((size_t)(row0 + row) * T + t) * H + unit] = h;
                    if (drop && l < L - 1) {
                        float m;
                        if (drop_masks) {
                            const int b = row0 + row;
                            m = (b < p.B) ? p.masks[(((size_t)l * p.B + b) * T + t) * H + unit] : 0.0f;
                        } else {
                            // uniform in [0,1) from 24 random bits; keep with probability 1-p
                            const float uf = (float)(rnd[i] >> 8) * (1.0f / 16777216.0f);
                            m = (uf >= p.dropout_p) ? 1.0f / (1.0f - p.dropout_p) : 0.0f;
                        }
                        dbuf[(l * APE_TILE_ROWS + row) * SH + unit] = h * m;
                    }
                }
            }
            if (!XPRE && l == L - 1 && t + 1 < T) store_x(prv);   // x_{t+1} -> the other xin buffer
            TS(3)                                        // 3: gates + h store (+ x staging of the other paths)
            __syncthreads();                             // h^l_t (and x_{t+1}) visible
            TS(4)                                        // 4: barrier
        }

        // ---- linear head on h^{L-1}_t: output_layer of nn_models.py:189 -----------------------------
        if (p.y != nullptr && (all_steps || t == T - 1)) {
            if (tid < APE_TILE_ROWS * O) {
                const int row = tid / O, o = tid - row * O;
                const float* hv = hbuf + (((L - 1) * 2 + cur) * APE_TILE_ROWS + row) * SH;
                const float* wv = wout_s + o * (H + 1);
                float s = 0.0f;
                for (int k = 0; k < H; ++k) s = fmaf(hv[k], wv[k], s);
                s += p.b_out[o];
                const int b = row0 + row;
                if (b < p.B) {
                    if (all_steps) p.y[((size_t)b * T + t) * O + o] = s;
                    else p.y[(size_t)b * O + o] = s;
                }
            }
        }
    }
#ifdef APE_T16_STAMPS
    TS(5)                                                // 5: head
    if (blockIdx.x == 300 && tid == 0 && p.x_group > 0)
        printf("t16 stamps (cycles over %d steps): setup %llu  k-loop %llu  xstage %llu  gates %llu  barrier %llu  head %llu  total %llu\n",
               T, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], __builtin_amdgcn_s_memtime() - t_begin);
#endif
}

template <int H, int L, int XE = 4>
hipError_t launch(const LstmParams& p, hipStream_t stream) {
    const bool drop = (p.flags & (APE_FLAG_DROPOUT_MASKS | APE_FLAG_DROPOUT_PHILOX)) != 0 && L > 1;
    const size_t smem = ape_lstm_tile16_smem_bytes(H, L, p.KX, p.O, drop);
    const int grid = (p.B + APE_TILE_ROWS - 1) / APE_TILE_ROWS;
    hipLaunchKernelGGL((ape_lstm_tile16<H, L, XE>), dim3(grid), dim3(256), smem, stream, p);
    return hipGetLastError();
}

// the attribute is per kernel instantiation and process-wide: it is raised to the CU's whole LDS, never to one
// model's own need (a later model with a smaller layout must not lower it under an earlier one's launches)
template <int H, int L, int XE = 4>
hipError_t prepare(size_t smem) {
    if (smem > APE_LDS_BYTES) return hipErrorInvalidValue;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_lstm_tile16<H, L, XE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
}

}  // namespace

size_t ape_lstm_tile16_smem_bytes(int H, int L, int KX, int O, bool dropout) {
    const size_t SX = KX + 8, SH = H + 8;
    size_t fl = 2 * APE_TILE_ROWS * SX + (size_t)L * 2 * APE_TILE_ROWS * SH +
                (size_t)((dropout && L > 1) ? L - 1 : 0) * APE_TILE_ROWS * SH + (size_t)O * (H + 1) + (size_t)L * 4 * H;
    return fl * sizeof(float);
}

#define APE_DISPATCH(FN, ...)                                   \
    if (H == 256) {                                             \
        if (L == 1) return FN<256, 1>(__VA_ARGS__);             \
        if (L == 2) return FN<256, 2>(__VA_ARGS__);             \
        if (L == 3) return FN<256, 3>(__VA_ARGS__);             \
    } else if (H == 128) {                                      \
        if (L == 1) return FN<128, 1>(__VA_ARGS__);             \
        if (L == 2) return FN<128, 2>(__VA_ARGS__);             \
        if (L == 3) return FN<128, 3>(__VA_ARGS__);             \
    }                                                           \
    return hipErrorInvalidValue;

// raise the dynamic-LDS limit of the instantiation once, at model creation (not in the launch
// path, which must stay graph-capturable)
hipError_t ape_prepare_lstm_tile16(int H, int L, size_t smem_bytes) { APE_DISPATCH(prepare, smem_bytes) }

hipError_t ape_launch_lstm_tile16(int H, int L, const LstmParams& p, hipStream_t stream) {
    if (p.KX > 64) {      // wide layer-0 input: the 256 activations of ImuPoseLSTM's input layer (H=256 L=2), or
                          // the layer-0 output sequence in front of layer 1 run on its own (H=256 L=1)
        if (H == 256 && L == 2 && p.KX == 256) return launch<256, 2, 16>(p, stream);
        if (H == 256 && L == 1 && p.KX == 256) return launch<256, 1, 16>(p, stream);
        if (H == 128 && L == 2 && p.KX == 128) return launch<128, 2, 8>(p, stream);       // layers 1-2 of the 3 x 128 upper-arm model
        return hipErrorInvalidValue;
    }
    APE_DISPATCH(launch, p, stream)
}

hipError_t ape_prepare_lstm_tile16_wide(size_t smem_bytes) { return prepare<256, 2, 16>(smem_bytes); }
hipError_t ape_prepare_lstm_tile16_upper(int H, int L, size_t smem_bytes) {
    if (H == 256 && L == 1) return prepare<256, 1, 16>(smem_bytes);
    if (H == 128 && L == 2) return prepare<128, 2, 8>(smem_bytes);
    return hipErrorInvalidValue;
}
