// MLP pose regressor for gfx950: the reference's DropoutFF (estimate/nn_models.py:313-370), which its
// loader can dispatch instead of DropoutLSTM (nn_models.py:395-396):
//     Linear(I,H) -> leaky_relu -> n x [Linear(H,H) -> leaky_relu] -> dropout -> Linear(H,O)
// applied to the last axis of x, i.e. independently to every row (window step).
//
// One workgroup (4 wave64, one per SIMD) owns 16 * NMT rows (NMT = 2 once the batch is 64 rows per CU or more: every weight
// fragment streamed from L2 then feeds two row tiles; NMT = 1 below that, for the latency of small batches.  Measured on MI355X,
// 22 -> 256 -> 256 -> 256 -> 14: 65 536 rows 245 us with either, 262 144 rows 888 us = 82.8 TFLOP/s with NMT = 2; NMT = 4 -- one
// workgroup per CU, 145 KB of LDS -- is SLOWER, 291 us / 1173 us: this kernel lives on several workgroups per CU hiding each
// other's L2 latency, not on weight reuse) and walks the layers; activations ping-pong between
// two LDS buffers (row stride H+8 floats: conflict-free ds_read_b128), wave w owns output columns
// [w*H/4, (w+1)*H/4).  Each layer is a [16 x K] x [K x H] product on v_mfma_f32_16x16x4_f32 (exact f32):
// A operand from LDS (one ds_read_b128 per 16 k-values, reused by all the wave's tiles), B operand = the
// layer's weights streamed from L2 in a host-prepacked fragment order (1 KiB per wave-instruction,
// double-buffered one k-block ahead).  Bias is the initial accumulator, leaky_relu (slope 0.01, the torch
// default) and the optional dropout mask of the last hidden layer are fused into the accumulator write-back;
// the narrow output layer is a VALU dot product per (row, target).  f64 z-score of the inputs fused into the
// load (estimator.py:103-104).  Work per row: 2*(I*H + n*H*H + H*O) FLOP; HBM bytes: I*4 in + O*4 out.
#include "ape_internal.h"
#include "../../include/ape_hip.h"

namespace {

template <int NT>
__device__ __forceinline__ void load_b(f32x4 (&b)[NT], const f32x4* __restrict__ p) {
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = p[n * 64];
}

template <int NMT, int NT>
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[NMT][NT], const f32x4 (&a)[NMT], const f32x4 (&b)[NT]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], b[n][j], acc[mt][n], 0, 0, 0);
    }
}

template <int H, int NMT>
__global__ __launch_bounds__(256, 1) void ape_mlp_tile16(const MlpParams p) {
    constexpr int NT = H / 64;        // 16-column tiles per wave
    constexpr int SH = H + 8;
    constexpr int ROWS = APE_TILE_ROWS * NMT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * ROWS;
    const int KX = p.KX, SX = KX + 8;
    const bool normalize = (p.flags & APE_FLAG_NORMALIZE_INPUT) != 0;
    const bool drop_masks = (p.flags & APE_FLAG_DROPOUT_MASKS) != 0;
    const bool drop_philox = (p.flags & APE_FLAG_DROPOUT_PHILOX) != 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                              // [ROWS][SX]
    float* act = xin + ROWS * SX;                   // [2][ROWS][SH]

    // ---- inputs: row n of the batch lives at x[n * row_stride + row_offset + k] ------------------------------
    for (int idx = tid; idx < ROWS * KX; idx += 256) {
        const int row = idx / KX, k = idx - row * KX;
        const int n = row0 + row;
        float v = 0.0f;
        if (k < p.I && n < p.N) {
            v = p.x[(size_t)n * p.row_stride + p.row_offset + k];
            if (normalize) v = (float)(((double)v - p.xx_m[k]) / p.xx_s[k]);
        }
        xin[row * SX + k] = v;
    }
    __syncthreads();

    // ---- input layer + hidden layers on the matrix cores -------------------------------------------------------
    const int n_layers = p.n_hidden + 1;
#pragma unroll 1
    for (int j = 0; j < n_layers; ++j) {
        const int K = (j == 0) ? KX : H;
        const int nq = K / 16;                      // even (KX % 32 == 0, H % 64 == 0)
        const int sstride = (j == 0) ? SX : SH;      // row stride of the layer's input
        const float* src = (j == 0) ? xin + r * SX + 4 * g : act + (((j - 1) & 1) * ROWS + r) * SH + 4 * g;
        const f32x4* wl = p.wpack[j] + (size_t)wave * nq * NT * 64 + lane;
        f32x4 acc[NMT][NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float bv = p.bias[j][wave * (H / 4) + n * 16 + r];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) acc[mt][n] = f32x4{bv, bv, bv, bv};
        }
        auto load_a = [&](f32x4 (&a)[NMT], int q) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) a[mt] = *reinterpret_cast<const f32x4*>(src + mt * 16 * sstride + 16 * q);
        };
        f32x4 b0[NT], b1[NT], a0[NMT], a1[NMT];
        load_b<NT>(b0, wl);
#pragma unroll 1
        for (int q = 0; q < nq; q += 2) {
            load_b<NT>(b1, wl + (size_t)(q + 1) * NT * 64);
            load_a(a0, q);
            mfma_block<NMT, NT>(acc, a0, b0);
            if (q + 2 < nq) load_b<NT>(b0, wl + (size_t)(q + 2) * NT * 64);
            load_a(a1, q + 1);
            mfma_block<NMT, NT>(acc, a1, b1);
        }
        // leaky_relu (+ dropout on the last hidden activation, nn_models.py:351) into the other buffer
        float* dst = act + ((j & 1) * ROWS) * SH;
        const bool last = (j == n_layers - 1);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = wave * (H / 4) + n * 16 + r;
            uint32_t rnd[4] = {0, 0, 0, 0};
            if (last && drop_philox)
                philox4x32((uint32_t)(row0 + 16 * mt + 4 * g), 0u, (uint32_t)col, 0xFFu, (uint32_t)p.seed, (uint32_t)(p.seed >> 32), rnd);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * mt + 4 * g + i;
                float v = acc[mt][n][i];
                v = (v > 0.0f) ? v : p.neg_slope * v;
                if (last && (drop_masks || drop_philox)) {
                    float m;
                    if (drop_masks) {
                        const int nrow = row0 + row;
                        m = (nrow < p.N) ? p.mask[(size_t)nrow * H + col] : 0.0f;
                    } else {
                        const float uf = (float)(rnd[i] >> 8) * (1.0f / 16777216.0f);
                        m = (uf >= p.dropout_p) ? 1.0f / (1.0f - p.dropout_p) : 0.0f;
                    }
                    v *= m;
                }
                dst[row * SH + col] = v;
                // ImuPoseLSTM's input layer (nn_models.py:242): the activation itself is the result
                if (last && p.hidden_out != nullptr && row0 + row < p.N) p.hidden_out[(size_t)(row0 + row) * H + col] = v;
            }
        }
        __syncthreads();
    }

    // ---- output layer: (row, target) dot products, k ascending, one fma per term -----------------------------------------
    if (p.hidden_out == nullptr) {
        for (int idx = tid; idx < ROWS * p.O; idx += 256) {
            const int row = idx / p.O, o = idx - row * p.O;
            const int n = row0 + row;
            if (n < p.N) {
                const float* hv = act + (((n_layers - 1) & 1) * ROWS + row) * SH;
                const float* wv = p.w_out + (size_t)o * H;
                float s = 0.0f;
                for (int k = 0; k < H; ++k) s = fmaf(hv[k], wv[k], s);
                p.y[(size_t)n * p.O + o] = s + p.b_out[o];
            }
        }
    }
}

template <int H, int NMT>
size_t smem_of(int KX) { return ((size_t)APE_TILE_ROWS * NMT * (KX + 8) + 2 * (size_t)APE_TILE_ROWS * NMT * (H + 8)) * sizeof(float); }

}  // namespace

// Linear head over rows: y[n][o] = b_out[o] + sum_k hseq[n][k] * w_out[o][k] (k ascending, one fma per term: the same
// order as the heads inside the LSTM kernels).  64 rows per workgroup, rows and weights staged in LDS; HBM-bound
// (H*4 bytes in, O*4 out per row).  Used for the all-steps output of the cluster kernel.
namespace {
constexpr int HR_ROWS = 64;
__global__ __launch_bounds__(256) void ape_head_rows_kernel(const float* __restrict__ hseq, int N, int H, int O,
                                                            const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                            float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float hr_smem[];
    float* rows = hr_smem;                          // [64][H+4]
    float* ws = rows + HR_ROWS * (H + 4);           // [O][H+1]
    const int tid = threadIdx.x;
    const size_t n0 = (size_t)blockIdx.x * HR_ROWS;
    const int nr = (int)min((size_t)HR_ROWS, (size_t)N - n0);
    for (int idx = tid; idx < O * H; idx += 256) ws[(idx / H) * (H + 1) + idx % H] = w_out[idx];
    for (int idx = tid; idx < nr * (H / 4); idx += 256) {
        const int rr = idx / (H / 4), c4 = idx - rr * (H / 4);
        *reinterpret_cast<f32x4*>(rows + rr * (H + 4) + 4 * c4) = *reinterpret_cast<const f32x4*>(hseq + (n0 + rr) * H + 4 * c4);
    }
    __syncthreads();
    for (int idx = tid; idx < nr * O; idx += 256) {
        const int rr = idx / O, o = idx - rr * O;
        const float* hv = rows + rr * (H + 4);
        const float* wv = ws + o * (H + 1);
        float s = 0.0f;
        for (int k = 0; k < H; ++k) s = fmaf(hv[k], wv[k], s);
        y[(n0 + rr) * O + o] = s + b_out[o];
    }
}
}  // namespace

hipError_t ape_launch_head_rows(const float* hseq, int N, int H, int O, const float* w_out, const float* b_out, float* y,
                                hipStream_t stream) {
    const size_t smem = ((size_t)HR_ROWS * (H + 4) + (size_t)O * (H + 1)) * sizeof(float);      // <= 64 KB + 33 KB
    // (the dynamic-LDS limit is raised per DEVICE in ape_prepare_mlp_tile16, from ape_model_create)
    hipLaunchKernelGGL(ape_head_rows_kernel, dim3((N + HR_ROWS - 1) / HR_ROWS), dim3(256), smem, stream, hseq, N, H, O, w_out,
                       b_out, y);
    return hipGetLastError();
}

template <int H, int NMT>
static hipError_t launch_mlp(const MlpParams& p, hipStream_t stream) {
    const size_t smem = smem_of<H, NMT>(p.KX);      // (limit raised per device in ape_prepare_mlp_tile16)
    const int rows = APE_TILE_ROWS * NMT;
    hipLaunchKernelGGL((ape_mlp_tile16<H, NMT>), dim3((p.N + rows - 1) / rows), dim3(256), smem, stream, p);
    return hipGetLastError();
}

// `n_cus`: CU count of the device -- 32-row workgroups from 64 rows per CU on (a weight fragment then feeds two row tiles),
// 16-row workgroups below that (more workgroups, shorter latency); KX > 64 keeps the 16-row form (for safety: the input
// layers this kernel serves are narrow)
hipError_t ape_launch_mlp_tile16(int H, const MlpParams& p, hipStream_t stream, int n_cus) {
    const bool wide = p.N >= 64 * (n_cus > 0 ? n_cus : 256) && p.KX <= 64;
    if (H == 256) return wide ? launch_mlp<256, 2>(p, stream) : launch_mlp<256, 1>(p, stream);
    if (H == 128) return wide ? launch_mlp<128, 2>(p, stream) : launch_mlp<128, 1>(p, stream);
    return hipErrorInvalidValue;
}

// The dynamic-LDS limit is a per-DEVICE attribute of each kernel instantiation: raised to the CU's whole LDS for the
// current device, from ape_model_create (after its hipSetDevice) -- never from a process-wide flag in the launch path,
// which would leave every device but the first at the 64 KB default.
hipError_t ape_prepare_mlp_tile16(int H) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_head_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    if (e != hipSuccess) return e;
    if (H == 256) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_mlp_tile16<256, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_mlp_tile16<256, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
    } else if (H == 128) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_mlp_tile16<128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ape_mlp_tile16<128, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, APE_LDS_BYTES);
    } else {
        return hipErrorInvalidValue;
    }
    return e;
}
