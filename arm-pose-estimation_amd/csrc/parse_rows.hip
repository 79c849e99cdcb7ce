// Batched feature builder: raw smartwatch(+phone) messages -> calibrated regressor inputs.
//
// Replaces parse_row_to_xx of the three estimators (reference estimate/watch_phone_pocket_nn.py:41-96,
// estimate/watch_only.py:46-82, estimate/watch_phone_uarm_nn.py:43-105) and the helpers they call in
// utility/transformations.py: android_quat_to_global(_no_north) :225-241, reduce_global_quat_to_y_rot
// :200-207, euler_to_quat :152-174 (y rotation only), quat_invert :244-254, hamilton_product :129-149,
// quat_rotate_vector :83-126, quat_to_6drr_1x6 :476-518, calib_watch_left_to_north_quat :182-197.
// Message layouts: data_types/messaging.py:20-68 (28 floats, watch only) and :96-187 (55 floats).
//
// One thread per row; float64 arithmetic (the reference computes on Python floats); rows are staged through
// LDS both ways so the global reads AND writes are coalesced and shared by the block's threads (a row's features go out
// rep times -- T slots on a cold start, n_mc window copies in Monte-Carlo mode: as a per-thread loop over a private array
// that was 68 us for ONE stream with 25 copies, scratch loads and dependent scalar stores).  HBM-bound and tiny: 220 B in,
// 88-304 B out per row.
#include "ape_internal.h"
#include "../../include/ape_hip.h"
#include "parse_device.h"

#pragma clang fp contract(off)

namespace {

using namespace ape_parsedev;

constexpr int PR_BLOCK = 64;                    // threads per workgroup
constexpr int PR_ROWS = 16;                     // rows per workgroup: a row is one dependent f64 chain (atan2 -> sin / cos -> quaternion
                                                // products -> atan2, ~5 us) whatever the block shape, so few rows per block = short copy
                                                // loops either side of it and more CUs busy (1024 rows: 11.7 -> 7.0 us)
constexpr int XW = 39;                          // feature row stride in LDS (odd: conflict-free per-thread rows)

// Output row n goes to out[n * out_stride + j * rep_stride + 0..I) for j < rep: rep = 1 and out_stride = I is the
// plain [N,I] matrix; the stream bank (ape_streams_push_rows) points it at one slot of every stream's window ring
// (out_stride = T*I) or, on a cold start, at all T of them (rep = T, rep_stride = I: estimator.py:96-97).
// big_endian: the rows are the UDP payload as received (55 or 28 big-endian float32, stream_listener/imu.py:53).
template <typename TOut>
__global__ __launch_bounds__(PR_BLOCK) void ape_parse_rows_kernel(const float* __restrict__ rows, int N, int width,
                                                                  int kind, TOut* __restrict__ out, int I,
                                                                  size_t out_stride, int rep, size_t rep_stride,
                                                                  int big_endian) {
    __shared__ float slab[PR_ROWS * 57];        // row stride 57: odd -> conflict-free per-thread rows
    const int tid = threadIdx.x;
    const size_t r0 = (size_t)blockIdx.x * PR_ROWS;
    const int n = (int)min((size_t)PR_ROWS, (size_t)N - r0);
    for (int idx = tid; idx < n * width; idx += PR_BLOCK) {
        const int rr = idx / width, c = idx - rr * width;
        float v = rows[r0 * width + idx];
        if (big_endian) v = __builtin_bit_cast(float, __builtin_bswap32(__builtin_bit_cast(unsigned, v)));
        slab[rr * 57 + c] = v;
    }
    __shared__ double xout[PR_ROWS * XW];
    __syncthreads();
    if (tid < n) {
    parse_row(slab + tid * 57, width, kind, xout + tid * XW);
    }
    __syncthreads();
    // all threads of the block write all rows: consecutive threads = consecutive features of one (row, copy)
    const int per_row = rep * I;
    for (int idx = tid; idx < n * per_row; idx += PR_BLOCK) {
        const int rr = idx / per_row, rem = idx - rr * per_row, j = rem / I, i = rem - j * I;
        out[(r0 + rr) * out_stride + j * rep_stride + i] = (TOut)xout[rr * XW + i];
    }
}

}  // namespace

hipError_t ape_launch_parse_rows(const float* rows, int N, int width, int kind, void* out, int out_dtype, int I,
                                 size_t out_stride, int rep, size_t rep_stride, int big_endian, hipStream_t stream) {
    const int grid = (N + PR_ROWS - 1) / PR_ROWS;
    if (out_dtype == APE_F32)
        hipLaunchKernelGGL(ape_parse_rows_kernel<float>, dim3(grid), dim3(PR_BLOCK), 0, stream, rows, N, width, kind,
                           static_cast<float*>(out), I, out_stride, rep, rep_stride, big_endian);
    else
        hipLaunchKernelGGL(ape_parse_rows_kernel<double>, dim3(grid), dim3(PR_BLOCK), 0, stream, rows, N, width, kind,
                           static_cast<double*>(out), I, out_stride, rep, rep_stride, big_endian);
    return hipGetLastError();
}

// features computed by the caller -> window rings (same addressing as above)
__global__ void ape_ring_write_kernel(const float* __restrict__ xx, int N, int I, float* __restrict__ out,
                                      size_t out_stride, int rep, size_t rep_stride) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * I) return;
    const size_t n = idx / I;
    const int i = (int)(idx - n * I);
    const float v = xx[idx];
    for (int j = 0; j < rep; ++j) out[n * out_stride + j * rep_stride + i] = v;
}

hipError_t ape_launch_ring_write(const float* xx, int N, int I, float* out, size_t out_stride, int rep,
                                 size_t rep_stride, hipStream_t stream) {
    const size_t total = (size_t)N * I;
    hipLaunchKernelGGL(ape_ring_write_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, xx, N, I, out,
                       out_stride, rep, rep_stride);
    return hipGetLastError();
}
