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

#pragma clang fp contract(off)

namespace {

struct Q { double w, x, y, z; };

__device__ __forceinline__ Q qmul(const Q a, const Q b) {
    return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
             a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
             a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q qconj(const Q q) { return Q{q.w, -q.x, -q.y, -q.z}; }
__device__ __forceinline__ Q qinv(const Q q) {
    const double n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return Q{q.w / n, -q.x / n, -q.y / n, -q.z / n};
}
// android (X east, Y north, Z up) -> global (X right, Y up, Z forward): [-w, x, z, y]
__device__ __forceinline__ Q no_north(const Q q) { return Q{-q.w, q.x, q.z, q.y}; }
// azimuth of q * (0,0,1): atan2(x, z) of the rotated forward axis
__device__ __forceinline__ double y_rot_of(const Q q) {
    const Q t = qmul(qmul(q, Q{0.0, 0.0, 0.0, 1.0}), qconj(q));
    return atan2(t.x, t.z);
}
__device__ __forceinline__ Q y_quat(double a) { return Q{cos(0.5 * a), 0.0, sin(0.5 * a), 0.0}; }
__device__ __forceinline__ Q rd(const float* row, int i) {
    return Q{(double)row[i], (double)row[i + 1], (double)row[i + 2], (double)row[i + 3]};
}
// first two columns of the rotation matrix, row-interleaved [m11,m12,m21,m22,m31,m32] (transforms3d formula)
__device__ __forceinline__ void six_drr(const Q q, double* o) {
    const double nq = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    if (nq < 2.220446049250313e-16) { o[0] = 1; o[1] = 0; o[2] = 0; o[3] = 1; o[4] = 0; o[5] = 0; return; }
    const double s = 2.0 / nq;
    const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    o[0] = 1.0 - (q.y * ys + q.z * zs); o[1] = q.x * ys - q.w * zs;
    o[2] = q.x * ys + q.w * zs;         o[3] = 1.0 - (q.x * xs + q.z * zs);
    o[4] = q.x * zs - q.w * ys;         o[5] = q.y * zs + q.w * xs;
}

// column positions (data_types/messaging.py)
struct Cols { int rot, fwd, ph_rot, ph_fwd, init_pres; };
__device__ __forceinline__ Cols cols_of(int width) {
    return width == 28 ? Cols{5, 23, -1, -1, 27} : Cols{5, 46, 28, 50, 54};
}
// sw_dt, gyro, lvel, lacc at 0,10..18; grav at 20..22 (pressure sits at 19)
__device__ __forceinline__ int sw_sensor_col(int i) { return i == 0 ? 0 : (i <= 9 ? 9 + i : 10 + i); }
// ph gyro, lvel, lacc at 33..41; grav at 43..45
__device__ __forceinline__ int ph_sensor_col(int i) { return i < 9 ? 33 + i : 34 + i; }

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
    const float* row = slab + tid * 57;
    const Cols cl = cols_of(width);
    double* xx = xout + tid * XW;
    int o = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) xx[o++] = (double)row[sw_sensor_col(i)];
    const double r_pres = (double)row[19] - (double)row[cl.init_pres];
    const Q sw_fwd = rd(row, cl.fwd), sw_rot = rd(row, cl.rot);
    Q north = y_quat(-y_rot_of(no_north(sw_fwd)));
    if (kind == APE_PARSE_WATCH_PHONE_UARM) {
        // north incl. the left-hand calibration turn; watch / phone offsets to the calibration pose
        north = qmul(Q{0.7071068, 0.0, -0.7071068, 0.0}, north);
        const Q larm_dst{-0.7071068, 0.0, -0.7071068, 0.0}, uarm_dst{0.7071068, 0.0, 0.7071068, 0.0};
        const Q sw_cal = qmul(qmul(north, no_north(sw_rot)), qmul(qinv(qmul(north, no_north(sw_fwd))), larm_dst));
        six_drr(sw_cal, xx + o); o += 6;
        xx[o++] = r_pres;
#pragma unroll
        for (int i = 0; i < 12; ++i) xx[o++] = (double)row[ph_sensor_col(i)];
        const Q ph_rot = rd(row, cl.ph_rot), ph_fwd = rd(row, cl.ph_fwd);
        const Q ph_cal = qmul(qmul(north, no_north(ph_rot)), qmul(qinv(qmul(north, no_north(ph_fwd))), uarm_dst));
        six_drr(ph_cal, xx + o); o += 6;
    } else {
        six_drr(qmul(north, no_north(sw_rot)), xx + o); o += 6;
        xx[o++] = r_pres;
        if (kind == APE_PARSE_WATCH_PHONE_POCKET) {
            const Q ph_rot_g = qmul(north, no_north(rd(row, cl.ph_rot)));
            const Q ph_fwd_g = qmul(north, no_north(rd(row, cl.ph_fwd)));
            const double hy = y_rot_of(qmul(ph_rot_g, qinv(ph_fwd_g)));
            xx[o++] = sin(hy);
            xx[o++] = cos(hy);
        }
    }
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
