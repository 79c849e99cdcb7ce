// Quaternion / forward-kinematics post-filter kernels for gfx950.
//
// ape_fk_kernel      replaces estimate_joints.arm_pose_from_nn_targets (reference
//                    estimate/estimate_joints.py:16-92) incl. utility/transformations.py:
//                    six_drr_1x6_to_rot_mat_1x9 (:602-637), rot_mat_to_quat (:521-545),
//                    hips_sin_cos_to_quat (:177-179), quat_rotate_vector (:83-126),
//                    hamilton_product (:129-149); optionally the `pred*yy_s+yy_m` of
//                    estimate/estimator.py:108-109 in front of it.
// ape_msg_kernel     replaces compose_msg.msg_from_nn_targets_est (estimate/compose_msg.py:13-108)
//                    incl. average_quaternions (transformations.py:32-51).
//
// The reference does this arithmetic in float64 (numpy), so both kernels compute in float64
// whatever the storage type: the work is a few hundred flops per row and HBM/latency bound.
//
// rot_mat_to_quat: the reference takes the dominant eigenvector of a symmetric 4x4 matrix
// (numpy.linalg.eigh, ~70 % of its frame time).  For the orthonormal matrices Gram-Schmidt
// produces that eigenvector is the rotation's unit quaternion, computed here in closed form
// with Shepperd's pivoting (largest of trace, m00, m11, m22) and the reference's w >= 0 sign
// rule; agreement with the eigh route is ~1e-15 (tests/test_oracle_golden.py).
#include "ape_internal.h"
#include "../../include/ape_hip.h"
#include "fk_device.h"
#include "stream_post_device.h"

// numpy evaluates a*b+c with two roundings; keep the device arithmetic the same
#pragma clang fp contract(off)

namespace {

using namespace ape_fkdev;

using namespace ape_postdev;

// row -> est row, shared by the batched FK kernel and the message kernel
__device__ void fk_row(const double* pr, const double* body, int layout, double* e) {
    const Vec3 larm_vec{body[0], body[1], body[2]};
    const Vec3 uarm_vec{body[3], body[4], body[5]};
    const Vec3 uarm_orig_rh{body[6], body[7], body[8]};
    if (layout == APE_LAYOUT_ORI_CAL_LARM_UARM_HIPS) {             // estimate_joints.py:48-71
        const Quat uq = six_drr_to_quat(pr + 6), lq = six_drr_to_quat(pr);
        const Quat hq = hips_quat(pr[12], pr[13]);
        const Vec3 uo = qrot(hq, uarm_orig_rh);
        const Vec3 lo = vadd(qrot(uq, uarm_vec), uo);
        const Vec3 ho = vadd(qrot(lq, larm_vec), lo);
        put_v(e, ho); put_v(e + 3, lo); put_v(e + 6, uo); put_q(e + 9, lq); put_q(e + 13, uq); put_q(e + 17, hq);
    } else if (layout == APE_LAYOUT_ORI_CAL_LARM_UARM) {           // estimate_joints.py:74-92
        const Quat uq = six_drr_to_quat(pr + 6), lq = six_drr_to_quat(pr);
        const Vec3 lo = vadd(qrot(uq, uarm_vec), uarm_orig_rh);
        const Vec3 ho = vadd(qrot(lq, larm_vec), lo);
        put_v(e, ho); put_v(e + 3, lo); put_q(e + 6, lq); put_q(e + 10, uq);
    } else {                                                       // estimate_joints.py:20-45
        const Quat uq = six_drr_to_quat(pr + 12), lq = six_drr_to_quat(pr + 3);
        const Quat hq = hips_quat(pr[18], pr[19]);
        const Vec3 uo = qrot(hq, uarm_orig_rh);
        e[0] = pr[0]; e[1] = pr[1]; e[2] = pr[2];
        e[3] = pr[9]; e[4] = pr[10]; e[5] = pr[11];
        put_v(e + 6, uo); put_q(e + 9, lq); put_q(e + 13, uq); put_q(e + 17, hq);
    }
}

// A row's three independent chains side by side: a workgroup of two waves owns 32 rows.  Wave 0, lane 2 r + s: the 6D -> quaternion chain of row r's lower (s = 0) / upper (s = 1) arm and the rotated
// bone vector -- one instruction stream for both, no divergence; wave 1, lane r: the hips chain (atan2, cos, sin) and the rotated
// shoulder origin, lanes 32 + r the columns that pass through.  The two sums that join the chains (estimate_joints.py:60-62,
// 84-85) follow a barrier, in the reference's order.  (One thread per row, the first form, is a chain of ~1400 f64 instructions:
// 7.1 us for 1024 rows against 2-3 here, 4.2 us for the one row of the latency path -- through an LDS slab -- against ~3.)
constexpr int FK3_ROWS = 32;

template <typename TIn, typename TOut>
__global__ __launch_bounds__(128) void ape_fk3_kernel(const FkParams p) {
    __shared__ double rot[FK3_ROWS][3][3];                 // [row][lower arm, upper arm, shoulder origin][xyz]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool hips = p.layout != APE_LAYOUT_ORI_CAL_LARM_UARM;
    const bool full = p.layout != APE_LAYOUT_ORI_CAL_LARM_UARM_HIPS && hips;      // the 20-column layout (estimate_joints.py:20-45)
    // column of the first 6D value of the lower / upper arm, of the hips' sine; est columns of the quaternions
    const int c_l = full ? 3 : 0, c_u = full ? 12 : 6, c_h = full ? 18 : 12;
    const int e_lq = hips ? 9 : 6, e_uq = hips ? 13 : 10, e_hq = 17;
    auto load = [&](const TIn* src, int c) -> double {
        double v = (double)src[c];
        if (p.yy_m) v = v * p.yy_s[c] + p.yy_m[c];         // de-normalisation in f64: estimator.py:108-109
        return v;
    };
    if (wave == 0) {
        const int r = lane >> 1, sub = lane & 1;
        const size_t row = (size_t)blockIdx.x * FK3_ROWS + r;
        if (row < (size_t)p.N) {
            const TIn* src = static_cast<const TIn*>(p.preds) + row * p.O;
            TOut* dst = static_cast<TOut*>(p.est) + row * p.W;
            double s6[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) s6[c] = load(src, (sub ? c_u : c_l) + c);
            const Quat q = six_drr_to_quat(s6);
            const Vec3 bone = sub ? Vec3{p.body[3], p.body[4], p.body[5]} : Vec3{p.body[0], p.body[1], p.body[2]};
            const Vec3 v = qrot(q, bone);
            rot[r][sub][0] = v.x; rot[r][sub][1] = v.y; rot[r][sub][2] = v.z;
            const int eq = sub ? e_uq : e_lq;
            dst[eq] = (TOut)q.w; dst[eq + 1] = (TOut)q.x; dst[eq + 2] = (TOut)q.y; dst[eq + 3] = (TOut)q.z;
        }
    } else {
        const int r = lane & 31;
        const size_t row = (size_t)blockIdx.x * FK3_ROWS + r;
        if (row < (size_t)p.N) {
            const TIn* src = static_cast<const TIn*>(p.preds) + row * p.O;
            TOut* dst = static_cast<TOut*>(p.est) + row * p.W;
            if (lane < 32) {
                Vec3 uo{p.body[6], p.body[7], p.body[8]};
                if (hips) {
                    const Quat hq = hips_quat(load(src, c_h), load(src, c_h + 1));
                    uo = qrot(hq, uo);
                    dst[e_hq] = (TOut)hq.w; dst[e_hq + 1] = (TOut)hq.x; dst[e_hq + 2] = (TOut)hq.y; dst[e_hq + 3] = (TOut)hq.z;
                    dst[6] = (TOut)uo.x; dst[7] = (TOut)uo.y; dst[8] = (TOut)uo.z;
                }
                rot[r][2][0] = uo.x; rot[r][2][1] = uo.y; rot[r][2][2] = uo.z;
            } else if (full) {                               // hand and lower-arm positions are network outputs here
#pragma unroll
                for (int c = 0; c < 3; ++c) { dst[c] = (TOut)load(src, c); dst[3 + c] = (TOut)load(src, 9 + c); }
            }
        }
    }
    __syncthreads();
    if (wave == 0 && !full) {
        const int r = lane >> 1, c = lane & 1;             // two lanes per row: the lower-arm origin / the hand origin
        const size_t row = (size_t)blockIdx.x * FK3_ROWS + r;
        if (row < (size_t)p.N) {
            TOut* dst = static_cast<TOut*>(p.est) + row * p.W;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double lo = rot[r][1][k] + rot[r][2][k];                 // qrot(uq, uarm_vec) + uo
                if (c == 0) dst[3 + k] = (TOut)lo;
                else dst[k] = (TOut)(rot[r][0][k] + lo);                       // qrot(lq, larm_vec) + lo
            }
        }
    }
}

// ---- message: N est rows -> 25 doubles ------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* scratch) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__global__ __launch_bounds__(256) void ape_msg_kernel(const MsgParams p) {
    __shared__ double scratch[4];
    const int tid = threadIdx.x;
    const int W = p.W, N = p.N;
    const bool hips = p.layout != APE_LAYOUT_ORI_CAL_LARM_UARM;
    const int qc_l = hips ? 9 : 6, qc_u = hips ? 13 : 10, qc_h = 17;
    const int nq = hips ? 3 : 2;

    double out_q[3][4];
    double orig_mean[9];
    if (N > 1) {
        // sign-aligned mean: sum_i sign(q_i . q_0) q_i / N, normalised (transformations.py:40-51)
        const double wgt = 1.0 / (double)N;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < nq) {      // uniform across the block
                const int col = (k == 0) ? qc_l : (k == 1) ? qc_u : qc_h;
                const double* q0 = p.est + col;
                const double r0 = q0[0], r1 = q0[1], r2 = q0[2], r3 = q0[3];
                double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                for (int i = tid; i < N; i += 256) {
                    const double* qi = p.est + (size_t)i * W + col;
                    // sign rule of transformations.py:44: np.dot(qi, q0) < 0.0.  numpy hands this to BLAS ddot,
                    // whose 4-element loop is an FMA chain; at (near-)exact orthogonality only that chain
                    // reproduces the sign of the rounding residue, so the same chain is used here.
                    const double d = fma(qi[3], r3, fma(qi[2], r2, fma(qi[1], r1, qi[0] * r0)));
                    const double sg = (i > 0 && d < 0.0) ? -wgt : wgt;
                    a0 += qi[0] * sg; a1 += qi[1] * sg; a2 += qi[2] * sg; a3 += qi[3] * sg;
                }
                a0 = block_sum(a0, scratch); a1 = block_sum(a1, scratch);
                a2 = block_sum(a2, scratch); a3 = block_sum(a3, scratch);
                const double nrm = sqrt(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3);
                out_q[k][0] = a0 / nrm; out_q[k][1] = a1 / nrm; out_q[k][2] = a2 / nrm; out_q[k][3] = a3 / nrm;
            }
        }
        if (p.layout == APE_LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS) {   // compose_msg.py:26-29: plain means
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                double a = 0;
                for (int i = tid; i < N; i += 256) a += p.est[(size_t)i * W + c];
                orig_mean[c] = block_sum(a, scratch) / (double)N;
            }
        }
    }
    if (tid != 0) return;
    finish_msg(p.layout, N, out_q, orig_mean, p.est, p.body, p.msg);
}

// ---- stream bank: per stream, the smoothing stack + FK + message of one frame ---------------------
// Replaces, for S streams at once, the tail of Estimator.add_xx_to_row_hist_and_make_prediction
// (estimator.py:108-118: de-normalise, push onto the smoothing history -- padded with the newest prediction on a
// cold start --, stack oldest..newest) and Estimator.msg_from_pred (:122-137: FK per row, message, the
// 6-float hand/elbow tail of every row).  One wave per stream, lane i = smoothing row i (oldest first,
// smooth <= 64); sums over rows by wave shuffles.
template <typename TMsg>
__global__ __launch_bounds__(256) void ape_stream_post_kernel(const StreamPostParams p) {
    stream_post<TMsg, false>(p, (int)blockIdx.x, 0, 1);
}
// ... a stream's stack dealt over `chunks` workgroups (stream_post, SPLIT): a few streams with tall stacks
template <typename TMsg>
__global__ __launch_bounds__(256) void ape_stream_post_split_kernel(const StreamPostParams p, const int chunks) {
    stream_post<TMsg, true>(p, (int)blockIdx.x / chunks, (int)blockIdx.x % chunks, chunks);
}
// ... and the form for banks without stacking: 64 streams per workgroup (stream_post_wide)
template <typename TMsg>
__global__ __launch_bounds__(256) void ape_stream_post_wide_kernel(const StreamPostParams p) {
    stream_post_wide<TMsg>(p, (int)blockIdx.x * 64);
}

template <typename TIn, typename TOut>
hipError_t launch_fk(const FkParams& p, hipStream_t stream) {
    hipLaunchKernelGGL((ape_fk3_kernel<TIn, TOut>), dim3((p.N + FK3_ROWS - 1) / FK3_ROWS), dim3(128), 0, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t ape_launch_fk(const FkParams& p, int preds_dtype, int est_dtype, hipStream_t stream) {
    if (preds_dtype == APE_F32 && est_dtype == APE_F32) return launch_fk<float, float>(p, stream);
    if (preds_dtype == APE_F32 && est_dtype == APE_F64) return launch_fk<float, double>(p, stream);
    if (preds_dtype == APE_F64 && est_dtype == APE_F32) return launch_fk<double, float>(p, stream);
    if (preds_dtype == APE_F64 && est_dtype == APE_F64) return launch_fk<double, double>(p, stream);
    return hipErrorInvalidValue;
}

hipError_t ape_launch_msg_reduce(const MsgParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(ape_msg_kernel, dim3(1), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t ape_launch_stream_post(const StreamPostParams& p, hipStream_t stream) {
    if (p.smooth == 1 && p.n_mc == 1 && p.S >= 8) {        // no stacking: lanes = streams
        const int wide = (p.S + 63) / 64;
        if (p.msg_dtype == APE_F32) hipLaunchKernelGGL(ape_stream_post_wide_kernel<float>, dim3(wide), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(ape_stream_post_wide_kernel<double>, dim3(wide), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    const int grid = p.S;                                  // one workgroup (four role waves per row group) per stream
    const int chunks = ape_stream_post_chunks(p.smooth * p.n_mc);
    if (p.part != nullptr && chunks > 1) {                 // (ape_api.hip hands the workspace over when the bank is small enough)
        if (p.msg_dtype == APE_F32) hipLaunchKernelGGL(ape_stream_post_split_kernel<float>, dim3(grid * chunks), dim3(256), 0, stream, p, chunks);
        else hipLaunchKernelGGL(ape_stream_post_split_kernel<double>, dim3(grid * chunks), dim3(256), 0, stream, p, chunks);
        return hipGetLastError();
    }
    if (p.msg_dtype == APE_F32) hipLaunchKernelGGL(ape_stream_post_kernel<float>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(ape_stream_post_kernel<double>, dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}
