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

// numpy evaluates a*b+c with two roundings; keep the device arithmetic the same
#pragma clang fp contract(off)

namespace {

using namespace ape_fkdev;

__device__ __forceinline__ void put_q(double* d, const Quat q) { d[0] = q.w; d[1] = q.x; d[2] = q.y; d[3] = q.z; }
__device__ __forceinline__ void put_v(double* d, const Vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
__device__ __forceinline__ Vec3 vadd(const Vec3 a, const Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }

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

// The message from its parts (compose_msg.py:48-108): for N > 1 the sign-aligned mean quaternions `out_q`
// (lower arm, upper arm, hips) with the origins recomputed from them (or plain origin means for the position
// layout), for N == 1 a copy of est row `e0`.  Fixed joint layout, compose_msg.py:72-78.
__device__ void finish_msg(int layout, int N, const double (&out_q)[3][4], const double (&orig_mean)[9],
                           const double* e0, const double* body, double* m) {
    const bool hips = layout != APE_LAYOUT_ORI_CAL_LARM_UARM;
    const int qc_l = hips ? 9 : 6, qc_u = hips ? 13 : 10;
    const Vec3 larm_vec{body[0], body[1], body[2]};
    const Vec3 uarm_vec{body[3], body[4], body[5]};
    const Vec3 uarm_orig_rh{body[6], body[7], body[8]};
    Quat lq, uq, hq{1.0, 0.0, 0.0, 0.0};
    Vec3 ho, lo, uo = uarm_orig_rh;
    if (N > 1) {
        lq = Quat{out_q[0][0], out_q[0][1], out_q[0][2], out_q[0][3]};
        uq = Quat{out_q[1][0], out_q[1][1], out_q[1][2], out_q[1][3]};
        if (hips) hq = Quat{out_q[2][0], out_q[2][1], out_q[2][2], out_q[2][3]};
        if (layout == APE_LAYOUT_ORI_CAL_LARM_UARM_HIPS) {        // compose_msg.py:58-61
            uo = qrot(hq, uarm_orig_rh);
            lo = vadd(qrot(uq, uarm_vec), uo);
            ho = vadd(qrot(lq, larm_vec), lo);
        } else if (layout == APE_LAYOUT_ORI_CAL_LARM_UARM) {      // compose_msg.py:92-94
            lo = vadd(qrot(uq, uarm_vec), uarm_orig_rh);
            ho = vadd(qrot(lq, larm_vec), lo);
        } else {
            ho = Vec3{orig_mean[0], orig_mean[1], orig_mean[2]};
            lo = Vec3{orig_mean[3], orig_mean[4], orig_mean[5]};
            uo = Vec3{orig_mean[6], orig_mean[7], orig_mean[8]};
        }
    } else {                                                        // single row: copy (compose_msg.py:63-68)
        ho = Vec3{e0[0], e0[1], e0[2]};
        lo = Vec3{e0[3], e0[4], e0[5]};
        lq = Quat{e0[qc_l], e0[qc_l + 1], e0[qc_l + 2], e0[qc_l + 3]};
        uq = Quat{e0[qc_u], e0[qc_u + 1], e0[qc_u + 2], e0[qc_u + 3]};
        if (hips) {
            uo = Vec3{e0[6], e0[7], e0[8]};
            hq = Quat{e0[17], e0[18], e0[19], e0[20]};
        }
    }
    // (hand rot duplicates the lower-arm quaternion)
    put_q(m + 0, lq); put_v(m + 4, ho); put_q(m + 7, lq); put_v(m + 11, lo);
    put_q(m + 14, uq); put_v(m + 18, uo); put_q(m + 21, hq);
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
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return __shfl(v, 0, 64);
}

// One workgroup per stream, a wave per chain (as ape_fk3_kernel: one wave computing a whole row is a chain of ~1400 f64
// instructions): wave 0 the lower arm's 6D -> quaternion chain, its rotated bone and its quaternion mean, wave 1 the upper arm's,
// wave 2 the hips' and the shoulder origin, wave 3 the copy into the ring, the columns that pass through and -- behind a barrier
// -- the two sums that join the chains (hand / elbow origins: the 6-float tail of every row).  Sums over rows stay what they
// were: a lane's rows in order, then the wave shuffle tree.
template <typename TMsg>
__global__ __launch_bounds__(256) void ape_stream_post_kernel(const StreamPostParams p) {
    __shared__ double rot[64][3][3];                        // per row of a 64-row chunk: rotated lower-arm bone, upper-arm bone, shoulder origin
    __shared__ double e0_s[21];                             // row 0 of the stack (the N == 1 message)
    __shared__ double outq_s[3][4];
    __shared__ double omean_s[9];
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int s = blockIdx.x;
    const int M = p.n_mc, N = p.smooth * M, O = p.O;
    const bool hips = p.layout != APE_LAYOUT_ORI_CAL_LARM_UARM;
    const bool full = p.layout == APE_LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS;
    const int nq = hips ? 3 : 2;
    const int qc[3] = {hips ? 9 : 6, hips ? 13 : 10, 17};
    const int c_in[3] = {full ? 3 : 0, full ? 12 : 6, full ? 18 : 12};     // first input column of the role's chain
    const double wgt = 1.0 / (double)N;
    double ref[4] = {0, 0, 0, 0};                           // row 0's quaternion of this role: the sign reference of the mean
    double acc[4] = {0, 0, 0, 0};
    double osum[6] = {0, 0, 0, 0, 0, 0};
    if (threadIdx.x < 21) e0_s[threadIdx.x] = 0.0;
    if (threadIdx.x < 12) outq_s[threadIdx.x >> 2][threadIdx.x & 3] = 0.0;
    if (threadIdx.x < 9) omean_s[threadIdx.x] = 0.0;
    // host frames: the health of the regressor launch in front of this kernel travels with the datagrams
    if (p.status_out != nullptr && s == 0 && threadIdx.x == 255)
        *p.status_out = __hip_atomic_load(p.status_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // stacked row i = (prediction j of the last `smooth`, oldest first; Monte-Carlo sample k of it): lanes take
    // rows lane, lane + 64, ... (trip count uniform over the workgroup)
    for (int base = 0; base < N; base += 64) {
        const int i = base + lane;
        const bool act = i < N;
        const float* src = nullptr;
        bool fresh = false;
        int j = 0, k = 0;
        if (act) {
            j = i / M; k = i - j * M;
            // the newest prediction sits in ring slot `pos`, the oldest one slot further
            fresh = p.cold || j == p.smooth - 1;
            const int slot = (p.pos + 1 + j) % p.smooth;
            src = fresh ? p.y_new + ((size_t)s * M + k) * O : p.yring + (((size_t)s * p.smooth + slot) * M + k) * O;
        }
        auto load = [&](int c) -> double {
            double v = (double)src[c];
            if (p.yy_m) v = v * p.yy_s[c] + p.yy_m[c];      // estimator.py:108-109
            return v;
        };
        double q[4] = {0, 0, 0, 0};
        if (role < 2) {
            if (act) {
                double s6[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) s6[c] = load(c_in[role] + c);
                const Quat qq = six_drr_to_quat(s6);
                const Vec3 bone = role ? Vec3{p.body[3], p.body[4], p.body[5]} : Vec3{p.body[0], p.body[1], p.body[2]};
                const Vec3 v = qrot(qq, bone);
                rot[lane][role][0] = v.x; rot[lane][role][1] = v.y; rot[lane][role][2] = v.z;
                q[0] = qq.w; q[1] = qq.x; q[2] = qq.y; q[3] = qq.z;
            }
        } else if (role == 2) {
            if (act) {
                Vec3 uo{p.body[6], p.body[7], p.body[8]};
                if (hips) {
                    const Quat hq = hips_quat(load(c_in[2]), load(c_in[2] + 1));
                    uo = qrot(hq, uo);
                    q[0] = hq.w; q[1] = hq.x; q[2] = hq.y; q[3] = hq.z;
                }
                rot[lane][2][0] = uo.x; rot[lane][2][1] = uo.y; rot[lane][2][2] = uo.z;
                if (full) { osum[0] += uo.x; osum[1] += uo.y; osum[2] += uo.z; }
                if (i == 0 && hips) { e0_s[6] = uo.x; e0_s[7] = uo.y; e0_s[8] = uo.z; }
            }
        } else if (act && fresh) {                          // keep the prediction for the next frames
            float* dst = p.yring + (((size_t)s * p.smooth + (p.cold ? j : p.pos)) * M + k) * O;
#pragma unroll
            for (int c = 0; c < 20; ++c)
                if (c < O) dst[c] = src[c];
        }
        if (role < nq) {
            if (base == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) ref[c] = __shfl(q[c], 0, 64);
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) e0_s[qc[role] + c] = q[c];
                }
            }
            if (N > 1) {
                const double d = fma(q[3], ref[3], fma(q[2], ref[2], fma(q[1], ref[1], q[0] * ref[0])));   // the sign rule of ape_msg_kernel
                const double sg = !act ? 0.0 : ((i > 0 && d < 0.0) ? -wgt : wgt);
                acc[0] += q[0] * sg; acc[1] += q[1] * sg; acc[2] += q[2] * sg; acc[3] += q[3] * sg;
            }
        }
        __syncthreads();                                    // the chunk's rotated vectors are in LDS
        if (role == 3 && act) {
            double e6[6];
            if (full) {                                     // hand and lower-arm positions are network outputs (estimate_joints.py:20-45)
#pragma unroll
                for (int c = 0; c < 3; ++c) { e6[c] = load(c); e6[3 + c] = load(9 + c); }
#pragma unroll
                for (int c = 0; c < 6; ++c) osum[c] += e6[c];
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    e6[3 + c] = rot[lane][1][c] + rot[lane][2][c];          // qrot(uq, uarm_vec) + uo
                    e6[c] = rot[lane][0][c] + e6[3 + c];                    // qrot(lq, larm_vec) + lo
                }
            }
            if (i == 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) e0_s[c] = e6[c];
            }
            if (p.tail || p.packed) {                       // estimator.py:131-137: est[i, :6] of every row
                TMsg* t = p.packed ? static_cast<TMsg*>(p.msg) + (size_t)s * (25 + 6 * N) + 25 + (size_t)i * 6
                                   : static_cast<TMsg*>(p.tail) + ((size_t)s * N + i) * 6;
#pragma unroll
                for (int c = 0; c < 6; ++c) t[c] = (TMsg)e6[c];
            }
        }
        __syncthreads();                                    // ... and read: the next chunk may overwrite them
    }
    if (N > 1) {
        if (role < nq) {
            const double a0 = wave_sum(acc[0]), a1 = wave_sum(acc[1]), a2 = wave_sum(acc[2]), a3 = wave_sum(acc[3]);
            const double nrm = sqrt(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3);
            if (lane == 0) { outq_s[role][0] = a0 / nrm; outq_s[role][1] = a1 / nrm; outq_s[role][2] = a2 / nrm; outq_s[role][3] = a3 / nrm; }
        }
        if (full && role >= 2) {                            // compose_msg.py:26-29: plain means of the three origins
            const int n_o = (role == 2) ? 3 : 6, o0 = (role == 2) ? 6 : 0;
            for (int c = 0; c < n_o; ++c) {
                const double m_c = wave_sum(osum[c]) / (double)N;
                if (lane == 0) omean_s[o0 + c] = m_c;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double out_q[3][4], orig_mean[9], e0[21], m[25];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) out_q[k][c] = outq_s[k][c];
#pragma unroll
    for (int c = 0; c < 9; ++c) orig_mean[c] = omean_s[c];
#pragma unroll
    for (int c = 0; c < 21; ++c) e0[c] = e0_s[c];
    finish_msg(p.layout, N, out_q, orig_mean, e0, p.body, m);
    TMsg* dst = static_cast<TMsg*>(p.msg) + (size_t)s * (p.packed ? 25 + 6 * N : 25);
#pragma unroll
    for (int c = 0; c < 25; ++c) dst[c] = (TMsg)m[c];
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
    const int grid = p.S;                                  // one workgroup (four role waves) per stream
    if (p.msg_dtype == APE_F32) hipLaunchKernelGGL(ape_stream_post_kernel<float>, dim3(grid), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(ape_stream_post_kernel<double>, dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}
