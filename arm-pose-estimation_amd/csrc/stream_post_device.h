// The per-stream smoothing + message step of the stream bank as a device function (fk.hip's ape_stream_post_kernel: one workgroup per
// stream; in round 4 it was also tried as the tail of lstm_mc_small.hip's launch -- no gain over its own launch, not kept).
// Replaces the tail of Estimator.add_xx_to_row_hist_and_make_prediction and Estimator.msg_from_pred (reference estimate/estimator.py:108-137).
// float64 with separate roundings for a * b + c, like numpy: contraction is off inside every function, whatever the including
// translation unit does.  Call with 256 threads; `s` = the stream.
#pragma once
#include "ape_internal.h"
#include "fk_device.h"
#include "../../include/ape_hip.h"

namespace ape_postdev {

using namespace ape_fkdev;

__device__ __forceinline__ void put_q(double* d, const Quat q) { d[0] = q.w; d[1] = q.x; d[2] = q.y; d[3] = q.z; }
__device__ __forceinline__ void put_v(double* d, const Vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
__device__ __forceinline__ Vec3 vadd(const Vec3 a, const Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }

// The message from its parts (compose_msg.py:48-108): for N > 1 the sign-aligned mean quaternions `out_q`
// (lower arm, upper arm, hips) with the origins recomputed from them (or plain origin means for the position
// layout), for N == 1 a copy of est row `e0`.  Fixed joint layout, compose_msg.py:72-78.
__device__ inline void finish_msg(int layout, int N, const double (&out_q)[3][4], const double (&orig_mean)[9],
                           const double* e0, const double* body, double* m) {
#pragma clang fp contract(off)
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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return __shfl(v, 0, 64);
}

// One workgroup per stream, a wave per chain (as ape_fk3_kernel: one wave computing a whole row is a chain of ~1400 f64
// instructions): wave 0 the lower arm's 6D -> quaternion chain, its rotated bone and its quaternion mean, wave 1 the upper arm's,
// wave 2 the hips' and the shoulder origin, wave 3 the copy into the ring, the columns that pass through and -- behind a barrier
// -- the two sums that join the chains (hand / elbow origins: the 6-float tail of every row).  Sums over rows: a lane's rows in
// order, then the wave shuffle tree.
//
// SPLIT (round 4): a stream's stack dealt over `C` workgroups, one 64-row chunk each -- for a few streams with tall stacks (one
// estimator's frame at 60 samples x smooth 5 = 300 rows, the watch-only default 25 x 10 = 250).  The float64 chains keep a SIMD's
// issue port busy by themselves (~5 cycles per instruction: a second wave on the same SIMD buys nothing -- four row groups in ONE
// 1024-thread workgroup were measured at 18.1 us per 300-row frame against 15.9 for the plain loop over five chunks), so more rows
// at once need more CUs.  Chunk 0 holds rows 0 .. 63; chunk c > 0 rows 64 + 63 (c - 1) ... on lanes 0 .. 62 and, on lane 63, row 0
// once more: its quaternions are the sign reference of the means (transformations.py:44).  Partial sums go to `p.part`, the last
// workgroup of a stream to arrive (ticket in `p.part_cnt`, release / acquire fences around it) adds them in chunk order and
// writes the message; the counter is left at zero.
template <typename TMsg, bool SPLIT>
__device__ inline void stream_post(const StreamPostParams& p, const int s, const int chunk, const int C) {
#pragma clang fp contract(off)

    __shared__ double rot[64][3][3];                        // per row of a 64-row chunk: rotated lower-arm bone, upper-arm bone, shoulder origin
    __shared__ double e0_s[21];                             // row 0 of the stack (the N == 1 message)
    __shared__ double ref_s[3][4];                          // row 0's quaternions: the sign reference of the means
    __shared__ double outq_s[3][4];
    __shared__ double omean_s[9];
    __shared__ int last_s;
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int M = p.n_mc, N = p.smooth * M, O = p.O;
    const bool hips = p.layout != APE_LAYOUT_ORI_CAL_LARM_UARM;
    const bool full = p.layout == APE_LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS;
    const int nq = hips ? 3 : 2;
    const int qc[3] = {hips ? 9 : 6, hips ? 13 : 10, 17};
    const int c_in[3] = {full ? 3 : 0, full ? 12 : 6, full ? 18 : 12};     // first input column of the role's chain
    const double wgt = 1.0 / (double)N;
    double acc[4] = {0, 0, 0, 0};
    double osum[6] = {0, 0, 0, 0, 0, 0};
    if (threadIdx.x < 21) e0_s[threadIdx.x] = 0.0;
    if (threadIdx.x < 12) { outq_s[threadIdx.x >> 2][threadIdx.x & 3] = 0.0; ref_s[threadIdx.x >> 2][threadIdx.x & 3] = 0.0; }
    if (threadIdx.x < 9) omean_s[threadIdx.x] = 0.0;
    // host frames: the health of the regressor launch in front of this kernel travels with the datagrams
    if (p.status_out != nullptr && s == 0 && chunk == 0 && threadIdx.x == 255)
        *p.status_out = __hip_atomic_load(p.status_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // stacked row i = (prediction j of the last `smooth`, oldest first; Monte-Carlo sample k of it): lanes take
    // rows lane, lane + 64, ... (trip count uniform over the workgroup); SPLIT: the one chunk of this workgroup
    const int first = SPLIT ? (chunk == 0 ? 0 : 64 + 63 * (chunk - 1)) : 0;
    const int past = SPLIT ? min(N, chunk == 0 ? 64 : first + 63) : N;
    for (int base = first; base < past; base += 64) {
        const bool refrow = SPLIT && chunk > 0 && lane == 63;       // row 0 once more, for its quaternions only
        const int i = refrow ? 0 : base + lane;
        const bool act = i < past && !refrow;
        const float* src = nullptr;
        bool fresh = false;
        int j = 0, k = 0;
        if (act || refrow) {
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
            if (act || refrow) {
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
            if (act || refrow) {
                Vec3 uo{p.body[6], p.body[7], p.body[8]};
                if (hips) {
                    const Quat hq = hips_quat(load(c_in[2]), load(c_in[2] + 1));
                    uo = qrot(hq, uo);
                    q[0] = hq.w; q[1] = hq.x; q[2] = hq.y; q[3] = hq.z;
                }
                rot[lane][2][0] = uo.x; rot[lane][2][1] = uo.y; rot[lane][2][2] = uo.z;
                if (full && act) { osum[0] += uo.x; osum[1] += uo.y; osum[2] += uo.z; }
                if (i == 0 && act && hips) { e0_s[6] = uo.x; e0_s[7] = uo.y; e0_s[8] = uo.z; }
            }
        } else if (act && fresh) {                          // keep the prediction for the next frames
            float* dst = p.yring + (((size_t)s * p.smooth + (p.cold ? j : p.pos)) * M + k) * O;
#pragma unroll
            for (int c = 0; c < 20; ++c)
                if (c < O) dst[c] = src[c];
        }
        if (role < nq && i == 0 && (act || refrow)) {       // row 0: lane 0 of the first chunk (SPLIT: lane 63 of the others)
#pragma unroll
            for (int c = 0; c < 4; ++c) ref_s[role][c] = q[c];
            if (act) {
#pragma unroll
                for (int c = 0; c < 4; ++c) e0_s[qc[role] + c] = q[c];
            }
        }
        __syncthreads();                                    // the chunk's rotated vectors and row 0's quaternions are in LDS
        if (role < nq && N > 1) {
            const double r0 = ref_s[role][0], r1 = ref_s[role][1], r2 = ref_s[role][2], r3 = ref_s[role][3];
            const double d = fma(q[3], r3, fma(q[2], r2, fma(q[1], r1, q[0] * r0)));   // the sign rule of ape_msg_kernel
            const double sg = !act ? 0.0 : ((i > 0 && d < 0.0) ? -wgt : wgt);
            acc[0] += q[0] * sg; acc[1] += q[1] * sg; acc[2] += q[2] * sg; acc[3] += q[3] * sg;
        }
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
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (role < nq) { a0 = wave_sum(acc[0]); a1 = wave_sum(acc[1]); a2 = wave_sum(acc[2]); a3 = wave_sum(acc[3]); }
        double om[6] = {0, 0, 0, 0, 0, 0};
        const int n_o = (role == 2) ? 3 : 6, o0 = (role == 2) ? 6 : 0;
        if (full && role >= 2)                              // compose_msg.py:26-29: plain means of the three origins
            for (int c = 0; c < n_o; ++c) om[c] = wave_sum(osum[c]);
        if constexpr (SPLIT) {
            // partial sums out, release, ticket; the last chunk to arrive acquires and adds them in chunk order
            double* mine = p.part + ((size_t)s * C + chunk) * 21;
            if (lane == 0 && role < nq) { mine[role * 4] = a0; mine[role * 4 + 1] = a1; mine[role * 4 + 2] = a2; mine[role * 4 + 3] = a3; }
            if (lane == 0 && full && role >= 2)
                for (int c = 0; c < n_o; ++c) mine[12 + o0 + c] = om[c];
            if (p.done_out != nullptr) __threadfence_system();      // (this chunk's tail rows may sit in host memory: out before the ticket)
            else __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned tk = __hip_atomic_fetch_add(p.part_cnt + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last_s = (tk == (unsigned)(C - 1)) ? 1 : 0;
                if (last_s) {
                    __hip_atomic_store(p.part_cnt + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __threadfence();
                }
            }
            __syncthreads();
            if (!last_s) return;
            const double* all = p.part + (size_t)s * C * 21;
            a0 = a1 = a2 = a3 = 0;
            if (lane == 0 && role < nq)
                for (int c = 0; c < C; ++c) {
                    a0 += all[c * 21 + role * 4]; a1 += all[c * 21 + role * 4 + 1]; a2 += all[c * 21 + role * 4 + 2]; a3 += all[c * 21 + role * 4 + 3];
                }
            if (lane == 0 && full && role >= 2)
                for (int k = 0; k < n_o; ++k) {
                    om[k] = 0;
                    for (int c = 0; c < C; ++c) om[k] += all[c * 21 + 12 + o0 + k];
                }
        }
        if (lane == 0 && role < nq) {
            const double nrm = sqrt(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3);
            outq_s[role][0] = a0 / nrm; outq_s[role][1] = a1 / nrm; outq_s[role][2] = a2 / nrm; outq_s[role][3] = a3 / nrm;
        }
        if (lane == 0 && full && role >= 2)
            for (int c = 0; c < n_o; ++c) omean_s[o0 + c] = om[c] / (double)N;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
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
    if (p.done_out != nullptr) {                     // host frames: outputs first (system scope), then the word the host is watching
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(p.done_out + s, p.done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The same step for banks WITHOUT stacking (smooth = 1, one sample per stream: the deterministic bank of configs[3]): lane = stream, 64 streams
// per workgroup, the four role waves as above.  stream_post() gives every stream a workgroup of its own whose waves then run with ONE
// active lane -- 4096 waves at 1024 streams, four per SIMD, each a chain of ~500 float64 instructions: the kernel was bound by their
// issue slots (13.6 us at 1024 streams against 6 us for one stream).  Same device functions in the same order: bit-identical outputs.
template <typename TMsg>
__device__ inline void stream_post_wide(const StreamPostParams& p, const int s0) {
#pragma clang fp contract(off)

    __shared__ double rot[64][3][3];
    __shared__ double e0w[64][21];                          // est row of every stream (the N == 1 message is a copy of it)
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int s = s0 + lane, O = p.O;
    const bool act = s < p.S;
    const bool hips = p.layout != APE_LAYOUT_ORI_CAL_LARM_UARM;
    const bool full = p.layout == APE_LAYOUT_ORI_POS_CAL_LARM_UARM_HIPS;
    const int qc[3] = {hips ? 9 : 6, hips ? 13 : 10, 17};
    const int c_in[3] = {full ? 3 : 0, full ? 12 : 6, full ? 18 : 12};
    if (p.status_out != nullptr && s0 == 0 && threadIdx.x == 255)
        *p.status_out = __hip_atomic_load(p.status_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int c = role; c < 21; c += 4) e0w[lane][c] = 0.0;
    __syncthreads();
    const float* src = p.y_new + (size_t)(act ? s : 0) * O;
    auto load = [&](int c) -> double {
        double v = (double)src[c];
        if (p.yy_m) v = v * p.yy_s[c] + p.yy_m[c];          // estimator.py:108-109
        return v;
    };
    if (role < 2) {
        if (act) {
            double s6[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) s6[c] = load(c_in[role] + c);
            const Quat qq = six_drr_to_quat(s6);
            const Vec3 bone = role ? Vec3{p.body[3], p.body[4], p.body[5]} : Vec3{p.body[0], p.body[1], p.body[2]};
            const Vec3 v = qrot(qq, bone);
            rot[lane][role][0] = v.x; rot[lane][role][1] = v.y; rot[lane][role][2] = v.z;
            e0w[lane][qc[role]] = qq.w; e0w[lane][qc[role] + 1] = qq.x; e0w[lane][qc[role] + 2] = qq.y; e0w[lane][qc[role] + 3] = qq.z;
        }
    } else if (role == 2) {
        if (act) {
            Vec3 uo{p.body[6], p.body[7], p.body[8]};
            if (hips) {
                const Quat hq = hips_quat(load(c_in[2]), load(c_in[2] + 1));
                uo = qrot(hq, uo);
                e0w[lane][17] = hq.w; e0w[lane][18] = hq.x; e0w[lane][19] = hq.y; e0w[lane][20] = hq.z;
                e0w[lane][6] = uo.x; e0w[lane][7] = uo.y; e0w[lane][8] = uo.z;
            }
            rot[lane][2][0] = uo.x; rot[lane][2][1] = uo.y; rot[lane][2][2] = uo.z;
        }
    } else if (act) {                                       // keep the prediction (the one ring slot of a bank without stacking)
        float* dst = p.yring + (size_t)s * O;
#pragma unroll
        for (int c = 0; c < 20; ++c)
            if (c < O) dst[c] = src[c];
    }
    __syncthreads();
    if (role == 3 && act) {
        double e6[6];
        if (full) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { e6[c] = load(c); e6[3 + c] = load(9 + c); }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                e6[3 + c] = rot[lane][1][c] + rot[lane][2][c];              // qrot(uq, uarm_vec) + uo
                e6[c] = rot[lane][0][c] + e6[3 + c];                        // qrot(lq, larm_vec) + lo
            }
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) e0w[lane][c] = e6[c];
        if (p.tail || p.packed) {
            TMsg* t = p.packed ? static_cast<TMsg*>(p.msg) + (size_t)s * 31 + 25 : static_cast<TMsg*>(p.tail) + (size_t)s * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) t[c] = (TMsg)e6[c];
        }
    }
    __syncthreads();
    if (act) {                                              // the message: every role wave writes a quarter of its stream's 25 columns
        double out_q[3][4] = {}, orig_mean[9] = {}, e0[21], m[25];
#pragma unroll
        for (int c = 0; c < 21; ++c) e0[c] = e0w[lane][c];
        finish_msg(p.layout, 1, out_q, orig_mean, e0, p.body, m);
        TMsg* dst = static_cast<TMsg*>(p.msg) + (size_t)s * (p.packed ? 31 : 25);
#pragma unroll
        for (int c = 0; c < 25; ++c)
            if (c / 7 == role) dst[c] = (TMsg)m[c];
    }
    if (p.done_out != nullptr) {
        __threadfence_system();
        __syncthreads();
        if (role == 0 && act) __hip_atomic_store(p.done_out + s, p.done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace ape_postdev
