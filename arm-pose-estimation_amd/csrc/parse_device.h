// The feature builder of one raw message as a device function: shared by parse_rows.hip's batched kernel and by the single-launch
// frame of lstm_mc_small.hip (one workgroup per stream builds the new row beside the regressor's prologue).  Replaces parse_row_to_xx
// of the three estimators (reference estimate/watch_phone_pocket_nn.py:41-96, watch_only.py:46-82, watch_phone_uarm_nn.py:43-105) and
// the helpers of utility/transformations.py they call (see parse_rows.hip).  float64 with separate roundings for a * b + c (the
// reference computes on Python floats): contraction is off inside every function, whatever the including translation unit does.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ape_hip.h"
#include "angle_device.h"

namespace ape_parsedev {

struct Q { double w, x, y, z; };

__device__ __forceinline__ Q qmul(const Q a, const Q b) {
#pragma clang fp contract(off)
    return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
             a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
             a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q qconj(const Q q) { return Q{q.w, -q.x, -q.y, -q.z}; }
__device__ __forceinline__ Q qinv(const Q q) {
#pragma clang fp contract(off)
    const double n = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return Q{q.w / n, -q.x / n, -q.y / n, -q.z / n};
}
// android (X east, Y north, Z up) -> global (X right, Y up, Z forward): [-w, x, z, y]
__device__ __forceinline__ Q no_north(const Q q) { return Q{-q.w, q.x, q.z, q.y}; }
// q * (0,0,1): the rotated forward axis; its azimuth is atan2(x, z)
__device__ __forceinline__ Q fwd_of(const Q q) {
#pragma clang fp contract(off)
    return qmul(qmul(q, Q{0.0, 0.0, 0.0, 1.0}), qconj(q));
}
// the y rotation by MINUS the azimuth of q's forward axis (reduce_global_quat_to_y_rot -> euler_to_quat(0, -a, 0)): angle_device.h
__device__ __forceinline__ Q north_of(const Q q) {
#pragma clang fp contract(off)
    const Q t = fwd_of(q);
    const ape_angledev::CS h = ape_angledev::half_of_atan2(t.x, t.z);
    return Q{h.c, 0.0, -h.s, 0.0};
}
__device__ __forceinline__ Q rd(const float* row, int i) {
#pragma clang fp contract(off)
    return Q{(double)row[i], (double)row[i + 1], (double)row[i + 2], (double)row[i + 3]};
}
// first two columns of the rotation matrix, row-interleaved [m11,m12,m21,m22,m31,m32] (transforms3d formula)
__device__ __forceinline__ void six_drr(const Q q, double* o) {
#pragma clang fp contract(off)
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
#pragma clang fp contract(off)
    return width == 28 ? Cols{5, 23, -1, -1, 27} : Cols{5, 46, 28, 50, 54};
}
// sw_dt, gyro, lvel, lacc at 0,10..18; grav at 20..22 (pressure sits at 19)
__device__ __forceinline__ int sw_sensor_col(int i) { return i == 0 ? 0 : (i <= 9 ? 9 + i : 10 + i); }
// ph gyro, lvel, lacc at 33..41; grav at 43..45
__device__ __forceinline__ int ph_sensor_col(int i) { return i < 9 ? 33 + i : 34 + i; }

// row: the message (55 or 28 floats, host byte order); xx: the features (22 / 20 / 38 doubles by `kind`)
__device__ inline void parse_row(const float* row, int width, int kind, double* xx) {
#pragma clang fp contract(off)
    const Cols cl = cols_of(width);
    int o = 0;
#pragma unroll
    for (int i = 0; i < 13; ++i) xx[o++] = (double)row[sw_sensor_col(i)];
    const double r_pres = (double)row[19] - (double)row[cl.init_pres];
    const Q sw_fwd = rd(row, cl.fwd), sw_rot = rd(row, cl.rot);
    Q north = north_of(no_north(sw_fwd));
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
            const Q t = fwd_of(qmul(ph_rot_g, qinv(ph_fwd_g)));
            const ape_angledev::CS a = ape_angledev::of_atan2(t.x, t.z);
            xx[o++] = a.s;
            xx[o++] = a.c;
        }
    }
}

}  // namespace ape_parsedev
