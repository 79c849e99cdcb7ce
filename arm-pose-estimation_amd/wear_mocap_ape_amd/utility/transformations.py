"""Quaternion helpers of the FEATURE BUILDER (``parse_row_to_xx``), host side.

Scope note: the hot-path quaternion algebra (6D -> quaternion, hips sin/cos -> quaternion,
vector rotation, quaternion averaging: reference ``utility/transformations.py:32-51,83-179,
471-473,521-637``) lives in the HIP kernels of ``csrc/fk.hip`` and is not duplicated here.
This module only holds the per-row calibration math in front of the window
(``android_quat_to_global`` :225-241, ``reduce_global_quat_to_y_rot`` :200-207, ``quat_invert``
:244-254, ``quat_to_6drr_1x6`` :476-518, ``calib_watch_left_to_north_quat`` :182-197), which is
the "next" row f1 of SURVEY.md section 8 and still runs on the host in float64, one row at a time.
All quaternions are ``[w,x,y,z]``."""
import math

import numpy as np


def quat_mul(a, b) -> np.ndarray:
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def quat_conj(q) -> np.ndarray:
    return np.array([q[0], -q[1], -q[2], -q[3]])


def quat_invert(q) -> np.ndarray:
    q = np.asarray(q, dtype=np.float64)
    return quat_conj(q) / float(np.dot(q, q))


def rotate_vec(q, v) -> np.ndarray:
    return quat_mul(quat_mul(q, np.array([0.0, v[0], v[1], v[2]])), quat_conj(q))[1:]


def android_to_global_no_north(q) -> np.ndarray:
    """android (X east, Y north, Z up) -> global (X right, Y up, Z forward, left-handed)"""
    return np.array([-q[0], q[1], q[3], q[2]], dtype=np.float64)


def android_to_global(q, north_quat) -> np.ndarray:
    return quat_mul(north_quat, android_to_global_no_north(q))


def y_rotation_of(q) -> float:
    """azimuth of the rotated forward axis: atan2(x, z) of q * (0,0,1)"""
    fwd = rotate_vec(q, (0.0, 0.0, 1.0))
    return math.atan2(fwd[0], fwd[2])


def y_rot_quat(angle: float) -> np.ndarray:
    return np.array([math.cos(0.5 * angle), 0.0, math.sin(0.5 * angle), 0.0])


def north_quat_from_forward(sw_fwd) -> np.ndarray:
    """rotation about the up axis that aligns the watch's calibration heading with global Z"""
    return y_rot_quat(-y_rotation_of(android_to_global_no_north(sw_fwd)))


def quat_to_six_drr(q) -> np.ndarray:
    """first two columns of the rotation matrix of q, interleaved row-wise:
    [m11, m12, m21, m22, m31, m32]"""
    w, x, y, z = (float(c) for c in q)
    nq = w * w + x * x + y * y + z * z
    if nq < np.finfo(np.float64).eps:
        return np.array([1.0, 0.0, 0.0, 1.0, 0.0, 0.0])
    s = 2.0 / nq
    xs, ys, zs = x * s, y * s, z * s
    return np.array([1.0 - (y * ys + z * zs), x * ys - w * zs,
                     x * ys + w * zs, 1.0 - (x * xs + z * zs),
                     x * zs - w * ys, y * zs + w * xs])
