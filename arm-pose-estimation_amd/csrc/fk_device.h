// Device functions of the forward-kinematics post-filter shared by fk.hip and the latency kernel's fused tail
// (lstm_cluster_small.hip): quaternion product / rotation (utility/transformations.py:83-149), 6D rotation -> unit quaternion
// (:602-637, :521-545 in closed form, see fk.hip) and the hips quaternion (:177-179).  float64 with separate roundings for
// a * b + c, like numpy: contraction is off inside every function, whatever the including translation unit does.
#pragma once
#include <hip/hip_runtime.h>
#include "angle_device.h"

namespace ape_fkdev {

struct Quat { double w, x, y, z; };
struct Vec3 { double x, y, z; };

__device__ __forceinline__ Quat qmul(const Quat a, const Quat b) {   // transformations.py:140-145
#pragma clang fp contract(off)
    return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}

__device__ __forceinline__ Vec3 qrot(const Quat q, const Vec3 v) {   // q (0,v) q*
#pragma clang fp contract(off)
    const Quat t = qmul(q, Quat{0.0, v.x, v.y, v.z});
    const Quat o = qmul(t, Quat{q.w, -q.x, -q.y, -q.z});
    return Vec3{o.x, o.y, o.z};
}

// s = [m11,m12,m21,m22,m31,m32] -> unit quaternion, w >= 0.  No zero-norm guard: NaN propagates
// exactly like the reference.
__device__ inline Quat six_drr_to_quat(const double* s) {
#pragma clang fp contract(off)
    const double a1x = s[0], a1y = s[2], a1z = s[4];
    const double a2x = s[1], a2y = s[3], a2z = s[5];
    const double n1 = sqrt(a1x * a1x + a1y * a1y + a1z * a1z);
    const double b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const double d = b1x * a2x + b1y * a2y + b1z * a2z;
    const double ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
    const double n2 = sqrt(ux * ux + uy * uy + uz * uz);
    const double b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    const double b3x = b1y * b2z - b1z * b2y;
    const double b3y = b1z * b2x - b1x * b2z;
    const double b3z = b1x * b2y - b1y * b2x;
    // R = [b1 b2 b3] as columns
    const double m00 = b1x, m01 = b2x, m02 = b3x;
    const double m10 = b1y, m11 = b2y, m12 = b3y;
    const double m20 = b1z, m21 = b2z, m22 = b3z;
    const double tr = m00 + m11 + m22;
    Quat q;
    if (!(tr < m00) && !(tr < m11) && !(tr < m22)) {          // trace is the pivot (also the NaN path)
        const double s4 = 2.0 * sqrt(tr + 1.0);
        q = Quat{0.25 * s4, (m21 - m12) / s4, (m02 - m20) / s4, (m10 - m01) / s4};
    } else if (m00 >= m11 && m00 >= m22) {
        const double s4 = 2.0 * sqrt(1.0 + m00 - m11 - m22);
        q = Quat{(m21 - m12) / s4, 0.25 * s4, (m01 + m10) / s4, (m02 + m20) / s4};
    } else if (m11 >= m22) {
        const double s4 = 2.0 * sqrt(1.0 + m11 - m00 - m22);
        q = Quat{(m02 - m20) / s4, (m01 + m10) / s4, 0.25 * s4, (m12 + m21) / s4};
    } else {
        const double s4 = 2.0 * sqrt(1.0 + m22 - m00 - m11);
        q = Quat{(m10 - m01) / s4, (m02 + m20) / s4, (m12 + m21) / s4, 0.25 * s4};
    }
    if (q.w < 0.0) q = Quat{-q.w, -q.x, -q.y, -q.z};           // transformations.py:543-544
    return q;
}

__device__ __forceinline__ Quat hips_quat(double sn, double cs) {   // transformations.py:177-179
#pragma clang fp contract(off)
    const ape_angledev::CS h = ape_angledev::half_of_atan2(sn, cs);   // (cos, sin) of atan2(sn, cs) / 2, without the angle
    return Quat{h.c, 0.0, h.s, 0.0};
}

}  // namespace ape_fkdev
