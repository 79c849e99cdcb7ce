// cos / sin of atan2(y, x) and of HALF that angle, without the angle.
//
// The reference turns an azimuth into a y-rotation quaternion in two places -- the north calibration of the feature builder
// (utility/transformations.py:200-207 reduce_global_quat_to_y_rot -> :152-174 euler_to_quat) and the hips quaternion of the
// post-filter (:177-179) -- as  a = atan2(y, x);  (cos(a / 2), 0, sin(a / 2), 0),  and the pocket features as sin(a), cos(a)
// (estimate/watch_phone_pocket_nn.py:88-93).  In float64 on the device atan2 -> cos -> sin is three software routines of a few
// hundred dependent instructions each; a row of the feature builder is ONE such chain, issued at an instruction per ~9 cycles, so the
// chain's length is the kernel's duration whatever the grid (round 6, 1024 rows: pocket 7.1 -> 6.3 us, watch 5.3 -> 5.0, upper arm
// 7.3 -> 6.9; profiles/r06_post_in_tail.md).  The half-angle identities give the same numbers from two square roots and three divisions:
//     r = sqrt(x^2 + y^2), c = x / r, |s| = |y| / r;  big = sqrt((1 + |c|) / 2) >= 0.707, small = |s| / (2 big);
//     c >= 0:  cos(a/2) = big,   |sin(a/2)| = small;     c < 0:  cos(a/2) = small, |sin(a/2)| = big;     sin(a/2) carries y's sign
// (|a/2| <= pi/2, so the cosine is never negative; the division is always by the well-conditioned one of the pair).  Each result is
// within a few 1e-16 of the routine's (the tests hold the feature builder to the reference's float32 / float64 outputs at 1e-6 / 1e-12 and
// the messages at 1e-10).  Whatever the identities cannot state -- r = 0 with its signed-zero cases, an overflowing or underflowing
// square, infinities, NaN -- takes the reference's own route through atan2, so degenerate messages give what they always gave.
#pragma once
#include <hip/hip_runtime.h>

namespace ape_angledev {

struct CS { double c, s; };

__device__ __forceinline__ bool plain_radius(double r) { return r > 0.0 && r <= 1.7976931348623157e308; }

// (cos(a / 2), sin(a / 2)) of a = atan2(y, x)
__device__ inline CS half_of_atan2(double y, double x) {
#pragma clang fp contract(off)
    const double r = sqrt(x * x + y * y);
    if (!plain_radius(r)) {
        const double h = 0.5 * atan2(y, x);
        return CS{cos(h), sin(h)};
    }
    const double c = x / r, sa = fabs(y) / r;
    const double big = sqrt(0.5 * (1.0 + fabs(c)));
    const double small = sa / (2.0 * big);
    return c >= 0.0 ? CS{big, copysign(small, y)} : CS{small, copysign(big, y)};
}

// (cos(a), sin(a)) of a = atan2(y, x)
__device__ inline CS of_atan2(double y, double x) {
#pragma clang fp contract(off)
    const double r = sqrt(x * x + y * y);
    if (!plain_radius(r)) {
        const double a = atan2(y, x);
        return CS{cos(a), sin(a)};
    }
    return CS{x / r, y / r};
}

}  // namespace ape_angledev
