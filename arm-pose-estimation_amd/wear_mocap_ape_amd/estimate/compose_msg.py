"""est rows -> the 25-float joint message.  Same function and message layout as the reference's
``estimate/compose_msg.py:13-108``; the sign-aligned quaternion means and the re-run of the
kinematic chain happen in ``ape_msg_kernel`` (csrc/fk.hip)."""
import numpy as np

from wear_mocap_ape_amd.estimate import _post
from wear_mocap_ape_amd.utility.names import NNS_TARGETS, TARGET_LAYOUT


def msg_from_nn_targets_est(est: np.array, body_measure: np.array, y_targets: NNS_TARGETS):
    """est [N,21|14] -> msg float64 [25]:
    hand rot [0:4] (= lower-arm rot), hand orig [4:7], larm rot [7:11], larm orig [11:14],
    uarm rot [14:18], uarm orig [18:21], hips rot [21:25]   (compose_msg.py:72-78)."""
    layout = TARGET_LAYOUT[y_targets]
    ctx = _post.context(layout)
    with ctx.lock:
        return _post.msg_rows(ctx.handle, layout, ctx.device, est, body_measure)
