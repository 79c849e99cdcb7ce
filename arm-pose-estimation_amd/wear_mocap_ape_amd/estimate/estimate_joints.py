"""NN targets -> arm pose rows ("FK").  Same function and row layouts as the reference's
``estimate/estimate_joints.py:16-92``; the arithmetic runs in ``ape_fk_kernel`` (csrc/fk.hip)."""
import numpy as np

from wear_mocap_ape_amd.estimate import _post
from wear_mocap_ape_amd.utility.names import NNS_TARGETS, TARGET_LAYOUT


def arm_pose_from_nn_targets(preds: np.array, body_measurements: np.array, y_targets: NNS_TARGETS):
    """preds [N,O], body_measurements [1,9] = [larm_vec, uarm_vec, uarm_orig_rh] -> est float64
    [N,21] (targets with hips) or [N,14] (without): estimate_joints.py:48-71 / :74-92 / :20-45."""
    layout = TARGET_LAYOUT[y_targets]
    ctx = _post.context(layout)
    with ctx.lock:
        return _post.fk_rows(ctx.handle, layout, ctx.device, preds, body_measurements)
