"""Arm measurements used by the forward-kinematics post-filter.

Only the nine floats of ``Estimator.body_measurements`` are on the hot path (reference
``estimate/estimator.py:57-68``).  The defaults below are the reference's
``data_types/bone_map.py:42-45``; a ``BoneMap`` can also be built from explicit measurements.
Parsing a mocap skeleton XML (bone_map.py:47-99) is host-side configuration and out of scope."""
import numpy as np


class BoneMap:
    DEFAULT_LARM_LEN = 0.22
    DEFAULT_UARM_LEN = 0.26
    # default left shoulder origin relative to hip
    DEFAULT_UARM_ORIG_RH = np.array([-0.1704612, 0.4309841, -0.00670862])

    def __init__(self, left_lower_arm_length: float = DEFAULT_LARM_LEN,
                 left_upper_arm_length: float = DEFAULT_UARM_LEN,
                 left_upper_arm_origin_rh=None):
        self._larm_len = float(left_lower_arm_length)
        self._uarm_len = float(left_upper_arm_length)
        self._uarm_orig = np.array(self.DEFAULT_UARM_ORIG_RH if left_upper_arm_origin_rh is None
                                   else left_upper_arm_origin_rh, dtype=np.float64)

    @property
    def left_lower_arm_length(self):
        return self._larm_len

    @property
    def left_upper_arm_length(self):
        return self._uarm_len

    @property
    def left_upper_arm_origin_rh(self):
        return self._uarm_orig
