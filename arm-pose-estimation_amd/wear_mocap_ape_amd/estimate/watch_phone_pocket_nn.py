"""``WatchPhonePocketNN`` -- the ``watch_phone_pocket_lstm`` estimator (reference
``estimate/watch_phone_pocket_nn.py:12-112``): smartwatch on the wrist + phone in the pocket,
22 features -> 2x256 LSTM -> 14 targets (two 6D rotations + hips sin/cos)."""
import math

import numpy as np
import torch

from wear_mocap_ape_amd.data_types import messaging
from wear_mocap_ape_amd.data_types.bone_map import BoneMap
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.estimate.estimator import Estimator
from wear_mocap_ape_amd.utility import transformations as ts
from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS

_SW_SENSORS = ["sw_dt"] + [f"sw_{g}_{a}" for g in ("gyro", "lvel", "lacc", "grav") for a in "xyz"]


def _quat(row, lookup, prefix):
    return np.array([row[lookup[f"{prefix}_{c}"]] for c in "wxyz"], dtype=np.float64)


def features_from_row(row, slp) -> np.ndarray:
    """55-float watch+phone message -> float32[22] (watch_phone_pocket_nn.py:41-96)."""
    r_pres = row[slp["sw_pres"]] - row[slp["sw_init_pres"]]
    north = ts.north_quat_from_forward(_quat(row, slp, "sw_forward"))
    sw_cal_g = ts.android_to_global(_quat(row, slp, "sw_rotvec"), north)
    # phone orientation relative to its calibration pose -> hip yaw
    ph_rot_g = ts.android_to_global(_quat(row, slp, "ph_rotvec"), north)
    ph_fwd_g = ts.android_to_global(_quat(row, slp, "ph_forward"), north)
    hips_y = ts.y_rotation_of(ts.quat_mul(ph_rot_g, ts.quat_invert(ph_fwd_g)))
    return np.hstack([
        [row[slp[n]] for n in _SW_SENSORS],
        ts.quat_to_six_drr(sw_cal_g),
        r_pres,
        math.sin(hips_y),
        math.cos(hips_y),
    ]).astype(np.float32)


class WatchPhonePocketNN(Estimator):
    def __init__(self,
                 model_hash: str,
                 smooth: int = 1,
                 add_mc_samples=True,
                 monte_carlo_samples=25,
                 bonemap: BoneMap = None,
                 tag: str = "NN POCKET PHONE"):
        self.__tag = tag
        self.__mc_samples = monte_carlo_samples
        self.__slp = messaging.WATCH_PHONE_IMU_LOOKUP
        self.__nn_model, params = nn_models.load_deployed_model_from_hash(hash_str=model_hash)
        super().__init__(
            x_inputs=NNS_INPUTS[params["x_inputs_n"]],
            y_targets=NNS_TARGETS[params["y_targets_n"]],
            smooth=smooth,
            normalize=params["normalize"],
            seq_len=params["sequence_len"],
            add_mc_samples=add_mc_samples,
            tag=tag,
            bonemap=bonemap
        )

    def _hip_model(self):
        return self.__nn_model

    def _frame_samples(self):
        return self.__mc_samples

    _parse_kind = _hip.PARSE_WATCH_PHONE_POCKET

    def parse_row_to_xx(self, row):
        return features_from_row(row, self.__slp)

    def make_prediction_from_row_hist(self, xx):
        """normalised window float64 [T,22] -> float32 [n_mc,14]: last step of the MC forward
        (watch_phone_pocket_nn.py:98-112)."""
        xx = torch.tensor(xx[None, :, :], dtype=torch.float32)
        t_preds = self.__nn_model.monte_carlo_predictions(x=xx, n_samples=self.__mc_samples, last_step_only=True)
        return t_preds.numpy()[:, -1, :]
