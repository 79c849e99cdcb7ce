"""``WatchPhoneUarmNN`` -- watch on the wrist + phone strapped to the upper arm (reference
``estimate/watch_phone_uarm_nn.py:13-121``): 38 features -> 3x128 LSTM -> 12 targets."""
import numpy as np
import torch

from wear_mocap_ape_amd.data_deploy.nn import deploy_models
from wear_mocap_ape_amd.data_types import messaging
from wear_mocap_ape_amd.data_types.bone_map import BoneMap
from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.estimate import nn_models
from wear_mocap_ape_amd.estimate.estimator import Estimator
from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import _SW_SENSORS, _quat
from wear_mocap_ape_amd.utility import transformations as ts
from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS

_PH_SENSORS = [f"ph_{g}_{a}" for g in ("gyro", "lvel", "lacc", "grav") for a in "xyz"]
# arm orientations in the calibration pose (left arm stretched forward)
_LARM_DST_G = np.array([-0.7071068, 0, -0.7071068, 0])
_UARM_DST_G = np.array([0.7071068, 0, 0.7071068, 0])
_LEFT_HAND_CAL = np.array([0.7071068, 0, -0.7071068, 0])


def features_from_row(row, slp) -> np.ndarray:
    """55-float watch+phone message -> float64[38] (watch_phone_uarm_nn.py:43-105; the reference
    returns float64 here)."""
    r_pres = row[slp["sw_pres"]] - row[slp["sw_init_pres"]]
    # north quaternion incl. the left-hand calibration turn (transformations.py:182-197)
    north = ts.quat_mul(_LEFT_HAND_CAL, ts.north_quat_from_forward(_quat(row, slp, "sw_forward")))

    def calibrated(dev, dst_g):
        rot_g = ts.android_to_global(_quat(row, slp, f"{dev}_rotvec"), north)
        fwd_g = ts.android_to_global(_quat(row, slp, f"{dev}_forward"), north)
        return ts.quat_mul(rot_g, ts.quat_mul(ts.quat_invert(fwd_g), dst_g))

    return np.hstack([
        [row[slp[n]] for n in _SW_SENSORS],
        ts.quat_to_six_drr(calibrated("sw", _LARM_DST_G)),
        r_pres,
        [row[slp[n]] for n in _PH_SENSORS],
        ts.quat_to_six_drr(calibrated("ph", _UARM_DST_G)),
    ]).astype(np.float64)


class WatchPhoneUarmNN(Estimator):
    def __init__(self,
                 model_hash: str = deploy_models.LSTM.WATCH_PHONE_UARM.value,
                 smooth: int = 1,
                 add_mc_samples=True,
                 monte_carlo_samples=50,
                 bonemap: BoneMap = None,
                 tag: str = "NN UARM PHONE"):
        self.__tag = tag
        self._stream_mc = add_mc_samples
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

    _parse_kind = _hip.PARSE_WATCH_PHONE_UARM

    def parse_row_to_xx(self, row: np.array):
        return features_from_row(row, self.__slp)

    def make_prediction_from_row_hist(self, xx):
        xx = torch.tensor(xx[None, :, :], dtype=torch.float32)
        t_preds = self.__nn_model.monte_carlo_predictions(x=xx, n_samples=self.__mc_samples, last_step_only=True)
        return t_preds.numpy()[:, -1, :]
