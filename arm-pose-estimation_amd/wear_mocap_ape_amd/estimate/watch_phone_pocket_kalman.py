"""``WatchPhonePocketKalman`` -- the ensemble-Kalman phone-in-pocket estimator of the reference's main example script
(``estimate/watch_phone_pocket_kalman.py:12-169``, ``example_scripts/stream/watch_phone_pocket.py``): same features and
targets as ``WatchPhonePocketNN``, the regressor replaced by ``KalmanSmartwatchModel``; the corrected ensemble takes the
place of the Monte-Carlo samples.  PARITY UNPINNED (see ``estimate/kalman_models.py``)."""
from pathlib import Path

import torch

from wear_mocap_ape_amd import _hip
from wear_mocap_ape_amd.data_types import messaging
from wear_mocap_ape_amd.estimate import kalman_models
from wear_mocap_ape_amd.estimate.estimator import Estimator
from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import features_from_row
from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS


class WatchPhonePocketKalman(Estimator):
    def __init__(self,
                 model_path: Path,
                 smooth: int = 1,
                 num_ensemble: int = 32,
                 window_size: int = 10,
                 add_mc_samples: bool = True,
                 normalize: bool = True,
                 tag: str = "KALMAN POCKET PHONE"):
        super().__init__(
            x_inputs=NNS_INPUTS.WATCH_PHONE_CAL_HIP,
            y_targets=NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS,
            smooth=smooth,
            seq_len=window_size,
            add_mc_samples=add_mc_samples,
            normalize=normalize,
            tag=tag
        )
        self.__tag = tag
        self.__slp = messaging.WATCH_PHONE_IMU_LOOKUP
        self.__num_ensemble = num_ensemble
        self.__win_size = window_size
        self.__dim_x = 14
        self.__model = kalman_models.KalmanSmartwatchModel(self.__num_ensemble, self.__win_size)
        self.__model.eval()
        # the pretrained model: the reference's checkpoint file ({"model": state_dict}, :50-54) or a ready state_dict
        if isinstance(model_path, dict):
            checkpoint = model_path if "model" in model_path else {"model": model_path}
        else:
            checkpoint = torch.load(model_path, map_location=torch.device("cpu"))
        self.__model.load_state_dict(checkpoint["model"])
        self._device = self.__model.torch_device
        self.__init_state()

    def __init_state(self):
        # the filter starts from a history of zero tensors (:57-63)
        self.__init_step = 0
        self.__input_state = torch.zeros((1, self.__num_ensemble, self.__win_size, self.__dim_x), dtype=torch.float32,
                                         device=self._device)

    def reset(self):
        super().reset()
        self.__init_state()

    model = property(lambda self: self.__model)

    _parse_kind = _hip.PARSE_WATCH_PHONE_POCKET

    def parse_row_to_xx(self, row):
        return features_from_row(row, self.__slp)            # the same 22 features as the LSTM estimator (:73-131)

    def make_prediction_from_row_hist(self, xx_hist):
        # -> batch size, window_size, ensembles, raw_obs (:135)
        xx_seq = torch.tensor(xx_hist, dtype=torch.float32).to(self._device)[None, :, None, :]
        output = self.__model(xx_seq, self.__input_state)
        # not enough history yet: sensor-model predictions until a time window worth of states exists (:141-156)
        if self.__init_step <= self.__win_size:
            self.__init_step += 1
            pred = self.__model.format_state(output[3][0])[None, :, None, :]       # -> bs en k dim
            self.__input_state = torch.cat((self.__input_state[:, :, 1:, :], pred), axis=2)
            return output[3].cpu().numpy()[0][:, :14]
        # initialised: the corrected ensemble is the next input state and the output (:159-169)
        ensemble = output[0]
        self.__input_state = torch.cat((self.__input_state[:, :, 1:, :], ensemble[:, :, None, :]), axis=2)
        return ensemble.cpu().numpy()[0][:, :14]
