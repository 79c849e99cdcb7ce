"""Normalisation statistics of a (inputs, targets) pair -- load branch of the reference's
``utility/data_stats.py:11-40``.  Fitting new statistics from recordings (:42-92) is
training-time code and out of scope.

File name: ``"{x_inputs.name}_{y_targets.name}"`` (:30) under ``<deploy>/data_stats/``; this
package ships the reference's three stat sets re-exported as plain ``.json``; a user-provided
``.pkl`` of the reference's format is read as well."""
import json
import logging
import pickle
from pathlib import Path

import numpy as np

from wear_mocap_ape_amd import config
from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS


def get_norm_stats(x_inputs: NNS_INPUTS, y_targets: NNS_TARGETS, data_list: list = None) -> dict:
    """-> dict with float64 arrays ``xx_m, xx_s`` [I] and ``yy_m, yy_s`` [O]."""
    if data_list is not None:
        raise UserWarning("fitting normalisation stats from recordings is out of scope of this package")
    stem = "{}_{}".format(x_inputs.name, y_targets.name)
    f_dir = Path(config.PATHS["deploy"]) / "data_stats"
    if (f_dir / (stem + ".json")).exists():
        raw = json.loads((f_dir / (stem + ".json")).read_text())
        logging.info("loaded data stats from {}".format(f_dir / (stem + ".json")))
        stats = {k: np.asarray(raw[k], dtype=np.float64) for k in ("xx_m", "xx_s", "yy_m", "yy_s")}
        stats["x_inputs"], stats["y_targets"] = raw.get("x_inputs"), raw.get("y_targets")
        return stats
    if (f_dir / (stem + ".pkl")).exists():
        with open(f_dir / (stem + ".pkl"), "rb") as handle:
            return pickle.load(handle)
    raise UserWarning("attempting to create stats file without data list")   # data_stats.py:50-51
