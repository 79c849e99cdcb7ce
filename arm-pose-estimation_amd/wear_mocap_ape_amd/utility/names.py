"""Column-name enums of the regressor inputs and targets.

Same member names and the same ordered column lists as the reference's
``utility/names.py:4-110`` (``NNS_TARGETS`` :4-29, ``NNS_INPUTS`` :32-110) -- the order is the
feature / target index bookkeeping the kernels rely on, so it is checked bit-for-bit against the
reference's lists in ``tests/test_host_bookkeeping.py``.  The lists are assembled from their
repeating groups instead of being spelled out, and the stdlib ``Enum`` is enough here (no two
members share a value, so ``aenum.NoAlias`` is not needed)."""
from enum import Enum


def _xyz(prefix):
    return [f"{prefix}_{a}" for a in "xyz"]


def _six(prefix):
    return [f"{prefix}_{i}" for i in range(1, 7)]


def _imu(dev):
    """gyro, integrated velocity, linear acceleration, gravity of one device"""
    return _xyz(f"{dev}_gyro") + _xyz(f"{dev}_lvel") + _xyz(f"{dev}_lacc") + _xyz(f"{dev}_grav")


_LARM_UARM = _six("gt_larm_6drr_cal") + _six("gt_uarm_6drr_cal")
_HIPS = ["gt_hips_yrot_cal_sin", "gt_hips_yrot_cal_cos"]
_PH_HIPS = ["ph_hips_yrot_cal_sin", "ph_hips_yrot_cal_cos"]
_WATCH_CAL = ["sw_dt"] + _imu("sw") + _six("sw_6drr_cal") + ["sw_pres_cal"]
_WATCH_ACC = ["sw_dt"] + _xyz("sw_lacc") + _six("sw_6drr_cal")


class NNS_TARGETS(Enum):
    ORI_CAL_LARM_UARM_HIPS = _LARM_UARM + _HIPS                                   # 14
    ORI_CAL_LARM_UARM = list(_LARM_UARM)                                          # 12
    ORI_POS_CAL_LARM_UARM_HIPS = (_xyz("gt_hand_orig_cal") + _six("gt_larm_6drr_cal") + _xyz("gt_larm_orig_cal")
                                  + _six("gt_uarm_6drr_cal") + _HIPS)             # 20


class NNS_INPUTS(Enum):
    WATCH_ONLY_CAL = list(_WATCH_CAL)                                             # 20
    WATCH_ONLY_ACC_ONLY = list(_WATCH_ACC)
    WATCH_PHONE_CAL_HIP = _WATCH_CAL + _PH_HIPS                                   # 22
    WATCH_HIP_ACC_ONLY = _WATCH_ACC + _PH_HIPS
    WATCH_HIP_ACC_AND_BAR = _WATCH_ACC + ["sw_pres_cal"] + _PH_HIPS
    WATCH_PHONE_CAL_ALL = _WATCH_CAL + _imu("ph") + _six("ph_6drr_cal")           # 38
    WATCH_ONLY_RAW = ["sw_dt"] + _imu("sw") + _six("sw_6drr_raw") + ["sw_pres_cal"]


# NN-target enum -> layout id of the C ABI (include/ape_hip.h)
TARGET_LAYOUT = {
    NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS: 0,
    NNS_TARGETS.ORI_CAL_LARM_UARM: 1,
    NNS_TARGETS.ORI_POS_CAL_LARM_UARM_HIPS: 2,
}
