"""Positions of the measurements inside the smartwatch (+phone) UDP message -- same tables as
the reference's ``data_types/messaging.py:20-68`` (watch only, 28 floats) and ``:96-187``
(watch + phone, 55 floats), generated from their field groups."""


def _device_block(dev, with_calibration):
    names = [f"{dev}_dt", f"{dev}_h", f"{dev}_m", f"{dev}_s", f"{dev}_ns"]
    names += [f"{dev}_rotvec_{c}" for c in ("w", "x", "y", "z", "conf")]
    for grp in ("gyro", "lvel", "lacc"):
        names += [f"{dev}_{grp}_{a}" for a in "xyz"]
    names += [f"{dev}_pres"]
    names += [f"{dev}_grav_{a}" for a in "xyz"]
    if with_calibration:
        names += [f"{dev}_forward_{c}" for c in "wxyz"] + [f"{dev}_init_pres"]
    return names


def _lookup(names):
    return {n: i for i, n in enumerate(names)}


WATCH_ONLY_IMU_LOOKUP = _lookup(_device_block("sw", True))
watch_only_imu_msg_len = len(WATCH_ONLY_IMU_LOOKUP) * 4

WATCH_PHONE_IMU_LOOKUP = _lookup(
    _device_block("sw", False) + _device_block("ph", False)
    + [f"sw_forward_{c}" for c in "wxyz"] + [f"ph_forward_{c}" for c in "wxyz"] + ["sw_init_pres"])
watch_phone_imu_msg_len = len(WATCH_PHONE_IMU_LOOKUP) * 4
