"""Paths and ports (mirror of the reference's config.py:5-19).

``PATHS["deploy"]`` may be redirected with the environment variable ``WEAR_MOCAP_APE_DEPLOY``
(e.g. to the reference's own ``data_deploy`` directory holding trained ``checkpoint.pt`` files)."""
import os
from pathlib import Path

proj_path = Path(__file__).parent.absolute()

_deploy = Path(os.environ.get("WEAR_MOCAP_APE_DEPLOY", proj_path / "data_deploy"))
PATHS = {"deploy": _deploy, "skeleton": _deploy}

PORT_PUB_LEFT_ARM = 50003
PORT_LISTEN_WATCH_PHONE_IMU = 65000
PORT_LISTEN_WATCH_IMU = 46000
PORT_LISTEN_AUDIO = 65001
PORT_PUB_TRANSCRIBED_KEYS = 50006
