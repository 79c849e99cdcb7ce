"""Hashes of the deployed regressors (reference data_deploy/nn/deploy_models.py:4-7)."""
from enum import Enum


class LSTM(Enum):
    WATCH_PHONE_POCKET = "670b66fa7664252d1cfb3b5a8a362002ffeeba5c"   # 22 -> 2x256 -> 14, T=6
    WATCH_PHONE_UARM = "7cb5cdf94ef4c66388c7f15f642005d5e008146a"     # 38 -> 3x128 -> 12, T=6
    WATCH_ONLY = "04f4ad63bfccb3668f7598c9375403e10b1fae2a"           # 20 -> 2x256 -> 12, T=8
