"""Hashes (directory names under ``<deploy>/nn/``) of the three deployed regressors -- values of the reference's
``data_deploy/nn/deploy_models.py:4-7``, checked against it in tests/test_host_bookkeeping.py."""
from enum import Enum

LSTM = Enum("LSTM", {
    "WATCH_PHONE_POCKET": "670b66fa7664252d1cfb3b5a8a362002ffeeba5c",   # 22 features -> 2 x 256 -> 14 targets, T = 6
    "WATCH_PHONE_UARM": "7cb5cdf94ef4c66388c7f15f642005d5e008146a",     # 38 features -> 3 x 128 -> 12 targets, T = 6
    "WATCH_ONLY": "04f4ad63bfccb3668f7598c9375403e10b1fae2a",           # 20 features -> 2 x 256 -> 12 targets, T = 8
})
