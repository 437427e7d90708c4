"""GPU parity tests added in round 3 (run with -m gpu): BASELINE config 4's shape (512^2 render = IS 1024, 5120 faces)
through the train_s2 render-and-compare module and the loss kernels at H = 512, regulariser / eval kernels / rotate_cam
goldens, capped super-block bins.

Every comparison is HIP (through the C ABI) against golden vectors written by the reference itself, the CPU oracle on the same
seeded inputs, tolerances are
written at each check and the measured figures are appended to gpurun_out/parity_measured.jsonl (helpers.assert_close_frac)."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import scene, assert_close_frac, t2n

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RARGS = ([0.1, 0.2, 0.3], 1, 100, True, 1e-3, 1e-5, 'euclidean', 1e-10, 1e-4)


