"""Reading the full-load fixtures of tests/golden/ (full_load_inputs.npz, ref_full_load_outputs.npz): shared by make_golden_ref.py, the CPU
test and the GPU test.  No oracle, no reference: numpy and the configuration table only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from pkgload import load_pkg  # noqa: E402

abi = load_pkg().abi

FULL_LOAD = [("A", "half", None), ("B", "all2", None), ("C", "half", None), ("E", "half", 160)]   # (configuration, length mix, features; E: 160 keeps mini-Eigen to seconds)
N_PROBE = 4


def probes(d):
    """fixed probe vectors: the covariance is stored as its diagonal and P V (a discrepancy anywhere in P shows in P V)"""
    return np.random.default_rng(123).standard_normal((d, N_PROBE))


def load_full_load_case(g, name):
    """-> cfg, x1, P1 (symmetric, from its stored upper triangle), types, lens, meas"""
    cfg = abi.config_named(name)
    x1 = g[name + "_x1"]
    d = 24 + 6 * ((len(x1) - 26) // 7)
    P1 = np.zeros((d, d))
    P1[np.triu_indices(d)] = g[name + "_P1u"]
    P1 = P1 + np.triu(P1, 1).T
    lens = g[name + "_lens"]
    meas = np.zeros((len(lens), cfg.max_track_len, 2), np.float32)
    meas[:, : g[name + "_meas"].shape[1]] = g[name + "_meas"]
    return cfg, x1, P1, g[name + "_types"], lens, meas


