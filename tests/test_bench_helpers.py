"""bench.py's algorithmic-work model (W_filter) — the figure `batched_filter.achieved_tflops_fp64` is priced with."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing bench does not touch the GPU or the oracle)


def test_filter_flops_matches_the_survey_figures():
    abi = bench.abi
    cfg = abi.config_named("B")
    Fu = abi.fu(cfg)
    # worst case of SURVEY.md 8d: every one of the ceil(F/2) features at maximum length -> "cfg B 51 MFLOP/frame"
    # (gate 18.9 + compression 21.3 incl. the 2.3 of the nullspace step, which the W_filter formula itself does not carry, + EKF 7.4 + propagate 0.9)
    w = bench.filter_flops(cfg, 10, [11] * Fu, [ord("1")] * Fu, 10) / 1e6
    assert 46.0 < w < 51.5, w
    gate = sum(2 * 19 * 60 ** 2 + 2 * 19 ** 2 * 60 + (4 / 3) * 19 ** 3 for _ in range(Fu)) / 1e6
    assert abs(gate - 18.9) < 0.1
    # no update while the window is too short: propagation only (m * 6 * 24^3)
    assert bench.filter_flops(cfg, 1, [3] * 5, [ord("1")] * 5, 10) == 10 * 6.0 * 24 ** 3
    # a type-'2' feature contributes its first ceil(L/2) observations only
    a = bench.filter_flops(cfg, 10, [11], [ord("2")], 0)
    b = bench.filter_flops(cfg, 10, [6], [ord("1")], 0)
    assert a == b
