"""How far can ANY implementation of the filter that is not bit-identical to the reference sit from it, free-running, on a badly observable
motion?  CPU study on the oracle (test infrastructure; no GPU):

  (a) the LITERAL form (oracle/filter.cpp compress_and_apply = Updater.cc:469-619: sequential Givens QR, leading-row rank scan, S = Hn P Hn^T
      + s2 I) against ITSELF with +-1 ulp of noise on every entry of the state and of the covariance after every frame — less than what two
      correct implementations of ANY stage differ by (another summation order in propagate, Householder instead of Givens in the nullspace
      projection, Eigen's own LU);
  (b) the information-form mirror of the device's U7-U10 (orc_update_local/global: [A|b] = Hw^T [Hw | r], T = s2 I + A Pcc, structural rank
      truncation) against the literal form — every OTHER stage shared bit for bit.

    python tools/at_rest_sensitivity.py [--motion stationary|rotation|line] [--frames 100] [--seeds 5]

Round 4, direct tracks with 15 % drops (the sequences of tests/test_gpu_truncation.py), 100 frames, max per-state delta over the sequence:
    stationary   (a) 2.0e-7 .. 2.0e-6 over five noise seeds     (b) 3.3e-7      device (MI355X): 2.9e-6
    rotation     (a) 5.4e-7 .. 1.9e-6                            (b) —           device: 3.6e-7
The reference's own sensitivity to last-bit noise reaches 2e-6 on these motions (position and velocity are unobservable at rest; the sequence
amplifies a difference ~1e7-fold): a free-running 1e-6 bar is not a property of an implementation there, whatever form its update takes —
the measurement-space form included.  What an implementation owes is agreement PER UPDATE (tests hold 1e-9; measured 6.5e-14 on direct tracks)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O       # noqa: E402
import scenarios as S    # noqa: E402

abi, rv = O.abi, O.rv


def run(cfg, seq, n, ulps=0, seed=0, information_form=False):
    rng = np.random.default_rng(seed)
    w, a, ni = seq.init_from_static(38)
    s = O.System(cfg, information_form=information_form)
    s.set_state(*O.initialize(cfg, w, a, ni))
    drv = rv.synth.DirectTrackDriver(seq)
    out = []
    for k in range(39, 39 + n):
        inp = drv.inputs(k)
        if ulps:
            x, P = s.get_state()
            x = x + np.spacing(np.abs(x)) * rng.integers(-ulps, ulps + 1, x.shape)
            E = np.triu(np.spacing(np.abs(P)) * rng.integers(-ulps, ulps + 1, P.shape))
            s.set_state(x, P + E + np.triu(E, 1).T)
        s.frame(inp["imu"], inp["cand"], tracked=inp["tracked"], status=inp["status"])
        drv.after(s.tracker().get_points()[0])
        out.append(s.get_state()[0].copy())
    return out


def study(motion="stationary", frames=100, seeds=5, config="B", seq_seed=4, drop_prob=0.15):
    cfg = abi.config_named(config, enable_equalizer=0)
    seq = rv.synth.SynthSequence(cfg, duration=(38 + frames + 4) / 20.0, seed=seq_seed, drop_prob=drop_prob, motion=motion)
    lit = run(cfg, seq, frames)
    noise = [max(S.state_delta(a, b) for a, b in zip(run(cfg, seq, frames, ulps=1, seed=sd), lit)) for sd in range(seeds)]
    mirror = max(S.state_delta(a, b) for a, b in zip(run(cfg, seq, frames, information_form=True), lit))
    return noise, mirror


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--motion", default="stationary")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--seeds", type=int, default=5)
    a = ap.parse_args()
    noise, mirror = study(a.motion, a.frames, a.seeds)
    print("%s, %d frames: literal vs literal with +-1 ulp noise per frame: %s   |   information-form mirror vs literal: %.2e"
          % (a.motion, a.frames, " ".join("%.2e" % v for v in noise), mirror))
