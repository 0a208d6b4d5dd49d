"""The feature-sharded updater over the REAL collective: rvio_hip_frame_sharded_dev with ncclAllGather (RCCL) on the handle's filter stream, one
process per GPU (tests/sharded_worker.py under torch.distributed.run).  World 1 runs on any box (a communicator of one rank: the collective,
the block sum and the replicated global stage are all on the path); world 2 and 8 run wherever that many GPUs are visible — the first
multi-GPU box executes code that has otherwise only seen more than one rank through the caller-supplied rendezvous of
tests/test_gpu_sharded_ranks.py (threads on one GPU) and the gloo tests.  Asserted: no device error, every rank's end state BIT-identical
(the replicas of SURVEY.md 8e), tracker tables equal to the plain frame path's, state within 1e-9 of the plain frame path on the same frames."""
import os
import subprocess
import sys

import numpy as np
import pytest

import scenarios as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg_name,n", [("B", 40), ("E", 40)], ids=["cfgB", "cfgE"])
@pytest.mark.parametrize("world", [1, 2, 8])
def test_sharded_frame_over_rccl(gpu_required, tmp_path, world, cfg_name, n):
    import torch
    have = torch.cuda.device_count()
    if have < world:
        pytest.skip("%d GPU(s) visible, world %d needs %d" % (have, world, world))
    out = str(tmp_path / "sharded.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    port = 29600 + (os.getpid() % 200) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "sharded_worker.py"), cfg_name, str(n), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    d = np.load(out)
    assert int(d["world"]) == world and int(d["n_clones"]) > 0
    assert not d["err"].any(), d["err"]
    for q in range(1, world):      # replicas: the same bits on every rank
        assert np.array_equal(d["x"][0], d["x"][q]) and np.array_equal(d["P"][0], d["P"][q]) and np.array_equal(d["pts"][0], d["pts"][q]), q
    npts = int(d["pts"][0][0])
    assert np.array_equal(d["pts"][0][1:1 + 2 * npts], d["pts_plain"])          # the tracker never reads the filter: identical tables
    assert S.state_delta(d["x"][0], d["x_plain"]) <= 1e-9, S.state_delta(d["x"][0], d["x_plain"])
    assert float(np.max(np.abs(d["P"][0] - d["P_plain"]))) <= 1e-9 * max(1.0, float(np.max(np.abs(d["P_plain"]))))
