"""N>1 path on CPU: world_size-2 (and 4) gloo groups run the feature-sharded updater protocol
 — rank r builds the information block [A|b] of features f % world == r, ONE all-gather
exchanges the blocks, every rank runs the identical global stage.  The compute legs here are the CPU
oracle's mirrors of rvio_hip_update_local/_global (tests only); what is under test is the protocol:
partition, payload layout, gather order, replica determinism."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle as O
    import scenarios as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=30)
    r = recs[-1]
    types, lens, meas = S.worst_case_tracks(cfg, r, seq)
    own = O.update_local(cfg, r["x1"], r["P1"], types, lens, meas, rank, world)
    # what travels is the device's WIRE FORMAT (csrc/rvio_dev.h shard_layout, NumPy mirror abi.shard_pack): 8 counters + the upper-triangle
    # 16 x 16 tiles of both parts that can be non-zero — 31 KB instead of 60 KB at this window, 215 instead of 524 KB at cfg E
    abi = O.abi
    ldh = 6 * (cfg.max_track_len - 1) + 1
    part = ldh * (ldh - 1)
    c6 = 6 * ((len(r["x1"]) - 26) // 7)
    parts = np.stack([own[pt * part: (pt + 1) * part].reshape(ldh - 1, ldh) for pt in range(2)])
    pay, _ = abi.shard_pack(parts, own[2 * part: 2 * part + 8], c6, cfg.max_track_len)
    assert len(pay) == abi.shard_payload_doubles(c6, cfg.max_track_len) and len(pay) * 8 <= 0.55 * 2 * ldh * ldh * 8
    blk = torch.from_numpy(pay)
    gathered = [torch.zeros_like(blk) for _ in range(world)]
    dist.all_gather(gathered, blk)                       # the ONE collective of the frame
    blocks = []
    for g in gathered:                                   # back to the oracle's own layout for its global stage (lower triangle mirrored from the upper tiles)
        pp, cnt = abi.shard_unpack(g.numpy(), c6, cfg.max_track_len, ldh)
        blocks.append(np.concatenate([pp[0][: ldh - 1].reshape(-1), pp[1][: ldh - 1].reshape(-1), cnt]))
    blocks = np.stack(blocks)
    x, P, info = O.update_global(cfg, r["x1"], r["P1"], blocks)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, P=P, good=info["n_good"], rows=info["n_rows"], own=own, wire_doubles=len(pay))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_update_gloo(world, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    import scenarios as S
    cfg = O.abi.config_named("B", enable_equalizer=0)
    seq, recs = S.record_sequence(cfg, n_frames=30)
    r = recs[-1]
    types, lens, meas = S.worst_case_tracks(cfg, r, seq)
    xo, Po, d = O.update(cfg, r["x1"], r["P1"], types, lens, meas)        # unsharded reference-form update
    res = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(world)]
    for k in range(1, world):                                                # replicas are bit-identical
        assert np.array_equal(res[0]["x"], res[k]["x"]) and np.array_equal(res[0]["P"], res[k]["P"])
    assert int(res[0]["good"]) == d["n_good"] and int(res[0]["rows"]) == d["n_rows"]
    assert S.state_delta(res[0]["x"], xo) < 1e-9
    assert np.max(np.abs(res[0]["P"] - Po)) < 1e-9 * np.max(np.abs(Po))
    # every feature is owned by exactly one rank: per-rank accepted counts add up
    # (payload: type-'2' part, type-'1' part, then 8 counters starting with n_good, n_rows — oracle/filter.cpp:orc_update_local)
    assert sum(int(res[k]["own"][-8]) for k in range(world)) == d["n_good"]
    assert sum(int(res[k]["own"][-7]) for k in range(world)) == d["n_rows"]
