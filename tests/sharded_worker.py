"""One rank of the feature-sharded updater over the REAL collective (ncclAllGather = RCCL, through rvio_hip_frame_sharded_dev): launched by
tests/test_gpu_multi_rccl.py as `python -m torch.distributed.run --nproc-per-node W tests/sharded_worker.py <cfg> <frames> <out.npz>`, one
process per GPU (one process on GPU 0 for W = 1).  Every rank renders the same synthetic frames, runs them through the sharded frame path and
sends its end state to rank 0, which also runs the plain (unsharded) frame path on the same frames and writes everything to <out.npz>."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
K0 = 38


def main():
    cfg_name, n, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    import torch
    import torch.distributed as dist
    from pkgload import load_pkg
    rv = load_pkg()
    from rvio_amd import hip, rccl
    abi = rv.abi
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    cfg = abi.config_named(cfg_name, enable_equalizer=1)
    seq = rv.synth.SynthSequence(cfg, duration=(K0 + n + 4) / 20.0)
    ks = list(range(K0 + 1, K0 + 1 + n))
    imgs = np.stack([seq.render(k) for k in ks])
    imus = [seq.imu_between(k) for k in ks]
    init = seq.init_from_static(K0)
    d_imgs = torch.from_numpy(imgs).cuda()
    d_imus = [torch.from_numpy(i.view(np.uint8)).cuda() for i in imus]
    torch.cuda.synchronize()
    comm = rccl.RcclComm(rank, world, dist, torch)
    h = hip.RvioHip(cfg, device=local)
    h.initialize(*init)
    collectives = 0
    for i in range(n):
        h.frame_sharded_dev(d_imgs[i].data_ptr(), cfg.width, d_imus[i].data_ptr(), len(imus[i]), 0, 0, rank, world, comm.comm)
        collectives += 1 if i + 1 > cfg.min_track_len else 0
    h.sync()
    x, P = h.get_state()
    pts, _ = h.get_points()
    err = h.frame_info()["device_error"]
    h.close()
    # end states of every rank to rank 0 (fixed shapes)
    xs = [torch.zeros(len(x), dtype=torch.float64, device="cuda") for _ in range(world)]
    Ps = [torch.zeros(P.size, dtype=torch.float64, device="cuda") for _ in range(world)]
    ps = [torch.zeros(2 * cfg.n_features + 1, dtype=torch.float32, device="cuda") for _ in range(world)]
    es = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(world)]
    pp = np.zeros(2 * cfg.n_features + 1, np.float32)
    pp[0] = len(pts)
    pp[1:1 + 2 * len(pts)] = np.asarray(pts, np.float32).reshape(-1)
    dist.all_gather(xs, torch.from_numpy(np.ascontiguousarray(x)).cuda())
    dist.all_gather(Ps, torch.from_numpy(np.ascontiguousarray(P).reshape(-1)).cuda())
    dist.all_gather(ps, torch.from_numpy(pp).cuda())
    dist.all_gather(es, torch.tensor([err], dtype=torch.int64, device="cuda"))
    if rank == 0:
        hp = hip.RvioHip(cfg, device=local)      # the plain frame path on the same frames
        hp.initialize(*init)
        for i in range(n):
            hp.frame_dev(d_imgs[i].data_ptr(), cfg.width, d_imus[i].data_ptr(), len(imus[i]), 0, 0)
        hp.sync()
        xp, Pp = hp.get_state()
        ptp, _ = hp.get_points()
        hp.close()
        np.savez(out_path, world=world, x=np.stack([t.cpu().numpy() for t in xs]), P=np.stack([t.cpu().numpy() for t in Ps]),
                 pts=np.stack([t.cpu().numpy() for t in ps]), err=np.array([int(t.item()) for t in es]), x_plain=xp, P_plain=np.asarray(Pp).reshape(-1),
                 pts_plain=np.asarray(ptp, np.float32).reshape(-1), n_clones=(len(x) - 26) // 7)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
