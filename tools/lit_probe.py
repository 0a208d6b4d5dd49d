"""Time of the literal path (csrc/literal.h) per update: stacks of the random sweeps that take it, by rows; HIP events around rvio_hip_update_tracked."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle as O, test_truncation as TT
from rvio_amd import hip
abi = O.abi

def probe(name, sweep, want=6):
    h, seen = None, {}
    for trial, mode, cfg, n, x, P, ty, ln, me in sweep:
        if len(ln) > 24 or mode == 3:
            continue
        blk = O.update_local(cfg, x, P, ty, ln, me, 0, 1)
        if blk[-8 + 5] != 1:
            continue
        rows = int(blk[-8 + 1])
        if rows // 50 in seen:
            continue
        seen[rows // 50] = 1
        if h is None:
            h = hip.RvioHip(cfg)
        ts = []
        for rep in range(6):
            h.set_state(x, P)
            h.update(ty, ln, me)          # uploads the tracks
            h.sync()
            h.set_state(x, P)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = torch.cuda.ExternalStream(h.stream())
            e0.record(st); h.update_tracked(); e1.record(st); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ptr, nd = h.update_local(ty, ln, me, 0, 1); h.sync()
        h.set_state(x, P); h.update(ty, ln, me); h.sync()
        class DA:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
        blkd = torch.as_tensor(DA(ptr, nd), device="cuda").cpu().numpy()
        ldh = 6 * (cfg.max_track_len - 1) + 1
        st = blkd[ldh * ldh + 200: ldh * ldh + 205]
        print("   phases (us): nullspace %.0f, staging %.0f, sweep %.0f, scan + Gram %.0f" % tuple((st[1:] - st[:-1]) / 100.0))
        if len(seen) >= want:
            break

probe("cfgB", TT.sweep_wider(1500))
probe("cfgA", TT.sweep_few(1500))
