"""rvio_hip_frame_sharded_dev with world = 2 / 3 on ONE GPU: one handle per rank, driven by one host thread each over the same frames, the
collective supplied by the caller (the entry point's `allgather` argument: same signature as ncclAllGather) as a rendezvous of the threads
+ device-to-device copies on each handle's filter stream.  RCCL itself refuses two ranks of a communicator on one device, so this is how the library's OWN sharded frame —
propagate and the per-feature stage of the shard f % 2 == rank in one launch, the block sum in rank order, the replicated global stage, the
long-window factor on its own queue — is exercised beyond world 1 before the driver's multi-GPU run:
  * all ranks end bit-identical (the replicas of SURVEY.md 8e),
  * and within rounding of the unsharded frame path on the same frames (the shares are summed in another order)."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle as O
import scenarios as S

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu
K0 = 38


def frames(cfg_name, n):
    cfg = abi.config_named(cfg_name, enable_equalizer=1)
    seq = rv.synth.SynthSequence(cfg, duration=(K0 + n + 4) / 20.0)
    ks = list(range(K0 + 1, K0 + 1 + n))
    return cfg, seq.init_from_static(K0), np.stack([seq.render(k) for k in ks]), [seq.imu_between(k) for k in ks]


@pytest.mark.parametrize("cfg_name,n,world", [("B", 50, 2), ("B", 50, 3), ("C", 45, 2)], ids=["cfgB-world2", "cfgB-world3", "cfgC-long-window-world2"])
def test_ranks_on_one_gpu(gpu_required, cfg_name, n, world):
    import torch
    from rvio_amd import hip
    cfg, init, imgs, imus = frames(cfg_name, n)
    d_imgs = torch.from_numpy(imgs).cuda()
    d_imus = [torch.from_numpy(i.view(np.uint8)).cuda() for i in imus]
    torch.cuda.synchronize()
    rt = C.CDLL("libamdhip64.so")
    rt.hipStreamSynchronize.argtypes = [C.c_void_p]
    rt.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    bar = threading.Barrier(world, timeout=120)
    slots = [None] * world
    calls = [0] * world
    errors = []

    def make_allgather(rank):
        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)
        def ag(send, recv, count, dtype, comm, stream):
            try:
                assert dtype == 8                                   # ncclFloat64
                assert rt.hipStreamSynchronize(stream) == 0         # this rank's block is complete
                slots[rank] = send
                calls[rank] += 1
                bar.wait()
                for q in range(world):                               # on the handle's own filter stream, like the collective it stands in for (a device-to-device
                    assert rt.hipMemcpyAsync(recv + q * count * 8, slots[q], count * 8, 3, stream) == 0   # hipMemcpy on the NULL stream is not ordered with a non-blocking stream)
                assert rt.hipStreamSynchronize(stream) == 0
                bar.wait()                                           # nobody rewrites its block before the other ranks have copied it
                return 0
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))
                bar.abort()
                return 1
        return ag

    out = [None] * world

    def drive(rank):
        try:
            h = hip.RvioHip(cfg)
            h.initialize(*init)
            ag = make_allgather(rank)
            fn = C.cast(ag, C.c_void_p).value
            for i in range(n):
                h.frame_sharded_dev(d_imgs[i].data_ptr(), cfg.width, d_imus[i].data_ptr(), len(imus[i]), 0, 0, rank, world, comm=1, allgather=fn)
            h.sync()
            x, P = h.get_state()
            pts, hl = h.get_points()
            info = h.frame_info()
            h.close()
            out[rank] = dict(x=x, P=P, pts=pts, hl=hl, info=info)
        except Exception as e:   # noqa: BLE001
            errors.append("rank %d: %r" % (rank, e))
            bar.abort()

    ts = [threading.Thread(target=drive, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    assert not errors, errors
    assert all(o is not None for o in out)
    a = out[0]
    assert len(set(calls)) == 1 and calls[0] > n // 2, calls           # the collective ran in (nearly) every frame, in lockstep
    for b in out:
        assert b["info"]["device_error"] == 0
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["P"], b["P"]) and np.array_equal(a["pts"], b["pts"])   # bit-identical replicas
    # the unsharded frame path on the same frames
    h = hip.RvioHip(cfg)
    h.initialize(*init)
    for i in range(n):
        h.frame_dev(d_imgs[i].data_ptr(), cfg.width, d_imus[i].data_ptr(), len(imus[i]), 0, 0)
    h.sync()
    x, P = h.get_state()
    pts, _ = h.get_points()
    assert h.frame_info()["updated"] in (0, 1)
    h.close()
    assert np.array_equal(pts, a["pts"])
    assert S.state_delta(x, a["x"]) <= 1e-10, S.state_delta(x, a["x"])
    assert float(np.max(np.abs(P - a["P"]))) <= 1e-12 * max(1.0, float(np.max(np.abs(P))))
