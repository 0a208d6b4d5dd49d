"""Phase stamps of feat_build_kernel<4> (workgroup 0 of a batch launch) with the chip full: RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so python tools/batch_phase_clocks.py [B]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
rv, abi = bench.rv, bench.abi
from rvio_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = abi.config_named("B", enable_equalizer=0)
Fu, ML = abi.fu(cfg), cfg.max_track_len
seq = rv.synth.SynthSequence(cfg, duration=(bench.K0 + 40) / 20.0 + 1.0)
h1 = hip.RvioHip(cfg)
h1.initialize(*seq.init_from_static(bench.K0))
drv = rv.synth.DirectTrackDriver(seq)
nfill = cfg.max_track_len + 8
for f in range(nfill):
    inp = drv.inputs(bench.K0 + 1 + f)
    h1.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
    drv.after(h1.get_points()[0])
x0, P0 = h1.get_state()
imu = seq.imu_between(bench.K0 + 1 + nfill)
h1.propagate(imu)
x1, _ = h1.get_state()
h1.close()
d_imu = torch.from_numpy(np.ascontiguousarray(imu).view(np.uint8)).cuda()
for mix, nfeat in (("half", None), ("half", 12)):
    types, lens, meas = rv.synth.worst_case_tracks(cfg, x1, n_feat=nfeat, mix=mix)
    nf = len(types)
    t_nf = np.full(B, nf, np.int32)
    t_ty, t_ln, t_me = np.zeros((B, Fu), np.uint8), np.zeros((B, Fu), np.int32), np.zeros((B, Fu, ML, 2), np.float32)
    t_ty[:, :nf], t_ln[:, :nf] = types, lens
    t_me[:, :nf, : meas.shape[1]] = meas
    d = [torch.from_numpy(a_).cuda() for a_ in (t_nf, t_ty, t_ln, t_me)]
    hb = hip.RvioHip(cfg, batch=B)
    torch.cuda.synchronize()
    for r in range(3):
        hb.set_state(x0, P0)
        hb.frame_tracks_dev(d_imu.data_ptr(), 0, len(imu), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr())
        hb.sync()
    out = (C.c_longlong * 64)()
    hb.L.rvio_hip_debug_clocks(hb.h, out)
    t = np.array(list(out))
    idx = [i for i in range(30, 41) if t[i] != 0]
    names = {31: "loads", 32: "U1(skip)", 33: "LM(skip)", 34: "jacobians", 35: "reflectors", 36: "apply", 37: "gate H Pcc", 38: "S", 39: "LDLt", 40: "shares"}
    print("B=%d, %d features per instance (type/len of feature 0: %s/%d):" % (B, nf, chr(types[0]), lens[0]), " ".join("%s:%d" % (names.get(b, b), t[b] - t[a]) for a, b in zip(idx[:-1], idx[1:])), "total", t[idx[-1]] - t[idx[0]])
    hb.close()
