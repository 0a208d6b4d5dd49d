"""Phases of feat_build_kernel<4> with the chip full (B instances per launch, natural flow of the batched-filter leg): DBG_P samples the first four feature slots of every
64th instance (mean and longest per phase, microseconds on the constant 100 MHz clock).   RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so python tools/batch_feat_phases.py [B]"""
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
names = ["loads", "barrier", "U1 (geom4)", "U2 (geom4)", "U3 Jacobians", "blocks + reflectors", "apply reflectors", "gate H Pcc", "S", "LDLt", "shares"]
for mix, nfeat in (("half", 6), ("half", None)):
    types, lens, meas = rv.synth.worst_case_tracks(cfg, x1, n_feat=nfeat, mix=mix)
    nf = len(types)
    t_nf = np.full(B, nf, np.int32)
    t_ty, t_ln, t_me = np.zeros((B, Fu), np.uint8), np.zeros((B, Fu), np.int32), np.zeros((B, Fu, ML, 2), np.float32)
    t_ty[:, :nf], t_ln[:, :nf] = types, lens
    t_me[:, :nf, : meas.shape[1]] = meas
    d = [torch.from_numpy(a_).cuda() for a_ in (t_nf, t_ty, t_ln, t_me)]
    hb = hip.RvioHip(cfg, batch=B)
    torch.cuda.synchronize()
    out, mx = (C.c_longlong * 64)(), (C.c_longlong * 64)()
    for r in range(3):
        hb.set_state(x0, P0)
        hb.L.rvio_hip_debug_phases(hb.h, out, mx)
        hb.frame_tracks_dev(d_imu.data_ptr(), 0, len(imu), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr())
        hb.sync()
    hb.L.rvio_hip_debug_phases(hb.h, out, mx)
    s, m = np.array(list(out), float), np.array(list(mx), float)
    cnt = np.maximum(s[32:43], 1)
    print("B = %d, %d features per instance (types %s): sampled workgroups %d" % (B, nf, "".join(chr(t) for t in types[:4]), int(cnt[0])))
    for k, nm in enumerate(names):
        print("  %-22s mean %7.2f us | longest %7.2f us" % (nm, s[k] / cnt[k] / 100.0, m[30 + k] / 100.0))
    print("  sum of the means %.1f us" % float(np.sum(s[0:11] / cnt / 100.0)))
    hb.close()
