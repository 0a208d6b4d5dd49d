"""In-situ timeline of the filter chain in the pipelined single-stream run (instrumented build):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRVIO_DBG_CLOCKS r-vio_amd/csrc/rvio_hip.hip -o r-vio_amd/librvio_dbg.so
    RVIO_HIP_LIB=r-vio_amd/librvio_dbg.so python tools/chain_clocks.py [frames]
Every stage's first workgroup stamps the constant 100 MHz clock at its start (DBG_R, rvio_dev.h); printed: the mean start-to-start
interval of consecutive stages over the last 48 frames (microseconds) = what each stage costs the serial chain, launch gap included,
and the frame period (feat_prop to feat_prop)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

rv, abi = bench.rv, bench.abi
from rvio_amd import hip  # noqa: E402

cfg = abi.config_named("B", enable_equalizer=1)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_frames = 1 + K
seq, imgs, imu_arr, imu_cnt, cand_arr, cand_cnt = bench.build_inputs(cfg, n_frames)
h = hip.RvioHip(cfg)
d_imgs = torch.from_numpy(imgs).cuda()
d_imu = torch.from_numpy(imu_arr.view(np.uint8).reshape(n_frames, -1)).cuda()
torch.cuda.synchronize()
h.initialize(*seq.init_from_static(bench.K0))
for i in range(n_frames):
    h.frame_dev(d_imgs.data_ptr() + i * cfg.width * cfg.height, cfg.width, d_imu.data_ptr() + i * d_imu.shape[1], int(imu_cnt[i]), 0, 0)
h.sync()
out = (C.c_longlong * 512)()
fr = C.c_int(0)
h.L.rvio_hip_debug_ring(h.h, out, C.byref(fr))
t = np.array(list(out), dtype=np.int64).reshape(64, 8)
last = fr.value
rows = [t[(last - j) & 63] for j in range(48, -1, -1)]    # oldest .. newest
names = ["feat_prop", "gram", "solve", "ug", "final", "augcomp", "augcomp end", "bookkeep"]
r = np.array(rows, dtype=np.float64) / 100.0               # microseconds
print("frames stamped:", last)
print("period (feat_prop -> feat_prop): %.1f us" % np.mean(np.diff(r[:, 0])))
fused = bool(np.all(r[1:, 4] < r[1:, 3]))     # round 4: U / G / P1 and the Joseph form are ONE launch at 6n <= 60 (joseph_lds_kernel stamps slot 3 only; slot 4 keeps its old value)
stages = [0, 1, 2, 3, 5, 6] if fused else [0, 1, 2, 3, 4, 5, 6]
if fused:
    names[3] = "joseph_lds"
for a, b in zip(stages[:-1], stages[1:]):
    print("%-12s -> %-12s %.1f us" % (names[a], names[b], np.mean(r[1:, b] - r[1:, a])))
print("solve start -> solve end (thread 0): %.1f us; solve end -> ug start: %.1f us" % (np.mean(r[1:, 7] - r[1:, 2]), np.mean(r[1:, 3] - r[1:, 7])))
print("augcomp end -> next feat_prop: %.1f us" % np.mean(r[1:, 0] - r[:-1, 6]))
out2 = (C.c_longlong * 512)()
fr2 = C.c_int(0)
h.L.rvio_hip_debug_ring2(h.h, out2, C.byref(fr2))
t2 = np.array(list(out2), dtype=np.int64).reshape(64, 8)
rows2 = np.array([t2[(fr2.value - j) & 63] for j in range(48, -1, -1)], dtype=np.float64) / 100.0
n2 = {0: "klt", 2: "ransac", 3: "bookkeep (launch)", 4: "bookkeep (after the wait)", 5: "bookkeep end"}
print("side stream: period (klt -> klt): %.1f us" % np.mean(np.diff(rows2[:, 0])))
ids = [0, 2, 3, 4, 5]
for a, b in zip(ids[:-1], ids[1:]):
    print("%-26s -> %-26s %.1f us" % (n2[a], n2[b], np.mean(rows2[1:, b] - rows2[1:, a])))
print("bookkeep end -> next klt: %.1f us" % np.mean(rows2[1:, 0] - rows2[:-1, 5]))
# how far the side chain runs ahead of the filter: book-keeping(k) end -> feat_prop(k) start (frame numbers: side = fr2, filter = last)
off = fr2.value - last
print("side chain frames ahead of the filter at the end:", off)
# ---- absolute timeline of the three chains by FRAME (image chain tag = frame number; the side ring advances once per frame at KLT, the
# filter ring once per update at feat_prop): who waits for whom
out3 = (C.c_longlong * 512)()
h.L.rvio_hip_debug_ring3(h.h, out3)
t3 = np.array(list(out3), dtype=np.int64).reshape(64, 8) / 100.0
side_of = lambda f: t2[(fr2.value - (n_frames - 1 - f)) & 63] / 100.0     # KLT of frame f was the (f+1)-th KLT launch
filt_of = lambda f: t[(last - (n_frames - 1 - f)) & 63] / 100.0
rows = []
for f in range(n_frames - 40, n_frames - 2):
    im, sd, fl = t3[f & 63], side_of(f), filt_of(f)
    rows.append([im[3] - im[0],            # image chain: CLAHE start -> last cornerSubPix workgroup done
                 sd[6] - im[3],            # corners ready -> refill half of book-keeping starts (negative: book-keeping waited for the detector)
                 sd[6] - sd[4],            # hand-over half (after its wait) -> refill half starts
                 sd[0] - side_of(f - 1)[5],  # book-keeping(f-1) end -> KLT(f) start
                 fl[0] - sd[4],            # hand-over half of book-keeping(f) after its wait -> feat_prop(f) start
                 sd[4] - sd[3],            # the hand-over half's wait for filter(f-2)
                 im[0] - side_of(f - 3)[5]])   # book-keeping(f-3) end -> CLAHE(f) start
rows = np.array(rows)
for name, col in zip(["image chain CLAHE start -> cornerSubPix end", "cornerSubPix(f) end -> bookkeep_b(f) start (<0: waited for the detector)",
                      "bookkeep_a(f) after wait -> bookkeep_b(f) start", "bookkeep_b(f-1) end -> KLT(f) start", "bookkeep_a(f) after wait -> feat_prop(f) start",
                      "bookkeep_a(f): wait for filter(f-2)", "bookkeep_b(f-3) end -> CLAHE(f) start"], rows.T):
    print("%-78s mean %7.1f  min %7.1f  max %7.1f us" % (name, col.mean(), col.min(), col.max()))
g = []
for f in range(n_frames - 40, n_frames - 2):
    im, fl, flp = t3[f & 63], filt_of(f), filt_of(f - 1)
    if im[4] > 0:
        g.append([im[4] - flp[6], im[5] - im[4], fl[0] - im[5]])
if g:
    g = np.array(g)
    print("filter stream: augcomp(f-1) block 0 end -> gate(f) start %.1f us, gate start -> end %.1f us, gate end -> feat_prop(f) start %.1f us (means)" % tuple(g.mean(0)))
    print("gate duration per frame (us):", " ".join("%.0f" % v for v in g[:, 1]))
ck = (C.c_longlong * 64)()
h.L.rvio_hip_debug_clocks(h.h, ck)
ck = np.array(list(ck), dtype=np.float64) / 100.0
if ck[20] > 0 and ck[25] > ck[20]:
    print("joseph_lds_kernel workgroup 0, last frame (clock64 ticks / 100, ~2.35 GHz): loads %.2f | U = Pc W %.2f | G = U A %.2f | P1c %.2f | closing products %.2f" % tuple(ck[21:26] - ck[20:25]))
if ck[38] > ck[30] > 0:
    print("solve9_small_kernel, last frame (clock64 ticks / 100): " + " | ".join("%d:+%.1f" % (i, ck[i] - ck[i - 1]) for i in range(31, 39)))
if ck[27] > ck[26] > 0:
    print("feat_prop_kernel, last frame: the propagate workgroup starts %.2f us after feature workgroup 0 and runs %.2f us" % (ck[26] - ck[28], ck[27] - ck[26]))
h.close()
