"""frame rate per consecutive 20-frame chunk of one free-running stream (is the driver's --steps 20 --warmup 5 window representative?)"""
import sys, time, os, numpy as np
sys.path.insert(0, os.getcwd())
sys.argv=['x']
import bench
import torch
from rvio_amd import hip
abi=bench.abi
cfg=abi.config_named("B",enable_equalizer=1)
n_frames=246
seq,imgs,imu_arr,imu_cnt,cand_arr,cand_cnt=bench.build_inputs(cfg,n_frames)
wi,ai,ni=seq.init_from_static(bench.K0)
torch.cuda.set_device(0)
fs=bench.FrameSet(torch,cfg,imgs,imu_arr,imu_cnt,None,None)
torch.cuda.synchronize()
for rep in range(2):
    h=hip.RvioHip(cfg)
    h.initialize(wi,ai,ni)
    for i in range(6): h.frame_dev(*fs.args(i))
    h.sync()
    out=[]
    for c in range(6,n_frames,20):
        t0=time.perf_counter()
        for i in range(c,c+20): h.frame_dev(*fs.args(i))
        te=time.perf_counter()-t0
        h.sync()
        el=time.perf_counter()-t0
        info=h.frame_info()
        out.append("%d:%.0f/%.0f(%d,%d)"%(c,1e6*el/20,1e6*te/20,info["n_feat_update"],info["n_rows"]))
    h.close()
    print("us per frame total/enqueue (n_feat_update, rows of last frame):"," ".join(out))
