import sys, time, os, numpy as np
sys.path.insert(0, os.getcwd())
mode=sys.argv[1] if len(sys.argv)>1 else "A"
sys.argv=['x']
import bench
import torch
from rvio_amd import hip
abi=bench.abi
cfg=abi.config_named("B",enable_equalizer=1)
K,W=20,5
n_frames=1+W+K
seq,imgs,imu_arr,imu_cnt,cand_arr,cand_cnt=bench.build_inputs(cfg,n_frames)
wi,ai,ni=seq.init_from_static(bench.K0)
torch.cuda.set_device(0)
h0=None
if mode=="A":
    h0=hip.RvioHip(cfg)
fs=bench.FrameSet(torch,cfg,imgs,imu_arr,imu_cnt,None,None)
torch.cuda.synchronize()
def run(tag, h=None):
    h=h or hip.RvioHip(cfg)
    h.initialize(wi,ai,ni)
    for i in range(1+W): h.frame_dev(*fs.args(i))
    h.sync()
    t0=time.perf_counter()
    for i in range(1+W,n_frames):
        h.frame_dev(*fs.args(i))
    tenq=time.perf_counter()-t0
    h.sync()
    el=time.perf_counter()-t0
    h.close()
    print(mode,tag,"fps %.0f ms/step %.4f enq %.4f"%(K/el,1e3*el/K,1e3*tenq/K))
run("first",h0)
run("second")
run("third")
hs=[hip.RvioHip(cfg) for _ in range(3)]
for i,hh in enumerate(hs): run("multi%d"%i,hh)
