import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import oracle as O, scenarios as S
from rvio_amd import hip
abi,rv=O.abi,O.rv
cfg = abi.config_named("E", enable_equalizer=1)
seq = rv.synth.SynthSequence(cfg, duration=3.0, n_landmarks=12000)
w, a, n = seq.init_from_static(38)
h = hip.RvioHip(cfg); h.initialize(w, a, n)
x, P = O.initialize(cfg, w, a, n)
trk = O.Tracker(cfg)
img_count=0
for k in range(39, 39 + 45):
    img, imu = seq.render(k), seq.imu_between(k)
    trk.track(img, imu, None)
    img_count+=1
    ncl=(len(x)-26)//7
    x1,P1=O.propagate(cfg,x,P,imu)
    types,lens,meas=trk.get_tracks()
    d=None
    if ncl>cfg.min_track_len-1:
        x2,P2,d=O.update(cfg,x1,P1,types,lens,meas)
    else: x2,P2=x1,P1
    x,P,_,_=O.augment_compose(cfg,x2,P2,img_count>1)
    h.frame(img, imu, None); h.sync()
    gi=h.frame_info()
    if d is not None:
        dg=h.update_diag()
        same=np.array_equal(dg["accepted"],d["accepted"])
        if not same or k>=74:
            print(k,"nfeat",len(types),"types",bytes(types[:12]),"lens",lens[:12])
            print("  dev acc",dg["accepted"][:12],"gamma",np.array2string(dg["gamma"][:12],precision=6),"ndof",dg["ndof"][:12])
            print("  orc acc",d["accepted"][:12],"gamma",np.array2string(d["gamma"][:12],precision=6),"ndof",d["ndof"][:12])
            print("  pfinv dev",dg["pfinv"][:3],"\n  pfinv orc",d["pfinv"][:3])
            print("  state delta",S.state_delta(h.get_state()[0],x))
        if not same: break
