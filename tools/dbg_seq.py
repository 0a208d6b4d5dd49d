import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, oracle as O, scenarios as S
from rvio_amd import hip
abi=O.abi
cfg=abi.config_named('B',enable_equalizer=0)
seq,recs=S.record_sequence(cfg,n_frames=14)
h=hip.RvioHip(cfg); w,a,n=seq.init_from_static(38); h.initialize(w,a,n)
for r in recs:
    inp=r['inp']
    h.track_points(inp['tracked'],inp['status'],inp['imu'],inp['cand'])
    ty,le,me=h.get_tracks()
    same_tr = (np.array_equal(ty,r['types']) and np.array_equal(le,r['lens']) and np.array_equal(me[:, :], r['meas']))
    info=h.frame_info()
    # manual frame tail through public API
    h.L.rvio_hip_sync(h.h)
    x0,P0=h.get_state()
    h.propagate(inp['imu']); x1,P1=h.get_state()
    d1=S.state_delta(x1,r['x1'])
    ncl=(len(x1)-26)//7
    if ncl>cfg.min_track_len-1:
        h.update_tracked(); x2,P2=h.get_state(); d2=S.state_delta(x2,r['x2']); dg=h.update_diag()
        accsame=np.array_equal(dg['accepted'],r['diag']['accepted'])
    else: d2=-1; accsame=None
    h.augment_compose(r['do_augment']); x3,P3=h.get_state(); d3=S.state_delta(x3,r['x3'])
    print(r['k'], 'tracks_same',same_tr, 'nf',len(ty),len(r['types']), 'd1 %.2e d2 %.2e d3 %.2e'%(d1,d2,d3), 'acc',accsame, {k:info[k] for k in ('n_klt_ok','n_ransac_inliers','ransac_winner')}, {k:r['info'][k] for k in ('n_klt_ok','n_ransac_inliers','ransac_winner')})
