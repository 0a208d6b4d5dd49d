import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, oracle as O, scenarios as S
from rvio_amd import hip
abi=O.abi
cfg=abi.config_named('B',enable_equalizer=0)
seq,recs=S.record_sequence(cfg,n_frames=20)
r=recs[-1]
h=hip.RvioHip(cfg)
h.set_state(r['x1'],r['P1'])
ty,le,me=S.worst_case_tracks(cfg,r,seq)
for rep in range(3):
    for k in range(4): h.propagate(r['inp']['imu'])
    for k in range(4): h.augment_compose(False)
    for k in range(4): h.update(r['types'],r['lens'],r['meas'])
h.sync()
