import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import oracle as O, scenarios as S
from rvio_amd import hip
abi,rv=O.abi,O.rv
cfg = abi.config_named("B", enable_equalizer=1)
n = 80
seq = rv.synth.SynthSequence(cfg, duration=(38 + n + 4) / 20.0, seed=2, motion="stationary")
w, a, ni = seq.init_from_static(38)
x0, P0 = O.initialize(cfg, w, a, ni)
h = hip.RvioHip(cfg); h.initialize(w, a, ni)
h1 = hip.RvioHip(cfg); h1.initialize(w, a, ni)
lit = O.System(cfg); lit.set_state(x0, P0)
for k in range(39, 39 + n):
    img, imu = seq.render(k), seq.imu_between(k)
    xs,Ps = lit.get_state()
    if k > 39: h1.set_state(xs,Ps)
    oi = lit.frame(imu, None, img=img)[0]
    h.frame(img, imu, None); h.sync()
    h1.frame(img, imu, None); h1.sync()
    gi = h.frame_info()
    xl,Pl = lit.get_state()
    d = np.abs(S.qfix(h.get_state()[0])-S.qfix(xl)); d1=np.abs(S.qfix(h1.get_state()[0])-S.qfix(xl))
    print(k, "free %.2e@%d  reseeded %.2e@%d"%(d.max(), d.argmax(), d1.max(), d1.argmax()), "upd", gi["updated"], "feat", gi["n_feat_update"], gi["n_feat_accepted"], "rows", gi["n_rows"], "trunc dev", gi["rank_truncated_at"], "lit rank", lit.last_rank(), "Pmax %.1e"%np.abs(Pl).max())
