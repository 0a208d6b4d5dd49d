"""Batched filter (rvio_hip_create_batch / rvio_hip_frame_tracks_dev): B instances advanced by one launch
per stage against B plain handles fed the same inputs through the per-stage calls.  The front end (integer / float32 with order-free
sums) is bit for bit the same; the filter agrees to rounding — a batch handle runs the throughput forms of three filter kernels
(solve6 behind gemm_T instead of solve7, the compact share reduction, fewer loads in flight in the gate), the same algorithms with
a different summation grouping — and therefore both agree with the oracle within the filter tolerance."""

import numpy as np
import pytest

import oracle as O
import scenarios as S

abi, rv = O.abi, O.rv
pytestmark = pytest.mark.gpu


def same_filter_state(xa, Pa, xb, Pb, tol=1e-11):
    """rounding-level agreement of two filter states.  1e-11 is the bound for FREE-RUNNING sequences (the last bits of one update are
    amplified by the next ones); one update from identical inputs is held to 1e-13 (test_one_update_batch_vs_plain_is_rounding_only)."""
    return S.state_delta(xa, xb) <= tol and float(np.max(np.abs(Pa - Pb))) <= tol * max(1e-30, float(np.max(np.abs(Pb))))


def pack_inputs(cfg, recs_f):
    """one frame of B instances -> the device layout of rvio_hip_frame_tracks_dev"""
    B, Fu, ML = len(recs_f), abi.fu(cfg), cfg.max_track_len
    n_feat = np.zeros(B, np.int32)
    types = np.zeros((B, Fu), np.uint8)
    lens = np.zeros((B, Fu), np.int32)
    meas = np.zeros((B, Fu, ML, 2), np.float32)
    m = len(recs_f[0]["inp"]["imu"])
    imu = np.zeros((B, m), dtype=recs_f[0]["inp"]["imu"].dtype)
    for i, r in enumerate(recs_f):
        n = len(r["lens"])
        n_feat[i] = n
        types[i, :n], lens[i, :n], meas[i, :n] = r["types"], r["lens"], r["meas"]
        assert len(r["inp"]["imu"]) == m
        imu[i] = r["inp"]["imu"]
    return n_feat, types, lens, meas, imu, m


@pytest.fixture(scope="module")
def recs3():
    cfg = abi.config_named("B", enable_equalizer=0)
    return cfg, [S.record_sequence(cfg, n_frames=26, seed=s)[1] for s in (0, 1, 2)]


def spread(recs, B):
    """B instances over the recorded sequences, round-robin (instance i replays sequence i % 3)"""
    return [recs[i % len(recs)] for i in range(B)]


@pytest.mark.parametrize("shared_imu,B", [(False, 3), (True, 3), (False, 128)], ids=["own-imu", "shared-imu", "128-instances"])
def test_batch_equals_plain_handles_bit_for_bit(gpu_required, recs3, shared_imu, B):
    """(B = 128 selects the forms of large batches — gram_reduce_batch_kernel, the fat augmentation workgroups, joseph_batch_kernel — which
    the small batches of the other tests never launch.)"""
    from rvio_amd import hip
    import torch
    cfg, recs = recs3
    n_seq = len(recs)
    recs = spread(recs, B)
    hb = hip.RvioHip(cfg, batch=B)
    hs = [hip.RvioHip(cfg) for _ in range(n_seq)]     # one plain handle per distinct sequence
    hb.set_state(recs[0][0]["x0"], recs[0][0]["P0"])
    for i in range(B):
        hb.set_state_at(i, recs[i][0]["x0"], recs[i][0]["P0"])
    for i in range(n_seq):
        hs[i].set_state(recs[i][0]["x0"], recs[i][0]["P0"])
    n_upd = 0
    for f in range(len(recs[0])):
        rf = [recs[i][f] for i in range(B)]
        n_feat, types, lens, meas, imu, m = pack_inputs(cfg, rf)
        if shared_imu:
            imu[:] = imu[0]
        d = [torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).cuda() for a in (imu, n_feat, types, lens, meas)]
        torch.cuda.synchronize()
        hb.frame_tracks_dev(d[0].data_ptr(), 0 if shared_imu else m, m, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr())
        for i in range(n_seq):
            do_update, do_augment = hs[i].frame_plan()
            hs[i].propagate(imu[i])
            if do_update:
                hs[i].update(rf[i]["types"], rf[i]["lens"], rf[i]["meas"])
                n_upd += 1
            hs[i].augment_compose(do_augment)
        hb.sync()
        plain = [hs[i].get_state() for i in range(n_seq)]
        got = [hb.get_state_at(i) for i in range(B)]
        for i in range(B):
            xa, Pa = got[i]
            xb, Pb = plain[i % n_seq]
            assert same_filter_state(xa, Pa, xb, Pb), (f, i)
            if not shared_imu:      # the recorded oracle states belong to the recorded IMU
                assert S.state_delta(xa, rf[i]["x3"]) <= 1e-9, (f, i)
            if i >= n_seq and not shared_imu:   # instances replaying the same sequence: the same bits, wherever they sit in the launch
                assert np.array_equal(xa, got[i - n_seq][0]) and np.array_equal(Pa, got[i - n_seq][1]), (f, i)
    assert n_upd > 20
    xa, _ = hb.get_state()
    assert np.array_equal(xa, hb.get_state_at(0)[0])
    hb.close()
    for h in hs:
        h.close()


def test_frame_tracks_dev_on_a_plain_handle_and_front_end_refused_on_a_batch(gpu_required, recs3):
    from rvio_amd import hip
    import torch
    cfg, recs = recs3
    h1, h2 = hip.RvioHip(cfg), hip.RvioHip(cfg, batch=2)
    h1.set_state(recs[1][0]["x0"], recs[1][0]["P0"])
    for f in range(12):
        r = recs[1][f]
        n_feat, types, lens, meas, imu, m = pack_inputs(cfg, [r])
        d = [torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).cuda() for a in (imu, n_feat, types, lens, meas)]
        torch.cuda.synchronize()
        h1.frame_tracks_dev(d[0].data_ptr(), 0, m, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr())
        h1.sync()
        assert S.state_delta(h1.get_state()[0], r["x3"]) <= 1e-9, f
    with pytest.raises(hip.RvioHipError):
        h2.track(np.zeros((cfg.height, cfg.width), np.uint8), recs[0][0]["inp"]["imu"])
    with pytest.raises(hip.RvioHipError):
        h2.update(recs[0][5]["types"], recs[0][5]["lens"], recs[0][5]["meas"])
    assert h2.L.rvio_hip_batch_size(h2.h) == 2 and h1.L.rvio_hip_batch_size(h1.h) == 1
    h1.close()
    h2.close()


@pytest.fixture(scope="module")
def scenes3():
    cfg = abi.config_named("B")                     # stock: CLAHE on, corners from the device detector
    n_frames, k0 = 12, 60
    seqs = [rv.synth.SynthSequence(cfg, duration=8.0, seed=s) for s in range(3)]
    imgs = [[q.render(k0 + f) for q in seqs] for f in range(n_frames)]                     # [frame][scene][H][W]
    imus = [[q.imu_between(k0 + f) for q in seqs] for f in range(n_frames)]
    return cfg, seqs, imgs, imus


@pytest.mark.parametrize("B", [3, 8])
def test_batch_with_front_end_equals_plain_handles_bit_for_bit(gpu_required, scenes3, B):
    """rvio_hip_frame_batch_dev: B camera streams (different scenes) through CLAHE, detector, KLT, RANSAC, book-keeping and the filter
    in one launch per stage — every instance must end in exactly the state of a plain handle fed its own stream.  B = 8 selects the
    throughput forms of the image kernels (4 pixels per thread, one wave per corner): they must give the same bits as the 1-pixel forms
    the plain handles run."""
    from rvio_amd import hip
    import torch
    cfg, seqs3, imgs3, imus3 = scenes3
    n_frames = len(imgs3)
    pick = [b % 3 for b in range(B)]
    seqs = [seqs3[k] for k in pick]
    imgs = np.stack([[imgs3[f][k] for k in pick] for f in range(n_frames)])              # [frame][instance][H][W]
    imus = [[imus3[f][k] for k in pick] for f in range(n_frames)]
    hb = hip.RvioHip(cfg, batch=B, front_end=True)
    hs = [hip.RvioHip(cfg) for _ in range(B)]
    for i, q in enumerate(seqs):
        w, a, n = q.init_from_static(38)
        hs[i].initialize(w, a, n)
        if i == 0:
            hb.initialize(w, a, n)
        hb.set_state_at(i, *hs[i].get_state())
    keep = []
    for f in range(n_frames):
        m = len(imus[f][0])
        assert all(len(u) == m for u in imus[f])
        imu = np.stack(imus[f])
        d_img = torch.from_numpy(imgs[f]).cuda()
        d_imu = torch.from_numpy(imu.view(np.uint8).reshape(B, -1)).cuda()
        keep += [d_img, d_imu]
        torch.cuda.synchronize()
        hb.frame_batch_dev(d_img.data_ptr(), cfg.width, cfg.width * cfg.height, d_imu.data_ptr(), m, m)
        for i in range(B):
            hs[i].frame_dev(d_img[i].data_ptr(), cfg.width, d_imu[i].data_ptr(), m, 0, 0)
    hb.sync()
    n_upd = 0
    for i in range(B):
        hs[i].sync()
        xa, Pa = hb.get_state_at(i)
        xb, Pb = hs[i].get_state()
        assert same_filter_state(xa, Pa, xb, Pb), i
        n_upd += hs[i].frame_info()["updated"]
        pa, la = hb.get_points_at(i)
        pb, lb = hs[i].get_points()
        assert np.array_equal(pa, pb) and np.array_equal(la, lb), i
    assert not np.array_equal(hb.get_state_at(0)[0], hb.get_state_at(1)[0])      # the streams really differ
    hb.close()
    for h in hs:
        h.close()


@pytest.mark.parametrize("case", ["odd-750x481", "D-1920x1080"])
def test_batch_front_end_other_image_sizes(gpu_required, case):
    """the throughput forms need word-aligned rows: a width that is not a multiple of 4 must fall back to the 1-pixel forms kernel by
    kernel (750 x 481: also a CLAHE grid that needs padding), and 1920 x 1080 / 800 features takes the detector's general path —
    both bit-identical to plain handles"""
    from rvio_amd import hip
    import torch
    if case.startswith("odd"):
        cfg = abi.config_named("B", width=750, height=481)
        n_frames = 5
    else:
        cfg = abi.config_named("D")
        n_frames = 3
    B, k0 = 8, 60
    seqs2 = [rv.synth.SynthSequence(cfg, duration=5.0, seed=s) for s in range(2)]
    imgs2 = [[q.render(k0 + f) for q in seqs2] for f in range(n_frames)]
    imus2 = [[q.imu_between(k0 + f) for q in seqs2] for f in range(n_frames)]
    pick = [b % 2 for b in range(B)]
    hb = hip.RvioHip(cfg, batch=B, front_end=True)
    hs = [hip.RvioHip(cfg) for _ in range(2)]
    for i, q in enumerate(seqs2):
        w, a, n = q.init_from_static(38)
        hs[i].initialize(w, a, n)
        if i == 0:
            hb.initialize(w, a, n)
    for b in range(B):
        hb.set_state_at(b, *hs[pick[b]].get_state())
    keep = []
    for f in range(n_frames):
        m = len(imus2[f][0])
        assert len(imus2[f][1]) == m
        d_img = torch.from_numpy(np.stack([imgs2[f][k] for k in pick])).cuda()
        d_imu = torch.from_numpy(np.stack([imus2[f][k] for k in pick]).view(np.uint8).reshape(B, -1)).cuda()
        keep += [d_img, d_imu]
        torch.cuda.synchronize()
        hb.frame_batch_dev(d_img.data_ptr(), cfg.width, cfg.width * cfg.height, d_imu.data_ptr(), m, m)
        for i in range(2):
            hs[i].frame_dev(d_img[i].data_ptr(), cfg.width, d_imu[i].data_ptr(), m, 0, 0)
    hb.sync()
    for i in range(2):
        hs[i].sync()
    for b in range(B):
        xa, Pa = hb.get_state_at(b)
        xb, Pb = hs[pick[b]].get_state()
        assert same_filter_state(xa, Pa, xb, Pb), (case, b)
        pa, la = hb.get_points_at(b)                     # the tracker's feature list and history lengths (the window is still too short
        pb, lb = hs[pick[b]].get_points()                # for an update after these few frames, so the states alone would not see the tracker)
        assert len(pa) > 100 and np.array_equal(pa, pb) and np.array_equal(la, lb), (case, b)
    hb.close()
    for h in hs:
        h.close()


@pytest.mark.parametrize("B", [3, 128])
def test_one_update_batch_vs_plain_is_rounding_only(gpu_required, recs3, B):
    """ONE frame from identical (x, P) through a batch handle (solve6 behind gemm_T, compact share reduction, feat_build<4>) and through a
    plain handle (solve7, tiled reduction, feat_build<16>): the same algorithms with a different summation grouping — 1e-13, two orders under
    the free-running bound, so that a real divergence of a throughput form cannot hide behind the sequence-level tolerance."""
    from rvio_amd import hip
    import torch
    cfg, recs = recs3
    n_seq = len(recs)
    recs = spread(recs, B)      # (B = 128: the forms of large batches, joseph_batch_kernel among them)
    hb = hip.RvioHip(cfg, batch=B)
    h1 = hip.RvioHip(cfg)
    worst_x, worst_p, n = 0.0, 0.0, 0
    for f in (8, 14, 20, 25):
        rf = [recs[i][f] for i in range(B)]
        if not all(r["did_update"] for r in rf):
            continue
        ncl = (len(rf[0]["x0"]) - 26) // 7
        assert all((len(r["x0"]) - 26) // 7 == ncl for r in rf)
        hb.set_state(rf[0]["x0"], rf[0]["P0"])
        for i in range(B):
            hb.set_state_at(i, rf[i]["x0"], rf[i]["P0"])
        hb.L.rvio_hip_frame_plan(hb.h, None, None)      # (img_count > 1 from here on: augmentation on, as in the recorded frame)
        hb.L.rvio_hip_frame_plan(hb.h, None, None)
        n_feat, types, lens, meas, imu, m = pack_inputs(cfg, rf)
        d = [torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).cuda() for a in (imu, n_feat, types, lens, meas)]
        torch.cuda.synchronize()
        hb.frame_tracks_dev(d[0].data_ptr(), m, m, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr())
        hb.sync()
        plain = []
        for i in range(n_seq):
            h1.set_state(rf[i]["x0"], rf[i]["P0"])
            h1.propagate(rf[i]["inp"]["imu"])
            h1.update(rf[i]["types"], rf[i]["lens"], rf[i]["meas"])
            h1.augment_compose(True)
            plain.append(h1.get_state())
        for i in range(B):
            xa, Pa = hb.get_state_at(i)
            xb, Pb = plain[i % n_seq]
            assert xa.shape == xb.shape
            worst_x = max(worst_x, S.state_delta(xa, xb))
            worst_p = max(worst_p, float(np.max(np.abs(Pa - Pb))) / float(np.max(np.abs(Pb))))
            n += 1
    hb.close()
    h1.close()
    assert n >= 6 and worst_x <= 1e-13 and worst_p <= 1e-13, (n, worst_x, worst_p)


@pytest.mark.parametrize("name,frames,B", [("A", 30, 2), ("E", 44, 2), ("A", 30, 128)], ids=["A-30", "E-44", "A-30-128-instances"])
def test_batch_handles_cover_the_long_windows(gpu_required, name, frames, B):
    """cfg A (the stock 14-clone window) and cfg E (config 5 of BASELINE.json: 1600 features / 30 clones, 6n = 180): the instance-sharded
    fleet — the multi-GPU mode that scales — needs batch handles at these windows too.  Two differently seeded instances behind one handle
    against the recorded oracle states and against plain handles.  (128 instances at cfg A: the large-batch forms at a window of 6 x 6 tiles —
    gram_reduce_batch_kernel<6>, the fat augmentation workgroups; the k-loop Joseph kernels, 6n = 84 being beyond joseph_batch_kernel.)"""
    from rvio_amd import hip
    import torch
    cfg = abi.config_named(name, enable_equalizer=0)
    n_seq = 2
    recs = spread([S.record_sequence(cfg, n_frames=frames, seed=s, duration=5.0)[1] for s in range(n_seq)], B)
    hb = hip.RvioHip(cfg, batch=B)
    hs = [hip.RvioHip(cfg) for _ in range(n_seq)]
    hb.set_state(recs[0][0]["x0"], recs[0][0]["P0"])
    for i in range(B):
        hb.set_state_at(i, recs[i][0]["x0"], recs[i][0]["P0"])
    for i in range(n_seq):
        hs[i].set_state(recs[i][0]["x0"], recs[i][0]["P0"])
    n_upd = 0
    for f in range(frames):
        rf = [recs[i][f] for i in range(B)]
        n_feat, types, lens, meas, imu, m = pack_inputs(cfg, rf)
        d = [torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).cuda() for a in (imu, n_feat, types, lens, meas)]
        torch.cuda.synchronize()
        hb.frame_tracks_dev(d[0].data_ptr(), m, m, d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr())
        for i in range(n_seq):
            do_update, do_augment = hs[i].frame_plan()
            hs[i].propagate(imu[i])
            if do_update:
                hs[i].update(rf[i]["types"], rf[i]["lens"], rf[i]["meas"])
                n_upd += 1
            hs[i].augment_compose(do_augment)
        hb.sync()
        plain = [h.get_state() for h in hs]
        for i in range(B):
            xa, Pa = hb.get_state_at(i)
            xb, Pb = plain[i % n_seq]
            assert same_filter_state(xa, Pa, xb, Pb, 1e-10), (name, f, i)
            assert S.state_delta(xa, rf[i]["x3"]) <= 1e-8, (name, f, i)
    assert n_upd > frames and (len(xa) - 26) // 7 == cfg.max_track_len - 1
    hb.close()
    for h in hs:
        h.close()
