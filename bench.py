#!/usr/bin/env python3
"""bench.py — camera frames/s of the MI355X-native R-VIO hot path (BASELINE.json metric).

A "step" is one pass of the per-frame hot path (System::MonoVIO timed body, System.cc:253-367:
CLAHE + corner detection + KLT track + RANSAC + book-keeping -> IMU propagate -> MSCKF update ->
augment/compose) over one synthetic EuRoC-shaped 752x480 frame with its ~10 IMU samples, all
resident in HBM before the timed region.  N=1 workload = BASELINE.json configs[1] (cfg B: 200
features, 10-clone window).  N>1: the feature-sharded updater — every rank runs the
replicated front end + propagate, builds the Jacobians / nullspace / gate / compression of its
feature shard, and the per-rank information blocks [A|b] are exchanged with ONE all-gather (RCCL)
per frame, enqueued by the library on the handle's filter stream (rvio_hip_frame_sharded_dev: one
C-ABI call per frame); the same frames are processed by the whole group => "scaling": "strong".
Every frame index goes through FrameSet.args, which refuses anything outside the resident sequence;
secondary legs fail soft ({"error": ...}) — tests/test_gpu_bench.py runs the driver's own command.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from pkgload import load_pkg  # noqa: E402

rv = load_pkg()
abi, synth = rv.abi, rv.synth

K0 = 38  # last stationary frame of the synthetic sequence (t = 1.9 s)
PARITY_FRAMES = 141   # frames of the free-running device-vs-CPU comparison
ROOFLINE_FRAMES = 141  # frames the stage latencies and the roofline candidates are measured after (fixed: independent of --steps)
PEAK_F64_ = 78.6      # TFLOP/s FP64 (vector == matrix), public MI355X spec
CPU_WARM, CPU_TIMED = 50, 300   # the CPU baseline is p50 / p95 over >= 300 frames after 50 warm-up frames


def build_inputs(cfg, n_frames, seed=0):
    """Render the sequence on the host (untimed): frames, IMU batches, detector corner lists."""
    seq = synth.SynthSequence(cfg, duration=(K0 + n_frames + 3) / 20.0, seed=seed)
    F = cfg.n_features
    imgs = np.zeros((n_frames, cfg.height, cfg.width), np.uint8)
    imus, cands = [], []
    mmax = 0
    for i in range(n_frames):
        k = K0 + 1 + i
        imgs[i] = seq.render(k)
        xy, vis = seq.project(k, noise=False)
        cand, _ = seq.candidates(k, xy, vis)
        imu = seq.imu_between(k)
        imus.append(imu)
        cands.append(cand)
        mmax = max(mmax, len(imu))
    imu_arr = np.zeros((n_frames, mmax), dtype=abi.IMU_DTYPE)
    imu_cnt = np.zeros(n_frames, np.int32)
    cand_arr = np.zeros((n_frames, F, 2), np.float32)
    cand_cnt = np.zeros(n_frames, np.int32)
    for i in range(n_frames):
        imu_arr[i, : len(imus[i])] = imus[i]
        imu_cnt[i] = len(imus[i])
        cand_arr[i, : len(cands[i])] = cands[i]
        cand_cnt[i] = len(cands[i])
    return seq, imgs, imu_arr, imu_cnt, cand_arr, cand_cnt


class DeviceArray:
    """__cuda_array_interface__ view of a raw device pointer (torch.as_tensor wraps it without a copy)."""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class FrameSet:
    """A rendered input sequence resident in HBM + the only place where a frame index becomes a device pointer: `args(i)` checks
    0 <= i < n before any arithmetic (a bad device pointer cannot be caught afterwards — the GPU faults and the process aborts)."""

    def __init__(self, torch, cfg, imgs, imu_arr, imu_cnt, cand_arr=None, cand_cnt=None):
        self.n = int(len(imgs))
        self.cfg = cfg
        self.imgs, self.imu_arr, self.imu_cnt = imgs, imu_arr, np.asarray(imu_cnt)
        self.cand_arr = cand_arr
        self.cand_cnt = np.zeros(self.n, np.int32) if cand_arr is None else np.asarray(cand_cnt)
        assert len(imu_arr) == self.n and len(self.imu_cnt) == self.n and len(self.cand_cnt) == self.n
        self.d_imgs = torch.from_numpy(imgs).cuda()
        self.d_imu = torch.from_numpy(imu_arr.view(np.uint8).reshape(self.n, -1)).cuda()
        self.d_cand = None if cand_arr is None else torch.from_numpy(cand_arr).cuda()
        self.isb = cfg.width * cfg.height
        self.msb = int(self.d_imu.shape[1])
        self.csb = 0 if cand_arr is None else cfg.n_features * 2 * 4
        self.p_img, self.p_imu = self.d_imgs.data_ptr(), self.d_imu.data_ptr()
        self.p_cand = 0 if cand_arr is None else self.d_cand.data_ptr()
        assert self.d_imgs.numel() == self.n * self.isb

    @classmethod
    def fake(cls, cfg, n, m=10, base=1 << 40):
        """no device: made-up base addresses, for the CPU test that walks every leg's index arithmetic (tests/test_bench_helpers.py)"""
        self = cls.__new__(cls)
        self.n, self.cfg = int(n), cfg
        self.imgs = np.zeros((n, 1, 1), np.uint8)
        self.imu_arr = np.zeros((n, m), dtype=abi.IMU_DTYPE)
        self.imu_cnt = np.full(n, m, np.int32)
        self.cand_arr, self.cand_cnt = None, np.zeros(n, np.int32)
        self.isb, self.msb, self.csb = cfg.width * cfg.height, m * abi.IMU_DTYPE.itemsize, 0
        self.p_img, self.p_imu, self.p_cand = base, 2 * base, 0
        return self

    def check(self, i):
        i = int(i)
        if not 0 <= i < self.n:
            raise IndexError("frame index %d outside the resident sequence [0, %d)" % (i, self.n))
        return i

    def args(self, i):
        """(d_img, stride, d_imu, m, d_cand, n_cand) of frame i for rvio_hip_frame_dev / rvio_hip_track_dev"""
        i = self.check(i)
        return (self.p_img + i * self.isb, self.cfg.width, self.p_imu + i * self.msb, int(self.imu_cnt[i]),
                self.p_cand + i * self.csb, int(self.cand_cnt[i]))

    def host(self, i):
        """(img, imu, cand) of frame i as host arrays for rvio_hip_frame"""
        i = self.check(i)
        return self.imgs[i], self.imu_arr[i, : self.imu_cnt[i]], None if self.cand_arr is None else self.cand_arr[i, : self.cand_cnt[i]]


def pose_latency_plan(n_frames, max_track_len, want=60, min_timed=8):
    """(n_warm, n_timed) of the pose-latency leg for a resident sequence of n_frames frames, or None when the sequence is too short to
    fill the window first (the leg then runs on the longer parity sequence).  0 <= n_warm, n_warm + n_timed <= n_frames always."""
    n_warm = 3 * max_track_len + 10
    if n_frames < n_warm + min_timed:
        return None
    return n_warm, min(want, n_frames - n_warm)


def safe_leg(out, key, fn, *a, **kw):
    """A secondary leg never costs the headline: a Python-level failure becomes {"error": ...} in its object."""
    try:
        out[key] = fn(*a, **kw)
    except Exception as e:   # noqa: BLE001
        import traceback
        out[key] = {"error": repr(e)[:300], "where": traceback.format_exc().strip().splitlines()[-3:][0].strip()[:200]}
    return out[key]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", default="B", choices=list(abi.BASELINE_CONFIGS))
    ap.add_argument("--cpu-frames", type=int, default=120, help="frames of the same sequence timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-streams", action="store_true")
    ap.add_argument("--no-equalizer", action="store_true", help="Tracker.EnableEqualizer: 0 (skip CLAHE)")
    ap.add_argument("--force-sharded", action="store_true", help="run the sharded-updater frame path (RCCL all-gather) even with one rank")
    ap.add_argument("--host-corners", action="store_true",
                    help="feed a caller-side corner list (projected landmarks) instead of running the device detector")
    ap.add_argument("--batch", default="1,16,256,2048", help="instance counts of the batched-filter leg (SURVEY.md 8d (ii)); '' skips it")
    ap.add_argument("--batch-streams", default="1,16,128", help="instance counts of the batched camera-stream leg (whole frame, B streams per launch); '' skips it")
    ap.add_argument("--no-defined-load", action="store_true", help="skip the batched filter at SURVEY 8d's defined load (profiling passes of the natural-flow leg)")
    ap.add_argument("--streams", type=int, default=8, help="independent filter instances for the aggregate-throughput leg")
    ap.add_argument("--stream-threads", type=int, default=1, help="host threads issuing the launches of the aggregate-throughput leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rvio_amd import hip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # Tracker.EnableEqualizer: 1 in the stock config (config/rvio_euroc.yaml): CLAHE runs on every frame
    cfg = abi.config_named(args.config, enable_equalizer=0 if args.no_equalizer else 1)
    K, W = args.steps, args.warmup
    if K < 1 or W < 0:
        raise SystemExit("--steps >= 1 and --warmup >= 0")
    n_frames = 1 + W + K   # first image (seed) + warmup + timed
    seq, imgs, imu_arr, imu_cnt, cand_arr, cand_cnt = build_inputs(cfg, n_frames)
    if not args.host_corners:
        # stock behaviour: FeatureDetector::DetectWithSubPix runs inside the library (NULL corner list), also in the CPU baseline
        cand_arr, cand_cnt = None, np.zeros_like(cand_cnt)

    h = hip.RvioHip(cfg, device=local_rank)
    stream = torch.cuda.ExternalStream(h.stream(), device=torch.device("cuda", local_rank))
    fs = FrameSet(torch, cfg, imgs, imu_arr, imu_cnt, cand_arr, cand_cnt)
    torch.cuda.synchronize()

    wi, ai, ni = seq.init_from_static(K0)
    h.initialize(wi, ai, ni)

    gathered, comm = None, None
    if sharded:
        if not os.environ.get("RVIO_TORCH_COLLECTIVE"):
            from rvio_amd import rccl
            try:
                comm = rccl.RcclComm(rank, world, dist, torch)
            except (OSError, RuntimeError, AttributeError) as e:
                print("rank %d: direct RCCL communicator unavailable (%s); using the process-group collective" % (rank, e), file=sys.stderr)
            ok = torch.tensor([1 if comm is not None else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # every rank takes the same path
            if int(ok.item()) == 0 and comm is not None:
                comm.close()
                comm = None
        nblk = abi.shard_payload_doubles(6 * (cfg.max_track_len - 1), cfg.max_track_len)    # a full window's payload per rank (rvio_hip_update_local; the wire format of csrc/rvio_dev.h shard_layout)
        gathered = torch.zeros(world * nblk, dtype=torch.float64, device="cuda")

    def make_frame(h_, fs_, gathered_):
        def frame_(i):
            if not sharded:
                h_.frame_dev(*fs_.args(i))
            elif comm is not None:
                # ONE C-ABI call per frame: pipelined like N=1, the ncclAllGather enqueued by the library on the handle's filter stream
                h_.frame_sharded_dev(*fs_.args(i), rank, world, comm.comm)
            else:
                # fall-back without a direct RCCL communicator: the same frame split open, the collective through torch.distributed
                st_ = torch.cuda.ExternalStream(h_.stream(), device=torch.device("cuda", local_rank))
                h_.frame_sharded_piped(*fs_.args(i), rank, world, gathered_, dist, DeviceArray, torch, st_, force_collective=args.force_sharded, comm=None)
        return frame_

    frame = make_frame(h, fs, gathered)

    for i in range(1 + W):
        frame(i)
    h.sync()
    torch.cuda.synchronize()
    if sharded:      # RCCL may have printed its version banner through C stdio: push it out now, the JSON line must come last
        import ctypes
        ctypes.CDLL(None).fflush(None)
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record()
    for i in range(1 + W, n_frames):
        frame(i)
    t_enq = time.perf_counter() - t0          # host time to enqueue the K frames (launch-rate bound if close to the total)
    with torch.cuda.stream(stream):
        ev1.record()
    h.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    gpu_ms = ev0.elapsed_time(ev1)
    info = h.frame_info()
    x_gpu, P_gpu = h.get_state()
    h.close()      # (the first live handle of a process owns private hardware queues: every secondary leg below measures a handle of its own)

    out = {
        "metric": "camera frames/sec (KLT+RANSAC track, IMU propagate, MSCKF update, augment/compose), 200 feat / 10-clone window",
        "value": K / elapsed, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None, "dtype": "f64 (filter) / u8+i32+f32 (KLT)", "data": "synthetic",
        "config": {"workload": "cfg%s: synthetic EuRoC-shaped %dx%d @20Hz, %d features, %d-clone window, IMU 200 Hz, single stream, equalizer %s, corners from %s"
                               % (args.config, cfg.width, cfg.height, cfg.n_features, cfg.max_track_len - 1, "on (CLAHE 3.0, 5x5)" if cfg.enable_equalizer else "off",
                                  "a caller-side list" if args.host_corners else "the device detector (GFTT + cornerSubPix)"),
                   "parallelism": ("1 process/GPU; feature-sharded updater + 1 all-gather/frame (%s)"
                                   % ("rvio_hip_frame_sharded_dev: one C-ABI call per frame, ncclAllGather on the filter stream" if comm is not None else "torch.distributed")) if sharded else "single GPU"},
        "gpu_ms_per_step_events": gpu_ms / K, "host_enqueue_ms_per_step": 1e3 * t_enq / K,
        "last_frame": {k: info[k] for k in ("n_tracked_in", "n_klt_ok", "n_ransac_inliers", "n_feat_update", "n_feat_accepted", "n_rows", "updated", "device_error")},
    }

    # the long sequence the parity leg, the CPU baseline and (when --steps is short) the pose-latency leg run on: >= PARITY_FRAMES
    # frames whatever --steps says (the reference's rank truncation first bites at frame 51 of the stock sequence: a 26-frame
    # comparison would not see it, and a 10-clone window is not even full after 26 frames)
    long_in = None

    def long_inputs():
        nonlocal long_in
        if long_in is None:
            need = PARITY_FRAMES if args.no_cpu else max(PARITY_FRAMES, CPU_WARM + CPU_TIMED)
            if n_frames >= need:
                long_in = (imgs, imu_arr, imu_cnt, cand_arr, cand_cnt)
            else:
                _, pi, pa, pc, pca, pcc = build_inputs(cfg, need)
                long_in = (pi, pa, pc, pca, pcc) if args.host_corners else (pi, pa, pc, None, np.zeros_like(pcc))
        return long_in

    if sharded and not args.no_streams:
        def side_modes():
            # beside the sharded figure: every rank ALSO runs the same K frames as an independent camera stream on its own handle (N cameras
            # on N GPUs, no collective) — the weak-scaling counterpart of `value`, reported in its own object
            h2 = hip.RvioHip(cfg, device=local_rank)
            h2.initialize(wi, ai, ni)
            for i in range(1 + W):
                h2.frame_dev(*fs.args(i))
            h2.sync()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            for i in range(1 + W, n_frames):
                h2.frame_dev(*fs.args(i))
            h2.sync()
            if world > 1:
                dist.barrier()
            el2 = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([el2], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el2 = float(tt.item())
            h2.close()
            out["independent_streams"] = {"value": world * K / el2, "unit": "frames/s", "scaling": "weak",
                                          "note": "one camera stream per GPU, no collective; `value` above is ONE stream with its updater sharded over the GPUs"}
            # the multi-GPU mode that pays at this window size: a FLEET of filter instances sharded by instance —
            # rank r owns instances r, r + N, ... of a fixed fleet, one batch handle per GPU, no collective at all (strong scaling of the fleet)
            fleet = 2048
            per = max(1, fleet // world)
            leg = batched_filter_leg(cfg, torch, [per], name=args.config, seed0=4 * rank, barrier=(dist.barrier if world > 1 else None))
            el3 = torch.tensor([leg["sizes"][0]["ms_per_batched_frame"]], dtype=torch.float64, device="cuda")
            if world > 1:
                dist.all_reduce(el3, op=dist.ReduceOp.MAX)
            ms = float(el3.item())
            out["instance_sharded_fleet"] = {"instances_total": per * world, "instances_per_gpu": per, "ms_per_fleet_frame": ms,
                                             "filter_frames_per_s": per * world / (ms * 1e-3), "scaling": "strong", "collectives_per_frame": 0,
                                             "algorithmic_mflop_per_filter_frame": leg["algorithmic_mflop_per_filter_frame"],
                                             "achieved_tflops_fp64_per_gpu": leg["algorithmic_mflop_per_filter_frame"] * 1e6 * per / (ms * 1e-3) / 1e12,
                                             "note": "rvio_hip_create_batch per GPU, instances r, r+N, ... of a %d-instance fleet on rank r; max over ranks of the batched-frame time" % (per * world)}
        def sharded_cfg_e(WE=36, KE=24):
            # the configuration north_star designs the feature split around (cfg E: 1600 features, 30-clone window): the SAME sharded frame path at that
            # size, window full, every rank rendering the same frames — `value` above stays the headline configuration
            cfgE = abi.config_named("E", enable_equalizer=0 if args.no_equalizer else 1)
            nE = 1 + WE + KE
            seqE, imgsE, imuE, cntE, _, ccE = build_inputs(cfgE, nE)
            fsE = FrameSet(torch, cfgE, imgsE, imuE, cntE, None, np.zeros_like(ccE))
            hE = hip.RvioHip(cfgE, device=local_rank)
            hE.initialize(*seqE.init_from_static(K0))
            nblkE = abi.shard_payload_doubles(6 * (cfgE.max_track_len - 1), cfgE.max_track_len)
            gE = torch.zeros(world * nblkE, dtype=torch.float64, device="cuda")
            fE = make_frame(hE, fsE, gE)
            for i in range(1 + WE):
                fE(i)
            hE.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            for i in range(1 + WE, nE):
                fE(i)
            hE.sync()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            elE = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([elE], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                elE = float(tt.item())
            infoE = hE.frame_info()
            hE.close()
            out["sharded_cfgE"] = {"value": KE / elE, "unit": "frames/s", "ms_per_step": 1e3 * elE / KE, "steps": KE, "warmup": WE, "n_gpus": world, "scaling": "strong",
                                   "workload": "cfgE: %dx%d, %d features, %d-clone window (full after %d frames), feature-sharded updater, 1 all-gather of the information block per frame"
                                               % (cfgE.width, cfgE.height, cfgE.n_features, cfgE.max_track_len - 1, cfgE.max_track_len),
                                   "payload_bytes_per_rank": int(8 * nblkE),
                                   "last_frame": {k: infoE[k] for k in ("n_feat_update", "n_feat_accepted", "n_rows", "updated", "device_error")}}
        if world > 1:
            side_modes()          # (collectives inside: a one-sided failure would dead-lock the group anyway)
            if args.config != "E":
                sharded_cfg_e()
        else:
            try:
                side_modes()
            except Exception as e:   # noqa: BLE001
                out["independent_streams"] = {"error": repr(e)[:300]}
            if args.config != "E":
                safe_leg(out, "sharded_cfgE", lambda: (sharded_cfg_e(), out["sharded_cfgE"])[1])
    if rank == 0 and world == 1 and not args.no_streams:
        # (first of the extra legs: a process normally owns ONE handle.  HIP multiplexes its streams onto 4 hardware queues in creation
        # order; a handle created after dozens of other streams — the later legs — can find two of its three streams on one queue and
        # loses their overlap: 2.3 k instead of 4.1 k frames/s were measured for this leg when it ran last)
        safe_leg(out, "host_buffers", host_buffer_leg, cfg, fs, wi, ai, ni, 1 + W)
    if rank == 0 and not args.no_latency:
        # (also at N>1: the per-kernel roofline is a property of one GPU; the other ranks wait in the barrier below)
        x_single = None
        try:
            # The stage latencies and the roofline candidates are measured on a FIXED prefix of the sequence (ROOFLINE_FRAMES frames: window full,
            # the same operands whatever --steps says) so that the object does not change identity between a 20-step and a 200-step run; the
            # sharded comparison below needs the state after the timed run's own frames and keeps them.
            if sharded or fs.n == ROOFLINE_FRAMES:
                fl_ = fs
            else:
                li_ = long_inputs()
                fl_ = FrameSet(torch, cfg, *[a_[:ROOFLINE_FRAMES] if a_ is not None else None for a_ in li_])
            out.update(latency_pass(cfg, torch, fl_, wi, ai, ni, device=local_rank, name=args.config))
            x_single = out.pop("x_final")
        except Exception as e:   # noqa: BLE001
            out["latency_pass"] = {"error": repr(e)[:300]}
        if not sharded:
            def pose_leg():
                plan = pose_latency_plan(fs.n, cfg.max_track_len)
                if plan is not None:
                    return pose_latency_leg(cfg, fs, plan, wi, ai, ni, local_rank)
                li = long_inputs()
                plan = pose_latency_plan(len(li[0]), cfg.max_track_len)
                if plan is None:
                    return {"skipped": "window of %d clones needs %d frames to fill; %d resident" % (cfg.max_track_len - 1, 3 * cfg.max_track_len + 18, len(li[0]))}
                fl = FrameSet(torch, cfg, *li)
                return dict(pose_latency_leg(cfg, fl, plan, wi, ai, ni, local_rank), sequence="the %d-frame parity sequence (--steps too short to fill the window)" % fl.n)
            safe_leg(out, "pose_latency_unpipelined", pose_leg)
        if sharded and x_single is not None:   # the sharded updater against the same frames through the one-GPU updater (block sums in rank order: rounding only)
            out["max_state_delta_sharded_vs_single_gpu"] = float(np.max(np.abs(_qfix(x_gpu) - _qfix(x_single))))
        out.pop("x_at_cpu_frames", None)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1:
        if not args.no_streams:
            safe_leg(out, "multi_stream", multi_stream, cfg, fs, wi, ai, ni, 1 + W, streams=args.streams, threads=args.stream_threads)
        if not args.no_latency:
            safe_leg(out, "update_at_load", update_at_load_leg, cfg, torch, name=args.config)
        if args.batch_streams:
            safe_leg(out, "batched_streams", batched_streams_leg, cfg, torch, [int(b) for b in args.batch_streams.split(",") if b], name=args.config)
        if args.batch:
            safe_leg(out, "batched_filter", batched_filter_leg, cfg, torch, [b for b in args.batch.split(",") if b], name=args.config)
            big = [parse_batch_size(b)[0] for b in args.batch.split(",") if b and parse_batch_size(b)[0] >= 256 and parse_batch_size(b)[1] == 1]
            if big and not args.no_defined_load:
                safe_leg(out, "batched_filter_at_defined_load", batched_at_load_leg, cfg, torch, big[-1:], name=args.config)
        bf = out.get("batched_filter", {}).get("sizes") if isinstance(out.get("batched_filter"), dict) else None
        if bf:
            top = max(bf, key=lambda e: e["instances"])
            rb = {"bound": "mfma", "kernel": "the batched filter frame: propagate, per-feature stage, reduction, solve, Joseph form, augment/compose for %d instances per launch" % top["instances"],
                  "achieved": top["achieved_tflops_fp64"], "peak": top["achieved_tflops_fp64"] / top["frac_fp64_peak"] if top["frac_fp64_peak"] else None, "unit": "TFLOP/s",
                  "frac": top["frac_fp64_peak"], "by": "W_filter of SURVEY.md 8(d) (algorithmic FP64 work of the frames run) / wall time of the timed batched frames",
                  "ms_per_batched_frame": top["ms_per_batched_frame"], "traffic": None}
            mc = os.path.join(ROOT, "profiles", "r06_batched_mfma_util.json")
            if not os.path.exists(mc):
                mc = os.path.join(ROOT, "profiles", "r05_batched_mfma_util.json")
            if os.path.exists(mc):   # the matrix pipe's own counter (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64, its own pass): committed, cannot be read live
                try:
                    with open(mc) as f:
                        cm = json.load(f)
                    if cm.get("instances") == top["instances"]:
                        a = cm["mfma_flop_per_batched_frame"] / (top["ms_per_batched_frame"] * 1e-3) / 1e12
                        rb["by_mfma_counter"] = {"mfma_flop_per_batched_frame": cm["mfma_flop_per_batched_frame"], "achieved": a, "unit": "TFLOP/s", "frac": a / PEAK_F64_,
                                                 "what": "FP64 MFMA flop the matrix pipe itself counted for one batched frame (committed rocprofv3 --pmc pass, "
                                                         "profiles/%s) / the frame time measured live here" % os.path.basename(mc)}
                    else:
                        rb["by_mfma_counter"] = {"skipped": "committed counters are for %s instances" % cm.get("instances")}
                except Exception as e:   # noqa: BLE001
                    rb["by_mfma_counter"] = {"error": repr(e)[:100]}
            dl = out.get("batched_filter_at_defined_load", {}).get("sizes") if isinstance(out.get("batched_filter_at_defined_load"), dict) else None
            if dl:
                rb["at_defined_load"] = {k: {"frac": v["frac_fp64_peak"], "achieved": v["achieved_tflops_fp64"], "ms_per_batched_frame": v["ms_per_batched_frame"]}
                                         for k, v in dl[-1].items() if isinstance(v, dict) and "frac_fp64_peak" in v}
            out["roofline_batched"] = rb
        if not args.no_cpu:
            _CFG_NAME[0] = args.config
            try:
                pin = long_inputs()
                # (the CPU sample covers the frames of the timed run when that is affordable — <= 300 frames, ~4 s of oracle — so that the END
                # STATE OF THE TIMED, UN-SYNCHRONISED RUN itself is compared below, not only the synchronised replay of the parity leg)
                ncpu = min(len(pin[0]), CPU_WARM + CPU_TIMED)
                npar = min(ncpu, PARITY_FRAMES)     # the free-running device-vs-CPU comparison keeps its 141 frames (the rank truncation bites from frame 51 on)
                out["cpu_baseline"], cpu_states = cpu_baseline(cfg, seq, *pin, wi, ai, ni, ncpu)
                if n_frames <= len(cpu_states):
                    xc, Pc = cpu_states[n_frames - 1][0], cpu_states[n_frames - 1][1]
                    out["timed_run_max_state_delta"] = float(np.max(np.abs(_qfix(x_gpu) - _qfix(xc))))
                    out["timed_run_max_cov_rel_delta"] = float(np.max(np.abs(P_gpu - Pc)) / max(1e-300, float(np.max(np.abs(Pc)))))
                    out["timed_run_parity"] = {"frames": n_frames, "tolerance": 1e-6, "ok": bool(out["timed_run_max_state_delta"] <= 1e-6),
                                               "what": "state and covariance the TIMED loop itself left behind (1 + warmup + steps frames enqueued back to back, no host "
                                                       "synchronisation) against the literal CPU oracle after the same frames"}
                else:
                    out["timed_run_parity"] = {"skipped": "%d frames in the timed run, CPU sample of %d" % (n_frames, len(cpu_states))}
                if _MULTI[0] is not None:
                    out["cpu_baseline_multicore"] = _MULTI[0]
                out["parity"] = parity_leg(cfg, torch, pin, wi, ai, ni, npar, cpu_states)
                out["max_state_delta_vs_cpu"] = out["parity"]["max_state_delta"]
                # last, so that nothing above can depend on it: the same frames through the reference's own compiled sources (a child process with a time-out)
                safe_leg(out, "cpu_baseline_reference", cpu_baseline_reference, cfg, pin[0], pin[1], pin[2], pin[3], wi, ai, ni, ncpu, cpu_states[-1][0])
            except Exception as e:   # noqa: BLE001
                out.setdefault("cpu_baseline", {"error": repr(e)[:300]})
                out.setdefault("parity", {"error": repr(e)[:300]})
    if sharded:
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
        import ctypes
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        # the driver keeps the head of the line: the contract's keys first, then the objects the review reads — roofline, cpu_baseline, the drop-in
        # call on host buffers (PCIe inclusive) and the update at the defined full load — then everything else in the order it was measured
        head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "host_buffers", "update_at_load", "p50_ekf_update_ms", "timed_run_parity"]
        out = dict([(k, out[k]) for k in head if k in out] + [(k, v) for k, v in out.items() if k not in head])
        print(json.dumps(out), flush=True)


def _qfix(x):
    x = np.array(x, float)
    for i in [0, 10] + list(range(26, len(x), 7)):
        if x[i + 3] < 0:
            x[i:i + 4] *= -1
    return x


def pose_latency_leg(cfg, fs, plan, wi, ai, ni, device):
    """What a 20 Hz camera sees: host wall clock from handing ONE frame over (rvio_hip_frame_dev, nothing else in flight) until its pose is
    on the host (rvio_hip_get_pose: the filter stream only).  The Updater's hand-over leaves book-keeping before the detector has finished
    (bookkeep_a_kernel / bookkeep_b_kernel), so the pose does not wait for CLAHE + GFTT + cornerSubPix, the long pole of the front end; the
    refill for the NEXT frame finishes behind it (the next call would find it done at any real frame rate).
    `plan` = pose_latency_plan(fs.n, ...): every index below goes through fs.args, which refuses anything outside the resident sequence."""
    from rvio_amd import hip
    n_warm, n_timed = plan
    h = hip.RvioHip(cfg, device=device)
    h.initialize(wi, ai, ni)
    for i in range(n_warm):
        h.frame_dev(*fs.args(i))
    h.sync()
    ts = []
    for i in range(n_warm, n_warm + n_timed):
        a = fs.args(i)
        t0 = time.perf_counter()
        h.frame_dev(*a)
        h.pose()
        ts.append(1e3 * (time.perf_counter() - t0))
        h.sync()                                  # the refill half of book-keeping, the next frame's image chain inputs ...
    upd = h.frame_info()["updated"]
    h.close()
    return {"p50_ms": float(np.median(ts)), "p95_ms": float(np.percentile(ts, 95)), "frames": len(ts), "updated_last_frame": int(upd),
            "note": "host wall clock, frame handed over (frames resident in HBM) -> pose on the host, one frame in flight; includes the host's enqueue time"}


def latency_pass(cfg, torch, fs, wi, ai, ni, device=0, name="B"):
    """Second, untimed-for-throughput pass on a fresh handle: per-stage device latencies with HIP events on the
    handle's stream (p50 EKF-update ms of the metric), the dominant kernel's roofline, and the state after
    `cpu_frames` frames for the parity figure."""
    from rvio_amd import hip
    h = hip.RvioHip(cfg, device=device)
    st = torch.cuda.ExternalStream(h.stream(), device=torch.device("cuda", device))
    h.initialize(wi, ai, ni)
    n = fs.n
    names = ["track", "propagate", "update", "augment_compose"]
    lat = {k: [] for k in names}
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    x_at = None
    ncpu = min(120, n)
    skip = min(20, max(0, n - 4))     # (short runs: keep at least the last frames)
    for i in range(n):
        a = fs.args(i)
        with torch.cuda.stream(st):
            evs[0].record()
        h.track_dev(*a)
        with torch.cuda.stream(st):
            evs[1].record()
        did_update = h.frame_tail_staged(a[2], a[3], evs, st, torch)
        h.sync()
        if i >= skip:
            lat["track"].append(evs[0].elapsed_time(evs[1]))
            lat["propagate"].append(evs[1].elapsed_time(evs[2]))
            if did_update:
                lat["update"].append(evs[2].elapsed_time(evs[3]))
            lat["augment_compose"].append(evs[3].elapsed_time(evs[4]))
        if i == ncpu - 1:
            x_at, _ = h.get_state()
    p50 = {k: float(np.median(v)) for k, v in lat.items() if v}
    res = {"latency_ms_p50": p50, "frame_latency_ms_unpipelined": float(sum(p50.values())),
           "latency_ms_p95": {k: float(np.percentile(v, 95)) for k, v in lat.items() if v},
           "p50_ekf_update_ms": float(np.median(lat["update"])) if lat["update"] else None,
           "x_at_cpu_frames": x_at, "x_final": h.get_state()[0]}
    # ---- roofline (live, HIP events on the handle's stream; the rocprofv3 summary in profiles/ must agree).
    # Candidates = the longest kernels of the section-8(a) hot path (KLT chain -> HBM, filter -> FP64: SURVEY.md 8d), each timed by
    # rvio_hip_debug_time_kernel in the very form the handle launches, on the operands the last frame left in HBM; the DOMINANT one
    # (largest live average) is `roofline`, the others and the detector's longest kernel (cornerSubPix, section 8(f)) go to `roofline_other`.
    n = cfg.max_track_len - 1
    c6 = 6 * n
    F = cfg.n_features
    PEAK_F64 = 78.6   # TFLOP/s, FP64 vector == FP64 matrix on MI355X (public spec; not in the measured tables of the guide)
    t_solve = h.time_kernel(0, 50) * 1e-6
    t_klt = h.time_kernel(1, 50) * 1e-6
    try:
        t_feat = h.time_kernel(8, 50) * 1e-6       # feat_prop_kernel itself, as the pipelined frame launches it (state put aside and restored)
        feat_in_situ = True
    except Exception:   # noqa: BLE001  (a window whose fused launch does not fit LDS: the per-feature workgroups in a launch of their own)
        t_feat = h.time_kernel(2, 50) * 1e-6
        feat_in_situ = False
    t_subpix = h.time_kernel(6, 50) * 1e-6
    t_greedy = h.time_kernel(9, 50) * 1e-6
    # (rvio_hip.hip: the blocked SPD solve; at 6n <= 64 behind the Cholesky role its all-LDS form; at 6n > 96 its split form — six launches)
    solve_name = "solve9_small_kernel" if c6 <= 64 else "solve9_kernel" if c6 <= 96 else "solve9_prod_kernel<0..3> + solve9_sweep_kernel + solve9_dx_kernel"
    solve_match = "solve9_small_kernel" if c6 <= 64 else "solve9_kernel" if c6 <= 96 else "solve9_sweep_kernel"
    # solve: the work SURVEY.md 8d prices for U8 — T = s2 I + A Pcc (2 c6^3) and the inverse of T (c6 steps x c6 rows x (c6 + 1) columns x 2 flops).  The
    # kernel timed here runs BEHIND the Cholesky role (the factor of the clone block rides in the per-feature launch / on a queue of its own,
    # off the chain), so the flops of that factor, c6^3 / 3, are taken out of the numerator: the figure prices what the timed kernel does.
    fl_solve = 2.0 * c6 ** 3 + 2.0 * c6 * c6 * (c6 + 1) - c6 ** 3 / 3.0
    # per-feature stage (U1-U5): the gate term of W_filter for the tracks the last frame handed over, sum_f 2 rho (6n)^2 + 2 rho^2 (6n) + 4/3 rho^3
    ty_l, ln_l, _ = h.get_tracks()
    fl_feat = 0.0
    for L, t in zip(ln_l, ty_l):
        rho = max(2 * ((int(L) + 1) // 2 if t == ord("2") else int(L)) - 3, 0)
        fl_feat += 2.0 * rho * c6 ** 2 + 2.0 * rho ** 2 * c6 + (4.0 / 3.0) * rho ** 3
    it_l = 10
    by_klt = F * 4 * (16 * 16 * 5) + F * 4 * it_l * 16 * 16     # B_klt of SURVEY.md 8d, it_l = 10: template + it_l bilinear windows per level
    by_subpix = F * (4.0 * 17 * 17) * 5                          # cornerSubPix: a 17x17 float window re-sampled per iteration, ~5 iterations per corner
    n_cand_l = int(h.frame_info().get("n_tracked_in", F))
    cands = [
        {"bound": "mfma", "kernel": "%s (W = (s2 I + A Pcc)^-1, dx, state injection%s)" % (solve_name, "; one workgroup" if c6 <= 96 else "; avg_us = the six launches together"),
         "match": solve_match, "launched_by_timed_path": solve_name,
         "achieved": fl_solve / t_solve / 1e12, "peak": PEAK_F64, "unit": "TFLOP/s", "avg_us": t_solve * 1e6,
         "note": "blocked symmetric sweep of M = s2 I + L^T A L + Woodbury on FP64 MFMA tiles; 6n <= 96: one workgroup on ONE CU (%.2f TFLOP/s of the chip's %.1f), the Cholesky of the "
                 "clone block rides in the per-feature launch (off the chain: NOT in avg_us, and its c6^3 / 3 flops are not in `achieved`), dx = Pc y and the state injection are role "
                 "workgroups of the Joseph launch behind it (not in avg_us either: timed as the frame's update launches it); 6n > 96: the four product phases are "
                 "chip-wide launches, the sweep one workgroup, the Cholesky factor runs beside the filter chain on a queue of its own; latency bound: %d 16 x 16 in-wave "
                 "factorisations in sequence" % (PEAK_F64 / 256.0, PEAK_F64, (c6 + 15) // 16)},
        {"bound": "hbm", "kernel": "klt_kernel3 (4-level pyramidal LK, one wave per feature)", "match": "klt_kernel3", "launched_by_timed_path": "klt_kernel3 (forward match)",
         "achieved": by_klt / t_klt / 1e9, "peak": 8000.0, "unit": "GB/s", "avg_us": t_klt * 1e6,
         "note": "timed matching the current image back onto the previous one from the current feature positions (the forward match's displacements, "
                 "reversed): the frame's own inputs are gone once book-keeping has moved the features"},
        {"bound": "mfma", "kernel": "feat_prop_kernel (U1-U5 per feature with the FP64 MFMA gate, %d features handed over by the last frame, + PreIntegrator::propagate and the Cholesky role as "
                                    "two more workgroups)" % len(ln_l) if feat_in_situ else
                                    "the per-feature workgroups of feat_prop_kernel in a launch of their own (feat_build_kernel<16>; %d features)" % len(ln_l), "match": "feat_",
         "launched_by_timed_path": "feat_prop_kernel" + ("" if feat_in_situ else " (timed as feat_build_kernel<16>: the fused launch of this window does not fit LDS)"),
         "achieved": (fl_feat / t_feat / 1e12) if fl_feat > 0 else None, "peak": PEAK_F64, "unit": "TFLOP/s", "avg_us": t_feat * 1e6},
    ]
    det = [{"bound": "hbm", "kernel": "subpix_kernel (cornerSubPix, 4 waves per corner; detector = section 8(f))", "match": "subpix_kernel", "launched_by_timed_path": "subpix_kernel",
            "achieved": by_subpix / t_subpix / 1e9, "peak": 8000.0, "unit": "GB/s", "avg_us": t_subpix * 1e6},
           {"bound": "hbm", "kernel": "greedy_kernel (goodFeaturesToTrack's min-distance selection as a priority-ordered maximal independent set, ONE workgroup; detector = section 8(f))",
            "match": "greedy_kernel", "launched_by_timed_path": "greedy_kernel",
            "achieved": None, "peak": 8000.0, "unit": "GB/s", "avg_us": t_greedy * 1e6,
            "note": "a serial selection over the candidate lists in LDS: no bytes or flops worth pricing (latency of the rounds); listed because it is one of the longest kernels "
                    "of the image chain, which runs two frames deep beside the chains that set the period"}]
    for c in cands + det:
        c["frac"] = None if c["achieved"] is None else c["achieved"] / c["peak"]
        # HBM-side bytes per launch: PMC counters cannot be read live, so this is the committed rocprofv3 --pmc result OF THIS CONFIGURATION
        # (profiles/r06_pmc_traffic_cfg<name>.json, else an earlier round's: FETCH_SIZE + WRITE_SIZE as reported, separate passes) or null
        c["traffic"], c["traffic_unit"] = pmc_traffic(name, c.pop("match"))
    # identity of `roofline`: the section-8(a) kernel with the largest AVERAGE in the committed rocprofv3 kernel trace of the default run of THIS configuration
    # (profiles/r06_kernel_stats_stream[_cfgX].md: every frame of the run, the fast-motion ones included) — a fixed identity per tree, the same for a 20-step and
    # a 200-step invocation; kernels whose duration contains a device-side wait (ransac_book_kernel, stage_gate_kernel) are not candidates.  Without a committed
    # trace: the largest live average.  `achieved` is always the LIVE figure (HIP events, 50 launches, the fixed state after frame 140).
    trace = kernel_trace_averages(name)
    for c in cands + det:
        tm = [v for k, v in trace.items() if c["launched_by_timed_path"].split()[0].split("<")[0] in k]
        if tm and "+" not in c["launched_by_timed_path"]:       # (the split solve is six launches: no single trace row stands for it — live rule for those windows)
            c["trace_avg_us"], c["trace_median_us"] = tm[0]
            if c["achieved"] is not None:       # the same algorithmic work over the AVERAGE launch of the traced run (every frame, the fast-motion ones included): the conservative figure
                c["achieved_at_trace_avg"] = c["achieved"] * c["avg_us"] / c["trace_avg_us"]
                c["frac_at_trace_avg"] = c["achieved_at_trace_avg"] / c["peak"]
    if all("trace_avg_us" in c for c in cands):
        cands.sort(key=lambda c: -c["trace_avg_us"])
        how = ("the largest average in the committed kernel trace of the default run (profiles/%s: %s) among the section-8(a) kernels of the frame; kernels whose duration contains a "
               "device-side wait are not candidates" % (trace_file(name), ", ".join("%s %.1f us" % (c["launched_by_timed_path"].split()[0], c["trace_avg_us"]) for c in cands)))
    else:
        cands.sort(key=lambda c: -c["avg_us"])
        how = "no committed kernel trace for this configuration: the largest live average"
    res["roofline"] = dict(cands[0], dominant_by="rule: %s.  avg_us / achieved are LIVE (HIP events, 50 launches on the handle's stream, rvio_hip_debug_time_kernel, each kernel in the very "
                                                  "form the pipelined frame launches it) on the operands left after frame %d of the stock sequence — a FIXED state (window full), independent "
                                                  "of --steps; trace_avg_us is the average over every frame of the traced run, the fast-motion frames included (KLT iterates until its "
                                                  "slowest feature has converged).  Every candidate is listed (roofline + roofline_other) with its own fraction" % (how, fs.n - 1),
                           context="a single 752x480 stream offers 51 MFLOP and 3.7 MB per frame (SURVEY.md 8d), i.e. <<1% of either roof by construction; "
                                   "update_at_load.roofline prices the whole update at full load, batched_filter / batched_streams the same kernels with the chip full")
    res["roofline_other"] = cands[1:] + det
    h.close()
    return res


def trace_file(cfg_name):
    return "r06_kernel_stats_stream.md" if cfg_name == "B" else "r06_kernel_stats_stream_cfg%s.md" % cfg_name


def kernel_trace_averages(cfg_name):
    """{kernel name: (avg us, median us)} from the committed rocprofv3 kernel-trace summary of this configuration's default run (tools/rocpd_stats.py table), {} if absent"""
    out = {}
    try:
        with open(os.path.join(ROOT, "profiles", trace_file(cfg_name))) as fh:
            for ln in fh:
                f = [x.strip() for x in ln.strip().strip("|").split("|")]
                if len(f) >= 5 and f[1].isdigit():
                    out[f[0]] = (float(f[3]), float(f[4]))
    except (OSError, ValueError):
        return {}
    return out


def pmc_traffic(cfg_name, kernel_substr):
    """(bytes per launch, source) of a kernel from the committed rocprofv3 --pmc summary of THIS configuration, or (None, reason)."""
    path = None
    for rnd in ("r06", "r05", "r04", "r03"):      # (configurations whose PMC pass was not repeated this round keep an earlier round's: same kernels)
        path = os.path.join(ROOT, "profiles", "%s_pmc_traffic_cfg%s.json" % (rnd, cfg_name))
        if os.path.exists(path):
            break
    try:
        with open(path) as fh:
            e = [v for k, v in json.load(fh).items() if kernel_substr in k]
        e.sort(key=lambda v: -v.get("dispatches", 0))
        return 1024.0 * (e[0]["fetch_kb_mean"] + e[0]["write_kb_mean"]), "bytes/launch (profiles/%s)" % os.path.basename(path)
    except (OSError, IndexError, KeyError, ValueError):
        return None, "no committed PMC pass for cfg%s" % cfg_name


def update_at_load_leg(cfg, torch, name="B", reps=60):
    """The headline latency AT LOAD: Updater::update on ceil(F/2) features.  The free-running synthetic scene hands the
    update a dozen features per frame; here the window is filled by a direct-track sequence, then the update is driven with two
    full loads generated for that very state (rvio_amd.synth.worst_case_tracks): "half" = every second feature type '2' at the
    maximum length, the others type '1' with L ~ U[3, n+1]; "long" = every feature type '1' at L = n+1, the 2L-3 = 19-row worst
    case W_filter = 51 MFLOP of SURVEY.md 8d is quoted on.  Timed with HIP events on the handle's stream around
    rvio_hip_update_tracked (state re-seeded before every repetition); per-kernel times from rvio_hip_debug_time_kernel."""
    from rvio_amd import hip
    seq = rv.synth.SynthSequence(cfg, duration=(K0 + 40) / 20.0 + 1.0)
    h = hip.RvioHip(cfg)
    st = torch.cuda.ExternalStream(h.stream())
    h.initialize(*seq.init_from_static(K0))
    drv = rv.synth.DirectTrackDriver(seq)
    nfill = cfg.max_track_len + 8
    for f in range(nfill):
        inp = drv.inputs(K0 + 1 + f)
        h.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        drv.after(h.get_points()[0])
    imu = seq.imu_between(K0 + 1 + nfill)
    h.propagate(imu)
    x1, P1 = h.get_state()
    n = (len(x1) - 26) // 7
    c6 = 6 * n
    PEAK_F64 = 78.6
    res = {}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for mix in ("half", "long"):
        types, lens, meas = rv.synth.worst_case_tracks(cfg, x1, mix=mix)
        h.set_state(x1, P1)
        h.update(types, lens, meas)                     # uploads the hand-over tables (they stay on the device) + warm-up
        info = h.frame_info()
        ts = []
        for _ in range(reps):
            h.set_state(x1, P1)
            with torch.cuda.stream(st):
                ev0.record()
            h.update_tracked()
            with torch.cuda.stream(st):
                ev1.record()
            h.sync()
            ts.append(ev0.elapsed_time(ev1))
        h.set_state(x1, P1)
        h.update_tracked()
        h.sync()
        kern = {k: h.time_kernel(w, 20) for k, w in (("feat_build", 2), ("gram_reduce", 3), ("solve", 0), ("ug", 4), ("final", 5), ("ug_final_as_launched", 7))}
        w_alg = filter_flops(cfg, n, lens, types, 0)    # W_filter of SURVEY.md 8d for exactly these tracks (update only: m = 0)
        p50 = float(np.median(ts))
        res[mix] = {"n_feat": int(len(types)), "n_feat_accepted": int(info["n_feat_accepted"]), "n_rows": int(info["n_rows"]),
                    "p50_update_ms": p50, "p95_update_ms": float(np.percentile(ts, 95)), "kernel_us": kern,
                    "w_filter_mflop": w_alg / 1e6,
                    "roofline": {"bound": "mfma", "achieved": w_alg / (p50 * 1e-3) / 1e12, "peak": PEAK_F64, "unit": "TFLOP/s",
                                 "frac": w_alg / (p50 * 1e-3) / 1e12 / PEAK_F64,
                                 "note": "the reference's FP64 work for these tracks (gate, Givens nullspace + compression, EKF) over the p50 update time of ONE stream"}}
    h.close()
    return {"workload": "cfg%s, window full (%d clones, 6n = %d), one stream: rvio_hip_update_tracked on full loads" % (name, n, c6), **res}


def batched_at_load_leg(cfg, torch, sizes, name="B", reps=10):
    """The batched filter at SURVEY.md 8d's DEFINED direct-track load — exactly ceil(F/2) features per instance and frame: "half" = every
    second feature type '2' at the maximum length, the others type '1' with L ~ U[3, n+1]; "long" = every feature type '1' at L = n+1 (the
    worst case W_filter is quoted on) — beside `batched_filter`, whose load is the natural track flow of the synthetic scene (a dozen
    features per frame).  Every instance starts each repetition from the same full-window state (re-seeded, untimed); timed with HIP events
    on the handle's stream: ONE whole filter frame (propagate + update + augment / compose) of all B instances."""
    from rvio_amd import hip
    Fu, ML = abi.fu(cfg), cfg.max_track_len
    seq = rv.synth.SynthSequence(cfg, duration=(K0 + 40) / 20.0 + 1.0)
    h1 = hip.RvioHip(cfg)
    h1.initialize(*seq.init_from_static(K0))
    drv = rv.synth.DirectTrackDriver(seq)
    nfill = cfg.max_track_len + 8
    for f in range(nfill):
        inp = drv.inputs(K0 + 1 + f)
        h1.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
        drv.after(h1.get_points()[0])
    x0, P0 = h1.get_state()
    imu = seq.imu_between(K0 + 1 + nfill)
    h1.propagate(imu)
    x1, _ = h1.get_state()
    h1.close()
    n = (len(x1) - 26) // 7
    PEAK_F64 = 78.6
    d_imu = torch.from_numpy(np.ascontiguousarray(imu).view(np.uint8)).cuda()
    res = []
    for B in sizes:
        hb = hip.RvioHip(cfg, batch=B)
        st = torch.cuda.ExternalStream(hb.stream())
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        entry = {"instances": B}
        for mix in ("half", "long"):
            types, lens, meas = rv.synth.worst_case_tracks(cfg, x1, mix=mix)
            nf = len(types)
            t_nf = np.full(B, nf, np.int32)
            t_ty, t_ln, t_me = np.zeros((B, Fu), np.uint8), np.zeros((B, Fu), np.int32), np.zeros((B, Fu, ML, 2), np.float32)
            t_ty[:, :nf], t_ln[:, :nf] = types, lens
            t_me[:, :nf, : meas.shape[1]] = meas
            d = [torch.from_numpy(a_).cuda() for a_ in (t_nf, t_ty, t_ln, t_me)]
            torch.cuda.synchronize()
            ts, info = [], None
            for r in range(reps + 2):
                hb.set_state(x0, P0)
                with torch.cuda.stream(st):
                    ev0.record()
                hb.frame_tracks_dev(d_imu.data_ptr(), 0, len(imu), d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr())
                with torch.cuda.stream(st):
                    ev1.record()
                hb.sync()
                if r >= 2:
                    ts.append(ev0.elapsed_time(ev1))
            x_last = hb.get_state_at(B - 1)[0]
            w_alg = filter_flops(cfg, n, lens, types, len(imu))
            p50 = float(np.median(ts))
            tfl = w_alg * B / (p50 * 1e-3) / 1e12
            entry[mix] = {"n_feat": int(nf), "stacked_rows_if_all_accepted": int(np.sum(2 * np.where(types == ord("2"), (lens + 1) // 2, lens) - 3)),
                          "ms_per_batched_frame": p50, "filter_frames_per_s": B / (p50 * 1e-3), "w_filter_mflop_per_instance": w_alg / 1e6,
                          "achieved_tflops_fp64": tfl, "frac_fp64_peak": tfl / PEAK_F64, "finite": bool(np.all(np.isfinite(x_last)))}
            del d
        hb.close()
        res.append(entry)
    return {"workload": "cfg%s filter only, window full (%d clones): every instance runs ONE filter frame on ceil(F/2) = %d features (SURVEY.md 8d's defined "
                        "direct-track load), all instances from the same state, %d repetitions" % (name, n, Fu, reps),
            "peak_tflops_fp64": PEAK_F64, "sizes": res}


def multi_stream(cfg, fs, wi, ai, ni, n_warm, streams=8, threads=1):
    """Aggregate throughput of `streams` independent filter instances (own handle, own HIP streams) fed the same resident
    frames: kernels of different instances overlap on the 256 CUs.  `threads` host threads issue the launches (each drives
    streams/threads instances; the C-ABI calls release the GIL)."""
    import threading
    from rvio_amd import hip
    n_frames = fs.n
    n_warm = max(0, min(n_warm, n_frames - 1))
    hs = [hip.RvioHip(cfg) for _ in range(streams)]
    for h in hs:
        h.initialize(wi, ai, ni)

    def run(mine, lo, hi):
        for i in range(lo, hi):
            a = fs.args(i)
            for h in mine:
                h.frame_dev(*a)
        for h in mine:
            h.sync()

    def run_all(lo, hi):
        if threads <= 1:
            run(hs, lo, hi)
            return
        ts = [threading.Thread(target=run, args=(hs[t::threads], lo, hi)) for t in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    run_all(0, n_warm)
    t0 = time.perf_counter()
    run_all(n_warm, n_frames)
    el = time.perf_counter() - t0
    k = n_frames - n_warm
    for h in hs:
        h.close()
    return {"streams": streams, "host_threads": threads, "value": streams * k / el, "unit": "frames/s", "frames_per_stream": k,
            "note": "independent handles (3 streams each) issued from one host thread onto HIP's 4 hardware queues: superseded by batch handles (batched_streams)"}


def host_buffer_leg(cfg, fs, wi, ai, ni, n_warm):
    """PCIe-inclusive rate: the same frames handed over as HOST buffers (rvio_hip_frame: image + IMU + corners copied H2D on
    the tracker stream every frame).  Reported beside `value`, never as `value`."""
    from rvio_amd import hip
    h = hip.RvioHip(cfg)
    h.initialize(wi, ai, ni)
    n = fs.n
    n_warm = max(0, min(n_warm, n - 1))
    for i in range(n_warm):
        h.frame(*fs.host(i))
    h.sync()
    t0 = time.perf_counter()
    for i in range(n_warm, n):
        h.frame(*fs.host(i))
    h.sync()
    el = time.perf_counter() - t0
    h.close()
    return {"value": (n - n_warm) / el, "unit": "frames/s",
            "bytes_h2d_per_frame": int(fs.imgs[0].nbytes + fs.imu_arr[0].nbytes + (0 if fs.cand_arr is None else fs.cand_arr[0].nbytes)),
            "note": "caller's pageable buffers, packed into the library's pinned ring on the host, asynchronous H2D on the tracker stream"}


def filter_flops(cfg, n, lens, types, m):
    """W_filter of SURVEY.md 8d for one frame: the reference's FP64 work (gate, nullspace, Givens compression, EKF, propagate) for
    the tracks actually handed over (rho_f = 2 L_f - 3 rows per feature, type '2' contributes its first ceil(L/2) observations)"""
    c6, d = 6 * n, 24 + 6 * n
    w = m * 6.0 * 24 ** 3
    if n <= cfg.min_track_len - 1:
        return w
    M = 0
    for L, t in zip(lens, types):
        Le = (int(L) + 1) // 2 if t == ord("2") else int(L)
        rho = max(2 * Le - 3, 0)
        M += rho
        w += 2.0 * rho * c6 ** 2 + 2.0 * rho ** 2 * c6 + (4.0 / 3.0) * rho ** 3
    r = min(M, c6)
    w += 3.0 * M * c6 ** 2
    w += 2.0 * r * d * d + 2.0 * r * r * d + 2.0 * d * d * r + 2.0 * r ** 3 + 2.0 * d * r * r + 2.0 * d * d * r + 4.0 * d ** 3 + 2.0 * d * d * r
    return w


def parse_batch_size(size):
    """an entry of --batch: "2048" = one batch handle of 2048 instances; "2048x2" = 2048 instances as two handles of 1024, both in flight"""
    txt = str(size)
    B, nh = (int(v) for v in txt.split("x")) if "x" in txt else (int(txt), 1)
    if B < 1 or nh < 1 or B % nh:
        raise ValueError("--batch entry %r: instances must be a positive multiple of the handle count" % (size,))
    return B, nh


def batched_filter_leg(cfg, torch, sizes, name="B", seeds=4, n_warm=16, n_timed=40, seed0=0, barrier=None):
    """B independent filter instances advanced by ONE launch per stage (rvio_hip_create_batch).  The hand-over
    tables come from `seeds` direct-track sequences (different landmark/noise seeds) run through plain handles first; instance b
    replays sequence b mod seeds.  Reports filter-frames/s and the FP64 rate against W_filter of the tracks actually processed."""
    from rvio_amd import hip
    Fu, ML = abi.fu(cfg), cfg.max_track_len
    nf = n_warm + n_timed
    k0 = K0
    tabs, inits, ends, flops = [], [], [], np.zeros(nf)
    for sd in range(seed0, seed0 + seeds):
        seq = rv.synth.SynthSequence(cfg, duration=(k0 + nf + 3) / 20.0 + 1.0, seed=sd)
        wi, ai, ni = seq.init_from_static(k0)
        h = hip.RvioHip(cfg)
        h.initialize(wi, ai, ni)
        inits.append(h.get_state())
        drv = rv.synth.DirectTrackDriver(seq)
        n_feat = np.zeros(nf, np.int32)
        types = np.zeros((nf, Fu), np.uint8)
        lens = np.zeros((nf, Fu), np.int32)
        meas = np.zeros((nf, Fu, ML, 2), np.float32)
        imus = []
        for f in range(nf):
            inp = drv.inputs(k0 + 1 + f)
            ncl = min(max(f - 1, 0), ML - 1)           # window length when this frame's update runs (first augmentation: 2nd frame)
            h.frame_points(inp["tracked"], inp["status"], inp["imu"], inp["cand"])
            t, l, me = h.get_tracks()
            drv.after(h.get_points()[0])
            n_feat[f] = len(l)
            types[f, : len(l)], lens[f, : len(l)], meas[f, : len(l)] = t, l, me
            imus.append(inp["imu"])
            flops[f] += filter_flops(cfg, ncl, l, t, len(inp["imu"])) / seeds
        x_end = h.get_state()[0]           # the plain handle after the same nf frames: what every instance replaying this sequence must end at
        h.close()
        m = min(len(i) for i in imus)
        ends.append((x_end, all(len(i) == m for i in imus)))
        tabs.append((n_feat, types, lens, meas, np.stack([i[:m] for i in imus]), m))
    m = min(t[5] for t in tabs)
    w_frame = float(np.mean(flops[n_warm:]))
    PEAK_F64 = 78.6
    res = []
    for size in sizes:
        B, nh = parse_batch_size(size)
        idx = np.arange(B) % seeds
        # [frame][instance] tables, resident in HBM before the timed region
        d_nf = torch.from_numpy(np.stack([tabs[i][0] for i in idx], 1).copy()).cuda()
        d_ty = torch.from_numpy(np.stack([tabs[i][1] for i in idx], 1).copy()).cuda()
        d_ln = torch.from_numpy(np.stack([tabs[i][2] for i in idx], 1).copy()).cuda()
        d_me = torch.from_numpy(np.stack([tabs[i][3] for i in idx], 1).copy()).cuda()
        imu_h = np.stack([tabs[i][4][:, :m] for i in idx], 1).copy()            # [frame][instance][m]
        d_im = torch.from_numpy(imu_h.view(np.uint8).reshape(nf, B, -1)).cuda()
        per = B // nh
        hs = [hip.RvioHip(cfg, batch=per) for _ in range(nh)]
        for k, h in enumerate(hs):
            h.set_state(*inits[idx[k * per]])
            for b in range(per):
                if idx[k * per + b] != idx[k * per]:
                    h.set_state_at(b, *inits[idx[k * per + b]])
        torch.cuda.synchronize()

        def frame(f):       # the calls only enqueue: with several handles their chains run side by side on the device
            for k, h in enumerate(hs):
                a = k * per
                h.frame_tracks_dev(d_im[f, a].data_ptr(), m, m, d_nf[f, a:].data_ptr(), d_ty[f, a].data_ptr(), d_ln[f, a].data_ptr(), d_me[f, a].data_ptr())

        def sync():
            for h in hs:
                h.sync()
        for f in range(n_warm):
            frame(f)
        sync()
        if barrier is not None:
            barrier()
        t0 = time.perf_counter()
        for f in range(n_warm, nf):
            frame(f)
        sync()
        if barrier is not None:
            barrier()
        el = time.perf_counter() - t0
        x_last = hs[-1].get_state_at(per - 1)[0]
        x_first = hs[0].get_state_at(0)[0]
        same_m = all(ends[i][1] and tabs[i][5] == m for i in (idx[0], idx[B - 1]))   # (the plain handle integrated every IMU sample of its frames; the batch the common first m)
        d_plain = max(float(np.max(np.abs(_qfix(x_first) - _qfix(ends[idx[0]][0])))), float(np.max(np.abs(_qfix(x_last) - _qfix(ends[idx[B - 1]][0])))))
        for h in hs:
            h.close()
        del d_nf, d_ty, d_ln, d_me, d_im
        tfl = w_frame * B * n_timed / el / 1e12
        res.append({"instances": B, "handles": nh, "ms_per_batched_frame": 1e3 * el / n_timed, "filter_frames_per_s": B * n_timed / el,
                    "achieved_tflops_fp64": tfl, "frac_fp64_peak": tfl / PEAK_F64, "finite": bool(np.all(np.isfinite(x_last))),
                    "max_state_delta_vs_plain_handle": d_plain if same_m else None,
                    "plain_handle_note": "instances 0 and B-1 after the timed run against the plain (one-instance, latency-form) handles that ran the same %d frames" % nf})
    return {"workload": "cfg%s filter only (propagate + update + augment/compose), direct-track hand-over tables of %d seeded sequences, "
                        "%d frames timed after %d, one launch per stage for all instances" % (name, seeds, n_timed, n_warm),
            "algorithmic_mflop_per_filter_frame": w_frame / 1e6, "peak_tflops_fp64": PEAK_F64, "sizes": res}


def batched_streams_leg(cfg, torch, sizes, name="B", seeds=4, n_warm=12, n_timed=30):
    """The WHOLE frame (CLAHE, detector, KLT, RANSAC, book-keeping, propagate, update, augment/compose) for B camera streams in one
    launch per stage (rvio_hip_create_batch with front end, rvio_hip_frame_batch_dev).  `seeds` differently seeded scenes are
    rendered on the host; stream b replays scene b mod seeds.  Images and IMU batches are resident in HBM before the timed region."""
    from rvio_amd import hip
    nf = 1 + n_warm + n_timed
    scenes = [build_inputs(cfg, nf, seed=sd) for sd in range(seeds)]
    m = int(min(sc[3].min() for sc in scenes))
    inits = []
    for sc in scenes:
        h = hip.RvioHip(cfg)
        h.initialize(*sc[0].init_from_static(K0))
        inits.append(h.get_state())
        h.close()
    d_img_s = [torch.from_numpy(sc[1]).cuda() for sc in scenes]                                   # [seed][frame][H][W]
    d_imu_s = [torch.from_numpy(np.ascontiguousarray(sc[2][:, :m]).view(np.uint8).reshape(nf, -1)).cuda() for sc in scenes]
    npx = cfg.width * cfg.height
    it_l = 10
    by_klt = npx * (1 + 2 * (1 / 4 + 1 / 16 + 1 / 64)) + cfg.n_features * 4 * (16 * 16 * 5) + cfg.n_features * 4 * it_l * 16 * 16   # B_klt, SURVEY.md 8d
    res = []
    for size in sizes:
        B, nh = parse_batch_size(size)
        idx = np.arange(B) % seeds
        h = hip.RvioHip(cfg, batch=B, front_end=True)
        h.set_state(*inits[0])
        for b in range(B):
            if idx[b] != 0:
                h.set_state_at(b, *inits[idx[b]])
        img_buf = torch.empty((2, B, cfg.height, cfg.width), dtype=torch.uint8, device="cuda")      # double-buffered frame of B streams
        imu_buf = torch.empty((2, B, d_imu_s[0].shape[1]), dtype=torch.uint8, device="cuda")
        sel = torch.from_numpy(idx).cuda()
        stack_img = torch.stack(d_img_s)                                                            # [seed][frame][H][W]
        stack_imu = torch.stack(d_imu_s)

        def stage(f):          # gather this frame's B images / IMU batches (device-side copy, OUTSIDE the timed region: see below)
            img_buf[f & 1].copy_(stack_img[sel, f])
            imu_buf[f & 1].copy_(stack_imu[sel, f])
        # all frames are staged up front so that the timed region contains the library's work only
        frames_img = torch.empty((nf, B, cfg.height, cfg.width), dtype=torch.uint8, device="cuda") if B * nf * npx < 24e9 else None
        if frames_img is not None:
            frames_imu = torch.empty((nf, B, d_imu_s[0].shape[1]), dtype=torch.uint8, device="cuda")
            for f in range(nf):
                frames_img[f].copy_(stack_img[sel, f])
                frames_imu[f].copy_(stack_imu[sel, f])
        torch.cuda.synchronize()

        def frame(f):
            if frames_img is None:
                stage(f)
                torch.cuda.synchronize()
                ip, up_ = img_buf[f & 1].data_ptr(), imu_buf[f & 1].data_ptr()
            else:
                ip, up_ = frames_img[f].data_ptr(), frames_imu[f].data_ptr()
            h.frame_batch_dev(ip, cfg.width, npx, up_, m, m)
        for f in range(1 + n_warm):
            frame(f)
        h.sync()
        t0 = time.perf_counter()
        for f in range(1 + n_warm, nf):
            frame(f)
        h.sync()
        el = time.perf_counter() - t0
        x_last = h.get_state_at(B - 1)[0]
        info = h.frame_info()
        h.close()
        del frames_img, img_buf, imu_buf, stack_img, stack_imu
        fps = B * n_timed / el
        res.append({"streams": B, "ms_per_batched_frame": 1e3 * el / n_timed, "frames_per_s": fps,
                    "klt_chain_algorithmic_GBps": by_klt * fps / 1e9, "frac_hbm_peak": by_klt * fps / 1e9 / 8000.0,
                    "staged_up_front": True, "finite": bool(np.all(np.isfinite(x_last))), "updated_last_frame": int(info["updated"])})
    for e in res:
        e.update(issue_slots(e["streams"], e["ms_per_batched_frame"]))
    return {"workload": "cfg%s whole frame (stock: CLAHE + device detector), %d seeded scenes, stream b replays scene b mod %d, %d frames timed after %d, "
                        "one launch per stage for all streams" % (name, seeds, seeds, n_timed, 1 + n_warm),
            "algorithmic_MB_per_frame_klt_chain": by_klt / 1e6, "sizes": res}


SHADER_HZ, N_SIMD = 2.4e9, 256 * 4     # MI355X: peak engine clock, 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)


def issue_slots(streams, ms_per_frame, path=None):
    """The bound the batched camera-stream frame actually hits (not HBM: ~6 %): instruction issue.  The SQ counters of one batched frame
    (committed rocprofv3 --pmc pass, tools/issue_slots_json.py — counters cannot be read live) over the issue slots of the frame time
    measured live: one slot = one wave instruction on one SIMD = 4 cycles."""
    path = path or os.path.join(ROOT, "profiles", "r05_streams_issue_slots.json")
    if not os.path.exists(path):
        return {}
    try:
        with open(path) as f:
            cm = json.load(f)
        if cm.get("streams") != streams:
            return {}
        slots = ms_per_frame * 1e-3 * SHADER_HZ * N_SIMD / 4.0
        pf = cm["per_batched_frame"]
        o = {"bound": "instruction issue", "slots_per_batched_frame": slots, "unit": "SIMD issue slots (4 cycles each, 1024 SIMDs at 2.4 GHz)",
             "what": "SQ counters of one batched frame (committed pass: profiles/r05_streams_issue_slots.json) / the issue slots of the frame time measured here; "
                     "VALU is a utilisation (<= 1), ANY counts scalar / LDS / memory instructions of other waves issuing beside it (can exceed 1)"}
        if "SQ_ACTIVE_INST_VALU" in pf:
            o["valu_busy"] = pf["SQ_ACTIVE_INST_VALU"]; o["frac_valu"] = pf["SQ_ACTIVE_INST_VALU"] / slots
        if "SQ_ACTIVE_INST_ANY" in pf:
            o["any_busy"] = pf["SQ_ACTIVE_INST_ANY"]; o["frac_any"] = pf["SQ_ACTIVE_INST_ANY"] / slots
        return {"issue_slots": o}
    except Exception as e:   # noqa: BLE001
        return {"issue_slots": {"error": repr(e)[:100]}}


_CFG_NAME = ["B"]
_MULTI = [None]


def parity_leg(cfg, torch, pin, wi, ai, ni, n, cpu_states):
    """The device, free-running over the same n frames as the CPU oracle in its LITERAL form (sequential Givens QR + leading-row rank
    scan, Updater.cc:469-529): per-state and covariance deltas at EVERY frame, the update counters, and the frames in which the
    reference's scan cut informative rows off (the device must report the same nRank there)."""
    from rvio_amd import hip
    imgs, imu_arr, imu_cnt, cand_arr, cand_cnt = pin
    h = hip.RvioHip(cfg)
    h.initialize(wi, ai, ni)
    worst_x, worst_p, cuts, cuts_equal, counters_equal = 0.0, 0.0, [], True, True
    for i in range(n):
        h.frame(imgs[i], imu_arr[i, : imu_cnt[i]], None if cand_arr is None else cand_arr[i, : cand_cnt[i]])
        x, P = h.get_state()
        gi = h.frame_info()
        xc, Pc, ci, crank = cpu_states[i]
        worst_x = max(worst_x, float(np.max(np.abs(_qfix(x) - _qfix(xc)))))
        worst_p = max(worst_p, float(np.max(np.abs(P - Pc)) / max(1e-300, float(np.max(np.abs(Pc))))))
        counters_equal &= all(gi[k] == ci[k] for k in ("n_klt_ok", "n_ransac_inliers", "n_feat_update", "n_feat_accepted", "n_rows", "updated"))
        if gi["rank_truncated_at"] >= 0:
            cuts.append(i)
            cuts_equal &= gi["rank_truncated_at"] == crank
    h.close()
    return {"frames": n, "max_state_delta": worst_x, "max_cov_rel_delta": worst_p, "tolerance": 1e-6, "counters_equal_every_frame": bool(counters_equal),
            "rank_truncation_frames": cuts, "rank_truncation_nrank_equal": bool(cuts_equal),
            "against": "the CPU baseline run in its literal form (Givens QR + leading-row rank scan, Updater.cc:469-529); max over all frames of the free-running sequence"}


def cpu_baseline(cfg, seq, imgs, imu_arr, imu_cnt, cand_arr, cand_cnt, wi, ai, ni, n):
    """The CPU oracle (oracle/liborc.so, -O3, 1 core) on the first n frames of the same sequence; also returns (x, P, info, nRank) after every frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    s = O.System(cfg)
    x0, P0 = O.initialize(cfg, wi, ai, ni)
    s.set_state(x0, P0)
    tms, states, per = [], [], []
    for i in range(n):
        t0 = time.perf_counter()
        info, t, pp, pq = s.frame(imu_arr[i, : imu_cnt[i]], None if cand_arr is None else cand_arr[i, : cand_cnt[i]], img=imgs[i])
        per.append(time.perf_counter() - t0)
        tms.append(t)
        xs, Ps = s.get_state()
        states.append((xs, Ps, info, s.last_rank()))
    warm = CPU_WARM if n >= CPU_WARM + 100 else min(20, n // 4)   # 50 warm-up frames (short samples: a quarter of them)
    per = np.array(per)[warm:]
    el = float(per.sum())
    tms = np.array(tms)[warm:]
    xs_lit = states[-1][0]
    xs = states
    multi = None
    if cand_arr is None:   # secondary figure: the same sources with their OpenMP loops active, all host cores, in a child process
        import subprocess
        import tempfile
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = max(1, min(8, cores))     # SURVEY.md 8d brackets OpenCV's threading with 8 host cores; more threads only add fork/join cost here
        try:
            with tempfile.TemporaryDirectory() as td:
                f = os.path.join(td, "in.npz")
                np.savez(f, config=_CFG_NAME[0], equalizer=int(cfg.enable_equalizer), imgs=imgs[:n], imu=imu_arr[:n].view(np.uint8), imu_cnt=imu_cnt[:n],
                         wi=np.asarray(wi, float), ai=np.asarray(ai, float), ni=int(ni), warm=int(warm))
                env = dict(os.environ, ORC_LIB="liborc_omp.so", OMP_NUM_THREADS=str(cores), OMP_WAIT_POLICY="passive")
                r = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_omp.py"), f], env=env, timeout=120).decode().strip().splitlines()[-1])
            multi = {"value": r["value"], "unit": "frames/s", "cores": cores, "kind": "port",
                     "frame_ms_p50": r.get("frame_ms_p50"), "frame_ms_p95": r.get("frame_ms_p95"),
                     "sample": "the same frames after the same %d warm-up frames, oracle/liborc_omp.so (image rows, CLAHE tiles, KLT features and cornerSubPix corners in parallel)" % warm,
                     "same_state_as_single_thread": bool(np.array_equal(np.array(r["x"]), xs_lit))}
        except Exception as e:   # the baseline is a reported extra: never fail the bench line over it
            multi = {"error": repr(e)[:200]}
    _MULTI[0] = multi
    return ({"value": len(per) / el, "unit": "frames/s", "cores": 1, "kind": "port",
             "frames_timed": int(len(per)), "frames_warmup": int(warm),
             "frame_ms_p50": float(1e3 * np.median(per)), "frame_ms_p95": float(1e3 * np.percentile(per, 95)),
             "stage_ms_p50": dict(zip(("track", "propagate", "update", "augment_compose"), (float(v) for v in np.median(tms, axis=0)))),
             "stage_ms_p95": dict(zip(("track", "propagate", "update", "augment_compose"), (float(v) for v in np.percentile(tms, 95, axis=0)))),
             "sample": "frames %d..%d of the same synthetic sequence after %d warm-up frames (BASELINE.md section 3), oracle/liborc.so = the CPU RESTATEMENT of the "
                       "reference (kind: port — not Eigen / OpenCV; pinned against the reference's own sources by tests/test_ref_pins.py), g++ -O3, single thread; "
                       "the spans of System.cc:255-260,367 (track; propagate + update + augment + compose), no ROS publishing / debug images / usleep"
                       % (warm, n - 1, warm)}, xs)


def cpu_baseline_reference(cfg, imgs, imu_arr, imu_cnt, cand_arr, wi, ai, ni, n, port_end_state=None, timeout=150):
    """The reference's OWN translation units (oracle/_ref/libref.so: Updater.cc, PreIntegrator.cc, Ransac.cc, Tracker.cc, FeatureDetector.cc,
    InputBuffer.cc, System.cc compiled unmodified against oracle/refshim/) driven frame by frame through System::MonoVIO on the same frames
    as cpu_baseline — in a child process with a time-out (tools/cpu_baseline_ref.py: the reference's code can spin forever with 17..31 RANSAC
    candidates).  libref.so is built where /root/reference exists and travels with the repository snapshot; where it is
    absent the leg says so.  Test infrastructure, like the oracle: timed here as a baseline only."""
    import subprocess
    import tempfile
    if cand_arr is not None:
        return {"skipped": "host-corner mode: the reference leg replays the stock configuration (its own FeatureDetector)"}
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref.so")) and not os.path.isdir("/root/reference/src/rvio"):
        return {"skipped": "oracle/_ref/libref.so is not here (it is built by `make -C oracle ref` where the reference's sources exist)"}
    warm = CPU_WARM if n >= CPU_WARM + 100 else min(20, n // 4)
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "in.npz")
        np.savez(f, config=_CFG_NAME[0], equalizer=int(cfg.enable_equalizer), imgs=imgs[:n], imu=imu_arr[:n].view(np.uint8), imu_cnt=imu_cnt[:n],
                 wi=np.asarray(wi, float), ai=np.asarray(ai, float), ni=int(ni), warm=int(warm))
        r = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_ref.py"), f], timeout=timeout,
                                               stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1])
    if "skipped" in r:
        return r
    o = {"value": r["value"], "unit": "frames/s", "cores": 1, "kind": "reference", "frames_timed": r["frames_timed"], "frames_warmup": int(warm),
         "frame_ms_p50": r["frame_ms_p50"], "frame_ms_p95": r["frame_ms_p95"],
         "sample": "the same frames as cpu_baseline through System::MonoVIO of the reference's own sources, g++ -O2, single thread, on oracle/refshim: its Eigen is "
                   "a header of plain loops (no SIMD kernels: the filter stages run slower than on real Eigen) and its OpenCV IMAGE algorithms forward to the port's "
                   "restatements (the track span is the port's) - so this is the reference's CODE on the host cores, not the Eigen / OpenCV build BASELINE.md names"}
    if port_end_state is not None:
        xr = np.array(r["x"])
        o["max_state_delta_vs_port_after_these_frames"] = float(np.max(np.abs(_qfix(xr) - _qfix(port_end_state)))) if len(xr) == len(port_end_state) else None
    return o


if __name__ == "__main__":
    main()
