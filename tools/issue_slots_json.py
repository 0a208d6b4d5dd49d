#!/usr/bin/env python3
"""Instruction-issue work of one batched camera-stream frame from a rocprofv3 --pmc pass over SQ counters (its own pass, no trace domain).
usage: tools/issue_slots_json.py counter_collection.csv --streams 128 > profiles/r05_streams_issue_slots.json

The batched camera-stream frame is not bandwidth-bound (6 % of HBM): its bound is the shader's instruction issue.  Per kernel and per
batched frame this sums SQ_ACTIVE_INST_ANY / SQ_ACTIVE_INST_VALU (quad-cycles in which a wave had an instruction — any / a vector-ALU one —
in execution, summed over the chip's waves) over every launch that covers all `--streams` streams (>= that many workgroups; torch's own
kernels of the bench harness excluded).  bench.py's batched_streams leg divides the committed per-frame figures by the issue slots of the
frame time it measures live: slots = seconds x 2.4e9 Hz x 256 CUs x 4 SIMDs / 4 cycles.  VALU busy / slots is a true utilisation (one
vector instruction per SIMD at a time); ANY / slots can exceed 1 (scalar, LDS and memory instructions of OTHER waves issue beside it)."""
import csv
import json
import re
import sys


def collect(paths, streams):
    agg = {}
    for path in paths:
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if int(r["Grid_Size"]) < streams * int(r["Workgroup_Size"]):
                    continue
                k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
                if k.startswith("at::") or k.startswith("__amd_rocclr"):
                    continue
                agg.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return agg


def summarise(agg, streams, source=""):
    per_kernel, n_by = {}, {}
    for k, d in agg.items():
        n = max(len(v) for v in d.values())
        n_by[k] = n
        per_kernel[k] = dict({"dispatches": n}, **{c: sum(v) / len(v) for c, v in d.items()})
    # frames = launches of a once-per-frame kernel (the pyramid: one launch per batched frame)
    frames = n_by.get("pyramid_kernel") or (max(n_by.values()) if n_by else 1)
    tot = {}
    for k, d in agg.items():
        for c, v in d.items():
            tot[c] = tot.get(c, 0.0) + sum(v) / frames
    return {"streams": streams, "frames_profiled": frames, "per_batched_frame": tot, "per_kernel": per_kernel, "source": source}


if __name__ == "__main__":
    args = sys.argv[1:]
    streams = 128
    if "--streams" in args:
        i = args.index("--streams"); streams = int(args[i + 1]); del args[i:i + 2]
    print(json.dumps(summarise(collect(args, streams), streams,
                               "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -- python bench.py "
                               "--no-cpu --batch '' --no-streams --no-latency --batch-streams %d --steps 20 --warmup 5 (tools/final_measure.sh)" % streams)))
