#!/usr/bin/env python3
"""FP64 MFMA work of one batched filter frame from a rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 pass (its own pass, no trace domain).
usage: tools/mfma_util_json.py counter_collection.csv --instances 2048 > profiles/r05_batched_mfma_util.json
One MOPS_F64 count = 512 flop (v_mfma_f64_16x16x4: 16 x 16 x 4 x 2 = 2048 flop per wave instruction, the counter ticks 4 per instruction).
bench.py's roofline_batched divides the committed per-frame figure by the frame time it measures live."""
import csv
import json
import re
import sys

args = sys.argv[1:]
inst = 2048
if "--instances" in args:
    i = args.index("--instances"); inst = int(args[i + 1]); del args[i:i + 2]
agg = {}
for path in args:
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != "SQ_INSTS_VALU_MFMA_MOPS_F64" or int(r["Grid_Size"]) < inst * int(r["Workgroup_Size"]):
                continue
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            agg.setdefault(k, []).append(float(r["Counter_Value"]))
frames = max((len(v) for k, v in agg.items() if "joseph" in k), default=max((len(v) for v in agg.values()), default=1))
per_kernel = {k: {"dispatches": len(v), "mfma_flop_per_dispatch": 512.0 * sum(v) / len(v)} for k, v in agg.items() if sum(v) > 0}
total = sum(512.0 * sum(v) for v in agg.values()) / frames
print(json.dumps({"instances": inst, "frames_profiled": frames, "mfma_flop_per_batched_frame": total, "per_kernel": per_kernel,
                  "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -- python bench.py --no-streams --no-cpu --no-latency "
                            "--steps 5 --warmup 2 --batch %d --no-defined-load --batch-streams '' (tools/measure_all.sh)" % inst}))
