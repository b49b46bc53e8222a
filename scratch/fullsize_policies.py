#!/usr/bin/env python3
"""Full-size parity table over Winograd policies (oracle/fullsize.py::run_harness) -> stdout / gpurun_out/fullsize_policies.txt.
    python scratch/fullsize_policies.py [c2|c3 ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tf-faster-rcnn_amd"), os.path.join(ROOT, "tf-faster-rcnn_amd", "lib")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import fullsize as fs  # noqa: E402

configs = sys.argv[1:] or ["c2"]
dev = torch.device("cuda", 0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "fullsize_policies.txt"), "a") as f:
    for config in configs:
        for weights in (("damped", "calibrated") if config == "c2" else ("calibrated",)):
            for policy in os.environ.get("POLICIES", "direct,f2,f4,f4_b1_f2,f4_b12_f2,f4_b12_direct,f4_head_f2").split(","):
                try:
                    line = fs.format_report(fs.run_harness(config, weights, policy, dev))
                except Exception as e:  # noqa: BLE001
                    line = "%s %s %s EXCEPTION %r" % (config, weights, policy, e)
                print(line, flush=True)
                f.write(line + "\n")
                f.flush()
