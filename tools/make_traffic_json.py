#!/usr/bin/env python3
"""profiles/r02_traffic.json from the rocprofv3 --pmc summaries of tools/profile_r02.sh:
per config and dominant kernel, FETCH_SIZE / WRITE_SIZE per launch in KB as reported (1 KB =
1024 B; calibration of the counters: profiles/r01_traffic.json "_comment_v2").  bench.py uses
the file for `roofline.traffic` only when `engine_tag` equals its own ENGINE_TAG.

    python tools/make_traffic_json.py <engine_tag>
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
# the kernel bench.py prices per config: (name in the JSON, prefix of the profiled symbol)
KERNELS = {2: ("pcd_prb_kernel", "void spfm::pcd_prb_kernel<float, 2,"),
           3: ("pcd_prb_kernel", "void spfm::pcd_prb_kernel<float, 3,"),
           4: ("pbcd_prb_kernel", "void spfm::pbcd_prb_kernel<float, 2,")}


def per_launch(path, prefix):
    for line in open(path):
        if line.startswith(prefix):
            m = re.search(r"calls\s+(\d+)\s+sum\s+(\S+)\s+mean/launch\s+(\S+)", line)
            return int(m.group(1)), float(m.group(3))
    return None


out = {"engine_tag": sys.argv[1],
       "_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes "
                   "(tools/profile_r02.sh), bench.py --steps 1 --warmup 0; per-launch means in KB "
                   "as reported; sources: profiles/r02_c<config>_pmc_{fetch,write}_summary.txt"}
for cfg, (name, prefix) in KERNELS.items():
    f = os.path.join(PROF, "r02_c%d_pmc_fetch_summary.txt" % cfg)
    w = os.path.join(PROF, "r02_c%d_pmc_write_summary.txt" % cfg)
    if not (os.path.exists(f) and os.path.exists(w)):
        continue
    a, b = per_launch(f, prefix), per_launch(w, prefix)
    if a and b:
        out["config%d" % cfg] = {name: {"fetch_kb_per_launch": a[1], "write_kb_per_launch": b[1],
                                        "launches": a[0]}}
json.dump(out, open(os.path.join(PROF, "r02_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
