#!/usr/bin/env python3
"""profiles/<round>_traffic.json (default round r04) from the rocprofv3 --pmc summaries of
tools/profile_<round>.sh:
per config and dominant kernel, FETCH_SIZE / WRITE_SIZE per launch in KB as reported (1 KB =
1024 B; calibration of the counters: profiles/r01_traffic.json "_comment_v2" and
profiles/r03_fetch_calibration.txt).  `engine_tag` is the library's build tag (hash of its
sources, spfm_build_tag) read from the very library the counters were collected with;
bench.py quotes the file for `roofline.traffic` only when the library it runs has that tag.

    python tools/make_traffic_json.py [tag] [round]   (on the GPU box, after tools/profile_r04.sh)
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


sys.path.insert(0, ROOT)
from sparsepoly_amd import _capi  # noqa: E402

ROUND = sys.argv[2] if len(sys.argv) > 2 else "r04"
out = {"engine_tag": sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else _capi.build_tag(),
       "_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes "
                   "(tools/profile_%s.sh), bench.py --steps 1 --warmup 0; per-launch means in KB " % ROUND +
                   "as reported; sources: profiles/%s_c<config>_pmc_{fetch,write}_summary.txt" % ROUND}
for cfg, (name, prefix) in KERNELS.items():
    f = os.path.join(PROF, "%s_c%d_pmc_fetch_summary.txt" % (ROUND, cfg))
    w = os.path.join(PROF, "%s_c%d_pmc_write_summary.txt" % (ROUND, cfg))
    if not (os.path.exists(f) and os.path.exists(w)):
        continue
    a, b = per_launch(f, prefix), per_launch(w, prefix)
    if a and b:
        out["config%d" % cfg] = {name: {"fetch_kb_per_launch": a[1], "write_kb_per_launch": b[1],
                                        "launches": a[0]}}
json.dump(out, open(os.path.join(PROF, "%s_traffic.json" % ROUND), "w"), indent=1)
print(json.dumps(out, indent=1))
