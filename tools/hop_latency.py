#!/usr/bin/env python3
"""Measured latency of the exchange primitive of the persistent pass (agent-scope store ->
visible to another workgroup's agent-scope load), ping-pong between two workgroups of one
launch; see spfm_debug_hop_latency in include/spfm.h."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(0, "f32")
for partner in (1, 2, 4, 7, 8, 16, 9, 63):
    ns, xcc = eng.debug_hop_latency(partner, 20000)
    print(json.dumps(dict(partner_workgroup=partner, xcc_ids=list(xcc),
                          same_xcd=xcc[0] == xcc[1], ns_per_hop=round(ns, 1))), flush=True)
eng.close()
