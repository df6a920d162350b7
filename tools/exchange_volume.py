#!/usr/bin/env python3
"""Bare exchange round of the 64-column persistent pass (spfm_debug_exchange_cost) against the
number of slots read: how much of a round is volume, how much is latency."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd.engine import HipEngine  # noqa: E402

eng = HipEngine(0, "f32")
for G in (32, 64, 96, 128):
    for ncols in (4, 9, 18, 37, 64):
        ns = eng.debug_exchange_cost(G, ncols, -8, 5000)
        print(json.dumps(dict(exchange_workgroups=G, slots_read=ncols, sweepers="8 waves",
                              ns_per_round=round(ns, 1))), flush=True)
eng.close()
