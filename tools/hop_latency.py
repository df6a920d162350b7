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
for G in (2, 4, 8, 16, 32, 64, 128):
    for rm in (1, 8, -8, -12):
        if rm == 8 and G < 16:
            continue
        ns = eng.debug_exchange_cost(G, 37, rm, 5000)
        print(json.dumps(dict(exchange_workgroups=G, slots_read=37,
                              sweepers="all, 4 waves" if rm == 1 else ("all, 8 store-free waves x 1 round (12 waves)" if rm == -12 else "all, 8 waves x 1 round" if rm < 0
                                       else "1 in %d + follower hop" % rm),
                              ns_per_round=round(ns, 1))), flush=True)
eng.close()
