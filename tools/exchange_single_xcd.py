#!/usr/bin/env python3
"""Bare exchange cost: all XCDs (sc1 stores) vs one XCD (plain stores, L2-served sc1 loads)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd.engine import HipEngine
eng = HipEngine(0, "f32")
for xcd in (0, 1, 3):
    eng.set_option("probe_xcd", xcd)
    for G in (8, 16, 32, 64):
        for lds in (60 * 1024,):
            eng.set_option("probe_lds", lds)
            try:
                ns = eng.debug_exchange_cost(G, 37, -8, 5000)
            except Exception as e:
                ns = str(e)
            print(json.dumps(dict(xcd_mode=xcd, G=G, ns_per_round=ns)), flush=True)
eng.close()
