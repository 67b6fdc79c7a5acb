#!/usr/bin/env python3
"""gpurun_out/n3mr_prof (tools/collect_profiles_n3mr.sh) -> profiles/traffic_n3mr_latest.json: HBM-side bytes per launch of the
NMR forward (k_n3mr_zbuffer + k_n3mr_resolve) and backward (k_n3mr_pack + k_n3mr_backward_pixel_map* + k_n3mr_backward_line_walks + k_n3mr_backward_face),
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 as in tools/pmc_to_json.py, stamped with bench.csrc_hash().  usage: tools/pmc_n3mr_to_json.py <dir> <tag>"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import csrc_hash                 # noqa: E402

d, tag = sys.argv[1], sys.argv[2]
FWD, BWD = ("k_n3mr_zbuffer", "k_n3mr_resolve"), ("k_n3mr_pack", "k_n3mr_backward_pixel_map", "k_n3mr_backward_line_walks", "k_n3mr_backward_face")


def means(counter):
    for root, _, files in os.walk(os.path.join(d, counter)):
        for f in files:
            if f.endswith("counter_collection.csv"):
                acc = collections.defaultdict(list)
                for r in csv.DictReader(open(os.path.join(root, f))):
                    if r["Counter_Name"] == counter:
                        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
                return {k: sum(v) / len(v) for k, v in acc.items()}
    return {}


fetch, write = means("FETCH_SIZE"), means("WRITE_SIZE")
tot = lambda m, names: sum(v for k, v in m.items() if any(n in k for n in names))
out = {"fwd": (2 * tot(fetch, FWD) + tot(write, FWD)) * 1024, "bwd": (2 * tot(fetch, BWD) + tot(write, BWD)) * 1024,
       "per_kernel_fetch_kb": {k.split("(")[0][-40:]: v for k, v in fetch.items() if "n3mr" in k},
       "csrc_hash": csrc_hash(),
       "note": "HBM-side bytes per launch of the NMR forward (k_n3mr_zbuffer + k_n3mr_resolve) and backward (k_n3mr_pack + "
               "k_n3mr_backward_pixel_map_all + k_n3mr_backward_line_walks + k_n3mr_backward_face) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB): "
               "(2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH doubled per MI355X_MICROARCH.md. 78 000 faces, 1024^2, rgb+alpha+depth. "
               "Source: profiles/%s_n3mr_pmc_*.txt" % tag}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "traffic_n3mr_latest.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
