#!/usr/bin/env python3
"""profiles/<tag>_* (tools/collect_profiles.sh) -> profiles/traffic_latest.json, profiles/valu_latest.json:
the per-launch HBM bytes and VALU figures of the two raster kernels that bench.py attaches to its
`roofline` object.  usage: tools/pmc_to_json.py <dir with pmc csv summaries> <tag> [clock_ghz]

  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes   (rocprofv3 reports KB; FETCH_SIZE is doubled on gfx950
            per /opt/skills/guides/MI355X_MICROARCH.md §HBM; WRITE_SIZE is uncalibrated and includes atomics)
  busy      = SQ_INSTS_VALU * mean_issue_cycles / (1024 SIMDs * kernel cycles), clamped to 1.  mean_issue_cycles is the
              mean issue cost of the kernel's VALU opcode mix (static count over the kernel's ISA, MIX below) at the prices
              measured on MI355X (tools/ubench/valu_rates2.hip, profiles/r02_valu_rates2.txt: v_mov / add / mul / fma_f32,
              v_add_u32, v_and 2.7 cycles, v_rcp / v_sqrt 8.3, v_exp 5.0 more than its multiply, f64 4.7, the rest 4.3).
              (SQ_ACTIVE_INST_VALU charges every instruction 4 cycles, which gave 1.12 for the backward in round 3: kept as
              busy_raw_4cycle.)  bench.py recomputes busy against the launch time IT measures.
  lane_util = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  useful_lane_frac = busy * lane_util: the share of the chip's VALU lane-slots that carried a pair's arithmetic
  valu_insts = SQ_INSTS_VALU per launch
Both files are stamped with bench.csrc_hash() of the sources they were collected on."""
import csv
import collections
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import csrc_hash                 # noqa: E402

# static VALU opcode mix of k_softras_forward<2,1,16> / k_softras_backward<2,1,16> (1 628 / 1 641 VALU instructions in the
# ISA, 43 % / 40 % of them at the 2.7-cycle price): mean issue cycles per wavefront-instruction
MIX = {"fwd_raster": 3.74, "bwd_raster": 3.75}
d, tag = sys.argv[1], sys.argv[2]
clock = float(sys.argv[3]) if len(sys.argv) > 3 else 2.4
KERNELS = {"fwd_raster": "k_softras_forward", "bwd_raster": "k_softras_backward",
           "n3mr_fwd": "k_n3mr_resolve", "n3mr_bwd": "k_n3mr_backward_pixel_map"}


def means(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def pick(m, needle):
    for k, v in m.items():
        if needle in k:
            return v
    return {}


def find(sub):
    for root, _, files in os.walk(os.path.join(d, sub)):
        for f in files:
            if f.endswith("counter_collection.csv"):
                return os.path.join(root, f)
    return None


durations = {}
ks = os.path.join(d, "%s_kernel_stats.csv" % tag)
if os.path.exists(ks):
    with open(ks) as f:
        for r in csv.DictReader(f):
            durations[r["Name"]] = float(r["AverageNs"])
fetch, write, sq = (means(find(s)) if find(s) else {} for s in ("fetch", "write", "sq"))
traffic, valu = {}, {}
for key, needle in KERNELS.items():
    fs, ws = pick(fetch, needle).get("FETCH_SIZE"), pick(write, needle).get("WRITE_SIZE")
    if fs is not None and ws is not None:
        traffic[key] = (2 * fs + ws) * 1024
    s = pick(sq, needle)
    ns = next((v for k, v in durations.items() if needle in k), None)
    if s and ns:
        cycles = ns * clock
        mic = MIX.get(key, 4.0)
        busy = min(1.0, s["SQ_INSTS_VALU"] * mic / (1024 * cycles))
        lane = s["SQ_THREAD_CYCLES_VALU"] / (64 * s["SQ_ACTIVE_INST_VALU"])
        valu[key] = {"busy": busy, "busy_raw_4cycle": 4 * s["SQ_ACTIVE_INST_VALU"] / (1024 * cycles),
                     "lane_util": lane, "useful_lane_frac": busy * lane, "mean_issue_cycles": mic, "clock_ghz": clock, "simds": 1024,
                     "valu_insts_per_launch": s["SQ_INSTS_VALU"], "salu_insts_per_launch": s.get("SQ_INSTS_SALU"),
                     "lds_insts_per_launch": s.get("SQ_INSTS_LDS"), "avg_launch_ns": ns,
                     "source": "profiles/%s_pmc_sq.txt, %s_kernel_stats.csv; %.1f GHz, 1024 SIMDs" % (tag, tag, clock)}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
traffic["note"] = ("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units): "
                   "(2*FETCH_SIZE + WRITE_SIZE)*1024 - FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 reports "
                   "half of a wide coalesced read); WRITE_SIZE uncalibrated, includes the backward's float atomics. Source: profiles/%s_pmc_*.txt" % tag)
traffic["csrc_hash"] = valu["csrc_hash"] = csrc_hash()
json.dump(traffic, open(os.path.join(out, "traffic_latest.json"), "w"), indent=1)
json.dump(valu, open(os.path.join(out, "valu_latest.json"), "w"), indent=1)
print(json.dumps({"traffic": traffic, "valu": valu}, indent=1))
