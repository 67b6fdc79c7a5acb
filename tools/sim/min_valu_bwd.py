#!/usr/bin/env python3
"""VALU floor of the headline BACKWARD k_softras_backward<2,1,16,false>  (VERDICT r5 next #1a / #3; the forward's: min_valu.py).

  static   the kernel's ISA (hipcc -S, cached) cut into basic blocks and grouped by the LOOP NEST the compiler annotates
           (`; in Loop: Header=... Depth=n`): depth 0 = a tile's prologue (up to the "nothing buffered in this tile" exit: every
           launched wavefront; the rest: tiles with work), depth 1 = one union pass (table init, K hash probes, compaction),
           depth 2 = one batch (record staging, work items), depth 3 = one TRIP of the pair loop.  A trip is split by signature:
           holder search + 13 gathers + next-item look-up + row reduction + flush (ORGANISATION: paid per trip whatever the lanes
           do), the FAST copy of backward_pair (first of two structurally identical block runs; the second, full of IEEE
           divisions, is the copy for faces outside the fast-arithmetic range), inside it the two extra edge projections of
           INSIDE pairs, and the blocks of modes this workload does not run (per-texel atomics, the vertex-colour reduction).
  dynamic  tiles, union passes, batches, trips, lanes with a pair, trips / lanes with an inside pair, trips / lanes of non-FAST
           faces - MEASURED by the instrumented build `count_paths_bwd` (JR_TUNE_COUNT_PATHS=2; `--measure` on the GPU box
           -> profiles/r06_path_counts_bwd.json).
  model    issued VALU per launch = sum of static x dynamic -> must reproduce the PMC's SQ_INSTS_VALU (profiles/valu_latest.json);
  floor    the same stream with every trip at 64 lanes (pairs / 64 trips; organisation and tile code as issued per full trip).
  -> profiles/min_valu_bwd_latest.json: attainable_ms, frac_of_attainable, the split of the measured launch into dictated pair
     arithmetic / idle lanes / organisation per trip / union + staging + items + tile code / unused issue slots
     (bench.py: roofline.valu_bwd).

    python tools/sim/min_valu_bwd.py              # CPU: ISA + committed counts
    python tools/sim/min_valu_bwd.py --measure    # GPU box: run the instrumented build
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "jrender_amd", "csrc", "softras_backward.hip")
ASM = os.environ.get("MIN_VALU_BWD_ASM", "/tmp/min_valu_backward.s")
COUNTS = os.path.join(ROOT, "profiles", "r06_path_counts_bwd.json")
OUT = os.path.join(ROOT, "profiles", "min_valu_bwd_latest.json")
KERNEL = "k_softras_backwardILi2ELi1ELi16ELb0"
COUNTER_NAMES = ["tiles", "passes", "batches", "trips", "lanes", "trips_inside", "lanes_inside", "trips_slow", "lanes_slow", "faces", "items"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from min_valu import price, is_valu          # noqa: E402  (the measured opcode prices)


def blocks_with_loops():
    deps = [SRC] + [os.path.join(os.path.dirname(SRC), h) for h in ("softras_device.h", "jr_tuning.h", "jr_kernels.h")]
    if not os.path.exists(ASM) or os.path.getmtime(ASM) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                        "-x", "hip", "--cuda-device-only", "-S", SRC, "-o", ASM], check=True, stderr=subprocess.DEVNULL)
    lines = open(ASM).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(KERNEL) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    out, cur = [], dict(name="entry", depth=0, hdr=None, ins=[])
    for l in lines[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l) or re.match(r"^; %(bb\.\d+):(.*)", l)
        if m:
            out.append(cur)
            cur = dict(name=m.group(1), depth=0, hdr=None, ins=[])
            l = m.group(2)
        s = l.strip()
        if s.startswith(";") or (m and ";" in l):
            d = re.search(r"Depth=(\d+)", l)
            if "in Loop: Header=" in l and d:
                cur["depth"], cur["hdr"] = int(d.group(1)), re.search(r"Header=(BB\d+_\d+)", l).group(1)
            elif "This" in l and "Loop Header" in l and d:
                cur["depth"], cur["hdr"] = int(d.group(1)), cur["name"].replace(".L", "")
            continue
        if l.startswith("\t") and not s.startswith((".", ";")):
            cur["ins"].append(s)
    out.append(cur)
    return out


def n_of(ins, key):
    return sum(1 for i in ins if key in i.split()[0])


def valu_of(ins):
    v = [i for i in ins if is_valu(i)]
    return len(v), sum(price(i.split()[0]) for i in v)


def sig(b):
    i = b["ins"]
    return (valu_of(i)[0], n_of(i, "v_div_fmas"), n_of(i, "v_rcp"), n_of(i, "ds_read"), n_of(i, "v_cndmask"))


def classify(blocks):
    regions = {k: [0, 0.0] for k in ("exit_prologue", "tile", "pass", "probe_loops", "batch", "trip_search_gather", "trip_reduce_flush",
                                    "pair_fast", "pair_inside", "pair_slow", "modes_not_taken")}
    table = []

    def add(r, b):
        v, c = valu_of(b["ins"])
        regions[r][0] += v
        regions[r][1] += c
        table.append((b["name"], b["depth"], r, v, c))

    # loop headers by role
    d3 = [b for b in blocks if b["depth"] == 3]
    trip_hdr = max(set(b["hdr"] for b in d3), key=lambda h: sum(valu_of(b["ins"])[0] for b in d3 if b["hdr"] == h))
    trip = [k for k, b in enumerate(blocks) if b["depth"] >= 3 and (b["hdr"] == trip_hdr or b["depth"] > 3) and trip_idx_ok(blocks, k, trip_hdr)]
    lo, hi = trip[0], trip[-1]
    batch_hdr = next(b["hdr"] for b in blocks[lo - 1::-1] if b["depth"] == 2)
    # depth 0: the prologue up to the empty-tile exit = everything before the first block with >= 8 global loads (the pixel state)
    first_state = next(k for k, b in enumerate(blocks) if b["depth"] == 0 and n_of(b["ins"], "global_load") >= 8)
    for k, b in enumerate(blocks):
        if b["depth"] == 0:
            add("exit_prologue" if k < first_state else "tile", b)
        elif b["depth"] == 1:
            add("pass", b)
        elif b["depth"] == 2 and b["hdr"] != batch_hdr:
            add("probe_loops", b)                # a lane whose first probe hit another id walks on (rare; not counted dynamically)
        elif b["depth"] == 2:
            add("batch", b)
    # the trip: FAST copy = first of two runs of blocks with the same signatures
    sg = [sig(blocks[k]) for k in range(lo, hi + 1)]
    fast0 = slow0 = None
    for i in range(len(sg)):
        for j in range(i + 8, len(sg) - 5):
            if sg[i:i + 5] == sg[j:j + 5] and sum(s[0] for s in sg[i:i + 5]) >= 12:
                fast0, slow0 = lo + i, lo + j
                break
        if fast0 is not None:
            break
    assert fast0 is not None, "the two copies of backward_pair were not found"
    slow1 = next(k for k in range(slow0, hi + 1) if n_of(blocks[k]["ins"], "global_atomic") >= 1)       # the per-texel atomics follow the pair code
    proj = [k for k in range(fast0, slow0) if sig(blocks[k])[1] == 1 and sig(blocks[k])[2] == 1 and 24 <= sig(blocks[k])[0] <= 34 and sig(blocks[k])[3] in (2, 3)]
    inside = set()
    if len(proj) >= 2:                                            # the 2nd / 3rd edge projection + the register moves that select the nearest candidate
        k0, k1 = proj[0], proj[-1]
        inside.update(range(k0, k1 + 1))
        for k in (k0 - 2, k0 - 1, k1 + 1):
            v = valu_of(blocks[k]["ins"])[0]
            if v and n_of(blocks[k]["ins"], "v_mov_b32") >= v - 4 and v <= 10:
                inside.add(k)
    seen_dpp = 0
    for k in range(lo, hi + 1):
        b = blocks[k]
        if not valu_of(b["ins"])[0]:
            continue
        if k < fast0:
            add("trip_search_gather", b)
        elif k < slow0:
            sv = sig(b)
            if k not in inside and sv[1] == 1 and sv[0] <= 13:
                add("modes_not_taken", b)            # IEEE fall-backs behind rarely-true branches (1 / s outside the fast range, x / sigma for 'hard' alpha ...)
            else:
                add("pair_inside" if k in inside else "pair_fast", b)
        elif k < slow1:
            add("pair_slow", b)
        else:
            dpp = sum(1 for i in b["ins"] if "_dpp" in i)
            if n_of(b["ins"], "global_atomic") >= 3 or (b["depth"] > 3 and seen_dpp == 0 and n_of(b["ins"], "ds_read") == 0):
                add("modes_not_taken", b)            # per-texel atomics (T > 1)
            elif dpp >= 8:
                seen_dpp += 1
                add("trip_reduce_flush" if seen_dpp == 1 else "modes_not_taken", b)     # (the second reduction: vertex colours)
            elif seen_dpp >= 2:
                add("modes_not_taken", b)
            else:
                add("trip_reduce_flush", b)
    return regions, table


def trip_idx_ok(blocks, k, hdr):
    """block k lies inside the trip loop: its own header is the trip loop's, or it sits in a deeper loop nested there"""
    b = blocks[k]
    if b["hdr"] == hdr:
        return True
    # a depth-4 block: accept when the nearest depth-3 neighbours belong to the trip loop
    for j in range(k - 1, -1, -1):
        if blocks[j]["depth"] == 3:
            return blocks[j]["hdr"] == hdr
        if blocks[j]["depth"] < 3:
            return False
    return False


def measure():
    sys.path.insert(0, ROOT)
    os.environ.setdefault("JRENDER_LIB", os.path.join(ROOT, "jrender_amd", "csrc", "libjrender_hip_count_paths_bwd.so"))
    import numpy as np
    from jrender_amd import _ffi, synthetic as syn
    from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
    ctx = _ffi.Context(0)
    fv, tex = syn.sphere_views(39000, 8)
    fn = SoftRasterizeFunction(image_size=1024, ctx=ctx)
    fv, tex = ctx.array(fv), ctx.array(tex)
    g = ctx.array(np.random.default_rng(7).uniform(-1, 1, (8, 4, 1024, 1024)).astype(np.float32))
    fn.execute(fv, tex); fn.grad(g)
    ctx.section_clocks()
    n = 3
    for _ in range(n):
        fn.execute(fv, tex); fn.grad(g)
    c = np.asarray(ctx.section_clocks(), np.float64) / n
    out = {k: float(v) for k, v in zip(COUNTER_NAMES, c)}
    st = ctx.last_stats()
    out["launched_wavefronts"] = float(8 * st["bins_per_image"] * (ctx.bin_size() // 8) ** 2)
    out["workload"] = "8 views x 39 000 faces x 1024^2, K = 16, Renderer defaults; per launch of k_softras_backward<2,1,16,false>"
    json.dump(out, open(COUNTS, "w"), indent=1)
    print(json.dumps(out, indent=1))


def main():
    if "--measure" in sys.argv:
        return measure()
    regions, table = classify(blocks_with_loops())
    if "--blocks" in sys.argv:
        for name, d, r, v, c in table:
            if v:
                print("%-12s d%d %-20s valu %4d  cycles %6.0f" % (name.replace(".LBB", ""), d, r, v, c))
    print("k_softras_backward<2,1,16,false>: static VALU per region (instructions, issue cycles at the measured opcode prices)")
    for r, (v, c) in regions.items():
        print("   %-20s %5d  %7.0f" % (r, v, c))
    if not os.path.exists(COUNTS):
        print("\n(no dynamic counts: run `python tools/sim/min_valu_bwd.py --measure` on the GPU box)")
        return
    d = json.load(open(COUNTS))
    valu = json.load(open(os.path.join(ROOT, "profiles", "valu_latest.json")))["bwd_raster"]
    R = lambda r: regions[r]          # noqa: E731
    # issued: region x how often it runs;  floor: the pair loop at 64 lanes per trip
    trips, lanes = d["trips"], d["lanes"]
    full = lanes / 64.0                                           # trips if every lane always held a pair
    rows = [  # name, static region, issued multiplier, floor multiplier, class
        ("tile prologue up to the empty-tile exit", "exit_prologue", d["launched_wavefronts"], d["launched_wavefronts"], "tile"),
        ("tile state", "tile", d["tiles"], d["tiles"], "tile"),
        ("union pass (table, K probes, compaction)", "pass", d["passes"], d["passes"], "tile"),
        ("batch (record staging, work items)", "batch", d["batches"], d["batches"], "tile"),
        ("trip: holder search + gathers", "trip_search_gather", trips, full, "org"),
        ("trip: reduction, flush, next item", "trip_reduce_flush", trips, full, "org"),
        ("pair arithmetic (FAST faces)", "pair_fast", trips, full, "pair"),
        ("  two more projections of inside pairs", "pair_inside", d["trips_inside"], d["lanes_inside"] / 64.0, "pair"),
        ("pair arithmetic (other faces, x 0.6)", "pair_slow", 0.6 * d["trips_slow"], 0.6 * d["lanes_slow"] / 64.0, "pair"),
    ]
    tot = {"issued": 0.0, "issued_cyc": 0.0, "floor_cyc": 0.0}
    cls = {"tile": [0.0, 0.0], "org": [0.0, 0.0], "pair": [0.0, 0.0]}       # issued cycles, floor cycles
    print("\nper launch (%s):" % d["workload"])
    print("   %-44s %12s %12s | %12s" % ("region", "runs", "at 64 lanes", "issued VALU"))
    for name, r, mi, mf, c in rows:
        v, cy = R(r)
        tot["issued"] += v * mi; tot["issued_cyc"] += cy * mi; tot["floor_cyc"] += cy * mf
        cls[c][0] += cy * mi; cls[c][1] += cy * mf
        print("   %-44s %12.0f %12.0f | %12.3e" % (name, mi, mf, v * mi))
    print("\nmodel: %.3e VALU wavefront-instructions per launch; PMC SQ_INSTS_VALU %.3e -> model / measured = %.3f"
          % (tot["issued"], valu["valu_insts_per_launch"], tot["issued"] / valu["valu_insts_per_launch"]))
    simds, ghz = valu["simds"], valu["clock_ghz"]
    ms = lambda cyc: cyc / simds / (ghz * 1e6)      # noqa: E731
    t_meas = valu["avg_launch_ns"] * 1e-6
    t_issued, t_floor = ms(tot["issued_cyc"]), ms(tot["floor_cyc"])
    split = {"dictated_pair_arithmetic": ms(cls["pair"][1]), "idle_lanes": ms(cls["pair"][0] - cls["pair"][1] + cls["org"][0] - cls["org"][1]),
             "organisation_per_trip_at_full_lanes": ms(cls["org"][1]), "union_staging_items_tile": ms(cls["tile"][0]),
             "unused_issue_slots": max(0.0, t_meas - t_issued)}
    print("issue time of the modelled stream on %d SIMDs at %.1f GHz: %.3f ms (measured launch %.3f ms)" % (simds, ghz, t_issued, t_meas))
    print("the same stream with every trip at 64 lanes: %.3f ms = attainable_ms; frac_of_attainable = %.3f" % (t_floor, t_floor / t_meas))
    for k, v in split.items():
        print("   %-40s %.3f ms  %4.1f %%" % (k, v, 100 * v / t_meas))
    print("lane use of the pair loop: %.1f of 64 (%.0f trips for %.3e pairs); pair arithmetic alone at full lanes: %.3f ms"
          % (lanes / trips, trips, lanes, ms(cls["pair"][1])))
    json.dump({"kernel": "k_softras_backward<2,1,16,false>", "model_valu_per_launch": tot["issued"],
               "measured_valu_per_launch": valu["valu_insts_per_launch"], "model_over_measured": tot["issued"] / valu["valu_insts_per_launch"],
               "attainable_ms": t_floor, "issued_ms": t_issued, "measured_ms_of_the_profile": t_meas, "frac_of_attainable": t_floor / t_meas,
               "pair_arithmetic_only_ms": ms(cls["pair"][1]), "split_ms": split, "lanes_per_trip": lanes / trips,
               "static_valu": {k: v[0] for k, v in regions.items()},
               "source": "tools/sim/min_valu_bwd.py: ISA of the shipped kernel x profiles/r06_path_counts_bwd.json (instrumented GPU run)"},
              open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
