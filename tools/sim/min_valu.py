#!/usr/bin/env python3
"""How much of the forward's VALU work is dictated, how much is divergence, how much is overhead  (VERDICT r4 next #2).

The headline forward k_softras_forward<2,1,16> fills the VALU issue slots (92 % busy, profiles/valu_latest.json) at 56 % lane
use - so "11 % of the HBM roofline" is a statement about instructions, and this tool puts the claim "at the ceiling of this
formulation" on a checkable footing:

  static   the kernel's ISA (hipcc -S, cached), cut into true basic blocks; the blocks of the FAST-face raster loop are
           classified BY SIGNATURE into the regions of forward_pair (cuda/soft_rasterize.py:316-419 = softras_forward.hip:
           forward_pair): head + first edge projection (every pair), the two extra projections of inside pixels, coverage +
           clip + depth (pairs that pass the distance cull), K-buffer id store / append / replace + rescan, softmax update;
           the rest of the kernel as per-list-chunk (walk + staging), per-batch (ballots + pre-cull) and per-tile code.
           Every region gets its VALU count and its issue cycles at the measured opcode prices.
  dynamic  how often each region runs per launch - MEASURED on the GPU by the instrumented build `count_paths`
           (JR_TUNE_COUNT_PATHS: trips and lanes per region, batches, list chunks, tiles; profiles/r05_path_counts.json, written
           by `python tools/sim/min_valu.py --measure` on the GPU box).
  model    issued VALU per launch = sum over regions of static count x trips that execute it  -> must reproduce the PMC's
           SQ_INSTS_VALU (5.10e8) within 5 %: that is the check on the classification;
  floor    the same arithmetic with every trip at 64 lanes: sum over regions of static count x lanes / 64 - what THIS
           expression tree costs when no lane ever idles (the per-batch / per-tile code counted as it is);
  -> attainable_ms = floor issue cycles / (SIMDs x clock), frac_of_attainable = attainable / measured, and the split of the
     measured time into dictated arithmetic / idle lanes (divergence) / list walk, staging, masks (overhead).

bench.py reads profiles/min_valu_latest.json (written by the default run) for roofline.valu.attainable_ms / frac_of_attainable.

    python tools/sim/min_valu.py                 # CPU: ISA + committed counts -> table, profiles/min_valu_latest.json
    python tools/sim/min_valu.py --measure       # GPU box: run the instrumented build, write profiles/r05_path_counts.json
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "jrender_amd", "csrc", "softras_forward.hip")
ASM = os.environ.get("MIN_VALU_ASM", "/tmp/min_valu_forward.s")
COUNTS = os.path.join(ROOT, "profiles", "r05_path_counts.json")
OUT = os.path.join(ROOT, "profiles", "min_valu_latest.json")
KERNEL = "k_softras_forwardILi2ELi1ELi16ELi5"
CHEAP = ("v_mov_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32",
         "v_add_u32", "v_sub_u32", "v_and_b32")
SLOW = ("v_rcp_f32", "v_exp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32")
COUNTER_NAMES = ["trips", "lanes", "trips_inside", "lanes_inside", "trips_live", "lanes_live", "trips_insert", "lanes_insert",
                 "trips_append", "lanes_append", "trips_replace", "lanes_replace", "trips_softmax", "lanes_softmax",
                 "batches", "list_chunks", "tiles", "trips_slow", "lanes_slow"]


def price(op):
    """issue cycles per wavefront-instruction per SIMD, measured (tools/ubench/valu_rates2.hip, profiles/r02_valu_rates2.txt)"""
    if op.startswith(SLOW):
        return 8.3
    if op.startswith("v_pk_"):
        return 4.8
    if op.startswith(CHEAP) and "dpp" not in op:
        return 2.7
    return 4.3


def is_valu(i):
    return i.startswith("v_") and not i.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))


def basic_blocks():
    if not os.path.exists(ASM) or os.path.getmtime(ASM) < os.path.getmtime(SRC):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                        "-x", "hip", "--cuda-device-only", "-S", SRC, "-o", ASM], check=True, stderr=subprocess.DEVNULL)
    lines = open(ASM).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(KERNEL) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], ["entry", []]
    for l in lines[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            ins = l.strip()
            cur[1].append(ins)
            if ins.startswith(("s_cbranch", "s_branch")):       # a TRUE basic block ends at a branch
                blocks.append(cur)
                cur = [cur[0] + "+", []]
    blocks.append(cur)
    return blocks


def callee_valu(name_key):
    """(VALU, issue cycles) of a device function the kernel CALLS (pixel_masks<2,0,8>: the ballots + pre-cull of a batch
    are not inlined - s_swappc_b64 in the kernel's per-batch block)"""
    lines = open(ASM).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(name_key) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_setpc_b64" in lines[i])
    ins = [l.strip() for l in lines[start + 1:end + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    # (its few branches skip exec-masked regions - `lane < fill` - that every batch of the headline scene enters: counted whole)
    return valu_of(ins)


def n_of(ins, key):
    return sum(1 for i in ins if i.startswith(key))


def valu_of(ins):
    v = [i for i in ins if is_valu(i)]
    return len(v), sum(price(i.split()[0]) for i in v)


def classify(blocks):
    """-> {region: [valu, cycles]} and the annotated block table.  Signatures (checked against the source's structure):
         rescan      8 x v_max3 + 16 x (v_cmp, v_cndmask)          KBuffer::rescan
         replace     15-16 x (v_cmp, v_cndmask), no v_max3         KBuffer::insert, replace-the-maximum write
         append      >= 10 v_mov in a block without LDS reads      the register shift of an append
         inside      2 x v_rcp, no v_div_fmas, >= 8 v_pk_          the 2nd / 3rd edge projection (reciprocal multiply)
         project     1 x v_div_fmas + v_max3 + >= 8 v_cndmask      region choice + the first projection (IEEE quotient)
         clipdepth   >= 8 v_pk_ / v_fma with 3 ds_read and v_rcp   barycentric_clip + depth_of
         softmax     v_exp + v_pk_ + 2 v_cndmask                   softmax_accumulate
         coverage    v_exp + v_rcp in a block of <= 8 VALU         coverage_fast
       The kernel holds the loop twice (FAST faces / the rest); the FIRST copy is the FAST one, the second is priced by the
       dynamic `trips_slow` counter as a whole."""
    names = [b[0] for b in blocks]
    # the two loop copies: each starts at the block that pops the pixel's next face (v_ffbl_b32 x 2 + ds_read) and ends before the next one
    heads = [k for k, (_, ins) in enumerate(blocks) if n_of(ins, "v_ffbl_b32") >= 2 and n_of(ins, "ds_read") >= 1]
    assert len(heads) == 1, "raster loop head not found: %r" % [names[h] for h in heads]
    # the head pops the pixel's next face, reads its meta word and branches on FLAG_SAFE: the FAST copy follows, the SLOW copy
    # starts at the label that branch names
    lo = heads[0]
    target = [i.split()[1] for i in blocks[lo][1] if i.startswith("s_cbranch")][0]
    hi = names.index(target)
    regions = {k: [0, 0.0] for k in ("every_pair", "obtuse_corner", "inside", "live", "insert", "append", "replace", "softmax",
                                    "modes_not_taken", "slow_loop", "per_chunk", "per_batch", "per_tile")}
    table = []

    def add(region, ins, name):
        v, c = valu_of(ins)
        regions[region][0] += v
        regions[region][1] += c
        table.append((name, region, v, c))

    # the alpha-mode switch (uniform branches on p.alpha between the coverage and the clip / depth block): this workload's
    # mode is 'prod' - one fused multiply-add; the blocks of the other modes, including the ones the switch jumps FORWARD to,
    # are code this launch never runs
    cov = next(k for k in range(lo, hi) if n_of(blocks[k][1], "v_exp") >= 1 and n_of(blocks[k][1], "v_rcp") >= 1 and valu_of(blocks[k][1])[0] <= 8)
    clip = next(k for k in range(cov, hi) if n_of(blocks[k][1], "ds_read") >= 3 and n_of(blocks[k][1], "v_rcp") >= 1)
    alpha_other = set()
    far = set()
    for k in range(cov + 1, clip):
        ins = blocks[k][1]
        only_fma = all(i.startswith(("v_fma", "v_fmac")) for i in ins if is_valu(i))
        if valu_of(ins)[0] and not only_fma:
            alpha_other.add(k)
        far.update(i.split()[1] for i in ins if i.startswith(("s_cbranch", "s_branch")))
    for k in range(clip + 1, hi):
        if blocks[k][0].rstrip("+") in far:
            alpha_other.add(k)
    seen_project = False
    for k in range(lo, hi):
        name, ins = blocks[k]
        v, _c = valu_of(ins)
        if v == 0:
            continue
        if k in alpha_other:
            add("modes_not_taken", ins, name)
            continue
        mx3, cnd, cmp_, rcp, divf, exp, pk, mov, dsr = (n_of(ins, x) for x in ("v_max3", "v_cndmask", "v_cmp", "v_rcp", "v_div_fmas", "v_exp", "v_pk_", "v_mov_b32", "ds_read"))
        gl, gs = n_of(ins, "global_load"), n_of(ins, "global_store")
        if mx3 >= 6 and cnd >= 12:
            r = "replace"
        elif cnd >= 12 and cmp_ >= 12 and mx3 == 0:
            r = "replace"
        elif mov >= 10 and dsr == 0 and cnd <= 3:
            r = "append"
        elif mov >= 4 and v == mov and seen_project and any(t[1] == "append" for t in table):
            r = "append"
        elif divf >= 5:
            r = "modes_not_taken"          # vertex colours: nine IEEE quotients (texture_type 'vertex')
        elif gl >= 1:
            r = "modes_not_taken"          # per-texel surface colour (T > 1)
        elif divf == 1 and mx3 >= 1 and cnd >= 8:
            r, seen_project = "every_pair", True
        elif rcp >= 2 and divf == 0 and pk >= 8 and seen_project and regions["inside"][0] == 0:
            r = "inside"
        elif exp >= 1 and rcp >= 1 and v <= 8:
            r = "live"
        elif exp >= 1 and (pk >= 2 or cnd >= 2):
            r = "softmax"
        elif dsr >= 3 and rcp >= 1 and (pk + n_of(ins, "v_fma")) >= 12:
            r = "live"
        elif gs >= 1:
            r = "insert"
        elif not seen_project:
            r = "obtuse_corner" if (dsr == 2 and pk >= 2 and cnd >= 1) or (v <= 3 and cmp_ == 1 and table and table[-1][1] == "every_pair" and n_of(blocks[k + 1][1], "ds_read") == 2) else "every_pair"
        elif divf == 1 and v <= 13:
            r = "modes_not_taken"          # IEEE fall-backs behind uniform / never-true branches (1/s outside the fast range, alpha 'hard', x / gamma)
        else:
            r = None
        if r is None:
            # small blocks after the projection: order in the source - cull compare, alpha, depth cull / K-buffer entry, facing test
            r = "live" if regions["softmax"][0] == 0 else "softmax"
            if v <= 6 and cmp_ >= 1 and regions["live"][0] == 0:
                r = "every_pair"           # the distance-cull compare that ends a rejected pair
        add(r, ins, name)
    for k in range(hi, len(blocks)):
        name, ins = blocks[k]
        if n_of(ins, "global_store") >= 1 and n_of(ins, "v_ffbl_b32") == 0 and k > hi + 20:
            break
    # the SLOW copy: from its head to the block before the epilogue's first store block (found by the 16 id-plane stores after it)
    ep = next(k for k in range(hi + 1, len(blocks)) if n_of(blocks[k][1], "global_store") >= 4)
    for k in range(hi, ep):
        add("slow_loop", blocks[k][1], blocks[k][0])
    for k in range(ep, len(blocks)):
        add("per_tile", blocks[k][1], blocks[k][0])
    # before the loop: the tile's prologue ends with the jump into the batch loop (the largest straight block: pixel centres,
    # initial pixel state); then the list walk + record copies (per list chunk) up to the call of pixel_masks, and from that
    # call to the raster loop's head the per-batch code
    kp = max((k for k in range(0, lo) if blocks[k][1] and blocks[k][1][-1].startswith("s_branch")), key=lambda k: len(blocks[k][1]))
    kc = next(k for k in range(0, lo) if n_of(blocks[k][1], "s_swappc_b64") >= 1)
    for k in range(0, lo):
        name, ins = blocks[k]
        if k <= kp:
            add("per_tile", ins, name)
        elif k < kc:
            add("per_chunk", ins, name)
        else:
            add("per_batch", ins, name)
            if k == kc:
                v, c = callee_valu("pixel_masksILi2ELi0ELi8E")
                regions["per_batch"][0] += v; regions["per_batch"][1] += c
                table.append(("pixel_masks<2,0,8>", "per_batch", v, c))
    return regions, table


def measure():
    """GPU box: the instrumented build on the headline batch -> profiles/r05_path_counts.json"""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("JRENDER_LIB", os.path.join(ROOT, "jrender_amd", "csrc", "libjrender_hip_count_paths.so"))
    import numpy as np
    from jrender_amd import _ffi, synthetic as syn
    from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
    ctx = _ffi.Context(0)
    fv, tex = syn.sphere_views(39000, 8)
    fn = SoftRasterizeFunction(image_size=1024, ctx=ctx)
    fv, tex = ctx.array(fv), ctx.array(tex)
    fn.execute(fv, tex)
    ctx.section_clocks()
    n = 3
    for _ in range(n):
        fn.execute(fv, tex)
    c = np.asarray(ctx.section_clocks(), np.float64) / n
    out = {k: float(v) for k, v in zip(COUNTER_NAMES, c)}
    out["workload"] = "8 views x 39 000 faces x 1024^2, K = 16, Renderer defaults; per launch of k_softras_forward<2,1,16>"
    json.dump(out, open(COUNTS, "w"), indent=1)
    print(json.dumps(out, indent=1))


def main():
    if "--measure" in sys.argv:
        return measure()
    regions, table = classify(basic_blocks())
    if "--blocks" in sys.argv:
        for name, r, v, c in table:
            print("%-14s %-16s valu %4d  cycles %6.0f" % (name.replace(".LBB", ""), r, v, c))
    print("k_softras_forward<2,1,16>: static VALU per region (instructions, issue cycles at the measured opcode prices)")
    for r, (v, c) in regions.items():
        print("   %-16s %5d  %7.0f" % (r, v, c))
    if not os.path.exists(COUNTS):
        print("\n(no dynamic counts: run `python tools/sim/min_valu.py --measure` on the GPU box)")
        return
    d = json.load(open(COUNTS))
    valu = json.load(open(os.path.join(ROOT, "profiles", "valu_latest.json")))["fwd_raster"]
    # region -> (trips that execute it, lanes that need it)
    dyn = {"every_pair": ("trips", "lanes"), "inside": ("trips_inside", "lanes_inside"), "live": ("trips_live", "lanes_live"),
           "insert": ("trips_insert", "lanes_insert"), "append": ("trips_append", "lanes_append"),
           "replace": ("trips_replace", "lanes_replace"), "softmax": ("trips_softmax", "lanes_softmax")}
    issued = issued_cyc = floor = floor_cyc = 0.0
    print("\nper launch (%s):" % d["workload"])
    print("   %-16s %10s %12s | %12s %12s" % ("region", "trips", "lanes / 64", "issued VALU", "at 64 lanes"))
    for r, (tk, lk) in dyn.items():
        v, c = regions[r]
        issued += v * d[tk]; issued_cyc += c * d[tk]
        floor += v * d[lk] / 64.0; floor_cyc += c * d[lk] / 64.0
        print("   %-16s %10.0f %12.0f | %12.3e %12.3e" % (r, d[tk], d[lk] / 64.0, v * d[tk], v * d[lk] / 64.0))
    # the SLOW copy (faces outside the fast-arithmetic range): priced whole per trip
    vs, cs = regions["slow_loop"]
    issued += 0.6 * vs * d["trips_slow"]; issued_cyc += 0.6 * cs * d["trips_slow"]
    floor += 0.6 * vs * d["lanes_slow"] / 64.0; floor_cyc += 0.6 * cs * d["lanes_slow"] / 64.0
    over = over_cyc = 0.0
    for r, k in (("per_chunk", "list_chunks"), ("per_batch", "batches"), ("per_tile", "tiles")):
        v, c = regions[r]
        over += v * d[k]; over_cyc += c * d[k]
        print("   %-16s %10.0f %12s | %12.3e %12s" % (r, d[k], "-", v * d[k], "(as issued)"))
    total = issued + over
    print("\nmodel: %.3e VALU wavefront-instructions per launch (raster loop %.3e + walk / staging / masks / tile %.3e); PMC SQ_INSTS_VALU %.3e -> model / measured = %.3f"
          % (total, issued, over, valu["valu_insts_per_launch"], total / valu["valu_insts_per_launch"]))
    simds, ghz = valu["simds"], valu["clock_ghz"]
    ms = lambda cyc: cyc / simds / (ghz * 1e6)      # noqa: E731
    t_meas = valu["avg_launch_ns"] * 1e-6
    t_issued, t_floor, t_over = ms(issued_cyc + over_cyc), ms(floor_cyc + over_cyc), ms(over_cyc)
    print("issue time of the modelled stream on %d SIMDs at %.1f GHz: %.3f ms (measured launch %.3f ms: the issue slots are %.0f %% busy)"
          % (simds, ghz, t_issued, t_meas, 100 * t_issued / t_meas))
    print("the same arithmetic with every trip at 64 lanes: %.3f ms  = attainable_ms; frac_of_attainable = %.3f" % (t_floor, t_floor / t_meas))
    print("   dictated arithmetic (the reference's expression tree per surviving pair, 100 %% lanes)  %.3f ms  %4.1f %%" % (ms(floor_cyc), 100 * ms(floor_cyc) / t_meas))
    print("   idle lanes inside the raster loop (divergence: issued - floor)                        %.3f ms  %4.1f %%" % (ms(issued_cyc - floor_cyc), 100 * ms(issued_cyc - floor_cyc) / t_meas))
    print("   list walk, record staging, ballots + pre-cull, tile prologue / stores               %.3f ms  %4.1f %%" % (t_over, 100 * t_over / t_meas))
    print("   issue slots not used (latency the 20 wavefronts per CU do not cover)                 %.3f ms  %4.1f %%" % (t_meas - t_issued, 100 * (t_meas - t_issued) / t_meas))
    gap = (issued_cyc - floor_cyc) / max(issued_cyc + over_cyc, 1)
    owner = max(dyn, key=lambda r: regions[r][1] * (d[dyn[r][0]] - d[dyn[r][1]] / 64.0))
    print("gap between issued and floor: %.1f %% of the issued cycles; the region that owns most of it: %s" % (100 * gap, owner))
    json.dump({"kernel": "k_softras_forward<2,1,16>", "model_valu_per_launch": total, "measured_valu_per_launch": valu["valu_insts_per_launch"],
               "model_over_measured": total / valu["valu_insts_per_launch"], "attainable_ms": t_floor, "issued_ms": t_issued,
               "measured_ms_of_the_profile": t_meas, "frac_of_attainable": t_floor / t_meas,
               "split_ms": {"dictated_arithmetic": ms(floor_cyc), "idle_lanes": ms(issued_cyc - floor_cyc), "walk_staging_masks": t_over,
                            "unused_issue_slots": t_meas - t_issued},
               "gap_issued_vs_floor": gap, "gap_owner": owner,
               "source": "tools/sim/min_valu.py: ISA of the shipped kernel x profiles/r05_path_counts.json (instrumented GPU run)"},
              open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
