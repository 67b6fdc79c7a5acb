"""Consistency of the measurement tooling with the sources it drives (CPU only)."""
import importlib.util
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_ablation_variants_name_existing_switches():
    """Every -D of tools/ablate/variants.py is a switch csrc/jr_tuning.h declares (a renamed or removed switch
    would silently build the product again and report a bogus A/B)."""
    tuning = open(os.path.join(ROOT, "jrender_amd", "csrc", "jr_tuning.h")).read()
    declared = set(re.findall(r"#ifndef (JR_TUNE_[A-Z0-9_]+)", tuning))
    variants = _load(os.path.join(ROOT, "tools", "ablate", "variants.py"), "variants").VARIANTS
    assert "product" in variants and variants["product"] == []
    for name, defines in variants.items():
        for d in defines:
            key, _, val = d.partition("=")
            assert key in declared, "variant %r uses %s, which jr_tuning.h does not declare" % (name, key)
            assert val.lstrip("-").isdigit(), d
    # and every declared switch is used by the kernels (no orphan defaults)
    csrc = "".join(open(os.path.join(ROOT, "jrender_amd", "csrc", f)).read()
                   for f in os.listdir(os.path.join(ROOT, "jrender_amd", "csrc")) if f.endswith((".hip", ".h")))
    for key in declared:
        assert len(re.findall(key, csrc)) >= 2, "%s is declared but never used" % key


def test_profile_jsons_feed_the_bench_line():
    """profiles/*_latest.json carry the keys bench.py looks up (a missing key prints `null` in the driver's line)."""
    valu = json.load(open(os.path.join(ROOT, "profiles", "valu_latest.json")))
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    n3 = json.load(open(os.path.join(ROOT, "profiles", "traffic_n3mr_latest.json")))
    for k in ("fwd_raster", "bwd_raster"):
        assert {"busy", "lane_util", "valu_insts_per_launch", "source"} <= set(valu[k])
        assert k in traffic
    # sanity of the magnitudes (bytes per launch; the NMR backward moved 3.3 GB before round 4's per-line walks, 0.39 GB since)
    assert traffic["bwd_raster"] > 5e8 and n3["fwd"] > 1e8 and 1e8 < n3["bwd"] < 1e9
    # all three were collected on the same kernel sources
    assert valu["csrc_hash"] == traffic["csrc_hash"] == n3["csrc_hash"]


def test_stale_counters_are_accounted_for():
    """bench.py's `roofline.profile_stale` / `.profiled_kernels_device_code_unchanged`: counters taken on other kernel sources than the
    shipped ones are flagged, and the flag is only softened by a tools/isa_same.py record about exactly these two source states that
    found the forward's and the backward's device code identical.  If the committed counters ARE stale, such a record must exist."""
    import bench
    valu = json.load(open(os.path.join(ROOT, "profiles", "valu_latest.json")))
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    here = bench.csrc_hash()
    iso_path = os.path.join(ROOT, "profiles", "isa_same_latest.json")
    iso = json.load(open(iso_path)) if os.path.exists(iso_path) else None
    stale, same = bench.profile_staleness(traffic, valu, iso, here)
    if stale:
        assert same is True, "profiles/*_latest.json were collected on other kernel sources and no tools/isa_same.py record covers the difference"
        assert iso["units"]["binning.hip"]["same"] is False or any(not u["same"] for u in iso["units"].values())       # (something did change)
    else:
        assert same is None
    # the rule itself
    t, v = {"csrc_hash": "aaa"}, {"csrc_hash": "aaa"}
    rec = {"old_csrc_hash": "aaa", "new_csrc_hash": "bbb", "units": {"softras_forward.hip": {"same": True}, "softras_backward.hip": {"same": True}, "binning.hip": {"same": False}}}
    assert bench.profile_staleness(t, v, None, "aaa") == (False, None)
    assert bench.profile_staleness(t, v, rec, "aaa") == (False, None)
    assert bench.profile_staleness(t, v, None, "bbb") == (True, None)
    assert bench.profile_staleness(t, v, rec, "bbb") == (True, True)
    assert bench.profile_staleness(t, v, rec, "ccc") == (True, None)                    # a record about other sources says nothing
    assert bench.profile_staleness(t, {"csrc_hash": "zzz"}, rec, "bbb") == (True, None)
    rec["units"]["softras_backward.hip"]["same"] = False
    assert bench.profile_staleness(t, v, rec, "bbb") == (True, False)
    del rec["units"]["softras_backward.hip"]
    assert bench.profile_staleness(t, v, rec, "bbb") == (True, False)


def test_isa_same_tool_on_an_unchanged_unit():
    """tools/isa_same.py on the smallest translation unit, HEAD against the working tree: the device assembly of an unchanged source
    must compare SAME although the two compilations run in different directories (the compilation-unit id hipcc derives from the
    source path is normalised away)."""
    import shutil
    import subprocess
    import sys
    if not os.path.isdir(os.path.join(ROOT, ".git")) or shutil.which("git") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        import pytest
        pytest.skip("needs the git checkout and hipcc")
    if subprocess.run(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", "jrender_amd/csrc/loss_kernels.hip", "jrender_amd/csrc/jr_kernels.h",
                       "jrender_amd/csrc/jr_tuning.h", "include/jrender_hip.h"]).returncode != 0:
        import pytest
        pytest.skip("the unit or its headers differ from HEAD in the working tree")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_same.py"), "HEAD", "loss_kernels.hip"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["units"]["loss_kernels.hip"]["same"] is True and rec["units"]["loss_kernels.hip"]["asm_lines"] > 500
    assert rec["old_csrc_hash"] and rec["new_csrc_hash"]


def test_counter_bump_model_of_the_list_building_kernels():
    """tools/sim/bin_atomics.py replays binning.hip's wave_bin_match (lanes of a wavefront that target the same bin share one atomic,
    one group per ballot-matching round) on the bench's synthetic scenes.  With unlimited rounds a trip issues one atomic per DISTINCT
    bin; fewer rounds can only add atomics; a mesh's consecutive faces share bins, a soup's do not - the measured times of
    profiles/r06_c16_match_rounds.txt follow these counts."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bin_atomics", os.path.join(ROOT, "tools", "sim", "bin_atomics.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from jrender_amd import synthetic as syn
    rounds = [1, 4, 8, 16, ("adaptive", 16), 64]
    mesh = m.count(syn.sphere_views(3300, 2)[0], 256, 16, rounds)
    soup = m.count(syn.triangle_soup(3300, 2, seed=0)[0], 256, 16, rounds)
    for r in (mesh, soup):
        a = r["atomics"]
        assert a[1] >= a[4] >= a[8] >= a[16] >= a[64] and a[("adaptive", 16)] >= a[16] and a[("adaptive", 16)] <= a[4] + r["trips"]
        assert abs(a[64] - r["mean_distinct_bins_per_trip"] * r["trips"]) < 1e-6 * a[64] + 1      # unlimited rounds: one atomic per distinct bin
        assert a[1] <= r["pairs"] and a[64] >= r["trips"]
    assert mesh["atomics"][("adaptive", 16)] < 0.6 * mesh["atomics"][4]               # what round 6 gained on meshes ...
    assert soup["atomics"][("adaptive", 16)] > 0.95 * soup["atomics"][4] > 0.9 * soup["pairs"]     # ... and could not gain on a soup
    # the matching itself, on hand-made trips
    k = np.array([5, 5, 7, 5, 9, 7, 11, 13, 13], np.int64)
    assert m.atomics_of_a_trip(k, 64) == 5 and m.atomics_of_a_trip(k, 1) == 1 + 6 and m.atomics_of_a_trip(k, 2) == 2 + 4
    ones = np.arange(40, dtype=np.int64)
    assert m.atomics_of_a_trip(ones, 16) == 40 and m.atomics_of_a_trip(ones, ("adaptive", 16)) == 40


def test_pipelined_heavy_tile_step_machine_model():
    """tools/sim/heavy_pipe_model.py: wavefront 3's decisions of tile_heavy_pipe (softras_forward.hip) replayed with owner
    tags on every double / triple buffer: each round evaluated once and applied once, in order, no buffer written
    while a reader of the same step needs it, and the loop ends - for random batch / round structures including
    one-round batches (bubbles) and a single batch."""
    import importlib.util
    import random
    spec = importlib.util.spec_from_file_location("heavy_pipe_model", os.path.join(ROOT, "tools", "sim", "heavy_pipe_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = random.Random(3)
    shapes = [[1], [1, 1, 1], [2, 1], [6, 6, 6, 6], [1, 4, 1, 4]] + [[rng.choice([1, 1, 2, 3, 4, 6]) for _ in range(rng.randint(1, 7))] for _ in range(3000)]
    for rp in shapes:
        ev, ap = m.run(rp)
        want = [(b, r) for b in range(len(rp)) for r in range(rp[b])]
        assert ev == want and ap == want, rp
