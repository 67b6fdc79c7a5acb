#!/usr/bin/env python3
"""Which translation units of jrender_amd/csrc produce the SAME gfx950 device code in two source trees?

    python tools/isa_same.py <git-rev> [--write] [unit.hip ...]        (CPU only: hipcc cross-compiles; ~ 8 minutes for all units)

Compiles every .hip unit of <git-rev> (git archive into a temporary directory) and of the working tree to device assembly with the
product's flags (jrender_amd/_build.py), drops the lines that carry paths or compiler identification, and compares.  Purpose: the
counters under profiles/ are stamped with a hash of ALL kernel sources (bench.csrc_hash); when a later commit touches one unit
only, this says - reproducibly - for which kernels the committed counters still describe the shipped code object.
Prints one line per unit and a JSON summary (last line); --write also stores it as profiles/isa_same_latest.json with bench.csrc_hash()
of both trees, where bench.py finds it: a line whose counters are stale by the source hash then says whether the two profiled kernels'
device code is unchanged (`roofline.profiled_kernels_device_code_unchanged`)."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jrender_amd import _build                      # noqa: E402


CUID = re.compile(r"__hip_cuid_[0-9a-f]+")


def device_asm(tree, unit):
    src = os.path.join(tree, "jrender_amd", "csrc", unit)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "u.s")
        flags = [f for f in _build.FLAGS if f != "-fPIC"]
        subprocess.check_call([_build.HIPCC, *flags, "-S", "--cuda-device-only", "-I", os.path.join(tree, "include"), src, "-o", out],
                              cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
        keep = []
        for line in open(out, errors="replace"):
            t = line.strip()
            if t.startswith((".file", ".ident", ";")) or "clang version" in t:      # paths, compiler banner, comments
                continue
            keep.append(CUID.sub("__hip_cuid_X", line.split(";")[0].rstrip()))      # the compilation-unit id is a hash of the source PATH
        return hashlib.sha256("\n".join(keep).encode()).hexdigest()[:16], len(keep)


def main():
    import bench
    rev = sys.argv[1]
    write = "--write" in sys.argv
    units = [a for a in sys.argv[2:] if a != "--write"] or [s for s in _build.SOURCES if s.endswith(".hip")]
    with tempfile.TemporaryDirectory() as old:
        tar = subprocess.Popen(["git", "-C", ROOT, "archive", rev, "jrender_amd/csrc", "include"], stdout=subprocess.PIPE)
        subprocess.check_call(["tar", "-x", "-C", old], stdin=tar.stdout)
        tar.wait()
        jobs = [(tree, u) for u in units for tree in (old, ROOT) if os.path.exists(os.path.join(tree, "jrender_amd", "csrc", u))]
        with ThreadPoolExecutor(max_workers=int(os.environ.get("JOBS", "4"))) as ex:
            res = dict(zip(jobs, ex.map(lambda j: device_asm(*j), jobs)))
        old_hash = bench.csrc_hash(old)
        summary = {}
        for u in units:
            a, b = res.get((old, u)), res.get((ROOT, u))
            same = a is not None and b is not None and a[0] == b[0]
            summary[u] = {"same": same, "old": a and a[0], "new": b and b[0], "asm_lines": b and b[1]}
            print("%-32s %s   %s -> %s   (%s lines of device assembly)" % (u, "SAME     " if same else "DIFFERENT", a and a[0], b and b[0], b and b[1]))
    rev_full = subprocess.check_output(["git", "-C", ROOT, "rev-parse", rev], text=True).strip()
    out = {"old_rev": rev_full, "old_csrc_hash": old_hash, "new_csrc_hash": bench.csrc_hash(), "units": summary,
           "how": "tools/isa_same.py: hipcc -S --cuda-device-only with the product's flags on both trees, path-derived lines dropped"}
    print(json.dumps(out))
    if write:
        with open(os.path.join(ROOT, "profiles", "isa_same_latest.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
