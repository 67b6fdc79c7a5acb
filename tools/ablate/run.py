#!/usr/bin/env python3
"""GPU box: parity + timing of ablation variants, interleaved `rounds` times.
usage: python tools/ablate/run.py [--rounds 2] [--no-parity | --quick-parity] [names...]   -> table on stdout (+ JSON lines)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from variants import VARIANTS           # noqa: E402

args = sys.argv[1:]
rounds = int(args[args.index("--rounds") + 1]) if "--rounds" in args else 2
parity = "--no-parity" not in args
extra = args[args.index("--bench-args") + 1].split() if "--bench-args" in args else []
skip = {args[i + 1] for i, a in enumerate(args) if a in ("--rounds", "--bench-args")}
names = [a for a in args if not a.startswith("--") and a not in skip] or list(VARIANTS)


def lib(n):
    return os.path.join(ROOT, "jrender_amd", "csrc", "libjrender_hip.so" if n == "product" else "libjrender_hip_%s.so" % n)


res = {n: [] for n in names}
status = {}
for n in names:
    if not os.path.exists(lib(n)):
        status[n] = "missing"
        continue
    if parity:
        # every variant is claimed exactness-equivalent (csrc/jr_tuning.h): the curated parity suite on each, plus the
        # full-size suite unless --quick-parity (the product build gets the whole `pytest -m gpu` run anyway)
        files = ["tests/test_gpu_parity.py"] + ([] if "--quick-parity" in args else ["tests/test_gpu_fullsize.py"])
        p = subprocess.run([sys.executable, "-m", "pytest", *files, "-x", "-q", "-m", "gpu"],
                           cwd=ROOT, env=dict(os.environ, JRENDER_LIB=lib(n)), capture_output=True, text=True)
        status[n] = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "no output"
        if p.returncode:
            print("PARITY FAIL", n, "\n", p.stdout[-3000:], flush=True)
    else:
        status[n] = "parity not run"
for r in range(rounds):
    for n in names:
        if status.get(n) == "missing":
            continue
        p = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-secondary"] + extra, cwd=ROOT,
                           env=dict(os.environ, JRENDER_LIB=lib(n)), capture_output=True, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
            res[n].append((d["phase_ms_per_step"]["fwd_raster"], d["phase_ms_per_step"]["bwd_raster"], d["step_ms"]["median"]))
        except Exception as e:
            print("bench failed", n, e, p.stderr[-500:])
print("%-10s %9s %9s %9s   %s" % ("variant", "fwd ms", "bwd ms", "step ms", "parity"))
for n in names:
    if res[n]:
        best = [min(x[i] for x in res[n]) for i in range(3)]
        print("%-10s %9.4f %9.4f %9.4f   %s" % (n, best[0], best[1], best[2], status[n]))
        print(json.dumps({"variant": n, "defines": VARIANTS[n], "runs": res[n], "parity": status[n]}))
    else:
        print("%-10s %s" % (n, status.get(n)))
