#!/usr/bin/env python3
"""Static view of one kernel's ISA: tools/asm_blocks.py <file.hip> <kernel-name-substring> [--dump LABEL]
Compiles the file to gfx950 assembly and prints per-basic-block instruction counts with markers
(divisions, exp/rcp, DPP, LDS, atomics, scratch), or dumps one block."""
import re, subprocess, sys, os
src, key = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/asm_blocks.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                "-fno-fast-math", "-I" + os.path.join(root, "jrender_amd/csrc"), "-x", "hip",
                "--cuda-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
lines = lines[start:end + 1]
blocks, cur = [], ["entry", []]
for l in lines[1:]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
    elif l.startswith("\t") and not l.strip().startswith((".", ";")):
        cur[1].append(l.strip())
blocks.append(cur)
if "--dump" in sys.argv:
    lab = sys.argv[sys.argv.index("--dump") + 1]
    for name, ins in blocks:
        if name == lab:
            print("\n".join(ins))
    sys.exit(0)
keys = ("v_div_fmas", "v_exp_f32", "v_rcp_f32", "v_add_f32_dpp", "ds_bpermute", "ds_read", "ds_write", "global_atomic",
        "global_load", "global_store", "scratch_", "v_readlane", "v_writelane", "v_mov_b32", "v_cndmask", "s_waitcnt", "s_nop")
total = 0
for name, ins in blocks:
    total += len(ins)
    marks = []
    for k in keys:
        n = sum(1 for i in ins if i.startswith(k))
        if n:
            marks.append("%s:%d" % (k.replace("v_", "").replace("_f32", ""), n))
    br = [i.split()[0][2:] + "->" + i.split()[1].replace(".LBB", "") for i in ins if i.startswith(("s_cbranch", "s_branch"))]
    print("%-11s %4d  %-70s %s" % (name.replace(".LBB", ""), len(ins), " ".join(marks), " ".join(br)))
print("total", total)
