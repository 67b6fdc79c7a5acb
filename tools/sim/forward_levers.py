#!/usr/bin/env python3
"""Predicted instruction counts of the two forward levers VERDICT r3 (next #4) asked to model before building:

  (i)  `global_load ... lds` staging of the face records into a second record buffer;
  (ii) a wave-uniform raster loop with neutral operands (no second register set for the per-pixel state).

CPU only: compiles softras_forward.hip to gfx950 assembly (like tools/asm_blocks.py), takes the headline instantiation
k_softras_forward<2,1,16>, finds the blocks of the FAST-face raster loop and of the record staging, and prices them with the
measured issue costs of DESIGN.md 4 (cycles per wavefront-instruction per SIMD: v_mov / add / mul / fma 2.7, other VALU 4.3,
v_rcp / v_exp / v_sqrt 8.3).  The kernel is VALU-issue bound (92 % of the issue slots busy at 20 wavefronts per CU,
profiles/valu_latest.json), so a lever is worth what it removes from the VALU stream - instructions of other classes issue
in the shadow of the other wavefronts.

    python tools/sim/forward_levers.py            -> table on stdout (profiles/r04_forward_levers.txt)
"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "jrender_amd", "csrc", "softras_forward.hip")
OUT = "/tmp/forward_levers.s"
CHEAP = ("v_mov_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32",
         "v_add_u32", "v_sub_u32", "v_and_b32", "v_add_nc_u32")
SLOW = ("v_rcp_f32", "v_exp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32")


def price(op):
    if op.startswith(SLOW):
        return 8.3
    if op.startswith(CHEAP) and "dpp" not in op:
        return 2.7
    return 4.3


def blocks_of(kernel_key):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                    "-x", "hip", "--cuda-device-only", "-S", SRC, "-o", OUT], check=True, stderr=subprocess.DEVNULL)
    lines = open(OUT).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(kernel_key) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], ["entry", []]
    for l in lines[start + 1:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), []]
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur[1].append(l.strip())
    blocks.append(cur)
    return blocks


def census(ins):
    c = {"valu": 0, "v_mov": 0, "v_cndmask": 0, "salu": 0, "lds": 0, "vmem": 0, "cycles": 0.0}
    for i in ins:
        op = i.split()[0]
        if op.startswith("v_") and not op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            c["valu"] += 1
            c["cycles"] += price(op)
            c["v_mov"] += op.startswith("v_mov_b32") and "dpp" not in op
            c["v_cndmask"] += op.startswith("v_cndmask")
        elif op.startswith("s_") or op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
    return c


def main():
    blocks = blocks_of("k_softras_forwardILi2ELi1ELi16ELi5")
    count = lambda ins, key: sum(1 for i in ins if i.startswith(key))        # noqa: E731
    # the raster loop's head: the block that fetches the pixel's next face slot (ds_read) and carries the per-trip copies
    # of the live state (>= 8 plain v_mov); the code is emitted twice (FAST faces / the rest), the first copy is the FAST one
    heads = [k for k, (_, ins) in enumerate(blocks) if count(ins, "v_mov_b32") >= 8 and count(ins, "ds_read") >= 3 and count(ins, "v_cndmask") >= 2]
    assert len(heads) >= 2, "raster loop heads not found: %r" % (heads,)
    lo, hi = heads[0], heads[1]
    loop = [i for _, ins in blocks[lo:hi] for i in ins]
    head = blocks[lo][1]
    shift = [ins for _, ins in blocks[lo:hi] if count(ins, "v_mov_b32") >= 5 and not count(ins, "ds_read") and len(ins) <= 20]
    stage = [ins for _, ins in blocks if count(ins, "ds_write_b128") >= 8 and count(ins, "global_load_dwordx4") >= 8]
    c_loop, c_head = census(loop), census(head)
    print("k_softras_forward<2,1,16>: %d instructions in %d blocks" % (sum(len(b[1]) for b in blocks), len(blocks)))
    print()
    print("raster loop, FAST-face copy (blocks %s .. %s; ALL paths of a trip: inside and outside pixels, append / replace / reject):"
          % (blocks[lo][0].replace(".LBB", ""), blocks[hi - 1][0].replace(".LBB", "")))
    print("   VALU %d  (v_mov %d, v_cndmask %d)   SALU + lane moves %d   LDS %d   VMEM %d   VALU issue cycles %.0f"
          % (c_loop["valu"], c_loop["v_mov"], c_loop["v_cndmask"], c_loop["salu"], c_loop["lds"], c_loop["vmem"], c_loop["cycles"]))
    n_shift = sum(census(s)["v_mov"] for s in shift)
    print("   of the v_mov: %d in the loop head (copies of live state into the registers the divergent body works on), %d in the K-buffer's"
          % (c_head["v_mov"], n_shift))
    print("   append shift (z[k] = z[k-1]: the work itself), %d elsewhere (constants, operands of DPP / readlane)" % (c_loop["v_mov"] - c_head["v_mov"] - n_shift))
    print()
    print("lever (ii), wave-uniform loop with neutral operands:")
    bound = c_head["v_mov"] * 2.7
    print("   removes at most the loop head's %d v_mov = %.0f cycles of %.0f per trip = %.1f %% of the loop's VALU issue time"
          % (c_head["v_mov"], bound, c_loop["cycles"], 100 * bound / c_loop["cycles"]))
    state = 6 + 2      # alpha, softmax sum / max, three colours + the K-buffer's size and cached maximum: updated under a lane mask today
    print("   costs one v_cndmask (4.3 cycles) per state word that a masked-off lane must keep: >= %d words = %.0f cycles" % (state, state * 4.3))
    print("   predicted: %+.1f %% of the loop (between the bound and the bound plus the selects) -> not built"
          % (100 * (state * 4.3 - bound) / c_loop["cycles"]))
    print()
    print("lever (i), global_load ... lds staging:")
    for k, ins in enumerate(stage):
        c = census(ins)
        print("   staging block %d: %d x global_load_dwordx4 + %d x ds_write_b128 per 64-lane chunk of records, VALU %d (addresses), SALU %d"
              % (k, count(ins, "global_load_dwordx4"), count(ins, "ds_write_b128"), c["valu"], c["salu"]))
    print("   direct-to-LDS loads keep the address arithmetic and drop the ds_write_b128: VALU instructions removed: 0.")
    print("   The kernel's VALU issue slots are 92 % busy (profiles/valu_latest.json), LDS instructions are 6.7 % of its VALU count:")
    print("   predicted change of the launch time: 0 ... -1 % (the freed VGPRs do not reach the next occupancy step: 94 -> <= 80 needed) -> not built")


if __name__ == "__main__":
    main()
