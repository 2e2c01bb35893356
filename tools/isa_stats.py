#!/usr/bin/env python3
"""Static instruction statistics of a gfx950 kernel from hipcc's assembly listing (no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude --cuda-device-only -S -o k.s frizbee_amd/csrc/kernels_dp.hip
    python tools/isa_stats.py k.s k2b_dp [substring ...]

Per matching kernel: VGPR / SGPR / scratch / occupancy from the metadata, the instruction mix, and every basic block with its
instruction count and whether a later branch jumps back to it (a loop head).  For a VALU-issue-bound kernel (k2b_dp) the VALU count
of the row loop's blocks IS the cost model: ~4 cycles per wave64 VALU instruction (tools/ubench/valu_rate.hip)."""
import re
import sys


def classify(op):
    if op.startswith(("v_cmp", "v_")):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", lines[i])
        if not m or not all(p in m.group(1) for p in pats):
            i += 1
            continue
        name = m.group(1)
        j = i + 1
        blocks = [["entry", {}]]
        order = {"entry": 0}
        backs = set()
        loops = []  # (first block, last block) of every backward branch: the static extent of a loop
        while j < len(lines) and not lines[j].startswith("\t.section") and not re.match(r"^\s*\.end_amdhsa_kernel", lines[j]):
            ln = lines[j]
            lab = re.match(r"^(\.LBB\d+_\d+):", ln)
            if lab:
                order[lab.group(1)] = len(blocks)
                blocks.append([lab.group(1), {}])
            else:
                ins = re.match(r"^\t([a-z_0-9]+)\s*(.*)$", ln)
                if ins and not ins.group(1).startswith("."):
                    op = ins.group(1)
                    k = classify(op)
                    blocks[-1][1][k] = blocks[-1][1].get(k, 0) + 1
                    if k == "branch":
                        t = re.search(r"(\.LBB\d+_\d+)", ins.group(2))
                        if t and t.group(1) in order:
                            backs.add(t.group(1))
                            loops.append((order[t.group(1)], len(blocks) - 1))
            j += 1
        meta = {}
        for k in range(j, min(j + 400, len(lines))):
            mm = re.match(r"^; (NumVgprs|NumAgprs|TotalNumVgprs|NumSgprs|ScratchSize|Occupancy|codeLenInByte): (\d+)", lines[k].replace(" [waves/SIMD]", ""))
            if mm:
                meta[mm.group(1)] = int(mm.group(2))
            if lines[k].startswith("_Z") and k > j + 5:
                break
        tot = {}
        for _, c in blocks:
            for k, v in c.items():
                tot[k] = tot.get(k, 0) + v
        print(f"== {name}\n   {meta}\n   total {tot}")
        for lab, c in blocks:
            n = sum(c.values())
            if n >= 24 or lab in backs:
                print(f"   {lab:12s} {'LOOP' if lab in backs else '    '} n={n:5d} {c}")
        for a, b in sorted(set(loops)):
            if b - a >= 1:
                ext = {}
                for _, c in blocks[a:b + 1]:
                    for k, v in c.items():
                        ext[k] = ext.get(k, 0) + v
                print(f"   loop extent {blocks[a][0]} .. {blocks[b][0]} ({b - a + 1} blocks, every conditional block counted): {ext}")
        i = j


if __name__ == "__main__":
    main()
