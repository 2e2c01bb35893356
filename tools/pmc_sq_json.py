#!/usr/bin/env python3
"""Reduce tools/pmc_summary.py output (SQ counter passes of `bench.py --fast`) to profiles/latest_sq.json: the scorer kernel's
wave-instruction counts per launch, which bench.py turns into an issue fraction next to its live kernel time (and labels as stored).
Usage: pmc_sq_json.py pmc_sq.txt out.json [shader_clock_GHz]"""
import json, re, sys
rows = {}
for ln in open(sys.argv[1]):
    m = re.match(r"^(.*?)\s+(SQ_\w+|GRBM_\w+)\s+avg\s+([\d.]+) over (\d+)", ln)
    if m and "k2b_dp" in m.group(1):
        rows.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(3))
name, c = max(rows.items(), key=lambda kv: kv[1].get("SQ_INSTS_VALU", 0))
kern = re.search(r"(k2b_dp\w*<[^>]*>?)", name)
out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES / SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES / SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU "
                 "(separate passes, counters only) -- python bench.py --fast --steps 5 --warmup 2, MI355X; reduced by tools/pmc_summary.py + tools/pmc_sq_json.py",
       "scorer_kernel": (kern.group(1) if kern else name)[:60],
       "scorer_valu_wave_instructions_per_launch": c.get("SQ_INSTS_VALU"),
       "scorer_salu_wave_instructions_per_launch": c.get("SQ_INSTS_SALU"),
       "scorer_all_wave_instructions_per_launch": c.get("SQ_ACTIVE_INST_ANY"),
       "scorer_waves": c.get("SQ_WAVES"), "scorer_wave_quad_cycles": c.get("SQ_WAVE_CYCLES"),
       "scorer_wait_any_quad_cycles": c.get("SQ_WAIT_ANY"), "scorer_wait_inst_any_quad_cycles": c.get("SQ_WAIT_INST_ANY"),
       "shader_clock_GHz": float(sys.argv[3]) if len(sys.argv) > 3 else 2.25}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
