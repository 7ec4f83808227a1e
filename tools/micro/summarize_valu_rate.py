#!/usr/bin/env python3
"""valu_rate's JSON lines -> profiles/r03_valu_rate.json (what bench.py's roofline_valu prices instructions with).
usage: summarize_valu_rate.py gpurun_out/valu_rate.txt profiles/r03_valu_rate.json"""
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
out = {"_source": "tools/micro/valu_rate.hip on 1x MI355X: cycles one SIMD needs per wave64 instruction of an independent stream; "
                  "1 wave: from the wave's own s_memtime stamps; >= 2 waves: from the launch's wall time x the measured clock "
                  "(the waves of a launch do not all overlap, so per-wave stamps read low)",
       "_rows": rows}
for name in sorted({r["instr"] for r in rows}):
    by = {r["waves_per_simd_launched"]: r for r in rows if r["instr"] == name}
    out[name] = {"simd_cycles_1_wave": round(by[1]["cycles_per_instr_per_wave"], 2),
                 "simd_cycles_2_waves": round(by[2]["simd_cycles_per_wave_instr_from_wall"], 2),
                 "simd_cycles_ge4_waves": round(min(by[4]["simd_cycles_per_wave_instr_from_wall"], by[8]["simd_cycles_per_wave_instr_from_wall"]), 2),
                 "clock_ghz": round(by[4]["clock_ghz"], 2)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
