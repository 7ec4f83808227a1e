#!/usr/bin/env python3
"""Collapse a rocprofv3 kernel_stats.csv into families (conv / gemm / attention / groupnorm / elementwise / ...)."""
import csv
import re
import sys

FAM = [("gn_fused", r"gn_reduce|gn_apply"), ("conv_igemm", r"igemm|Conv|conv|naive_conv|SubTensorOp|batched_transpose|transpose"),
       ("gemm", r"Cijk|gemm|GEMM"), ("attention", r"attn|fmha|flash|Fmha"), ("layernorm", r"layer_norm|LayerNorm"),
       ("groupnorm_torch", r"RowwiseMoments|GroupNorm|group_norm"), ("softmax", r"softmax"),
       ("elementwise", r"elementwise|vectorized|CatArray|index|copy|fill|upsample")]
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
agg = {}
for r in rows:
    fam = next((f for f, pat in FAM if re.search(pat, r["Name"])), "other")
    a = agg.setdefault(fam, [0.0, 0])
    a[0] += float(r["TotalDurationNs"]); a[1] += int(r["Calls"])
for f, (ns, calls) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-18s %9.2f ms %6.1f%% %7d calls" % (f, ns / 1e6, 100 * ns / tot, calls))
print("total %.2f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%8.2f ms %6d  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), r["Name"][:150]))
