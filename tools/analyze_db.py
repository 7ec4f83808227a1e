#!/usr/bin/env python3
"""Per-step kernel-family breakdown from a rocprofv3 rocpd database (runs on the GPU box; prints text only).
usage: analyze_db.py results.db [marker_kernel_substring] [top_n]  — steps are delimited by the marker's launches."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "gip_preprocess_kernel"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
marks = [r[0] for r in db.execute("select start from kernels where name like ? order by start", ("%" + marker + "%",))]
lo, hi, nstep = marks[-5], marks[-1], 4
rows = db.execute("select name, count(*), sum(end-start) from kernels where start>=? and start<? group by name", (lo, hi)).fetchall()
# conv3x3_kernel<BN, 2, TAPS = 1, ...> is the MFMA nn.Linear / 1x1 convolution on the convolution kernel: its own family, so that
# "conv_fwd" is the 3x3 convolutions (forward and data gradient, incl. the split-K reduce and the few-channel kernels)
FAM = [("raster(gip)", r"gip_"), ("gn_fused", r"gn_reduce|gn_apply|gn_finalize"), ("conv_bwd", r"igemm_bwd|igemm_wrw|bwd_data|wrw"),
       ("linear(gip mfma)", r"conv3x3_kernelILi\d+ELi2ELi1E"),
       ("conv_fwd", r"igemm_fwd|Conv|conv"), ("miopen_aux", r"SubTensorOp|batched_transpose|transpose"),
       ("gemm", r"Cijk|gemm|GEMM"), ("attention", r"attn|fmha|flash|Fmha"), ("layernorm", r"layer_norm|LayerNorm|layernorm"),
       ("glue(gip)", r"geglu_kernel|cat2_stats|add_bias_residual"), ("winograd transforms(gip)", r"winograd_"),
       ("groupnorm_torch", r"RowwiseMoments|GroupNorm|group_norm"), ("softmax", r"softmax"), ("adam", r"adam|Adam|multi_tensor"),
       ("elementwise", r"elementwise|vectorized|CatArray|index|copy|fill|upsample|reduce")]
tot = sum(r[2] for r in rows)
agg = {}
for n, c, ns in rows:
    fam = next((f for f, pat in FAM if re.search(pat, n)), "other")
    a = agg.setdefault(fam, [0, 0])
    a[0] += ns
    a[1] += c
print("wall per step %.2f ms, GPU busy per step %.2f ms, %d kernels per step" % ((hi - lo) / nstep / 1e6, tot / nstep / 1e6, sum(r[1] for r in rows) // nstep))
for f, (ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-18s %9.3f ms %6.1f%% %7d" % (f, ns / nstep / 1e6, 100 * ns / tot, c // nstep))
for n, c, ns in sorted(rows, key=lambda r: -r[2])[:top]:
    print("%8.3f ms %5d  %s" % (ns / nstep / 1e6, c // nstep, n[:130]))

# ---- idle time: union of the kernel intervals over the steps against the wall time (what stream overlap could still fill)
iv = db.execute("select start, end from kernels where start>=? and start<? order by start", (lo, hi)).fetchall()
busy, cur_s, cur_e = 0, None, None
gaps = []
for s_, e_ in iv:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append(s_ - cur_e)
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
if cur_e is not None:
    busy += cur_e - cur_s
gaps.sort()
ng = len(gaps)
print("union of kernel intervals %.2f ms per step; idle %.2f ms per step in %d gaps (median %.1f us, p90 %.1f us, > 20 us: %.2f ms)" % (
    busy / nstep / 1e6, sum(gaps) / nstep / 1e6, ng // nstep, gaps[ng // 2] / 1e3 if ng else 0, gaps[int(ng * 0.9)] / 1e3 if ng else 0,
    sum(g_ for g_ in gaps if g_ > 20000) / nstep / 1e6))
# the largest gaps and the kernels on either side (host-bound stretches show up here)
named = db.execute("select start, end, name from kernels where start>=? and start<? order by start", (lo, hi)).fetchall()
big, cur_e, prev = [], None, None
for s_, e_, n_ in named:
    if cur_e is not None and s_ > cur_e:
        big.append((s_ - cur_e, prev, n_))
    if cur_e is None or e_ > cur_e:
        cur_e, prev = e_, n_
for g_, a_, b_ in sorted(big, reverse=True)[:16]:
    print("  gap %7.1f us   after %-60s before %s" % (g_ / 1e3, a_[:60], b_[:60]))
