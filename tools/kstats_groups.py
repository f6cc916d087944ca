#!/usr/bin/env python3
"""Group a rocprofv3 kernel_stats CSV into categories (time share of a training step)."""
import csv
import re
import sys

GROUPS = [
    ("hip kernels (td::)", r"(^|[^s])td::"),
    ("conv (MIOpen igemm/winograd/gemm)", r"igemm|Cijk|miopen.*[Cc]onv|gridwise|naive_conv|Winograd|winograd|sp3|gfx9.*conv|conv_|kernel_gemm|batched_transpose"),
    ("MIOpen tensor ops (cast/set/add)", r"SubTensorOp|Op1dTensor|Op2dTensor|Op4dTensor|OpTensor"),
    ("batchnorm", r"BatchNorm|batch_norm"),
    ("max_pool", r"max_pool"),
    ("reflection_pad", r"reflection_pad"),
    ("grid_sampler", r"grid_sampler"),
    ("upsample/interp", r"upsample|interp"),
    ("optimizer/foreach", r"multi_tensor|foreach|Adam|adam"),
    ("reduce", r"reduce_kernel"),
    ("cat/copy", r"CatArray|copy_kernel|direct_copy"),
    ("elementwise", r"elementwise|vectorized"),
]
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
acc = {g: 0.0 for g, _ in GROUPS}
acc["other"] = 0.0
other = []
for r in rows:
    t = float(r["TotalDurationNs"])
    for g, pat in GROUPS:
        if re.search(pat, r["Name"]):
            acc[g] += t
            break
    else:
        acc["other"] += t
        other.append((t, r["Name"][:90]))
print("total %.2f ms over %g steps -> %.2f ms/step" % (tot / 1e6, steps, tot / 1e6 / steps))
for g, t in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-40s %8.2f ms/step  %5.1f%%" % (g, t / 1e6 / steps, 100 * t / tot))
for t, n in sorted(other, reverse=True)[:8]:
    print("     other: %7.2f ms  %s" % (t / 1e6 / steps, n))
