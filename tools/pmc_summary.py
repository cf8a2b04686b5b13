"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel (short names), as JSON.
usage: pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import json
import re
import sys

KEYS = ("k_tree<true, true, false, false", "k_tree<true, true, false, true", "k_embed_fold", "k_embed_pool_c", "k_embed_pool_x", "k_gemm_x", "k_tail_gemm", "k_tail_lds", "k_move_async")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for key in KEYS:
                if key in n:
                    short = re.sub(r"^void \(anonymous namespace\)::", "", n).split("(")[0] if key.startswith(("k_gemm_x", "k_tail_gemm", "k_tail_lds")) else key
                    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    break
out = {k: dict({c: round(sum(x) / len(x), 1) for c, x in v.items()}, launches=len(next(iter(v.values())))) for k, v in acc.items()}
print(json.dumps(out, indent=1))
