"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel (short names)."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for key in ("k_tree<true, true", "k_embed_pool_c", "k_embed", "k_cls_pool", "k_gather", "k_cls_attn", "k_tail_gemm", "k_ln_heads", "k_ln_rows"):
                if key in n:
                    if key == "k_embed" and "k_embed_pool" in n:
                        continue
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
