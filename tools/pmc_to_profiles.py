"""gpurun_out/r04_pmc/pmc_merged.json (tools/profile_pmc.sh) -> profiles/r04_pmc_traffic.json + profiles/r04_pmc_sq_counters.json.
Unit and gfx950 corrections as MI355X_MICROARCH.md's HBM section prescribes; the `_how` strings say what was applied.
usage: python tools/pmc_to_profiles.py [merged.json] [out_dir]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04_pmc", "pmc_merged.json")
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
m = json.load(open(src))
HOW = ("rocprofv3 --kernel-trace --pmc <counters of one pass> --kernel-include-regex <ONE kernel family> --output-format csv -- python3 bench.py "
       "--steps 1 --warmup 1 --preroll-cheap 32 --preroll-full 2 --cpu-seconds 0 --fp32-steps 0 (tools/profile_pmc.sh: 5 counter sets x 4 kernel "
       "families = 20 passes, each run ONCE, all rc 0 - r04_pmc_abort_diagnosis.md says why one family per pass); means over the 3 712 launches of each "
       "kernel in a run (3 200 of them full 800-simulation searches on de-phased games, 512 from the cheap pre-roll). Units: rocprofv3 reports KB; "
       "bytes = KB * 1024. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide 16-B/lane streaming reads by 2x - applied (x2) "
       "to k_tail_lds, whose reads are all 16-B/lane LDS-DMA pieces; k_tree's reads are 8-16 B per lane gathers and k_embed_fold's byte loads and "
       "L2-resident table gathers (uncalibrated widths, taken as reported); WRITE_SIZE is exact for 16-B/lane stores.")
traffic, sq = {"_how": HOW}, {"_how": HOW + "  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles."}
for k, v in m.items():
    corr = 2.0 if k.startswith("k_tail_lds") else 1.0
    hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
    traffic[k] = {"fetch_kb": round(v["FETCH_SIZE"] * corr, 1), "write_kb": round(v["WRITE_SIZE"], 1), "fetch_correction": corr,
                  "bytes_per_launch": int(round((v["FETCH_SIZE"] * corr + v["WRITE_SIZE"]) * 1024)), "tcc_hit": round(hit, 1), "tcc_miss": round(miss, 1),
                  "l2_hit_rate": round(hit / max(hit + miss, 1.0), 3), "launches": v["launches"]}
    c = {n: round(x, 1) for n, x in v.items() if n.startswith("SQ_")}
    wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    c["derived"] = {"wait_any_share": round(v.get("SQ_WAIT_ANY", 0.0) / wc, 3), "issue_stall_share": round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                    "active_share": round(v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3),
                    "valu_per_mfma": round(v.get("SQ_INSTS_VALU", 0.0) / v["SQ_INSTS_MFMA"], 1) if v.get("SQ_INSTS_MFMA") else None,
                    "salu_per_valu": round(v.get("SQ_INSTS_SALU", 0.0) / max(v.get("SQ_INSTS_VALU", 0.0), 1.0), 2),
                    "lds_bank_conflict_cycles": round(v.get("SQ_LDS_BANK_CONFLICT", 0.0), 1)}
    sq[k] = c
json.dump(traffic, open(os.path.join(out, "r04_pmc_traffic.json"), "w"), indent=1)
json.dump(sq, open(os.path.join(out, "r04_pmc_sq_counters.json"), "w"), indent=1)
print(json.dumps({k: (v["bytes_per_launch"], v["l2_hit_rate"]) for k, v in traffic.items() if k != "_how"}))
