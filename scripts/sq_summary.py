"""Summarise the SQ counter passes of scripts/collect_profiles_r4.sh for k_solve into profiles/<tag>_sq_solve.json.
usage: python scripts/sq_summary.py <out.json> <pass_dir> [<pass_dir> ...]"""
import csv, glob, json, os, sys
out = {}
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if not row["Kernel_Name"].startswith("k_solve"):
                continue
            e = out.setdefault(row["Counter_Name"], {"launches": set(), "sum": 0.0})
            e["launches"].add(row["Dispatch_Id"]); e["sum"] += float(row["Counter_Value"])
doc = {k: {"launches": len(v["launches"]), "per_launch": v["sum"] / max(1, len(v["launches"]))} for k, v in sorted(out.items())}
g = lambda k: doc.get(k, {}).get("per_launch", 0.0)
der = {}
if g("SQ_WAVE_CYCLES"):
    der["wave_parked_share (SQ_WAIT_ANY / SQ_WAVE_CYCLES)"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    der["issue_stall_share (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
    der["issuing_share (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)"] = g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")
if g("SQ_LDS_IDX_ACTIVE"):
    der["lds_bank_conflict_share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
if g("SQ_INSTS_FLAT"):
    der["flat_share_of_memory_instructions"] = g("SQ_INSTS_FLAT") / max(1.0, g("SQ_INSTS_FLAT") + g("SQ_INSTS_LDS"))
json.dump({"kernel": "k_solve", "command": "rocprofv3 --pmc <8 SQ counters per pass> -- python3 bench.py --steps 2 --warmup 1 --no-cpu --exact-sample 0 --handles 1 --closed-loop-steps 0 --no-extra-legs",
           "counters_per_launch": doc, "derived": der,
           "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); instruction counters count wave instructions"}, open(sys.argv[1], "w"), indent=1)
print(json.dumps(der, indent=1))
