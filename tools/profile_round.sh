#!/bin/bash
# Profiles committed under profiles/<label>/: rocprofv3 kernel-trace statistics of the
# default bench.py run, and HBM traffic counters of the dominant kernel (two separate
# counter passes, as the TCC block cannot hold FETCH_SIZE and WRITE_SIZE together).
# usage (on the GPU box): tools/profile_round.sh <label>
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${1:-r01}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/ktrace.log || { tail -5 $OUT/ktrace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.log || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_write.json 2> $OUT/pmc_write.log || { tail -5 $OUT/pmc_write.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_sq.json 2> $OUT/pmc_sq.log || { tail -5 $OUT/pmc_sq.log; exit 1; }
# vector instructions by class (what tools/issue_account.py prices), and where waves wait
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_class -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_class.log || { tail -5 $OUT/pmc_class.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_wait -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_wait.log || { tail -5 $OUT/pmc_wait.log; exit 1; }
python3 - <<PY
import csv, glob, json, collections, os
out = "$OUT"
def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(n[k]) for c, v in agg[k].items()} | {"dispatches": len(n[k])} for k in agg}
res = {"fetch": counters("pmc_fetch"), "write": counters("pmc_write"), "sq": counters("pmc_sq"),
       "class": counters("pmc_class"), "wait": counters("pmc_wait")}
json.dump(res, open(os.path.join(out, "pmc_per_launch.json"), "w"), indent=1, sort_keys=True)
traffic = {}
for k in res["fetch"]:
    f = res["fetch"][k].get("FETCH_SIZE", 0.0); w = res["write"].get(k, {}).get("WRITE_SIZE", 0.0)
    # both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes
    traffic[k.split("::")[-1].split("<")[0] + "_hbm_bytes_per_launch"] = 2.0 * f * 1024 + w * 1024
for k, v in res["sq"].items():
    if "SQ_INSTS_VALU" in v:  # wave-level vector instructions of one launch
        traffic[k.split("::")[-1].split("<")[0] + "_valu_instructions_per_launch"] = v["SQ_INSTS_VALU"]
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1, sort_keys=True)
for f in glob.glob(os.path.join(out, "ktrace", "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read())
print(json.dumps(traffic, indent=1))
PY
