#!/bin/bash
# Profiles of the headline run, committed under profiles/ (run through gpurun from the repo root):
#   bash tools/profiles.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of `python bench.py --no-subrecords --no-cpu-baseline` (the headline alone: the
#    default command adds the sub-records' workloads to the same process)
# 2. PMC passes (one counter group per run, never combined with trace domains) of a one-chunk slice of the headline corpus
#    (250k docs = 1 GiB = one launch of every kernel, as in the headline's chunks) and of configs[1]
set +e
tag=$1
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ ! -s $out/bench_under_rocprof.json ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/bench.py --no-subrecords --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.log
fi
cd $root
cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
echo "kernel trace done"
bash tools/pmc.sh ${tag}_cfg3 --workload cfg3 --docs 250000 > $out/pmc_cfg3.log 2>&1
echo "pmc cfg3 done"
bash tools/pmc.sh ${tag}_cfg2 --workload cfg2 > $out/pmc_cfg2.log 2>&1
cp gpurun_out/pmc_${tag}_cfg3/summary.csv $out/pmc_cfg3_per_kernel.csv
cp gpurun_out/pmc_${tag}_cfg2/summary.csv $out/pmc_cfg2_per_kernel.csv
python3 - <<PY
import csv, json
res = {"build": "$tag", "note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md) and WRITE_SIZE, separate passes; cfg3 = one 1 GiB chunk of the headline corpus (250k docs), cfg2 = configs[1] (one chunk)", "workloads": {}}
for wl in ("cfg3", "cfg2"):
    ks = {}
    for r in csv.DictReader(open("$out/pmc_%s_per_kernel.csv" % wl)):
        if r.get("hbm_read_MB_corrected") and r.get("hbm_write_MB"):
            ks[r["kernel"]] = {"hbm_read_MB": float(r["hbm_read_MB_corrected"]), "hbm_write_MB": float(r["hbm_write_MB"])}
    res["workloads"][wl] = {"kernels": ks, "total_MB": round(sum(v["hbm_read_MB"] + v["hbm_write_MB"] for v in ks.values()), 1)}
json.dump(res, open("$out/pmc_traffic.json", "w"), indent=1)
print(json.dumps({w: v["total_MB"] for w, v in res["workloads"].items()}))
PY
head -12 $out/kernel_stats.csv | cut -c1-160
