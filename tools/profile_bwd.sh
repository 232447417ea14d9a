set -e
mkdir -p gpurun_out/r1b && export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r1b/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/r1b/pytest_gpu.log | cut -c1-250; exit 1; }
tail -1 gpurun_out/r1b/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --workload c2_bwd --no-cpu-baseline > gpurun_out/r1b/bench_c2_bwd.json 2> /dev/null
timeout -k 10 300 python bench.py --workload c3_bwd --steps 10 --no-cpu-baseline > gpurun_out/r1b/bench_c3_bwd.json 2> /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1b/kt -o c2b -- python3 bench.py --workload c2_bwd --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r1b/kt_bench.json 2> /dev/null
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r1b/pmc_a -o c2b -- python3 bench.py --workload c2_bwd --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("gpurun_out/r1b/pmc_a/*counter_collection.csv"):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "bwd_" in k:
            agg[k.split("IDF")[0][-16:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("gpurun_out/r1b/pmc_summary.txt", "w") as f:
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            f.write(f"{k:20s} {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open("gpurun_out/r1b/pmc_summary.txt").read())
PY
python -c "
import json
for w in ('c2_bwd','c3_bwd'):
    d=json.load(open(f'gpurun_out/r1b/bench_{w}.json')); print(w, d['value'], 'TF', d['ms_per_step'], 'ms')"
cut -c1-130 gpurun_out/r1b/kt/c2b_kernel_stats.csv | head -5
