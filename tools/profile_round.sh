# Round evidence run (one gpurun call): bench lines, rocprofv3 kernel stats + PMC passes for the bf16 headline (c2) and the
# native fp8 workload (c5), fast-loop cycle stamps, bare-MFMA ceiling.  Usage: bash tools/profile_round.sh r2
set -e
R=gpurun_out/${1:-r2}_evidence
mkdir -p $R && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py > $R/bench_c2.json 2> $R/bench_c2.err
for w in c3 c4 c5 c2_bwd c3_bwd decode; do timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline > $R/bench_$w.json 2> $R/bench_$w.err; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -o c2 -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/kt_bench_c2.json 2> $R/kt.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt5 -o c5 -- python3 bench.py --workload c5 --steps 100 --warmup 20 --no-cpu-baseline > $R/kt_bench_c5.json 2> $R/kt5.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/ktb -o c2_bwd -- python3 bench.py --workload c2_bwd --steps 20 --warmup 5 --no-cpu-baseline > $R/kt_bench_c2_bwd.json 2> $R/ktb.err
for w in c2 c5; do
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/pmc_a_$w -o $w -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/pmc_a_$w.err
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/pmc_b_$w -o $w -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/pmc_b_$w.err
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc_c_$w -o $w -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/pmc_c_$w.err
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/pmc_d_$w -o $w -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/pmc_d_$w.err
  python tools/pmc_sum.py $R/pmc_a_$w $R/pmc_b_$w $R/pmc_c_$w $R/pmc_d_$w > $R/pmc_summary_$w.txt
done
FA_FWD_LIB=tools/bin/libfa_cycles.so timeout -k 10 100 python tools/loop_cycles.py 2>&1 | grep -v amdgpu.ids | head -2 > $R/loop_cycles.txt
timeout -k 10 200 ./tools/bin/mfma_ceiling > $R/ceiling.txt 2>&1
timeout -k 10 300 python tools/fwd_grid.py 0 2>&1 | grep -v amdgpu > $R/fwd_grid.txt
python - <<PY
import json
for w in ("c2", "c3", "c4", "c5", "c2_bwd", "c3_bwd", "decode"):
    d = json.load(open(f"$R/bench_{w}.json"))
    r = d["roofline"]
    print(w, d["value"], d["unit"], r.get("kernel_ms_min", r.get("step_ms_min")), r["frac"])
PY
