# Round evidence run (one gpurun call): bench lines for every BASELINE config, rocprofv3 kernel stats + PMC passes for c2 / c3 /
# c4 / c5 (+ kernel stats of the c2 backward), fast-loop cycle stamps, bare-MFMA ceiling, the benchmark grids.
# Usage: bash tools/profile_round.sh r3      (needs tools/bin/libfa_cycles.so and tools/bin/mfma_ceiling, see tools/README.md)
set -e
R=gpurun_out/${1:-r3}_evidence
mkdir -p $R && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py > $R/bench_c2.json 2> $R/bench_c2.err
for w in c3 c4 c5 c2_bwd c3_bwd decode; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $R/bench_$w.json 2> $R/bench_$w.err; done
echo "bench lines done"
for w in c2 c3 c4 c5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt_$w -o $w -- python3 bench.py --workload $w --steps 100 --warmup 20 --no-cpu-baseline --no-check > $R/kt_bench_$w.json 2> $R/kt_$w.err
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt_c2_bwd -o c2_bwd -- python3 bench.py --workload c2_bwd --steps 20 --warmup 5 --no-cpu-baseline > $R/kt_bench_c2_bwd.json 2> $R/kt_c2_bwd.err
echo "kernel traces done"
for w in c2 c3 c4 c5; do
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/pmc_a_$w -o $w -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-check > /dev/null 2> $R/pmc_a_$w.err
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/pmc_b_$w -o $w -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-check > /dev/null 2> $R/pmc_b_$w.err
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc_c_$w -o $w -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-check > /dev/null 2> $R/pmc_c_$w.err
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/pmc_d_$w -o $w -- python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-check > /dev/null 2> $R/pmc_d_$w.err
  python tools/pmc_sum.py $R/pmc_a_$w $R/pmc_b_$w $R/pmc_c_$w $R/pmc_d_$w > $R/pmc_summary_$w.txt
  echo "pmc $w done"
done
# (developer builds, tools/README.md; skipped when they were not built in this container)
if [ -f tools/bin/libfa_cycles.so ]; then FA_FWD_LIB=tools/bin/libfa_cycles.so timeout -k 10 100 python tools/loop_cycles.py 2>&1 | grep -v amdgpu.ids | head -3 > $R/loop_cycles.txt || true; fi
if [ -x tools/bin/mfma_ceiling ]; then timeout -k 10 200 ./tools/bin/mfma_ceiling > $R/ceiling.txt 2>&1 || true; fi
timeout -k 10 300 python tools/fwd_grid.py 0 2>&1 | grep -v amdgpu > $R/fwd_grid.txt
timeout -k 10 300 python tools/hdim_bench.py 2>&1 | grep -v amdgpu > $R/hdim.txt
timeout -k 10 300 python tools/bwd_hdim_bench.py 2>&1 | grep -v amdgpu > $R/bwd_hdim.txt
timeout -k 10 300 python tools/varlen_short_bench.py 2>&1 | grep -v amdgpu > $R/varlen_short.txt
timeout -k 10 300 python tools/window_bench.py 2>&1 | grep -v amdgpu > $R/window_bench.txt
timeout -k 10 300 python tools/dropout_bench.py 2>&1 | grep -v amdgpu > $R/dropout.txt
timeout -k 10 300 python tools/feature_survey.py 2>&1 | grep -v amdgpu > $R/feature_survey.txt
timeout -k 10 300 python tools/inference_survey.py 2>&1 | grep -v amdgpu > $R/inference_survey.txt
timeout -k 10 300 python tools/dv_chunk_bench.py 2>&1 | grep -v amdgpu > $R/dv_chunk.txt
timeout -k 10 300 python tools/persist_sweep.py 2>&1 | grep -v amdgpu > $R/persist_sweep.txt || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt_bwd_d256 -o bwd_d256 -- python3 tools/bwd_hdim_bench.py 256 > $R/kt_bwd_d256.txt 2> $R/kt_bwd_d256.err || true
echo "grids done"
python - <<PY
import json
for w in ("c2", "c3", "c4", "c5", "c2_bwd", "c3_bwd", "decode"):
    d = json.load(open(f"$R/bench_{w}.json"))
    r = d["roofline"]
    print(w, d["value"], d["unit"], r.get("kernel_ms_min", r.get("step_ms_min")), r["frac"], d.get("checked"))
PY
