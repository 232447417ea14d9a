# Round-end evidence run (one gpurun call): full GPU suite, smoke, bench lines, rocprofv3 kernel stats + PMC passes.
set -e
R=gpurun_out/r1f
mkdir -p $R && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $R/pytest_gpu.log 2>&1 || { tail -20 $R/pytest_gpu.log | cut -c1-250; exit 1; }
tail -1 $R/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py > $R/bench_c2.json 2> $R/bench_c2.err
for w in c3 c4 c5 c2_bwd c3_bwd decode; do timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline > $R/bench_$w.json 2> $R/bench_$w.err; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -o c2 -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/kt_bench.json 2> $R/kt.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt_bwd -o c2b -- python3 bench.py --workload c2_bwd --steps 10 --warmup 2 --no-cpu-baseline > $R/kt_bwd_bench.json 2> $R/kt_bwd.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt_dec -o dec -- python3 bench.py --workload decode --steps 20 --warmup 3 > $R/kt_dec_bench.json 2> $R/kt_dec.err
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/pmc_a -o c2 -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/pmc_a.err
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES --output-format csv -d $R/pmc_b -o c2 -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/pmc_b.err
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc_c -o c2 -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/pmc_c.err
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/pmc_d -o c2 -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> $R/pmc_d.err
python tools/pmc_sum.py $R/pmc_a $R/pmc_b $R/pmc_c $R/pmc_d > $R/pmc_summary.txt
timeout -k 10 300 python tools/fwd_grid.py 0 2>&1 | grep -v amdgpu > $R/fwd_grid.txt
timeout -k 10 200 python tools/d256_bench.py 2>&1 | grep -v amdgpu > $R/d256.txt
timeout -k 10 200 python tools/decode_sweep.py 2>&1 | grep -v amdgpu > $R/decode_sweep.txt
timeout -k 10 200 python tools/dropout_bench.py 2>&1 | grep -v amdgpu > $R/dropout.txt
python - <<'PY'
import json
for w in ("c2", "c3", "c4", "c5", "c2_bwd", "c3_bwd", "decode"):
    d = json.load(open(f"gpurun_out/r1f/bench_{w}.json"))
    r = d["roofline"]
    print(w, d["value"], d["unit"], r.get("kernel_ms_min", r.get("step_ms_min")), r["frac"])
PY
