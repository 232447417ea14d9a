set -e
mkdir -p gpurun_out/r1f && cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r1f/pytest_gpu.log 2>&1
tail -2 gpurun_out/r1f/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/r1f/bench_c2.json 2> gpurun_out/r1f/bench_c2.err
for w in c3 c4 c5; do timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r1f/bench_$w.json 2> gpurun_out/r1f/bench_$w.err; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1f/kt -o c2 -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r1f/kt_bench.json 2> gpurun_out/r1f/kt.err
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/r1f/pmc_a -o c2 -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/r1f/pmc_a.err
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r1f/pmc_b -o c2 -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/r1f/pmc_b.err
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r1f/pmc_c -o c2 -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/r1f/pmc_c.err
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r1f/pmc_d -o c2 -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/r1f/pmc_d.err
python tools/pmc_sum.py gpurun_out/r1f/pmc_a gpurun_out/r1f/pmc_b gpurun_out/r1f/pmc_c gpurun_out/r1f/pmc_d > gpurun_out/r1f/pmc_summary.txt
cat gpurun_out/r1f/bench_c*.json | cut -c1-260
