set -e
mkdir -p gpurun_out/bw && export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_flash_attn_bwd_gpu.py -m gpu -x -q > gpurun_out/bw/t.log 2>&1 || { tail -20 gpurun_out/bw/t.log | cut -c1-250; exit 1; }
tail -1 gpurun_out/bw/t.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bw/kt -o c2b -- python3 bench.py --workload c2_bwd --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bw/c2_bwd.json 2> /dev/null
cut -c1-130 gpurun_out/bw/kt/c2b_kernel_stats.csv | head -5
python -c "
import json; d=json.load(open('gpurun_out/bw/c2_bwd.json')); print('c2_bwd', d['value'], 'TF', d['ms_per_step'], 'ms')"
