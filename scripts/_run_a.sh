set -x
cd /root/repo
python -m pytest tests/test_gpu_gemm_kernels.py tests/test_gpu_update_parity.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/hl_pref -o hl -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-env --no-subrecords --no-live-pmc > gpurun_out/hl_pref.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/hl_pref/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
tail -1 gpurun_out/hl_pref.log | cut -c1-300
for i in 1 2 3; do python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-env --no-subrecords --no-live-pmc 2>/dev/null | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; done
