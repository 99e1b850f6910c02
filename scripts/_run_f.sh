cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_gpu_optimizer_pass.py tests/test_gpu_golden.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ppA -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-env --no-subrecords --no-live-pmc > /dev/null 2>&1
f=$(find /tmp/ppA -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'adam' in r['Name'] or 'gather' in r['Name']: print(r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3)
PY
for i in 1 2 3; do python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-env --no-subrecords --no-live-pmc 2>/dev/null | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; done
