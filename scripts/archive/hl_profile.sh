#!/bin/bash
# headline kernel table in one gpurun call: parity tests of the fp32 GEMM path, then rocprofv3 --kernel-trace --stats of bench.py
#   usage: scripts/hl_profile.sh [extra bench args]
cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_gpu_gemm_kernels.py tests/test_gpu_update_parity.py tests/test_gpu_golden.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
out=/tmp/hlprof; rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-env --no-subrecords --no-live-pmc "$@" > gpurun_out/hl.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows[:16]:
    print("%-60s %6s %8.2f" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
grep '^{' gpurun_out/hl.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("under profiler:", d["value"], d["ms_per_step"])'
for i in 1 2 3; do python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-env --no-subrecords --no-live-pmc "$@" 2>/dev/null | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'; done
