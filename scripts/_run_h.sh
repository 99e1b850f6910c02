cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|assert" | tail -3
bash scripts/r03_profiles.sh all > gpurun_out/r03prof_run.log 2>&1; echo "profiles rc=$?"
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
