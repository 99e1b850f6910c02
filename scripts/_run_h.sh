cd /root/repo
bash scripts/r03_profiles.sh all > gpurun_out/r03prof_run.log 2>&1; echo "profiles rc=$?"
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
