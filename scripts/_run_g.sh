cd /root/repo
python -m pytest tests/test_gpu_update_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_env.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|error|assert" | tail -3
bash scripts/ab_libs.sh 2 | cut -c1-30
