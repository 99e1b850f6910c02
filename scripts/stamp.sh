#!/bin/bash
# Run HERE (where .git exists) before a gpurun call that makes profiles: .git is not shipped to the GPU box, so HEAD and
# the commit each kernel source last changed at travel as two small (git-ignored) files that scripts/pmc.sh stamps into
# its summary.
cd "$(dirname "$0")/.."
git rev-parse --short=8 HEAD > .git_head
python - <<'PY'
import json, subprocess
srcs = ["learner.hip", "learner_dp.hip", "learner_io.hip", "learner_env.hip", "learner_internal.hip.h", "gemm_direct.hip.h", "hgemm.hip.h", "small_kernels.hip.h", "env.hip.h"]
kv = {s: subprocess.run(["git", "log", "-1", "--format=%h", "--", "dqn-hfo_amd/csrc/" + s], capture_output=True, text=True).stdout.strip() for s in srcs}
json.dump(kv, open(".kernel_versions.json", "w"))
print(open(".git_head").read().strip(), kv)
PY
