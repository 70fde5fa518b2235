#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02_call25
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "free_proj" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" | cut -c1-250; grep -n "Error\|FAILED\|err=" "$OUT/pytest.log" | head | cut -c1-300
python - <<'PY'
import sys
sys.path.insert(0,'.')
from tests import gpu_checks as gc
for D in (128, 96):
    for b in (8, 16):
        r = gc.check_train_step(depth=18, image_size=32, batch=b, compute_dtype='f32', proj_out_dim=D)
        bad = [(x['name'], '%.2e' % x['err'], '%.2e' % x['tol']) for x in r if not x['ok']]
        med = [(x['name'].split()[0], '%.2e' % x['err'], '%.2e' % x['tol']) for x in r if 'new_params_rel_median' in x['name'] or 'grad_relnorm' in x['name']]
        print('D', D, 'batch', b, 'bad', bad, med)
PY
