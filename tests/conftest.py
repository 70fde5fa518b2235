import os
import sys

import pytest

# The product's default fp32 matrix arithmetic is the fast tolerance-meeting mode ('f16x3_3'); the suite's "f32" checks were calibrated on the
# exact fp32-input MFMA and name every other mode explicitly, so the DEFAULT is exact here (child processes inherit the variable).
os.environ.setdefault('SIMCLR_DEFAULT_F32_MATMUL', 'exact')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)')
    # The float64 CPU oracle (oracle/model_torch.py) is where the GPU suite spends its wall clock.  On the 128-thread GPU host torch's
    # default (all threads) oversubscribes the 32 ... 64 px convolutions; SIMCLR_ORACLE_THREADS (default 32 on hosts with more cores,
    # measured by tools/oracle_threads_probe.py -> profiles/r05_oracle_threads.json) caps it.  Child processes inherit the variable.
    try:
        import torch
        want = int(os.environ.get('SIMCLR_ORACLE_THREADS', '32'))
        if want > 0 and (os.cpu_count() or 1) > want:
            torch.set_num_threads(want)
            os.environ.setdefault('OMP_NUM_THREADS', str(want))
    except Exception:  # pragma: no cover
        pass


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
