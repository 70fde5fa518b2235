#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 200 python -m pytest tests/test_gpu_distributed.py -m gpu -q -k "falls_back" 2>&1 | tail -4 | cut -c1-300
