#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_records.py -m gpu -q -k "shipped" 2>&1 | tail -8
