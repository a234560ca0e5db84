#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_dist.py -q -rs 2>&1 | tail -14 | grep -h "passed\|failed\|SKIP\|skipped"
