#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_pairx.py -q 2>&1 | tail -12 | grep -h "passed\|failed\|Error\|assert"
