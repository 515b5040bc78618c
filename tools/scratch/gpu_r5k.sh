#!/bin/bash
timeout 900 python -m pytest tests/test_frontend_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
