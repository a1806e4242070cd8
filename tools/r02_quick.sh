#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "realtime or mapped" 2>&1 | tail -6
