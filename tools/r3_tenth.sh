#!/bin/bash
set -o pipefail
out=gpurun_out/r3j
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
grep -E "passed|failed|Error|assert" $out/pytest.log | tail -20
