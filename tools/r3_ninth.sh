#!/bin/bash
set -o pipefail
out=gpurun_out/r3i
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_ccp_step.py tests/test_gpu_association.py -m gpu -q > $out/pytest_ccp.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
grep -E "passed|failed|Error|assert" $out/pytest_ccp.log | tail -20
