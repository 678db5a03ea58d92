#!/bin/bash
# Build container only (needs /root/reference): regenerate tests/golden/ by running the reference itself and fail if any
# committed fixture changes - the goldens must be exactly what oracle/gen_golden.py and oracle/gen_golden_simulator.py write today (~3 minutes).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
[ -d /root/reference ] || { echo "no /root/reference here: this check runs in the build container"; exit 2; }
git diff --quiet -- tests/golden || { echo "tests/golden has uncommitted changes: commit or stash them first"; exit 2; }
PYTHONPATH=oracle/refshim:/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py > /tmp/gen_golden.log 2>&1 || { tail -20 /tmp/gen_golden.log; exit 1; }
PYTHONPATH=oracle/refshim:/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_simulator.py > /tmp/gen_golden_simulator.log 2>&1 || { tail -20 /tmp/gen_golden_simulator.log; exit 1; }
if git diff --quiet -- tests/golden && [ -z "$(git status --porcelain -- tests/golden)" ]; then
    echo "tests/golden reproduces byte for byte ($(ls tests/golden | wc -l) files)"
else
    git status --short -- tests/golden
    echo "FIXTURES DIFFER from what the reference produces now"; exit 1
fi
