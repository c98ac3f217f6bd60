#!/bin/bash
# tools/poison_check.py over its model configurations: no NaN gradient with every engine buffer NaN-filled at allocation, repeated
# steps bit-identical under any allocation pattern
cd "$GRAFT_REPO_ROOT"
for c in 0 1 2 3 4 5 6; do CASE=$c POISON_LDS=${POISON_LDS:-0} python tools/poison_check.py nan 96 2>&1 | grep -E "==|LDS|first step|differ|Error|error" | cut -c1-260; done
