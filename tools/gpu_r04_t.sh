#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python tools/probes/exchange_noise.py bf16 > /tmp/xn.log 2>&1; grep -E "vs plain|Error|error" /tmp/xn.log | cut -c1-600; tail -3 /tmp/xn.log | cut -c1-300
