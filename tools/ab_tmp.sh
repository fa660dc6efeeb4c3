#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5
python3 tools/soak.py 80 256 se 2>&1 | tail -3
