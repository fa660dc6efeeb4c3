#!/bin/bash
# Same-call sweep of engine grid parameters over bench.py (each honoured override is echoed in the line's "overrides"):
#   bash tools/sweep_env.sh "VAR=v1" "VAR=v2 VAR2=w" ...      -> one summary line per setting (first = defaults)
R=${GRAFT_REPO_ROOT:-$PWD}
run() {
  env $1 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-box 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); kc = d.get('kernel_classes', {})
print('%-40s %8.3f ms  ' % ('$1', d['ms_per_step']) + ' '.join('%s=%.3f' % (k.replace('k_', ''), v['ms_per_step']) for k, v in list(kc.items())[:7]))"
}
run "MNAS_NOP=0"
for S in "$@"; do run "$S"; done
run "MNAS_NOP=0"
