#!/bin/bash
# One lease of the reproducibility protocol: the DRIVER's exact command for HEAD, for an earlier commit's
# tree (tools/repro/make_tree.sh) and for HEAD again, back to back on one box.   bash tools/repro/run.sh <lease-tag> [tree]
#   -> gpurun_out/repro_<lease-tag>.jsonl : one bench line per run, each tagged {"repro": {"which": ..., "lease": ...}}
TAG=${1:-a}; TREE=${2:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/repro_$TAG.jsonl
mkdir -p $R/gpurun_out; : > $OUT
run() {   # which, dir
  ( cd $2 && python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $R/gpurun_out/repro_${TAG}_$1.err ) | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); d['repro'] = {'which': '$1', 'lease': '$TAG', 'host': open('/etc/hostname').read().strip()}
        print(json.dumps(d))
" >> $OUT
}
run head1 $R
[ -d $R/tools/repro/$TREE ] && run $TREE $R/tools/repro/$TREE
run head2 $R
python3 - $OUT <<'PY'
import sys, json
for l in open(sys.argv[1]):
    d = json.loads(l); b = d.get('box', {})
    print(d['repro']['which'], d['value'], d['ms_per_step'], 'roofline', d.get('roofline', {}).get('frac'),
          'box', {k: (b.get(k) or {}).get('copy_GBps') for k in ('before', 'after')}, {k: (b.get(k) or {}).get('valu_pk_fma_TFLOPps') for k in ('before', 'after')})
PY
