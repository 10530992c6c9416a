#!/bin/bash
# usage (GPU box): tools/ab_window.sh STEPS WARMUP "tune-a" "tune-b" ...   -- steps/s of the headline scene for each tuning set (name=value[,name=value]; "-" = defaults), three runs each
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
steps=$1; warmup=$2; shift 2
for t in "$@"; do
  args=""; if [ "$t" != "-" ]; then for kv in ${t//,/ }; do args="$args --tune $kv"; done; fi
  vals=""
  for r in 1 2 3; do
    v=$(python bench.py --steps $steps --warmup $warmup --no-cpu-baseline --no-dense-pcg --no-other-schedule --no-fast-forward --profile-steps 0 $args 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
    vals="$vals $v"
  done
  echo "$steps+$warmup [$t]:$vals"
done
