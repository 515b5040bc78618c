#!/bin/bash
# Runs ON THE GPU BOX: per-kernel GPU durations (rocprofv3 --kernel-trace --stats) of an arbitrary command.
# usage: tools/kstats.sh <tag> <command...>      -> gpurun_out/kstats_<tag>/k_kernel_stats.csv + a short table on stdout
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/kstats_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o k -- "$@" > "$OUT/cmd.log" 2>&1 ) || tail -5 "$OUT/cmd.log"
python3 - "$OUT/k_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print('%-110s n=%-5s avg=%8.1f us  min=%8.1f' % (r['Name'][:110], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
