#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for f in "$@"; do
  rm -rf /tmp/mmp
  LELE_HIP_GEMM_FORCE=$f timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/mmp -o t -- python $R/tools/scratch/mm_shapes.py > /tmp/mmp.log 2>&1 || tail -5 /tmp/mmp.log
  echo "== force $f"
  python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/mmp/t_kernel_trace.csv')))
agg = collections.OrderedDict()
for r in rows:
    if 'gemm' not in r['Kernel_Name']: continue
    key = (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['Kernel_Name'][:60])
    agg.setdefault(key, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    v = sorted(v)
    print(k[:3], k[3][-30:], 'n=%d med=%.1f us min=%.1f' % (len(v), v[len(v)//2] / 1e3, v[0] / 1e3))
PY
done
