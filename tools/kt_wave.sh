cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0 1 2 3 7 130 386; do
  JOVA_WAVE_DBG=$d timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_w$d -- python3 $R/tools/one_conv_fwd.py > /dev/null 2>&1 || exit 1
  echo "dbg=$d"; python3 - <<PY
import csv, glob
f = glob.glob('$R/gpurun_out/kt_w$d/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'wave' in r['Name'] : print('   %-60s calls %s avg %.1f us min %.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
