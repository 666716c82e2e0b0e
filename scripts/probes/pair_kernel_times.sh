# per-kernel times of cfg4b / cfg4f with the paired tail on (0) and off (1), resident (bench loop) and cold (--cold), from rocprofv3 kernel stats
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in cfg4b cfg4f; do for k in 0 1; do
  export BSQ_TOKENS_PB8_PAIR=$k
  O=$R/gpurun_out/pair_${w}_$k; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --full-line --workload $w --no-configs --no-cpu-baseline --no-e2e --no-sustained --steps 40 --warmup 10 > /dev/null 2> $O/err
  echo "== $w pair_knob=$k resident"; python3 -c "
import csv,glob,sys
for f in glob.glob('$O/t/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:5]:
        if 'k_' in r['Name'] and 'fill' not in r['Name']: print('   %-60s calls %5s avg %9.2f us' % (r['Name'][28:88], r['Calls'], float(r['AverageNs'])/1e3))
"
  rm -rf $O/t
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --full-line --workload $w --no-configs --no-cpu-baseline --no-e2e --no-sustained --steps 10 --warmup 3 --cold > /dev/null 2>> $O/err
  echo "== $w pair_knob=$k cold-dominated"; python3 -c "
import csv,glob,sys
for f in glob.glob('$O/t/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:5]:
        if 'k_' in r['Name'] and 'fill' not in r['Name']: print('   %-60s calls %5s avg %9.2f us' % (r['Name'][28:88], r['Calls'], float(r['AverageNs'])/1e3))
"
  rm -rf $O/t
done; done
