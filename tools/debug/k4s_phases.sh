# kernel times of the sparse correlation with phases switched off (debug build): bash tools/debug/k4s_phases.sh [tile bytes]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k4sp
for skip in ${SKIPS:-0 1 3 11 15}; do
  export SN_K4S_SKIP=$skip
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k4sp/p$skip -- python3 tools/debug/k4s_phases.py $1 > gpurun_out/k4sp/run$skip.txt 2>&1
  f=$(find gpurun_out/k4sp/p$skip -name '*kernel_stats.csv' | head -1)
  echo "skip $skip: $(grep -h 'per call' gpurun_out/k4sp/run$skip.txt)"
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'corr' in r['Name']: print('    ', r['Name'][:60], r['Calls'], r['AverageNs'])
"
  rm -rf gpurun_out/k4sp/p$skip
done
