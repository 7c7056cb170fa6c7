# kernel-level A/B of one environment switch under rocprofv3 (same box): bash tools/prof_ab.sh VAR "0 1" [grep pattern]
# prints the total kernel time per step and the lines of the kernels matching the pattern for each value
V=$1; VALS=$2; PAT=${3:-ln_bwd}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for x in $VALS; do
  rm -rf /tmp/pf_ab_$x
  env $V=$x rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_ab_$x -o st -- python bench.py --lean --steps 20 > /tmp/pf_ab_$x.log 2>&1
  f=$(find /tmp/pf_ab_$x -name "*kernel_stats.csv" | head -1)
  echo "== $V=$x"
  python tools/kstats.py $f 25 400 | grep -E "total kernel time|$PAT"
done
