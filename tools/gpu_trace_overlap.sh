# Concurrency profile of the GRAPH-REPLAYED step with the sub-network forks on and off (tools/trace_overlap.py): kernel trace
# of bench.py's replay loop, last three steps.  Arguments: the settings to run (default "1 0"); one bench run of ~1.5 min each; writes gpurun_out/overlap_forks_{on,off}.{txt,json}.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for forks in ${@:-1 0}; do
  tag=$([ $forks = 1 ] && echo on || echo off)
  rm -rf /tmp/trace_overlap_$tag
  export TD_BRANCH_STREAMS=$forks
  rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_overlap_$tag -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --entry step --miopen-find off > $R/gpurun_out/trace_overlap_$tag.log 2>&1 || { echo "bench failed (forks $tag)"; exit 1; }
  f=$(ls /tmp/trace_overlap_$tag/*/*kernel_trace.csv | head -1)
  python3 $R/tools/trace_overlap.py $f 3 --json $R/gpurun_out/overlap_forks_$tag.json > $R/gpurun_out/overlap_forks_$tag.txt
  echo "== forks $tag"; cat $R/gpurun_out/overlap_forks_$tag.txt
done
