mkdir -p gpurun_out
for st in ops fwd fwdbwd full; do
  echo "=== $st" >> gpurun_out/dbg.log
  timeout -k 10 300 python tools/debug_graph.py $st >> gpurun_out/dbg.log 2>&1
  echo "rc=$?" >> gpurun_out/dbg.log
done
grep -v "amdgpu.ids" gpurun_out/dbg.log | tail -60
