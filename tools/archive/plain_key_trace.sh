# kernel timeline of one MSM on a plain key (no pre-shifted tables) beside one on a pre-shifted key: rocprofv3 --kernel-trace of tools/time_plain_key.py
cd /tmp && export TMPDIR=/tmp
for l in 12 20; do
  rm -rf /tmp/pk; rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -- python $GRAFT_REPO_ROOT/tools/time_plain_key.py $l > /tmp/pk.log 2>&1
  tail -1 /tmp/pk.log
  cp /tmp/pk/*/*kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/plain_trace_$l.csv
done
