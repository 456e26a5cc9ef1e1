set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/sharddump
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/shard_trace.py > $O/run.log 2>&1
F=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python3 tools/trace_dump.py $F 20 40 > $O/dump.txt
cat $O/dump.txt
