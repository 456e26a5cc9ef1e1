set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/shard8k
mkdir -p $O
python3 bench.py --rows 8192 --force-sharded --no-cpu-baseline > $O/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --rows 8192 --force-sharded --no-cpu-baseline --steps 50 --warmup 5 > $O/bench_kt.log 2>&1
grep -h '"metric"' $O/bench.log | cut -c1-1500
