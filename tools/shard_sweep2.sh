set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/shard_sweep2.log
: > $O
for s in 5 6 7 8 9 10 12; do
echo "== TN 256x4352x8192 splits $s ==" >> $O
python3 tools/gemm_ab.py --form 2 --m 256 --n 4352 --k 8192 --splits $s --tiles 1,20,21 --rounds 5 --positive >> $O 2>&1
done
for s in 2 3 4 5 6; do
echo "== NT 8192x256x4096 splits $s ==" >> $O
python3 tools/gemm_ab.py --form 0 --m 8192 --n 256 --k 4096 --splits $s --tiles 1,20,16,18,5 --rounds 5 --positive >> $O 2>&1
done
grep -v amdgpu.ids $O
