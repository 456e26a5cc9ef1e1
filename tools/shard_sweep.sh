# Tile / split sweep of the products of ONE 8192-row shard of configs[1] (run on the GPU box).
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/shard_sweep.log
: > $O
echo "== NT Y.D^T 8192x256x4096, no split ==" >> $O
python3 tools/gemm_ab.py --form 0 --m 8192 --n 256 --k 4096 --splits 1 --tiles 15,16,19,20,21,22,2,5 --rounds 6 --positive >> $O 2>&1
echo "== NT Y.D^T 8192x256x4096, 8 splits (slabs + reduce) ==" >> $O
python3 tools/gemm_ab.py --form 0 --m 8192 --n 256 --k 4096 --splits 8 --tiles 7,1,20 --rounds 6 --positive >> $O 2>&1
echo "== NT 4 splits ==" >> $O
python3 tools/gemm_ab.py --form 0 --m 8192 --n 256 --k 4096 --splits 4 --tiles 1,20,21,5 --rounds 6 --positive >> $O 2>&1
echo "== TN x^T.[Y] 256x4096x8192, 15 splits ==" >> $O
python3 tools/gemm_ab.py --form 2 --m 256 --n 4096 --k 8192 --splits 15 --tiles 1,7,3,20 --rounds 6 --positive >> $O 2>&1
echo "== TN 8 splits ==" >> $O
python3 tools/gemm_ab.py --form 2 --m 256 --n 4096 --k 8192 --splits 8 --tiles 1,7,20,18 --rounds 6 --positive >> $O 2>&1
echo "== NN S.D 256x4096x256 ==" >> $O
python3 tools/gemm_ab.py --form 1 --m 256 --n 4096 --k 256 --splits 1 --tiles 2,17,18,16 --rounds 6 --positive >> $O 2>&1
echo "== NT gram D.D^T 256x256x4096, 16 splits ==" >> $O
python3 tools/gemm_ab.py --form 0 --m 256 --n 256 --k 4096 --splits 16 --tiles 2,17,18 --rounds 6 --positive >> $O 2>&1
echo "== NT gram 32 splits ==" >> $O
python3 tools/gemm_ab.py --form 0 --m 256 --n 256 --k 4096 --splits 32 --tiles 2,17,18 --rounds 6 --positive >> $O 2>&1
echo "== NN x.G 8192x256x256 ==" >> $O
python3 tools/gemm_ab.py --form 1 --m 8192 --n 256 --k 256 --splits 1 --tiles 2,17,18,16,15 --rounds 6 --positive >> $O 2>&1
cat $O
