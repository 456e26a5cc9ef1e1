set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/ahead_ab.log
: > $O
python3 -m pytest tests/test_gpu_gemm.py -q -m gpu -x -k "aligned or ragged" 2>&1 | tail -2 >> $O
echo "== NT Y.D^T 65536x256x4096: tile 7 (baseline) vs 25 (reads ahead) ==" >> $O
python3 tools/gemm_ab.py --form 0 --m 65536 --n 256 --k 4096 --tiles 7,25 --rounds 10 --positive >> $O 2>&1
echo "== TN 256x4096x65536 15 splits: 1 vs 26 ==" >> $O
python3 tools/gemm_ab.py --form 2 --m 256 --n 4096 --k 65536 --splits 15 --tiles 1,26 --rounds 10 --positive >> $O 2>&1
echo "== NN x.G 65536x256x256: 7 vs 25 ==" >> $O
python3 tools/gemm_ab.py --form 1 --m 65536 --n 256 --k 256 --tiles 7,25 --rounds 10 --positive >> $O 2>&1
echo "== shard NT 8192x256x4096 8 splits: 7 vs 25 ==" >> $O
python3 tools/gemm_ab.py --form 0 --m 8192 --n 256 --k 4096 --splits 8 --tiles 7,25 --rounds 10 --positive >> $O 2>&1
echo "== shard TN 256x4096x8192 15 splits: 1 vs 26 ==" >> $O
python3 tools/gemm_ab.py --form 2 --m 256 --n 4096 --k 8192 --splits 15 --tiles 1,26 --rounds 10 --positive >> $O 2>&1
grep -v amdgpu.ids $O
