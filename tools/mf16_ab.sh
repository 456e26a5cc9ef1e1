set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/mf16_ab.log
: > $O
python3 -m pytest tests/test_gpu_gemm.py -q -m gpu -x -k "aligned or ragged or identity" 2>&1 | tail -3 >> $O
echo "== NT Y.D^T 65536x256x4096 (full size): 256x256 tile, 32x32x2 (7) vs 16x16x4 (23) ==" >> $O
python3 tools/gemm_ab.py --form 0 --m 65536 --n 256 --k 4096 --tiles 7,23 --rounds 8 --positive >> $O 2>&1
echo "== TN x^T.Y 256x4096x65536 15 splits: 128x128 (1) vs mf16 (24) ==" >> $O
python3 tools/gemm_ab.py --form 2 --m 256 --n 4096 --k 65536 --splits 15 --tiles 1,24 --rounds 8 --positive >> $O 2>&1
echo "== NT 128 tile: (1) vs (24) at 65536x256x4096 ==" >> $O
python3 tools/gemm_ab.py --form 0 --m 65536 --n 256 --k 4096 --tiles 1,24 --rounds 6 --positive >> $O 2>&1
grep -v amdgpu.ids $O
