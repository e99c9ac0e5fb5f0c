cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kats.py -m gpu -x -q 2>&1 | tail -3
bash tools/r03_ab2.sh r03g P:base T:base w6
