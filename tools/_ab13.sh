cd /root/repo
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "loop_shape or persistent" 2>&1 | tail -8
