cd /root/repo
python -m pytest tests/test_gpu_rnn_seq.py -x -q 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode 2>&1 | tail -1 | cut -c1-330
