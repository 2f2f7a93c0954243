cd /root/repo
python -m pytest tests/test_gpu_rnn_seq.py -x -q 2>&1 | tail -2
python tools/bench_rnn.py 2>&1 | tail -2
python tools/prof_rnn.py 2>&1 | grep -A15 "lstm fwd"
