cd /root/repo
python -m pytest tests/test_gpu_rnn_seq.py -x -q 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-decode 2>&1 | tail -1 | cut -c100-200
LAS_NO_WARMERS=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-decode 2>&1 | tail -1 | cut -c100-200
done
