cd /root/repo
python -m pytest tests/test_gpu_rnn_seq.py tests/test_gpu_speller_bf16.py tests/test_gpu_las_parity.py -x -q -k "rows16 or shadows or reuses" 2>&1 | tail -15
