cd /root/repo
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode 2>&1 | tail -1 | cut -c1-330
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
