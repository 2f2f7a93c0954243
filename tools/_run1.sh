cd /root/repo
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-decode 2>&1 | tail -1 | cut -c100-200
LAS_NO_TAIL2=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-decode 2>&1 | tail -1 | cut -c100-200
done
