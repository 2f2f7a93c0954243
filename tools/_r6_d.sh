#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
P=r6d
LAS_PARITY_LOG=$PWD/gpurun_out/${P}_parity.jsonl timeout 2400 python3 -m pytest tests -m gpu -q -rs 2>&1 | grep -v amdgpu.ids > gpurun_out/${P}_pytest.log
tail -15 gpurun_out/${P}_pytest.log | cut -c1-300
python3 bench.py --decode-only > gpurun_out/${P}_decode_bench.json 2> /dev/null
LAS_NO_XCD_LOCAL_ROWS=1 python3 bench.py --decode-only > gpurun_out/${P}_decode_bench_noxcd.json 2> /dev/null
export LAS_ALLOW_SERIAL_STREAMS=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_dec_${P}_$c -o p -- python3 bench.py --decode-only > /tmp/pmc_dec_${P}_$c.log 2>&1
done
python3 tools/pmc_summary.py gpurun_out/${P}_decode /tmp/pmc_dec_${P}_FETCH_SIZE /tmp/pmc_dec_${P}_WRITE_SIZE > gpurun_out/${P}_decode_pmc_summary.log 2>&1
unset LAS_ALLOW_SERIAL_STREAMS
python3 - <<'PY'
import json
for f in ("gpurun_out/r6d_decode_bench.json", "gpurun_out/r6d_decode_bench_noxcd.json"):
    d = json.load(open(f))["decode"]
    print(f, d["value"], d["value_b16"], d["us_per_decode_step"], d["step_parts_us"], d["timing"]["spread"], d["value_b64_stream"]["value"] if d.get("value_b64_stream") else None)
p = json.load(open("gpurun_out/r6d_decode_pmc.json"))
for k, v in p["kernels"].items():
    if k.startswith(("dec_", "lstm_cell", "beam_")): print(k, v["calls"], v["avg_us"], v.get("hbm_bytes_per_launch"))
PY
