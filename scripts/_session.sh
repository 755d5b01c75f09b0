#!/bin/bash
# session 29: one-pass consumers
mkdir -p gpurun_out/s29
timeout 600 python -m pytest tests/test_consumers.py tests/test_aiming.py tests/test_advice_r01.py tests/test_chunked_trace_gpu.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/s29/pytest.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s29/bench.json 2> gpurun_out/s29/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/s29/bench.json'))
for c in d['consumers']:
    print(c['call'][:50], round(c['ms'],4), c.get('two_pass_ms'), round(c.get('frac',0),3))
print(d['ms_per_step'], d['roofline']['frac'])
P
cat gpurun_out/s29/pytest.txt
