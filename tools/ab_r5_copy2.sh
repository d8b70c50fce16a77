#!/bin/bash
# bench + R2D2 line + the tests that pin the tick (lock-step goldens, actor parity) after the tick's small copies became one launch
python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-threaded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('apex', d['value'], d['ms_per_step'], d['summary'])"
python bench.py --algo r2d2 --steps 60 --warmup 10 --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r2d2', d['value'], d['ms_per_step'], d.get('grad_steps_per_s'))"
python -m pytest tests/test_agent_ops_gpu.py tests/test_r2d2_actor_gpu.py tests/test_e2e_gpu.py tests/test_dedup_gpu.py -x -q -m gpu 2>&1 | tail -2
