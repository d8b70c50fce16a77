#!/bin/bash
# same-box A/B: the learner's stream joined to the actor's at the end of every step (default) or only at publishes
for v in 1 0 1 0; do RELA_BENCH_JOIN=$v timeout -k 10 200 python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-threaded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('join=$v', d['value'], d['ms_per_step'], d['summary'])"; done
