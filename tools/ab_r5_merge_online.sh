#!/bin/bash
# same-box A/B: the f32x3 learner's two online forwards merged into one 2B-row forward (default) or not
for v in 1 0 1 0; do RELA_LEARNER_MERGE_ONLINE=$v timeout -k 10 300 python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-threaded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge=$v', d['value'], d['ms_per_step'], d['summary'])"; done
for v in 1 0; do RELA_LEARNER_MERGE_ONLINE=$v RELA_BENCH_ONLY=learner timeout -k 10 200 python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-threaded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge=$v learner only', d['ms_per_step'])"; done
