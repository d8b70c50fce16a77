#!/bin/bash
# same-box A/B: actor tick and learner step on two streams (the default) against strictly one after the other
for v in 0 1 0 1; do RELA_BENCH_ONE_STREAM=$v timeout -k 10 200 python bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-threaded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one_stream=$v', d['value'], d['ms_per_step'], d['summary'])"; done
