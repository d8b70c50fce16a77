run() { echo "== $*"; env "$@" RELA_PRECISION=f32x3 timeout -k 10 200 python rela_amd/pyrela/benchmark.py --grid $GRID --epoch_sec 1.5 --num_epoch 3 --replay_buffer_size 4194304 --burn_in_frames 20000 --env $ENVK 2>&1 | grep -E "^act rate" | cut -c1-150; }
GRID=64x100 ENVK=sliding run RELA_CU_RESERVE=4
GRID=64x100 ENVK=sliding run RELA_CU_RESERVE=8
GRID=64x100 ENVK=sliding run RELA_CU_RESERVE=16
GRID=32x200 ENVK=sliding run RELA_CU_RESERVE=8
GRID=16x400 ENVK=sliding run RELA_CU_RESERVE=8
GRID=64x100 ENVK=null run RELA_CU_RESERVE=8
GRID=16x400 ENVK=null run RELA_CU_RESERVE=8
GRID=64x100 ENVK=fresh run RELA_CU_RESERVE=8
