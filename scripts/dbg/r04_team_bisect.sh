# which part of the batched / team path makes some sequences diverge (S32, groups of 16, 40 steps): per-sequence ATE of each variant against the unbatched run
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); a = d['config']['ate_rmse_m_vs_ground_truth_per_sequence']
print('$tag', d['value'], 'max', max(a), 'bad', [(i, v) for i, v in enumerate(a) if v > 0.02])"; }
run unbatched_t1        --sequences 32 --runner-threads 1 --steps 40
run g16_t2_no_teams     --sequences 32 --batched --group-size 16 --runner-threads 2 --steps 40
run g16_t4_teams2       --sequences 32 --batched --group-size 16 --runner-threads 4 --steps 40
run g16_t8_teams4       --sequences 32 --batched --group-size 16 --runner-threads 8 --steps 40
run g16_t8_nofront      --sequences 32 --batched --group-size 16 --runner-threads 8 --steps 40 --no-batch-front
run g16_t8_teams4_again --sequences 32 --batched --group-size 16 --runner-threads 8 --steps 40
run s16_default         --sequences 16 --batched --steps 60
run s16_default_again   --sequences 16 --batched --steps 60
