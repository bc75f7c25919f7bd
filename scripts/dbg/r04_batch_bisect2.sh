cd $GRAFT_REPO_ROOT
N=${1:-24}
v() { tag=$1; keys=$2; shift 2; DBG_KEYS=$keys python scripts/dbg/multiseq_first_diff.py "$@" $N 2>&1 > /tmp/v.out; grep "^run" /tmp/v.out | awk -v t="$tag" '{ n++; if ($0 !~ /: 0 of /) f++ } END { printf "%-34s failing runs %d of %d\n", t, f, n }'; grep "^    {" /tmp/v.out | python3 -c "
import sys, ast, collections
rows = [ast.literal_eval(l.strip()) for l in sys.stdin]
if rows:
    print('      events', len(rows), ' groups', dict(collections.Counter(r['group'] for r in rows)), ' iteration counts differ in', sum(1 for r in rows if r['ref'] and r['got'] and r['ref']['iters'] != r['got']['iters']), ' rows differ in', sum(1 for r in rows if not r['rows_identical_there']))
    e = sorted(r['pos_err_last'] for r in rows); print('      |dp| at the last frame: min %.2e median %.2e max %.2e' % (e[0], e[len(e)//2], e[-1]), ' first bad frame: min', min(r['first_frame_record_diff'] for r in rows), 'max', max(r['first_frame_record_diff'] for r in rows))
"; }
v "baseline (2 groups, 2 threads)"  ""                                                       16 8 2 75
v "solve single"                    batch_single_solve                                       16 8 2 75
v "all three single"                batch_single_eval,batch_single_reduce,batch_single_solve 16 8 2 75
v "reduce single"                   batch_single_reduce                                      16 8 2 75
