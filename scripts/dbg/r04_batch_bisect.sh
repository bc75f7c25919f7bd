# bisect of the open multi-sequence defect: failing runs out of N (any member's frame records differing from the single-thread run) per variant
cd $GRAFT_REPO_ROOT
N=${1:-12}
v() { tag=$1; keys=$2; shift 2; DBG_KEYS=$keys python scripts/dbg/multiseq_first_diff.py "$@" $N 2>&1 | grep "^run" | awk -v t="$tag" '{ n++; if ($0 !~ /: 0 of /) f++ } END { printf "%-34s failing runs %d of %d\n", t, f, n }'; }
v "2 groups, 2 threads (baseline)"   ""                                                       16 8 2 75
v "2 groups, ONE thread"             ""                                                       16 8 1 75
v "1 group of 16, one thread"        ""                                                       16 16 1 75
v "eval single"                      batch_single_eval                                        16 8 2 75
v "reduce single"                    batch_single_reduce                                      16 8 2 75
v "solve single"                     batch_single_solve                                       16 8 2 75
v "eval+reduce+solve single"         batch_single_eval,batch_single_reduce,batch_single_solve 16 8 2 75
