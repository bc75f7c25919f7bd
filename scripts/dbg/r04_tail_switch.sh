cd $GRAFT_REPO_ROOT
v() { tag=$1; keys=$2; light=$3; n=$4; DBG_LIGHT=$light DBG_KEYS=$keys timeout 900 python scripts/dbg/multiseq_first_diff.py 16 8 2 75 $n > /tmp/v.out 2>&1; grep "^run" /tmp/v.out | awk -v t="$tag" '{ n++; if ($0 !~ /: 0 of /) f++ } END { printf "%-40s failing runs %d of %d\n", t, f+0, n }'; grep "^    {" /tmp/v.out | cut -c1-330 | head -${5:-0}; }
v "SHARED accept+gauge launch, race fixed" batch_shared_tail 1 60 3
v "default (members' own launches), race fixed" "" 1 30 3
