# kernel timeline of the 16-sequence line (4 groups of 4): where does a group's BA queue wait?
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=12
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace16 -- python3 /root/repo/bench.py --sequences 16 --batched --steps 30 --no-cpu-baseline > /root/repo/gpurun_out/trace16_line.json 2>/dev/null
f=$(find /tmp/trace16 -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, json
rows=list(csv.DictReader(open(sys.argv[1])))
print('rows', len(rows), rows[0].keys())
# keep the last 40 % of the run (timed region)
ts=[(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60], r.get('Queue_Id'), r.get('Stream_Id')) for r in rows]
t0=min(t[0] for t in ts); t1=max(t[1] for t in ts)
cut=t0+0.7*(t1-t0)
ts=[t for t in ts if t[0]>=cut]
byq=collections.defaultdict(list)
for t in ts: byq[(t[3],t[4])].append(t)
out={}
for q,l in byq.items():
    l.sort()
    busy=sum(e-s for s,e,_,_,_ in l); span=l[-1][1]-l[0][0]
    names=collections.Counter(n for _,_,n,_,_ in l)
    gaps=[l[i+1][0]-l[i][1] for i in range(len(l)-1)]
    gaps_by=collections.defaultdict(list)
    for i in range(len(l)-1): gaps_by[(l[i][2][:28], l[i+1][2][:28])].append(l[i+1][0]-l[i][1])
    top=sorted(((sum(v)/1e3, len(v), k) for k,v in gaps_by.items()), reverse=True)[:8]
    dur=collections.defaultdict(list)
    for s0,e0,n0,_,_ in l: dur[n0[:40] or 'dv_copy_kernel'].append(e0-s0)
    durs=sorted(((sum(v)/1e3, len(v), round(sum(v)/len(v)/1e3,1), k) for k,v in dur.items()), reverse=True)[:12]
    out[str(q)]=dict(kernel_us_total_n_avg=durs, n=len(l), busy_ms=busy/1e6, span_ms=span/1e6, top_kernels=names.most_common(4), top_gap_pairs_us_total=[(round(a,1), n, k) for a,n,k in top])
json.dump(out, open('/root/repo/gpurun_out/trace16_queues.json','w'), indent=1)
for q,v in sorted(out.items(), key=lambda kv:-kv[1]['busy_ms'])[1:3]:
    print(q, v['n'], round(v['busy_ms'],1), round(v['span_ms'],1), v['top_kernels'][:3])
    for g in v['top_gap_pairs_us_total'][:3]: print('     gap', g)
    for d in v['kernel_us_total_n_avg'][:12]: print('     dur', d)
PY
