set -x
O=gpurun_out/s3k; mkdir -p $O
cp optimum_quanto_amd/lib/libquanto_hip.so /tmp/cur.so
for rep in 1 2; do
for v in cur glds_sc1 glds_nt glds_sc0_sc1; do
  if [ $v = cur ]; then cp /tmp/cur.so optimum_quanto_amd/lib/libquanto_hip.so; else cp scripts/probes/libquanto_hip_$v.so optimum_quanto_amd/lib/libquanto_hip.so; fi
  python scripts/ab.py --workloads w8a8 fp8a8 cfg4_fp8a8 cfg2 cfg4 int4_prefill --env QUANTO_HIP_GROUP_M=- --sequential --rounds 7 > $O/ab_${v}_$rep.jsonl 2>$O/ab_${v}_$rep.err
done
done
cp /tmp/cur.so optimum_quanto_amd/lib/libquanto_hip.so
python - <<'PY'
import json,glob,collections
d=collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/s3k/ab_*.jsonl')):
    v=f.split('ab_')[1].rsplit('_',1)[0]; rep=f[-7]
    for ln in open(f):
        r=json.loads(ln); d[r['workload']][(v,rep)]=r['us_median']
for w,x in d.items(): print(w, {k:v for k,v in sorted(x.items())})
PY
