set -x
O=gpurun_out/s3p; mkdir -p $O
export QUANTO_HIP_EXPERIMENT=1
timeout 900 python -m pytest tests/test_qconv2d.py tests/test_backward_and_workspace.py -m gpu -q -x -p no:cacheprovider > $O/pytest_auto.log 2>&1; tail -2 $O/pytest_auto.log
QUANTO_HIP_CONV_ROWS_DB=2 timeout 900 python -m pytest tests/test_qconv2d.py -m gpu -q -x -p no:cacheprovider > $O/pytest_db2.log 2>&1; tail -2 $O/pytest_db2.log
for db in 0 2 0 2; do
  for set in default grid strided; do
    if [ $set = default ]; then A=""; else A="$set"; fi
    QUANTO_HIP_CONV_ROWS_DB=$db TIME_CONV2D_DIRECT_ONLY=1 python scripts/time_conv2d.py qint8 $A 2>/dev/null | sed "s/^{/{\"db\": $db, \"set\": \"$set\", /" >> $O/conv_db_ab.jsonl
  done
done
python - <<'PY'
import json,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for ln in open('gpurun_out/s3p/conv_db_ab.jsonl'):
    r=json.loads(ln); d[(r['set'],r['B'],r['C'],r['H'],r['OC'],r['k'],r['stride'],r['kernel'])][r['db']].append(r['conv_kernel_direct_us'])
for k,v in d.items(): print(k, {a:b for a,b in v.items()})
PY
