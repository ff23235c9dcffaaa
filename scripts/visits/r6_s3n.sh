set -x
O=gpurun_out/s3n; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "test_qbytes or int8 or multi" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python scripts/ab.py --workloads int8_decode32 int8_qkv_fused32 --env QUANTO_HIP_GROUP_M=- --rounds 7 2>/dev/null | cut -c1-140
