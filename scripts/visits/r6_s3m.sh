set -x
O=gpurun_out/s3m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "dequant or prefill or large_tile_int4 or auto_takes or unpack or w4a8 or qbits_conv or int4" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python scripts/ab.py --workloads int4_prefill w4a8 --env QUANTO_HIP_DEQ_IDX64=1,0 --rounds 9 > $O/ab_deq.jsonl 2>$O/ab.err; cut -c1-170 $O/ab_deq.jsonl
