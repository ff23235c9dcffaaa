set -x
O=gpurun_out/s3i; mkdir -p $O
timeout 900 python -m pytest tests/test_qconv2d.py tests/test_backward_and_workspace.py -m gpu -q -x -p no:cacheprovider -k "depthwise or backward" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python scripts/time_depthwise.py > $O/depthwise_strip.jsonl 2>$O/err.txt
QUANTO_HIP_EXPERIMENT=1 QUANTO_HIP_DW_STRIP=0 python scripts/time_depthwise.py > $O/depthwise_quads.jsonl 2>>$O/err.txt
cat $O/depthwise_strip.jsonl; cat $O/depthwise_quads.jsonl | cut -c1-140
