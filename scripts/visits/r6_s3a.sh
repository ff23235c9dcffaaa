set -x
O=gpurun_out/s3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "fused4 or causal_lm or int4_prefill or reference_style" > $O/pytest_fused.log 2>&1; tail -3 $O/pytest_fused.log
python scripts/probes/fused4_hash.py > $O/hash_new.jsonl 2>$O/hash_new.err
python scripts/ab_prefill.py --shapes 4096x4096 14336x4096 4096x14336 --ms 128 256 512 1024 2048 --variants mfma_fused4 > $O/fused4_new.jsonl 2>$O/fused4_new.err
cp optimum_quanto_amd/lib/libquanto_hip.so /tmp/new.so
cp scripts/probes/libquanto_hip_prev.so optimum_quanto_amd/lib/libquanto_hip.so
python scripts/probes/fused4_hash.py > $O/hash_prev.jsonl 2>$O/hash_prev.err
python scripts/ab_prefill.py --shapes 4096x4096 14336x4096 4096x14336 --ms 128 256 512 1024 2048 --variants mfma_fused4 > $O/fused4_prev.jsonl 2>$O/fused4_prev.err
cp /tmp/new.so optimum_quanto_amd/lib/libquanto_hip.so
diff $O/hash_new.jsonl $O/hash_prev.jsonl && echo HASH_IDENTICAL
python scripts/cfg2_overlap_bound.py > $O/cfg2_overlap_bound.jsonl 2>$O/cfg2_overlap_bound.err
python scripts/power_probe.py --cfgs 0 2 3 --seconds 3 > $O/cfg2_power_cfgs.jsonl 2>$O/cfg2_power.err
tail -5 $O/cfg2_overlap_bound.jsonl
