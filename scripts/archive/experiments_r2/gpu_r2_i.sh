#!/bin/bash
# round-2 validation: full GPU suite, smoke, driver-style bench, sharded bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2i; O=gpurun_out/r2i
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python bench.py --shard --no-cpu-baseline > $O/bench_shard.json 2> $O/bench_shard.err; echo "shard rc=$?" >> $O/rc.txt
tail -n 3 $O/pytest.log; cat $O/rc.txt; cat $O/bench_default.json; cat $O/bench_shard.json; tail -n 3 $O/bench_shard.err
