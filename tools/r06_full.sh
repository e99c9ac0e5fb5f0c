#!/bin/bash
# Round-6 GPU call: PMC passes (mj_k_sp instruction counts, mj_k_encode HBM bytes at the bench's own 65,536 tables) -> their
# summaries under profiles/ of the box's copy, THEN the whole -m gpu suite, the default bench line (which reads those summaries),
# rocprofv3 kernel stats of the v4 / v3 bench, the select / packed-f32 issue microbenchmark, the 2-rank gloo smoke run of bench.py --gpus 2 and __graft_entry__.smoke().
# Output: gpurun_out/$1/ (profiles/ sub-directory = the files to commit).
TAG=${1:-r06}; OUT=/root/repo/gpurun_out/$TAG; mkdir -p $OUT/profiles; cd /root/repo
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex 'mj_k_encode' --output-format csv -d $OUT/pmc_write -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-matrix --version 4 > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'mj_k_encode' --output-format csv -d $OUT/pmc_fetch -- python /root/repo/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-matrix --version 4 > $OUT/pmc_fetch.log 2>&1
cd /root/repo; tools/pmc_sp.sh ${TAG}_pmc_sp 65536 > $OUT/pmc_sp.log 2>&1; tail -2 $OUT/pmc_sp.log
python tools/summarize_sp_pmc.py gpurun_out/${TAG}_pmc_sp r06 > $OUT/summarize_sp.log 2>&1; tail -3 $OUT/summarize_sp.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v4_stats -- python /root/repo/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-matrix --version 4 > $OUT/v4_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v3_stats -- python /root/repo/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-matrix --version 3 > $OUT/v3_stats.log 2>&1
cd /root/repo; python tools/summarize_profiles.py $OUT r06 65536 > $OUT/summarize.log 2>&1; tail -2 $OUT/summarize.log
cp profiles/r06_* profiles/pmc_encode.json $OUT/profiles/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
cp $OUT/bench.json $OUT/profiles/r06_bench_v4.json
timeout 300 python bench.py --gpus 2 --dist-backend gloo --tables 8192 --steps 10 --warmup 3 --preroll 256 --no-cpu-baseline --no-matrix 2> $OUT/gloo.err | grep '^{' > $OUT/profiles/r06_bench_gloo_2ranks_1gpu.json; echo "gloo 2-rank rc=$?"; cut -c1-300 $OUT/profiles/r06_bench_gloo_2ranks_1gpu.json
timeout 300 python bench.py --gpus 2 --dist-backend gloo --scaling strong --tables 16384 --steps 10 --warmup 3 --preroll 256 --no-cpu-baseline --no-matrix 2>> $OUT/gloo.err | grep '^{' > $OUT/profiles/r06_bench_gloo_2ranks_1gpu_strong.json; echo "gloo strong rc=$?"
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
# the small-pool schedule in a kernel trace (three kernels side by side) and its A/B against the schedule switched off
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v4_4096_stats -- python /root/repo/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-matrix --version 4 --tables 4096 > $OUT/v4_4096_stats.log 2>&1
cd /root/repo; python - <<PY > $OUT/profiles/r06_bench_v4_4096_kernel_stats.csv 2> $OUT/stats4096.err
import csv, glob, sys
f = glob.glob("$OUT/v4_4096_stats/*/*kernel_stats.csv")
w = csv.writer(sys.stdout)
for i, r in enumerate(csv.reader(open(f[0]))):
    if i == 0 or any(k in r[0] for k in ("mj_k", "vectorized", "Memset")):
        w.writerow(r)
PY
for n in 4096 8192 16384; do for m in 0 -1; do MJ_SP_WIDE=$m timeout 200 python bench.py --no-cpu-baseline --no-matrix --steps 60 --warmup 5 --tables $n 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'tables': $n, 'MJ_SP_WIDE': $m, 'env_steps_per_s': round(d['value']), 'ms_per_cycle': round(d['ms_per_step'], 3), 'mj_k_sp_ms': round(d['kernel_ms_per_step']['mj_k_sp'], 3), 'sp_schedule': d.get('sp_schedule')}))"; done; done > $OUT/profiles/r06_small_pool_schedule_ab.jsonl; cat $OUT/profiles/r06_small_pool_schedule_ab.jsonl
rm -rf $OUT/v4_4096_stats
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $OUT/gputest.log 2>&1; echo "gputest rc=$?"; tail -4 $OUT/gputest.log; tail -16 $OUT/gputest.log > $OUT/profiles/r06_gputest_tail.txt
rm -rf $OUT/pmc_write $OUT/pmc_fetch $OUT/v4_stats $OUT/v3_stats; du -sh $OUT
