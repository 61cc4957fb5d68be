R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 300 python bench.py > $OUT/bench.log 2> $OUT/bench.err; echo "rc=$?" >> $OUT/bench.err)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline > $OUT/prof_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_kt -o kt --output-format csv -- python $R/tools/profile_render.py > $OUT/prof_kt.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_wr -o wr --output-format csv -- python $R/tools/profile_render.py --iters 4 > $OUT/prof_wr.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_rd -o rd --output-format csv -- python $R/tools/profile_render.py --iters 4 > $OUT/prof_rd.log 2>&1
rm -f $OUT/bench_other.log
for w in MarlGrid-3AgentCluttered11x11-v0:4096 MarlGrid-4AgentEmpty9x9-v0:65536 Custom-8AgentCluttered30x30:131072; do
  (cd $R && timeout 300 python bench.py --workload ${w%%:*} --batch-per-gpu ${w##*:} --steps 50 --warmup 5 --no-cpu-baseline >> $OUT/bench_other.log 2>> $OUT/bench_other.err)
done
cat $OUT/bench.log | cut -c1-300; tail -1 $OUT/prof_bench.log | cut -c1-300; cut -c1-120 $OUT/prof_bench/kt_kernel_stats.csv | head -5
