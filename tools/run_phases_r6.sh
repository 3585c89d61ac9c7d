cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6meas; mkdir -p $O
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
rm -rf /tmp/prof_t; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 6 --warmup 2 $F > $O/bench_trace.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
python tools/step_phases.py $f $O/step_phases_800x800.txt 16 > /dev/null
python tools/step_phases.py $f $O/step_phases_800x800_inline.txt 3 > /dev/null
python tools/step_listing.py $f "" adamw_finish 15 > $O/step_listing_pipelined.txt 2>&1
python tools/step_gaps.py $f 15 8 > $O/step_gaps.txt 2>&1
cut -c1-110 $O/step_phases_800x800.txt; cat $O/step_gaps.txt
head -1 $f | tr ',' '\n' | head -30
