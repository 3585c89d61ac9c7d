cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F="--no-cpu-baseline --no-alt --no-extra --no-inference --no-real-data"
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -- python bench.py --mode graph --steps 4 --warmup 2 $F > /dev/null 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv")
python tools/step_listing.py $f mask_prep adamw_finish > gpurun_out/step_listing.txt 2>&1
python tools/step_listing.py $f "" adamw_finish 15 > gpurun_out/step_listing_steady.txt 2>&1
wc -l gpurun_out/step_listing.txt gpurun_out/step_listing_steady.txt
