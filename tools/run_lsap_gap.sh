cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/prof_l; CDETR_LSAP_COST_LDS=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_l -- python tools/lsap_gap.py > /dev/null 2>&1
  f=$(find /tmp/prof_l -name "*kernel_trace.csv")
  python - "$f" $v <<'PY'
import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["LDS_Block_Size"]))
rows.sort()
gaps=[]; durs=[]; after=[]
for i,r in enumerate(rows):
    if "lsap" in r[2] and i>0 and i+1 < len(rows):
        gaps.append((r[0]-rows[i-1][1])/1e3); durs.append((r[1]-r[0])/1e3); after.append((rows[i+1][0]-r[1])/1e3); lds=r[3]
gaps=gaps[5:]; durs=durs[5:]; after=after[5:]
import statistics as st
print(f"CDETR_LSAP_COST_LDS={sys.argv[2]}: LDS per workgroup {lds} B | gap in front of the solve: median {st.median(gaps):.1f} us (min {min(gaps):.1f}, max {max(gaps):.1f}) | solve {st.median(durs):.1f} us | gap behind it {st.median(after):.1f} us | n={len(gaps)}")
PY
done
