R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3_c; mkdir -p $O
for mode in materialized; do
  python3 $R/tools/mode_run.py $mode 7000 > $O/power_$mode.run 2>&1 &
  PID=$!
  sleep 14
  for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|junction\|fclk" ; echo ---; sleep 0.5; done > $O/power_$mode.txt 2>&1
  wait $PID
  cat $O/power_$mode.run | tail -1
done
rocm-smi --showmaxpower 2>&1 | grep -i power > $O/power_cap.txt
cat $O/power_cap.txt
