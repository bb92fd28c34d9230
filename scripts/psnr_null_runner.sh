#!/bin/bash
# Null-control windows of scripts/psnr_windows.py in the build container: the oracle restarted from its own kept states with ANOTHER
# thread count (the trajectories ran with 1), one window after another as the states appear; finished windows are skipped on restart.
#   nohup bash scripts/psnr_null_runner.sh "1 2" 2 &        (seeds, threads)
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
SEEDS=${1:-"1 2 3 4"}; THREADS=${2:-2}
STARTS=${STARTS:-"0 500 1000 2000 3000 4000 5000 5750 6000 6250 6500 6750 7000 7500 8000 8500 9000 9500 9750"}
OUT=profiles/r06_psnr/null_windows_${THREADS}threads.log; mkdir -p profiles/r06_psnr; touch $OUT
while true; do
  pending=0
  for k in $STARTS; do for s in $SEEDS; do
    grep -q "mode null seed $s start $k " $OUT && continue
    f0=psnr_states_r06/cpu_seed$s.state.it$(printf %05d $k); f1=psnr_states_r06/cpu_seed$s.state.it$(printf %05d $((k+250)))
    if [ -f $f0 ] && [ -f $f1 ]; then
      OMP_NUM_THREADS=$THREADS nice -n 10 python scripts/psnr_windows.py --modes null --threads $THREADS --seeds $s --at $k --states psnr_states_r06 2>/dev/null | grep WINDOW >> $OUT
    else pending=1; fi
  done; done
  [ $pending = 0 ] && break
  sleep 120
done
echo "null runner done: $SEEDS" >> $OUT
