#!/bin/bash
# CPU-oracle trajectories (scripts/psnr_trajectory.py) of several seeds on the GPU box's host, NPROC processes x THREADS threads (keep the
# product <= 32: scripts/gpu_cpu_psnr.sh's header), then the HIP fp32 / bf16 trajectories of the same seeds.  -> gpurun_out/trajb/
#   SEEDS="4 5 6 7" ITERS=1000 THREADS=4 bash scripts/gpu_traj_batch.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$R/gpurun_out/trajb; mkdir -p $OUT; cd $R
ITERS=${ITERS:-1000}; THREADS=${THREADS:-4}
python -c "import torch" > /dev/null 2>&1
t0=$(date +%s)
for s in ${SEEDS:-4 5 6 7}; do
  OMP_NUM_THREADS=$THREADS timeout ${CPU_TIMEOUT:-1500} python scripts/psnr_trajectory.py --mode cpu --seed $s --iters $ITERS --threads $THREADS --out $OUT/traj_cpu_seed${s}_$ITERS.json > $OUT/cpu_$s.log 2>&1 &
done
wait
echo "# CPU trajectories: $(( $(date +%s) - t0 )) s for $ITERS iterations, $(echo ${SEEDS:-4 5 6 7} | wc -w) processes x $THREADS threads" | tee $OUT/timing.log
for s in ${SEEDS:-4 5 6 7}; do
  for m in fp32 bf16; do python scripts/psnr_trajectory.py --mode $m --seed $s --iters $ITERS --out $OUT/traj_${m}_seed${s}_$ITERS.json 2>&1 | tail -1; done
done
tail -1 $OUT/cpu_*.log
