set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r02
# kernel trace + stats of the bench command itself (cfg3, one chunk of 4096 to keep the trace small)
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02/kt -o cfg3_b4096 -- python3 bench.py --workload cfg3 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r02/cfg3_b4096_bench.json 2> gpurun_out/prof_r02/kt.err
echo kt rc=$?
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/prof_r02/pmc_$c -o cfg3 -- python3 bench.py --workload cfg3 --batch 4096 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_r02/pmc_$c.err
  echo $c rc=$?
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace -d gpurun_out/prof_r02/pmc_sq -o cfg3 -- python3 bench.py --workload cfg3 --batch 4096 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_r02/pmc_sq.err
echo sq rc=$?
find gpurun_out/prof_r02 -name "*.csv" | head -20
du -sh gpurun_out/prof_r02
