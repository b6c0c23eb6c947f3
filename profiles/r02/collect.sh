# Collects the round-2 rocprofv3 evidence on the GPU box:  bash profiles/r02/collect.sh [full]
#   kt/      --kernel-trace --stats of `python3 bench.py --workload cfg3 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline`
#   pmc_*/   separate --pmc passes of the same command with --steps 1 (FETCH_SIZE | WRITE_SIZE | SQ_* MFMA / VALU counters)
#   full/    (with `full`) --kernel-trace --stats of the DEFAULT command `python3 bench.py --no-cpu-baseline`
#            (cfg3 at its full 65 536-signal batch, chunks of 8192)
# ROCm 7.2's rocprofv3 writes rocpd SQLite databases; profiles/r02/summarize.py turns them into the CSVs kept here.
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/kt -o cfg3_b4096 -- python3 bench.py --workload cfg3 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/cfg3_b4096_bench.json 2> $OUT/kt.err
echo kt rc=$?
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o cfg3 -- python3 bench.py --workload cfg3 --batch 4096 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$c.err
  echo $c rc=$?
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_sq -o cfg3 -- python3 bench.py --workload cfg3 --batch 4096 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
echo sq rc=$?
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_wait -o cfg3 -- python3 bench.py --workload cfg3 --batch 4096 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_wait.err
echo wait rc=$?
if [ "$1" = "full" ]; then
  rocprofv3 --kernel-trace --stats -d $OUT/full/kt -o cfg3_full -- python3 bench.py --no-cpu-baseline > $OUT/cfg3_full_bench.json 2> $OUT/full.err
  echo full rc=$?
fi
du -sh $OUT
