# Collects the round-3 rocprofv3 evidence on the GPU box:  bash profiles/r03/collect.sh <workload> <batch> [full]
#   kt/      --kernel-trace --stats of `python3 bench.py --workload W --batch B --steps 3 --warmup 1 --no-cpu-baseline`
#   pmc_*/   separate --pmc passes of the same command with --steps 1 (FETCH_SIZE | WRITE_SIZE | SQ_* MFMA / VALU | wait counters)
#   full/    (with `full`, cfg3 only) --kernel-trace --stats of the DEFAULT command `python3 bench.py --no-cpu-baseline`
# ROCm 7.2's rocprofv3 writes rocpd SQLite databases; profiles/r03/summarize.py turns them into the CSVs kept here.
# The program itself follows `--` (no wrapper, no env / bash -c hop): the profiler's library initialises the GPU first.
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
W=${1:-cfg3}; B=${2:-4096}
# TAG names a second set of files for the same workload (e.g. TAG=_gfn: the default matrix-function route; the untagged files of
# this directory were collected on the eigensolver pipeline, which is what ADMMNET_SPECTRAL=0 runs today)
TAG=${TAG:-}
OUT=gpurun_out/prof_r03${TAG}_$W
rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --workload $W --batch $B --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o $W -- $CMD --steps 3 > $OUT/${W}${TAG}_bench.json 2> $OUT/kt.err
echo kt rc=$?
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o $W -- $CMD --steps 1 > /dev/null 2> $OUT/pmc_$c.err
  echo $c rc=$?
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_sq -o $W -- $CMD --steps 1 > /dev/null 2> $OUT/pmc_sq.err
echo sq rc=$?
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_wait -o $W -- $CMD --steps 1 > /dev/null 2> $OUT/pmc_wait.err
echo wait rc=$?
if [ "$3" = "full" ]; then
  rocprofv3 --kernel-trace --stats -d $OUT/full/kt -o ${W}_full -- python3 bench.py --workload $W --no-cpu-baseline > $OUT/${W}${TAG}_full_bench.json 2> $OUT/full.err
  echo full rc=$?
fi
python3 profiles/r03/summarize.py $OUT $OUT/${W}${TAG}_b$B
[ "$3" = "full" ] && python3 profiles/r03/summarize.py $OUT/full $OUT/${W}${TAG}_full
find $OUT -name '*.db' -delete      # the rocpd databases (tens of MB) stay on the box: gpurun_out/ returns at most 64 MiB
du -sh $OUT
