# rocprofv3 evidence for profiles/: (1) kernel-trace stats of the bench command, (2)-(4) PMC passes (SQ set, FETCH_SIZE, WRITE_SIZE).
# The conv autotuner runs first, unprofiled, into a persisted tile cache, so the profiles contain the steady-state launches only.
# usage (on the MI355X box, from the repo root): bash tools/profile_round.sh r01
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=16
export CUTIE_AMD_TILE_CACHE=/tmp/cutie_tiles.json
R=${1:-r01}
BENCH="python bench.py --steps 60 --warmup 10 --preroll 60 --cpu-frames 0 --no-roofline --clips-in-flight 0"
OUT=/tmp/prof_$R
rm -rf $OUT; mkdir -p $OUT gpurun_out
python bench.py --steps 20 --warmup 5 --preroll 30 --cpu-frames 0 --no-roofline --clips-in-flight 0 > $OUT/tune.log 2>&1
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16" "FETCH_SIZE" "WRITE_SIZE" ${EXTRA_PMC:+"$EXTRA_PMC"}; do
  i=$((i+1))
  timeout 500 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc$i -- $BENCH --no-lookahead > $OUT/pmc$i.log 2>&1 || tail -3 $OUT/pmc$i.log
done
#  counter evidence for the STACKED score launches: the SQ set once more on the HINTED leg (one read-out per bank version:
# aff_score4_kernel<4, 0> etc. exist only there); summarised per kernel instantiation as `affinity_score_pass_hinted`
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/hinted -- $BENCH > $OUT/hinted.log 2>&1 || tail -3 $OUT/hinted.log
python tools/profile_summarize.py $OUT $R
cp profiles/${R}_* gpurun_out/
tail -2 $OUT/stats.log
