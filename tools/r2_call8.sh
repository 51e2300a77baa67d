OUT=gpurun_out/c8; mkdir -p $OUT
EXTRA_PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS" bash tools/profile_round.sh r02a > $OUT/profile.log 2>&1
tail -5 $OUT/profile.log
cp profiles/r02a_* $OUT/
