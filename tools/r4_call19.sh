cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c19
mkdir -p $O
bash tools/ab.sh r4c19 2 "CUTIE_AMD_DEFER_MEM=0" "CUTIE_AMD_DEFER_MEM=1" "CUTIE_AMD_GRAPHS=1" "CUTIE_AMD_GRAPHS=1 CUTIE_AMD_DEFER_MEM=1" 2>&1 | tee $O/1_ab.log
