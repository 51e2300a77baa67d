# Round 5, call 19: in-frame A/B of the re-swept tile entries (same K-order classes: bit-identical results)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c19
mkdir -p $O
cp tools/abl/tiles_r5_candidates.json /tmp/t_all.json; cp tools/abl/tiles_r5_only134.json /tmp/t_134.json
bash tools/ab.sh tiles 3 "CUTIE_AMD_X=0" "CUTIE_AMD_TILE_CACHE=/tmp/t_134.json" "CUTIE_AMD_TILE_CACHE=/tmp/t_all.json" 2>&1 | tee $O/ab.log
