# Proposed first GPU call of the next round on this tree (build the diagnostic library HERE first: bash tools/build_diag_aff.sh).
#   1. the whole GPU suite (what the driver runs)
#   2. timelines of both score passes with TWO blocks per CU (call 32 of round 4 ran them with one: profiles/r04_affinity.md section 5)
#   3. the driver's command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c1
mkdir -p $O
timeout 480 python -m pytest tests/ -x -q -m gpu > $O/suite.log 2>&1; tail -3 $O/suite.log
CUTIE_AMD_LIB=$GRAFT_REPO_ROOT/tools/abl/libcutie_hip_ATL.so timeout 60 python tools/aff_timeline.py 2:0 p1:2:0 > $O/aff_timeline.txt 2>&1; head -c 1200 $O/aff_timeline.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_line.json 2> $O/driver_line.err; tail -c 600 $O/driver_line.json
