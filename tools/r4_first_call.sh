# Round 4, first GPU call, on branch next/all (= main + the five branches prepared at the end of round 3 without GPU minutes:
# next/touch-rewire, next/pc-pairstep, next/bank-write, next/cout1-rows, next/up4-vec).  Nothing in them has run on a GPU yet.
#   1. the kernel tests of what is new, the frame's parity tests with everything on
#   2. cold / warm timing: pair-step tiles against the table's tiles, the Cout = 1 rows kernel
#   3. A/B inside this box: everything off (= main's behaviour) | everything on | each feature alone
# Keep what wins, drop what does not (DESIGN.md 9 lists the motivation of each), then run the full suite on the result.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4first
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "140 or 141 or 142 or 143 or 144 or 145 or 146 or 147 or bank_write or cout1 or seg_epilogue or next_weights" > $O/1_kernels.log 2>&1; tail -3 $O/1_kernels.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "stages or trajectory or 480p or bank_contents or other_baseline" > $O/2_parity.log 2>&1; tail -3 $O/2_parity.log
timeout 150 python tools/cold_probe.py \
  3,30,54,256,256,3,129 3,30,54,256,256,3,145 3,30,54,256,256,3,110 3,30,54,256,256,3,141 3,30,54,256,256,3,142 3,30,54,256,256,3,101 3,30,54,256,256,3,140 \
  1,30,54,1024,256,1,102 1,30,54,1024,256,1,143 1,30,54,256,1024,1,101 1,30,54,256,1024,1,140 1,30,54,256,1024,1,110 1,30,54,256,1024,1,141 \
  1,30,54,256,256,3,109 1,30,54,256,256,3,143 1,30,54,256,256,3,130 1,30,54,256,256,3,147 \
  3,120,216,128,128,3,122 3,120,216,128,128,3,146 3,120,216,128,128,3,103 3,120,216,128,128,3,144 \
  3,60,108,128,128,3,108 3,60,108,128,128,3,144 3,60,108,128,128,3,141 \
  3,120,216,128,1,3,19 2>&1 | tee $O/3_cold.log | tail -28
CUTIE_AMD_COUT1_ROWS=0 timeout 60 python tools/cold_probe.py 3,120,216,128,1,3,19 2>&1 | tee $O/3_cold_cout1_plain.log | tail -1
OFF="CUTIE_AMD_WPF_REWIRE=0 CUTIE_AMD_BANK_WRITE=0 CUTIE_AMD_COUT1_ROWS=0 CUTIE_AMD_UP4_VEC=0"
bash tools/ab.sh r4first 2 "$OFF" "CUTIE_AMD_WPF_REWIRE=1" \
  "CUTIE_AMD_WPF_REWIRE=1 CUTIE_AMD_BANK_WRITE=0 CUTIE_AMD_COUT1_ROWS=0 CUTIE_AMD_UP4_VEC=0" \
  "CUTIE_AMD_WPF_REWIRE=0 CUTIE_AMD_BANK_WRITE=1 CUTIE_AMD_COUT1_ROWS=0 CUTIE_AMD_UP4_VEC=0" \
  "CUTIE_AMD_WPF_REWIRE=0 CUTIE_AMD_BANK_WRITE=0 CUTIE_AMD_COUT1_ROWS=1 CUTIE_AMD_UP4_VEC=0" \
  "CUTIE_AMD_WPF_REWIRE=0 CUTIE_AMD_BANK_WRITE=0 CUTIE_AMD_COUT1_ROWS=0 CUTIE_AMD_UP4_VEC=1" 2>&1 | tee $O/4_ab.log
