cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c5
mkdir -p $O
AFF_VARIANTS=1 timeout 300 python tools/aff_batch_ab.py 12200 2>&1 | tee $O/aff_batch_variants.txt
