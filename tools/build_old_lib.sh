# tools/abl/libcutie_hip_OLD.so = the kernel library of a given commit (default HEAD), for A/B runs of uncommitted kernel changes under the
# same Python ($CUTIE_AMD_LIB).  Builds in /tmp from `git archive`; nothing of it is committed (tools/abl/ is git-ignored, but travels with gpurun).
set -e
REV=${1:-HEAD}
cd "$(dirname "$0")/.."
rm -rf /tmp/cutie_old && mkdir -p /tmp/cutie_old tools/abl
git archive $REV cutie_amd/csrc include | tar -x -C /tmp/cutie_old
make -s -C /tmp/cutie_old/cutie_amd/csrc -j8 > /tmp/cutie_old/build.log 2>&1
cp /tmp/cutie_old/cutie_amd/libcutie_hip.so tools/abl/libcutie_hip_OLD.so
ls -la tools/abl/libcutie_hip_OLD.so
