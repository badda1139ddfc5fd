cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 420 bash tools/profile.sh r6_fnav10 fnav10 > gpurun_out/r6_profile_fnav10.log 2>&1; echo "fnav10 rc=$?"
