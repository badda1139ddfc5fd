cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in n10 fnav10 fnav; do timeout -k 10 420 bash tools/profile.sh r6_$c $c > gpurun_out/r6_profile_$c.log 2>&1; echo "$c rc=$?"; done
PROF_ARGS="--launch span" timeout -k 10 420 bash tools/profile.sh r6s_fnav fnav > gpurun_out/r6_profile_fnav_span.log 2>&1; echo "fnav span rc=$?"
