set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
sed -n '/^export TMPDIR/,$p' tools/prof_r02_final.sh > /tmp/prof_tail.sh
bash /tmp/prof_tail.sh
