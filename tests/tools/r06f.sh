cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in "config2 1024" "config5 256" "config3_strict 256" "config3 256" "config4 2" "config4_strict 2"; do
  set -- $c
  timeout 1500 python tests/full_config_parity.py $1 $2 > gpurun_out/parity_$1.log 2>&1; echo "$1 rc $?"; tail -c 600 gpurun_out/parity_$1.log; echo
done
