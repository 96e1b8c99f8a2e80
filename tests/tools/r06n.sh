cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 4300 python tests/full_config_parity.py config5 4096 > gpurun_out/parity_config5_4096.log 2>&1; echo "config5@4096 rc $?"; tail -c 900 gpurun_out/parity_config5_4096.log; echo
cp gpurun_out/config5_full_parity.json gpurun_out/config5_4096spp_full_parity.json
