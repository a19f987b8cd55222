O=gpurun_out/r02_k; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
cd cuda-aho-corasick-wu-manber_amd && mkdir -p /tmp/smd && cd /tmp/smd && $GRAFT_REPO_ROOT/cuda-aho-corasick-wu-manber_amd/smatcher ac -m 8 -p_size 100 -n 1048576 -alphabet 4 -c 2>&1 | tail -12
