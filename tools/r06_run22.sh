mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06r_pytest.log 2>&1; tail -3 gpurun_out/r06r_pytest.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
