cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2
./profiles/ubench/pcie_handoff > gpurun_out/g2/pcie_handoff.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_literal.py -m gpu -q -k "outliers" > gpurun_out/g2/pytest_pop.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/g2/bench_driver.json 2> gpurun_out/g2/bench_driver.err
cat gpurun_out/g2/pcie_handoff.txt; tail -n 3 gpurun_out/g2/pytest_pop.log
