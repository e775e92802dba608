#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them at round end
mkdir -p gpurun_out/suite
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/suite/suite.log 2>&1; echo "suite rc=$?"
grep -E "passed|failed|error" gpurun_out/suite/suite.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
