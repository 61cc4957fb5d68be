#!/bin/bash
# Round 3, visit 12: put_obj evicting agents (interact goldens), reject tables, the kernel back to the whole-batch
# step — suite, then the evidence set of the final kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v12}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -n 4 $OUT/pytest.log
bash $R/tools/gpu_r3_final.sh ${1:-v12}/final
