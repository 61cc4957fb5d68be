#!/bin/bash
# Visit 19: the -m gpu suite and the final evidence set on the restructured head (clean build).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v19}
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -n 4 $OUT/pytest.log
bash tools/gpu_r3_final.sh ${1:-v19}/final
