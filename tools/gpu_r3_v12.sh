#!/bin/bash
# Visit 12: the whole -m gpu suite on the final kernels (put_obj eviction, reject tables, whole-batch step), then
# the final evidence set.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v12}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -n 4 $OUT/pytest.log
bash tools/gpu_r3_final.sh ${1:-v12}/final
