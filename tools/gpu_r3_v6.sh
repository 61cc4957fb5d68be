#!/bin/bash
# Round 3, visit 6: off-path instantiations (compile-time tile sizes 6 / 11, 12-wave 'prestige' workgroups) A/B,
# SQ counters for tile 5 / 11 / the human_player case / tile 8, golden 'prestige' + generic-raster tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v6}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout 900 python -m pytest tests -m gpu -q -x -k "golden or fuzz or batch_vs_oracle or fused or view" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log)
tail -n 4 $OUT/pytest_subset.log
(cd $R && timeout 400 python tools/ab_offpath.py 2>&1 | grep -v amdgpu.ids > $OUT/ab_offpath.jsonl); cut -c1-330 $OUT/ab_offpath.jsonl
bash $R/tools/pmc_offpath.sh $OUT > /dev/null 2>&1; cat $OUT/pmc_offpath.txt
