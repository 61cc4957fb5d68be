#!/bin/bash
# Visit 13: the -m gpu suite again (one test adapted to the eviction semantics), what the env step inside the raster
# launch executes (SQ counters, fused minus raster) and where step_run's time goes (stamps inside it; 16 / 8 / 4
# waves per CU).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-v13}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -n 4 $OUT/pytest.log
bash tools/step_counters.sh $OUT 2>&1 | tail -n 30
cd $R
(OPTS="0 0" timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_16.txt); head -n 22 $OUT/phase_stamps_16.txt
(MG_RENDER_WPB=8 timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_wpb8.txt); head -n 22 $OUT/phase_stamps_wpb8.txt
(MG_RENDER_WPB=4 MG_RENDER_PER_CU=1 timeout 200 python tools/phase_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_wpb4_1percu.txt); head -n 22 $OUT/phase_stamps_wpb4_1percu.txt
