# Round-6 switch matrix: the whole GPU suite under every run-time routing switch that is left (each run: full `pytest -m gpu`), + smoke().
mkdir -p gpurun_out
OUT=gpurun_out/r06_switch_matrix.txt; : > $OUT
run() { echo "== $*" | tee -a $OUT; env "$@" timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --tb=line -x 2>&1 | grep -vE "of the bound|^observed|warnings summary|DeprecationWarning|warnings.warn|Docs:|^$|test_clip_load_runs" | tail -3 | tee -a $OUT; }
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -1 | tee -a $OUT
run PCLIP_GEMM_4W=0
run PCLIP_CLASSIFY_MID=0
run PCLIP_CLASSIFY_PANEL=0
run PCLIP_GEMM_BAND=6 PCLIP_GEMM_BAND_N=2
run PCLIP_CLASSIFY_PANEL_PASSES=1
run PCLIP_CLASSIFY_PANEL_EXACT=1
