cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CDETR_GEMM_PD=4 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "gemm or conv or linear or fuzz or variant" 2>&1 | tail -3
for pd in 2 4; do echo "=== PD=$pd"; CDETR_GEMM_PD=$pd python tools/pd_sweep.py 2>&1 | grep -v amdgpu; done
for pd in 2 4; do CDETR_GEMM_PD=$pd python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>&1 | tail -1 | cut -c1-200; done
