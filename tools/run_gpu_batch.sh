set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
O=gpurun_out/r04
MX_RCCL_LOG=$O/rccl_r04.log MX_PARITY_LOG=$O/parity_r04_whole_config_vs_oracle.log timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
