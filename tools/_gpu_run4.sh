set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nlmeans.py tests/test_gpu_tiled.py tests/test_gpu_parity_at_size.py -m gpu -x -q > gpurun_out/r02d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02d_pytest.log; tail -4 gpurun_out/r02d_pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-legs --no-verify > gpurun_out/r02d_bench.log 2>&1; tail -1 gpurun_out/r02d_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['full_pipe']['ms_per_step'], d['config']['full_pipe']['kernels_ms_per_step'])"
Q="--no-cpu-baseline --no-host-legs --no-verify --steps 2 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU --output-format csv -d gpurun_out/r02d_sqa -- python bench.py $Q > gpurun_out/r02d_sqa.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/r02d_sqc -- python bench.py $Q > gpurun_out/r02d_sqc.log 2>&1
A=$(find gpurun_out/r02d_sqa -name '*counter_collection.csv' | head -1)
C=$(find gpurun_out/r02d_sqc -name '*counter_collection.csv' | head -1)
python tools/pmc_sq_json.py gpurun_out/r02d_pmc_sq.json "r02d" $A $C > gpurun_out/r02d_pmc_sq.txt
rm -rf gpurun_out/r02d_sqa gpurun_out/r02d_sqc
grep -E "nlm_chunks|diffuse_pde" -A0 gpurun_out/r02d_pmc_sq.txt
