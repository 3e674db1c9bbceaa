set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python tools/bench_module.py denoiseprofile --size 100MP --steps 1 > gpurun_out/r02j_dn.log 2>&1
python tools/kernel_trace_list.py /tmp/kt dn_ > gpurun_out/r02j_dn_trace.txt; cat gpurun_out/r02j_dn_trace.txt
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python tools/bench_module.py diffuse --size 100MP --steps 1 > gpurun_out/r02j_diffuse.log 2>&1
python tools/kernel_trace_list.py /tmp/kt > gpurun_out/r02j_diffuse_trace.txt; tail -45 gpurun_out/r02j_diffuse_trace.txt
