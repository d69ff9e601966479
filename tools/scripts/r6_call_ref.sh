# round 6: the reference itself on the GPU box's host cores (staged copy, .refstage/): thread sweep with a per-setting limit, and the
# real-container drop-in test
mkdir -p gpurun_out/ref
timeout 1500 python tools/cpu_reference.py --ref .refstage --where "gpu box" --threads 16,8,32,64,128,256 --per-setting-timeout 150 --repeats 2 --quick --out gpurun_out/ref/r06_cpu_reference_gpubox.json > gpurun_out/ref/cpu_reference.log 2>&1
grep -E "torch_threads" gpurun_out/ref/cpu_reference.log | cut -c1-220 | head -8
NF_REFERENCE_PATH=.refstage timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "reference_own_containers" -rs > gpurun_out/ref/r06_reference_containers_gpubox.log 2>&1; tail -3 gpurun_out/ref/r06_reference_containers_gpubox.log
nproc; lscpu | grep "Model name" | head -1
