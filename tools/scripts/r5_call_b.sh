mkdir -p gpurun_out/b
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "maf" > gpurun_out/b/pytest_maf.log 2>&1; tail -3 gpurun_out/b/pytest_maf.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_hygiene.py -x -q -k "counted_waits" > gpurun_out/b/pytest_waits.log 2>&1; tail -2 gpurun_out/b/pytest_waits.log | cut -c1-300
NF_REFERENCE_PATH=.refstage timeout 200 python -m pytest tests/test_gpu_parity.py -k "reference_own_containers or reference_style_container" -v -rs 2>&1 | tail -12 | cut -c1-400 > gpurun_out/r05_reference_containers_gpubox.log; tail -4 gpurun_out/r05_reference_containers_gpubox.log
timeout 600 python tools/maf_ablate5.py run > gpurun_out/b/ablate.jsonl 2>&1; cat gpurun_out/b/ablate.jsonl | cut -c1-300
