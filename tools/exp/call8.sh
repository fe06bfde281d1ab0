O=gpurun_out/r04h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "replay" 2>&1 | tail -15
