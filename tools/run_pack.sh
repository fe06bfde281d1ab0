MMX_FUSE=6 timeout 200 python tools/kbench.py --blocks 64 --mask --sigmas 3 3.5 4 4.5 5 2>&1 | grep -E "zx path|zxpass"
ZX_CHECK_MODE=6 timeout 600 python tools/zx4_check.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_zx or kernel_radius or sweep" 2>&1 | tail -2
