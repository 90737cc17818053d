# conv2_bwd_x6: parity, timing beside the f32 kernel, phase counters (timing build), product build restored
timeout 300 python -m pytest tests/test_conv_gpu.py -q -x -k "backward or autograd_with_gather or fused_vs_miopen or run_to_run" 2>&1 | tail -4
python scripts/conv_bench.py 8192 --no-model --only=conv2_bwd_fused,conv2_bwd_x6 2>/dev/null
scripts/debug/phase_sweep.sh ${1:-r3_x6} conv2_bwd_x6 8 "" > /dev/null 2>&1; cat gpurun_out/${1:-r3_x6}_sweep.log
