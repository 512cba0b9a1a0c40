python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "both_sides or thresholds or mid_shapes_through" 2>&1 | tail -3
TS=1 build/potrf_check 2>&1 | head -40
