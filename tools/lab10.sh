#!/bin/bash
# fp32-bulk mode, config 3: training points per split-K slice of the SYRK launches (fp32 accumulation runs over one
# slice before the fp64 slice sum): time and the evidence / gradient against the fp64 evaluation of the same problem.
for r in 2048 4096 8192 16384; do echo "SLICE_ROWS=$r"; GPRHIP_SLICE_ROWS=$r PREC=f32 python3 tools/run_config3.py | cut -c1-150; done
PREC=f64 python3 tools/run_config3.py | head -1
