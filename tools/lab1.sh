#!/bin/bash
out=gpurun_out/lab5; mkdir -p $out
for w in 4 8; do GPRHIP_ENG_WAVES=$w LAB=1 ./build/gemm_check > $out/lab_w$w.txt 2>&1; GPRHIP_ENG_WAVES=$w NOADV=1 LAB=1 ./build/gemm_check > $out/lab_noadv_w$w.txt 2>&1; done
tail -n 6 $out/lab_w4.txt $out/lab_noadv_w4.txt $out/lab_w8.txt $out/lab_noadv_w8.txt
