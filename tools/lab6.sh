#!/bin/bash
set -u
for sk in 0 1; do echo "SKIP=$sk"; SKIP=$sk ORD=3 LAB=1 ./build/gemm_check | grep "^time" | grep "triu\|full" | tail -6; done
