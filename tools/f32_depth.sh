#!/bin/bash
# operand prefetch depth / occupancy of the fp32 wide-accumulation product (GPAK_F32_RSD: 2 / 4 at two waves per SIMD,
# 8 / 12 / 16 / 24 at one): tools/time_predict.py at M = 131072 for each
for d in 4 2 8 16 4; do   # the values gemm_f32.hip has builds for
  echo "== GPAK_F32_RSD=$d"
  GPAK_F32_RSD=$d python tools/time_predict.py depth 131072 | tail -1
done
