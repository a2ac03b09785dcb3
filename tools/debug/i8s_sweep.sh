#!/bin/bash
# timing experiments on the stride-4 kernel: SN_CONV_I8_DBG bits (results are wrong with dbg != 0 except 256)
for dbg in 0 256 128 384 1; do
    echo "== dbg=$dbg"
    SN_CONV_I8_DBG=$dbg timeout -k 10 100 python tools/conv_ab.py --rounds 3 --iters 30 2>&1 | grep -E "stride4 noguard"
done
