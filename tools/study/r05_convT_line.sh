#!/bin/bash
python tools/costreg_layers_timing.py 8 2>/dev/null | grep "conv9\|conv11\|network" | sed 's/convT3d_k3_s2_bf16x3//; s/weight splitting.*network/network/; s/;  checksum.*//' | tr -s ' ' | tr '\n' '|'
echo
