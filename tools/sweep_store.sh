#!/bin/bash
# same-box A/B of the bf16 storage of final_layer.0's activation gradient at batch 64: tools/sweep_store.sh <out dir>
# (the sweep over more layers and over the conv outputs that chose it used developer switches that no longer exist; its result:
#  dact[11] alone 0.843 -> 0.827 ms, dact[10..11] / raw[10..11] on top 0.827 - 0.831, all decoder layers 0.862)
exec "$(dirname "$0")/sweep_env.sh" FAVAE_DACT11_BF16 "0" "$1" 1
