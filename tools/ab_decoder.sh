#!/bin/bash
# A/B timing of two library builds on ONE box (box-to-box spread is +-2.5 %): tools/ab_decoder.sh <libA> <libB> [B ...]
A=$1; Bq=$2; shift 2
for rep in 1 2; do for L in $A $Bq; do for B in "$@"; do
  MMT_AMD_LIB=$L timeout -k 10 100 python tools/time_decoder.py $B cluster_pipeline 2>&1 | grep -o "B=[0-9]*\|decoder_ms.: [0-9.]*" | tr "\n" " "; echo $(basename $L)
done; done; done
