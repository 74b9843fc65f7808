#!/bin/bash
# occupancy sweep of seed+verify for the 250-bp (16-word) and 300-bp (20-word) classes
for L in 250 300; do for pad in 0 28000 50000; do
  echo "read_len=$L pad=$pad"
  GF_SV_PAD_LDS=$pad bash tools/ab_variants.sh "libgfmatch.so:0" --read-len $L --pairs 5000000
done; done
