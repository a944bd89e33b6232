#!/bin/bash
# library variants for kernel experiments: tools/variants/<name>.so built with extra -D flags
#   bash tools/build_variants.sh name1:"-DJTK_ENC_G=1" name2:"-DJTK_ENC_WAVES=8 -DJTK_ENC_G=2" ...
set -e
mkdir -p tools/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  make -s -C jtokkit_amd/csrc OUT=../../tools/variants/$name.so EXTRA="$flags" &
done
wait
ls -la tools/variants
